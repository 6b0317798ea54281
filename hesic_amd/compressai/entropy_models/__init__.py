from .entropy_models import EntropyBottleneck, EntropyModel, GaussianConditional, GaussianMixtureConditional

__all__ = ["EntropyModel", "EntropyBottleneck", "GaussianConditional", "GaussianMixtureConditional"]
