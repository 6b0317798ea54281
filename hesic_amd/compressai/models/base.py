"""CompressionModel base (reference: compressai/models/priors.py:36-103).  The reference's single-image
zoo (FactorizedPrior ... Cheng2020) is not on the stereo path and is not rebuilt;
the stereo models live in ``hesic_amd.models``."""
import torch.nn as nn

from compressai.entropy_models import EntropyBottleneck


class CompressionModel(nn.Module):
    """Auto-encoder base with one entropy bottleneck; ``parameters()`` skips the bottleneck (it is trained by
    the auxiliary optimiser through ``aux_parameters()``)."""

    def __init__(self, entropy_bottleneck_channels, init_weights=True):
        super().__init__()
        self.entropy_bottleneck = EntropyBottleneck(entropy_bottleneck_channels)
        if init_weights:
            self._initialize_weights()

    def aux_loss(self):
        return sum(m.loss() for m in self.modules() if isinstance(m, EntropyBottleneck))

    def _initialize_weights(self):
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
                nn.init.kaiming_normal_(m.weight)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    def forward(self, *args):
        raise NotImplementedError()

    def parameters(self):
        for m in self.children():
            if not isinstance(m, EntropyBottleneck):
                yield from m.parameters()

    def aux_parameters(self):
        for m in self.children():
            if isinstance(m, EntropyBottleneck):
                yield from m.parameters()

    def update(self, force=False):
        for m in self.children():
            if isinstance(m, EntropyBottleneck):
                m.update(force=force)
