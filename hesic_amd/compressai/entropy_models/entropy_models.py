"""Entropy models of the HESIC path (reference: compressai/entropy_models/entropy_models.py).

Same classes, constructor arguments, parameter / buffer names and exceptions as the reference.  The
``forward`` of each model is ONE fused HIP kernel (quantise + likelihood, csrc/entropy.hip) with a
hand-written backward; ``update`` / ``compress`` / ``decompress`` are host-side table building and
range-ANS coding (C++: csrc/host/hesic_host.cpp) exactly as in the reference -- they are not on the
throughput path.
"""
import numpy as np
import scipy.stats
import torch
import torch.nn as nn
import torch.nn.functional as F

from compressai._CXX import pmf_to_quantized_cdf as _pmf_to_quantized_cdf
from compressai.ops import LowerBound
from hesic_amd import functional as Fn


class _EntropyCoder:
    """Proxy to the actual entropy coder ('ans' = C++ rANS; 'rangecoder' if the PyPI module exists)."""

    def __init__(self, method):
        if not isinstance(method, str):
            raise ValueError(f'Invalid method type "{type(method)}"')
        from compressai import available_entropy_coders
        if method not in available_entropy_coders():
            methods = ", ".join(available_entropy_coders())
            raise ValueError(f'Unknown entropy coder "{method}" (available: {methods})')
        if method == "ans":
            from compressai import ans
            self._encoder, self._decoder = ans.RansEncoder(), ans.RansDecoder()
        else:
            import range_coder
            self._encoder, self._decoder = range_coder.RangeEncoder(), range_coder.RangeDecoder()

    def encode_with_indexes(self, *args, **kwargs):
        return self._encoder.encode_with_indexes(*args, **kwargs)

    def decode_with_indexes(self, *args, **kwargs):
        return self._decoder.decode_with_indexes(*args, **kwargs)


def default_entropy_coder():
    from compressai import get_entropy_coder
    return get_entropy_coder()


def pmf_to_quantized_cdf(pmf, precision=16):
    return torch.IntTensor(_pmf_to_quantized_cdf(pmf.tolist(), precision))


class EntropyModel(nn.Module):
    r"""Entropy model base class (reference :56-239)."""

    def __init__(self, likelihood_bound=1e-9, entropy_coder=None, entropy_coder_precision=16):
        super().__init__()
        if entropy_coder is None:
            entropy_coder = default_entropy_coder()
        self.entropy_coder = _EntropyCoder(entropy_coder)
        self.entropy_coder_precision = int(entropy_coder_precision)
        self.likelihood_bound = float(likelihood_bound)
        self.use_likelihood_bound = likelihood_bound > 0
        if self.use_likelihood_bound:
            self.likelihood_lower_bound = LowerBound(likelihood_bound)
        # filled by update()
        self.register_buffer("_offset", torch.IntTensor())
        self.register_buffer("_quantized_cdf", torch.IntTensor())
        self.register_buffer("_cdf_length", torch.IntTensor())

    def forward(self, *args):
        raise NotImplementedError()

    @staticmethod
    def _noise_like(x):
        # fresh U(-1/2, 1/2) per call: no shared mutable cache (the reference's _get_noise_cached, :88-96,
        # makes modules non re-entrant)
        return torch.empty_like(x).uniform_(-0.5, 0.5)

    def _quantize(self, inputs, mode, means=None):
        """'noise': x + U(-1/2,1/2); 'dequantize': round(x-mu)+mu; 'symbols': int32 round(x-mu)."""
        if mode not in ("noise", "dequantize", "symbols"):
            raise ValueError(f'Invalid quantization mode: "{mode}"')
        if mode == "noise":
            return inputs + self._noise_like(inputs)
        if inputs.is_cuda and inputs.dim() == 4 and inputs.dtype in (torch.float32, torch.bfloat16):
            if mode == "symbols":
                return Fn.quantize_symbols(inputs, means)
            if means is None:
                # round-half-even of the stored value: one elementwise kernel (the conditional kernel below would also evaluate a
                # likelihood nobody reads, on a ones / zeros scale and mean map filled just for it)
                return torch.round(inputs.detach())
            sc = torch.ones_like(inputs)
            out, _ = Fn.gaussian_conditional(inputs.detach(), sc, None if means is None else means.detach().expand_as(inputs))
            return out
        # host / odd-shaped tensors (update(), tests of the error paths): plain tensor ops
        outputs = inputs.clone()
        if means is not None:
            outputs -= means
        outputs = torch.round(outputs)
        if mode == "dequantize":
            if means is not None:
                outputs += means
            return outputs
        return outputs.int()

    @staticmethod
    def _dequantize(inputs, means=None):
        if means is not None:
            outputs = inputs.type_as(means)
            outputs += means
        else:
            outputs = inputs.float()
        return outputs

    def _pmf_to_cdf(self, pmf, tail_mass, pmf_length, max_length):
        cdf = torch.zeros((len(pmf_length), max_length + 2), dtype=torch.int32)
        for i, p in enumerate(pmf):
            prob = torch.cat((p[:pmf_length[i]], tail_mass[i]), dim=0)
            _cdf = pmf_to_quantized_cdf(prob, self.entropy_coder_precision)
            cdf[i, :_cdf.size(0)] = _cdf
        return cdf

    def _check_cdf_size(self):
        if self._quantized_cdf.numel() == 0:
            raise ValueError("Uninitialized CDFs. Run update() first")
        if len(self._quantized_cdf.size()) != 2:
            raise ValueError(f"Invalid CDF size {self._quantized_cdf.size()}")

    def _check_offsets_size(self):
        if self._offset.numel() == 0:
            raise ValueError("Uninitialized offsets. Run update() first")
        if len(self._offset.size()) != 1:
            raise ValueError(f"Invalid offsets size {self._offset.size()}")

    def _check_cdf_length(self):
        if self._cdf_length.numel() == 0:
            raise ValueError("Uninitialized CDF lengths. Run update() first")
        if len(self._cdf_length.size()) != 1:
            raise ValueError(f"Invalid offsets size {self._cdf_length.size()}")

    def compress(self, inputs, indexes, means=None):
        """Tensors -> list of byte strings (one per batch element), reference :165-196."""
        symbols = self._quantize(inputs, "symbols", means)
        if len(inputs.size()) != 4:
            raise ValueError("Invalid `inputs` size. Expected a 4-D tensor.")
        if inputs.size() != indexes.size():
            raise ValueError("`inputs` and `indexes` should have the same size.")
        self._check_cdf_size()
        self._check_cdf_length()
        self._check_offsets_size()
        cdf = self._quantized_cdf.cpu().tolist()
        lengths = self._cdf_length.reshape(-1).int().cpu().tolist()
        offsets = self._offset.reshape(-1).int().cpu().tolist()
        symbols, indexes = symbols.cpu(), indexes.cpu()
        return [self.entropy_coder.encode_with_indexes(symbols[i].reshape(-1).int().tolist(),
                                                       indexes[i].reshape(-1).int().tolist(), cdf, lengths, offsets)
                for i in range(symbols.size(0))]

    def decompress(self, strings, indexes, means=None):
        """List of byte strings -> tensor, reference :199-239."""
        if not isinstance(strings, (tuple, list)):
            raise ValueError("Invalid `strings` parameter type.")
        if not len(strings) == indexes.size(0):
            raise ValueError("Invalid strings or indexes parameters")
        if len(indexes.size()) != 4:
            raise ValueError("Invalid `indexes` size. Expected a 4-D tensor.")
        self._check_cdf_size()
        self._check_cdf_length()
        self._check_offsets_size()
        if means is not None:
            if means.size()[:-2] != indexes.size()[:-2]:
                raise ValueError("Invalid means or indexes parameters")
            if means.size() != indexes.size() and (means.size(2) != 1 or means.size(3) != 1):
                raise ValueError("Invalid means parameters")
        cdf = self._quantized_cdf.cpu().tolist()
        lengths = self._cdf_length.reshape(-1).int().cpu().tolist()
        offsets = self._offset.reshape(-1).int().cpu().tolist()
        outputs = torch.empty(indexes.size(), dtype=torch.int32)
        for i, s in enumerate(strings):
            values = self.entropy_coder.decode_with_indexes(s, indexes[i].reshape(-1).int().cpu().tolist(), cdf, lengths, offsets)
            outputs[i] = torch.tensor(values, dtype=torch.int32).reshape(outputs[i].size())
        dev = self._quantized_cdf.device
        return self._dequantize(outputs.to(dev), None if means is None else means.to(dev))


class EntropyBottleneck(EntropyModel):
    r"""Factorised entropy bottleneck (Balle et al. 2018), reference :242-430.

    Per channel a 1-3-3-3-3-1 cumulative-logit network; likelihood of z_hat is
    |sigmoid(s*c(z_hat+1/2)) - sigmoid(s*c(z_hat-1/2))|.  ``forward`` = one HIP kernel."""

    def __init__(self, channels, *args, tail_mass=1e-9, init_scale=10, filters=(3, 3, 3, 3), **kwargs):
        super().__init__(*args, **kwargs)
        self.channels = int(channels)
        self.filters = tuple(int(f) for f in filters)
        self.init_scale = float(init_scale)
        self.tail_mass = float(tail_mass)

        self._biases = nn.ParameterList()
        self._factors = nn.ParameterList()
        self._matrices = nn.ParameterList()
        widths = (1,) + self.filters + (1,)
        scale = self.init_scale ** (1 / (len(self.filters) + 1))
        for i in range(len(self.filters) + 1):
            init = np.log(np.expm1(1 / scale / widths[i + 1]))
            self._matrices.append(nn.Parameter(torch.full((self.channels, widths[i + 1], widths[i]), float(init))))
            self._biases.append(nn.Parameter(torch.empty(self.channels, widths[i + 1], 1).uniform_(-0.5, 0.5)))
            if i < len(self.filters):
                self._factors.append(nn.Parameter(torch.zeros(self.channels, widths[i + 1], 1)))
        self.quantiles = nn.Parameter(torch.tensor([-self.init_scale, 0.0, self.init_scale]).repeat(self.channels, 1, 1))
        target = np.log(2 / self.tail_mass - 1)
        self.register_buffer("target", torch.Tensor([-target, 0, target]))

    def _medians(self):
        return self.quantiles[:, :, 1:2]

    def _logits_cumulative(self, inputs, stop_gradient):
        """(C,1,L) -> (C,1,L) cumulative logits; tensor-op form used only by update() and loss()
        (3*C values); the per-latent evaluation is fused in csrc/entropy.hip."""
        logits = inputs
        for i in range(len(self.filters) + 1):
            matrix, bias = self._matrices[i], self._biases[i]
            if stop_gradient:
                matrix, bias = matrix.detach(), bias.detach()
            logits = torch.matmul(F.softplus(matrix), logits) + bias
            if i < len(self._factors):
                factor = self._factors[i].detach() if stop_gradient else self._factors[i]
                logits = logits + torch.tanh(factor) * torch.tanh(logits)
        return logits

    def loss(self):
        logits = self._logits_cumulative(self.quantiles, stop_gradient=True)
        return torch.abs(logits - self.target).sum()

    def update(self, force=False):
        if self._offset.numel() > 0 and not force:
            return
        with torch.no_grad():
            medians = self.quantiles[:, 0, 1]
            minima = torch.clamp(torch.ceil(medians - self.quantiles[:, 0, 0]).int(), min=0)
            maxima = torch.clamp(torch.ceil(self.quantiles[:, 0, 2] - medians).int(), min=0)
            self._offset = -minima
            pmf_start = medians - minima
            pmf_length = maxima + minima + 1
            max_length = int(pmf_length.max())
            samples = torch.arange(max_length, device=medians.device)[None, :] + pmf_start[:, None, None]
            lower = self._logits_cumulative(samples - 0.5, stop_gradient=True)
            upper = self._logits_cumulative(samples + 0.5, stop_gradient=True)
            sign = -torch.sign(lower + upper)
            pmf = torch.abs(torch.sigmoid(sign * upper) - torch.sigmoid(sign * lower))[:, 0, :]
            tail_mass = torch.sigmoid(lower[:, 0, :1]) + torch.sigmoid(-upper[:, 0, -1:])
            quantized_cdf = self._pmf_to_cdf(pmf.cpu(), tail_mass.cpu(), pmf_length.cpu(), max_length)
            self._quantized_cdf = quantized_cdf.to(medians.device)
            self._cdf_length = (pmf_length + 2).int()

    def forward(self, x):
        if len(self.filters) != 4 or any(f != 3 for f in self.filters):
            raise NotImplementedError("hesic_amd EntropyBottleneck kernel is specialised for filters=(3,3,3,3)")
        noise = self._noise_like(x) if self.training else None
        return self.forward_with_noise(x, noise)

    def forward_with_noise(self, x, noise):
        """noise=None: eval (round(x-med)+med); else x+noise (parity tests inject the draw)."""
        if not hasattr(self, "_eb_packer"):
            self._eb_packer = Fn.PackedEb()
        return Fn.entropy_bottleneck(x, list(self._matrices), list(self._biases), list(self._factors), self.quantiles, noise,
                                     packer=self._eb_packer)

    @staticmethod
    def _build_indexes(size):
        N, C, H, W = size
        return torch.arange(C).view(1, -1, 1, 1).int().repeat(N, 1, H, W)

    def compress(self, x):
        indexes = self._build_indexes(x.size())
        medians = self._medians().detach().view(1, -1, 1, 1)
        return super().compress(x, indexes, medians)

    def decompress(self, strings, size):
        output_size = (len(strings), self._quantized_cdf.size(0), size[0], size[1])
        indexes = self._build_indexes(output_size)
        medians = self._medians().detach().view(1, -1, 1, 1)
        return super().decompress(strings, indexes, medians)


class _GaussianBase(EntropyModel):
    """scale table / bound handling shared by the two conditional models."""

    def _init_scales(self, scale_table, scale_bound, tail_mass):
        if scale_table and (scale_table != sorted(scale_table) or any(s <= 0 for s in scale_table)):
            raise ValueError(f'Invalid scale_table "({scale_table})"')
        self.register_buffer("scale_table", self._prepare_scale_table(scale_table) if scale_table else torch.Tensor())
        self.register_buffer("scale_bound", torch.Tensor([float(scale_bound)]) if scale_bound is not None else None)
        self.tail_mass = float(tail_mass)
        if scale_bound is None and scale_table:
            self.lower_bound_scale = LowerBound(self.scale_table[0])
        elif scale_bound is not None and scale_bound > 0:
            self.lower_bound_scale = LowerBound(scale_bound)
        else:
            raise ValueError("Invalid parameters")

    @staticmethod
    def _prepare_scale_table(scale_table):
        return torch.Tensor(tuple(float(s) for s in scale_table))

    def _standardized_cumulative(self, inputs):
        return 0.5 * torch.erfc(-(2 ** -0.5) * inputs)

    @staticmethod
    def _standardized_quantile(quantile):
        return scipy.stats.norm.ppf(quantile)

    def update_scale_table(self, scale_table, force=False):
        if self._offset.numel() > 0 and not force:
            return
        self.scale_table = self._prepare_scale_table(scale_table)
        self.update()

    def update(self):
        multiplier = -self._standardized_quantile(self.tail_mass / 2)
        pmf_center = torch.ceil(self.scale_table * multiplier).int()
        pmf_length = 2 * pmf_center + 1
        max_length = torch.max(pmf_length).item()
        samples = torch.abs(torch.arange(max_length, device=pmf_center.device).int() - pmf_center[:, None]).float()
        samples_scale = self.scale_table.unsqueeze(1).float()
        upper = self._standardized_cumulative((.5 - samples) / samples_scale)
        lower = self._standardized_cumulative((-.5 - samples) / samples_scale)
        pmf = upper - lower
        tail_mass = 2 * lower[:, :1]
        quantized_cdf = self._pmf_to_cdf(pmf.cpu(), tail_mass.cpu(), pmf_length.cpu(), max_length)
        self._quantized_cdf = quantized_cdf.to(pmf_center.device)
        self._offset = -pmf_center
        self._cdf_length = pmf_length + 2

    def build_indexes(self, scales):
        scales = self.lower_bound_scale(scales)
        indexes = scales.new_full(scales.size(), len(self.scale_table) - 1).int()
        for s in self.scale_table[:-1]:
            indexes -= (scales <= s).int()
        return indexes

    def _bound(self):
        return self.lower_bound_scale.value()


class GaussianConditional(_GaussianBase):
    r"""Gaussian conditional layer (reference :433-562): y_hat = round(y-mu)+mu (eval) or y+U (train),
    likelihood = Phi((1/2-|y_hat-mu|)/s) - Phi((-1/2-|y_hat-mu|)/s), s = max(scale, bound)."""

    def __init__(self, scale_table, *args, scale_bound=0.11, tail_mass=1e-9, **kwargs):
        super().__init__(*args, **kwargs)
        if not isinstance(scale_table, (type(None), list, tuple)):
            raise ValueError(f'Invalid type for scale_table "{type(scale_table)}"')
        if isinstance(scale_table, (list, tuple)) and len(scale_table) < 1:
            raise ValueError(f'Invalid scale_table length "{len(scale_table)}"')
        self._init_scales(scale_table, scale_bound, tail_mass)

    def _likelihood(self, inputs, scales, means=None):
        """Stand-alone tensor-op form (reference :528-544) for host code; forward() uses the fused kernel."""
        values = torch.abs(inputs - means if means is not None else inputs)
        scales = self.lower_bound_scale(scales)
        return self._standardized_cumulative((.5 - values) / scales) - self._standardized_cumulative((-.5 - values) / scales)

    def forward(self, inputs, scales, means=None, noise=None):
        if self.training and noise is None:
            noise = self._noise_like(inputs)
        lb = self.likelihood_bound if self.use_likelihood_bound else 0.0
        return Fn.gaussian_conditional(inputs, scales, means, noise=noise if self.training else None,
                                       scale_bound=self._bound(), lik_bound=lb)


class GaussianMixtureConditional(_GaussianBase):
    r"""K-component Gaussian mixture (the HESIC addition, reference :565-710): quantisation ignores the
    means; likelihood = sum_k w[:, kM:(k+1)M] * (Phi(u_k) - Phi(l_k)); scales/means/weights carry K*M
    channels with channel index k*M+m."""

    def __init__(self, K, scale_table=None, mean_table=None, weight_table=None, *args, scale_bound=0.11,
                 tail_mass=1e-9, **kwargs):
        super().__init__(*args, **kwargs)
        self.K = K
        self._init_scales(scale_table, scale_bound, tail_mass)

    def _likelihood(self, inputs, scales, means=None, weights=None):
        M = inputs.size()[1]
        likelihood = None
        for k in range(self.K):
            sl = slice(M * k, M * (k + 1))
            v = torch.abs(inputs - means[:, sl])
            s = self.lower_bound_scale(scales[:, sl])
            term = (self._standardized_cumulative((.5 - v) / s) - self._standardized_cumulative((-.5 - v) / s)) * weights[:, sl]
            likelihood = term if likelihood is None else likelihood + term
        return likelihood

    def forward(self, inputs, scales, means=None, weights=None, noise=None):
        if self.training and noise is None:
            noise = self._noise_like(inputs)
        lb = self.likelihood_bound if self.use_likelihood_bound else 0.0
        return Fn.gaussian_mixture(inputs, scales, means, weights, self.K, noise=noise if self.training else None,
                                   scale_bound=self._bound(), lik_bound=lb)
