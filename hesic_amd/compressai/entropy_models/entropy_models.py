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
from hesic_amd import handover as _ho


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
    """1-D pmf tensor -> int32 tensor of len(pmf) + 1 cumulative counts summing to 2**precision."""
    return torch.tensor(_pmf_to_quantized_cdf(pmf.tolist(), precision), dtype=torch.int32)


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
        if _ho.is_carrier(inputs):
            # inference hand-over (hesic_amd/handover.py): the latent comes from the conv's fp32 accumulators and is rounded in fp32, its
            # integer values stored in the 16-bit format for the convs that read them (models._round_latent)
            if mode == "dequantize" and means is None and Fn.fp32_latents():
                return Fn.round_to(inputs.f32(), Fn.compute_dtype())
            inputs = inputs.plain()
        means = _ho.plain(means) if means is not None else None
        if mode == "noise":
            return inputs + self._noise_like(inputs)
        if inputs.is_cuda and inputs.dim() == 4 and inputs.dtype in (torch.float32, torch.bfloat16, torch.float16):
            if mode == "symbols":
                return Fn.quantize_symbols(inputs, means)
            if means is None:
                # round-half-even of the stored value: one elementwise kernel (the conditional kernel below would also evaluate a
                # likelihood nobody reads, on a ones / zeros scale and mean map filled just for it)
                return Fn.round_to(inputs, inputs.dtype)            # hesic_round: round-half-even, one launch
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
        """int32 symbols -> values: + means in the means' dtype, else plain fp32 (reference :127-134)."""
        return inputs.float() if means is None else inputs.to(means.dtype) + means

    def _pmf_to_cdf(self, pmf, tail_mass, pmf_length, max_length):
        """All rows' quantised CDFs from the pmf matrix in one native call (``hesic_pmf_rows_to_quantized_cdfs``): row i codes
        its first ``pmf_length[i]`` bins + the tail-mass escape bin; (rows, max_length + 2) int32 (reference :136-142)."""
        from hesic_amd import _host
        table = _host.quantized_cdf_rows(pmf.detach().cpu().numpy(), pmf_length.detach().cpu().numpy(),
                                         tail_mass.detach().cpu().numpy(), self.entropy_coder_precision, int(max_length) + 2)
        return torch.from_numpy(table)

    # (buffer, what it is called in the messages): filled by update(), checked before every coding call
    _TABLES = (("_quantized_cdf", "CDFs", "CDF", 2), ("_cdf_length", "CDF lengths", "offsets", 1), ("_offset", "offsets", "offsets", 1))

    def _require_tables(self, only=None):
        for name, plural, noun, ndim in self._TABLES:
            if only is not None and name != only:
                continue
            buf = getattr(self, name)
            if buf.numel() == 0:
                raise ValueError(f"Uninitialized {plural}. Run update() first")
            if buf.dim() != ndim:
                raise ValueError(f"Invalid {noun} size {buf.size()}")

    def _check_cdf_size(self):
        self._require_tables("_quantized_cdf")

    def _check_offsets_size(self):
        self._require_tables("_offset")

    def _check_cdf_length(self):
        self._require_tables("_cdf_length")

    def _coding_tables(self):
        """(cdf table, lengths, offsets) as int32 numpy arrays for the native coder."""
        self._require_tables()
        as_np = lambda t: t.detach().reshape(t.shape[0], -1).int().cpu().numpy()
        return as_np(self._quantized_cdf), as_np(self._cdf_length).reshape(-1), as_np(self._offset).reshape(-1)

    def _native_coder(self):
        from compressai import ans
        return isinstance(self.entropy_coder._encoder, ans.RansEncoder)

    def compress(self, inputs, indexes, means=None):
        """Tensors -> one byte string per batch element (reference :165-196).  Symbols, indexes and tables go to the
        native rANS coder as int32 buffers."""
        symbols = self._quantize(inputs, "symbols", means)
        if inputs.dim() != 4:
            raise ValueError("Invalid `inputs` size. Expected a 4-D tensor.")
        if inputs.size() != indexes.size():
            raise ValueError("`inputs` and `indexes` should have the same size.")
        table, lengths, offsets = self._coding_tables()
        sym = symbols.detach().int().cpu().numpy().reshape(symbols.shape[0], -1)
        idx = indexes.detach().int().cpu().numpy().reshape(indexes.shape[0], -1)
        if self._native_coder():
            from hesic_amd import _host
            return [_host.rans_encode_arrays(s_, i_, table, lengths, offsets) for s_, i_ in zip(sym, idx)]
        rows = table.tolist()
        return [self.entropy_coder.encode_with_indexes(s_.tolist(), i_.tolist(), rows, lengths.tolist(), offsets.tolist())
                for s_, i_ in zip(sym, idx)]

    def decompress(self, strings, indexes, means=None):
        """Byte strings -> tensor (reference :199-239)."""
        if not isinstance(strings, (tuple, list)):
            raise ValueError("Invalid `strings` parameter type.")
        if len(strings) != indexes.size(0):
            raise ValueError("Invalid strings or indexes parameters")
        if indexes.dim() != 4:
            raise ValueError("Invalid `indexes` size. Expected a 4-D tensor.")
        table, lengths, offsets = self._coding_tables()
        if means is not None:
            if means.size()[:-2] != indexes.size()[:-2]:
                raise ValueError("Invalid means or indexes parameters")
            if means.size() != indexes.size() and (means.size(2) != 1 or means.size(3) != 1):
                raise ValueError("Invalid means parameters")
        idx = indexes.detach().int().cpu().numpy().reshape(indexes.shape[0], -1)
        if self._native_coder():
            from hesic_amd import _host
            decoded = [_host.rans_decode_arrays(s_, i_, table, lengths, offsets) for s_, i_ in zip(strings, idx)]
        else:
            rows = table.tolist()
            decoded = [np.asarray(self.entropy_coder.decode_with_indexes(s_, i_.tolist(), rows, lengths.tolist(), offsets.tolist()),
                                  dtype=np.int32) for s_, i_ in zip(strings, idx)]
        dev = self._quantized_cdf.device
        symbols = torch.from_numpy(np.stack(decoded)).reshape(indexes.size()).to(dev)
        return self._dequantize(symbols, None if means is None else means.to(dev))


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
        if self.quantiles.is_cuda and self.filters == (3, 3, 3, 3) and self.quantiles.dtype == torch.float32:
            # forward + quantile gradient in one HIP launch (hesic_eb_aux_loss)
            return Fn.eb_aux_loss(list(self._matrices), list(self._biases), list(self._factors), self.quantiles, self.tail_mass)
        logits = self._logits_cumulative(self.quantiles, stop_gradient=True)
        return torch.abs(logits - self.target).sum()

    def update(self, force=False):
        """Fill ``_offset`` / ``_quantized_cdf`` / ``_cdf_length`` from the learned quantiles (reference :302-343): per channel the
        support is [median - ceil(median - q_lo), median + ceil(q_hi - median)], the pmf is the model's likelihood at those
        integer offsets from the median and the mass outside is the escape bin."""
        if self._offset.numel() > 0 and not force:
            return
        with torch.no_grad():
            q_lo, med, q_hi = self.quantiles[:, 0, 0], self.quantiles[:, 0, 1], self.quantiles[:, 0, 2]
            below = torch.ceil(med - q_lo).clamp_(min=0).int()
            above = torch.ceil(q_hi - med).clamp_(min=0).int()
            support = below + above + 1
            width = int(support.max())
            grid = (med - below)[:, None, None] + torch.arange(width, device=med.device)[None, None, :]    # (C, 1, width)
            lo = self._logits_cumulative(grid - 0.5, stop_gradient=True)
            hi = self._logits_cumulative(grid + 0.5, stop_gradient=True)
            flip = -torch.sign(lo + hi)
            pmf = (torch.sigmoid(flip * hi) - torch.sigmoid(flip * lo)).abs()[:, 0, :]
            outside = torch.sigmoid(lo[:, 0, :1]) + torch.sigmoid(-hi[:, 0, -1:])
            self._quantized_cdf = self._pmf_to_cdf(pmf, outside, support, width).to(med.device)
            self._offset = -below
            self._cdf_length = (support + 2).int()

    def forward(self, x):
        if len(self.filters) != 4 or any(f != 3 for f in self.filters):
            raise NotImplementedError("hesic_amd EntropyBottleneck kernel is specialised for filters=(3,3,3,3)")
        if _ho.active(x) and not self.training:
            # inference hand-over: z from the producing conv's fp32 accumulators; z_hat goes on as a Carrier so that the bilinear
            # up-sampling + cat in front of gmm_hyper_y2 (newnet1.py:556-557) can be recorded
            z = _ho.f32_of(x) if Fn.fp32_latents() else _ho.plain(x)
            od = Fn.compute_dtype() if (z.dtype == torch.float32 and Fn.fp32_latents()) else None
            z_hat, lik = self.forward_with_noise(z, None, out_dtype=od)
            return _ho.value(z_hat), lik
        x = _ho.plain(x)
        noise = self._noise_like(x) if self.training else None
        return self.forward_with_noise(x, noise)

    def forward_with_noise(self, x, noise, out_dtype=None):
        """noise=None: eval (round(x-med)+med); else x+noise (parity tests inject the draw).  ``out_dtype``: storage of the
        returned z_hat when x is an fp32 latent of the bf16 mode (inference)."""
        if not hasattr(self, "_eb_packer"):
            self._eb_packer = Fn.PackedEb()
        return Fn.entropy_bottleneck(x, list(self._matrices), list(self._biases), list(self._factors), self.quantiles, noise,
                                     packer=self._eb_packer, out_dtype=out_dtype,
                                     lik_bound=self.likelihood_bound if self.use_likelihood_bound else 0.0)

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
        """Tables for the scale levels (reference :494-514): level s gets the zero-mean Gaussian pmf on
        [-c, c], c = ceil(s * Phi^-1(1 - tail_mass / 2)), plus the two-sided tail as escape bin."""
        reach = -self._standardized_quantile(self.tail_mass / 2)
        table = self.scale_table.float()
        half = torch.ceil(table * reach).int()
        support = 2 * half + 1
        width = int(support.max())
        dist = (torch.arange(width, device=table.device).int()[None, :] - half[:, None]).abs().float()
        sigma = table[:, None]
        hi = self._standardized_cumulative((0.5 - dist) / sigma)
        lo = self._standardized_cumulative((-0.5 - dist) / sigma)
        self._quantized_cdf = self._pmf_to_cdf(hi - lo, 2 * lo[:, :1], support, width).to(table.device)
        self._offset = -half
        self._cdf_length = support + 2

    def build_indexes(self, scales):
        """Index of the first table level >= scale (the last level for anything larger), reference :516-521."""
        levels = self.scale_table[:-1].to(scales.device, scales.dtype).contiguous()
        return torch.bucketize(self.lower_bound_scale(scales).contiguous(), levels).int()

    def _bound(self):
        return self.lower_bound_scale.value()


def _handed_over(model, inputs, scales, means, out_dtype):
    """Inference hand-over (hesic_amd/handover.py): latents and parameter maps that arrive as Carriers are taken from their producers' fp32
    accumulators, and y_hat is stored in the 16-bit format (``out_dtype``) -- what ``models.HSIC._forward_eval`` asks for explicitly."""
    if not any(_ho.is_carrier(t) for t in (inputs, scales, means)):
        return inputs, scales, means, out_dtype
    if not model.training and Fn.fp32_latents():
        inputs, scales = _ho.f32_of(inputs), _ho.f32_of(scales)
        means = _ho.f32_of(means) if means is not None else None
        if out_dtype is None and inputs.dtype == torch.float32:
            out_dtype = Fn.compute_dtype()
        return inputs, scales, means, out_dtype
    return _ho.plain(inputs), _ho.plain(scales), (_ho.plain(means) if means is not None else None), out_dtype


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

    def forward(self, inputs, scales, means=None, noise=None, out_dtype=None):
        inputs, scales, means, out_dtype = _handed_over(self, inputs, scales, means, out_dtype)
        if self.training and noise is None:
            noise = self._noise_like(inputs)
        lb = self.likelihood_bound if self.use_likelihood_bound else 0.0
        return Fn.gaussian_conditional(inputs, scales, means, noise=noise if self.training else None,
                                       scale_bound=self._bound(), lik_bound=lb, out_dtype=out_dtype)


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

    def forward(self, inputs, scales, means=None, weights=None, noise=None, out_dtype=None):
        inputs, scales, means, out_dtype = _handed_over(self, inputs, scales, means, out_dtype)
        weights = _ho.plain(weights) if weights is not None else None
        if self.training and noise is None:
            noise = self._noise_like(inputs)
        lb = self.likelihood_bound if self.use_likelihood_bound else 0.0
        return Fn.gaussian_mixture(inputs, scales, means, weights, self.K, noise=noise if self.training else None,
                                   scale_bound=self._bound(), lik_bound=lb, out_dtype=out_dtype)
