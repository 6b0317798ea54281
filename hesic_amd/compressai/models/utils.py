"""conv()/deconv() factories and state-dict helpers (reference: compressai/models/utils.py).

``conv`` / ``deconv`` are *the* hook where the HIP convolutions are substituted: they
return real ``nn.Conv2d`` / ``nn.ConvTranspose2d`` subclasses (same ``.weight`` / ``.bias``, same
state-dict keys, ``isinstance`` checks and ``nn.Sequential`` indexing keep working) whose forward
launches the implicit-GEMM MFMA kernel (csrc/conv_igemm.hip) or, for the 3-channel image side, the
strided VALU kernels (csrc/sconv.hip).
"""
import torch
import torch.nn as nn

from hesic_amd import functional as Fn
from hesic_amd import handover as _ho
from hesic_amd import _lib as L


class HipConv2d(nn.Conv2d):
    """nn.Conv2d (square kernel, stride 1|2, padding k//2, no dilation/groups) on the HIP path."""

    def _check(self):
        k, s, p = self.kernel_size, self.stride, self.padding
        if k[0] != k[1] or s[0] != s[1] or p[0] != p[1] or self.dilation != (1, 1) or self.groups != 1 \
                or self.padding_mode != "zeros":
            raise NotImplementedError("hesic_amd conv: square kernel, equal strides, zero padding, no dilation/groups")

    def run(self, x, act=L.ACT_NONE, in_abs=False, mask=None, tap_mask=0):
        self._check()
        if not hasattr(self, "_packer"):
            self._packer = Fn.PackedWeight(shaped=getattr(self, "shaped_weights", False))
        return Fn.conv2d(x, self.weight, self.bias, kernel_size=self.kernel_size[0], stride=self.stride[0],
                         padding=self.padding[0], transposed=False, act=act, in_abs=in_abs, packer=self._packer,
                         mask=mask, tap_mask=tap_mask)

    def run_into(self, x, out, c_off, act=L.ACT_NONE, mask=None, tap_mask=0):
        """out[:, c_off:c_off + out_channels] = act(self(x)) at inference, written in place (no cat afterwards)."""
        self._check()
        if not hasattr(self, "_packer"):
            self._packer = Fn.PackedWeight(shaped=getattr(self, "shaped_weights", False))
        return Fn.conv2d_into(x, self.weight, self.bias, out, c_off, kernel_size=self.kernel_size[0], stride=self.stride[0],
                              padding=self.padding[0], transposed=False, act=act, packer=self._packer, mask=mask, tap_mask=tap_mask)

    def run_slice(self, x, c_off, act=L.ACT_NONE):
        """self(x[:, c_off:c_off + in_channels]) at inference, reading the channel slice in place."""
        self._check()
        if not hasattr(self, "_packer"):
            self._packer = Fn.PackedWeight(shaped=getattr(self, "shaped_weights", False))
        return Fn.conv2d_slice(x, c_off, self.weight, self.bias, kernel_size=self.kernel_size[0], stride=self.stride[0],
                               padding=self.padding[0], transposed=False, act=act, packer=self._packer)

    def run_latent(self, x, act=L.ACT_NONE, in_abs=False, want_lo=True):
        """(lo, hi) of a conv that feeds an entropy model: ``hi`` is fp32 from the accumulators at bf16 inference
        (``Fn.conv2d_latent``), ``lo`` (optional) the storage-dtype copy for the next conv."""
        self._check()
        if not hasattr(self, "_packer"):
            self._packer = Fn.PackedWeight(shaped=getattr(self, "shaped_weights", False))
        return Fn.conv2d_latent(x, self.weight, self.bias, kernel_size=self.kernel_size[0], stride=self.stride[0],
                                padding=self.padding[0], transposed=False, act=act, in_abs=in_abs, packer=self._packer,
                                want_lo=want_lo)

    def run_hilo(self, x_hilo, act=L.ACT_NONE, out="hilo", out_abs=False, gdn=None, products=3):
        """self on a hi/lo map (the pair analysis modes, ``Fn.conv2d_hilo``): ``x_hilo`` is the (B, 2*Cin, H, W) tensor;
        ``gdn``: the GDN module behind the conv (fused hi/lo epilogue); ``products`` = 2: single error-feedback weights ("x3c2")."""
        self._check()
        if not hasattr(self, "_packer_hl"):
            self._packer_hl = Fn.PackedWeightHiLo()
        g = None
        if gdn is not None:
            if not hasattr(gdn, "_packer_lo"):
                gdn._packer_lo = Fn.PackedGdnLo()
            gp, bp = gdn.packer().get(gdn.beta, gdn.gamma, gdn.beta_min)
            g = (gp, gdn._packer_lo.get(gdn.gamma), bp, gdn.inverse)
        if products == 2:
            if not hasattr(self, "_packer_hl1"):
                self._packer_hl1 = Fn.PackedWeightHiLo()
            wp = self._packer_hl1.get(self.weight, single=True)
        else:
            wp = self._packer_hl.get(self.weight)
        return Fn.conv2d_hilo(x_hilo, wp, self.bias, self.weight.shape[1], self.weight.shape[0],
                              kernel_size=self.kernel_size[0], stride=self.stride[0], padding=self.padding[0], gdn=g, act=act, out=out,
                              out_abs=out_abs, products=products)

    def run_gdn_hilo_out(self, x, gdn):
        """gdn(self(x)) for the "x3c2" analysis mode: single 16-bit operands in the conv, the GDN on pairs, a hi/lo map out
        (``Fn.conv2d_gdn_hilo_out``)."""
        self._check()
        if not hasattr(self, "_packer"):
            self._packer = Fn.PackedWeight(shaped=getattr(self, "shaped_weights", False))
        if not hasattr(gdn, "_packer_lo"):
            gdn._packer_lo = Fn.PackedGdnLo()
        gp, bp = gdn.packer().get(gdn.beta, gdn.gamma, gdn.beta_min)
        cout, cin, kh, kw = self.weight.shape
        wp = self._packer.get(self.weight, None, cout, cin, kh, kw, False, False, x.dtype)
        return Fn.conv2d_gdn_hilo_out(x, wp, self.bias, cin, kernel_size=kh, stride=self.stride[0], padding=self.padding[0],
                                      gdn=(gp, gdn._packer_lo.get(gdn.gamma), bp, gdn.inverse))

    def run_gdn(self, x, gdn):
        """gdn(self(x)); one fused kernel when eligible (inference, bf16 storage, 128 channels), else two ops."""
        if Fn.conv2d_gdn_fusable(x, self.weight, gdn.beta.numel(), False) and (self.weight.shape[1] != 3 or self.stride[0] == 2):
            self._check()
            if not hasattr(self, "_packer"):
                self._packer = Fn.PackedWeight(shaped=getattr(self, "shaped_weights", False))
            return Fn.conv2d_gdn(x, self.weight, self.bias, gdn.beta, gdn.gamma, kernel_size=self.kernel_size[0],
                                 stride=self.stride[0], padding=self.padding[0], transposed=False, inverse=gdn.inverse,
                                 beta_min=gdn.beta_min, packer=self._packer, gdn_packer=gdn.packer())
        return Fn.gdn(self.run(x), gdn.beta, gdn.gamma, gdn.inverse, gdn.beta_min)

    def run_cat(self, xa, xb, gdn=None, gdn_on_input=False):
        """self(torch.cat((xa, xb), 1)) without materialising the cat where the kernels allow it (inference); ``gdn``: the
        3-channel GDN module behind the conv (or, ``gdn_on_input``, in front of it on ``xa``) rides on the same launch."""
        self._check()
        return Fn.conv2d_cat(xa, xb, self.weight, self.bias, kernel_size=self.kernel_size[0], stride=self.stride[0],
                             padding=self.padding[0], transposed=False, gdn=gdn, gdn_on_input=gdn_on_input)

    def forward(self, x):
        if _ho.active(x):
            return _ho.conv(self, x)          # inference: deferred until the next module of the package sees it (hesic_amd/handover.py)
        return self.run(x)


class HipConvTranspose2d(nn.ConvTranspose2d):
    """nn.ConvTranspose2d with output_padding = stride-1 (what ``deconv`` builds) on the HIP path."""

    def _check(self):
        k, s, p, op = self.kernel_size, self.stride, self.padding, self.output_padding
        if k[0] != k[1] or s[0] != s[1] or p[0] != p[1] or op != (s[0] - 1, s[0] - 1) or self.dilation != (1, 1) \
                or self.groups != 1:
            raise NotImplementedError("hesic_amd deconv: square kernel, output_padding = stride-1, no dilation/groups")

    def run(self, x, act=L.ACT_NONE):
        self._check()
        if not hasattr(self, "_packer"):
            self._packer = Fn.PackedWeight(shaped=getattr(self, "shaped_weights", False), tr_stride=self.stride[0])
        return Fn.conv2d(x, self.weight, self.bias, kernel_size=self.kernel_size[0], stride=self.stride[0],
                         padding=self.padding[0], transposed=True, act=act, packer=self._packer)

    def run_slice(self, x, c_off, act=L.ACT_NONE):
        self._check()
        if not hasattr(self, "_packer"):
            self._packer = Fn.PackedWeight(shaped=getattr(self, "shaped_weights", False), tr_stride=self.stride[0])
        return Fn.conv2d_slice(x, c_off, self.weight, self.bias, kernel_size=self.kernel_size[0], stride=self.stride[0],
                               padding=self.padding[0], transposed=True, act=act, packer=self._packer)

    def run_latent(self, x, act=L.ACT_NONE, want_lo=True):
        self._check()
        if not hasattr(self, "_packer"):
            self._packer = Fn.PackedWeight(shaped=getattr(self, "shaped_weights", False), tr_stride=self.stride[0])
        return Fn.conv2d_latent(x, self.weight, self.bias, kernel_size=self.kernel_size[0], stride=self.stride[0],
                                padding=self.padding[0], transposed=True, act=act, packer=self._packer, want_lo=want_lo)

    def run_gdn(self, x, gdn):
        if Fn.conv2d_gdn_fusable(x, self.weight, gdn.beta.numel(), True):
            self._check()
            if not hasattr(self, "_packer"):
                self._packer = Fn.PackedWeight(shaped=getattr(self, "shaped_weights", False), tr_stride=self.stride[0])
            return Fn.conv2d_gdn(x, self.weight, self.bias, gdn.beta, gdn.gamma, kernel_size=self.kernel_size[0],
                                 stride=self.stride[0], padding=self.padding[0], transposed=True, inverse=gdn.inverse,
                                 beta_min=gdn.beta_min, packer=self._packer, gdn_packer=gdn.packer())
        return Fn.gdn(self.run(x), gdn.beta, gdn.gamma, gdn.inverse, gdn.beta_min)

    def run_cat(self, xa, xb, gdn=None, gdn_on_input=False):
        self._check()
        return Fn.conv2d_cat(xa, xb, self.weight, self.bias, kernel_size=self.kernel_size[0], stride=self.stride[0],
                             padding=self.padding[0], transposed=True, gdn=gdn, gdn_on_input=gdn_on_input)

    def forward(self, x, output_size=None):
        if output_size is not None:
            raise NotImplementedError("hesic_amd deconv: output_size is fixed by output_padding = stride-1")
        if _ho.active(x):
            return _ho.conv(self, x, transposed=True)
        return self.run(x)


def _plain_inputs(fn):
    """The ``run*`` entry points launch at once; a Carrier among their tensor arguments (a deferred output of another module of the package,
    hesic_amd/handover.py) is produced first."""
    import functools

    @functools.wraps(fn)
    def wrapped(self, *args, **kwargs):
        if any(type(a) is _ho.Carrier for a in args):
            args = tuple(_ho.plain(a) for a in args)
        return fn(self, *args, **kwargs)
    return wrapped


for _cls in (HipConv2d, HipConvTranspose2d):
    for _name in ("run", "run_into", "run_slice", "run_latent", "run_gdn", "run_cat", "run_gdn_hilo_out"):
        if _name in _cls.__dict__:
            setattr(_cls, _name, _plain_inputs(_cls.__dict__[_name]))


def conv(in_channels, out_channels, kernel_size=5, stride=2):
    return HipConv2d(in_channels, out_channels, kernel_size=kernel_size, stride=stride, padding=kernel_size // 2)


def deconv(in_channels, out_channels, kernel_size=5, stride=2):
    return HipConvTranspose2d(in_channels, out_channels, kernel_size=kernel_size, stride=stride,
                              output_padding=stride - 1, padding=kernel_size // 2)


# ---- state-dict helpers for the entropy-model buffers that change size on update()
def find_named_module(module, query):
    return next((m for n, m in module.named_modules() if n == query), None)


def find_named_buffer(module, query):
    return next((b for n, b in module.named_buffers() if n == query), None)


def _update_registered_buffer(module, buffer_name, state_dict_key, state_dict, policy="resize_if_empty",
                              dtype=torch.int):
    new_size = state_dict[state_dict_key].size()
    registered_buf = find_named_buffer(module, buffer_name)
    if policy in ("resize_if_empty", "resize"):
        if registered_buf is None:
            raise RuntimeError(f'buffer "{buffer_name}" was not registered')
        if policy == "resize" or registered_buf.numel() == 0:
            registered_buf.resize_(new_size)
    elif policy == "register":
        if registered_buf is not None:
            raise RuntimeError(f'buffer "{buffer_name}" was already registered')
        module.register_buffer(buffer_name, torch.empty(new_size, dtype=dtype).fill_(0))
    else:
        raise ValueError(f'Invalid policy "{policy}"')


def update_registered_buffers(module, module_name, buffer_names, state_dict, policy="resize_if_empty",
                              dtype=torch.int):
    """Resize / register ``buffer_names`` of ``module`` so a checkpoint whose CDF tables were filled by
    ``update()`` loads strictly (reference: compressai/models/utils.py:70-101)."""
    valid = [n for n, _ in module.named_buffers()]
    for name in buffer_names:
        if name not in valid:
            raise ValueError(f'Invalid buffer name "{name}"')
    for name in buffer_names:
        _update_registered_buffer(module, name, f"{module_name}.{name}", state_dict, policy, dtype)
