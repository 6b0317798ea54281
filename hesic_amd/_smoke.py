"""One small end-to-end invocation of the hot path on the GPU, checked against the CPU oracle.
(The oracle import here is the checker -- allowed for smoke(), see oracle/hesic_oracle.py header.)"""
import torch


def run(device):
    from . import functional as Fn, synthetic
    from oracle import hesic_oracle as O
    x = synthetic._uniform("smoke.x", (1, 128, 16, 16), -1, 1)
    w = synthetic._uniform("smoke.w", (128, 128, 5, 5), -0.03, 0.03)
    b = synthetic._uniform("smoke.b", (128,), -0.1, 0.1)
    y = Fn.conv2d(x.to(device), w.to(device), b.to(device), kernel_size=5, stride=2, padding=2)
    ref = O.conv(x, w, b, 2)
    err = float((y.cpu() - ref).abs().max() / ref.abs().max())
    assert err < 1e-4, f"smoke: conv mismatch {err}"
    print(f"smoke ok: conv rel err {err:.2e}")
