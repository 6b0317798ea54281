"""Rate-distortion evaluation helpers on top of ``models.rate_distortion`` (no kernels of their own).

``LambdaSweep`` is BASELINE config C5: four lambda-models (ywz/mywork/newtrain1.py:184-185 -- lambda only names an independently
trained model) evaluated on InStereo2K-size pairs zero-padded to multiples of 64, metrics over the original pixels
(ywz/mywork/test3real.py:69-72,110-122).  ``bench.py --sweep`` times it, ``tests/test_gpu_baseline_workloads.py`` checks its
accumulators against the reference's own run.
"""
import math

import torch

from . import models, synthetic

SWEEP_LAMBDAS = (0.0018, 0.0035, 0.0067, 0.0130)


class LambdaSweep:
    """Four models (weight salts 0..3 of ``synthetic.fill_state_dict_`` -- there are no trained checkpoints offline) and a (4, 4)
    fp64 device accumulator per model: total bits, squared error of view 1, of view 2, pairs seen.  ``step(m, ...)`` is one eval
    forward of model ``m`` on a padded batch plus the reductions; nothing is read back until ``summary()``."""

    def __init__(self, kind, device, nets=None):
        if nets is None:
            nets = []
            for m in range(4):
                net = (models.HSIC if kind == "hsic" else models.HSICJoint)()
                synthetic.fill_state_dict_(net.state_dict(), salt=m)
                nets.append(net.to(device).eval())
        self.nets = nets
        self.acc = torch.zeros((4, 4), dtype=torch.float64, device=device)

    def reset(self):
        self.acc.zero_()

    def step(self, m, x1, x2, x1_padded, x2_padded, h_matrix, record=True):
        """x1 / x2: the original (H, W) images the metrics are taken on; x*_padded: the same zero-padded to multiples of 64."""
        with torch.no_grad():
            out = self.nets[m](x1_padded, x2_padded, h_matrix)
            rd = models.rate_distortion(out, x1, x2)
        if record:
            bits = sum(v for kk, v in rd.items() if kk.startswith("bits_"))
            self.acc[m, 0:1] += bits
            self.acc[m, 1:2] += rd["sse1"]
            self.acc[m, 2:3] += rd["sse2"]
            self.acc[m, 3] += x1.shape[0]
        return out, rd

    def summary(self, height, width, acc=None):
        """{lambda: {pairs, bpp, psnr}} over the ORIGINAL pixel count (bpp counts both views: / 2 per view as the reference prints it)."""
        acc = self.acc if acc is None else acc
        per = {}
        for m, lam in enumerate(SWEEP_LAMBDAS):
            bits, s1, s2, n = (float(v) for v in acc[m])
            if n > 0:
                npx = n * height * width
                p1, p2 = 10 * math.log10(npx * 3 / s1), 10 * math.log10(npx * 3 / s2)
                per[str(lam)] = {"pairs": int(n), "bpp": bits / npx / 2, "psnr": (p1 + p2) / 2, "bits": bits, "sse1": s1, "sse2": s2}
        return per
