"""The C-ABI calls of one HESIC+ wavefront group step (what a group graph captures): name and argument count."""
import sys, os, tempfile, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hesic_amd
from hesic_amd import models, synthetic, _lib as L, functional as Fn
hesic_amd.set_compute_dtype(torch.float16)
net = models.HSICJoint(); synthetic.fill_state_dict_(net.state_dict()); net = net.cuda().eval(); net.update(force=True)
x1, x2, Hm = (t.cuda() for t in synthetic.stereo_batch(0, 1, 128, 192))
with tempfile.TemporaryDirectory() as td:
    enc = net.compress(x1, x2, Hm, "w", td)
    dec = net.decompress(None, None, Hm, "w", td)
    st = net._wf_cache[(2, 8, 12, 0)]
    orig = L.call
    log = []
    def call(name, *a):
        log.append((name, len(a)))
        return orig(name, *a)
    L.call = call
    st["pos"].zero_(); st["state"].zero_()
    with torch.no_grad(), Fn.no_split_k():
        net._wavefront_step_body(st, st["groups"][0])
    torch.cuda.synchronize()
    L.call = orig
    for n in log: print(n)
