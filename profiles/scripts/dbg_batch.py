import torch, sys
sys.path.insert(0, "/root/repo")
import hesic_amd
from hesic_amd import models, synthetic, functional as Fn
hesic_amd.set_compute_dtype(torch.float16)
enc = models.Encoder1(128, 192).cuda().eval()
synthetic.fill_state_dict_(enc.state_dict())
x = synthetic.stereo_batch(0, 8, 512, 512)[0].cuda()
with torch.no_grad():
    for mode in ("x3c2", "x3"):
        Fn.set_analysis_precision(mode)
        _, y8 = enc.latent(x, want_lo=False, exact=True)
        _, y8b = enc.latent(x, want_lo=False, exact=True)
        _, y1 = enc.latent(x[-1:], want_lo=False, exact=True)
        _, y2 = enc.latent(x[-2:], want_lo=False, exact=True)
        print(mode, "repeat equal", torch.equal(y8, y8b), "last alone equal", torch.equal(y8[-1:], y1), "n diff", int((y8[-1:] != y1).sum()), "maxdiff", float((y8[-1:] - y1).abs().max()),
              "last of two", torch.equal(y8[-1:], y2[-1:]))
        # layer by layer in x3c2
    Fn.set_analysis_precision("x3c2")
    c1, g1 = enc.g_a_conv1, enc.g_a_gdn1
    gp, bp = g1.packer().get(g1.beta, g1.gamma, g1.beta_min)
    if not hasattr(enc, "_hl1"): enc._hl1 = Fn.PackedWeightHiLo(), Fn.PackedGdnLo(), Fn.PackedN2wHiLo()
    img = enc._hl1[2].get(c1.weight, g1.gamma, out1=True)
    t8 = Fn.sconv_gdn_hilo(x, img, c1.bias, bp, g1.inverse, out1=True)
    t1 = Fn.sconv_gdn_hilo(x[-1:], img, c1.bias, bp, g1.inverse, out1=True)
    print("conv1 out1: last alone equal", torch.equal(t8[-1:], t1))
    u8 = enc.g_a_conv2.run_gdn_hilo_out(t8, enc.g_a_gdn2); u1 = enc.g_a_conv2.run_gdn_hilo_out(t8[-1:].contiguous(memory_format=torch.channels_last), enc.g_a_gdn2)
    print("conv2: last alone equal", torch.equal(u8[-1:], u1), int((u8[-1:] != u1).sum()))
    v8 = enc.g_a_conv3.run_hilo(u8, gdn=enc.g_a_gdn3); v1 = enc.g_a_conv3.run_hilo(u8[-1:].contiguous(memory_format=torch.channels_last), gdn=enc.g_a_gdn3)
    print("conv3: last alone equal", torch.equal(v8[-1:], v1), int((v8[-1:] != v1).sum()))
