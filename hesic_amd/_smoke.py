"""One small end-to-end invocation of the hot path on the GPU, checked against the CPU oracle.
(The oracle import here is the checker -- allowed for smoke(), see oracle/hesic_oracle.py header.)"""
import torch


def run(device):
    import hesic_amd
    from hesic_amd import models, synthetic
    from oracle import hesic_oracle as O
    if device.type != "cuda":
        raise RuntimeError("smoke(): needs a ROCm device, the HIP path has no CPU fallback")
    hesic_amd.set_compute_dtype(torch.float32)
    net = models.HSIC()
    synthetic.fill_state_dict_(net.state_dict())
    P = {k: v.clone() for k, v in net.state_dict().items()}
    net = net.to(device).eval()
    x1, x2, Hm = synthetic.stereo_batch(0, 1, 64, 64)
    with torch.no_grad():
        out = net(x1.to(device), x2.to(device), Hm.to(device))
        m = models.metrics_from(models.rate_distortion(out, x1.to(device), x2.to(device)))
        ref = O.hsic_forward(P, x1, x2, Hm)
    mr = O.metrics(ref, x1, x2)
    flips = float((out["y1_hat"].cpu() != ref["y1_hat"]).float().mean())
    assert flips < 1e-3, f"smoke: {flips:.2e} of the y1 latents differ from the oracle"
    assert abs(m["bpp"] - mr["bpp"]) < 1e-3 * max(1.0, mr["bpp"]), (m["bpp"], mr["bpp"])
    assert abs(m["psnr"] - mr["psnr"]) < 1e-3, (m["psnr"], mr["psnr"])
    # and one bf16 forward (the benchmark configuration) for shape / finiteness
    hesic_amd.set_compute_dtype(torch.bfloat16)
    with torch.no_grad():
        ob = net(x1.to(device), x2.to(device), Hm.to(device))
    hesic_amd.set_compute_dtype(torch.float32)
    assert ob["x2_hat"].shape == (1, 3, 64, 64) and bool(torch.isfinite(ob["x2_hat"]).all())
    print(f"smoke ok: HESIC 64x64 fp32 vs oracle  bpp {m['bpp']:.5f}/{mr['bpp']:.5f}  psnr {m['psnr']:.4f}/{mr['psnr']:.4f}  "
          f"latent flips {flips:.1e}; bf16 forward finite")
