"""Weight gradients of the wide convs (the backward pass of newtrain1.py:85-96): the finishing passes of several layers in one launch
(hesic_conv2d_wgrad_finish_batched) against one hesic_conv2d_wgrad_direct call per layer.  Both sum the K slices of a value in the
same fixed order, so dW is compared bit for bit (also when a weight receives two gradients per step, encoder1); the bias column
sums are per-slice partials of the weight-gradient kernel added in a fixed order: bit for bit as well, and checked against torch."""
import ctypes as C

import pytest
import torch

from hesic_amd import synthetic

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rnd(name, shape, lo=-1.0, hi=1.0):
    return synthetic._uniform("wb." + name, shape, lo, hi)


LAYERS = [
    # Cin, Cout, k, stride, transposed, (B, H, W) of the conv input, bias
    (128, 128, 5, 2, 0, (2, 24, 40), True),
    (128, 192, 5, 2, 1, (1, 12, 20), True),
    (192, 128, 3, 1, 0, (2, 16, 16), False),
    (128, 128, 5, 2, 1, (2, 8, 8), True),
    (64, 128, 5, 1, 0, (1, 20, 12), True),
    (128, 128, 5, 2, 0, (3, 16, 16), True),
    (128, 64, 3, 1, 1, (2, 9, 11), False),
    (128, 128, 5, 2, 0, (1, 32, 32), True),
    (128, 128, 5, 2, 1, (1, 16, 16), True),       # ninth job: a second launch inside one call
]


def _layer(i, L):
    Cin, Cout, k, s, tr, (B, H, W), has_b = LAYERS[i]
    pad = k // 2
    Ho, Wo = (H * s, W * s) if tr else ((H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1)
    x = rnd(f"x{i}", (B, Cin, H, W), -2, 2).to(DEV, torch.bfloat16).contiguous(memory_format=torch.channels_last)
    gy = rnd(f"g{i}", (B, Cout, Ho, Wo)).to(DEV, torch.bfloat16).contiguous(memory_format=torch.channels_last)
    d = L.ConvDesc(B, H, W, Cin, Ho, Wo, Cout, k, k, s, pad, tr, L.BF16, 0, 0, Cin, 0, Cout, 0, 0)
    wshape = (Cin, Cout, k, k) if tr else (Cout, Cin, k, k)
    return d, x, gy, wshape, Cout, has_b


def test_batched_finish_equals_the_per_layer_launches():
    from hesic_amd import _lib as L
    st = L.stream()
    n = len(LAYERS)
    layers = [_layer(i, L) for i in range(n)]
    ref_dw, ref_db, bat_dw, bat_db, wss = [], [], [], [], []
    for d, x, gy, wshape, Cout, has_b in layers:
        nws = int(L.lib().hesic_conv2d_wgrad_ws_bytes(C.byref(d)))
        ws = torch.empty(max(nws, 16), dtype=torch.uint8, device=DEV)
        dw = torch.full(wshape, 0.25, dtype=torch.float32, device=DEV)         # accumulate: the slot already holds a value
        db = torch.full((Cout,), -0.5, dtype=torch.float32, device=DEV) if has_b else None
        L.call("hesic_conv2d_wgrad_direct", C.byref(d), L.ptr(x), L.ptr(gy), L.ptr(dw), L.ptr(db), 1, L.ptr(ws), nws, st)
        ref_dw.append(dw); ref_db.append(db)
        ws2 = torch.empty(max(nws, 16), dtype=torch.uint8, device=DEV)
        L.call("hesic_conv2d_wgrad_partial", C.byref(d), L.ptr(x), L.ptr(gy), L.ptr(ws2), nws, st)
        wss.append(ws2)
        bat_dw.append(torch.full(wshape, 0.25, dtype=torch.float32, device=DEV))
        bat_db.append(torch.full((Cout,), -0.5, dtype=torch.float32, device=DEV) if has_b else None)
    vp = C.c_void_p * n
    descs = (L.ConvDesc * n)(*[l[0] for l in layers])
    L.call("hesic_conv2d_wgrad_finish_batched", n, descs, vp(*[w.data_ptr() for w in wss]), vp(*[l[2].data_ptr() for l in layers]),
           vp(*[w.data_ptr() for w in bat_dw]), vp(*[(b.data_ptr() if b is not None else None) for b in bat_db]), 1, st)
    torch.cuda.synchronize()
    for i in range(n):
        assert torch.equal(bat_dw[i], ref_dw[i]), f"layer {i}: dW differs"
        assert float((ref_dw[i] - 0.25).abs().max()) > 1e-3
        if ref_db[i] is not None:
            # the column sums of dY come out of the weight-gradient kernel itself (a fragment of ones on the matrix cores, per-slice
            # partials summed in a fixed order): deterministic, and equal to the plain fp32 sum up to the summation order
            assert torch.equal(bat_db[i], ref_db[i]), f"layer {i}: dbias differs"
            want = layers[i][2].float().sum((0, 2, 3)) - 0.5
            assert float((bat_db[i] - want).abs().max()) <= 2e-5 * float(want.abs().max() + 1.0), f"layer {i}: dbias vs torch"


def test_batched_finish_rejects_two_jobs_on_one_gradient():
    from hesic_amd import _lib as L
    d, x, gy, wshape, Cout, _ = _layer(0, L)
    nws = int(L.lib().hesic_conv2d_wgrad_ws_bytes(C.byref(d)))
    ws = torch.zeros(max(nws, 16), dtype=torch.uint8, device=DEV)
    dw = torch.zeros(wshape, dtype=torch.float32, device=DEV)
    vp = C.c_void_p * 2
    descs = (L.ConvDesc * 2)(d, d)
    with pytest.raises(RuntimeError, match="same gradient"):
        L.call("hesic_conv2d_wgrad_finish_batched", 2, descs, vp(ws.data_ptr(), ws.data_ptr()), vp(gy.data_ptr(), gy.data_ptr()),
               vp(dw.data_ptr(), dw.data_ptr()), vp(None, None), 1, L.stream())


def test_trainer_step_is_the_same_with_and_without_the_batched_finish():
    """One Trainer.step from the same state through the queued route (default: split-K launches in shared grids with their own K-slice
    counts, batched finishing passes) and with one launch pair per layer: the same weight gradients up to the summation order of the K
    slices and the scatter-adds of the warp's backward (run to run ~5e-7 of the largest gradient, profiles/scripts/train_determinism.py;
    between the routes up to ~1e-5 was seen once in a dozen runs): 5e-5 of the largest gradient, parameters after Adam likewise."""
    import hesic_amd
    from hesic_amd import functional as Fn, models, train
    hesic_amd.set_compute_dtype(torch.bfloat16)
    outs = []
    x1, x2, h = (t.to(DEV) for t in synthetic.stereo_batch(0, 2, 128, 128))
    try:
        for batch in (8, 0):
            prev, Fn.WGRAD_FINISH_BATCH = Fn.WGRAD_FINISH_BATCH, batch
            try:
                net = models.HSIC()
                synthetic.fill_state_dict_(net.state_dict())
                net = net.to(DEV)
                tr = train.Trainer(net, lmbda=0.0067)
                torch.manual_seed(3)                     # the quantisation noise of the step
                crit = tr.step(x1, x2, h)
                torch.cuda.synchronize()
                outs.append((float(crit["loss"]), tr.main_group.flat_g.clone(), tr.main_group.flat_p.clone()))
                tr.main_reducer.close()
                tr.aux_reducer.close()
            finally:
                Fn.WGRAD_FINISH_BATCH = prev
    finally:
        hesic_amd.set_compute_dtype(torch.float32)
    (l0, g0, p0), (l1, g1, p1) = outs
    assert l0 == l1
    assert float(g1.abs().max()) > 0
    assert float((g0 - g1).abs().max()) <= 5e-5 * float(g1.abs().max())
    # parameters after the first Adam step move by lr * g / (|g| + eps): compared where the gradient is not within rounding of zero
    big = g1.abs() > 1e-4 * float(g1.abs().max())
    assert float((p0 - p1)[big].abs().max()) <= 5e-5 * float(p1.abs().max())


def test_batched_gdn_param_finish_equals_the_per_layer_backward():
    """hesic_gdn_backward_partial + ONE hesic_gdn_param_finish_batched call for several GDN / IGDN backwards (round 5: what a Trainer step
    issues) against hesic_gdn_backward_acc per layer: dx, dgamma and dbeta bit for bit (the block partials are summed in the same order),
    accumulating into slots that already hold a value; two jobs on one gradient in one call are refused."""
    import hesic_amd
    from hesic_amd import _lib as L
    hesic_amd.set_compute_dtype(torch.bfloat16)
    try:
        st = L.stream()
        cases = [(2, 24, 40, False), (1, 64, 64, True), (3, 16, 16, False)]
        ref, bat, keep = [], [], []
        for i, (B, H, W, inv) in enumerate(cases):
            P = B * H * W
            x = rnd(f"gx{i}", (B, 128, H, W), -2, 2).to(DEV, torch.bfloat16).contiguous(memory_format=torch.channels_last)
            gy = rnd(f"gg{i}", (B, 128, H, W)).to(DEV, torch.bfloat16).contiguous(memory_format=torch.channels_last)
            beta = rnd(f"gb{i}", (128,), 0.5, 1.5).to(DEV)
            gamma = (rnd(f"gm{i}", (128, 128), 0.0, 0.02) + 0.1 * torch.eye(128)).to(DEV)
            assert L.lib().hesic_gdn_backward_partial_ok(P, 128, L.BF16) == 1
            nws = int(L.lib().hesic_gdn_backward_ws_bytes(P, 128))
            out = []
            for mode in ("ref", "bat"):
                ws = torch.empty(nws, dtype=torch.uint8, device=DEV)
                dx = torch.empty_like(x)
                dg = torch.full((128, 128), 0.25, device=DEV)
                db = torch.full((128,), -0.5, device=DEV)
                if mode == "ref":
                    L.call("hesic_gdn_backward_acc", L.ptr(x), L.ptr(gy), L.ptr(beta), L.ptr(gamma), L.ptr(dx), L.ptr(db), L.ptr(dg), 1, L.ptr(ws), P, 128,
                           int(inv), 1e-6, L.BF16, st)
                else:
                    L.call("hesic_gdn_backward_partial", L.ptr(x), L.ptr(gy), L.ptr(beta), L.ptr(gamma), L.ptr(dx), L.ptr(ws), P, 128, int(inv), 1e-6, L.BF16, st)
                out.append((dx, dg, db, ws, beta, gamma, P))
            ref.append(out[0]); bat.append(out[1])
        n = len(cases)
        vp, i64, f32 = C.c_void_p * n, C.c_int64 * n, C.c_float * n
        L.call("hesic_gdn_param_finish_batched", n, vp(*[b[3].data_ptr() for b in bat]), i64(*[b[6] for b in bat]), vp(*[b[4].data_ptr() for b in bat]),
               vp(*[b[5].data_ptr() for b in bat]), vp(*[b[1].data_ptr() for b in bat]), vp(*[b[2].data_ptr() for b in bat]), f32(*[1e-6] * n), 1, st)
        torch.cuda.synchronize()
        for i in range(n):
            assert torch.equal(ref[i][0], bat[i][0]), f"case {i}: dx"
            assert torch.equal(ref[i][1], bat[i][1]), f"case {i}: dgamma"
            assert torch.equal(ref[i][2], bat[i][2]), f"case {i}: dbeta"
            assert float((ref[i][1] - 0.25).abs().max()) > 1e-4
        with pytest.raises(RuntimeError, match="same gradient"):
            L.call("hesic_gdn_param_finish_batched", 2, vp(bat[0][3].data_ptr(), bat[0][3].data_ptr(), None), i64(bat[0][6], bat[0][6], 0),
                   vp(bat[0][4].data_ptr(), bat[0][4].data_ptr(), None), vp(bat[0][5].data_ptr(), bat[0][5].data_ptr(), None),
                   vp(bat[0][1].data_ptr(), bat[0][1].data_ptr(), None), vp(bat[0][2].data_ptr(), bat[0][2].data_ptr(), None), f32(1e-6, 1e-6, 0), 1, st)
    finally:
        hesic_amd.set_compute_dtype(torch.float32)


def test_batched_split_k_launch_leaves_the_same_partials_as_the_per_layer_launches():
    """hesic_conv2d_wgrad_partial_batched (round 5: the split-K launches of several layers in shared grids, longest K slices first) against one
    hesic_conv2d_wgrad_partial per layer: the same blocks do the same work in the same order, so every workspace -- K-slice partial tiles
    and the per-slice bias column sums behind them -- is compared bit for bit.  Sixteen jobs: two shared grids (14 + 2) inside one call,
    with a non-multiple-of-8 block count per job (the 8-aligned ranges' surplus ids exit) and a large layer that goes to the row kernel."""
    from hesic_amd import _lib as L
    st = L.stream()
    idx = list(range(len(LAYERS))) + [0, 3, 5, 7, 1, 8]
    jobs = [_layer(i, L) for i in idx]
    # one layer the row kernel takes (5x5 stride 2, rows of 64 pixels, >= 100 000 pixels): launched on its own inside the batched call
    B, H, W = 8, 256, 256
    xb = rnd("xrow", (B, 128, H, W), -2, 2).to(DEV, torch.bfloat16).contiguous(memory_format=torch.channels_last)
    gb = rnd("grow", (B, 128, H // 2, W // 2)).to(DEV, torch.bfloat16).contiguous(memory_format=torch.channels_last)
    jobs.append((L.ConvDesc(B, H, W, 128, H // 2, W // 2, 128, 5, 5, 2, 2, 0, L.BF16, 0, 0, 128, 0, 128, 0, 0), xb, gb, None, 128, True))
    n = len(jobs)
    ref, bat, nbytes = [], [], []
    for d, x, gy, *_ in jobs:
        nws = int(L.lib().hesic_conv2d_wgrad_ws_bytes(C.byref(d)))
        a = torch.full((max(nws, 16),), 0x5a, dtype=torch.uint8, device=DEV)
        L.call("hesic_conv2d_wgrad_partial", C.byref(d), L.ptr(x), L.ptr(gy), L.ptr(a), nws, st)
        ref.append(a)
        bat.append(torch.full((max(nws, 16),), 0x5a, dtype=torch.uint8, device=DEV))
        nbytes.append(nws)
    vp = C.c_void_p * n
    L.call("hesic_conv2d_wgrad_partial_batched", n, (L.ConvDesc * n)(*[j[0] for j in jobs]), vp(*[j[1].data_ptr() for j in jobs]),
           vp(*[j[2].data_ptr() for j in jobs]), vp(*[b.data_ptr() for b in bat]), (C.c_int64 * n)(*nbytes), None, st)
    torch.cuda.synchronize()
    for i in range(n):
        assert torch.equal(bat[i], ref[i]), f"job {i}: workspace differs"
    # a workspace that is too small is refused, naming the job
    short = list(nbytes)
    short[2] -= 4
    with pytest.raises(RuntimeError, match="job 2"):
        L.call("hesic_conv2d_wgrad_partial_batched", n, (L.ConvDesc * n)(*[j[0] for j in jobs]), vp(*[j[1].data_ptr() for j in jobs]),
               vp(*[j[2].data_ptr() for j in jobs]), vp(*[b.data_ptr() for b in bat]), (C.c_int64 * n)(*short), None, st)
    L.call("hesic_conv2d_wgrad_partial_batched", 0, None, None, None, None, None, None, st)       # an empty queue is a no-op


def test_batched_route_with_its_own_k_slice_counts_matches_the_direct_gradients():
    """The Trainer's route: hesic_conv2d_wgrad_nsplit(d, 1) K slices per layer (fewer than a launch of its own takes), workspaces from
    hesic_conv2d_wgrad_ws_bytes_n, hesic_conv2d_wgrad_partial_batched + hesic_conv2d_wgrad_finish_batched_n with the same counts.  Another slice
    count is another summation order: dW / dbias against hesic_conv2d_wgrad_direct within fp32 rounding of the sums, and a count the finishing
    call is not told about must not be silently accepted for a layer another kernel takes."""
    from hesic_amd import _lib as L
    st = L.stream()
    idx = list(range(len(LAYERS)))
    jobs = [_layer(i, L) for i in idx]
    # a layer of the training step's size (128 -> 128 5x5 on 32^2 maps, B = 8: 8192 pixels), where the two policies differ
    xb = rnd("xb8", (8, 128, 32, 32), -2, 2).to(DEV, torch.bfloat16).contiguous(memory_format=torch.channels_last)
    gb = rnd("gb8", (8, 128, 32, 32)).to(DEV, torch.bfloat16).contiguous(memory_format=torch.channels_last)
    jobs.append((L.ConvDesc(8, 32, 32, 128, 32, 32, 128, 5, 5, 1, 2, 0, L.BF16, 0, 0, 128, 0, 128, 0, 0), xb, gb, (128, 128, 5, 5), 128, True))
    n = len(jobs)
    nsp = [int(L.lib().hesic_conv2d_wgrad_nsplit(C.byref(j[0]), 1)) for j in jobs]
    nsp0 = [int(L.lib().hesic_conv2d_wgrad_nsplit(C.byref(j[0]), 0)) for j in jobs]
    assert all(a >= 1 and a <= b for a, b in zip(nsp, nsp0)) and nsp[-1] < nsp0[-1], (nsp, nsp0)
    # a layer another kernel takes keeps its count (fp32 storage: the VALU fallback)
    df = L.ConvDesc(2, 16, 16, 64, 16, 16, 64, 3, 3, 1, 1, 0, L.F32, 0, 0, 64, 0, 64, 0, 0)
    assert L.lib().hesic_conv2d_wgrad_nsplit(C.byref(df), 1) == L.lib().hesic_conv2d_wgrad_nsplit(C.byref(df), 0)
    nbytes = [int(L.lib().hesic_conv2d_wgrad_ws_bytes_n(C.byref(j[0]), k)) for j, k in zip(jobs, nsp)]
    assert all(int(L.lib().hesic_conv2d_wgrad_ws_bytes_n(C.byref(j[0]), k)) == int(L.lib().hesic_conv2d_wgrad_ws_bytes(C.byref(j[0])))
               for j, k in zip(jobs, nsp0))
    ws = [torch.empty(max(b, 16), dtype=torch.uint8, device=DEV) for b in nbytes]
    dw = [torch.full(j[3], 0.25, dtype=torch.float32, device=DEV) for j in jobs]
    db = [torch.full((j[4],), -0.5, dtype=torch.float32, device=DEV) if j[5] else None for j in jobs]
    vp, i32 = C.c_void_p * n, C.c_int32 * n
    descs = (L.ConvDesc * n)(*[j[0] for j in jobs])
    L.call("hesic_conv2d_wgrad_partial_batched", n, descs, vp(*[j[1].data_ptr() for j in jobs]), vp(*[j[2].data_ptr() for j in jobs]),
           vp(*[w.data_ptr() for w in ws]), (C.c_int64 * n)(*nbytes), i32(*nsp), st)
    L.call("hesic_conv2d_wgrad_finish_batched_n", n, descs, vp(*[w.data_ptr() for w in ws]), vp(*[j[2].data_ptr() for j in jobs]),
           vp(*[w.data_ptr() for w in dw]), vp(*[(b.data_ptr() if b is not None else None) for b in db]), 1, i32(*nsp), st)
    for i, (d, x, gy, wshape, Cout, has_b) in enumerate(jobs):
        nws = int(L.lib().hesic_conv2d_wgrad_ws_bytes(C.byref(d)))
        w0 = torch.empty(max(nws, 16), dtype=torch.uint8, device=DEV)
        rdw = torch.full(wshape, 0.25, dtype=torch.float32, device=DEV)
        rdb = torch.full((Cout,), -0.5, dtype=torch.float32, device=DEV) if has_b else None
        L.call("hesic_conv2d_wgrad_direct", C.byref(d), L.ptr(x), L.ptr(gy), L.ptr(rdw), L.ptr(rdb), 1, L.ptr(w0), nws, st)
        torch.cuda.synchronize()
        scale = float((rdw - 0.25).abs().max())
        assert float((dw[i] - rdw).abs().max()) <= 2e-5 * scale + 1e-6, f"layer {i}: dW"
        if has_b:
            assert float((db[i] - rdb).abs().max()) <= 2e-5 * float(rdb.abs().max() + 1.0), f"layer {i}: dbias"
