"""INTEGRATION.md path A as a callable: the reference's forward, replayed in ITS call order with nothing but module ``__call__``s of the
drop-in package (``hesic_amd/compressai``: ``conv()`` / ``deconv()`` modules, ``GDN``, ``MaskedConv2d``, the entropy models), plain
``nn.Sequential`` / ``nn.ReLU`` / ``nn.LeakyReLU`` / ``nn.UpsamplingBilinear2d``, ``torch.cat`` / ``torch.abs`` / ``softmax`` and the
kornia-shaped ``warp_perspective`` -- what the reference's own ``newnet1{,_joint}.py`` executes after ``import hesic_amd``
(ywz/mywork/newnet1.py:590-601, :615-624, :641-655, :676-692, :433-437, :441-453, :496-512, :562-577, :724-783;
newnet1_joint.py:675-753).  Every module's output goes to the next module as it is, like in the reference; no fused ``run_*`` entry point, no
``_forward_eval`` schedule.  At inference the modules hand over among themselves (``hesic_amd/handover.py``: deferred conv -> GDN fusion,
hi/lo pairs and fp32 latents between consecutive modules of the package); ``HESIC_NO_HANDOVER=1`` gives round 5's literal launches.  The reference .py files do not travel to the GPU box, so the call order is restated here: ``tests/test_gpu_path_a.py`` checks
it against the reference-recorded goldens and ``bench.py`` times it (``secondary.path_a``); the container-only
``tests/test_dropin_reference_model.py`` loads the real files against the same package."""
import torch
import torch.nn.functional as F


def encoder1(m, x):
    t = m.g_a_conv1(x)                    # newnet1.py:590-600: one module call after the other, each output handed to the next as it is
    t = m.g_a_gdn1(t)
    t = m.g_a_conv2(t)
    t = m.g_a_gdn2(t)
    t = m.g_a_conv3(t)
    t = m.g_a_gdn3(t)
    return m.g_a_conv4(t)


def encoder2(m, x1_warp, x2):
    pre = m.pre_gdn(m.pre_conv(torch.cat((x1_warp, x2), dim=-3)))
    return encoder1(m, pre)


def decoder1(m, y_hat):
    t = m.g_s_conv1(y_hat)
    t = m.g_s_gdn1(t)
    t = m.g_s_conv2(t)
    t = m.g_s_gdn2(t)
    t = m.g_s_conv3(t)
    t = m.g_s_gdn3(t)
    return m.g_s_conv4(t)


def decoder2(m, y_hat, x1_hat_warp):
    after1 = m.after_gdn(decoder1(m, y_hat))
    return m.after_conv(torch.cat((after1, x1_hat_warp.to(after1.dtype)), dim=-3))


def gmm_head(m, inp):
    sigma, means = m.gmm_sigma(inp), m.gmm_means(inp)            # nn.Sequential.__call__ over conv/deconv + nn.ReLU / nn.LeakyReLU
    temp = torch.reshape(m.gmm_weights(inp), (-1, m.K, m.M, 1, 1))
    weights = torch.reshape(F.softmax(temp.float(), dim=-4), (-1, m.M * m.K, 1, 1))
    return sigma, means, weights


def hsic_forward(net, x1, x2, h):
    from hesic_amd.geometry import warp_perspective
    size = (x1.shape[-2], x1.shape[-1])
    y1 = encoder1(net.encoder1, x1)
    z1 = net._h_a1.encode_hyper(torch.abs(y1))
    z1_hat, z1_lik = net.entropy_bottleneck1(z1)
    s1, m1, w1 = gmm_head(net._h_s1, z1_hat)
    y1_hat, y1_lik = net.gaussian1(y1, s1, m1, w1)
    x1_hat = decoder1(net.decoder1, y1_hat)
    x1_warp = warp_perspective(x1, h, size)
    y2 = encoder2(net.encoder2, x1_warp, x2)
    x1_warp_aftercodec = warp_perspective(x1_hat, h, size)
    y1_warpf2 = encoder1(net.encoder1, x1_warp_aftercodec)
    y1_hat_warpf2 = net.gaussian1._quantize(y1_warpf2, "noise" if net.training else "dequantize")
    z2 = net._h_a2.encode_hyper(torch.abs(y2))
    z2_hat, z2_lik = net.entropy_bottleneck2(z2)
    hs2 = net._h_s2
    cat_in = torch.cat((hs2.upsample_layer(z2_hat), y1_hat_warpf2.to(z2_hat.dtype)), dim=-3)
    s2, m2, w2 = gmm_head(hs2, cat_in)
    y2_hat, y2_lik = net.gaussian2(y2, s2, m2, w2)
    x1_hat_warp = warp_perspective(x1_hat, h, size)
    x2_hat = decoder2(net.decoder2, y2_hat, x1_hat_warp)
    return {"x1_hat": x1_hat, "x2_hat": x2_hat, "y1_hat": y1_hat, "y2_hat": y2_hat,
            "likelihoods": {"y1": y1_lik, "y2": y2_lik, "z1": z1_lik, "z2": z2_lik}}


def joint_forward(net, x1, x2, h):
    from hesic_amd.geometry import warp_perspective
    size = (x1.shape[-2], x1.shape[-1])
    mode = "noise" if net.training else "dequantize"
    y1 = encoder1(net.encoder1, x1)
    z1 = net.h_a1(y1)
    z1_hat, z1_lik = net.entropy_bottleneck1(z1)
    params1 = net.h_s1(z1_hat)
    y1_hat = net.gaussian_conditional1._quantize(y1, mode)
    ctx1 = net.context_prediction1(y1_hat)
    gp1 = net.entropy_parameters1(torch.cat((params1, ctx1), dim=1))
    sc1, mu1 = gp1.chunk(2, 1)
    _, y1_lik = net.gaussian_conditional1(y1, sc1, means=mu1)
    x1_hat = decoder1(net.decoder1, y1_hat)
    x1_warp = warp_perspective(x1, h, size)
    y2 = encoder2(net.encoder2, x1_warp, x2)
    z2 = net.h_a2(y2)
    z2_hat, z2_lik = net.entropy_bottleneck2(z2)
    x1_warp_aftercodec = warp_perspective(x1_hat, h, size)
    y1_hat_warpf2 = net.gaussian1._quantize(encoder1(net.encoder1, x1_warp_aftercodec), mode)
    params2 = net.h_s2(z2_hat)
    y2_hat = net.gaussian_conditional2._quantize(y2, mode)
    ctx2 = net.context_prediction2(y2_hat)
    gp2 = net.entropy_parameters2(torch.cat((params2, ctx2, y1_hat_warpf2.to(params2.dtype)), dim=1))
    sc2, mu2 = gp2.chunk(2, 1)
    _, y2_lik = net.gaussian_conditional1(y2, sc2, means=mu2)
    x1_hat_warp = warp_perspective(x1_hat, h, size)
    x2_hat = decoder2(net.decoder2, y2_hat, x1_hat_warp)
    return {"x1_hat": x1_hat, "x2_hat": x2_hat, "y1_hat": y1_hat, "y2_hat": y2_hat,
            "likelihoods": {"y1": y1_lik, "y2": y2_lik, "z1": z1_lik, "z2": z2_lik}}


FWD = {"hsic": hsic_forward, "joint": joint_forward}
