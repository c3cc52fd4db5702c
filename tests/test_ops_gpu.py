"""GPU parity of every C-ABI op against the CPU oracle ops (torch functional, fp64 reference)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import detgen, ae_oracle

pytestmark = pytest.mark.gpu

ACTS = list(ae_oracle.ACTS)


def dev():
    return torch.device("cuda:0")


def T(name, shape, lo=-1.0, hi=1.0):
    return torch.from_numpy(detgen.uniform(name, shape, lo, hi))


def close(got, ref, tol=2e-5, name=""):
    got = got.detach().cpu().double()
    ref = ref.detach().cpu().double()
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    err = (got - ref).abs().max().item()
    scale = ref.abs().max().item() + 1e-30
    assert err <= tol * scale, f"{name}: max err {err:.3e} vs scale {scale:.3e}"


def act_and_grad(z, act):
    z = z.clone().requires_grad_(True)
    a = ae_oracle.act_forward(z, act)
    (d,) = torch.autograd.grad(a.sum(), z)
    return a.detach(), d.detach()


CONV_SHAPES = [
    # N, Cin, H, W, Cout, k, stride
    (2, 3, 32, 32, 64, 3, 1),
    (2, 64, 32, 32, 64, 3, 2),
    (3, 64, 16, 16, 64, 3, 1),
    (2, 64, 16, 16, 64, 3, 2),
    (5, 64, 8, 8, 64, 3, 1),
    (2, 64, 32, 32, 3, 3, 1),
    (2, 16, 16, 16, 16, 3, 1),
    (1, 8, 8, 8, 8, 5, 1),
    (2, 8, 16, 16, 8, 5, 2),
    (1, 16, 64, 64, 16, 3, 1),
    (16, 64, 16, 16, 64, 3, 1),   # enough base positions for the 64x128 tile
    (3, 24, 8, 8, 40, 3, 1),      # ragged channel counts
    # >= 256 blocks of 32 tiles x 64 channels: forward and data gradient run the Winograd F(2x2,3x3) kernel (conv_wino.hip):
    # the train-step shape; 2 chunks + 48 of 64 output channels; 8x8 images (2 images per block, one chunk);
    # an odd number of chunks + two output-channel blocks, the second one ragged
    (128, 64, 16, 16, 64, 3, 1),
    (32, 16, 32, 32, 48, 3, 1),
    (512, 8, 8, 8, 16, 3, 1),
    (64, 24, 16, 16, 80, 3, 1),
    (256, 8, 8, 16, 16, 3, 1),    # Winograd on non-square images (tile rows of 8, 4 tile rows per image)
    # <= 4 channels on the image side and >= 1024 pixels: forward (thin input) and data gradient (thin output of the
    # conv) run conv_thin_in.hip: odd channel count across two channel blocks, 2 and 4 channels, wide and narrow rows
    (2, 3, 32, 32, 21, 3, 1),
    (1, 2, 16, 64, 10, 3, 1),
    (2, 4, 128, 8, 10, 3, 1),
    (2, 21, 32, 32, 3, 3, 1),
    (1, 10, 64, 64, 4, 3, 1),
    (2, 10, 128, 8, 2, 3, 1),
    # <= 4 output channels, Cin % 8 == 0, widths 16..64: the forward runs conv_thin_outv.hip (packed FMAs, DPP neighbours)
    (3, 32, 16, 16, 4, 3, 1),
    (1, 16, 64, 64, 1, 3, 1),
    (2, 48, 16, 32, 2, 3, 1),
    (2, 24, 16, 16, 3, 3, 1),
    # channel counts that are not a multiple of the 8-channel K-chunk: pipelined full chunks + masked tail chunk,
    # odd / even numbers of full chunks (zero-weight padding chunk), stride 1 and 2 (dgrad runs the other modes)
    (2, 20, 16, 16, 12, 3, 1),
    (2, 100, 16, 16, 36, 3, 2),
    (2, 28, 16, 16, 12, 3, 2),
    (1, 9, 8, 8, 17, 3, 1),
    # thin-channel wgrad kernel (conv_wgrad_thin.h): thin input / thin output, ragged + multi-tile wide side, k=5,
    # 64x64 images, and enough 256-pixel chunks (N*H*W/256 > 256 blocks) for several double-buffered iterations
    (3, 3, 64, 64, 16, 3, 1),
    (2, 16, 32, 32, 3, 3, 1),
    (2, 3, 32, 32, 80, 3, 1),
    (2, 3, 32, 32, 8, 5, 1),
    (2, 40, 16, 16, 2, 5, 1),
    (1, 1, 16, 16, 4, 3, 1),
    (200, 3, 32, 32, 64, 3, 1),
    (200, 32, 32, 32, 3, 3, 1),
    # thin-output forward kernel (conv_thin_out.hip): ragged / >64 input channels, k=5 (3 column tiles), 64x64, 16x16
    (2, 70, 32, 32, 3, 3, 1),
    (2, 16, 32, 32, 3, 5, 1),
    (1, 24, 64, 64, 2, 3, 1),
    (3, 8, 16, 16, 1, 3, 1),
    # 3x3 weight-gradient kernel (conv_wgrad16.h): several chunks per block so that the double-buffered pipeline reaches
    # its steady state and its look-ahead runs past the end (even / odd / ragged chunk counts per block), every width
    (64, 64, 16, 16, 64, 3, 1),
    (200, 32, 16, 16, 32, 3, 1),
    (40, 64, 32, 32, 64, 3, 2),
    (150, 64, 16, 16, 64, 3, 2),
    (300, 32, 8, 8, 32, 3, 1),
    (6, 64, 32, 32, 64, 3, 1),
    (5, 16, 64, 64, 16, 3, 1),
    (3, 40, 64, 64, 24, 3, 2),
]


@pytest.mark.parametrize("shape", CONV_SHAPES)
@pytest.mark.parametrize("act", ["gelu", None])
def test_conv2d_fwd(shape, act):
    from gmk import ops
    N, Cin, H, W, Cout, k, s = shape
    x = T(f"cx{shape}", (N, Cin, H, W))
    w = T(f"cw{shape}", (Cout, Cin, k, k)) / np.sqrt(Cin * k * k)
    b = T(f"cb{shape}", (Cout,))
    pre = F.conv2d(x.double(), w.double(), b.double(), s, (k - 1) // 2)
    y, dact = ops.conv2d_fwd(x.to(dev()), w.to(dev()), b.to(dev()), stride=s, act=act, want_dact=True)
    if act is None:
        close(y, pre, name="y")
    else:
        a, d = act_and_grad(pre, act)
        close(y, a, name="y")
        close(dact, d, tol=1e-4, name="dact")


@pytest.mark.parametrize("act", ACTS)
def test_activations_in_conv_epilogue(act):
    from gmk import ops
    x = T("ax", (2, 8, 8, 8), -3, 3)
    w = T("aw", (8, 8, 3, 3), -1, 1)
    b = T("ab", (8,), -2, 2)
    pre = F.conv2d(x.double(), w.double(), b.double(), 1, 1)
    a, d = act_and_grad(pre, act)
    y, dact = ops.conv2d_fwd(x.to(dev()), w.to(dev()), b.to(dev()), act=act, want_dact=True)
    close(y, a, name="y")
    close(dact, d, tol=1e-4, name="dact")


@pytest.mark.parametrize("shape", CONV_SHAPES)
def test_conv2d_dgrad_wgrad(shape):
    from gmk import ops
    N, Cin, H, W, Cout, k, s = shape
    x = T(f"gx{shape}", (N, Cin, H, W)).double().requires_grad_(True)
    w = (T(f"gw{shape}", (Cout, Cin, k, k)) / np.sqrt(Cin * k * k)).double().requires_grad_(True)
    b = T(f"gb{shape}", (Cout,)).double().requires_grad_(True)
    dy = T(f"gdy{shape}", (N, Cout, H // s, W // s))
    dact_in = T(f"gda{shape}", (N, Cin, H, W), 0.0, 1.0)
    y = F.conv2d(x, w, b, s, (k - 1) // 2)
    gx, gw, gb = torch.autograd.grad(y, (x, w, b), dy.double())
    dx = ops.conv2d_dgrad(dy.to(dev()), w.detach().float().to(dev()), (N, Cin, H, W), stride=s, dact_in=dact_in.to(dev()))
    close(dx, gx * dact_in.double(), name="dx")
    dx2 = ops.conv2d_dgrad(dy.to(dev()), w.detach().float().to(dev()), (N, Cin, H, W), stride=s)
    close(dx2, gx, name="dx(no dact)")
    dw, db = ops.conv2d_wgrad(dy.to(dev()), x.detach().float().to(dev()), k, stride=s)
    close(dw, gw, name="dw")
    close(db, gb, name="db")


DECONV_SHAPES = [(2, 64, 8, 8, 64, 3), (2, 64, 16, 16, 64, 3), (3, 16, 8, 8, 24, 3), (1, 8, 8, 8, 8, 5), (9, 64, 16, 16, 64, 3),
                 (2, 20, 8, 8, 12, 3), (2, 44, 16, 16, 100, 3),  # K-chunk tails in every mode
                 (70, 64, 8, 8, 64, 3), (50, 64, 16, 16, 64, 3), (3, 24, 32, 32, 40, 3)]  # several chunks per wgrad block


@pytest.mark.parametrize("shape", DECONV_SHAPES)
def test_deconv2d_all(shape):
    from gmk import ops
    N, Cin, H, W, Cout, k = shape
    p = (k - 1) // 2
    x = T(f"dx{shape}", (N, Cin, H, W)).double().requires_grad_(True)
    w = (T(f"dw{shape}", (Cin, Cout, k, k)) / np.sqrt(Cin * k * k)).double().requires_grad_(True)
    b = T(f"db{shape}", (Cout,)).double().requires_grad_(True)
    dy = T(f"ddy{shape}", (N, Cout, 2 * H, 2 * W))
    pre = F.conv_transpose2d(x, w, b, 2, p, 1)
    a, d = act_and_grad(pre.detach(), "gelu")
    y, dact = ops.deconv2d_fwd(x.detach().float().to(dev()), w.detach().float().to(dev()), b.detach().float().to(dev()),
                               act="gelu", want_dact=True)
    close(y, a, name="y")
    close(dact, d, tol=1e-4, name="dact")
    gx, gw, gb = torch.autograd.grad(pre, (x, w, b), dy.double())
    dxx = ops.deconv2d_dgrad(dy.to(dev()), w.detach().float().to(dev()), (N, Cin, H, W))
    close(dxx, gx, name="dx")
    dw, db = ops.deconv2d_wgrad(dy.to(dev()), x.detach().float().to(dev()), k)
    close(dw, gw, name="dw")
    close(db, gb, name="db")


LIN_SHAPES = [(8, 4096, 512), (8, 256, 4096), (256, 4096, 512), (256, 256, 4096), (5, 100, 37), (50, 4096, 256),
              # configs[4]'s FC layers (deep-K split forward, 1024- and 4096-tile grids) and a ragged shape of that size
              (256, 16384, 1024), (256, 512, 16384), (260, 8200, 1028)]


@pytest.mark.parametrize("shape", LIN_SHAPES)
def test_linear_all(shape):
    from gmk import ops
    M, K, N = shape
    x = T(f"lx{shape}", (M, K)).double().requires_grad_(True)
    w = (T(f"lw{shape}", (N, K)) / np.sqrt(K)).double().requires_grad_(True)
    b = T(f"lb{shape}", (N,)).double().requires_grad_(True)
    dy = T(f"ldy{shape}", (M, N))
    pre = F.linear(x, w, b)
    a, d = act_and_grad(pre.detach(), "gelu")
    y, dact = ops.linear_fwd(x.detach().float().to(dev()), w.detach().float().to(dev()), b.detach().float().to(dev()),
                             act="gelu", want_dact=True)
    close(y, a, name="y")
    close(dact, d, tol=1e-4, name="dact")
    y0 = ops.linear_fwd(x.detach().float().to(dev()), w.detach().float().to(dev()), b.detach().float().to(dev()))
    close(y0, pre, name="y(no act)")
    gx, gw, gb = torch.autograd.grad(pre, (x, w, b), dy.double())
    dact_in = T(f"lda{shape}", (M, K), 0.0, 1.0)
    dx = ops.linear_dgrad(dy.to(dev()), w.detach().float().to(dev()), dact_in=dact_in.to(dev()))
    close(dx, gx * dact_in.double(), name="dx")
    dw, db = ops.linear_wgrad(dy.to(dev()), x.detach().float().to(dev()))
    close(dw, gw, name="dw")
    close(db, gb, name="db")


@pytest.mark.parametrize("B,L", [(8, 256), (3, 16), (256, 256)])
def test_reparam(B, L):
    from gmk import ops
    stats = T(f"rs{B}", (B, L, 2), -2, 2).double().requires_grad_(True)
    eps = torch.from_numpy(detgen.normal(f"re{B}", (B, L)))
    z = stats[:, :, 0] + torch.exp(0.5 * stats[:, :, 1]) * eps.double()
    dz = T(f"rdz{B}", (B, L))
    (gs,) = torch.autograd.grad(z, stats, dz.double())
    zg = ops.reparam_fwd(stats.detach().float().to(dev()), eps.to(dev()))
    close(zg, z, name="z")
    ds = ops.reparam_bwd(dz.to(dev()), stats.detach().float().to(dev()), eps.to(dev()))
    close(ds, gs, name="dstats")
    base = T(f"rb{B}", (B, L, 2))
    ds2 = ops.reparam_bwd(dz.to(dev()), stats.detach().float().to(dev()), eps.to(dev()), dstats=base.to(dev()).clone())
    close(ds2, gs + base.double(), name="dstats(acc)")


@pytest.mark.parametrize("B", [1, 8, 50, 256])
@pytest.mark.parametrize("sigmoid", [True, False])
@pytest.mark.parametrize("variational,mse_prior", [(True, False), (True, True), (False, False)])
def test_vae_loss_fwd_bwd(B, sigmoid, variational, mse_prior):
    from gmk import ops
    L, beta = 256, 0.05
    pre = T(f"vp{B}", (B, 3, 32, 32), -4, 4).double().requires_grad_(True)
    img = T(f"vi{B}", (B, 3, 32, 32), 0, 1)
    stats = T(f"vs{B}", (B, L, 2), -2, 2).double().requires_grad_(True) if variational else None
    recon = torch.sigmoid(pre) if sigmoid else pre
    rec, kl, total = ae_oracle.train_losses(recon, img.double(), stats, beta, variational=variational, mse_prior=mse_prior)
    grads = torch.autograd.grad(total, (pre, stats) if variational else (pre,))
    losses, dpre, dstats, rout = ops.vae_loss_fwd_bwd(pre.detach().float().to(dev()), img.to(dev()),
                                                      None if stats is None else stats.detach().float().to(dev()),
                                                      beta, sigmoid=sigmoid, mse_prior=mse_prior, want_recon=True)
    lo = losses.cpu().double()
    assert abs(lo[0] - rec.item()) <= 1e-5 * abs(rec.item())
    assert abs(lo[2] - total.item()) <= 1e-5 * abs(total.item())
    if variational:
        assert abs(lo[1] - kl.item()) <= 1e-5 * abs(kl.item())
        close(dstats, grads[1], name="dstats")
    close(dpre, grads[0], name="dpre")
    close(rout, recon, name="recon")


@pytest.mark.parametrize("shape", [(2, 8, 16, 16), (3, 5, 32, 32), (1, 4, 8, 8), (2, 3, 64, 64), (1, 2, 12, 12)])   # 12: the generic (non power-of-two) kernel
def test_maxpool2(shape):
    from gmk import ops
    x = T(f"mp{shape}", shape).double().requires_grad_(True)
    y = F.max_pool2d(x, 2, 2)
    dy = T(f"mpd{shape}", tuple(y.shape))
    (gx,) = torch.autograd.grad(y, x, dy.double())
    dact = T(f"mpa{shape}", shape, 0, 1)
    yg, idx = ops.maxpool2_fwd(x.detach().float().to(dev()))
    close(yg, y, name="y")
    dx = ops.maxpool2_bwd(dy.to(dev()), idx, dact_in=dact.to(dev()))
    close(dx, gx * dact.double(), name="dx")
    # ties (post-ReLU zeros): gradient goes to the first element in scan order, like PyTorch
    xz = torch.zeros(1, 1, 4, 4).requires_grad_(True)
    yz = F.max_pool2d(xz, 2, 2)
    (gz,) = torch.autograd.grad(yz, xz, torch.ones_like(yz))
    _, idz = ops.maxpool2_fwd(xz.detach().to(dev()))
    dz = ops.maxpool2_bwd(torch.ones(1, 1, 2, 2, device=dev()), idz)
    close(dz, gz, name="tie")


@pytest.mark.parametrize("mode", ["nearest", "bilinear", "bicubic"])
@pytest.mark.parametrize("shape", [(2, 8, 8, 8), (1, 3, 16, 16), (2, 2, 4, 4), (2, 3, 32, 32), (1, 2, 6, 6)])   # 6: the generic nearest kernel
def test_upsample2(mode, shape):
    from gmk import ops
    x = T(f"up{shape}", shape).double().requires_grad_(True)
    y = F.interpolate(x, scale_factor=2, mode=mode)
    dy = T(f"upd{shape}", tuple(y.shape))
    (gx,) = torch.autograd.grad(y, x, dy.double())
    dact = T(f"upa{shape}", shape, 0, 1)
    close(ops.upsample2_fwd(x.detach().float().to(dev()), mode), y, name="y")
    close(ops.upsample2_bwd(dy.to(dev()), mode, dact_in=dact.to(dev())), gx * dact.double(), name="dx")
    close(ops.upsample2_bwd(dy.to(dev()), mode), gx, name="dx(no dact)")


# ---------------------------------------------------------------------------------------------- fused VAE tail
@pytest.mark.parametrize("shape", [(256, 4096, 256), (8, 4096, 256), (50, 1024, 32), (4, 64, 8), (300, 96, 20), (256, 16384, 512)])
def test_linear_reparam_head_and_tail(shape):
    """gmk_linear_fwd_reparam / gmk_linear_dgrad_reparam (encoder head + reparameterisation, decoder fc dgrad +
    reparameterisation backward + beta*KL gradient, /root/reference/models/autoencoders.py:57-61,169-176 and
    main.py:36-48) against autograd of the oracle's formulas; split-K shapes and the unsplit fallback, both priors."""
    from gmk import ops
    M, K, L = shape
    x = T(f"rx{shape}", (M, K))
    w = T(f"rw{shape}", (2 * L, K)) / np.sqrt(K)
    b = T(f"rb{shape}", (2 * L,))
    eps = T(f"re{shape}", (M, L), -2, 2)
    stats, z = ops.linear_fwd_reparam(x.to(dev()), w.to(dev()), b.to(dev()), eps.to(dev()))
    ref = (x.double() @ w.double().T + b.double()).view(M, L, 2)
    close(stats, ref, name="stats")
    close(z, ref[:, :, 0] + torch.exp(0.5 * ref[:, :, 1]) * eps.double(), name="z")
    # tail: N = decoder fc width
    for N, msep in ((512, False), (64, True)):
        beta = 0.37
        st = (T(f"rs{shape}{N}", (M, L, 2)) * 0.8).double().requires_grad_(True)
        wd = (T(f"rwd{shape}{N}", (N, L)) / np.sqrt(L)).double()
        dy = T(f"rdy{shape}{N}", (M, N))
        mu, lv = st[:, :, 0], st[:, :, 1]
        zz = mu + torch.exp(0.5 * lv) * eps.double()
        _, kl, _ = ae_oracle.train_losses(torch.zeros(1, dtype=torch.float64), torch.zeros(1, dtype=torch.float64), st, beta,
                                          variational=True, mse_prior=msep)
        obj = ((zz @ wd.T) * dy.double()).sum() + beta * kl      # d obj / d stats = reparam backward of dz + beta * dKL
        (g,) = torch.autograd.grad(obj, st)
        dstats = ops.linear_dgrad_reparam(dy.to(dev()), wd.float().to(dev()), st.detach().float().to(dev()), eps.to(dev()),
                                          beta / M, mse_prior=msep)
        close(dstats, g, name=f"dstats N={N} msep={msep}")


@pytest.mark.parametrize("B,variational", [(256, True), (3, True), (1100, False)])
def test_one_launch_loss_matches_two_launch_form(B, variational):
    """gmk_vae_loss_fwd_bwd_fused (the last block finalizes) == gmk_vae_loss_fwd_bwd, bit for bit, on repeated calls
    (the sync block must come back to zero each time); also with the KL gradient left out (GMK_LOSS_NO_DSTATS)."""
    from gmk import ops
    pre = (T(f"lp{B}", (B, 3, 32, 32)) * 3).to(dev())
    img = T(f"li{B}", (B, 3, 32, 32), 0.0, 1.0).to(dev())
    stats = (T(f"ls{B}", (B, 64, 2)) * 0.5).to(dev()) if variational else None
    ref = ops.vae_loss_fwd_bwd(pre, img, stats, 0.01)
    for rep in range(3):
        got = ops.vae_loss_fwd_bwd(pre, img, stats, 0.01, fused=True, want_dstats=(rep != 1))
        assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]), rep
        if variational and rep != 1:
            assert torch.equal(got[2], ref[2])
        assert got[2] is None or rep != 1


def test_thin_path_switch_agrees():
    """gmk_conv_set_thin_paths: both kernel paths of the 3-channel-side layers give the same results (fp32 rounding)."""
    from gmk import ops, tune
    x = T("tpx", (4, 3, 32, 32)).to(dev())
    w = (T("tpw", (64, 3, 3, 3)) / np.sqrt(27)).to(dev())
    b = T("tpb", (64,)).to(dev())
    h = T("tph", (4, 64, 32, 32)).to(dev())
    w3 = (T("tpw3", (3, 64, 3, 3)) / np.sqrt(576)).to(dev())
    b3 = T("tpb3", (3,)).to(dev())
    dy3 = T("tpd", (4, 3, 32, 32)).to(dev())
    before = tune.get_thin_paths()
    outs = []
    try:
        for choice in ((0, 0), (3, 1)):
            tune.set_thin_paths(*choice)
            assert tune.get_thin_paths() == choice
            y, d = ops.conv2d_fwd(x, w, b, act="gelu", want_dact=True)
            y3 = ops.conv2d_fwd(h, w3, b3)
            dx = ops.conv2d_dgrad(dy3, w3, (4, 64, 32, 32), 1, dact_in=h)
            outs.append([t.clone() for t in (y, d, y3, dx)])
        tune.set_thin_paths(-1, -1)   # negative: unchanged
        assert tune.get_thin_paths() == (3, 1)
    finally:
        tune.set_thin_paths(*before)
    for a, c, name in zip(outs[0], outs[1], ("y", "dact", "y3", "dx")):
        close(c, a.double().cpu(), tol=1e-5, name=name)


def test_tune_thin_paths_picks_and_applies_a_candidate(monkeypatch):
    from gmk import ops, tune
    for k in ("GMK_TUNE", "GMK_THIN_IN", "GMK_THIN_OUTV"):
        monkeypatch.delenv(k, raising=False)
    x = T("ttx", (8, 3, 32, 32)).to(dev())
    w = (T("ttw", (64, 3, 3, 3)) / np.sqrt(27)).to(dev())
    b = T("ttb", (64,)).to(dev())
    before = tune.get_thin_paths()
    try:
        r = tune.tune_thin_paths(lambda: ops.conv2d_fwd(x, w, b, act="gelu", want_dact=True), steps=5, rounds=2, warmup=2)
        assert r["tuned"] and tuple(r["choice"]) in tune.CANDIDATES
        assert tune.get_thin_paths() == tuple(r["choice"])
        assert set(r["ms"]) == {"3,1", "0,0"} and all(v > 0 for v in r["ms"].values())
        monkeypatch.setenv("GMK_TUNE", "0")
        tune.set_thin_paths(3, 1)
        r = tune.tune_thin_paths(lambda: None)
        assert not r["tuned"] and tuple(r["choice"]) == (3, 1)
    finally:
        tune.set_thin_paths(*before)
