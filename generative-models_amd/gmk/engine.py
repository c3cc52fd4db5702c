"""Explicit forward / backward of the reference Encoder and Decoder on libgmk kernels.

The layer sequences follow /root/reference/models/autoencoders.py:44-65 (Encoder.forward) and
:93-115 (Decoder.forward).  Every conv/linear kernel fuses bias + activation and, in training
mode, also stores act'(pre-activation); the backward pass multiplies by that derivative in the
epilogue of the kernel that produces the incoming gradient, so no activation is ever evaluated
twice and no elementwise kernel runs on its own.

`p` is a dict of parameter tensors keyed by the module-local names (conv1.weight, ...).
Gradients are returned as a dict with the same keys.
"""
import torch

from . import ops


def _wg(side, fn, *tensors):
    """Weight-gradient launch: on the side queue when there is one (overlaps the dgrad chain), else inline."""
    if side is None:
        return fn()
    return side.run(fn, *tensors)


def _o(gout, name):
    """(dw, db) destination tensors for layer `name`, or None to allocate."""
    if gout is None:
        return None
    return gout[name + ".weight"], gout[name + ".bias"]


# ------------------------------------------------------------------------------ encoder
def encoder_forward(p, x, *, act, no_pool, ksize, train, reparam_eps=None):
    """Returns (out (B, mult*L), saved).  x: (B,3,H,W) contiguous fp32 CUDA.
    reparam_eps (B, L): VAE head fused with the reparameterisation (gmk_linear_fwd_reparam): out is then the
    interleaved stats (B, L, 2) and saved["z"] the latent z = mu + sigma * eps."""
    sv = {"x": x}
    conv = ops.conv2d_fwd

    def cv(inp, name, stride):
        r = conv(inp, p[name + ".weight"], p[name + ".bias"], stride=stride, act=act, want_dact=train)
        return r if train else (r, None)

    a1, d1 = cv(x, "conv1", 1)
    if no_pool:
        a2, d2 = cv(a1, "downsample1", 2)
        i1 = None
    else:
        a2, i1 = ops.maxpool2_fwd(a1, want_idx=True)
        d2 = None
    a3, d3 = cv(a2, "conv2", 1)
    if no_pool:
        a4, d4 = cv(a3, "downsample2", 2)
        i2 = None
    else:
        a4, i2 = ops.maxpool2_fwd(a3, want_idx=True)
        d4 = None
    a5, d5 = cv(a4, "conv3", 1)
    B = x.shape[0]
    flat = a5.view(B, -1)
    if reparam_eps is not None:
        out, sv["z"] = ops.linear_fwd_reparam(flat, p["fc.weight"], p["fc.bias"], reparam_eps)
    else:
        out = ops.linear_fwd(flat, p["fc.weight"], p["fc.bias"])
    if train:
        sv.update(a1=a1, d1=d1, a2=a2, d2=d2, i1=i1, a3=a3, d3=d3, a4=a4, d4=d4, i2=i2, a5=a5, d5=d5)
    return out, sv


def encoder_backward(p, sv, dout, *, no_pool, ksize, gout=None, on_fc_done=None, side=None, need_dx=False, defer=None):
    """dout: (B, mult*L) gradient of the encoder output.  Returns param grads (+ "__dx__", the gradient w.r.t. the
    input image, when need_dx).  defer: list collecting the layers whose slab reduction is left to ONE later
    ops.wgrad_reduce_batch() launch (train_step: per gradient bucket)."""
    g = {}
    B = dout.shape[0]
    x = sv["x"]
    a5 = sv["a5"]
    g["fc.weight"], g["fc.bias"] = _wg(side, lambda: ops.linear_wgrad(dout, a5.view(B, -1), out=_o(gout, "fc")), dout)
    dz5 = ops.linear_dgrad(dout, p["fc.weight"], dact_in=sv["d5"].view(B, -1)).view_as(a5)
    if on_fc_done is not None:
        # fc gradients are enqueued and the last reader of fc.weight (the dgrad above) is too: the bucket's all-reduce
        # -- and its early AdamW update -- can start under the conv backward
        on_fc_done()
    g["conv3.weight"], g["conv3.bias"] = _wg(side, lambda: ops.conv2d_wgrad(dz5, sv["a4"], ksize, 1, out=_o(gout, "conv3"), defer=defer), dz5)
    if no_pool:
        dz4 = ops.conv2d_dgrad(dz5, p["conv3.weight"], tuple(sv["a4"].shape), 1, dact_in=sv["d4"])
        g["downsample2.weight"], g["downsample2.bias"] = _wg(side, lambda: ops.conv2d_wgrad(dz4, sv["a3"], ksize, 2, out=_o(gout, "downsample2"), defer=defer), dz4)
        dz3 = ops.conv2d_dgrad(dz4, p["downsample2.weight"], tuple(sv["a3"].shape), 2, dact_in=sv["d3"])
    else:
        da4 = ops.conv2d_dgrad(dz5, p["conv3.weight"], tuple(sv["a4"].shape), 1)
        dz3 = ops.maxpool2_bwd(da4, sv["i2"], dact_in=sv["d3"])
    g["conv2.weight"], g["conv2.bias"] = _wg(side, lambda: ops.conv2d_wgrad(dz3, sv["a2"], ksize, 1, out=_o(gout, "conv2"), defer=defer), dz3)
    if no_pool:
        dz2 = ops.conv2d_dgrad(dz3, p["conv2.weight"], tuple(sv["a2"].shape), 1, dact_in=sv["d2"])
        g["downsample1.weight"], g["downsample1.bias"] = _wg(side, lambda: ops.conv2d_wgrad(dz2, sv["a1"], ksize, 2, out=_o(gout, "downsample1"), defer=defer), dz2)
        dz1 = ops.conv2d_dgrad(dz2, p["downsample1.weight"], tuple(sv["a1"].shape), 2, dact_in=sv["d1"])
    else:
        da2 = ops.conv2d_dgrad(dz3, p["conv2.weight"], tuple(sv["a2"].shape), 1)
        dz1 = ops.maxpool2_bwd(da2, sv["i1"], dact_in=sv["d1"])
    g["conv1.weight"], g["conv1.bias"] = _wg(side, lambda: ops.conv2d_wgrad(dz1, x, ksize, 1, out=_o(gout, "conv1"), defer=defer), dz1)
    if need_dx:
        g["__dx__"] = ops.conv2d_dgrad(dz1, p["conv1.weight"], tuple(x.shape), 1)
    return g


# ------------------------------------------------------------------------------ decoder
def decoder_forward(p, z, *, act, num_channels, no_upsample, interpolation, ksize, base, train, final_act):
    """final_act: None -> return the pre-sigmoid output of conv3 ('pre'); 'sigmoid' -> sigmoid fused
    into conv3's epilogue (its derivative saved when train).  Returns (out, saved)."""
    sv = {"z": z}
    B = z.shape[0]
    C = num_channels

    def cv(inp, name):
        r = ops.conv2d_fwd(inp, p[name + ".weight"], p[name + ".bias"], stride=1, act=act, want_dact=train)
        return r if train else (r, None)

    def dc(inp, name):
        r = ops.deconv2d_fwd(inp, p[name + ".weight"], p[name + ".bias"], act=act, want_dact=train)
        return r if train else (r, None)

    r = ops.linear_fwd(z, p["fc.weight"], p["fc.bias"], act=act, want_dact=train)
    h0, e0 = r if train else (r, None)
    h0v = h0.view(B, C, base, base)
    h1, e1 = cv(h0v, "conv1")
    if no_upsample:
        h2, e2 = dc(h1, "deconv1")
    else:
        h2, e2 = ops.upsample2_fwd(h1, interpolation), None
    h3, e3 = cv(h2, "conv2")
    if no_upsample:
        h4, e4 = dc(h3, "deconv2")
    else:
        h4, e4 = ops.upsample2_fwd(h3, interpolation), None
    r = ops.conv2d_fwd(h4, p["conv3.weight"], p["conv3.bias"], stride=1, act=final_act,
                       want_dact=(train and final_act is not None))
    out, e5 = r if (train and final_act is not None) else (r, None)
    if train:
        sv.update(h0v=h0v, e0=e0, h1=h1, e1=e1, h2=h2, e2=e2, h3=h3, e3=e3, h4=h4, e4=e4, e5=e5)
    return out, sv


def decoder_backward(p, sv, dpre, *, no_upsample, interpolation, ksize, need_dz=True, gout=None, side=None, reparam=None, defer=None):
    """dpre: gradient w.r.t. conv3's pre-activation output (B,3,H,W).  Returns (param grads, dz).
    reparam = (stats, eps, kl_scale, mse_prior): the fc dgrad is fused with the reparameterisation backward and the
    beta*KL gradient (gmk_linear_dgrad_reparam) and the second return value is dstats (B, L, 2) instead of dz."""
    g = {}
    B = dpre.shape[0]
    g["conv3.weight"], g["conv3.bias"] = _wg(side, lambda: ops.conv2d_wgrad(dpre, sv["h4"], ksize, 1, out=_o(gout, "conv3"), defer=defer), dpre)
    if no_upsample:
        dz4 = ops.conv2d_dgrad(dpre, p["conv3.weight"], tuple(sv["h4"].shape), 1, dact_in=sv["e4"])
        g["deconv2.weight"], g["deconv2.bias"] = _wg(side, lambda: ops.deconv2d_wgrad(dz4, sv["h3"], ksize, out=_o(gout, "deconv2"), defer=defer), dz4)
        dz3 = ops.deconv2d_dgrad(dz4, p["deconv2.weight"], tuple(sv["h3"].shape), dact_in=sv["e3"])
    else:
        dh4 = ops.conv2d_dgrad(dpre, p["conv3.weight"], tuple(sv["h4"].shape), 1)
        dz3 = ops.upsample2_bwd(dh4, interpolation, dact_in=sv["e3"])
    g["conv2.weight"], g["conv2.bias"] = _wg(side, lambda: ops.conv2d_wgrad(dz3, sv["h2"], ksize, 1, out=_o(gout, "conv2"), defer=defer), dz3)
    if no_upsample:
        dz2 = ops.conv2d_dgrad(dz3, p["conv2.weight"], tuple(sv["h2"].shape), 1, dact_in=sv["e2"])
        g["deconv1.weight"], g["deconv1.bias"] = _wg(side, lambda: ops.deconv2d_wgrad(dz2, sv["h1"], ksize, out=_o(gout, "deconv1"), defer=defer), dz2)
        dz1 = ops.deconv2d_dgrad(dz2, p["deconv1.weight"], tuple(sv["h1"].shape), dact_in=sv["e1"])
    else:
        dh2 = ops.conv2d_dgrad(dz3, p["conv2.weight"], tuple(sv["h2"].shape), 1)
        dz1 = ops.upsample2_bwd(dh2, interpolation, dact_in=sv["e1"])
    g["conv1.weight"], g["conv1.bias"] = _wg(side, lambda: ops.conv2d_wgrad(dz1, sv["h0v"], ksize, 1, out=_o(gout, "conv1"), defer=defer), dz1)
    dz0 = ops.conv2d_dgrad(dz1, p["conv1.weight"], tuple(sv["h0v"].shape), 1, dact_in=sv["e0"].view_as(sv["h0v"]))
    dz0 = dz0.view(B, -1)
    g["fc.weight"], g["fc.bias"] = _wg(side, lambda: ops.linear_wgrad(dz0, sv["z"], out=_o(gout, "fc")), dz0)
    if reparam is not None:
        stats, eps, kl_scale, mse_prior = reparam
        return g, ops.linear_dgrad_reparam(dz0, p["fc.weight"], stats, eps, kl_scale, mse_prior)
    dz = ops.linear_dgrad(dz0, p["fc.weight"]) if need_dz else None
    return g, dz
