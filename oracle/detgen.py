"""Closed-form deterministic tensors for parity tests (TEST INFRASTRUCTURE).

Every value is a pure function of (tensor name, flat index), computed with integer
arithmetic in numpy uint64 so that this container, the GPU box and the golden generator
(oracle/gen_golden.py, which feeds the same tensors to the imported reference) all see
bit-identical inputs without shipping multi-MB weight files.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import oracle/.
"""
import zlib

import numpy as np

_M32 = np.uint64(0xFFFFFFFF)


def _mix(x):
    # 32-bit avalanche hash (lowbias32 constants), evaluated in uint64 and masked.
    x = x & _M32
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7FEB352D)) & _M32
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x846CA68B)) & _M32
    x ^= x >> np.uint64(16)
    return x


def uniform(name, shape, lo=-1.0, hi=1.0):
    """float32 array, U[lo,hi) with 24-bit resolution, keyed by `name`."""
    n = int(np.prod(shape)) if len(shape) else 1
    seed = np.uint64(zlib.crc32(name.encode()))
    idx = np.arange(n, dtype=np.uint64)
    h = _mix(idx * np.uint64(0x9E3779B1) + seed)
    h = _mix(h + seed * np.uint64(0x85EBCA6B))
    u = (h >> np.uint64(8)).astype(np.float64) / float(1 << 24)  # [0,1)
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def normal(name, shape):
    """float32 approx N(0,1) (Box-Muller on two keyed uniforms)."""
    u1 = uniform(name + "/u1", shape, 0.0, 1.0).astype(np.float64)
    u2 = uniform(name + "/u2", shape, 0.0, 1.0).astype(np.float64)
    u1 = np.maximum(u1, 2.0 ** -24)
    z = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
    return z.astype(np.float32)


def ae_param_shapes(variational, latent_size, num_channels, kernel_size, no_pool, no_upsample, base=8):
    """state_dict key -> shape, in the reference's registration order
    (/root/reference/models/autoencoders.py:33-41, 81-87)."""
    C, k, L = num_channels, kernel_size, latent_size
    mult = 2 if variational else 1
    s = {}
    s["encoder.conv1.weight"] = (C, 3, k, k); s["encoder.conv1.bias"] = (C,)
    s["encoder.conv2.weight"] = (C, C, k, k); s["encoder.conv2.bias"] = (C,)
    s["encoder.conv3.weight"] = (C, C, k, k); s["encoder.conv3.bias"] = (C,)
    if no_pool:
        s["encoder.downsample1.weight"] = (C, C, k, k); s["encoder.downsample1.bias"] = (C,)
        s["encoder.downsample2.weight"] = (C, C, k, k); s["encoder.downsample2.bias"] = (C,)
    s["encoder.fc.weight"] = (mult * L, base * base * C); s["encoder.fc.bias"] = (mult * L,)
    s["decoder.fc.weight"] = (C * base * base, L); s["decoder.fc.bias"] = (C * base * base,)
    s["decoder.conv1.weight"] = (C, C, k, k); s["decoder.conv1.bias"] = (C,)
    s["decoder.conv2.weight"] = (C, C, k, k); s["decoder.conv2.bias"] = (C,)
    s["decoder.conv3.weight"] = (3, C, k, k); s["decoder.conv3.bias"] = (3,)
    if no_upsample:
        s["decoder.deconv1.weight"] = (C, C, k, k); s["decoder.deconv1.bias"] = (C,)
        s["decoder.deconv2.weight"] = (C, C, k, k); s["decoder.deconv2.bias"] = (C,)
    return s


def ae_params(tag, **cfg):
    """Deterministic parameters scaled like PyTorch's default init (U(+-1/sqrt(fan_in)))."""
    out = {}
    for key, shp in ae_param_shapes(**cfg).items():
        if key.endswith(".weight"):
            if "deconv" in key:
                fan_in = shp[1] * shp[2] * shp[3]  # ConvTranspose2d: fan_in from dim 1
            else:
                fan_in = int(np.prod(shp[1:]))
            wkey = key
        else:
            wshape = ae_param_shapes(**cfg)[key[:-4] + "weight"]
            fan_in = wshape[1] * wshape[2] * wshape[3] if "deconv" in key else int(np.prod(wshape[1:]))
        b = 1.0 / np.sqrt(fan_in)
        out[key] = uniform(f"{tag}/{key}", shp, -b, b)
    return out


def images(tag, n, hw=32):
    return uniform(f"{tag}/images", (n, 3, hw, hw), 0.0, 1.0)


def classifier_param_shapes(latent_size):
    """state_dict key -> shape of the reference Classifier, in registration order
    (/root/reference/models/classifiers.py:15-26)."""
    L = latent_size
    s = {}
    s["conv1.weight"] = (64, 3, 5, 5); s["conv1.bias"] = (64,)
    s["conv2.weight"] = (128, 64, 5, 5); s["conv2.bias"] = (128,)
    s["conv3.weight"] = (256, 128, 3, 3); s["conv3.bias"] = (256,)
    s["fc.weight"] = (L, 256 * 4 * 4); s["fc.bias"] = (L,)
    s["fc_out.weight"] = (10, L); s["fc_out.bias"] = (10,)
    for name, c in (("bn1", 64), ("bn2", 128), ("bn3", 256), ("bn4", L)):
        s[name + ".weight"] = (c,); s[name + ".bias"] = (c,)
        s[name + ".running_mean"] = (c,); s[name + ".running_var"] = (c,)
        s[name + ".num_batches_tracked"] = ()
    return s


def classifier_params(tag, latent_size):
    """Deterministic Classifier state_dict: conv/linear like PyTorch's default init, BatchNorm with non-trivial
    affine parameters and running statistics (so that eval-mode folding is actually exercised)."""
    shapes = classifier_param_shapes(latent_size)
    out = {}
    for key, shp in shapes.items():
        mod, leaf = key.rsplit(".", 1)
        if mod.startswith("bn"):
            if leaf == "weight":
                out[key] = uniform(f"{tag}/{key}", shp, 0.5, 1.5)
            elif leaf == "bias":
                out[key] = uniform(f"{tag}/{key}", shp, -0.2, 0.2)
            elif leaf == "running_mean":
                out[key] = uniform(f"{tag}/{key}", shp, -0.1, 0.1)
            elif leaf == "running_var":
                out[key] = uniform(f"{tag}/{key}", shp, 0.5, 1.5)
            else:
                out[key] = np.array(7, dtype=np.int64)
        else:
            fan_in = int(np.prod(shapes[mod + ".weight"][1:]))
            b = 1.0 / np.sqrt(fan_in)
            if mod == "fc":
                b *= 6.0    # spread the features / logits out so that the metrics are not degenerate
            if mod == "fc_out":
                b *= 25.0
            out[key] = uniform(f"{tag}/{key}", shp, -b, b)
    return out
