"""CPU restatement of the GPU-resident CIFAR feeder (TEST INFRASTRUCTURE; numpy integer / byte arithmetic).

The transform is the reference's train pipeline /root/reference/dataloaders.py:23-29 -- RandomCrop(32, padding=4),
RandomHorizontalFlip, ToTensor (uint8 -> fp32 / 255, no normalisation) -- and its test pipeline (:32-35, ToTensor
only).  The reference draws (dy, dx, flip) from torch's unseeded global RNG ("no seeding anywhere"), so
there is no reference random stream to reproduce: the draw is the build's own counter-based hash, documented in
include/gmk.h (gmk_cifar_batch) and restated here.  Parity pinned against torch's own crop / flip ops on the same
(dy, dx, flip) in tests/test_feeder.py; the GPU kernel must match this file bit for bit.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import oracle/.
"""
import numpy as np

_M32 = np.uint64(0xFFFFFFFF)


def mix32(x):
    x = np.asarray(x, dtype=np.uint64) & _M32
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7FEB352D)) & _M32
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x846CA68B)) & _M32
    x ^= x >> np.uint64(16)
    return x


def draws(B, pad, seed):
    """(dy, dx, flip) int arrays for batch positions 0..B-1."""
    lo, hi = np.uint64(seed & 0xFFFFFFFF), np.uint64((seed >> 32) & 0xFFFFFFFF)
    b = np.arange(B, dtype=np.uint64)
    h = mix32(mix32((b * np.uint64(0x9E3779B1) + lo) & _M32) + ((hi * np.uint64(0x85EBCA6B)) & _M32))
    span = np.uint64(2 * pad + 1)
    return (h % span).astype(np.int64), ((h >> np.uint64(8)) % span).astype(np.int64), ((h >> np.uint64(16)) & np.uint64(1)).astype(np.int64)


def cifar_batch(data_u8, index, pad=4, augment=True, seed=0):
    """data_u8 (n,C,H,W) uint8, index (B,) -> (B,C,H,W) float32."""
    x = data_u8[np.asarray(index)]
    B, C, H, W = x.shape
    if augment:
        dy, dx, flip = draws(B, pad, seed)
        padded = np.zeros((B, C, H + 2 * pad, W + 2 * pad), np.uint8)
        padded[:, :, pad:pad + H, pad:pad + W] = x
        out = np.empty_like(x)
        for b in range(B):
            crop = padded[b, :, dy[b]:dy[b] + H, dx[b]:dx[b] + W]     # RandomCrop(H, padding=pad), dataloaders.py:25
            out[b] = crop[:, :, ::-1] if flip[b] else crop           # RandomHorizontalFlip, :26
        x = out
    return x.astype(np.float32) / np.float32(255.0)                   # ToTensor, :27
