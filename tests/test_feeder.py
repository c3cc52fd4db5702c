"""GPU-resident CIFAR feeder: the numpy oracle against torch's own crop / flip ops (CPU), the HIP
kernel against the oracle bit for bit (GPU), and the host loader's sharding / epoch contract."""
import numpy as np
import pytest
import torch

from oracle import detgen, feeder_oracle


def _dataset(n, tag="feed"):
    return (detgen.uniform(tag, (n, 3, 32, 32), 0.0, 256.0)).astype(np.uint8)


def test_oracle_matches_torch_crop_flip():
    """The oracle's transform == pad(4) -> crop at (dy,dx) -> flip, done with torch ops on ToTensor'd images, for the
    oracle's own draws (dataloaders.py:23-29 semantics; the reference's draw itself is unseeded torch RNG)."""
    data = _dataset(64)
    idx = np.arange(64)[::-1].copy()
    out = feeder_oracle.cifar_batch(data, idx, pad=4, augment=True, seed=0x1234567890ABCDEF)
    dy, dx, flip = feeder_oracle.draws(64, 4, 0x1234567890ABCDEF)
    assert dy.min() >= 0 and dy.max() <= 8 and dx.min() >= 0 and dx.max() <= 8 and set(flip.tolist()) == {0, 1}
    x = torch.from_numpy(data[idx]).float().div(255.0)          # ToTensor
    padded = torch.nn.functional.pad(x, (4, 4, 4, 4))
    for b in range(64):
        ref = padded[b, :, dy[b]:dy[b] + 32, dx[b]:dx[b] + 32]
        if flip[b]:
            ref = ref.flip(-1)
        assert np.array_equal(out[b], ref.numpy())
    plain = feeder_oracle.cifar_batch(data, idx, augment=False)
    assert np.array_equal(plain, x.numpy())


def test_draws_are_roughly_uniform():
    dy, dx, flip = feeder_oracle.draws(90000, 4, 77)
    for d in (dy, dx):
        cnt = np.bincount(d, minlength=9)
        assert cnt.min() > 9000 and cnt.max() < 11000
    assert abs(flip.mean() - 0.5) < 0.01


@pytest.mark.gpu
@pytest.mark.parametrize("B,augment,seed", [(1, 1, 0), (50, 1, 3), (256, 1, 0xFEDCBA9876543210), (256, 0, 5), (777, 1, 1 << 40)])
def test_kernel_matches_oracle_bit_exact(B, augment, seed):
    from gmk import lib, check
    data = _dataset(1000)
    idx = (detgen.uniform(f"feed/idx{B}", (B,), 0.0, 1000.0)).astype(np.int64)
    d = torch.from_numpy(data).cuda()
    i = torch.from_numpy(idx).cuda()
    out = torch.empty(B, 3, 32, 32, device="cuda")
    check(lib().gmk_cifar_batch(d.data_ptr(), i.data_ptr(), out.data_ptr(), B, 3, 32, 32, 4, augment, seed,
                                torch.cuda.current_stream().cuda_stream))
    ref = feeder_oracle.cifar_batch(data, idx, pad=4, augment=bool(augment), seed=seed)
    assert np.array_equal(out.cpu().numpy(), ref)


@pytest.mark.gpu
def test_loader_contract_and_sharding():
    from dataloaders import GpuCifarLoader
    data = torch.from_numpy(_dataset(1003))
    labels = torch.arange(1003) % 10
    # test loader: no augmentation, every sample exactly once, values = u8/255
    te = GpuCifarLoader(data, labels, 100, False, "cuda:0", shuffle=False)
    assert len(te) == 11
    got = torch.cat([x for x, _ in te]).cpu()
    assert torch.equal(got, data.float().div(255.0))
    # two ranks: equal shards of one shared permutation, disjoint, epoch changes the order
    loaders = [GpuCifarLoader(data, torch.arange(1003), 64, True, "cuda:0", rank=r, world=2, seed=9) for r in range(2)]
    assert len(loaders[0]) == len(loaders[1]) == 8
    seen = [torch.cat([y for _, y in ld]).cpu() for ld in loaders]
    assert seen[0].numel() == seen[1].numel() == 501
    assert len(set(seen[0].tolist()) & set(seen[1].tolist())) == 0
    again = torch.cat([y for _, y in loaders[0]]).cpu()
    assert not torch.equal(again, seen[0])
    x, y = next(iter(loaders[0]))
    assert x.shape == (64, 3, 32, 32) and x.dtype == torch.float32 and x.is_cuda and float(x.max()) <= 1.0
