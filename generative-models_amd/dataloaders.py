"""Data feeders with the reference's get_cifar() surface (/root/reference/dataloaders.py:8-43).

torchvision is not a dependency.  `get_cifar` reads the CIFAR-10 python-pickle batch files if they
are already under data_dir (the reference downloads them; there is no network here), and otherwise
falls back to synthetic U[0,1) images of the same shape so that main.py can still be exercised.
Augmentation matches the reference's train transform: RandomCrop(32, padding=4) + horizontal flip,
values in [0,1], no normalisation (dataloaders.py:23-29).
"""
import os
import pickle

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset


class _Images(Dataset):
    def __init__(self, images, labels, train):
        self.images, self.labels, self.train = images, labels, train

    def __len__(self):
        return self.images.shape[0]

    def __getitem__(self, i):
        img = self.images[i]
        if self.train:
            padded = torch.nn.functional.pad(img, (4, 4, 4, 4))
            dy, dx = np.random.randint(0, 9, size=2)
            img = padded[:, dy:dy + 32, dx:dx + 32]
            if np.random.rand() < 0.5:
                img = img.flip(-1)
        return img, self.labels[i]


class GpuCifarLoader:
    """GPU-resident batch feeder: the whole dataset lives in HBM as uint8 (150 MB for CIFAR-10
    train), every batch is assembled by one gmk_cifar_batch launch (gather by shuffled index + ToTensor + the
    reference's RandomCrop(32, padding=4) / horizontal flip, dataloaders.py:23-29), so the train loop never waits
    for the single-process PIL pipeline of the reference.  Iterates like the reference's DataLoader:
    `for image, label in loader`, `len(loader)`; with world > 1 every rank takes its own contiguous shard of one
    shared epoch permutation (equal local batches, as the DP step needs)."""

    def __init__(self, images_u8, labels, batch_size, train, device, rank=0, world=1, seed=0, shuffle=True):
        if images_u8.dtype != torch.uint8 or images_u8.dim() != 4:
            raise ValueError("GpuCifarLoader needs a (n,C,H,W) uint8 tensor")
        self.data = images_u8.contiguous().to(device)
        self.labels = labels.to(device)
        self.batch_size, self.train, self.device = int(batch_size), bool(train), torch.device(device)
        self.rank, self.world, self.seed, self.shuffle = rank, world, int(seed), shuffle
        self.epoch = 0
        n = self.data.shape[0]
        self.per_rank = n // world  # equal shards; the tail that does not divide is dropped like a DistributedSampler would pad
        self._gen = torch.Generator(device="cpu")

    def __len__(self):
        return (self.per_rank + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        from gmk import lib, check
        n = self.data.shape[0]
        self._gen.manual_seed(self.seed * 1000003 + self.epoch)  # the same permutation on every rank
        perm = torch.randperm(n, generator=self._gen) if self.shuffle else torch.arange(n)
        shard = perm[self.rank * self.per_rank:(self.rank + 1) * self.per_rank].to(self.device)
        epoch = self.epoch
        self.epoch += 1
        _, C, H, W = self.data.shape
        for i, start in enumerate(range(0, self.per_rank, self.batch_size)):
            idx = shard[start:start + self.batch_size].contiguous()
            out = torch.empty((idx.numel(), C, H, W), device=self.device, dtype=torch.float32)
            seed = ((self.seed * 0x9E3779B97F4A7C15) ^ (epoch << 40) ^ (self.rank << 24) ^ i) & 0xFFFFFFFFFFFFFFFF
            check(lib().gmk_cifar_batch(self.data.data_ptr(), idx.data_ptr(), out.data_ptr(), idx.numel(), C, H, W,
                                        4, 1 if self.train else 0, seed, torch.cuda.current_stream().cuda_stream))
            yield out, self.labels[idx]


def _load_cifar_batches(root, names):
    xs, ys = [], []
    for n in names:
        with open(os.path.join(root, n), 'rb') as f:
            d = pickle.load(f, encoding='latin1')  # the user's own dataset files
        xs.append(np.asarray(d['data'], dtype=np.uint8).reshape(-1, 3, 32, 32))
        ys.append(np.asarray(d['labels'], dtype=np.int64))
    return torch.from_numpy(np.concatenate(xs)), torch.from_numpy(np.concatenate(ys))


def get_cifar(dataset='CIFAR10', data_dir=None, num_samples=None, train_batch_size=None, test_batch_size=None,
              synthetic_size=(50000, 10000), device=None, rank=0, world=1, seed=0):
    """Signature of the reference (dataloaders.py:8: dataset, data_dir, num_samples, train_batch_size, test_batch_size, same
    order and defaults; `num_samples` only acts in the reference's MNIST branch, dataloaders.py:14-18, and is accepted and
    ignored here as it is there for CIFAR-10) plus keyword-only extras of this build.
    device = a CUDA device -> GPU-resident loaders (GpuCifarLoader); None / CPU -> torch DataLoaders over CPU tensors
    (what the reference returns, dataloaders.py:37-43)."""
    data_dir = 'data' if data_dir is None else data_dir
    train_batch_size = 50 if train_batch_size is None else train_batch_size     # arguments.py:16-17 defaults
    test_batch_size = 500 if test_batch_size is None else test_batch_size
    if str(dataset).lower() not in ('cifar', 'cifar10'):
        raise NotImplementedError(f'dataset {dataset!r}: only CIFAR-10 is on the path (the reference\'s MNIST branch is unused)')
    root = os.path.join(data_dir, 'cifar-10-batches-py')
    if os.path.isdir(root):
        xtr, ytr = _load_cifar_batches(root, [f'data_batch_{i}' for i in range(1, 6)])
        xte, yte = _load_cifar_batches(root, ['test_batch'])
    else:
        print(f'\n\nCIFAR-10 not found under {root}: using synthetic uniform uint8 images of the same shape\n\n')
        g = torch.Generator().manual_seed(1234)
        xtr = torch.randint(0, 256, (synthetic_size[0], 3, 32, 32), generator=g, dtype=torch.uint8)
        ytr = torch.randint(0, 10, (synthetic_size[0],), generator=g)
        xte = torch.randint(0, 256, (synthetic_size[1], 3, 32, 32), generator=g, dtype=torch.uint8)
        yte = torch.randint(0, 10, (synthetic_size[1],), generator=g)
    if device is not None and torch.device(device).type == 'cuda':
        return (GpuCifarLoader(xtr, ytr, train_batch_size, True, device, rank, world, seed),
                GpuCifarLoader(xte, yte, test_batch_size, False, device, 0, 1, seed, shuffle=False))  # dataloaders.py:41
    xtr, xte = xtr.float().div_(255.0), xte.float().div_(255.0)
    train_dataloader = DataLoader(_Images(xtr, ytr, True), batch_size=train_batch_size, shuffle=True)
    test_dataloader = DataLoader(_Images(xte, yte, False), batch_size=test_batch_size, shuffle=False)  # dataloaders.py:41
    return train_dataloader, test_dataloader
