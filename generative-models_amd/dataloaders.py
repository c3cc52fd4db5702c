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


def _load_cifar_batches(root, names):
    xs, ys = [], []
    for n in names:
        with open(os.path.join(root, n), 'rb') as f:
            d = pickle.load(f, encoding='latin1')  # the user's own dataset files
        xs.append(np.asarray(d['data'], dtype=np.uint8).reshape(-1, 3, 32, 32))
        ys.append(np.asarray(d['labels'], dtype=np.int64))
    return torch.from_numpy(np.concatenate(xs)).float().div_(255.0), torch.from_numpy(np.concatenate(ys))


def get_cifar(dataset='cifar', data_dir='data', train_batch_size=50, test_batch_size=500, synthetic_size=(50000, 10000)):
    root = os.path.join(data_dir, 'cifar-10-batches-py')
    if os.path.isdir(root):
        xtr, ytr = _load_cifar_batches(root, [f'data_batch_{i}' for i in range(1, 6)])
        xte, yte = _load_cifar_batches(root, ['test_batch'])
    else:
        print(f'\n\nCIFAR-10 not found under {root}: using synthetic U[0,1) images of the same shape\n\n')
        g = torch.Generator().manual_seed(1234)
        xtr = torch.rand(synthetic_size[0], 3, 32, 32, generator=g)
        ytr = torch.randint(0, 10, (synthetic_size[0],), generator=g)
        xte = torch.rand(synthetic_size[1], 3, 32, 32, generator=g)
        yte = torch.randint(0, 10, (synthetic_size[1],), generator=g)
    train_dataloader = DataLoader(_Images(xtr, ytr, True), batch_size=train_batch_size, shuffle=True)
    test_dataloader = DataLoader(_Images(xte, yte, False), batch_size=test_batch_size, shuffle=True)
    return train_dataloader, test_dataloader
