"""Helpers with the reference's names (/root/reference/utils.py); plotting is optional here
(matplotlib is imported lazily; torchvision is not required)."""
import os

import torch
import torch.nn as nn


def print_debug(x, debug=False, name=''):
    """--debug trace line: shape and the first eight values (format of reference utils.py:13-15)."""
    if not debug:
        return
    print(f'{name} shape: {list(x.shape)}\n\tvalues: {x.flatten()[:8]}')


def make_folder(name):
    os.makedirs(name, exist_ok=True)


def _make_grid(images, nrow=8, padding=2, normalize=False):
    """Minimal stand-in for torchvision.utils.make_grid (N,C,H,W) -> (C, H', W')."""
    images = images.detach().cpu().float()
    if normalize:
        lo, hi = images.min(), images.max()
        images = (images - lo) / (hi - lo).clamp_min(1e-5)
    n, c, h, w = images.shape
    ncol = min(nrow, n)
    nrw = (n + ncol - 1) // ncol
    grid = torch.zeros(c, nrw * (h + padding) + padding, ncol * (w + padding) + padding)
    for i in range(n):
        r, q = divmod(i, ncol)
        y0, x0 = padding + r * (h + padding), padding + q * (w + padding)
        grid[:, y0:y0 + h, x0:x0 + w] = images[i]
    return grid


def save_image_grid(images, path, nrow=4, figsize=(7, 4.5), normalize=False):
    import matplotlib
    matplotlib.use('Agg')
    import matplotlib.pyplot as plt
    grid = _make_grid(images, nrow=nrow, normalize=normalize).permute(1, 2, 0).clamp(0, 1)
    plt.figure(figsize=figsize)
    plt.imshow(grid)
    plt.axis('off')
    os.makedirs(os.path.dirname(path) or '.', exist_ok=True)
    plt.savefig(path)
    plt.close()


def plot_image(image, name='image.png'):
    image = image.squeeze(0)
    image = image - image.min()
    image = image / image.max()
    save_image_grid(image.unsqueeze(0), 'plots/' + name, nrow=1)


def plot_grid(model, input_images=None, name=''):
    model.eval()
    with torch.no_grad():
        reconstructed_images = model(input_images.to(next(model.parameters()).device))
    stack = torch.stack([input_images.cpu(), reconstructed_images.cpu()], dim=1).flatten(0, 1)
    save_image_grid(stack, f'plots/{name}.png', nrow=input_images.shape[0], normalize=True)


_ACTIVATIONS = {
    'relu': nn.ReLU, 'leaky-relu': nn.LeakyReLU, 'elu': nn.ELU, 'selu': nn.SELU,
    'gelu': nn.GELU, 'swish': nn.SiLU, 'mish': nn.Mish,
}


def get_activation_function(act_str):
    """--act string -> nn.Module instance (same seven names as the reference, utils.py:49-66); the modules are
    only tokens for gmk.act_id, the kernels apply the activation in their epilogues."""
    try:
        return _ACTIVATIONS[act_str]()
    except KeyError:
        raise NotImplementedError(act_str) from None
