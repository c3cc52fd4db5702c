"""CIFAR-10 Classifier with the reference's Python surface (feature / logit provider for the metrics).

Mirrors /root/reference/models/classifiers.py: same class name and module path (the reference stores the
classifier as a full-model pickle, `models.classifiers.Classifier`, classifiers.py:111), same constructor
(`Classifier(args)`), same child modules / parameter / buffer names, `.latent_vector` after forward.

Two forward paths, chosen by the module's mode:

* eval mode (what metrics.compute_fid / compute_inception_score / compute_confidence_diversity and
  evaluate_classifier run): hand-written gfx950 kernels through libgmk -- BatchNorm is folded
  into the preceding conv / linear weights, the activation into the conv / GEMM epilogue, so the whole forward is
  3 implicit-GEMM convs, 3 max-pools and 2 GEMMs.  CUDA tensors only (no CPU path).
* train mode (train_classifier, an auxiliary model's training loop: out of the hot-path scope):
  ordinary PyTorch modules under autograd.
"""
import numpy as np
import torch
import torch.nn as nn

from gmk import ops, act_id
from utils import print_debug


class Classifier(nn.Module):
    def __init__(self, args=None) -> None:
        super().__init__()
        self.debug = args.debug
        self.latent_size = args.latent_size
        self.conv1 = nn.Conv2d(in_channels=3, out_channels=64, kernel_size=5, stride=1, padding=2)
        self.conv2 = nn.Conv2d(in_channels=64, out_channels=128, kernel_size=5, stride=1, padding=2)
        self.conv3 = nn.Conv2d(in_channels=128, out_channels=256, kernel_size=3, stride=1, padding=1)
        self.max_pool = nn.MaxPool2d(2, 2)
        self.fc = nn.Linear(in_features=256 * 4 * 4, out_features=self.latent_size)
        self.fc_out = nn.Linear(in_features=self.latent_size, out_features=10)
        self.bn1 = nn.BatchNorm2d(num_features=64)
        self.bn2 = nn.BatchNorm2d(num_features=128)
        self.bn3 = nn.BatchNorm2d(num_features=256)
        self.bn4 = nn.BatchNorm1d(num_features=self.latent_size)
        self.dropout = nn.Dropout(p=args.dropout, inplace=False)
        self.act = args.act_fn

    # ------------------------------------------------------------------ eval: libgmk
    def _folded(self):
        """BatchNorm (eval: running statistics) folded into the producing layer:
        bn(conv(x)) = conv(x; w * s) + (b - mean) * s + beta,  s = gamma / sqrt(var + eps).
        Cached until a parameter or buffer is modified in place (tensor version counters) or replaced."""
        tensors = list(self.parameters()) + list(self.buffers())
        key = tuple((t.data_ptr(), t._version) for t in tensors)
        cache = self.__dict__.get("_gmk_fold")
        if cache is not None and cache[0] == key:
            return cache[1]
        out = {}
        with torch.no_grad():
            for layer, bn in ((self.conv1, self.bn1), (self.conv2, self.bn2), (self.conv3, self.bn3), (self.fc, self.bn4)):
                s = bn.weight / torch.sqrt(bn.running_var + bn.eps)
                w = layer.weight * s.view(-1, *([1] * (layer.weight.dim() - 1)))
                b = (layer.bias - bn.running_mean) * s + bn.bias
                out[layer] = (w.float().contiguous(), b.float().contiguous())
        self.__dict__["_gmk_fold"] = (key, out)
        return out

    def _forward_gmk(self, x):
        if not x.is_cuda:
            raise RuntimeError("Classifier eval forward runs gfx950 kernels (libgmk): move the model and the images to "
                               "an MI355X (there is no CPU path)")
        a = act_id(self.act)
        f = self._folded()
        x = x.detach().float().contiguous()
        for conv in (self.conv1, self.conv2, self.conv3):
            w, b = f[conv]
            x = ops.conv2d_fwd(x, w, b, stride=1, act=a)       # conv + folded BN + activation, classifiers.py:33,37,42
            x = ops.maxpool2_fwd(x, want_idx=False)            # :35,39,44
        x = x.view(-1, 4 * 4 * 256)                            # :46
        w, b = f[self.fc]
        x = ops.linear_fwd(x, w, b, act=a)                     # fc + folded BN1d + activation, :48 (dropout: identity in eval)
        self.latent_vector = x                                 # :51
        return ops.linear_fwd(x, self.fc_out.weight.detach(), self.fc_out.bias.detach())  # :52

    # ------------------------------------------------------------------ train: PyTorch autograd (out of scope for libgmk)
    def _forward_torch(self, x):
        x = self.max_pool(self.act(self.bn1(self.conv1(x))))
        x = self.max_pool(self.act(self.bn2(self.conv2(x))))
        x = self.max_pool(self.act(self.bn3(self.conv3(x))))
        x = x.view(-1, 4 * 4 * 256)
        x = self.act(self.bn4(self.fc(x)))
        x = self.dropout(x)
        self.latent_vector = x
        return self.fc_out(x)

    def forward(self, x):
        print_debug(x, self.debug, name='\nClassifier: input')
        x = self._forward_torch(x) if self.training else self._forward_gmk(x)
        print_debug(x, self.debug, name='Classifier: after fc_out')
        return x

    def __getstate__(self):
        state = self.__dict__.copy()
        state.pop("_gmk_fold", None)  # the folded-weight cache is not part of a saved model
        return state


def compute_accuracy(outputs, labels):
    """Top-1 accuracy of a batch in percent (same value as reference classifiers.py:69-74)."""
    with torch.no_grad():
        hits = (outputs.argmax(dim=1) == labels).sum().item()
    return 100.0 * hits / labels.size(0)


def _epoch_accuracy(classifier, dataloader, device, step=None):
    """Mean of the per-batch accuracies over one pass; `step(outputs, labels)` runs the optimisation when given."""
    accs = []
    for images, labels in dataloader:
        images, labels = images.to(device), labels.to(device)
        outputs = classifier(images)
        if step is not None:
            step(outputs, labels)
        accs.append(compute_accuracy(outputs, labels))
    return np.mean(accs)


def train_classifier(args, classifier=None, train_dataloader=None, test_dataloader=None, device=None):
    """Auxiliary model training (reference classifiers.py:77-113): cross entropy, AdamW, cosine LR stepped every
    iteration over epochs * len(loader) steps; the full model object is saved, as the metrics' loader expects."""
    print('\n\nTraining CIFAR-10 Classifier\n\n')
    classifier = Classifier(args=args).to(device) if classifier is None else classifier
    optim = torch.optim.AdamW(classifier.parameters(), lr=args.lr, weight_decay=args.wd)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(optimizer=optim, T_max=args.epochs * len(train_dataloader))
    criterion = nn.CrossEntropyLoss()

    def step(outputs, labels):
        optim.zero_grad()
        criterion(outputs, labels).backward()
        optim.step()
        sched.step()

    for epoch in range(args.epochs):
        classifier.train()
        train_acc = _epoch_accuracy(classifier, train_dataloader, device, step=step)
        test_acc = evaluate_classifier(classifier, test_dataloader=test_dataloader, device=device)
        print(f'Epoch {epoch}  train {train_acc:.2f}  test {test_acc:.2f}  LR {sched.get_last_lr()[0]:.4f}')
    checkpoint_path = f'checkpoints/cifar_classifier_{args.latent_size}.pth'
    print(f'\n\nSaving classifier model checkpoint to {checkpoint_path}')
    torch.save(classifier, checkpoint_path)
    return classifier


def evaluate_classifier(classifier, test_dataloader=None, device=None):
    """Eval-mode pass (libgmk forward), reference classifiers.py:115-124."""
    classifier.eval()
    with torch.no_grad():
        return _epoch_accuracy(classifier, test_dataloader, device)
