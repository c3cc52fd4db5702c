"""CPU restatement of the reference Classifier's eval-mode forward (TEST INFRASTRUCTURE; torch-CPU functional ops).

Follows /root/reference/models/classifiers.py:31-65 with BatchNorm in eval mode (running statistics) and dropout
off.  Pinned against the imported reference by oracle/gen_golden.py (tests/golden/classifier.json).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import oracle/.
"""
import torch
import torch.nn.functional as F

from . import ae_oracle

BN_EPS = 1e-5  # nn.BatchNorm2d / nn.BatchNorm1d default


def _bn(x, p, name):
    return F.batch_norm(x, p[name + ".running_mean"], p[name + ".running_var"], p[name + ".weight"], p[name + ".bias"],
                        training=False, eps=BN_EPS)


def classifier_forward(p, x, act):
    """p: name -> tensor (state_dict of the reference Classifier); x (B,3,32,32).  Returns (logits, latent_vector)."""
    a = lambda t: ae_oracle.act_forward(t, act)
    x = a(_bn(F.conv2d(x, p["conv1.weight"], p["conv1.bias"], 1, 2), p, "bn1"))   # classifiers.py:33
    x = F.max_pool2d(x, 2, 2)                                                     # :35
    x = a(_bn(F.conv2d(x, p["conv2.weight"], p["conv2.bias"], 1, 2), p, "bn2"))   # :37
    x = F.max_pool2d(x, 2, 2)                                                     # :39
    x = a(_bn(F.conv2d(x, p["conv3.weight"], p["conv3.bias"], 1, 1), p, "bn3"))   # :42
    x = F.max_pool2d(x, 2, 2)                                                     # :44
    x = x.reshape(-1, 4 * 4 * 256)                                                # :46
    x = a(_bn(F.linear(x, p["fc.weight"], p["fc.bias"]), p, "bn4"))               # :48
    latent = x                                                                    # :50-51 (dropout is identity in eval)
    logits = F.linear(x, p["fc_out.weight"], p["fc_out.bias"])                    # :52
    return logits, latent
