"""gmk -- Python host side of libgmk.so (hand-written gfx950 kernels behind a C ABI).

`gmk.lib()` loads the shared library (and fails loudly if it is missing: there is no CPU or
PyTorch fallback for the product path); `gmk.ops` wraps each C entry point for torch tensors.
"""
from ._lib import lib, check, LIB_PATH, ACT_IDS, UPSAMPLE_IDS, act_id  # noqa: F401
