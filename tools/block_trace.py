"""Per-block timeline of one implicit-GEMM launch (tools-only build of libgmk with -DGMK_CONV_TRACE, see conv_igemm.h):
   GMK_LIB_PATH=generative-models_amd/gmk/libgmk_trace.so python3 tools/block_trace.py [deconv|s2|s1_8]
Every block records {HW_ID, XCC_ID, t_start, t_k_loop_done, t_stores_acknowledged} (100 MHz wall clock).  Prints, per phase,
the distribution over blocks and the timeline of the blocks that shared a few CUs: are the two residents of a CU in lockstep?"""
import ctypes, os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "generative-models_amd"))
import torch
import numpy as np
from gmk import ops, lib
dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "deconv"
B, C = 256, 64
w = (torch.rand(C, C, 3, 3, device=dev) - 0.5) * 0.1
b = torch.zeros(C, device=dev)
if which == "deconv":
    x = torch.rand(B, C, 16, 16, device=dev); fn = lambda: ops.deconv2d_fwd(x, w, b, act="gelu", want_dact=True); nblk = 1024
elif which == "s2":
    x = torch.rand(B, C, 32, 32, device=dev); fn = lambda: ops.conv2d_fwd(x, w, b, stride=2, act="gelu", want_dact=True); nblk = 512
else:
    x = torch.rand(B, C, 8, 8, device=dev); fn = lambda: ops.conv2d_fwd(x, w, b, act="gelu", want_dact=True); nblk = 256
for _ in range(5):
    fn()
torch.cuda.synchronize()
fn()
torch.cuda.synchronize()
buf = (ctypes.c_uint64 * (4 * nblk))()
h = lib()
h.gmk_debug_conv_trace_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert h.gmk_debug_conv_trace_read(buf, nblk) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(nblk, 4)
hw = a[:, 0] & 0xFFFFFFFF; xcc = (a[:, 0] >> 32) & 0xF
t = (a[:, 1:4].astype(np.int64) - int(a[:, 1].min())) * 0.01   # us (100 MHz)
cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 7; simd = (hw >> 4) & 3; slot = hw & 0xF
key = [(int(xcc[i]), int(se[i]), int(sh[i]), int(cu[i])) for i in range(nblk)]
print(f"{which}: {nblk} blocks on {len(set(key))} distinct (xcc, se, sh, cu); kernel span {t[:, 2].max():.1f} us")
print("phase durations over blocks (us): prologue+K  mean %.1f  min %.1f  max %.1f | epilogue mean %.1f  min %.1f  max %.1f"
      % ((t[:, 1] - t[:, 0]).mean(), (t[:, 1] - t[:, 0]).min(), (t[:, 1] - t[:, 0]).max(),
         (t[:, 2] - t[:, 1]).mean(), (t[:, 2] - t[:, 1]).min(), (t[:, 2] - t[:, 1]).max()))
starts = np.sort(t[:, 0])
print("block start times (us): first 8", np.round(starts[:8], 1), " #started before 2 us:", int((starts < 2).sum()),
      " quartiles", np.round(np.percentile(starts, [25, 50, 75, 100]), 1))
# chip-wide: how many blocks are in their epilogue at each instant
grid = np.arange(0, t[:, 2].max(), 1.0)
in_epi = [(int(((t[:, 1] <= g) & (t[:, 2] > g)).sum()), int(((t[:, 0] <= g) & (t[:, 1] > g)).sum())) for g in grid]
print("t(us): blocks in epilogue / in prologue+K:", " ".join(f"{int(g)}:{e}/{k}" for g, (e, k) in zip(grid, in_epi) if int(g) % 3 == 0))
eb = (ctypes.c_uint64 * (8 * nblk))()
h.gmk_debug_conv_trace_read_epi.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert h.gmk_debug_conv_trace_read_epi(eb, nblk) == 0
ep = (np.frombuffer(eb, dtype=np.uint64).reshape(nblk, 8)[:, :4].astype(np.int64) - int(a[:, 1].min())) * 0.01
by = collections.defaultdict(list)
for i, k in enumerate(key):
    by[k].append(i)
print("blocks per CU: min %d max %d" % (min(len(v) for v in by.values()), max(len(v) for v in by.values())))
for k in list(by)[:4]:
    print("CU", k)
    for i in sorted(by[k], key=lambda i: t[i, 0]):
        print(f"   block {i:5d} wave slot {int(slot[i])} simd {int(simd[i])}: start {t[i,0]:6.1f}  K done {t[i,1]:6.1f}  | pass0 transposed {ep[i,0]:6.1f} stores issued {ep[i,1]:6.1f} | pass1 transposed {ep[i,2]:6.1f} stores issued {ep[i,3]:6.1f} | acknowledged {t[i,2]:6.1f}")
