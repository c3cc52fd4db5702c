"""tools/fwd_split.py again, with the host taken out: both variants recorded as launch plans (gmk/plan.py) and replayed
from C.  One full-batch forward chain (B = 256, one stream) against two half-batch chains (2 x 128, two streams)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "generative-models_amd"))
import torch
import bench
from gmk import ops, engine
from gmk._lib import lib, check
from gmk.plan import StepPlan
dev = torch.device("cuda:0")
model, optim, sched = bench.build(dev)
enc, dec = model.encoder, model.decoder
pe = {n: p.data for n, p in enc.named_parameters()}
pd = {n: p.data for n, p in dec.named_parameters()}
x = torch.rand(256, 3, 32, 32, device=dev)
eps = torch.randn(256, 256, device=dev)
main = torch.cuda.current_stream()
s2 = torch.cuda.Stream()
NS = int(os.environ.get("SPLIT", "2"))
streams = [main] + [torch.cuda.Stream() for _ in range(NS - 1)]

def fwd(xb, eb):
    stats, sve = engine.encoder_forward(pe, xb, act=enc._act_id, no_pool=enc.no_pool, ksize=enc.kernel_size, train=True, reparam_eps=eb)
    pre, svd = engine.decoder_forward(pd, sve["z"], act=dec._act_id, num_channels=dec.num_channels, no_upsample=dec.no_upsample,
                                      interpolation=dec._interp, ksize=dec.kernel_size, base=dec.base, train=True, final_act=None)
    return pre, sve, svd

def full():
    with ops.stream_scope(main):
        return fwd(x, eps)

hb = 256 // NS
xs = [x[i * hb:(i + 1) * hb].contiguous() for i in range(NS)]
es = [eps[i * hb:(i + 1) * hb].contiguous() for i in range(NS)]
def split():
    out = []
    for i, st in enumerate(streams):
        if i:
            check(lib().gmk_stream_wait_stream(st.cuda_stream, main.cuda_stream))
        with ops.stream_scope(st):
            out.append(fwd(xs[i], es[i]))
    for st in streams[1:]:
        check(lib().gmk_stream_wait_stream(main.cuda_stream, st.cuda_stream))
    return out

def record(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    p = StepPlan()
    with p: p.out = fn()
    p.set_inputs()
    return p

pf, ps = record(full), record(split)
for name, p in (("full", pf), (f"split{NS}", ps), ("full", pf), (f"split{NS}", ps)):
    for _ in range(10): p.enqueue(0, p.nseg)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200): p.enqueue(0, p.nseg)
    torch.cuda.synchronize()
    print(name, f"{(time.perf_counter() - t0) * 5e3:.1f} us per forward", flush=True)
