"""Would two half-batch forward chains on two streams beat one full-batch chain?  Times encoder+decoder forward (train mode)
for B=256 on one stream against 2 x B=128 on two streams (fork/join through gmk_stream_wait_stream)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "generative-models_amd"))
import torch
import bench
from gmk import ops, engine
from gmk._lib import lib, check
dev = torch.device("cuda:0")
model, optim, sched = bench.build(dev)
enc, dec = model.encoder, model.decoder
pe = {n: p.data for n, p in enc.named_parameters()}
pd = {n: p.data for n, p in dec.named_parameters()}
x = torch.rand(256, 3, 32, 32, device=dev)
eps = torch.randn(256, 256, device=dev)
main = torch.cuda.current_stream()
s2 = torch.cuda.Stream()

def fwd(xb, eb):
    stats, sve = engine.encoder_forward(pe, xb, act=enc._act_id, no_pool=enc.no_pool, ksize=enc.kernel_size, train=True, reparam_eps=eb)
    pre, svd = engine.decoder_forward(pd, sve["z"], act=dec._act_id, num_channels=dec.num_channels, no_upsample=dec.no_upsample,
                                      interpolation=dec._interp, ksize=dec.kernel_size, base=dec.base, train=True, final_act=None)
    return pre, sve, svd

def full():
    with ops.stream_scope(main):
        return fwd(x, eps)

xa, xb_, ea, eb_ = x[:128].contiguous(), x[128:].contiguous(), eps[:128].contiguous(), eps[128:].contiguous()
def split():
    check(lib().gmk_stream_wait_stream(s2.cuda_stream, main.cuda_stream))
    with ops.stream_scope(main):
        a = fwd(xa, ea)
    with torch.cuda.stream(s2):
        with ops.stream_scope(s2):
            b = fwd(xb_, eb_)
    check(lib().gmk_stream_wait_stream(main.cuda_stream, s2.cuda_stream))
    return a, b

for name, fn in (("full", full), ("split", split), ("full", full), ("split", split)):
    for _ in range(10): keep = fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100): keep = fn()
    torch.cuda.synchronize()
    print(name, f"{(time.perf_counter() - t0) * 1e4:.1f} us per forward", flush=True)
