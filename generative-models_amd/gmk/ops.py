"""Tensor-level wrappers over the C ABI (no autograd here: the autograd.Functions live in models/autoencoders.py).

Every function takes contiguous fp32 CUDA tensors, allocates its outputs with torch (device
memory plumbing), and launches on torch's current stream.
"""
import os
import torch

from ._lib import lib, check, act_id, UPSAMPLE_IDS

_ws_cache = {}
_ws_retired = []


# Stream handle override: torch.cuda.current_stream() costs ~7 us of host time per call and every op asks for it; at ~80
# launches per train step that alone was half of the 1.0 ms of host time that bounds the B = 256 step.  train_step sets
# the handle once (stream_scope) and the side queue swaps it around the calls it redirects.
_cur_stream = None


def _stream():
    return _cur_stream if _cur_stream is not None else torch.cuda.current_stream().cuda_stream


class stream_scope:
    """with stream_scope(torch_stream): every gmk op inside launches on `torch_stream` (which must be torch's current
    stream, so that torch's allocator and ops agree) without asking torch for it again."""

    def __init__(self, stream):
        self.handle = stream.cuda_stream

    def __enter__(self):
        global _cur_stream
        self.prev = _cur_stream
        _cur_stream = self.handle

    def __exit__(self, *exc):
        global _cur_stream
        _cur_stream = self.prev


def _p(t):
    return None if t is None else t.data_ptr()


# While a launch plan is being recorded (gmk/plan.py) every tensor these wrappers allocate is appended to the plan's keep
# list: the recorded pointers must stay valid, and owned by nobody else, for as long as the plan is replayed.
_rec_keep = None


def _new(shape, device, dtype=torch.float32):
    t = torch.empty(shape, device=device, dtype=dtype)
    if _rec_keep is not None:
        _rec_keep.append(t)
    return t


def _new_like(x):
    t = torch.empty_like(x)
    if _rec_keep is not None:
        _rec_keep.append(t)
    return t


def _chk(*ts):
    for t in ts:
        if t is None:
            continue
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise ValueError("gmk ops need contiguous fp32 CUDA tensors")


def workspace(nbytes, device):
    """A cached byte workspace per (device, current stream), grown on demand: kernels of one stream run in
    order, so they can share scratch; two streams (the wgrad side queue) must not."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), _stream())
    ws = _ws_cache.get(key)
    if ws is None or ws.numel() < nbytes:
        # A workspace that has to grow is RETIRED, not freed: under a side-stream handle (SideQueue.run swaps the handle,
        # not torch's current stream) the tensor belongs to the main stream's allocator pool, so dropping it would hand its
        # memory to the next main-stream torch.empty while an earlier weight-gradient kernel on the side stream may still
        # be using it as slab scratch; recorded launch plans also name it.  Growth is geometric, so the retired buffers
        # of a key sum to less than the live one.
        if ws is not None:
            _ws_retired.append(ws)
        grow = 0 if ws is None else 2 * ws.numel()
        ws = torch.empty(max(int(nbytes), grow, 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = ws
    return ws


class SideQueue:
    """Runs independent work (the weight-gradient kernels of backward) on a second HIP stream so that it
    overlaps the dgrad chain on the main stream: the two kinds of kernel fill each other's prologue /
    epilogue / tail bubbles.  Tensors read on the side stream are kept alive until join() has ordered the
    main stream behind it (the caching allocator would otherwise hand their memory to later main-stream work)."""

    def __init__(self, device):
        self.device = device
        # stream priority knob (GMK_SIDE_PRIORITY=-1 = high).  High-priority weight-gradient streams are 0.8 % faster in a
        # single process (1.025 -> 1.017 ms) but HANG when two processes share a card (the 2-rank rehearsal test, 3 of 3
        # runs; priority 0: never): a high-priority queue waiting on an event of a queue that is not scheduled.  A DP
        # job also runs more queues than the 4 hardware queues of a process (RCCL's), so the default stays 0.
        prio = int(os.environ.get("GMK_SIDE_PRIORITY", "0"))
        if prio != 0 and int(os.environ.get("WORLD_SIZE", "1")) > 1:
            raise RuntimeError("GMK_SIDE_PRIORITY != 0 is refused in a multi-process job: high-priority weight-gradient "
                               "streams hung 3 of 3 two-rank runs (a high-priority queue waiting on an event of a queue "
                               "that is not scheduled; a DP rank runs main + 2 side + RCCL streams on the 4 hardware "
                               "queues of a process).  Unset it, or run single-process.")
        # GMK_SIDE_STREAMS=n: the weight-gradient launches alternate over n side streams.  Default 2: 1.056 -> 1.021 ms per step
        # once the host stopped bounding the step (1 -> 2 -> 3 -> 4 streams: 1.056 / 1.021 / 1.021 / 1.040 ms; while the step was
        # host-bound, round 2 first measured two streams 1-3 % slower)
        n = max(1, int(os.environ.get("GMK_SIDE_STREAMS", "2")))
        # (CU-masked side streams -- hipExtStreamCreateWithCUMask, 192 of 256 CUs -- were measured in round 3: 1.90 ms per step
        # against 0.98, the masked queues serialise against the main one; not kept)
        self.streams = [torch.cuda.Stream(device, priority=prio) for _ in range(n)]
        self.stream = self.streams[0]
        self._next = 0
        self._keep = []
        self._main = None
        self._handles = [st.cuda_stream for st in self.streams]

    def begin(self, main_stream):
        """Names the stream that run() forks from until join() (saves a torch.cuda.current_stream() per call)."""
        self._main = main_stream

    def run(self, fn, *tensors):
        """fn() is enqueued on a side stream after everything enqueued so far on the main stream.  The gmk ops inside
        launch on the side stream through the handle override; torch's own current stream is left alone (fn must not
        allocate tensors that die before join(): the weight-gradient calls write into the flat gradient buffer)."""
        global _cur_stream
        main = self._main if self._main is not None else torch.cuda.current_stream(self.device)
        i = self._next
        self._next = (i + 1) % len(self.streams)
        check(lib().gmk_stream_wait_stream(self._handles[i], main.cuda_stream))
        self._keep.extend(tensors)
        prev = _cur_stream
        _cur_stream = self._handles[i]
        try:
            return fn()
        finally:
            _cur_stream = prev

    def reduce_batch(self, items):
        """The deferred slab reductions of a gradient bucket in one launch on streams[0], behind the weight-gradient kernels
        of every side stream (forks through gmk_stream_wait_stream so that a launch plan records them)."""
        global _cur_stream
        for h in self._handles[1:]:
            check(lib().gmk_stream_wait_stream(self._handles[0], h))
        prev = _cur_stream
        _cur_stream = self._handles[0]
        try:
            wgrad_reduce_batch(items)
        finally:
            _cur_stream = prev

    def gather(self):
        """Orders streams[0] behind the other side streams (so that work enqueued on it next, e.g. a gradient bucket's
        all-reduce, sees every weight gradient launched so far)."""
        for st in self.streams[1:]:
            self.stream.wait_stream(st)

    def join(self):
        """The current stream waits for all side work; kept tensors may be freed afterwards."""
        cur = self._main if self._main is not None else torch.cuda.current_stream(self.device)
        for h in self._handles:
            check(lib().gmk_stream_wait_stream(cur.cuda_stream, h))
        self._next = 0
        self._main = None
        self._keep.clear()


_side_cache = {}
_aux_cache = {}
_seg_cache = {}


def aux_stream(device):
    """A third stream per device (early AdamW updates of finished gradient buckets)."""
    key = device.index if device.index is not None else torch.cuda.current_device()
    st = _aux_cache.get(key)
    if st is None:
        st = _aux_cache[key] = torch.cuda.Stream(torch.device("cuda", key))
    return st


def bucket_segments(model, optimizer):
    """Flat (start, end) ranges of the three gradient buckets of an Autoencoder in `optimizer`'s buffers (the same split
    as gmk.parallel.GradReducer), cached per (model, optimizer)."""
    key = (id(model), id(optimizer), optimizer.flat_param.data_ptr())
    seg = _seg_cache.get(key)
    if seg is None:
        enc_fc = [model.encoder.fc.weight, model.encoder.fc.bias]
        enc_conv = [p for n, p in model.encoder.named_parameters() if not n.startswith("fc.")]
        seg = _seg_cache[key] = {"decoder": optimizer.segment(list(model.decoder.parameters())),
                                 "encoder.fc": optimizer.segment(enc_fc), "encoder.conv": optimizer.segment(enc_conv)}
    return seg


def side_queue(device):
    """The process-wide wgrad side queue of `device` (GMK_SIDE_WGRAD=0 disables it: one stream for everything)."""
    import os
    if os.environ.get("GMK_SIDE_WGRAD", "1") == "0":
        return None
    key = device.index if device.index is not None else torch.cuda.current_device()
    q = _side_cache.get(key)
    if q is None:
        q = _side_cache[key] = SideQueue(torch.device("cuda", key))
    return q


def conv2d_fwd(x, w, bias, stride=1, act=None, want_dact=False):
    _chk(x, w, bias)
    N, Cin, H, W = x.shape
    Cout, _, k, _ = w.shape
    y = _new((N, Cout, H // stride, W // stride), x.device, torch.float32)
    dact = _new_like(y) if want_dact else None
    check(lib().gmk_conv2d_fwd(_p(x), _p(w), _p(bias), _p(y), _p(dact), N, Cin, H, W, Cout, k, stride, act_id(act), _stream()))
    return (y, dact) if want_dact else y


def conv2d_dgrad(dy, w, x_shape, stride=1, dact_in=None):
    _chk(dy, w, dact_in)
    N, Cin, H, W = x_shape
    Cout, _, k, _ = w.shape
    dx = _new(x_shape, dy.device, torch.float32)
    check(lib().gmk_conv2d_dgrad(_p(dy), _p(w), _p(dact_in), _p(dx), N, Cin, H, W, Cout, k, stride, _stream()))
    return dx


_layer_ws = {}


def _layer_workspace(key, nbytes, device):
    """A workspace of a layer's own (deferred slab reduction: the slabs must survive until the bucket's batch reduce), keyed
    by the layer's gradient destination; grown buffers are retired like the shared ones (workspace())."""
    ws = _layer_ws.get(key)
    if ws is None or ws.numel() < nbytes:
        if ws is not None:
            _ws_retired.append(ws)
        ws = _layer_ws[key] = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
    return ws


def wgrad_reduce_batch(items):
    """One launch summing the slabs of the deferred weight gradients in `items` (conv2d_wgrad / deconv2d_wgrad, defer=)."""
    from ._lib import WgradLayer
    for a in range(0, len(items), 8):
        chunk = items[a:a + 8]
        arr = (WgradLayer * len(chunk))()
        for k, (ws, dw, db, geo) in enumerate(chunk):
            arr[k] = WgradLayer(ws.data_ptr(), dw.data_ptr(), _p(db), *geo)
        check(lib().gmk_conv_wgrad_reduce_batch(arr, len(chunk), _stream()))


def conv2d_wgrad(dy, x, ksize, stride=1, want_bias=True, out=None, defer=None):
    """out: optional (dw, db) tensors to write into (e.g. views of a flat gradient buffer).
    defer: a list -> where the library can (gmk_conv_wgrad_can_defer) only the partial slabs are written now, into a workspace
    of the layer's own, and (ws, dw, db, geometry) is appended for a later wgrad_reduce_batch(); needs `out`."""
    _chk(dy, x)
    N, Cin, H, W = x.shape
    Cout = dy.shape[1]
    if defer is not None and out is not None and lib().gmk_conv_wgrad_can_defer(N, Cin, H, W, Cout, ksize, stride, 0):
        dw, db = out
        _chk(dw, db)
        nb = lib().gmk_conv2d_wgrad_workspace_bytes(N, Cin, H, W, Cout, ksize, stride)
        ws = _layer_workspace(dw.data_ptr(), nb, x.device)
        check(lib().gmk_conv2d_wgrad(_p(dy), _p(x), None, _p(db), N, Cin, H, W, Cout, ksize, stride, _p(ws), ws.numel(), _stream()))
        defer.append((ws, dw, db, (N, Cin, H, W, Cout, ksize, stride, 0)))
        return dw, db
    if out is not None:
        dw, db = out
        _chk(dw, db)
    else:
        dw = _new((Cout, Cin, ksize, ksize), x.device, torch.float32)
        db = _new((Cout,), x.device, torch.float32) if want_bias else None
    nb = lib().gmk_conv2d_wgrad_workspace_bytes(N, Cin, H, W, Cout, ksize, stride)
    ws = workspace(nb, x.device)
    check(lib().gmk_conv2d_wgrad(_p(dy), _p(x), _p(dw), _p(db), N, Cin, H, W, Cout, ksize, stride, _p(ws), ws.numel(), _stream()))
    return dw, db


def deconv2d_fwd(x, w, bias, act=None, want_dact=False):
    _chk(x, w, bias)
    N, Cin, H, W = x.shape
    _, Cout, k, _ = w.shape
    y = _new((N, Cout, 2 * H, 2 * W), x.device, torch.float32)
    dact = _new_like(y) if want_dact else None
    check(lib().gmk_deconv2d_fwd(_p(x), _p(w), _p(bias), _p(y), _p(dact), N, Cin, H, W, Cout, k, act_id(act), _stream()))
    return (y, dact) if want_dact else y


def deconv2d_dgrad(dy, w, x_shape, dact_in=None):
    _chk(dy, w, dact_in)
    N, Cin, H, W = x_shape
    _, Cout, k, _ = w.shape
    dx = _new(x_shape, dy.device, torch.float32)
    check(lib().gmk_deconv2d_dgrad(_p(dy), _p(w), _p(dact_in), _p(dx), N, Cin, H, W, Cout, k, _stream()))
    return dx


def deconv2d_wgrad(dy, x, ksize, want_bias=True, out=None, defer=None):
    _chk(dy, x)
    N, Cin, H, W = x.shape
    Cout = dy.shape[1]
    if defer is not None and out is not None and lib().gmk_conv_wgrad_can_defer(N, Cin, H, W, Cout, ksize, 2, 1):
        dw, db = out
        _chk(dw, db)
        nb = lib().gmk_deconv2d_wgrad_workspace_bytes(N, Cin, H, W, Cout, ksize)
        ws = _layer_workspace(dw.data_ptr(), nb, x.device)
        check(lib().gmk_deconv2d_wgrad(_p(dy), _p(x), None, _p(db), N, Cin, H, W, Cout, ksize, _p(ws), ws.numel(), _stream()))
        defer.append((ws, dw, db, (N, Cin, H, W, Cout, ksize, 2, 1)))
        return dw, db
    if out is not None:
        dw, db = out
        _chk(dw, db)
    else:
        dw = _new((Cin, Cout, ksize, ksize), x.device, torch.float32)
        db = _new((Cout,), x.device, torch.float32) if want_bias else None
    nb = lib().gmk_deconv2d_wgrad_workspace_bytes(N, Cin, H, W, Cout, ksize)
    ws = workspace(nb, x.device)
    check(lib().gmk_deconv2d_wgrad(_p(dy), _p(x), _p(dw), _p(db), N, Cin, H, W, Cout, ksize, _p(ws), ws.numel(), _stream()))
    return dw, db


def linear_fwd(x, w, bias, act=None, want_dact=False):
    _chk(x, w, bias)
    M, K = x.shape
    N = w.shape[0]
    y = _new((M, N), x.device, torch.float32)
    dact = _new_like(y) if want_dact else None
    ws = workspace(lib().gmk_linear_workspace_bytes(M, K, N), x.device)
    check(lib().gmk_linear_fwd(_p(x), _p(w), _p(bias), _p(y), _p(dact), M, K, N, act_id(act), _p(ws), ws.numel(), _stream()))
    return (y, dact) if want_dact else y


def linear_dgrad(dy, w, dact_in=None):
    _chk(dy, w, dact_in)
    M, N = dy.shape
    K = w.shape[1]
    dx = _new((M, K), dy.device, torch.float32)
    ws = workspace(lib().gmk_linear_workspace_bytes(M, K, N), dy.device)
    check(lib().gmk_linear_dgrad(_p(dy), _p(w), _p(dact_in), _p(dx), M, K, N, _p(ws), ws.numel(), _stream()))
    return dx


def linear_wgrad(dy, x, want_bias=True, out=None):
    _chk(dy, x)
    M, N = dy.shape
    K = x.shape[1]
    if out is not None:
        dw, db = out
        _chk(dw, db)
    else:
        dw = _new((N, K), x.device, torch.float32)
        db = _new((N,), x.device, torch.float32) if want_bias else None
    ws = workspace(lib().gmk_linear_workspace_bytes(M, K, N), x.device)
    check(lib().gmk_linear_wgrad(_p(dy), _p(x), _p(dw), _p(db), M, K, N, _p(ws), ws.numel(), _stream()))
    return dw, db


def linear_fwd_reparam(x, w, bias, eps):
    """Encoder head + reparameterisation in one GEMM (+ its reduction): returns (stats (M, L, 2), z (M, L))."""
    _chk(x, w, bias, eps)
    M, K = x.shape
    N = w.shape[0]
    stats = _new((M, N // 2, 2), x.device, torch.float32)
    z = _new((M, N // 2), x.device, torch.float32)
    ws = workspace(lib().gmk_linear_workspace_bytes(M, K, N), x.device)
    check(lib().gmk_linear_fwd_reparam(_p(x), _p(w), _p(bias), _p(eps), _p(stats), _p(z), M, K, N, _p(ws), ws.numel(), _stream()))
    return stats, z


def linear_dgrad_reparam(dy, w, stats, eps, kl_scale, mse_prior=False):
    """Decoder fc dgrad + reparameterisation backward + beta*KL gradient: returns dstats (M, L, 2)."""
    _chk(dy, w, stats, eps)
    M, N = dy.shape
    K = w.shape[1]
    dstats = _new_like(stats)
    ws = workspace(lib().gmk_linear_workspace_bytes(M, K, N), dy.device)
    check(lib().gmk_linear_dgrad_reparam(_p(dy), _p(w), _p(stats), _p(eps), _p(dstats), M, K, N, float(kl_scale),
                                         1 if mse_prior else 0, _p(ws), ws.numel(), _stream()))
    return dstats


def reparam_fwd(stats, eps):
    _chk(stats, eps)
    B, L, _ = stats.shape
    z = _new((B, L), stats.device, torch.float32)
    check(lib().gmk_reparam_fwd(_p(stats), _p(eps), _p(z), B, L, _stream()))
    return z


def reparam_bwd(dz, stats, eps, dstats=None):
    _chk(dz, stats, eps, dstats)
    B, L, _ = stats.shape
    acc = 1
    if dstats is None:
        dstats = _new_like(stats)
        acc = 0
    check(lib().gmk_reparam_bwd(_p(dz), _p(stats), _p(eps), _p(dstats), B, L, acc, _stream()))
    return dstats


LOSS_SIGMOID, LOSS_MSE_PRIOR, LOSS_NO_DSTATS = 1, 2, 4


_loss_sync = {}


def _loss_state(device):
    """Private (partials workspace, zeroed 256-byte sync word block) of the one-launch loss kernel per (device, stream):
    unlike the shared byte workspace these must not be touched by any other op (the kernel keeps the sync block at zero),
    and two streams must not share them (their launches could overlap and split one ticket count)."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), _stream())
    st = _loss_sync.get(key)
    if st is None and torch.cuda.is_current_stream_capturing():
        # nothing may be created (zero fill + synchronize) under HIP graph capture: take the state a warm-up step made
        # on this device (the captured step is the only user of it while the graph replays: its streams are graph edges)
        st = next((v for k, v in _loss_sync.items() if k[0] == key[0]), None)
        if st is None:
            raise RuntimeError("run one eager train step on this device before capturing it into a graph")
        return st
    if st is None:
        ws = torch.empty(int(lib().gmk_vae_loss_workspace_bytes()), dtype=torch.uint8, device=device)
        sync = torch.zeros(256, dtype=torch.uint8, device=device)
        torch.cuda.current_stream(device).synchronize()
        st = _loss_sync[key] = (ws, sync)
    return st


def vae_loss_fwd_bwd(pre, image, stats, beta, sigmoid=True, mse_prior=False, want_recon=False, want_dstats=True, fused=False):
    """Returns (losses[3] = rec, kl, total ; dpre ; dstats or None ; recon or None).  want_dstats=False: the KL gradient
    is left to linear_dgrad_reparam (fused train step)."""
    _chk(pre, image, stats)
    B = pre.shape[0]
    chw = pre.numel() // B
    L = 0 if stats is None else stats.shape[1]
    dpre = _new_like(pre)
    dstats = None if (stats is None or not want_dstats) else _new_like(stats)
    recon = _new_like(pre) if want_recon else None
    losses = _new(3, pre.device, torch.float32)
    ws = workspace(lib().gmk_vae_loss_workspace_bytes(), pre.device)
    flags = (LOSS_SIGMOID if sigmoid else 0) | (LOSS_MSE_PRIOR if mse_prior else 0) | (0 if want_dstats else LOSS_NO_DSTATS)
    if fused:   # one launch (the last block finalizes); private workspace + sync block, main-stream callers only
        lws, sync = _loss_state(pre.device)
        check(lib().gmk_vae_loss_fwd_bwd_fused(_p(pre), _p(image), _p(stats), _p(recon), _p(dpre), _p(dstats), _p(losses),
                                               B, chw, L, float(beta), flags, _p(lws), lws.numel(), _p(sync), _stream()))
        return losses, dpre, dstats, recon
    check(lib().gmk_vae_loss_fwd_bwd(_p(pre), _p(image), _p(stats), _p(recon), _p(dpre), _p(dstats), _p(losses),
                                     B, chw, L, float(beta), flags, _p(ws), ws.numel(), _stream()))
    return losses, dpre, dstats, recon


def vae_loss_eval(pre, image, stats, beta, sigmoid=True, mse_prior=False, want_recon=False):
    """No-grad form (compute_test_loss, main.py:73-111): returns (out5 = rec, kl, total, mean|mu|, mean sigma ; recon or None)."""
    _chk(pre, image, stats)
    B = pre.shape[0]
    chw = pre.numel() // B
    L = 0 if stats is None else stats.shape[1]
    recon = _new_like(pre) if want_recon else None
    out5 = _new(5, pre.device, torch.float32)
    ws = workspace(lib().gmk_vae_loss_workspace_bytes(), pre.device)
    flags = (LOSS_SIGMOID if sigmoid else 0) | (LOSS_MSE_PRIOR if mse_prior else 0)
    check(lib().gmk_vae_loss_eval(_p(pre), _p(image), _p(stats), _p(recon), _p(out5), B, chw, L, float(beta), flags,
                                  _p(ws), ws.numel(), _stream()))
    return out5, recon


def maxpool2_fwd(x, want_idx=True):
    _chk(x)
    N, C, H, W = x.shape
    y = _new((N, C, H // 2, W // 2), x.device, torch.float32)
    idx = _new((N, C, H // 2, W // 2), x.device, torch.uint8) if want_idx else None
    check(lib().gmk_maxpool2_fwd(_p(x), _p(y), _p(idx), N * C, H, W, _stream()))
    return (y, idx) if want_idx else y


def maxpool2_bwd(dy, idx, dact_in=None):
    _chk(dy, dact_in)
    N, C, OH, OW = dy.shape
    dx = _new((N, C, 2 * OH, 2 * OW), dy.device, torch.float32)
    check(lib().gmk_maxpool2_bwd(_p(dy), _p(idx), _p(dact_in), _p(dx), N * C, 2 * OH, 2 * OW, _stream()))
    return dx


def upsample2_fwd(x, mode="nearest"):
    _chk(x)
    if mode not in UPSAMPLE_IDS:
        raise NotImplementedError(f"interpolation mode {mode!r} is not valid for 4-D input (nearest|bilinear|bicubic)")
    N, C, H, W = x.shape
    y = _new((N, C, 2 * H, 2 * W), x.device, torch.float32)
    check(lib().gmk_upsample2_fwd(_p(x), _p(y), N * C, H, W, UPSAMPLE_IDS[mode], _stream()))
    return y


def upsample2_bwd(dy, mode="nearest", dact_in=None):
    _chk(dy, dact_in)
    N, C, OH, OW = dy.shape
    dx = _new((N, C, OH // 2, OW // 2), dy.device, torch.float32)
    check(lib().gmk_upsample2_bwd(_p(dy), _p(dact_in), _p(dx), N * C, OH // 2, OW // 2, UPSAMPLE_IDS[mode], _stream()))
    return dx


def ew_mul(a, b):
    _chk(a, b)
    out = _new_like(a)
    check(lib().gmk_ew_mul(_p(a), _p(b), _p(out), a.numel(), _stream()))
    return out


def gram_accumulate(x, G, s):
    """G (D,D) f64 += x^T x ; s (D) f64 += x.sum(0) for fp32 features x (n, D)."""
    _chk(x)
    n, D = x.shape
    ws = workspace(lib().gmk_gram_workspace_bytes(n, D), x.device)
    check(lib().gmk_gram_accumulate(_p(x), n, D, G.data_ptr(), s.data_ptr(), _p(ws), ws.numel(), _stream()))


def fid_from_moments(m1, m2, iters=50, delta=1e-12, diag_eps=0.0):
    """m = (n, s f64 (D), G f64 (D,D)).  Returns an 8-double device tensor
    {dmu2, tr1, tr2, fro, c, tr_sqrt, fid, -}."""
    n1, s1, G1 = m1
    n2, s2, G2 = m2
    D = s1.numel()
    out = torch.zeros(8, dtype=torch.float64, device=s1.device)
    ws = workspace(lib().gmk_fid_workspace_bytes(D), s1.device)
    check(lib().gmk_fid_from_moments(G1.data_ptr(), s1.data_ptr(), int(n1), G2.data_ptr(), s2.data_ptr(), int(n2), D,
                                     int(iters), float(delta), float(diag_eps), out.data_ptr(), _p(ws), ws.numel(), _stream()))
    return out


def softmax_stats(logits, out=None, topn=3, eps=1e-7):
    """Accumulates the (2C+4)-double statistics vector of gmk_softmax_stats for logits (n, C)."""
    _chk(logits)
    n, C = logits.shape
    if out is None:
        out = torch.zeros(2 * C + 4, dtype=torch.float64, device=logits.device)
    check(lib().gmk_softmax_stats(_p(logits), n, C, int(topn), float(eps), out.data_ptr(), _stream()))
    return out
