"""Data-parallel gradient reduction over RCCL (torch.distributed backend "nccl" on ROCm).

The train step shards by batch (pure DP): losses are batch means, so equal local
batches + gradient averaging reproduce the single-process global-batch gradient.  Gradients live
in FusedAdamW's flat buffer; it is all-reduced (SUM) in contiguous buckets that are launched
as soon as the backward pass has produced them -- decoder + encoder.fc (95 % of the parameters, final ~35 us apart) as one
collective, then the encoder convs -- on the process group's own stream, overlapping the rest of backward.  The 1/world factor is
folded into the AdamW kernel (optimizer.grad_scale), so there is no extra pass over the gradients.
"""
import torch
import torch.distributed as dist


def _on_rccl(group=None):
    return dist.is_initialized() and dist.get_backend(group) == "nccl"


def all_reduce_(t, op=None, group=None):
    """In-place all-reduce of a device tensor.  RCCL takes it as it is.  gloo (CPU tests, 1-GPU rehearsals) gets a host
    copy: its device-tensor path hangs intermittently on this stack when two ranks share a card
    (tools/rehearse/gloo_probe.py: plain torch, no libgmk)."""
    op = dist.ReduceOp.SUM if op is None else op
    if _on_rccl(group) or not t.is_cuda:
        dist.all_reduce(t, op=op, group=group)
    else:
        h = t.cpu()
        dist.all_reduce(h, op=op, group=group)
        t.copy_(h)
    return t


def broadcast_(t, src=0, group=None):
    if _on_rccl(group) or not t.is_cuda:
        dist.broadcast(t, src=src, group=group)
    else:
        h = t.cpu()
        dist.broadcast(h, src=src, group=group)
        t.copy_(h)
    return t


class GradReducer:
    def __init__(self, model, optimizer, process_group=None, force=False):
        """`force`: run the bucketed all-reduce path even in a one-rank group (a sum over one rank is the identity).  A
        one-GPU box can then put the REAL transport -- RCCL's own stream, its event edges to the step's streams, the
        segment-wise replay of the launch plan -- under the step (tests/test_dp_equiv_gpu.py); never set in production."""
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.active = self.world > 1 or (bool(force) and dist.is_initialized())
        self.opt = optimizer
        enc_fc = [model.encoder.fc.weight, model.encoder.fc.bias]
        enc_conv = [p for n, p in model.encoder.named_parameters() if not n.startswith("fc.")]
        dec = list(model.decoder.parameters())
        # bucket name -> flat (start, end); order of completion in backward: decoder, enc.fc, enc convs
        self.buckets = {"decoder": optimizer.segment(dec), "encoder.fc": optimizer.segment(enc_fc),
                        "encoder.conv": optimizer.segment(enc_conv)}
        # encoder.fc's gradient is final ~35 us after the decoder's (1.0 ms step) and sits next to it in the flat buffer: by
        # default the two travel as ONE all-reduce issued at ready("encoder.fc") -- 13.6 MB instead of 5.4 + 8.2 at
        # configs[1], and one collective's worth of stream edges less per step (~1 % of the step on one rank, measured)
        self._fc_dec = optimizer.segment(enc_fc + dec)
        self._work = []
        self._by_name = {}
        # async device-tensor collectives only on RCCL; gloo (rehearsal / CPU-test transport) goes through all_reduce_()
        self._async = _on_rccl(process_group)
        optimizer.grad_scale = 1.0 / self.world

    def ready(self, name, merged=True):
        """Called by train_step when the gradients of bucket `name` have been enqueued.  merged=False: every bucket is its
        own all-reduce (the bucket-wise early AdamW wants the decoder's before encoder.fc's exists)."""
        if not self.active:
            return
        if merged and name == "decoder":
            return                      # travels with encoder.fc
        a, b = self._fc_dec if (merged and name == "encoder.fc") else self.buckets[name]
        if self._async:
            w = dist.all_reduce(self.opt.flat_grad[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self._work.append(w)
            self._by_name[name] = w
        else:
            all_reduce_(self.opt.flat_grad[a:b], group=self.group)
            if self.opt.flat_grad.is_cuda:   # the copy back ran on the caller's current stream: wait() orders behind it
                ev = torch.cuda.Event()
                ev.record()
                self._work.append(ev)

    def wait_bucket(self, name):
        """Orders the CURRENT stream behind bucket `name`'s all-reduce (no host wait on RCCL; gloo reduced synchronously)."""
        w = self._by_name.pop(name, None)
        if w is not None:
            w.wait()

    def wait(self):
        """Orders the CURRENT stream behind every all-reduce issued since the last wait() (no host wait on RCCL)."""
        for w in self._work:
            w.wait()      # c10d Work and torch.cuda.Event alike: a stream-side wait
        self._work = []
        self._by_name = {}


MIN_HW_QUEUES_DP = 8


def ensure_hw_queues(n=MIN_HW_QUEUES_DP):
    """Raises GPU_MAX_HW_QUEUES (HIP's hardware queues per process, default 4) to `n` for a data-parallel rank.  Must run
    before the process's first HIP call (torch.cuda.is_available() is one): the runtime reads it once.

    Why: the step runs on three streams (dgrad chain + two weight-gradient streams, gmk.ops.SideQueue) and HIP maps streams
    onto its hardware queues round-robin.  RCCL takes queues of its own when the group is created; with 4 in all, one
    weight-gradient stream lands on the MAIN stream's hardware queue and its kernels serialise with the dgrad chain.
    Measured on one MI355X with a one-rank RCCL group under the step (tools/rehearse/rccl_world1.sh): 1.173 ms per step
    with 4 queues, 1.050 ms with 8, 1.000 ms without the group."""
    import os
    try:
        cur = int(os.environ.get("GPU_MAX_HW_QUEUES", "4"))
    except ValueError:
        cur = 4
    if cur < n:
        if torch.cuda.is_initialized():
            import warnings
            warnings.warn("gmk.parallel: the HIP runtime is already initialised, GPU_MAX_HW_QUEUES=%d stays; export "
                          "GPU_MAX_HW_QUEUES=%d before starting a data-parallel rank" % (cur, n))
            return cur
        os.environ["GPU_MAX_HW_QUEUES"] = str(n)
        return n
    return cur


def init_from_env():
    """One process per GPU; reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torchrun contract)."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 or os.environ.get("GMK_DP_REHEARSE", "0") == "1":
        ensure_hw_queues()
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # Rehearsal knobs (a 1-GPU box cannot run RCCL between two ranks): GMK_DIST_BACKEND=gloo + GMK_FORCE_DEVICE=0 put
    # every rank on one card and exchange device tensors over gloo; the code path above the transport is the same.
    if os.environ.get("GMK_FORCE_DEVICE") is not None:
        local = int(os.environ["GMK_FORCE_DEVICE"])
    # GMK_DP_REHEARSE=1: a group even for ONE rank, so that GradReducer(force=True) puts RCCL under the step on a one-GPU box
    rehearse = os.environ.get("GMK_DP_REHEARSE", "0") == "1"
    if rehearse:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
    if (world > 1 or rehearse) and not dist.is_initialized():
        backend = os.environ.get("GMK_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available() and backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        elif torch.cuda.is_available():
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    elif torch.cuda.is_available():
        torch.cuda.set_device(local)
    return rank, local, world
