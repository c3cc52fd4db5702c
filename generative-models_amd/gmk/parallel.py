"""Data-parallel gradient reduction over RCCL (torch.distributed backend "nccl" on ROCm).

The train step shards by batch (pure DP): losses are batch means, so equal local
batches + gradient averaging reproduce the single-process global-batch gradient.  Gradients live
in FusedAdamW's flat buffer; it is all-reduced (SUM) in a few contiguous buckets that are launched
as soon as the backward pass has produced them -- decoder first, then encoder.fc, then the encoder
convs -- on the process group's own stream, overlapping the rest of backward.  The 1/world factor is
folded into the AdamW kernel (optimizer.grad_scale), so there is no extra pass over the gradients.
"""
import torch
import torch.distributed as dist


class GradReducer:
    def __init__(self, model, optimizer, process_group=None):
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.opt = optimizer
        enc_fc = [model.encoder.fc.weight, model.encoder.fc.bias]
        enc_conv = [p for n, p in model.encoder.named_parameters() if not n.startswith("fc.")]
        dec = list(model.decoder.parameters())
        # bucket name -> flat (start, end); order of completion in backward: decoder, enc.fc, enc convs
        self.buckets = {"decoder": optimizer.segment(dec), "encoder.fc": optimizer.segment(enc_fc),
                        "encoder.conv": optimizer.segment(enc_conv)}
        self._work = []
        optimizer.grad_scale = 1.0 / self.world

    def ready(self, name):
        """Called by train_step when the gradients of bucket `name` have been enqueued."""
        if self.world == 1:
            return
        a, b = self.buckets[name]
        self._work.append(dist.all_reduce(self.opt.flat_grad[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def wait(self):
        for w in self._work:
            w.wait()
        self._work = []


def init_from_env():
    """One process per GPU; reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torchrun contract)."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        backend = "nccl" if torch.cuda.is_available() else "gloo"
        if torch.cuda.is_available():
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    elif torch.cuda.is_available():
        torch.cuda.set_device(local)
    return rank, local, world
