"""Encoder / Decoder / Autoencoder with the reference's Python surface on libgmk kernels.

Mirrors /root/reference/models/autoencoders.py: same class names and module path (full-model
pickles resolve `models.autoencoders.Autoencoder`), same constructor arguments, same child
modules and parameter names/shapes (state_dicts are interchangeable), same side-channel
attributes (`.stats`, `.normal`, `.latent_vector`) and `.sample()`.

nn.Conv2d / nn.ConvTranspose2d / nn.Linear children are kept as *parameter containers* (their
default initialisation is therefore bit-identical to the reference under the same seed); their
forward is never called: all arithmetic runs in hand-written gfx950 kernels (gmk/engine.py).
There is no PyTorch fallback: CPU tensors or a missing libgmk.so raise.

Besides the drop-in autograd path (`recon = model(image)` then any loss), `train_step()` runs the
whole step (forward, fused sigmoid+MSE+KL loss kernel, backward) without autograd bookkeeping.
"""
import os

import torch
import torch.nn as nn

from gmk import engine, ops, act_id
from utils import print_debug


def _local_params(module):
    return {n: p for n, p in module.named_parameters()}


class _EncoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, x, names, train, *params):
        p = dict(zip(names, params))
        out, sv = engine.encoder_forward(p, x.contiguous(), act=mod._act_id, no_pool=mod.no_pool,
                                         ksize=mod.kernel_size, train=train)
        ctx.mod, ctx.names, ctx.p, ctx.sv = mod, names, p, sv
        return out

    @staticmethod
    def backward(ctx, dout):
        mod = ctx.mod
        g = engine.encoder_backward(ctx.p, ctx.sv, dout.contiguous(), no_pool=mod.no_pool, ksize=mod.kernel_size,
                                    need_dx=ctx.needs_input_grad[1])
        dx = g.pop("__dx__", None)
        return (None, dx, None, None) + tuple(g[n] for n in ctx.names)


class _DecoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, z, names, train, *params):
        p = dict(zip(names, params))
        out, sv = engine.decoder_forward(p, z.contiguous(), act=mod._act_id, num_channels=mod.num_channels,
                                         no_upsample=mod.no_upsample, interpolation=mod._interp,
                                         ksize=mod.kernel_size, base=mod.base, train=train,
                                         final_act="sigmoid" if mod.use_sigmoid else None)
        ctx.mod, ctx.names, ctx.p, ctx.sv = mod, names, p, sv
        return out

    @staticmethod
    def backward(ctx, dout):
        mod = ctx.mod
        dpre = dout.contiguous()
        if mod.use_sigmoid:
            dpre = ops.ew_mul(dpre, ctx.sv["e5"])
        g, dz = engine.decoder_backward(ctx.p, ctx.sv, dpre, no_upsample=mod.no_upsample,
                                        interpolation=mod._interp, ksize=mod.kernel_size,
                                        need_dz=ctx.needs_input_grad[1])
        return (None, dz, None, None) + tuple(g[n] for n in ctx.names)


class _ReparamFn(torch.autograd.Function):
    """z = mu + exp(0.5*log_var) * eps on the interleaved (B,L,2) stats (autoencoders.py:169-176)."""
    @staticmethod
    def forward(ctx, stats, eps):
        stats = stats.contiguous()
        ctx.save_for_backward(stats, eps)
        return ops.reparam_fwd(stats, eps)

    @staticmethod
    def backward(ctx, dz):
        stats, eps = ctx.saved_tensors
        return ops.reparam_bwd(dz.contiguous(), stats, eps), None


def _padding(kernel_size):
    if kernel_size == 3:
        return 1
    if kernel_size == 5:
        return 2
    print(f'\n\nkernel_size {kernel_size} is not supported\n\n')  # reference: autoencoders.py:31-32
    raise NotImplementedError(f'kernel_size {kernel_size} is not supported')


class Encoder(nn.Module):
    # A full-model pickle written by the reference (main.py:338-339, loaded at main.py:189-195) restores __dict__
    # without running __init__: its Encoder has no `base` (8x8 is hard-coded there, autoencoders.py:41), so the
    # class-level default stands in.
    base = 8

    def __init__(self, variational=False, latent_size=None, num_channels=None, kernel_size=None, act=None,
                 no_pool=False, debug=False, base=8) -> None:
        super().__init__()
        self.debug = debug
        self.variational = variational
        mult = 2 if variational else 1
        self.latent_size = latent_size
        self.no_pool = no_pool
        self.num_channels = num_channels
        self.kernel_size = kernel_size
        self.base = base  # spatial size after two /2 stages (reference hard-codes 8, autoencoders.py:41)
        padding = _padding(kernel_size)
        C = num_channels
        self.conv1 = nn.Conv2d(3, C, kernel_size, stride=1, padding=padding)
        self.conv2 = nn.Conv2d(C, C, kernel_size, stride=1, padding=padding)
        self.conv3 = nn.Conv2d(C, C, kernel_size, stride=1, padding=padding)
        if self.no_pool:
            self.downsample1 = nn.Conv2d(C, C, kernel_size, stride=2, padding=padding)
            self.downsample2 = nn.Conv2d(C, C, kernel_size, stride=2, padding=padding)
        else:
            self.max_pool = nn.MaxPool2d(kernel_size=2, stride=2)
        self.fc = nn.Linear(in_features=base * base * C, out_features=mult * latent_size)
        self.act = act

    @property
    def _act_id(self):
        return act_id(self.act)

    def forward(self, x):  # input x is (bs, 3, 32, 32) for cifar images
        print_debug(x, self.debug, name='\nEncoder: input')
        if x.shape[-1] != 4 * self.base or x.shape[-2] != 4 * self.base:
            raise RuntimeError(f'Encoder built for {4 * self.base}x{4 * self.base} inputs, got {tuple(x.shape)}')
        lp = _local_params(self)
        names = tuple(lp.keys())
        train = torch.is_grad_enabled() and any(p.requires_grad for p in lp.values())
        x = _EncoderFn.apply(self, x, names, train, *lp.values())
        print_debug(x, self.debug, name='Encoder: after fc')
        if self.variational:
            out = x.view(x.shape[0], self.latent_size, 2)
        else:
            out = x
        return out


class Decoder(nn.Module):
    base = 8   # see Encoder.base: reference pickles carry neither `base` nor `interpolation` (autoencoders.py:68-91)

    @property
    def _interp(self):
        """Upsampling mode: our own attribute, or (reference-written pickle) the nn.Upsample child's mode."""
        mode = self.__dict__.get("interpolation")
        if mode is None and not self.no_upsample:
            mode = self.upsample.mode
        return mode or "nearest"

    def __init__(self, latent_size, num_channels=None, kernel_size=None, act=None, sigmoid=None,
                 interpolation='bilinear', no_upsample=False, debug=False, base=8) -> None:
        super().__init__()
        self.debug = debug
        self.num_channels = num_channels
        self.use_sigmoid = sigmoid
        self.no_upsample = no_upsample
        self.kernel_size = kernel_size
        self.interpolation = interpolation
        self.base = base
        padding = _padding(kernel_size)
        C = num_channels
        self.fc = nn.Linear(in_features=latent_size, out_features=C * base * base)
        self.conv1 = nn.Conv2d(C, C, kernel_size, stride=1, padding=padding)
        self.conv2 = nn.Conv2d(C, C, kernel_size, stride=1, padding=padding)
        self.conv3 = nn.Conv2d(C, 3, kernel_size, stride=1, padding=padding)
        if self.no_upsample:
            self.deconv1 = nn.ConvTranspose2d(C, C, kernel_size, stride=2, padding=padding, output_padding=1)
            self.deconv2 = nn.ConvTranspose2d(C, C, kernel_size, stride=2, padding=padding, output_padding=1)
        else:
            self.upsample = nn.Upsample(scale_factor=2, mode=interpolation)
        self.act = act
        self.sigmoid = nn.Sigmoid()

    @property
    def _act_id(self):
        return act_id(self.act)

    def forward(self, x):  # x shape: (bs, latent_size)
        print_debug(x, self.debug, name='Decoder: input')
        lp = _local_params(self)
        names = tuple(lp.keys())
        train = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in lp.values()))
        out = _DecoderFn.apply(self, x, names, train, *lp.values())
        print_debug(out, self.debug, name='Decoder: after sigmoid')
        return out


class Autoencoder(nn.Module):
    def __init__(
        self,
        variational=False,
        latent_size=256,
        num_channels=64,
        kernel_size=3,
        act=nn.ReLU,
        sigmoid=True,
        interpolation='bilinear',
        no_pool=False,
        no_upsample=False,
        device=None,
        debug=False,
        image_size=32,
    ) -> None:
        super().__init__()
        base = image_size // 4  # reference: 8 (32x32 CIFAR); config 5 uses 64x64 -> 16
        self.encoder = Encoder(variational=variational, latent_size=latent_size, num_channels=num_channels,
                               kernel_size=kernel_size, act=act, no_pool=no_pool, debug=debug, base=base)
        self.decoder = Decoder(latent_size, num_channels=num_channels, kernel_size=kernel_size, act=act,
                               sigmoid=sigmoid, interpolation=interpolation, no_upsample=no_upsample,
                               debug=debug, base=base)
        self.variational = variational
        self.latent_size = latent_size
        self.sigmoid = nn.Sigmoid()
        self.device = device

    def forward(self, x):
        out = self.encoder(x)
        if self.variational:
            self.stats = out
            means = self.stats[:, :, 0]
            # reparametrization trick: sample a value as mean + std * N(0, 1)
            self.normal = torch.randn_like(means).contiguous()
            self.latent_vector = _ReparamFn.apply(self.stats, self.normal)  # (num_images, latent)
        else:
            self.latent_vector = out
        return self.decoder(self.latent_vector)

    def sample(self, name=None, num_images=8, return_samples=False):
        self.eval()
        latent_vector = torch.randn(size=(num_images, self.latent_size), device=self.device)
        out = self.decoder(latent_vector)
        if return_samples:
            return out
        from utils import save_image_grid
        save_image_grid(out.detach().cpu(), f'plots/samples_{name}.png', nrow=4, figsize=(7, 4.5))

    @torch.no_grad()
    def encode_features(self, x):
        """latent_vector of `x` without running the decoder (what metrics.compute_fid consumes;
        the reference runs the full model and discards the reconstruction, metrics.py:70-73)."""
        enc = self.encoder
        pe = {n: p.data for n, p in enc.named_parameters()}
        if self.variational:
            self.normal = torch.randn((x.shape[0], self.latent_size), device=x.device, dtype=torch.float32)
            self.stats, sv = engine.encoder_forward(pe, x.contiguous(), act=enc._act_id, no_pool=enc.no_pool,
                                                    ksize=enc.kernel_size, train=False, reparam_eps=self.normal)
            self.latent_vector = sv["z"]
        else:
            self.latent_vector, _ = engine.encoder_forward(pe, x.contiguous(), act=enc._act_id, no_pool=enc.no_pool,
                                                           ksize=enc.kernel_size, train=False)
        return self.latent_vector

    @torch.no_grad()
    def eval_losses(self, image, beta=0.0, mse_prior=False, eps=None):
        """No-grad forward + the fused loss kernel in its eval form: what one iteration of the reference's
        compute_test_loss evaluates (/root/reference/main.py:84-102).  Returns a device tensor
        [rec, kl (unweighted), rec + beta*kl, mean |mu|, mean sigma]; sets .stats/.normal/.latent_vector like forward().
        `eps` replaces the randn_like draw (the reference samples in eval mode as well, autoencoders.py:174)."""
        enc, dec = self.encoder, self.decoder
        pe = {n: p.data for n, p in enc.named_parameters()}
        pd = {n: p.data for n, p in dec.named_parameters()}
        image = image.contiguous()
        B = image.shape[0]
        if self.variational:
            if eps is None:
                eps = torch.randn((B, self.latent_size), device=image.device, dtype=torch.float32)
            eps = eps.contiguous()
            stats, sv = engine.encoder_forward(pe, image, act=enc._act_id, no_pool=enc.no_pool, ksize=enc.kernel_size,
                                               train=False, reparam_eps=eps)
            z = sv["z"]
            self.stats, self.normal = stats, eps
        else:
            z, _ = engine.encoder_forward(pe, image, act=enc._act_id, no_pool=enc.no_pool, ksize=enc.kernel_size, train=False)
            stats = None
        self.latent_vector = z
        pre, _ = engine.decoder_forward(pd, z, act=dec._act_id, num_channels=dec.num_channels,
                                        no_upsample=dec.no_upsample, interpolation=dec._interp,
                                        ksize=dec.kernel_size, base=dec.base, train=False, final_act=None)
        out5, _ = ops.vae_loss_eval(pre, image, stats, beta, sigmoid=bool(dec.use_sigmoid), mse_prior=mse_prior)
        return out5

    def _step_tensors(self):
        """name -> parameter / gradient tensors of encoder and decoder for train_step, rebuilt only when a parameter's or
        gradient's storage moved (walking named_parameters() four times cost 0.15 ms of host time per step)."""
        c = getattr(self, "_step_cache", None)
        mods = (id(self.encoder), id(self.decoder))
        if c is None or c[0] != mods:
            c = self._step_cache = [mods, list(self.encoder.named_parameters()), list(self.decoder.named_parameters()), None, None]
        for _, p in c[1] + c[2]:
            if p.grad is None:
                p.grad = torch.empty_like(p.data)
        key = tuple((p.data_ptr(), p.grad.data_ptr()) for _, p in c[1] + c[2])
        if c[3] != key:
            c[3] = key
            c[4] = ({n: p.data for n, p in c[1]}, {n: p.data for n, p in c[2]},
                    {n: p.grad for n, p in c[1]}, {n: p.grad for n, p in c[2]})
        return c[4]

    # ------------------------------------------------------------------ fused train step
    def train_step(self, image, beta=0.0, mse_prior=False, eps=None, reducer=None, optimizer=None):
        """Forward + fused loss + backward of /root/reference/main.py:29-51 without autograd.

        Overwrites `.grad` of every parameter (what zero_grad + backward leave behind) and returns a
        device tensor [rec, kl, total] (kl unweighted, total = rec + beta*kl).  The optimizer step
        stays with the caller (main.py:52-53).  `eps` replaces torch.randn for parity tests.
        `reducer` (gmk.parallel.GradReducer) is told as soon as each gradient bucket is enqueued so
        that its all-reduce overlaps the rest of backward.
        `optimizer` (gmk.optim.FusedAdamW, optional): the decoder and encoder.fc buckets (95 % of the parameters) get
        their AdamW update as soon as their gradients are final (and all-reduced), on a third stream under the encoder's
        backward pass -- nothing later in backward reads those weights.  The caller still calls optimizer.step() after
        train_step, as main.py:52 does: it updates the rest and advances the step count (FusedAdamW.step_range).
        Bit-identical to the plain order; measured at B = 256 on one GPU it hides 15 us of GPU time and costs 120 us of
        host time, so nothing passes it by default.

        Launch plan (gmk/plan.py; GMK_STEP_PLAN=0 turns it off): the first step with a given (batch shape, beta, parameter
        storage, stream) runs through the Python path below while libgmk records its ~85 entry-point calls; every later
        step with the same key is ONE C call per gradient bucket that re-issues them with the new image / eps pointers
        (gmk_train_step_enqueue): bit-identical results, 0.8 -> ~0.2 ms of host time per step.  A replayed step returns
        the SAME tensors every step (losses, .stats, .latent_vector are overwritten in stream order): read or copy what
        you want to keep before the next step.  One plan is cached per model (a new key re-records).
        """
        if os.environ.get("GMK_STEP_PLAN", "1") == "0" or torch.cuda.is_current_stream_capturing():
            return self._train_step_py(image, beta, mse_prior, eps, reducer, optimizer)
        from gmk.plan import StepPlan
        image = image.contiguous()
        B = image.shape[0]
        if self.variational:
            if eps is None:
                eps = torch.randn((B, self.latent_size), device=image.device, dtype=torch.float32)
            eps = eps.contiguous()
        else:
            eps = None
        main_stream = torch.cuda.current_stream(image.device)
        self._step_tensors()
        dp = reducer is not None and reducer.active
        side = ops.side_queue(image.device)
        early = self._early_ctx(optimizer, side, image.device, main_stream)
        key = (tuple(image.shape), float(beta), bool(mse_prior), main_stream.cuda_stream, dp, self._step_cache[3],
               bool(self.decoder.use_sigmoid), self.encoder._act_id, self.decoder._interp, os.environ.get("GMK_SIDE_WGRAD", "1"),
               os.environ.get("GMK_DEFER_REDUCE", "0"))
        cached = getattr(self, "_step_plan", None)
        if cached is None or cached[0] != key:
            if cached is not None:      # frees the old plan's activations before the new recording allocates its own
                cached[1].close()
                self._step_plan = None
            plan = StepPlan()
            with plan:
                plan.out = self._train_step_py(image, beta, mse_prior, eps, reducer, optimizer, rec=plan)
            plan.set_inputs(image, eps)
            plan.stats, plan.z = (self.stats if self.variational else None), self.latent_vector
            self._step_plan = (key, plan)
            return plan.out
        plan = cached[1]
        if dp or early is not None:   # the host has work of its own between the segments
            a = 0
            for name, seg in plan.marks:
                plan.enqueue(a, seg + 1, image, eps)
                a = seg + 1
                self._bucket_ready(name, reducer, side, early)
            plan.enqueue(a, plan.nseg, image, eps)
            if early is not None:
                main_stream.wait_stream(early[2])
        else:
            plan.enqueue(0, plan.nseg, image, eps)
        if self.variational:
            self.stats, self.normal = plan.stats, eps
        self.latent_vector = plan.z
        return plan.out

    def _early_ctx(self, optimizer, side, device, main_stream):
        """(optimizer, bucket segments, update stream, main stream) for the bucket-wise early AdamW, or None."""
        if optimizer is None or side is None or not hasattr(optimizer, "step_range"):
            return None
        return (optimizer, ops.bucket_segments(self, optimizer), ops.aux_stream(device), main_stream)

    @staticmethod
    def _bucket_ready(name, reducer, side, early=None):
        """Gradient bucket `name` is enqueued: start its all-reduce behind the weight-gradient streams and, with `early`,
        its AdamW update behind that."""
        if reducer is not None and reducer.active:
            merged = early is None
            if side is None:
                reducer.ready(name, merged)
            elif not (merged and name == "decoder"):
                # the bucket's gradients are ordered on the weight-gradient streams: the all-reduce is issued behind them.
                # (Issued from a stream of its own that waits for both -- so that neither waits for the other's backlog --
                # the step took 1.73 ms instead of 1.01: a queue whose head is a long-pending wait slows every dispatch.)
                side.gather()
                with torch.cuda.stream(side.stream):
                    reducer.ready(name, merged)
        if early is not None and name != "encoder.conv":
            optimizer, seg, opt_stream, main_stream = early
            # the update must follow the bucket's weight gradients (side stream), the kernels that read its weights
            # (main stream, all enqueued by now) and its all-reduce
            side.gather()
            opt_stream.wait_stream(side.stream)
            opt_stream.wait_stream(main_stream)
            if reducer is not None:   # RCCL work handles order torch's current stream
                with torch.cuda.stream(opt_stream):
                    reducer.wait_bucket(name)
            with ops.stream_scope(opt_stream):
                optimizer.step_range(*seg[name])

    def _train_step_py(self, image, beta=0.0, mse_prior=False, eps=None, reducer=None, optimizer=None, rec=None):
        """The step launch by launch from Python (and the recording pass of a launch plan: `rec`)."""
        # one stream lookup per step: every gmk op below takes the handle from the scope (gmk.ops.stream_scope)
        main_stream = torch.cuda.current_stream(image.device)
        with ops.stream_scope(main_stream):
            sigmoid = bool(self.decoder.use_sigmoid)
            enc, dec = self.encoder, self.decoder
            pe, pd, ge, gd = self._step_tensors()
            image = image.contiguous()
            B = image.shape[0]
            if self.variational:
                if eps is None:
                    eps = torch.randn((B, self.latent_size), device=image.device, dtype=torch.float32)
                eps = eps.contiguous()
                # encoder head + reparameterisation: one GEMM whose reduction epilogue writes stats and z
                stats, sve = engine.encoder_forward(pe, image, act=enc._act_id, no_pool=enc.no_pool, ksize=enc.kernel_size,
                                                    train=True, reparam_eps=eps)
                z = sve["z"]
                self.stats, self.normal = stats, eps
            else:
                out, sve = engine.encoder_forward(pe, image, act=enc._act_id, no_pool=enc.no_pool, ksize=enc.kernel_size, train=True)
                stats, z = None, out
            self.latent_vector = z
            pre, svd = engine.decoder_forward(pd, z, act=dec._act_id, num_channels=dec.num_channels,
                                              no_upsample=dec.no_upsample, interpolation=dec._interp,
                                              ksize=dec.kernel_size, base=dec.base, train=True, final_act=None)
            # sigmoid + MSE + beta*KL and d/d pre in one pass; the KL gradient rides with the reparameterisation backward in
            # the decoder fc's dgrad epilogue below, so no dstats round trip and no separate reparam launches
            losses, dpre, _, _ = ops.vae_loss_fwd_bwd(pre, image, stats, beta, sigmoid=sigmoid, mse_prior=mse_prior,
                                                      want_dstats=False, fused=True)
            # weight gradients run on a side stream, overlapping the dgrad chain on this one (gmk.ops.SideQueue)
            side = ops.side_queue(image.device)
            if side is not None:
                side.begin(main_stream)

            early = self._early_ctx(optimizer, side, image.device, main_stream)

            def bucket_ready(name):
                if rec is not None:
                    rec.mark(name)
                self._bucket_ready(name, reducer, side, early)

            # GMK_DEFER_REDUCE=1: the slab reductions of the 3x3 weight gradients are left to ONE launch per gradient bucket
            # (gmk_conv_wgrad_reduce_batch: 8 reduce launches -> 2).  Bit-identical; measured 1.037 against 1.026 ms per step
            # in the same call (the gradients of a whole bucket become final later), so it is off by default
            dfr = [] if os.environ.get("GMK_DEFER_REDUCE", "0") != "0" else None

            def flush_reduces():
                if dfr:
                    if side is not None:
                        side.reduce_batch(dfr)
                    else:
                        ops.wgrad_reduce_batch(dfr)
                    del dfr[:]

            rp = (stats, eps, float(beta) / B, mse_prior) if self.variational else None
            _, dz = engine.decoder_backward(pd, svd, dpre, no_upsample=dec.no_upsample, interpolation=dec._interp,
                                            ksize=dec.kernel_size, need_dz=True, gout=gd, side=side, reparam=rp, defer=dfr)
            flush_reduces()
            bucket_ready("decoder")
            dout = dz.view(B, -1)
            engine.encoder_backward(pe, sve, dout, no_pool=enc.no_pool, ksize=enc.kernel_size, gout=ge,
                                    on_fc_done=lambda: bucket_ready("encoder.fc"), side=side, defer=dfr)
            flush_reduces()
            bucket_ready("encoder.conv")
            if side is not None:
                side.join()
        if early is not None:
            main_stream.wait_stream(early[2])
        return losses
