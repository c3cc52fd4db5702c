"""FusedAdamW: a torch.optim.AdamW whose step() is ONE libgmk launch over a flat buffer.

It *is* a torch.optim.AdamW (isinstance, param_groups, state_dict()/load_state_dict() layout
identical to the reference's checkpoint['optimizer'], /root/reference/main.py:214,219,326), but
parameters, gradients and both moment buffers live in four flat device buffers; the per-parameter
tensors (p.data, p.grad, state['exp_avg'], state['exp_avg_sq']) are views into them.  The flat
gradient buffer is also what the data-parallel reducer all-reduces (gmk/parallel.py).
"""
import torch

from .ops import _stream as ops_stream
from ._lib import lib, check


class FusedAdamW(torch.optim.AdamW):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, **kw):
        # the kernel implements plain AdamW (what the reference constructs, main.py:214); options that change the
        # update rule are refused instead of being silently ignored
        for flag in ("amsgrad", "maximize", "capturable", "differentiable"):
            if kw.pop(flag, False):
                raise NotImplementedError(f"FusedAdamW: {flag}=True is not implemented by gmk_adamw_step")
        for flag in ("foreach", "fused"):   # implementation selectors of torch's own step(): meaningless here
            kw.pop(flag, None)
        if kw:
            raise TypeError(f"FusedAdamW: unexpected arguments {sorted(kw)}")
        if isinstance(lr, torch.Tensor):
            raise NotImplementedError("FusedAdamW: a tensor lr is not supported (the kernel takes a host scalar)")
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        if len(self.param_groups) != 1:
            raise NotImplementedError("FusedAdamW supports a single param group (the reference uses one)")
        self.grad_scale = 1.0  # set to 1/world_size by the DP reducer (SUM all-reduce)
        self._early = []       # flat ranges already updated by step_range() for the step in progress
        self._t = 0            # optimizer steps taken (= the shared state['step'] tensor)
        self._state_ready = False
        self._flatten()

    # ------------------------------------------------------------------ flat storage
    def _flatten(self):
        ps = self.param_groups[0]["params"]
        dev = ps[0].device
        if any(p.device != dev or p.dtype != torch.float32 for p in ps):
            raise RuntimeError("FusedAdamW needs fp32 parameters on one device (move the model first)")
        sizes = [p.numel() for p in ps]
        # every tensor starts on a 4-float (16-byte) boundary so kernels can use float4 everywhere
        self._offsets, off = [], 0
        for n in sizes:
            self._offsets.append(off)
            off += (n + 3) // 4 * 4
        self._numel = off
        self.flat_param = torch.zeros(off, device=dev, dtype=torch.float32)
        self.flat_grad = torch.zeros(off, device=dev, dtype=torch.float32)
        self.flat_m = torch.zeros(off, device=dev, dtype=torch.float32)
        self.flat_v = torch.zeros(off, device=dev, dtype=torch.float32)
        for p, o in zip(ps, self._offsets):
            n = p.numel()
            self.flat_param[o:o + n].copy_(p.data.reshape(-1))
            p.data = self.flat_param[o:o + n].view(p.shape)
            g = self.flat_grad[o:o + n].view(p.shape)
            if p.grad is not None:
                g.copy_(p.grad)
            p.grad = g
            st = self.state.get(p)
            m = self.flat_m[o:o + n].view(p.shape)
            v = self.flat_v[o:o + n].view(p.shape)
            if st is not None and "exp_avg" in st:
                m.copy_(st["exp_avg"]); v.copy_(st["exp_avg_sq"])
                st["exp_avg"], st["exp_avg_sq"] = m, v
        self._views_ok()

    def _views_ok(self):
        ps = self.param_groups[0]["params"]
        base_p, base_g = self.flat_param.data_ptr(), self.flat_grad.data_ptr()
        for p, o in zip(ps, self._offsets):
            if p.data_ptr() != base_p + 4 * o or p.grad is None or p.grad.data_ptr() != base_g + 4 * o:
                return False
        return True

    def _views_ok_fast(self):
        """The per-step check: every parameter still has a gradient and the first and last still sit in the flat buffers
        (model.to() / load_state_dict re-home all of them together); the full walk costs 20 us of host time per step."""
        ps = self.param_groups[0]["params"]
        for p in ps:
            if p.grad is None:
                return False
        base_p, base_g = self.flat_param.data_ptr(), self.flat_grad.data_ptr()
        for p, o in ((ps[0], self._offsets[0]), (ps[-1], self._offsets[-1])):
            if p.data_ptr() != base_p + 4 * o or p.grad.data_ptr() != base_g + 4 * o:
                return False
        return True

    def segment(self, params):
        """(start, end) float offsets of the flat range spanned by `params` (must be contiguous in order)."""
        ps = self.param_groups[0]["params"]
        idx = sorted(next(i for i, q in enumerate(ps) if q is p) for p in params)
        assert idx == list(range(idx[0], idx[-1] + 1)), "params of a bucket must be adjacent"
        start = self._offsets[idx[0]]
        end = self._offsets[idx[-1] + 1] if idx[-1] + 1 < len(ps) else self._numel
        return start, end

    # ------------------------------------------------------------------ torch.optim API
    def zero_grad(self, set_to_none=False):
        """Keeps the flat gradient views alive (the kernels write into them); zeroes them in one fill."""
        if not self._views_ok():
            self._flatten()
        self.flat_grad.zero_()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._state_ready = False   # the step count is re-read from the loaded state['step']
        self._flatten()

    def _prepare(self):
        if not self._views_ok_fast() or (self._t % 64 == 0 and not self._views_ok()):
            self._flatten()  # someone re-assigned p.data / p.grad (e.g. model.to()): re-home them
        group = self.param_groups[0]
        ps = group["params"]
        if self.flat_param.device.type != "cuda":
            raise RuntimeError("FusedAdamW.step() runs a gfx950 kernel: parameters must be on an MI355X (no CPU path)")
        if not self._state_ready:
            # state['step'] bookkeeping like torch (a float32 scalar tensor per parameter) -- except that all parameters
            # share ONE tensor object: bumping 24 separate CPU tensors every step cost 50 us of host time.  state_dict()
            # still shows a 'step' per parameter, and a stock torch.optim.AdamW that loads it makes its own copies.
            first = self.state[ps[0]]
            t0 = float(first["step"]) if "step" in first else 0.0
            self._step_t = torch.tensor(t0, dtype=torch.float32)
            for p, o in zip(ps, self._offsets):
                st = self.state[p]
                n = p.numel()
                if "exp_avg" not in st:
                    st["exp_avg"] = self.flat_m[o:o + n].view(p.shape)
                    st["exp_avg_sq"] = self.flat_v[o:o + n].view(p.shape)
                st["step"] = self._step_t
            self._t = int(t0)
            self._state_ready = True
        return group, ps

    def _launch(self, group, a, b, step):
        b1, b2 = group["betas"]
        check(lib().gmk_adamw_step(self.flat_param.data_ptr() + 4 * a, self.flat_grad.data_ptr() + 4 * a,
                                   self.flat_m.data_ptr() + 4 * a, self.flat_v.data_ptr() + 4 * a, b - a, float(group["lr"]),
                                   float(b1), float(b2), float(group["eps"]), float(group["weight_decay"]), step,
                                   float(self.grad_scale), ops_stream()))

    @torch.no_grad()
    def step_range(self, a, b):
        """AdamW on the flat range [a, b) NOW, on the current stream, as an early part of the step that the next step()
        call completes: step() then updates only the ranges not yet done and advances the step count once.  Lets
        Autoencoder.train_step(optimizer=...) update a gradient bucket as soon as it is final, under the rest of backward
        (same arithmetic, same step number, same lr as the step() that follows)."""
        group, ps = self._prepare()
        self._launch(group, a, b, self._t + 1)
        self._early.append((a, b))

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        group, ps = self._prepare()
        self._t += 1
        self._step_t += 1
        step = self._t
        pos = 0
        for a, b in sorted(self._early) + [(self._numel, self._numel)]:   # the complement of the early ranges
            if a > pos:
                self._launch(group, pos, a, step)
            pos = max(pos, b)
        self._early = []
        return loss
