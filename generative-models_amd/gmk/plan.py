"""Launch plans: the train step's launches recorded once, replayed from one C call (gmk_train_step_enqueue).

The step of /root/reference/main.py:27-53 is ~60 kernel launches and ~25 stream fork/join edges.  Issued from Python +
ctypes they cost 0.8 ms of host time per step against 1.0 ms of GPU time at batch 256: the step was one kernel-level
optimisation away from being host-bound, and smaller per-GPU batches were.  A StepPlan runs the step ONCE through the
ordinary Python path while libgmk records every entry-point call (include/gmk.h, "launch plans"); afterwards a step is
one C call per segment that re-issues the same calls on the same streams with the same arguments, except for the
pointers of the step's inputs (image batch, noise), which are substituted.

The plan owns every tensor the recorded step allocated (activations, act' planes, gradients of activations, losses):
the recorded pointers stay valid and nobody else gets that memory.  Outputs of a replayed step are therefore the SAME
tensors every step (overwritten in stream order).
"""
import ctypes as C

from . import ops
from ._lib import lib, check


class StepPlan:
    def __init__(self):
        self.handle = C.c_void_p()
        self.keep = []        # tensors allocated while recording (kept for the plan's lifetime)
        self.marks = []       # (name, segment index) where the host has work of its own (gradient buckets)
        self.out = None       # what the recorded step returned
        self.nseg = 0
        self._from = self._bytes = None
        self._n = 0

    # ------------------------------------------------------------------ recording
    def __enter__(self):
        if ops._rec_keep is not None:
            raise RuntimeError("a launch plan is already being recorded")
        check(lib().gmk_plan_begin(C.byref(self.handle)))
        ops._rec_keep = self.keep
        return self

    def __exit__(self, et, ev, tb):
        ops._rec_keep = None
        check(lib().gmk_plan_end())
        self.nseg = lib().gmk_plan_segments(self.handle)
        if et is not None:
            self.close()
        return False

    def mark(self, name):
        """Closes the current segment; the host work named `name` happens between it and the next one."""
        seg = lib().gmk_plan_mark()
        if seg < 0:
            check(seg)
        self.marks.append((name, seg))

    def set_inputs(self, *tensors):
        """The recorded step's per-step inputs: replay(*new) substitutes their addresses."""
        ts = [t for t in tensors if t is not None]
        self._n = len(ts)
        self._from = (C.c_void_p * self._n)(*[t.data_ptr() for t in ts])
        self._bytes = (C.c_size_t * self._n)(*[t.numel() * t.element_size() for t in ts])
        self._shapes = [tuple(t.shape) for t in ts]
        self.keep.extend(ts)

    # ------------------------------------------------------------------ replay
    def enqueue(self, seg_begin, seg_end, *inputs):
        ts = [t for t in inputs if t is not None]
        if len(ts) != self._n or any(tuple(t.shape) != s for t, s in zip(ts, self._shapes)):
            raise ValueError("launch plan replayed with inputs of another shape than it was recorded with")
        to = (C.c_void_p * self._n)(*[t.data_ptr() for t in ts])
        check(lib().gmk_train_step_enqueue(self.handle, seg_begin, seg_end, self._n, self._from, to, self._bytes))

    def close(self):
        if self.handle:
            lib().gmk_plan_destroy(self.handle)
            self.handle = C.c_void_p()
        self.keep = []
        self.out = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
