"""Optimizer side of the hot path: timm's decay / no-decay parameter grouping (timm 0.4.12
optim_factory.add_weight_decay, call site main_pretrain.py:253) and AdamW(betas=(0.9, 0.95)) (:254) as ONE fused
HIP launch over the parameter arena (ecamp_adamw_grouped), which also refreshes the bf16 shadow weights."""
import collections
import os

import torch

from . import hip_ops as ops

# How many (micro-)steps the host may queue ahead of the GPU.  The host queues a step in ~15 ms, the GPU runs it in ~36: unbounded, the
# host's lead grows by 20 ms per step, and every tensor a side stream reads (record_stream: the weight-gradient stream's x and dy, i.e.
# most saved activations, ~10 GB per step) is only returned to the caching allocator when the GPU gets there -- round 6's bench record
# showed 250-570 hipMalloc calls and 27-60 GB of reserve growth inside 20 timed steps, and steps of 2x the median when such a burst met
# a small lead (BENCH_r05's 40.7 ms mean).  Two steps in flight keep the GPU fed through a host hiccup of ~70 ms and reach their steady state within a 5-step warm-up
# (three: +35 ms of slack, but the lead is still growing when the driver's timed region starts).  0 = unbounded (the old behaviour).
MAX_STEPS_IN_FLIGHT = int(os.environ.get("ECAMP_MAX_STEPS_IN_FLIGHT", "2"))
# ... and fewer than that (never fewer than one) while the tensors held for the side streams (hip_ops.hold: ~21 GB per step at B = 256) pass this
MAX_HELD_BYTES = int(float(os.environ.get("ECAMP_MAX_HELD_GB", "96")) * 2 ** 30)


def add_weight_decay(model, weight_decay=1e-5, skip_list=()):
    """1-D tensors and names ending in '.bias' get no decay; everything else (incl. the 3-D cls/mask tokens and the
    embedding tables) is decayed.  Returns [no_decay group, decay group] like timm."""
    decay, no_decay = [], []
    for name, param in model.named_parameters():
        if not param.requires_grad:
            continue
        if len(param.shape) == 1 or name.endswith(".bias") or name in skip_list:
            no_decay.append(param)
        else:
            decay.append(param)
    return [{"params": no_decay, "weight_decay": 0.0}, {"params": decay, "weight_decay": weight_decay}]


class FusedAdamW(torch.optim.Optimizer):
    """torch.optim.AdamW semantics (decoupled decay, bias correction, eps outside the sqrt) on the flat arenas.
    `param_groups` behave as usual (lr schedulers write `group['lr']`); parameters whose gradient never arrives in
    the reference (`_ecamp_unused`, the BERT pooler) are skipped entirely, exactly as torch skips `grad is None`."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        if len(self.param_groups) > 8:
            raise ValueError("FusedAdamW supports at most 8 param groups")
        b = {tuple(g["betas"]) for g in self.param_groups}
        e = {g["eps"] for g in self.param_groups}
        if len(b) != 1 or len(e) != 1:
            raise ValueError("all groups must share betas and eps")
        self._arena = None
        self._step = 0
        self._step_dev = None       # f32[1] on the device: the count of steps actually taken when a loss scaler decides skips there (`ctl`)
        self.grad_scale = 1.0
        self.bucketwise_steps = 0   # optimizer steps that ran bucket by bucket behind the gradient all-reduces
        self._inflight = collections.deque()

    def pace(self):
        """Called where a (micro-)step ends: records an event on the compute stream and blocks the host until the step
        MAX_STEPS_IN_FLIGHT back has finished on the GPU (see MAX_STEPS_IN_FLIGHT).  No device-wide synchronisation."""
        if not torch.cuda.is_available():
            return
        ops.seal()   # tensors the side streams read during this step: held until the GPU is past here (hip_ops.hold)
        if MAX_STEPS_IN_FLIGHT <= 0:
            return
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self._inflight.append(ev)
        while len(self._inflight) > MAX_STEPS_IN_FLIGHT or (len(self._inflight) > 1 and ops.held_bytes() > MAX_HELD_BYTES):
            self._inflight.popleft().synchronize()
            ops.reap()
        ops.reap()   # the step that wait was for is complete: its held tensors go back to the allocator now

    @property
    def arena(self):
        """The parameter arena this optimizer updates (bound on first use); `arena.reducer` is the data-parallel reducer, if any."""
        return self._bind()

    def _bind(self):
        if self._arena is not None:
            return self._arena
        arena = None
        for g in self.param_groups:
            for p in g["params"]:
                a = getattr(p, "_ecamp_arena", None)
                if a is None:
                    raise RuntimeError("FusedAdamW: parameter is not in an ecamp_amd arena -- call model.prepare() (or run one "
                                       "forward) after model.to('cuda') and before the first optimizer.step()")
                arena = arena or a
                if a is not arena:
                    raise RuntimeError("FusedAdamW: parameters from different arenas")
        nblk = arena.total // 64
        table = torch.full((nblk,), 255, dtype=torch.uint8)
        for gi, g in enumerate(self.param_groups):
            for p in g["params"]:
                if getattr(p, "_ecamp_unused", False):
                    continue
                i = arena.index[id(p)]
                o, n = arena.offsets[i], arena.sizes[i]
                table[o // 64:(o + n + 63) // 64] = gi
        self._table = table.to(arena.device)
        self._m = ops.zeros((arena.total,), arena.device)
        self._v = ops.zeros((arena.total,), arena.device)
        self._arena = arena
        return arena

    @property
    def steps_taken(self):
        """Optimizer steps applied so far (a step skipped by the dynamic loss scaler does not count; reading it waits for the device)."""
        if self._step_dev is not None:
            self._step = int(self._step_dev.item())
        return self._step

    def step_counter(self):
        """The device-side step count (created from the host's on first use): ecamp_loss_scale_update advances it only on clean steps."""
        if self._step_dev is None:
            self._step_dev = torch.full((1,), float(self._step), device=self._bind().device, dtype=torch.float32)
        return self._step_dev

    def _launch(self, A, lo, hi, grad_sumsq, ctl=None):
        g0 = self.param_groups[0]
        ops.adamw_grouped(A.flat_p[lo:hi], A.flat_g[lo:hi], self._m[lo:hi], self._v[lo:hi], A.flat_p16[lo:hi] if A.flat_p16 is not None else None,
                          self._table[lo // 64:hi // 64], [g["lr"] for g in self.param_groups], [g["weight_decay"] for g in self.param_groups],
                          g0["betas"][0], g0["betas"][1], g0["eps"], max(self._step, 1), self.grad_scale, grad_sumsq, ctl)

    @torch.no_grad()
    def step(self, closure=None, grad_sumsq=None, ctl=None):
        """`ctl` (device f32[4], hip_ops.loss_scale_update): the gradient scale, the bias corrections and whether this step happens at all
        are read on the device; the step count then lives in `step_counter()`."""
        loss = closure() if closure is not None else None
        A = self._bind()
        events = None
        if A.reducer is not None:
            A.reducer.assert_reduced()   # loud failure instead of a silent step on un-reduced gradients
            events = A.reducer.take_bucket_events()
        if ctl is None:
            if self._step_dev is not None:      # back from device-decided steps to host-counted ones
                self._step, self._step_dev = self.steps_taken, None
            self._step += 1
        if events and not A._fresh:
            # data parallel, lazy join (GradReducer.lazy): one slice of the fused kernel per gradient bucket, each behind its own
            # all-reduce, in the order the collectives were issued -- the update of the early buckets runs while the last ones
            # are still on the links
            cur = torch.cuda.current_stream()
            for lo, hi, ev in events:
                cur.wait_event(ev)
                self._launch(A, lo, hi, grad_sumsq, ctl)
            self.bucketwise_steps += 1
        else:
            for _, _, ev in events or ():    # (a module did not run this window: zero its weights only behind every collective)
                torch.cuda.current_stream().wait_event(ev)
            A.flush_fresh()
            self._launch(A, 0, A.total, grad_sumsq, ctl)
        A.version += 1   # the bf16 shadows changed: the fp8-forward mode re-quantises its weight copies on next use
        self.pace()
        return loss

    def covers(self, parameters):
        """True when the tensors of `parameters` that carry a gradient are exactly the ones this optimizer updates: then
        `step_with_grad_norm()` equals the reference's get_grad_norm_(parameters) followed by step()."""
        own = {id(p) for g in self.param_groups for p in g["params"] if p.grad is not None}
        given = {id(p) for p in parameters if p.grad is not None}
        return len(own) > 0 and given == own

    @torch.no_grad()
    def step_with_grad_norm(self):
        """One pass over the gradient arena: the update AND sum(g^2) of what it consumed (util/misc.py:280-292 computes the norm in
        a separate pass over 733 MB just before the step).  -> 0-d tensor, the global L2 gradient norm before the update."""
        A = self._bind()
        s = ops.zeros((1,), A.device)
        self.step(grad_sumsq=s)
        return s.sqrt().reshape(())

    def zero_grad(self, set_to_none=False):
        """p.grad stay views of the gradient arena (set_to_none would detach them).  After the first step this is LAZY for weight
        matrices: biases / LayerNorm / embedding gradients are zeroed now, a weight matrix keeps its old values until the next
        backward pass OVERWRITES it (its first weight-gradient GEMM runs with beta = 0), or until `flush_grads()` / `step()` find
        it untouched and zero it.  Read p.grad after backward, not between zero_grad() and backward.  ECAMP_LAZY_ZERO_GRAD=0 restores
        the plain memset."""
        self._bind().zero_grad()

    def flush_grads(self):
        """Make every p.grad consistent (zero the weight gradients no GEMM has written since zero_grad())."""
        self._bind().flush_fresh()

    # -- checkpoint format compatible with torch.optim.AdamW (misc.py:295-338) ---------------------------------
    def state_dict(self):
        A = self._bind()
        self._step = self.steps_taken
        state, packed_groups, idx = {}, [], 0
        for g in self.param_groups:
            ids = []
            for p in g["params"]:
                i = A.index[id(p)]
                o, n = A.offsets[i], A.sizes[i]
                if self._step > 0 and not getattr(p, "_ecamp_unused", False):
                    state[idx] = {"step": torch.tensor(float(self._step)), "exp_avg": self._m[o:o + n].view(p.shape).clone(),
                                  "exp_avg_sq": self._v[o:o + n].view(p.shape).clone()}
                ids.append(idx)
                idx += 1
            packed_groups.append({**{k: v for k, v in g.items() if k != "params"}, "params": ids})
        return {"state": state, "param_groups": packed_groups}

    def load_state_dict(self, sd):
        A = self._bind()
        idx = 0
        for g, sg in zip(self.param_groups, sd["param_groups"]):
            for k, v in sg.items():
                if k != "params":
                    g[k] = v
            for p in g["params"]:
                st = sd["state"].get(idx, sd["state"].get(str(idx)))
                if st is not None:
                    i = A.index[id(p)]
                    o, n = A.offsets[i], A.sizes[i]
                    self._m[o:o + n].view(p.shape).copy_(st["exp_avg"])
                    self._v[o:o + n].view(p.shape).copy_(st["exp_avg_sq"])
                    self._step, self._step_dev = int(float(st["step"])), None
                idx += 1
