"""Flat parameter / gradient arenas in HBM (MI355X-first memory layout).

All trainable parameters of the model live in ONE contiguous f32 buffer, their gradients in a second one of
identical layout, and (bf16 mode) a bf16 shadow copy in a third.  `p.data` / `p.grad` of every nn.Parameter
are views into these arenas, so
  * the fused AdamW, the global grad-norm and `zero_grad` are one kernel launch each over the whole model,
  * data-parallel gradient buckets are zero-copy slices of the gradient arena (no flatten/unflatten),
  * weight-gradient GEMMs accumulate straight into their final location (no autograd AccumulateGrad pass).
Parameters are laid out in registration order -- the reverse of the order in which backward finishes them --
each padded to 64 elements (256 B) so every weight matrix is 16-B aligned for the GEMM loads.
"""
import torch

from . import hip_ops as ops

ALIGN = 64
LAZY_ZERO = __import__("os").environ.get("ECAMP_LAZY_ZERO_GRAD", "1") != "0"   # 0: zero_grad() memsets the whole gradient arena
# fp8 forward, delayed scaling of the activation sites: the scale covers F8_MARGIN x the largest maximum of the last F8_HISTORY steps
# (validated here, where they are read: ecamp_fp8_roll refuses a history outside 1..64 steps and a margin outside [1, 16] on every step,
# with a message that does not name the environment.  The one-batch calibration needs no seeding of the history: its maximum sits in the
# site's amax slots until the first roll, which enters it as the history's first element.)
F8_HISTORY = int(__import__("os").environ.get("ECAMP_FP8_HISTORY", "4"))
F8_MARGIN = float(__import__("os").environ.get("ECAMP_FP8_MARGIN", "1.25"))
if not 1 <= F8_HISTORY <= 64:
    raise ValueError("ECAMP_FP8_HISTORY=%d: the delayed-scaling history is 1..64 optimizer steps" % F8_HISTORY)
if not 1.0 <= F8_MARGIN <= 16.0:
    raise ValueError("ECAMP_FP8_MARGIN=%g: the delayed-scaling margin is a factor in [1, 16]" % F8_MARGIN)


class ParamArena:
    def __init__(self, model, compute_dtype):
        # registration order, except that members of a fuse group (e.g. BERT query/key/value weights, which run as one
        # [3H,H] GEMM) are placed back to back at the position of the group's first member
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        name_of = {id(p): n for n, p in named}
        group_of = {}
        self._fuse_groups = [list(g) for g in (model.arena_fuse_groups() if hasattr(model, "arena_fuse_groups") else [])]
        for grp in self._fuse_groups:
            for p in grp:
                group_of[id(p)] = grp
        params, seen = [], set()
        for name, p in named:
            for q in group_of.get(id(p), [p]):
                if id(q) not in seen:
                    seen.add(id(q))
                    params.append((name_of[id(q)], q))
        if not params:
            raise ValueError("model has no trainable parameters")
        dev = params[0][1].device
        if dev.type != "cuda":
            raise ops._lib.EcampHipError("ParamArena needs the model on an MI355X device (model.to('cuda')); no CPU fallback exists")
        self.device = dev
        self.compute_dtype = compute_dtype
        self.names, self.params, self.offsets, self.sizes = [], [], [], []
        off = 0
        for name, p in params:
            n = p.numel()
            self.names.append(name)
            self.params.append(p)
            self.offsets.append(off)
            self.sizes.append(n)
            off += (n + ALIGN - 1) // ALIGN * ALIGN
        self.total = off
        self.flat_p = ops.zeros((off,), dev)
        self.flat_g = ops.zeros((off,), dev)
        self.flat_p16 = torch.empty((off,), device=dev, dtype=compute_dtype) if compute_dtype in (torch.bfloat16, torch.float16) else None
        self.index = {}
        for i, (p, o, n) in enumerate(zip(self.params, self.offsets, self.sizes)):
            self.flat_p[o:o + n].view(p.shape).copy_(p.data)
            p.data = self.flat_p[o:o + n].view(p.shape)
            p.grad = self.flat_g[o:o + n].view(p.shape)
            p._ecamp_slot = i
            p._ecamp_arena = self
            self.index[id(p)] = i
        self.unused = [i for i, p in enumerate(self.params) if getattr(p, "_ecamp_unused", False)]
        self.on_ready = None  # callback(list of slot ids) set by the data-parallel reducer
        self.reducer = None   # the GradReducer itself (parallel.DistributedDataParallel): the loss scaler / optimizer check it
        # Lazy zero_grad: a weight matrix whose gradient is produced by ONE weight-gradient GEMM per backward pass is never zeroed --
        # the first GEMM after zero_grad() OVERWRITES it (no 733 MB memset, no read of the zeros); everything else (biases, LayerNorm,
        # embeddings, tokens: kernels add into them with atomics) is zeroed by one table-driven kernel.
        self._gemm_written = set()   # slots seen in gradw(): learnt from the backward passes actually run
        self._fresh = set()          # of those, the ones not yet written since the last zero_grad()
        self._zero_flags = None      # uint8 per 64-element block: 1 = zero me in zero_grad(); rebuilt when _gemm_written grows
        self.version = 0      # bumped whenever the values the kernels read change (optimizer step, sync_shadow)
        self._w8 = {}         # slot -> (version, e4m3 copy, scale): the fp8-forward mode's weights, re-quantised once per step
        # fp8 forward, delayed scaling (configs[4]): one GEMM-input SITE per weight slot -- scale (what this step's producers quantise
        # with), 16 amax slots (what they have seen), and on the host whether the site has been calibrated (its first use runs the
        # two-pass current scaling, which also seeds the scale).  Allocated on first use.
        self.f8_scale = None
        self.f8_amax = None
        self.f8_cal = set()
        self.f8_rolled = 0    # the `version` the scales were last rolled at
        self.sync_shadow()

    # -- views ---------------------------------------------------------------------------------
    def grad(self, p):
        i = self.index[id(p)]
        o, n = self.offsets[i], self.sizes[i]
        return self.flat_g[o:o + n].view(p.shape)

    def w(self, p):
        """The tensor the kernels read for parameter p: the bf16 shadow in bf16 mode, the f32 master otherwise."""
        if self.flat_p16 is None:
            return p.data
        i = self.index[id(p)]
        o, n = self.offsets[i], self.sizes[i]
        return self.flat_p16[o:o + n].view(p.shape)

    def gradw(self, ps, shape=None):
        """(gradient view, accumulate flag) for the weight-gradient GEMM of parameter `ps` (or of a list of adjacent parameters
        computed as ONE GEMM, viewed as `shape`): accumulate is False for the first GEMM after zero_grad()."""
        plist = list(ps) if isinstance(ps, (list, tuple)) else [ps]
        slots = [self.index[id(p)] for p in plist]
        view = self.fused_grad(plist, shape) if len(plist) > 1 else (self.grad(plist[0]) if shape is None else self.grad(plist[0]).view(shape))
        new = [i for i in slots if i not in self._gemm_written]
        if new:
            self._gemm_written.update(new)   # still zeroed by the last zero_grad(): accumulate this time, overwrite from now on
            self._zero_flags = None
        fresh = [i in self._fresh for i in slots]
        if all(fresh):
            self._fresh.difference_update(slots)
            return view, False
        if any(fresh):   # a fused group written partly: make the rest zero first (not on the hot path)
            self.flush_fresh(slots)
        return view, True

    def flush_fresh(self, slots=None):
        """Zero the gradients of overwrite-mode weights that NO GEMM has written since zero_grad() (a module that did not run this
        step): called before anything consumes the gradient arena as a whole (grad-norm, all-reduce finalisation, optimizer)."""
        todo = [i for i in (self._fresh if slots is None else slots) if i in self._fresh]
        for i in todo:
            o, n = self.offsets[i], self.sizes[i]
            ops.zero_(self.flat_g[o:o + n])
        self._fresh.difference_update(todo)

    def w8(self, p, shape=None):
        """(uint8 e4m3 copy, f32[1] scale) of parameter p -- or of a list of arena-adjacent parameters viewed as ONE [N, K] matrix of
        `shape` (BERT's fused query/key/value block) -- for the fp8 forward GEMMs (configs[4]); quantised from the bf16 shadow the first
        time it is asked for after the values changed (once per optimizer step)."""
        plist = list(p) if isinstance(p, (list, tuple)) else [p]
        i = self.index[id(plist[0])]
        if self._w8.get("version") != self.version:
            self._quantize_weights()
        o, n = self._span(plist) if len(plist) > 1 else (self.offsets[i], self.sizes[i])
        q = self.flat_p8[o:o + n]
        return (q.view(shape) if shape is not None else q.view(plist[0].shape)), self.w8_scale[self._w8["sid"][i]:self._w8["sid"][i] + 1]

    def _quantize_weights(self):
        """All matrices of the bf16 shadow -> e4m3 arena `flat_p8`, each with its own per-matrix scale (the members of a fuse group
        share one: they run as ONE GEMM), in three launches: maxima, roll, quantise (ecamp_fp8_weights)."""
        if "items" not in self._w8:
            sid = list(range(len(self.params)))
            for grp in self._fuse_groups:
                first = min(self.index[id(q)] for q in grp)
                for q in grp:
                    sid[self.index[id(q)]] = first
            rows = []
            for i, (p, o, n) in enumerate(zip(self.params, self.offsets, self.sizes)):
                if p.dim() < 2 or n % 4:
                    continue
                for c in range(0, n, 65536):
                    rows.append((o + c, min(65536, n - c), sid[i], 0))
            self._w8["sid"] = sid
            self._w8["items"] = torch.tensor(rows, dtype=torch.int32).to(self.device)
            self.flat_p8 = torch.zeros((self.total,), device=self.device, dtype=torch.uint8)
            self.w8_scale = torch.ones((len(self.params),), device=self.device, dtype=torch.float32)
            self.w8_amax = ops.zeros((len(self.params) * 512,), self.device)
        it = self._w8["items"]
        ops.fp8_weights(self.flat_p16, self.flat_p8, it, self.w8_amax, self.w8_scale, 0)
        ops.fp8_roll(self.w8_amax, self.w8_scale)
        ops.fp8_weights(self.flat_p16, self.flat_p8, it, self.w8_amax, self.w8_scale, 1)
        self._w8["version"] = self.version

    def f8_site(self, p):
        """-> (site index, scale f32[1] view, amax-slot f32[512] view, calibrated?) of the GEMM whose weight (first weight) is p.  Rolls
        every site's amax into its next scale when an optimizer step has happened since the last roll (delayed scaling: this step
        quantises with the maxima the previous step saw)."""
        if self.f8_scale is None:
            n = len(self.params)
            self.f8_scale = torch.ones((n,), device=self.device, dtype=torch.float32)
            self.f8_amax = ops.zeros((n * 512,), self.device)
            # delayed scaling with a memory: a site quantises with F8_MARGIN x the largest maximum its producers saw over the
            # last F8_HISTORY optimizer steps -- a history of ONE step without a margin let any activation that grows from step to step
            # (the first steps after the one-batch calibration, warm-up) saturate silently at +-448
            self.f8_hist = ops.zeros((n, F8_HISTORY), self.device)
            self.f8_rolls = 0
            self.f8_rolled = self.version
        if self.f8_rolled != self.version:
            ops.fp8_roll(self.f8_amax, self.f8_scale, hist=self.f8_hist, hist_pos=self.f8_rolls, margin=F8_MARGIN)
            self.f8_rolls += 1
            self.f8_rolled = self.version
        i = self.index[id(p[0] if isinstance(p, (list, tuple)) else p)]
        return i, self.f8_scale[i:i + 1], self.f8_amax[i * 512:(i + 1) * 512], i in self.f8_cal

    def _span(self, ps):
        idx = [self.index[id(p)] for p in ps]
        for a, b, p in zip(idx[:-1], idx[1:], ps[:-1]):
            if b != a + 1 or self.sizes[a] % ALIGN != 0:
                raise RuntimeError("parameters are not adjacent in the arena; cannot fuse")
        return self.offsets[idx[0]], sum(self.sizes[i] for i in idx)

    def fused_w(self, ps, shape):
        o, n = self._span(ps)
        src = self.flat_p16 if self.flat_p16 is not None else self.flat_p
        return src[o:o + n].view(shape)

    def fused_f32(self, ps, shape):
        o, n = self._span(ps)
        return self.flat_p[o:o + n].view(shape)

    def fused_grad(self, ps, shape):
        o, n = self._span(ps)
        return self.flat_g[o:o + n].view(shape)

    # -- maintenance ---------------------------------------------------------------------------
    def sync_shadow(self):
        """Refresh the bf16 shadow from the f32 masters (after load_state_dict / manual edits; AdamW does it itself)."""
        if self.flat_p16 is not None:
            ops.cast(self.flat_p, self.flat_p16)
        self.version += 1

    def zero_grad(self):
        if not self._gemm_written or not LAZY_ZERO:
            ops.zero_(self.flat_g)
            self._fresh.clear()
            return
        if self._zero_flags is None:
            flags = torch.ones(self.total // ALIGN, dtype=torch.uint8)
            for i in self._gemm_written:
                o, n = self.offsets[i], self.sizes[i]
                flags[o // ALIGN:(o + n + ALIGN - 1) // ALIGN] = 0
            for i in self.unused:        # never written by anything: must read as zero
                o, n = self.offsets[i], self.sizes[i]
                flags[o // ALIGN:(o + n + ALIGN - 1) // ALIGN] = 1
            self._zero_flags = flags.to(self.device)
        ops.zero_blocks_(self.flat_g, self._zero_flags)
        self._fresh = set(self._gemm_written) - set(self.unused)

    def attach_grads(self):
        """Re-point p.grad at the arena if someone set it to None (stock optimizers' zero_grad(set_to_none=True))."""
        lost = False
        for p, o, n in zip(self.params, self.offsets, self.sizes):
            if p.grad is None or p.grad.data_ptr() != self.flat_g.data_ptr() + 4 * o:
                p.grad = self.flat_g[o:o + n].view(p.shape)
                lost = True
        if lost:
            self.zero_grad()

    def ready(self, *ps):
        if self.on_ready is not None:
            self.on_ready([self.index[id(p)] for p in ps])
