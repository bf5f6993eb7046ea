"""Data-parallel gradient exchange for one-process-per-GPU training over RCCL/xGMI.

The reference wraps the model in torch DDP with find_unused_parameters=True and no `no_sync()`
(main_pretrain.py:247-250): a full 733 MB all-reduce on EVERY micro-step plus a graph walk for the two unused
pooler tensors.  Here the gradient arena (ecamp_amd/arena.py) is cut into contiguous buckets in reverse
registration order == the order backward finishes them; the hand-written backward stages report finished
parameters (`arena.ready`), and a bucket whose parameters are all done is all-reduced IN PLACE (zero-copy slice
of the arena, op=AVG) on a side HIP stream while the remaining backward keeps the compute stream busy.  Unused
parameters are known statically and never waited for; accumulation micro-steps skip communication.
xGMI is point-to-point (7 links x ~153 GB/s): large buckets (default 32 MiB, 8 MiB for the parameters backward finishes last)
keep RCCL's rings/trees per-link efficient; priced with a serial communication stream at 70 % of (N-1) x 76 GB/s per GPU the whole
f32 exchange is 4.8 ms (8 GPUs) / 8.3 ms (4) / 15.2 ms (2 GPUs, one link) of link time against 22 ms of backward at B=256, of which
the first gradients are final after 3.3 ms (tools/bucket_timeline.py).  `grad_dtype=torch.bfloat16` halves the
payload (a bucket is cast-packed into a bf16 mirror on the communication stream, all-reduced there and unpacked into the f32 arena);
f32 stays the default.  RCCL's kernels share the CUs with the backward pass: `rccl_env_defaults()` caps their channel (= workgroup)
count below the point where a co-tenant starts to cost the step more than it hides (profiles/r03_cotenant_ab.txt).
"""
import contextlib
import os
import re

import torch
import torch.distributed as dist
import torch.nn as nn

# RCCL runs one workgroup per channel beside the backward pass.  A spinning co-tenant of 16 / 32 / 64 / 96 workgroups costs the step
# +1.6 / +1.7 / +1.9 / +2.9 ms and 128 workgroups +13 ms (a cliff: no CU is left without a resident stranger;
# profiles/r03_cotenant_ab.txt, two boxes), while the exchange has 22 ms of backward to hide 4.8-15.2 ms of link time in -- so the
# channel count is capped at 32, a quarter of the cliff and the reserve the weight-gradient launches leave free
# (`p8_wgrad_reserve_cus` = 32).  Applied with setdefault: an explicit NCCL_* setting of the user wins, and the `rccl` record of
# bench.py prints what was in force.
RCCL_ENV_DEFAULTS = {"NCCL_MAX_NCHANNELS": "32", "HSA_ENABLE_IPC_MODE_LEGACY": "0"}

# Switches that only matter at N > 1.  The two that change WHAT RUNS beside RCCL -- the optimizer bucket by bucket behind each bucket's
# all-reduce, and the data-gradient GEMMs as one output tile per workgroup -- were tuned against a spinning stand-in kernel
# (tools/hog_probe.py), never against RCCL with more than one rank (this pool has one GPU per box), so they are OFF until
# tests/test_rccl_gpu.py has run on a box with two GPUs (it arms itself there and checks both against the plain path, bit for bit):
# ECAMP_BUCKETWISE_ADAMW=1 / ECAMP_DDP_Q8_BWD_GRID=1048576 switch them on.  The weight-gradient launches' 32-CU reserve is only a
# smaller grid of the same kernel and stays.
def ddp_defaults(env=None):
    env = os.environ if env is None else env
    return {"bucketwise_adamw": env.get("ECAMP_BUCKETWISE_ADAMW", "0") != "0",
            "q8_bwd_grid": int(env.get("ECAMP_DDP_Q8_BWD_GRID", "0")),
            "p8_wgrad_reserve_cus": int(env.get("ECAMP_P8_RESERVE_CUS", "32"))}


def rccl_env_defaults(env=None):
    """setdefault() RCCL_ENV_DEFAULTS into `env` (default: this process's environment; call before init_process_group)."""
    env = os.environ if env is None else env
    for k, v in RCCL_ENV_DEFAULTS.items():
        env.setdefault(k, v)
    return env


def rccl_env_record(debug_file=None):
    """What the `rccl` record of bench.py says about the communication kernels: every NCCL_* / RCCL_* variable in force and, when
    RCCL wrote an INFO log (NCCL_DEBUG=INFO, NCCL_DEBUG_FILE), the channel count it reports."""
    rec = {"env": {k: v for k, v in sorted(os.environ.items()) if k.startswith(("NCCL_", "RCCL_"))}}
    rec["max_channels_cap"] = int(os.environ["NCCL_MAX_NCHANNELS"]) if os.environ.get("NCCL_MAX_NCHANNELS", "").isdigit() else None
    if debug_file and os.path.exists(debug_file):
        try:
            txt = open(debug_file, errors="replace").read()
        except OSError:
            txt = ""
        ch = [int(m) for m in re.findall(r"(\d+) coll channels", txt)]
        ch += [int(b) for _, b in re.findall(r"Channel (\d+)/(\d+)", txt)]
        rec["channels_reported"] = max(ch) if ch else None
        rec["channel_lines"] = [ln.strip()[-160:] for ln in txt.splitlines() if "coll channels" in ln or "nChannels" in ln][:4]
    return rec


class GradReducer:
    def __init__(self, flat_g, offsets, sizes, unused=(), bucket_mb=32.0, group=None, force_comm=False, tail_bucket_mb=None,
                 tail_span_mb=32.0, grad_dtype=None):
        self.flat_g, self.offsets, self.sizes = flat_g, list(offsets), list(sizes)
        # payload dtype of the exchange: None / float32 = the arena itself, in place (the default, bit-identical to round 3);
        # bfloat16 = a bucket is cast into its slice of a bf16 mirror of the arena on the communication stream, the mirror slice is
        # all-reduced and cast back into the f32 arena -- half the bytes on the links (366 instead of 733 MB per step)
        self.grad_dtype = None if grad_dtype in (None, torch.float32) else grad_dtype
        if self.grad_dtype not in (None, torch.bfloat16):
            raise ValueError("GradReducer: grad_dtype must be float32 or bfloat16")
        self.pack = None   # the bf16 mirror, allocated on first use
        self.unused = set(unused)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.enabled = True
        self.force_comm = force_comm  # run the collectives even in a 1-rank group (exercises the RCCL / stream plumbing in tests)
        # Bucket sizes follow the backward timeline (tools/bucket_timeline.py): a bucket is closed BEFORE a tensor that would
        # take it past the cap (the 92 MB vocabulary-head weight, final 2.6 ms into backward, must not wait for the small fusion
        # parameters registered after it, final at 11 ms), and the parameters registered FIRST -- the ones backward finishes
        # last, whose exchange nothing can hide -- go in small buckets (`tail_bucket_mb` over the first `tail_span_mb` of the
        # arena), so that the exposed tail of the exchange is one small message.
        esz = flat_g.element_size()
        cap = max(1, int(bucket_mb * 1024 * 1024) // esz)
        tail_cap = cap if tail_bucket_mb is None else max(1, min(cap, int(tail_bucket_mb * 1024 * 1024) // esz))
        tail_span = 0 if tail_bucket_mb is None else int(tail_span_mb * 1024 * 1024) // esz
        n = len(self.offsets)
        self.buckets = []  # (lo, hi, [slots]) in the order backward completes them (last registered first)
        hi, slots = flat_g.numel(), []
        for i in range(n - 1, -1, -1):
            c = tail_cap if self.offsets[i] < tail_span else cap
            if slots and hi - self.offsets[i] > c:
                self.buckets.append((self.offsets[i + 1], hi, slots))
                hi, slots = self.offsets[i + 1], []
            slots.append(i)
            if hi - self.offsets[i] >= c or i == 0:
                self.buckets.append((self.offsets[i], hi, slots))
                hi, slots = self.offsets[i], []
        self.slot2bucket = {s: b for b, (_, _, sl) in enumerate(self.buckets) for s in sl}
        self.side = torch.cuda.Stream(device=flat_g.device) if flat_g.is_cuda else None
        backend = dist.get_backend(group) if dist.is_initialized() else None
        self.use_avg = backend == "nccl"  # RCCL has ncclAvg; gloo does not
        # gloo on device tensors (two ranks sharing ONE GPU in the model-level equivalence test; RCCL refuses duplicate devices):
        # buckets are staged through host memory, synchronously -- a test / debugging path, never the production one
        self.host_staged = backend == "gloo" and flat_g.is_cuda
        self._cb_queued = False
        self.main_stream = None   # the step's compute stream (set by the wrapper's forward); the collectives also wait for it
        self.timing = False       # record event pairs around every collective (bench.py's `rccl` record)
        self.after_bucket = None  # callback(lo, hi, slots), run right behind a bucket's collective on the stream that carries it (with a
        #                           callback set even a 1-rank reducer walks its buckets: tests copy them there to prove that a bucket
        #                           is final when its collective may start)
        self._timed = []
        # lazy join (set for one backward pass by the loss scaler when the fused optimizer follows): finalize() leaves the buckets'
        # completion events in `bucket_events` instead of making the compute stream wait for all of them, and FusedAdamW.step()
        # updates bucket by bucket, each slice behind its own all-reduce -- the optimizer pass hides the tail of the exchange that
        # backward could not (2.0 / 0.5 / 0.35 ms at 2 / 4 / 8 GPUs, tools/bucket_timeline.py).  Everything stays on the compute stream.
        self.lazy = False
        self.bucket_events = None
        self.dirty = False      # a backward pass has reported gradients that finalize() has not yet reduced
        self.reset()

    def reset(self):
        self.pending = [sum(1 for s in sl if s not in self.unused) for (_, _, sl) in self.buckets]
        self.launched = [False] * len(self.buckets)
        self.works = []
        self._cb_queued = False

    def _launch(self, b):
        comm = not (self.world == 1 and not self.force_comm)
        if self.launched[b] or (not comm and self.after_bucket is None):
            self.launched[b] = True
            return
        self.launched[b] = True
        lo, hi, slots = self.buckets[b]
        buf = self.flat_g[lo:hi]
        op = dist.ReduceOp.AVG if self.use_avg else dist.ReduceOp.SUM
        if self.host_staged:
            from . import hip_ops
            cur = torch.cuda.current_stream()
            if self.main_stream is not None and self.main_stream != cur:
                cur.wait_stream(self.main_stream)
            for wst in hip_ops.side_streams(self.flat_g.device):
                cur.wait_stream(wst)
            host = buf.cpu() if self.grad_dtype is None else buf.to(self.grad_dtype).cpu()
            dist.all_reduce(host, op=op, group=self.group)
            if not self.use_avg and self.world > 1:
                host = host.float().div_(self.world)
            buf.copy_(host)
            if self.after_bucket is not None:
                self.after_bucket(lo, hi, slots)
            self.works.append((None, b))
            return
        if self.side is not None:
            # the collective may start once every stream that can hold gradient work of this bucket has reached this point: the stream
            # that reported the parameter (a backward node may run on the branch stream), the MAIN compute stream of the step (the
            # report side's gradients of the same bucket are written there even when the reporting node ran elsewhere), the weight-
            # gradient side stream and the branch stream
            cur = torch.cuda.current_stream()
            self.side.wait_stream(cur)
            if self.main_stream is not None and self.main_stream != cur:
                self.side.wait_stream(self.main_stream)
            from . import hip_ops
            for wst in hip_ops.side_streams(self.flat_g.device):
                self.side.wait_stream(wst)
            with torch.cuda.stream(self.side):
                t0 = None
                if self.timing and comm:
                    t0 = torch.cuda.Event(enable_timing=True)
                    t0.record(self.side)
                if comm:
                    wire = buf
                    if self.grad_dtype is not None:   # cast-pack on the communication stream
                        wire = self._mirror()[lo:hi]
                        wire.copy_(buf)
                    w = dist.all_reduce(wire, op=op, group=self.group, async_op=True)
                    w.wait()       # stream-level: the SIDE stream waits for RCCL's stream (the host does not block)
                    if wire is not buf:
                        buf.copy_(wire)
                    if not self.use_avg and self.world > 1:
                        buf.div_(self.world)
                if t0 is not None:
                    t1 = torch.cuda.Event(enable_timing=True)
                    t1.record(self.side)
                    self._timed.append((t0, t1))
                if self.after_bucket is not None:
                    self.after_bucket(lo, hi, slots)
                done = torch.cuda.Event()
                done.record(self.side)
            self.works.append((done, b))
        else:
            wire = buf
            if self.grad_dtype is not None:
                wire = self._mirror()[lo:hi]
                wire.copy_(buf)
            w = dist.all_reduce(wire, op=op, group=self.group, async_op=True)
            self.works.append((w, b))

    def _mirror(self):
        if self.pack is None:
            self.pack = torch.empty(self.flat_g.numel(), dtype=self.grad_dtype, device=self.flat_g.device)
        return self.pack

    def payload_bytes(self):
        """Bytes one optimizer step puts on the links per rank (before the collective's own 2(N-1)/N factor)."""
        return self.flat_g.numel() * (self.flat_g.element_size() if self.grad_dtype is None else 2)

    def comm_ms(self):
        """Sum of the all-reduce durations (ms, events on the communication stream) recorded since the last call; needs `timing`."""
        tot = 0.0
        for a, b in self._timed:
            b.synchronize()
            tot += a.elapsed_time(b)
        self._timed = []
        return tot

    def mark_ready(self, slots):
        """Called from inside the backward stages as soon as a parameter's gradient is final."""
        if not self.enabled:
            return
        self.dirty = True
        if not self._cb_queued:
            self._cb_queued = True
            try:
                torch.autograd.Variable._execution_engine.queue_callback(self.finalize)
            except RuntimeError:
                # not inside a backward pass (unit tests drive finalize() themselves).  `dirty` stays set: the loss scaler calls
                # finalize() before the gradients are read, and FusedAdamW refuses to step on an un-reduced arena (assert_reduced)
                self._cb_queued = False
        for s in slots:
            if s in self.unused:
                continue
            b = self.slot2bucket[s]
            self.pending[b] -= 1
            if self.pending[b] == 0:
                self._launch(b)

    def finalize(self):
        """Runs at the end of backward (autograd engine callback) and again, as a no-op, from the loss scaler: every bucket is
        reduced and visible to the compute stream afterwards."""
        if not self.enabled or not self.dirty:
            return
        for b in range(len(self.buckets)):
            if not self.launched[b]:
                self._launch(b)
        self.join()   # events an earlier lazy pass left behind (nobody consumed them)
        if self.lazy and self.works and all(isinstance(w, torch.cuda.Event) for w, _ in self.works):
            self.bucket_events = [(self.buckets[b][0], self.buckets[b][1], w) for w, b in self.works]   # in launch order
        else:
            for w, b in self.works:
                if isinstance(w, torch.cuda.Event):
                    torch.cuda.current_stream().wait_event(w)   # the compute stream sees the reduced bucket
                elif w is not None:
                    w.wait()
                    lo, hi, _ = self.buckets[b]
                    if self.grad_dtype is not None:              # unpack the reduced bf16 slice into the f32 arena
                        self.flat_g[lo:hi].copy_(self.pack[lo:hi])
                    if not self.use_avg and self.world > 1:      # host tensors over gloo (the device paths divide in _launch)
                        self.flat_g[lo:hi].div_(self.world)
        self.dirty = False
        self.reset()

    def take_bucket_events(self):
        """-> [(lo, hi, event)] of the last lazy finalize() (the caller now owes the waits), or None."""
        ev, self.bucket_events = self.bucket_events, None
        return ev

    def join(self):
        """The current stream waits for whatever a lazy finalize() left in flight."""
        for _, _, e in self.take_bucket_events() or ():
            torch.cuda.current_stream().wait_event(e)

    def assert_reduced(self):
        """Called by the optimizer before it reads the gradient arena."""
        if self.enabled and self.dirty:
            raise RuntimeError("GradReducer: the optimizer is about to step on gradients that were reported by backward but never "
                               "all-reduced (finalize() did not run: no autograd callback and no loss-scaler call)")


class DistributedDataParallel(nn.Module):
    """Minimal DDP surface used by main_pretrain.py (`.module`, forward passthrough, `no_sync`)."""

    def __init__(self, module, device_ids=None, find_unused_parameters=False, bucket_cap_mb=32.0, process_group=None, force_comm=False,
                 tail_bucket_mb=8.0, tail_span_mb=32.0, grad_dtype=None, **_ignored):
        super().__init__()
        self.module = module
        arena = module.prepare()
        if dist.is_initialized() and dist.get_world_size(process_group) > 1:
            if dist.get_backend(process_group) == "gloo" and arena.flat_p.is_cuda:   # host-staged (see GradReducer.host_staged)
                host = arena.flat_p.cpu()
                dist.broadcast(host, src=0, group=process_group)
                arena.flat_p.copy_(host)
            else:
                dist.broadcast(arena.flat_p, src=0, group=process_group)  # C1: parameters rank0 -> all (one 733 MB message)
            arena.sync_shadow()
            # RCCL's all-reduce workgroups share the CUs with the backward pass, and a persistent one-workgroup-per-CU GEMM
            # whose CU is taken starts that workgroup late and holds its static share of the tiles back (tools/hog_probe.py).
            # The weight-gradient launches of the side stream leave 32 CUs to the communication kernels; opt-in (ddp_defaults):
            # the data-gradient GEMMs (the critical chain) as one output tile per workgroup -- the dispatcher deals the tiles to
            # whatever CUs are free, however many the collective takes (+0.15 ms per step on a GPU of its own, tools/grid_ab.sh)
            from . import hip_ops
            dd = ddp_defaults()
            hip_ops.set_option("p8_wgrad_reserve_cus", dd["p8_wgrad_reserve_cus"])
            hip_ops.set_option("q8_bwd_grid", dd["q8_bwd_grid"])
        self.reducer = GradReducer(arena.flat_g, arena.offsets, arena.sizes, arena.unused, bucket_cap_mb, process_group, force_comm,
                                   tail_bucket_mb=tail_bucket_mb, tail_span_mb=tail_span_mb, grad_dtype=grad_dtype)
        arena.on_ready = self.reducer.mark_ready
        arena.reducer = self.reducer   # the loss scaler and the optimizer find it here

    def forward(self, *args, **kwargs):
        if self.module.arena is not None and self.module.arena.on_ready is None:
            self.module.arena.on_ready = self.reducer.mark_ready
        if self.reducer.flat_g.is_cuda:
            self.reducer.main_stream = torch.cuda.current_stream()
        return self.module(*args, **kwargs)

    def set_grad_sync(self, flag):
        self.reducer.enabled = bool(flag)

    @contextlib.contextmanager
    def no_sync(self):
        old = self.reducer.enabled
        self.reducer.enabled = False
        try:
            yield
        finally:
            self.reducer.enabled = old
