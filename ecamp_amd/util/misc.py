"""Runtime helpers with the API of ECAMP/Pre-training/util/misc.py (same names, argument meaning and return
values), rebuilt for the MI355X path:
  * init_distributed_mode  -> one process per GPU, backend "nccl" (= RCCL over xGMI on ROCm)      misc.py:216-248
  * NativeScalerWithGradNormCount -> bf16/f32 need no loss scaling; the grad-norm is ONE fused reduction over the
    gradient arena and the optimizer step ONE fused AdamW launch                                  misc.py:251-277
  * get_grad_norm_ / all_reduce_mean / MetricLogger / SmoothedValue / save_model / load_model      misc.py:24-213,280-347
Device syncs are lazy: meters accept 0-d device tensors and only read them back when a value is printed or
averaged (the reference forces 4 host syncs per micro-step, main_pretrain.py:143-145,155).
"""
import builtins
import datetime
import math
import os
import time
from collections import deque
from pathlib import Path

import torch
import torch.distributed as dist

inf = float("inf")


def _torch_stat(values, kind):
    """The window statistics are DEFINED by what torch computes on the window (the reference prints them to four decimals and
    tests/golden/meters.npz pins them to 1e-12): `median` = torch's lower median of the f32-converted values, `avg` = their f32 mean."""
    if kind == "median":
        return torch.tensor(values).median().item()
    return torch.tensor(values, dtype=torch.float32).mean().item()


class SmoothedValue(object):
    """A scalar series with a sliding window (median / avg / max / value) and running totals (global_avg); interface, format keys and
    printed values of misc.py:24-83.  `update` also takes 0-d device tensors: they are parked and read back together, once, the first
    time any statistic is asked for -- a training step never waits for the device because of a meter."""

    def __init__(self, window_size=20, fmt=None):
        self.fmt = "{median:.4f} ({global_avg:.4f})" if fmt is None else fmt
        self.deque = deque(maxlen=window_size)   # the window (the reference's attribute name)
        self.count, self.total = 0, 0.0
        self._parked = []                        # (device tensor, weight) not yet read back

    def _take(self, number, weight):
        self.deque.append(number)
        self.total += number * weight
        self.count += weight

    def update(self, value, n=1):
        if torch.is_tensor(value):
            self._parked.append((value.detach(), n))
        else:
            self._take(value, n)

    def _flush(self):
        if not self._parked:
            return
        parked, self._parked = self._parked, []
        host = torch.stack([t.float().reshape(()) for t, _ in parked]).tolist()   # ONE read-back for the lot
        for number, (_, weight) in zip(host, parked):
            self._take(number, weight)

    def synchronize_between_processes(self):
        """Sums count and total over the ranks; the window stays local (as in the reference)."""
        self._flush()
        if get_world_size() == 1:
            return
        pair = torch.tensor([self.count, self.total], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.barrier()
        dist.all_reduce(pair)
        self.count, self.total = int(pair[0].item()), pair[1].item()

    def _window(self):
        self._flush()
        return list(self.deque)

    median = property(lambda self: _torch_stat(self._window(), "median"))
    avg = property(lambda self: _torch_stat(self._window(), "avg"))
    max = property(lambda self: max(self._window()))
    value = property(lambda self: self._window()[-1])

    @property
    def global_avg(self):
        self._flush()
        return self.total / self.count

    def __str__(self):
        stats = {k: getattr(self, k) for k in ("median", "avg", "global_avg", "max", "value")}
        return self.fmt.format(**stats)


class MetricLogger(object):
    """Named SmoothedValue meters, created on first update; `logger.<name>` reaches a meter; `log_every` wraps the loader and prints the
    reference's progress line (misc.py:86-167)."""

    def __init__(self, delimiter="\t"):
        self.delimiter = delimiter
        self.meters = {}

    def add_meter(self, name, meter):
        self.meters[name] = meter

    def update(self, **kwargs):
        for name, value in kwargs.items():
            if value is None:
                continue
            if not isinstance(value, (float, int, torch.Tensor)):
                raise AssertionError("meter %r takes a number or a 0-d tensor, got %s" % (name, type(value).__name__))
            if name not in self.meters:
                self.meters[name] = SmoothedValue()
            self.meters[name].update(value)

    def __getattr__(self, attr):
        meters = self.__dict__.get("meters", {})
        if attr in meters:
            return meters[attr]
        raise AttributeError("'{}' object has no attribute '{}'".format(type(self).__name__, attr))

    def __str__(self):
        return self.delimiter.join("%s: %s" % (name, meter) for name, meter in self.meters.items())

    def synchronize_between_processes(self):
        for meter in self.meters.values():
            meter.synchronize_between_processes()

    def log_every(self, iterable, print_freq, header=None):
        header = header or ""
        n = len(iterable)
        start_time = end = time.time()
        iter_time, data_time = SmoothedValue(fmt="{avg:.4f}"), SmoothedValue(fmt="{avg:.4f}")
        width = str(len(str(n)))
        parts = [header, "[{0:" + width + "d}/{1}]", "eta: {eta}", "{meters}", "time: {time}", "data: {data}"]
        if torch.cuda.is_available():
            parts.append("max mem: {memory:.0f}")
        msg = self.delimiter.join(parts)
        for i, obj in enumerate(iterable):
            data_time.update(time.time() - end)
            yield obj
            iter_time.update(time.time() - end)
            if i % print_freq == 0 or i == n - 1:
                eta = str(datetime.timedelta(seconds=int(iter_time.global_avg * (n - i))))
                kw = dict(eta=eta, meters=str(self), time=str(iter_time), data=str(data_time))
                if torch.cuda.is_available():
                    kw["memory"] = torch.cuda.max_memory_allocated() / (1024.0 * 1024.0)
                print(msg.format(i, n, **kw))
            end = time.time()
        total = time.time() - start_time
        print("{} Total time: {} ({:.4f} s / it)".format(header, str(datetime.timedelta(seconds=int(total))), total / max(n, 1)))


# --------------------------------------------------------------------------------------------- distributed
def _in_process_group():
    return dist.is_available() and dist.is_initialized()


is_dist_avail_and_initialized = _in_process_group   # the reference's name (misc.py:187-192)


def get_world_size():
    return dist.get_world_size() if _in_process_group() else 1


def get_rank():
    return dist.get_rank() if _in_process_group() else 0


def is_main_process():
    return get_rank() == 0


def save_on_master(*args, **kwargs):
    if get_rank() == 0:
        torch.save(*args, **kwargs)


def setup_for_distributed(is_master):
    """After this call print() is silent on every rank but the master (pass force=True to print anyway; jobs of more than 8 ranks
    always print) and prefixes the wall-clock time -- the log format of misc.py:170-184."""
    plain_print = builtins.print

    def rank_aware_print(*args, force=False, **kwargs):
        if is_master or force or get_world_size() > 8:
            plain_print("[%s] " % datetime.datetime.now().time(), end="")
            plain_print(*args, **kwargs)

    builtins.print = rank_aware_print


def init_distributed_mode(args):
    """Same environment contract as misc.py:216-248 (OMPI / torchrun RANK+WORLD_SIZE+LOCAL_RANK / SLURM)."""
    if getattr(args, "dist_on_itp", False):
        args.rank = int(os.environ["OMPI_COMM_WORLD_RANK"])
        args.world_size = int(os.environ["OMPI_COMM_WORLD_SIZE"])
        args.gpu = int(os.environ["OMPI_COMM_WORLD_LOCAL_RANK"])
        args.dist_url = "tcp://%s:%s" % (os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"])
        os.environ["LOCAL_RANK"], os.environ["RANK"], os.environ["WORLD_SIZE"] = str(args.gpu), str(args.rank), str(args.world_size)
    elif "RANK" in os.environ and "WORLD_SIZE" in os.environ:
        args.rank = int(os.environ["RANK"])
        args.world_size = int(os.environ["WORLD_SIZE"])
        args.gpu = int(os.environ.get("LOCAL_RANK", 0))
    elif "SLURM_PROCID" in os.environ:
        args.rank = int(os.environ["SLURM_PROCID"])
        args.gpu = args.rank % max(torch.cuda.device_count(), 1)
    else:
        print("Not using distributed mode")
        setup_for_distributed(is_master=True)
        args.distributed = False
        return
    args.distributed = True
    use_gpu = torch.cuda.is_available()
    if use_gpu:
        torch.cuda.set_device(args.gpu)
    args.dist_backend = "nccl" if use_gpu else "gloo"  # "nccl" IS RCCL on ROCm
    print("| distributed init (rank {}): {}, gpu {}".format(args.rank, args.dist_url, args.gpu), flush=True)
    kw = {}
    if use_gpu:
        kw["device_id"] = torch.device("cuda", args.gpu)
        from ..parallel import rccl_env_defaults
        rccl_env_defaults()   # RCCL's channel (= workgroup) cap beside the backward pass; an explicit NCCL_* setting wins
    dist.init_process_group(backend=args.dist_backend, init_method=args.dist_url, world_size=args.world_size, rank=args.rank, **kw)
    dist.barrier()
    setup_for_distributed(args.rank == 0)


def all_reduce_mean(x):
    """misc.py:341-347.  Accepts a float or a 0-d/1-d tensor; tensors stay on the device (no host sync)."""
    world_size = get_world_size()
    if world_size == 1:
        return x
    if isinstance(x, torch.Tensor):
        y = x.detach().clone()
        dist.all_reduce(y)
        return y / world_size
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor(x, device=dev)
    dist.all_reduce(t)
    return (t / world_size).item()


# --------------------------------------------------------------------------------------------- scaler / grad norm
def get_grad_norm_(parameters, norm_type=2.0):
    """Global L2 norm = norm of per-tensor norms (misc.py:280-292).  When the parameters live in an ecamp_amd arena
    it is one fused sum-of-squares kernel over the flat gradient buffer; otherwise the reference formula."""
    if isinstance(parameters, torch.Tensor):
        parameters = [parameters]
    parameters = [p for p in parameters if p.grad is not None]
    if len(parameters) == 0:
        return torch.tensor(0.0)
    if float(norm_type) == 2.0:
        arena = getattr(parameters[0], "_ecamp_arena", None)
        if arena is not None and all(getattr(p, "_ecamp_arena", None) is arena for p in parameters) and len(parameters) == len(arena.params):
            from .. import hip_ops as ops
            arena.flush_fresh()
            s = ops.zeros((1,), arena.device)
            ops.sumsq(arena.flat_g, s)
            return s.sqrt().reshape(())
    device = parameters[0].grad.device
    if norm_type == inf:
        return max(p.grad.detach().abs().max().to(device) for p in parameters)
    return torch.norm(torch.stack([torch.norm(p.grad.detach(), norm_type).to(device) for p in parameters]), norm_type)


# data parallel: AdamW bucket by bucket behind each bucket's all-reduce -- opt-in until it has run beside RCCL with two ranks
# (ecamp_amd/parallel.py ddp_defaults; tests/test_rccl_gpu.py)
_BUCKETWISE_ADAMW = os.environ.get("ECAMP_BUCKETWISE_ADAMW", "0") != "0"
_FUSED_GRAD_NORM = os.environ.get("ECAMP_FUSED_GRAD_NORM", "1") != "0"   # 0: separate sum-of-squares pass, as the reference does


class NativeScalerWithGradNormCount:
    """Call signature and return value of misc.py:251-277.

    Two modes.  Default (`dynamic=False`): bf16 / f32 training needs no loss scaling, the scale is identically 1 (the state dict keeps
    GradScaler's keys so a reference checkpoint round-trips).  `dynamic=True` (`main_pretrain.py --loss_scale dynamic`,
    env ECAMP_LOSS_SCALE=dynamic) is the reference's `torch.cuda.amp.GradScaler()` restated (torch 1.13.1 grad_scaler.py; defaults
    init_scale 65536, growth 2, backoff 0.5, growth_interval 2000): the loss is multiplied by the scale before backward; at an update step
    the gradients are checked for inf / nan and un-scaled, the returned norm is that of the un-scaled gradients (inf / nan when the step
    overflowed, as the reference's get_grad_norm_ after unscale_), the optimizer step is SKIPPED on overflow, then the scale backs off
    (x 0.5, growth tracker to 0) or, after `growth_interval` clean steps in a row, grows (x 2).  A reference checkpoint's scaler state is
    loaded and continued.  On the arena path the check is one sum-of-squares pass over the flat gradient buffer (a non-finite sum <=> a
    non-finite element, up to |g| > 1.8e19), the un-scaling is folded into AdamW's read of the gradient (`grad_scale`) and into the norm:
    p.grad keeps the SCALED values until zero_grad().  The decision itself (skip / backoff / growth) is taken ON THE DEVICE
    (ecamp_loss_scale_update -> AdamW's `ctl`): GradScaler.step blocks the host on `found_inf.item()` once per optimizer step, this class
    never does -- scale, growth tracker, skipped-step count and the optimizer's step count are device values that get_scale(),
    skipped_steps, last_found_inf, state_dict() read back only when asked.  With `compute_dtype=torch.float16` (`--amp fp16`:
    IEEE-half activations, libecamp_hip_f16.so) this is the reference's autocast + GradScaler pair (main_pretrain.py:139)."""
    state_dict_key = "amp_scaler"

    def __init__(self, dynamic=None, init_scale=65536.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000):
        if dynamic is None:
            dynamic = os.environ.get("ECAMP_LOSS_SCALE", "none") == "dynamic"
        self.dynamic = bool(dynamic)
        self._state = {"scale": float(init_scale) if self.dynamic else 1.0, "growth_factor": float(growth_factor), "backoff_factor": float(backoff_factor),
                       "growth_interval": int(growth_interval), "_growth_tracker": 0}
        self._skipped = 0           # optimizer steps skipped because a gradient overflowed (dynamic mode)
        self._last_found_inf = False
        # fused path: scale / growth tracker / skipped count live on the device (f32[4]; hip_ops.loss_scale_update keeps them) so that no
        # step waits for the overflow flag; `_dev_ctl` is what the last update handed to AdamW.  The host copies above are refreshed on demand.
        self._dev = None
        self._dev_ctl = None
        self.last_step_fused = None   # whether the last dynamic update ran as the three device launches (tests assert it on the arena path)

    def _to_device(self, device):
        if self._dev is None:
            st = self._state
            self._dev = torch.tensor([st["scale"], float(st["_growth_tracker"]), float(self._skipped), 0.0], device=device, dtype=torch.float32)
            self._dev_ctl = torch.zeros((4,), device=device, dtype=torch.float32)
        return self._dev

    def _from_device(self):
        """Refresh the host copy of the scaler state (waits for the device).  No-op when the state lives on the host."""
        if self._dev is not None:
            scale, tracker, skipped, _ = self._dev.tolist()
            self._state["scale"], self._state["_growth_tracker"], self._skipped = float(scale), int(tracker), int(skipped)
            self._last_found_inf = bool(self._dev_ctl[1].item() != 0)

    @property
    def skipped_steps(self):
        self._from_device()
        return self._skipped

    @property
    def last_found_inf(self):
        self._from_device()
        return self._last_found_inf

    def get_scale(self):
        self._from_device()
        return self._state["scale"]

    def _update(self, found_inf):
        """GradScaler.update() (grad_scaler.py: _amp_update_scale_)."""
        st = self._state
        if found_inf:
            st["scale"] *= st["backoff_factor"]
            st["_growth_tracker"] = 0
        else:
            st["_growth_tracker"] += 1
            if st["_growth_tracker"] == st["growth_interval"]:
                st["scale"] *= st["growth_factor"]
                st["_growth_tracker"] = 0

    def _dynamic_step(self, optimizer, clip_grad, parameters):
        """unscale_ + grad norm + scaler.step + scaler.update of misc.py:262-269 for the scaled gradients sitting in p.grad."""
        ps = list(parameters) if parameters is not None else [p for g in optimizer.param_groups for p in g["params"]]
        live = [p for p in ps if p.grad is not None]         # (frozen tensors -- the sin-cos position tables -- carry no gradient and are not in the arena)
        arena = getattr(live[0], "_ecamp_arena", None) if live else None
        fused = (clip_grad is None and arena is not None and hasattr(optimizer, "covers") and optimizer.covers(ps)
                 and all(getattr(p, "_ecamp_arena", None) is arena for p in live))
        self.last_step_fused = fused
        if fused:
            # Everything on the device, nothing read back: one pass for sum(g^2) over the SCALED gradients (inf / nan = overflow), one
            # single-thread kernel for GradScaler's decision + update (and AdamW's bias corrections at the count of steps actually taken),
            # then AdamW, which un-scales as it reads and leaves the arenas alone when told to skip.  The reference blocks on
            # found_inf.item() here; a host that waits once per step cannot queue the next one while this one runs (+16 ms per step).
            from .. import hip_ops as ops
            st = self._state
            dev = self._to_device(arena.device)
            s = ops.zeros((1,), arena.device)
            norm = torch.empty((1,), device=arena.device, dtype=torch.float32)
            ops.sumsq(arena.flat_g, s)
            b1, b2 = optimizer.param_groups[0]["betas"]
            ops.loss_scale_update(s, dev, optimizer.step_counter(), self._dev_ctl, norm, st["growth_factor"], st["backoff_factor"],
                                  st["growth_interval"], b1, b2)
            optimizer.step(ctl=self._dev_ctl)
            return norm.reshape(())                          # inf / nan on overflow, like the reference's norm of the unscaled gradients
        self._from_device()                                  # (a fused step ran before: continue from its state on the host)
        self._dev = None
        # any optimizer / CPU tensors / clip_grad: GradScaler's own order of operations in plain torch
        inv = 1.0 / self._state["scale"]
        grads = [p.grad for p in ps if p.grad is not None]
        for g in grads:
            g.mul_(inv)
        found_inf = any(not bool(torch.isfinite(g).all()) for g in grads)
        norm = torch.nn.utils.clip_grad_norm_(ps, clip_grad) if clip_grad is not None else get_grad_norm_(ps)
        if not found_inf:
            optimizer.step()
        self._last_found_inf = found_inf
        self._skipped += int(found_inf)
        self._update(found_inf)
        return norm

    def __call__(self, loss, optimizer, clip_grad=None, parameters=None, create_graph=False, update_grad=True):
        arena = getattr(optimizer, "arena", None) if loss.is_cuda else None
        reducer = getattr(arena, "reducer", None) if arena is not None else None
        lazy = False
        if (reducer is not None and update_grad and clip_grad is None and not create_graph and _FUSED_GRAD_NORM and _BUCKETWISE_ADAMW and not self.dynamic
                and hasattr(optimizer, "step_with_grad_norm") and getattr(reducer, "side", None) is not None
                and (reducer.world > 1 or reducer.force_comm) and not reducer.host_staged):
            ps = list(parameters) if parameters is not None else [p for g in optimizer.param_groups for p in g["params"]]
            lazy = optimizer.covers(ps)
        if lazy:
            reducer.lazy = True    # the end-of-backward callback leaves the buckets' events to the optimizer (GradReducer.lazy)
        try:
            if self.dynamic:   # scaler.scale(loss).backward(); the scale is read from the device when it lives there (no host wait)
                (loss * (self._dev[0] if self._dev is not None and loss.is_cuda else self._state["scale"])).backward(create_graph=create_graph)
            else:
                loss.backward(create_graph=create_graph)
        finally:
            if lazy:
                reducer.lazy = False
        if not update_grad:
            if hasattr(optimizer, "pace"):
                optimizer.pace()   # accumulation micro-steps count against the host's lead too (optim.MAX_STEPS_IN_FLIGHT)
            return None
        if lazy:
            reducer.finalize()     # a no-op when the autograd callback has run (then the bucket events are waiting)
            return optimizer.step_with_grad_norm()   # bucket by bucket behind the all-reduces; flushes unwritten weights itself
        if reducer is not None and hasattr(reducer, "join"):
            reducer.join()
        if hasattr(optimizer, "flush_grads"):
            optimizer.flush_grads()  # weights no GEMM wrote this window read as zero (lazy zero_grad)
        if reducer is not None:
            reducer.finalize()  # normally a no-op (the autograd callback has run); the safety net when no callback could be queued
        if self.dynamic:
            return self._dynamic_step(optimizer, clip_grad, parameters)
        if clip_grad is not None:
            assert parameters is not None
            norm = torch.nn.utils.clip_grad_norm_(parameters, clip_grad)
        else:
            ps = list(parameters) if parameters is not None else [p for g in optimizer.param_groups for p in g["params"]]
            if _FUSED_GRAD_NORM and hasattr(optimizer, "step_with_grad_norm") and optimizer.covers(ps):
                return optimizer.step_with_grad_norm()   # the norm falls out of the AdamW pass over the gradients
            norm = get_grad_norm_(ps)
        optimizer.step()
        return norm

    def state_dict(self):
        self._from_device()
        return dict(self._state)

    def load_state_dict(self, state_dict):
        """GradScaler.load_state_dict's keys.  In dynamic mode the reference's scale and growth tracker are continued; in the default
        mode the scale stays 1 (a reference checkpoint's 65536 must not scale a loss nobody un-scales)."""
        keep = {k: v for k, v in state_dict.items() if k in self._state}
        if not self.dynamic:
            keep.pop("scale", None)
            keep.pop("_growth_tracker", None)
        self._from_device()
        self._state.update(keep)
        self._dev = None   # re-created from the host state at the next fused step


# --------------------------------------------------------------------------------------------- checkpoints
def save_model(args, epoch, model, model_without_ddp, optimizer, loss_scaler):
    """`{model, optimizer, epoch, scaler, args}` -> output_dir/checkpoint-<epoch>.pth on rank 0 (misc.py:295-312)."""
    path = Path(args.output_dir) / ("checkpoint-%s.pth" % str(epoch))
    to_save = {"model": model_without_ddp.state_dict(), "optimizer": optimizer.state_dict(), "epoch": epoch,
               "scaler": loss_scaler.state_dict() if loss_scaler is not None else {}, "args": args}
    save_on_master(to_save, path)


def load_model(args, model_without_ddp, optimizer, loss_scaler):
    """Key-intersection load (a plain MAE ViT-B checkpoint initialises the encoder + decoder blocks); optimizer /
    epoch / scaler are restored only for paths starting with './ECAMP' -- misc.py:315-338."""
    if not args.resume:
        return
    if args.resume.startswith("https"):
        checkpoint = torch.hub.load_state_dict_from_url(args.resume, map_location="cpu", check_hash=True)
        model_without_ddp.load_state_dict(checkpoint["model"])
    else:
        checkpoint = torch.load(args.resume, map_location="cpu", weights_only=False)
        own = model_without_ddp.state_dict()
        own.update({k: v for k, v in checkpoint["model"].items() if k in own and tuple(v.shape) == tuple(own[k].shape)})
        model_without_ddp.load_state_dict(own)
    if args.resume.startswith("./ECAMP"):
        print("Resume checkpoint %s" % args.resume)
        if "optimizer" in checkpoint and "epoch" in checkpoint and not (hasattr(args, "eval") and args.eval):
            optimizer.load_state_dict(checkpoint["optimizer"])
            args.start_epoch = checkpoint["epoch"] + 1
            if "scaler" in checkpoint:
                loss_scaler.load_state_dict(checkpoint["scaler"])
            print("With optim & sched!")
