"""`train_one_epoch` -- the MAE-style engine step of ECAMP/Pre-training/main_pretrain.py:116-180 (the reference
defines it inline in main_pretrain.py; this module is the MAE-conventional home and main_pretrain re-exports it).

Same signature, same per-iteration LR schedule, same `(mim+res+mlm)/accum_iter` loss, same meters
(`mim_loss, res_loss, mlm_loss, lr`) and return value.  What changed is when the host waits for the device:
the reference calls `.item()` x3 + `torch.cuda.synchronize()` + three scalar all-reduces every micro-step
(:143-145,155,164-166); here the three losses stay on the device, are all-reduced as ONE 3-float message, and
are only read back when a meter is printed (every `print_freq` steps) or averaged at the end of the epoch.
Gradient all-reduce runs once per optimizer step (not per micro-step), overlapped with backward.
"""
import contextlib
import ctypes
import math
from typing import Iterable

import torch

from .util import lr_sched, misc


@contextlib.contextmanager
def _range(on, name):
    """roctx range (torch.cuda.nvtx is roctx on ROCm builds) when profiling is on; free otherwise."""
    if not on:
        yield
        return
    try:
        torch.cuda.nvtx.range_push(name)
        pushed = True
    except Exception:   # a build without roctx: profiling ranges are best effort
        pushed = False
    try:
        yield
    finally:
        if pushed:
            torch.cuda.nvtx.range_pop()


def _flush_tb(log_writer, pending):
    """Write the buffered (epoch_1000x, [mim, res, mlm] on the device, lr) records: ONE device read-back for all of them."""
    if not pending:
        return
    vals = torch.stack([r for _, r, _ in pending]).tolist()
    for (x, _, lr), r in zip(pending, vals):
        if not all(math.isfinite(v) for v in r):
            print("warning: non-finite loss {}".format(r))
        log_writer.add_scalar("mim_loss", r[0], x)
        log_writer.add_scalar("res_loss", r[1], x)
        log_writer.add_scalar("mlm_loss", r[2], x)
        log_writer.add_scalar("lr", lr, x)
    del pending[:]


def train_one_epoch(model: torch.nn.Module, data_loader: Iterable, optimizer: torch.optim.Optimizer, device: torch.device,
                    epoch: int, loss_scaler, log_writer=None, args=None):
    model.train(True)
    metric_logger = misc.MetricLogger(delimiter="  ")
    metric_logger.add_meter("lr", misc.SmoothedValue(window_size=1, fmt="{value:.6f}"))
    header = "Epoch: [{}]".format(epoch)
    print_freq = getattr(args, "print_freq", 20)
    accum_iter = args.accum_iter
    optimizer.zero_grad()
    if log_writer is not None:
        print("log_dir: {}".format(log_writer.log_dir))
    n_iter = len(data_loader)
    tb_pending = []
    prof = bool(getattr(args, "profile", False))
    if prof:
        from . import _lib
        lib = _lib.load()
        lib.ecamp_prof_collect(-1, None, None, None)
        lib.ecamp_prof_enable(1)   # HIP events around every GEMM / attention launch (csrc/profile.hip)
    if torch.device(device).type == "cuda" and getattr(args, "prefetch", True):
        from .data import DevicePrefetcher
        data_loader = DevicePrefetcher(data_loader, device)   # batch i+1 crosses PCIe while step i computes
    for data_iter_step, batch in enumerate(metric_logger.log_every(data_loader, print_freq, header)):
        # per-iteration (not per-epoch) lr schedule, updated at accumulation boundaries only (main_pretrain.py:137-138)
        if data_iter_step % accum_iter == 0:
            lr_sched.adjust_learning_rate(optimizer, data_iter_step / n_iter + epoch, args)
        update_grad = (data_iter_step + 1) % accum_iter == 0
        if hasattr(model, "set_grad_sync"):
            model.set_grad_sync(update_grad)
        with _range(prof, "ecamp/step %d" % data_iter_step):
            with _range(prof, "ecamp/forward"):
                mim_loss, res_loss, mlm_loss = model(batch, mask_ratio=args.mask_ratio)
                loss = (mim_loss + res_loss + mlm_loss) / accum_iter
            with _range(prof, "ecamp/backward+allreduce+adamw" if update_grad else "ecamp/backward"):
                loss_scaler(loss, optimizer, parameters=model.parameters(), update_grad=update_grad)
                if update_grad:
                    optimizer.zero_grad()
        losses = torch.stack([mim_loss.detach(), res_loss.detach(), mlm_loss.detach()])
        metric_logger.update(mim_loss=losses[0], res_loss=losses[1], mlm_loss=losses[2])
        lr = optimizer.param_groups[0]["lr"]
        metric_logger.update(lr=lr)
        reduced = misc.all_reduce_mean(losses)
        if log_writer is not None and update_grad:
            # epoch_1000x as the x-axis calibrates curves across batch sizes (main_pretrain.py:168-175).  The values stay on the device
            # until the meters are printed anyway (every `print_freq` steps) or the epoch ends: no host synchronisation per optimizer step
            tb_pending.append((int((data_iter_step / n_iter + epoch) * 1000), reduced, lr))
            if (data_iter_step + 1) % print_freq == 0 or data_iter_step == n_iter - 1:
                _flush_tb(log_writer, tb_pending)
        if getattr(loss_scaler, "dynamic", False) and ((data_iter_step + 1) % print_freq == 0 or data_iter_step == n_iter - 1):
            # GradScaler's state is on the device; it is read back where the meters are printed anyway.  A scale below 1 means every recent
            # step overflowed whatever the scale -- in IEEE half that is an activation past 65504 in the FORWARD pass (the residual stream is
            # stored in half here, f32 under the reference's autocast: INTEGRATION.md), and training is no longer making progress
            sc = loss_scaler.get_scale()
            metric_logger.update(loss_scale=sc)
            if sc < 1.0:
                print("warning: the dynamic loss scale has fallen to %g after %d skipped steps -- every step overflows; fp16 activations out of range?"
                      % (sc, loss_scaler.skipped_steps))
    if log_writer is not None:
        _flush_tb(log_writer, tb_pending)
    if prof:
        torch.cuda.synchronize()
        lib.ecamp_prof_enable(0)
        for cat, name in ((0, "bf16 GEMM"), (1, "f32 GEMM"), (2, "attention")):
            ms, fl, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
            lib.ecamp_prof_collect(cat, ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(n))
            if n.value:
                print("profile: %-10s %7d launches  %9.2f ms/step  %7.1f TFLOP/s (HIP events on the launch stream)"
                      % (name, n.value, ms.value / max(n_iter, 1), fl.value / max(ms.value, 1e-9) / 1e9))
    metric_logger.synchronize_between_processes()
    print("Averaged stats:", metric_logger)
    return {k: meter.global_avg for k, meter in metric_logger.meters.items()}
