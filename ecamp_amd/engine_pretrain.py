"""`train_one_epoch` -- the MAE-style engine step of ECAMP/Pre-training/main_pretrain.py:116-180 (the reference
defines it inline in main_pretrain.py; this module is the MAE-conventional home and main_pretrain re-exports it).

Same signature, same per-iteration LR schedule, same `(mim+res+mlm)/accum_iter` loss, same meters
(`mim_loss, res_loss, mlm_loss, lr`) and return value.  What changed is when the host waits for the device:
the reference calls `.item()` x3 + `torch.cuda.synchronize()` + three scalar all-reduces every micro-step
(:143-145,155,164-166); here the three losses stay on the device, are all-reduced as ONE 3-float message, and
are only read back when a meter is printed (every `print_freq` steps) or averaged at the end of the epoch.
Gradient all-reduce runs once per optimizer step (not per micro-step), overlapped with backward.
"""
import math
from typing import Iterable

import torch

from .util import lr_sched, misc


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
    for data_iter_step, batch in enumerate(metric_logger.log_every(data_loader, print_freq, header)):
        # per-iteration (not per-epoch) lr schedule, updated at accumulation boundaries only (main_pretrain.py:137-138)
        if data_iter_step % accum_iter == 0:
            lr_sched.adjust_learning_rate(optimizer, data_iter_step / n_iter + epoch, args)
        update_grad = (data_iter_step + 1) % accum_iter == 0
        if hasattr(model, "set_grad_sync"):
            model.set_grad_sync(update_grad)
        mim_loss, res_loss, mlm_loss = model(batch, mask_ratio=args.mask_ratio)
        loss = (mim_loss + res_loss + mlm_loss) / accum_iter
        loss_scaler(loss, optimizer, parameters=model.parameters(), update_grad=update_grad)
        if update_grad:
            optimizer.zero_grad()
        losses = torch.stack([mim_loss.detach(), res_loss.detach(), mlm_loss.detach()])
        metric_logger.update(mim_loss=losses[0], res_loss=losses[1], mlm_loss=losses[2])
        lr = optimizer.param_groups[0]["lr"]
        metric_logger.update(lr=lr)
        reduced = misc.all_reduce_mean(losses)
        if log_writer is not None and update_grad:
            # epoch_1000x as the x-axis calibrates curves across batch sizes (main_pretrain.py:168-175)
            epoch_1000x = int((data_iter_step / n_iter + epoch) * 1000)
            r = reduced.tolist()
            if not all(math.isfinite(v) for v in r):
                print("warning: non-finite loss {}".format(r))
            log_writer.add_scalar("mim_loss", r[0], epoch_1000x)
            log_writer.add_scalar("res_loss", r[1], epoch_1000x)
            log_writer.add_scalar("mlm_loss", r[2], epoch_1000x)
            log_writer.add_scalar("lr", lr, epoch_1000x)
    metric_logger.synchronize_between_processes()
    print("Averaged stats:", metric_logger)
    return {k: meter.global_avg for k, meter in metric_logger.meters.items()}
