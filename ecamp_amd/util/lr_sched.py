"""Per-iteration LR schedule of ECAMP/Pre-training/util/lr_sched.py:9-21: linear warm-up for `warmup_epochs`, then
a half cosine that reaches `min_lr` at `args.max_epoch` (NOT `args.epochs` -- run.sh trains 120 of 200)."""
import math


def adjust_learning_rate(optimizer, epoch, args):
    if epoch < args.warmup_epochs:
        lr = args.lr * epoch / args.warmup_epochs
    else:
        progress = (epoch - args.warmup_epochs) / (args.max_epoch - args.warmup_epochs)
        lr = args.min_lr + (args.lr - args.min_lr) * 0.5 * (1.0 + math.cos(math.pi * progress))
    for group in optimizer.param_groups:
        group["lr"] = lr * group["lr_scale"] if "lr_scale" in group else lr
    return lr
