"""ECAMP pre-training driver -- same command line as ECAMP/Pre-training/main_pretrain.py (run.sh:3-16 works
unchanged), on the MI355X-native model / optimizer / data-parallel reducer.

    python -m torch.distributed.run --nproc_per_node=8 -m ecamp_amd.main_pretrain --batch_size 256 --accum_iter 8 ...

Extra flags (all optional): --compute_dtype {bf16,fp16,fp32} (--amp fp16), --max_caption_length, --synthetic, --synthetic_len, --print_freq, --profile, --no_prefetch.
When `--data_path` holds the MIMIC-CXR CSVs the `ContextBertDataset` of module/pretrain_datasets.py is used (batched
entity-aware masker, bit-exact against the reference loop).  A missing CSV is an error (as in the reference) unless `--synthetic`
asks for the synthetic stand-in with the same batch schema.  `--profile` puts roctx ranges around every step and its phases
(visible to `rocprofv3 --marker-trace --kernel-trace -- python -m ecamp_amd.main_pretrain ...`) and prints the library's own
HIP-event totals of the GEMM and attention kernels per epoch.
"""
import argparse
import datetime
import json
import os
import random
import shutil
import time
from pathlib import Path

import numpy as np
import torch
from torch.utils.data import DataLoader, DistributedSampler

from . import optim as optim_factory
from .data import SyntheticContextBertDataset
from .engine_pretrain import train_one_epoch  # noqa: F401  (re-exported: the reference defines it in this module)
from .module import model_ecamp
from .parallel import DistributedDataParallel
from .util import misc
from .util.misc import NativeScalerWithGradNormCount as NativeScaler


def dump(file_path, args):
    import yaml
    with open(file_path, "w") as f:
        yaml.dump(data={k: (v if isinstance(v, (int, float, str, bool, type(None))) else str(v)) for k, v in vars(args).items()}, stream=f)


def get_args_parser():
    p = argparse.ArgumentParser("ECAMP pre-training", add_help=False)
    p.add_argument("--description", type=str, default="ecamp_pretrain")
    p.add_argument("--batch_size", default=256, type=int, help="Batch size per GPU (effective batch size is batch_size * accum_iter * # gpus")
    p.add_argument("--epochs", default=115, type=int)
    p.add_argument("--max_epoch", default=200, type=int)
    p.add_argument("--accum_iter", default=2, type=int)
    p.add_argument("--model", default="ecamp", type=str, metavar="MODEL")
    p.add_argument("--input_size", default=448, type=int, help="images input size")
    p.add_argument("--mask_ratio", default=0.75, type=float)
    p.add_argument("--norm_pix_loss", action="store_true")
    p.set_defaults(norm_pix_loss=False)
    p.add_argument("--weight_decay", type=float, default=0.05)
    p.add_argument("--lr", type=float, default=None, metavar="LR", help="learning rate (absolute lr)")
    p.add_argument("--min_lr", type=float, default=0.0, metavar="LR")
    p.add_argument("--warmup_epochs", type=int, default=40, metavar="N")
    p.add_argument("--data_path", default="./dataset_dir", type=str)
    p.add_argument("--output_dir", default="./output_dir")
    p.add_argument("--job_dir", default="code_repo")
    p.add_argument("--log_dir", default="./output_dir")
    p.add_argument("--seed", default=42, type=int)
    p.add_argument("--resume", default="")
    p.add_argument("--start_epoch", default=0, type=int, metavar="N")
    p.add_argument("--num_workers", default=16, type=int)
    p.add_argument("--pin_mem", action="store_true")
    p.add_argument("--no_pin_mem", action="store_false", dest="pin_mem")
    p.set_defaults(pin_mem=True)
    p.add_argument("--world_size", default=1, type=int)
    p.add_argument("--local_rank", "--local-rank", default=-1, type=int)
    p.add_argument("--dist_on_itp", action="store_true")
    p.add_argument("--dist_url", default="env://")
    # additions of this implementation
    p.add_argument("--compute_dtype", default="bf16", choices=["bf16", "fp16", "fp32"], help="16-bit storage format of the activations: bf16 "
                   "(default, the benchmarked mode), fp16 = IEEE half, what the reference's torch.cuda.amp.autocast() computes in "
                   "(main_pretrain.py:139; libecamp_hip_f16.so; implies --loss_scale dynamic, the reference's GradScaler), fp32 = parity mode")
    p.add_argument("--amp", default="", choices=["", "bf16", "fp16"], help="alias: --amp fp16 == --compute_dtype fp16 --loss_scale dynamic")
    p.add_argument("--gelu_saved_grad", default=1, type=int, choices=[0, 1], help="bf16 mode: 1 = the fc1 / BertIntermediate epilogue saves gelu'(x) "
                   "for the backward pass (no erf in backward; one more bf16 rounding of the derivative), 0 = save x and recompute gelu' in f32 like "
                   "the reference's GeluBackward")
    p.add_argument("--max_caption_length", default=256, type=int)
    p.add_argument("--synthetic", action="store_true", help="train on the synthetic stand-in dataset (random images and tokens) instead "
                   "of <data_path>/mimic-cxr-2.0.0-entity-llm.csv; without this flag a missing CSV is an error, as in the reference")
    p.add_argument("--synthetic_len", default=4096, type=int, help="samples per epoch of the synthetic dataset")
    p.add_argument("--image_u8", action="store_true", help="compact image schema: the dataset hands over the uint8 grayscale crop [448,448] instead of "
                   "the normalised f32 [3,448,448] (12x fewer bytes through the loader, PCIe and HBM; the kernels normalise on the fly, same bits)")
    p.add_argument("--loss_scale", default="none", choices=["none", "dynamic"], help="dynamic: the reference's torch.cuda.amp.GradScaler() semantics "
                   "(util/misc.py:251-271: loss x scale, inf / nan check, skipped step + backoff, growth every 2000 clean steps; the scaler state of a "
                   "reference checkpoint is continued).  none (default): bf16 / f32 need no loss scaling, the scale is 1")
    p.add_argument("--image_shard", default="", help="device image pipeline: a uint8 shard of the pre-decoded radiographs in CSV row order "
                   "(module/pretrain_datasets.py: U8ShardWriter; tools/make_image_shard.py writes one): loader workers hand over the bytes of each sample's crop box, "
                   "the GPU does RandomResizedCrop's resize + flip + Grayscale, byte for byte what the host transform gives on the same pixels")
    p.add_argument("--profile", action="store_true", help="roctx ranges around every optimizer step and its phases (rocprofv3 --marker-trace)")
    p.add_argument("--no_prefetch", action="store_false", dest="prefetch", help="copy each batch inside forward like the reference does")
    p.set_defaults(prefetch=True)
    p.add_argument("--print_freq", default=20, type=int)
    p.add_argument("--snapshot_code", action="store_true", help="copy ./ into output_dir/job_dir like the reference does")
    return p


def main(args):
    misc.init_distributed_mode(args)
    if args.lr is None:
        raise SystemExit("--lr is required (absolute learning rate), as in the reference")
    device = torch.device("cuda")
    seed = args.seed + misc.get_rank()
    torch.manual_seed(seed)
    np.random.seed(seed)

    csv = os.path.join(args.data_path, "mimic-cxr-2.0.0-entity-llm.csv")   # main_pretrain.py:195
    if args.synthetic:
        print("WARNING: --synthetic: training on RANDOM images and tokens (no dataset is read); checkpoints are meaningless")
        dataset_train = SyntheticContextBertDataset(args.synthetic_len, args.max_caption_length, args.input_size, seed=args.seed,
                                                    image_u8=args.image_u8)
    elif os.path.exists(csv):
        from .module.pretrain_datasets import ContextBertDataset
        random.seed(seed)  # the item pipeline draws from Python's `random` (pretrain_datasets.py:98,121,123)
        dataset_train = ContextBertDataset(os.path.join(args.data_path), max_caption_length=args.max_caption_length, image_u8=args.image_u8,
                                           image_shard=args.image_shard or None)
        if misc.is_main_process():
            from .module.pretrain_datasets import measure_item_rate
            rate = measure_item_rate(dataset_train)
            print("data loader: %.0f items/s per worker (decode + augmentation + masking on the host), x %d workers = %.0f items/s against "
                  "~6.8 k pairs/s per MI355X at B=256" % (rate, max(args.num_workers, 1), rate * max(args.num_workers, 1)))
    else:
        raise FileNotFoundError("%s not found: check --data_path (the reference fails in ContextBertDataset.__init__ here too); pass "
                                "--synthetic to run on the synthetic stand-in dataset instead" % csv)
    args.data = "synthetic" if args.synthetic else "mimic-cxr"   # recorded in config.yaml and log.txt
    num_tasks, global_rank = misc.get_world_size(), misc.get_rank()
    sampler_train = DistributedSampler(dataset_train, num_replicas=num_tasks, rank=global_rank, shuffle=True)
    print("Sampler_train = %s" % str(sampler_train))

    args.log_dir = os.path.join(args.output_dir, "tensorboard")
    log_writer = None
    if global_rank == 0 and args.log_dir is not None:
        os.makedirs(args.log_dir, exist_ok=True)
        try:
            from torch.utils.tensorboard import SummaryWriter
            log_writer = SummaryWriter(log_dir=args.log_dir)
        except Exception:  # tensorboard is optional
            print("tensorboard not available: scalars go to log.txt only")
        if args.snapshot_code:
            dst = os.path.join(args.output_dir, args.job_dir)
            if os.path.exists(dst):
                shutil.rmtree(dst)
            shutil.copytree(os.path.dirname(os.path.abspath(__file__)), dst)

    data_loader_train = DataLoader(dataset_train, sampler=sampler_train, batch_size=args.batch_size, num_workers=args.num_workers,
                                   pin_memory=args.pin_mem, drop_last=True, collate_fn=dataset_train.collate_fn)

    if args.amp:
        args.compute_dtype = args.amp
    if args.compute_dtype == "fp16":
        args.loss_scale = "dynamic"   # half's 5-bit exponent needs the reference's GradScaler (util/misc.py:251-271), as its autocast does
    dtype = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[args.compute_dtype]
    model = model_ecamp.__dict__[args.model](norm_pix_loss=args.norm_pix_loss, compute_dtype=dtype, gelu_saved_grad=bool(args.gelu_saved_grad))
    model.to(device)
    model_without_ddp = model
    print("Model = %s" % str(model_without_ddp))
    eff_batch_size = args.batch_size * args.accum_iter * misc.get_world_size()
    print("actual lr: %.2e" % args.lr)
    print("accumulate grad iterations: %d" % args.accum_iter)
    print("effective batch size: %d" % eff_batch_size)

    model_without_ddp.prepare()
    if args.distributed:
        gd = os.environ.get("ECAMP_DDP_GRAD_DTYPE", "f32")   # no reference counterpart: "bf16" halves the all-reduce payload
        model = DistributedDataParallel(model, device_ids=[args.gpu], find_unused_parameters=True,
                                        grad_dtype=torch.bfloat16 if gd == "bf16" else None)
        model_without_ddp = model.module

    # following timm: no weight decay for bias and norm layers
    param_groups = optim_factory.add_weight_decay(model_without_ddp, args.weight_decay)
    optimizer = optim_factory.FusedAdamW(param_groups, lr=args.lr, betas=(0.9, 0.95))
    print(optimizer)
    loss_scaler = NativeScaler(dynamic=(args.loss_scale == "dynamic"))
    if args.output_dir and misc.is_main_process():
        dump(os.path.join(args.output_dir, "config.yaml"), args)
    misc.load_model(args=args, model_without_ddp=model_without_ddp, optimizer=optimizer, loss_scaler=loss_scaler)

    print(f"Start training for {args.epochs} epochs")
    start_time = time.time()
    write_description = False
    for epoch in range(args.start_epoch, args.epochs):
        if args.distributed:
            data_loader_train.sampler.set_epoch(epoch)
        train_stats = train_one_epoch(model, data_loader_train, optimizer, device, epoch, loss_scaler, log_writer=log_writer, args=args)
        if args.output_dir:
            # checkpoint cadence of main_pretrain.py:274-292
            if (epoch == 0) or (60 <= epoch < 100 and epoch % 10 == 0) or (epoch >= 100 and (epoch % 5 == 0 or epoch + 1 == args.epochs)):
                misc.save_model(args=args, model=model, model_without_ddp=model_without_ddp, optimizer=optimizer, loss_scaler=loss_scaler, epoch=epoch)
        log_stats = {**{f"train_{k}": v for k, v in train_stats.items()}, "epoch": epoch}
        if args.output_dir and misc.is_main_process():
            if log_writer is not None:
                log_writer.flush()
            with open(os.path.join(args.output_dir, "log.txt"), mode="a", encoding="utf-8") as f:
                if not write_description:
                    f.write(args.description + "\n")
                    write_description = True
                f.write(json.dumps(dict(log_stats, data=args.data)) + "\n")
    total_time = time.time() - start_time
    print("Training time {}".format(str(datetime.timedelta(seconds=int(total_time)))))


if __name__ == "__main__":
    args = get_args_parser().parse_args()
    if args.output_dir:
        Path(args.output_dir).mkdir(parents=True, exist_ok=True)
    main(args)
