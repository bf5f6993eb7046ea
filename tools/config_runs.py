#!/usr/bin/env python3
"""One-GPU step times of the BASELINE.json configs that are not the bench line: configs[3] ViT-L/16 at 448^2 (B=64) and configs[4]
ViT-B/16 at B=512 in bf16 and with the fp8 forward.  python tools/config_runs.py [--steps 8]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ecamp_amd import optim
from ecamp_amd.data import synthetic_batch
from ecamp_amd.module import model_ecamp
from ecamp_amd.util.misc import NativeScalerWithGradNormCount
ap = argparse.ArgumentParser(); ap.add_argument("--steps", type=int, default=8)
ap.add_argument("--only", default="", help="substring of the config name, e.g. 'configs[3]' or 'fp8'")
ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16"], help="fp16: IEEE-half activations + dynamic loss scaling (--amp fp16); the fp8 line is skipped")
args = ap.parse_args()
CD = torch.float16 if args.dtype == "fp16" else torch.bfloat16
dev = torch.device("cuda:0")
def run(name, ctor, B, big, **kw):
    if (args.only and args.only not in name) or (args.dtype == "fp16" and kw.get("fp8_forward")):
        return
    torch.manual_seed(0)
    model = ctor(compute_dtype=CD, **kw).to(dev); model.prepare(); model.train()
    opt = optim.FusedAdamW(optim.add_weight_decay(model, 0.05), lr=1.5e-4, betas=(0.9, 0.95))
    scaler = NativeScalerWithGradNormCount(dynamic=(args.dtype == "fp16"))
    batch = synthetic_batch(B, 128, big, seed=0, device=dev)
    def step():
        mim, res, mlm = model(batch)
        scaler(mim + res + mlm, opt, parameters=model.parameters(), update_grad=True)
        opt.zero_grad()
        return mim, res, mlm
    for _ in range(10): out = step()   # allocator and weight-quantisation caches settle within a few steps (fp16 at B = 512 needs more than six)
    torch.cuda.synchronize(); n0 = torch.cuda.memory_stats().get("num_device_alloc", 0); t0 = time.perf_counter()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]; marks[0].record()
    for i in range(args.steps): out = step(); marks[i + 1].record()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / args.steps
    per_step = [round(marks[i].elapsed_time(marks[i + 1]), 1) for i in range(args.steps)]
    r = {"config": name, "dtype": args.dtype, "pairs_per_gpu": B, "ms_per_step": round(1e3 * dt, 2), "pairs_per_s": round(B / dt, 1),
         "peak_hbm_gb": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1), "losses": [round(float(t), 4) for t in out],
         "step_ms": per_step, "device_allocs_in_timed_region": torch.cuda.memory_stats().get("num_device_alloc", 0) - n0,
         **({"loss_scale": scaler.get_scale(), "skipped_steps": scaler.skipped_steps} if scaler.dynamic else {})}
    print(json.dumps(r), flush=True)
    del model, opt, batch
    torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()
run("configs[3] ViT-L/16 448^2 (784 visible tokens), bf16", model_ecamp.ecamp_large_448, 64, 896)
run("configs[4] ViT-B/16 bf16 reference point, B=512", model_ecamp.ecamp, 512, 448)
run("configs[4] ViT-B/16 fp8 forward (bf16 grads), B=512", model_ecamp.ecamp, 512, 448, fp8_forward=True)
