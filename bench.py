#!/usr/bin/env python3
"""bench.py -- ECAMP pre-training throughput on MI355X (the metric of BASELINE.json).

    python bench.py --gpus 1 --steps 50 --warmup 10      (the defaults: SURVEY.md 8(d))
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" = one full optimizer-inclusive pre-training micro-step of configs[1]: ViT-B/16 MAE encoder/decoder + SR
head + reference BERT (6L/6H/1536, vocab 30000) with context fusion, B=256 pairs per GPU, 224^2 encoder input (448^2
images resized on device), reports of S=128 tokens, bf16 activations / f32 master weights, train mode (dropout
active), host->HBM copy of the batch + forward + backward + grad all-reduce (N>1) + grad-norm + fused AdamW + zero_grad
(accum_iter=1).  `value` is timed with the batch ALREADY RESIDENT IN HBM when the timed region starts (the tier contract's definition
of `value`; rounds 2-5 printed the PCIe-inclusive rate there).  The PCIe-inclusive rate is measured right behind it and printed beside it
as `host_inclusive_*`: the batch starts in pinned HOST memory (what a DataLoader with pin_memory hands over) and crosses PCIe inside that
timed region on a copy stream, two steps ahead of its use (ecamp_amd.data.DevicePrefetcher; each timed step issues one batch copy and
consumes one issued two steps earlier), with one HIP event per step on the compute stream and an event pair around every batch copy on
the copy stream, so that a gap between the two rates is explained by the record itself (`step_ms` min / median / p90 / max,
`h2d_ms_per_step`, `h2d_gbps`; a step of 35.8 ms hides a 616 MB copy only above 17.2 GB/s).  Forward-only and forward+backward-only
rates are reported too.

`python bench.py --gpus N` with N > 1 and no RANK in the environment launches its own N ranks (one child process per GPU,
RCCL over xGMI) before anything in this process touches the GPU; under torch.distributed.run it uses the ranks it is given.

Prints ONE JSON line on rank 0 with `roofline` (dominant kernel = the bf16 MFMA GEMM family, timed live with HIP
events on the launch stream over the timed region) and `cpu_baseline` (the oracle -- a CPU restatement of the
reference validated against it -- timed on this box's host cores on a bounded sample; kind="port").
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0  # dense MFMA bf16 peak of MI355X (MI355X_MICROARCH.md; AMD's 5 PF figure is 2:1 sparse)
METRIC = "pretrain image-report pairs/sec (ViT-B/16, 224^2, seq=128)"


def cpu_info():
    """CPU model and physical core count of this box (/proc/cpuinfo)."""
    model, cores = None, set()
    try:
        phys = core = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and model is None:
                model = line.split(":", 1)[1].strip()
            elif line.startswith("physical id"):
                phys = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                core = line.split(":", 1)[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    cores.add((phys, core))
                phys = core = None
    except OSError:
        pass
    return model, (len(cores) or None), os.cpu_count()


def cpu_baseline(seq, budget_s=25.0):
    """Oracle fwd+bwd+AdamW on the host cores, B=8 pairs/step, fp32."""
    from oracle import ecamp_oracle as orc
    from oracle import recipe
    # torch CPU kernels stop scaling (and thrash) far below the 256 hardware threads of the GPU box: use 32
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    cfg = orc.cfg_base()
    B = 8
    P = orc.set_requires_grad(orc.load_state(orc.new_params(cfg), recipe.recipe_state(cfg, seed=0)), cfg)
    names = [k for k in orc.trainable_names(cfg)]
    no_decay = set(orc.weight_decay_groups(cfg)[0])
    m = {k: torch.zeros_like(P[k]) for k in names}
    v = {k: torch.zeros_like(P[k]) for k in names}
    batch = recipe.recipe_batch(cfg, B, seq, seed=0)

    def step(i):
        mim, res, mlm = orc.forward(P, cfg, batch, 0.75, recipe.recipe_noise(B, cfg.num_patches, seed=i), train=True)
        (mim + res + mlm).backward()
        with torch.no_grad():
            for k in names:
                if P[k].grad is not None:
                    orc.adamw_step(P[k], P[k].grad, m[k], v[k], i + 1, 1.5e-4, 0.0 if k in no_decay else 0.05)
                    P[k].grad = None

    tw = time.time()
    step(0)  # warm-up
    tw = time.time() - tw
    t0, n = time.time(), 0
    while n < 1 or (time.time() - t0 + tw < budget_s and n < 8):
        step(n + 1)
        n += 1
    dt = time.time() - t0
    model, phys, logical = cpu_info()
    # the other legs of BASELINE.md section 3 (S=256 at B=8; B=32): 1 warm-up + 3 timed optimizer-inclusive steps each
    legs = []
    for b2, s2 in ((8, 256), (32, seq)):
        try:
            batch = recipe.recipe_batch(cfg, b2, s2, seed=1)
            B = b2
            step(100)   # warm-up of this shape
            tl = time.time()
            for j in range(3):
                step(101 + j)
            legs.append({"B": b2, "S": s2, "pairs_per_s": round(3 * b2 / (time.time() - tl), 4), "steps": 3, "warmup": 1})
        except Exception as e:
            legs.append({"B": b2, "S": s2, "error": repr(e)})
    # BASELINE.md section 3 asks for "all physical cores": the same two shapes once more with one thread per physical core of the box
    # (reported beside the 32-thread legs; `value` stays the better-scaling 32-thread figure, and `cores` says so)
    all_legs = []
    if phys and phys != cores:
        torch.set_num_threads(phys)
        for b2, s2 in ((8, seq), (32, seq)):
            try:
                batch = recipe.recipe_batch(cfg, b2, s2, seed=2)
                B = b2
                step(200)
                tl = time.time()
                for j in range(2):
                    step(201 + j)
                all_legs.append({"B": b2, "S": s2, "threads": phys, "pairs_per_s": round(2 * b2 / (time.time() - tl), 4), "steps": 2, "warmup": 1})
            except Exception as e:
                all_legs.append({"B": b2, "S": s2, "threads": phys, "error": repr(e)})
        torch.set_num_threads(cores)
    return {"value": round(8 * n / dt, 4), "unit": "pairs/s", "cores": cores, "kind": "port", "cpu_model": model, "physical_cores": phys,
            "logical_cpus": logical, "other_legs": legs, "all_physical_cores_legs": all_legs,
            "sample": "oracle/ecamp_oracle.py (CPU restatement of the reference, fp32): %d optimizer-inclusive steps of B=8, S=%d, "
                      "448^2 images, dropout on, after 1 warm-up; torch %s, %d threads (torch's CPU kernels stop scaling far below the box's "
                      "thread count: `all_physical_cores_legs` has the same step with one thread per physical core)" % (n, seq, torch.__version__, cores)}


def launch_ranks(cmd, n, poll_s=0.2, grace_s=5.0, check_devices=True, limit_s=None):
    """Self-launch: `n` fresh child processes of `cmd`, one rank per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their
    environment), never re-exec'ed.  The children are polled: when one exits non-zero -- or when the whole job passes `limit_s`
    seconds of wall clock (env ECAMP_BENCH_LIMIT_S, default 1500: a rank hung inside an RCCL collective never exits by itself) --
    the others are terminated, the parent says which rank failed (or that the limit was hit, with every rank's stderr tail) and
    returns non-zero.  Every rank's stderr is also teed to this process's stderr line by line while the job runs, so a hang is not
    silent.  Returns 0 when every rank exited 0."""
    import tempfile
    import threading
    from ecamp_amd.parallel import rccl_env_defaults
    if limit_s is None:
        limit_s = float(os.environ.get("ECAMP_BENCH_LIMIT_S", "1500"))
    if check_devices:
        # (on ROCm this may call hipGetDeviceCount in the parent; harmless -- the ranks are fresh child processes, nothing is exec'ed)
        have = torch.cuda.device_count()
        if have < n:
            print("bench.py: --gpus %d but this box has %d visible GPU(s): nothing launched" % (n, have), file=sys.stderr)
            return 2
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    threads = max(1, (os.cpu_count() or n) // n)   # host threads per rank (pinning / copies / CPU-side torch ops): cores / N, not 256 each
    procs, logs, tees = [], [], []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        env.setdefault("OMP_NUM_THREADS", str(min(threads, 32)))
        rccl_env_defaults(env)   # the channel cap etc. (ecamp_amd/parallel.py); an explicit NCCL_* setting wins
        log = tempfile.TemporaryFile(mode="w+")
        logs.append(log)
        pr = subprocess.Popen(cmd, env=env, stderr=subprocess.PIPE, text=True, bufsize=1)
        procs.append(pr)

        def tee(pr=pr, log=log, r=r):   # the child's stderr, line by line: kept for the failure report AND passed on as it comes
            for line in pr.stderr:
                log.write(line)
                sys.stderr.write(line if n == 1 else "[rank %d] %s" % (r, line))
            pr.stderr.close()
        th = threading.Thread(target=tee, daemon=True)
        th.start()
        tees.append(th)
    failed = None
    t_limit = time.time() + limit_s
    while failed is None:
        codes = [pr.poll() for pr in procs]
        for r, c in enumerate(codes):
            if c is not None and c != 0:
                failed = (r, c)
                break
        if failed is None and all(c == 0 for c in codes):
            break
        if failed is None and time.time() > t_limit:
            failed = (-1, 124)
            break
        time.sleep(poll_s)
    if failed is not None:
        for pr in procs:
            if pr.poll() is None:
                pr.terminate()
        t_end = time.time() + grace_s
        for pr in procs:
            try:
                pr.wait(timeout=max(0.0, t_end - time.time()))
            except subprocess.TimeoutExpired:
                pr.kill()
                pr.wait()
    for th in tees:
        th.join(timeout=grace_s)
    if failed is not None:
        r, c = failed
        if r < 0:
            print("bench.py: the %d-rank job passed its wall-clock limit of %.0f s (ECAMP_BENCH_LIMIT_S); every rank was terminated.  "
                  "Stderr tails:" % (n, limit_s), file=sys.stderr)
            show = range(n)
        else:
            print("bench.py: rank %d of %d exited with code %d; the other ranks were terminated.  Its stderr (tail):" % (r, n, c), file=sys.stderr)
            show = [r]
        for q in show:
            logs[q].seek(0)
            for line in logs[q].read().splitlines()[-25:]:
                print("  [rank %d] %s" % (q, line), file=sys.stderr)
        return abs(c) or 1
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)   # >= 50 timed steps after 10 warm-up
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=256, help="pairs per GPU (configs[1] = 256)")
    ap.add_argument("--seq", type=int, default=128)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16", "fp32"], help="bf16: the headline mode.  fp16: IEEE-half activations (the reference's "
                    "autocast format) with its dynamic loss scaling inside the timed step (libecamp_hip_f16.so).  fp32: parity mode")
    ap.add_argument("--image-u8", action="store_true", help="compact image schema: uint8 [B,448,448] grayscale crops (51 MB per 256 pairs over PCIe instead "
                    "of 616 MB of f32 [B,3,448,448]); the default stays the reference's f32 schema")
    ap.add_argument("--grad-dtype", default=os.environ.get("ECAMP_DDP_GRAD_DTYPE", "f32"), choices=["f32", "bf16"],
                    help="payload of the gradient all-reduce at N > 1 (f32 = the reference's; bf16 halves the bytes on the links)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prof", action="store_true", help="do not bracket GEMM launches with HIP events")
    ap.add_argument("--only-value", action="store_true", help="time the K steps of `value` and nothing else (kernel traces of the production step)")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        raise SystemExit(launch_ranks([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], args.gpus))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d: launch `python bench.py --gpus N` (self-launching) or "
                         "`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    rccl_log = None
    if world > 1:
        from ecamp_amd.parallel import rccl_env_defaults
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        rccl_env_defaults()   # the driver launches the ranks through torch.distributed.run, not through launch_ranks: set the caps here too
        if rank == 0 and "NCCL_DEBUG" not in os.environ:   # RCCL's own account of its channels, for the `rccl` record
            import tempfile
            rccl_log = os.path.join(tempfile.gettempdir(), "ecamp_rccl_%d.log" % os.getpid())
            os.environ.update(NCCL_DEBUG="INFO", NCCL_DEBUG_SUBSYS="INIT,GRAPH", NCCL_DEBUG_FILE=rccl_log)
        dist.init_process_group("nccl", device_id=dev)  # RCCL over xGMI

    from ecamp_amd import _lib, optim
    from ecamp_amd.data import synthetic_batch
    from ecamp_amd.module import model_ecamp
    from ecamp_amd.parallel import DistributedDataParallel, ddp_defaults, rccl_env_record
    from ecamp_amd.util.misc import NativeScalerWithGradNormCount

    torch.manual_seed(42 + rank)  # main_pretrain.py:189
    cd = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[args.dtype]
    half16 = args.dtype in ("bf16", "fp16")
    model = model_ecamp.ecamp(compute_dtype=cd).to(dev)
    model.prepare()
    net = DistributedDataParallel(model, grad_dtype=torch.bfloat16 if args.grad_dtype == "bf16" else None) if world > 1 else model
    opt = optim.FusedAdamW(optim.add_weight_decay(model, 0.05), lr=1.5e-4, betas=(0.9, 0.95))
    scaler = NativeScalerWithGradNormCount(dynamic=(args.dtype == "fp16"))   # fp16: GradScaler's check + skip + update run in every timed step
    from ecamp_amd.data import DevicePrefetcher
    host_batch = {k: v.pin_memory() for k, v in synthetic_batch(args.batch, args.seq, 448, seed=rank, device="cpu", image_u8=args.image_u8).items()}   # what a
    # pin_memory DataLoader yields (main_pretrain.py:232-240); it crosses PCIe inside the timed region, one step ahead of its use
    batch = {k: v.to(dev) for k, v in host_batch.items()}   # resident copy for the side measurements
    net.train()
    opt.zero_grad()

    def step(b=None):
        mim, res, mlm = net(batch if b is None else b)
        norm = scaler(mim + res + mlm, opt, parameters=model.parameters(), update_grad=True)
        opt.zero_grad()
        return mim, res, mlm, norm

    def timed(fn, n):
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn(n)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        return time.perf_counter() - t0

    step_marks = []   # one HIP event per step on the compute stream (plus one in front of the first): where the time of a run went, step by step

    def mark():
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(torch.cuda.current_stream(dev))
        step_marks.append(ev)
        host_marks.append(time.perf_counter())

    host_marks = []   # host clock when each step's launches had been queued (beside the device events: is a slow step the GPU's or the host's?)

    def step_stats(marks):
        """ms between consecutive per-step events (call after a synchronize): min / median / p90 / max, the steps in order, and for
        the slowest step what the host was doing (ms the host took to queue that step; a host stall shows there, a GPU stall does not)."""
        seq = [a.elapsed_time(b) for a, b in zip(marks[:-1], marks[1:])]
        d = sorted(seq)
        if not d:
            return None
        q = lambda f: d[min(len(d) - 1, int(f * len(d)))]
        out = {"min": round(d[0], 3), "median": round(q(0.5), 3), "p90": round(q(0.9), 3), "max": round(d[-1], 3),
               "steps_over_1p1_median": sum(1 for v in seq if v > 1.1 * q(0.5)), "sequence": [round(v, 2) for v in seq]}
        if len(host_marks) == len(marks):
            hq = [1e3 * (b - a) for a, b in zip(host_marks[:-1], host_marks[1:])]
            out["host_queue_ms"] = {"median": round(sorted(hq)[len(hq) // 2], 3), "max": round(max(hq), 3),
                                    "of_slowest_step": round(hq[seq.index(d[-1])], 3)}
        return out

    def runtime_counters():
        """What can stall a step from outside the kernels: device allocations by the caching allocator (a hipMalloc / hipFree inside a step
        synchronises the device) and Python's cyclic garbage collector (a generation-2 pass over a large heap takes tens of ms)."""
        import gc
        ms = torch.cuda.memory_stats(dev)
        return {"device_allocs": ms.get("num_device_alloc", 0), "device_frees": ms.get("num_device_free", 0), "alloc_retries": ms.get("num_alloc_retries", 0),
                "reserved_mb": ms.get("reserved_bytes.all.current", 0) / 2 ** 20, "gc_gen2": gc.get_stats()[2]["collections"],
                "gc_all": sum(g["collections"] for g in gc.get_stats())}

    def counters_delta(a, b):
        return {k: (round(b[k] - a[k], 1) if isinstance(b[k], float) else b[k] - a[k]) for k in a}

    def run_resident(n):
        nonlocal out
        mark()
        for _ in range(n):
            out = step()
            mark()

    out = None
    if world > 1:
        net.reducer.timing = True   # event pairs around every bucket's all-reduce on the communication stream (the `rccl` record)
    if args.warmup > 0:
        run_resident(args.warmup)   # untimed warm-up on the SAME path
    lib = _lib.load()
    if world > 1:
        net.reducer.comm_ms()   # drop the warm-up's records
    del step_marks[:], host_marks[:]
    c0 = runtime_counters()
    dt = timed(run_resident, args.steps)          # THE metric: K full steps, inputs resident in HBM when the timed region starts
    value_steps = step_stats(step_marks)
    value_steps["runtime"] = counters_delta(c0, runtime_counters())
    rccl = None
    if world > 1:
        red = net.reducer
        rccl = {"world": world, "backend": dist.get_backend(), "buckets": len(red.buckets), "bucket_mb": round(max(hi - lo for lo, hi, _ in red.buckets) * 4 / 2 ** 20, 1),
                "payload_mb_per_step": round(red.payload_bytes() / 2 ** 20, 1), "payload_dtype": "bf16" if red.grad_dtype is not None else "f32",
                "op": "AVG" if red.use_avg else "SUM+div",
                "tail_bucket_mb": round((red.buckets[-1][1] - red.buckets[-1][0]) * 4 / 2 ** 20, 1),
                "p8_wgrad_reserve_cus": ddp_defaults()["p8_wgrad_reserve_cus"], "q8_bwd_grid": ddp_defaults()["q8_bwd_grid"],
                "bucketwise_adamw": ddp_defaults()["bucketwise_adamw"],
                "allreduce_ms_per_step": round(red.comm_ms() / args.steps, 3),
                "channels": rccl_env_record(rccl_log),
                "adamw": "%d of %d optimizer steps ran bucket by bucket behind each bucket's all-reduce (ECAMP_BUCKETWISE_ADAMW; the rest: one pass "
                         "after the last all-reduce)" % (opt.bucketwise_steps, opt.steps_taken),
                "note": "all-reduce of the f32 gradient arena in buckets on a side HIP stream, overlapped with backward; ms = sum of the "
                        "buckets' event-bracketed durations on that stream on rank 0 (they overlap compute, so this is not added step time)"}
        red.timing = False
    if args.only_value:
        if rank == 0:
            print(json.dumps({"metric": METRIC, "value": round(args.batch * world * args.steps / dt, 2), "unit": "pairs/s", "n_gpus": world,
                              "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3), "rccl": rccl,
                              "step_ms": value_steps, "note": "--only-value (inputs resident): side measurements, roofline pass and CPU baseline skipped"}))
        if world > 1:
            dist.destroy_process_group()
        return

    # The PCIe-inclusive leg (never `value`): ONE prefetch pipeline across its own warm-up and timed steps, as in a training run: asking it
    # for a batch (after the previous step has been queued) issues the host->HBM copy of the batch two steps ahead on the copy stream, where
    # it runs beside the GPU's current work.  Each of the K timed steps issues exactly one batch copy and consumes one issued two steps
    # earlier (steady state); the 3 untimed steps in front prime all three staging slots.
    prefetcher = DevicePrefetcher([host_batch] * (3 + args.steps + 2), dev)
    prefetcher.timing = True
    pipeline = iter(prefetcher)

    def run_inclusive(n):
        nonlocal out
        mark()
        for _ in range(n):
            out = step(next(pipeline))
            mark()

    run_inclusive(3)
    torch.cuda.synchronize()
    prefetcher.copy_ms()   # drop the priming copies' records
    del step_marks[:], host_marks[:]
    c0 = runtime_counters()
    dt_host = timed(run_inclusive, args.steps)
    host_steps = step_stats(step_marks)
    host_steps["runtime"] = counters_delta(c0, runtime_counters())
    h2d_ms, h2d_bytes, h2d_n = prefetcher.copy_ms()
    del pipeline

    side_n = max(1, min(args.steps, 5))

    def run_fwd(n):
        with torch.no_grad():
            for _ in range(n):
                net(batch)

    def run_fwd_bwd(n):
        if hasattr(net, "set_grad_sync"):
            net.set_grad_sync(False)
        for _ in range(n):
            mim, res, mlm = net(batch)
            (mim + res + mlm).backward()
        if hasattr(net, "set_grad_sync"):
            net.set_grad_sync(True)
        opt.zero_grad()

    def run_vit(n):   # the image side alone: stem -> encoder -> decoder -> image losses, forward + backward (the north-star's 40 % scope)
        if hasattr(net, "set_grad_sync"):
            net.set_grad_sync(False)
        for _ in range(n):
            mim, res, _ = model(batch, image_side_only=True)
            (mim + res).backward()
        if hasattr(net, "set_grad_sync"):
            net.set_grad_sync(True)
        opt.zero_grad()

    dt_fwd = timed(run_fwd, side_n) / side_n
    dt_fb = timed(run_fwd_bwd, side_n) / side_n
    run_vit(1)
    dt_vit = timed(run_vit, side_n) / side_n
    losses = [float(t.detach()) for t in out[:3]]
    # Roofline pass (not part of `value`): the same step, with the weight-gradient GEMMs back on the main stream so that every
    # launch runs alone and its HIP-event duration is its own (in the timed region above they overlap the dgrad chain on a side
    # stream, which makes the step faster but per-launch durations meaningless).
    from ecamp_amd import hip_ops
    prof_steps = 0
    if not args.no_prof:  # every rank runs it (the steps contain collectives)
        wgrad, hip_ops.OVERLAP_WGRAD = hip_ops.OVERLAP_WGRAD, False
        branches, hip_ops.OVERLAP_BRANCHES = hip_ops.OVERLAP_BRANCHES, False   # one kernel at a time: no second stream of any kind
        try:   # whatever happens in between, the process leaves the serialized mode
            step()
            torch.cuda.synchronize()
            lib.ecamp_prof_collect(-1, None, None, None)
            lib.ecamp_prof_enable(1)
            prof_steps = min(args.steps, 3)
            tp = time.perf_counter()
            for _ in range(prof_steps):
                step()
            torch.cuda.synchronize()
            serial_ms = 1e3 * (time.perf_counter() - tp) / prof_steps
        finally:
            lib.ecamp_prof_enable(0)
            hip_ops.OVERLAP_WGRAD = wgrad
            hip_ops.OVERLAP_BRANCHES = branches
    if world > 1:
        t = torch.tensor([dt, dt_host, dt_fwd, dt_fb, dt_vit], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, dt_host, dt_fwd, dt_fb, dt_vit = (float(x) for x in t.tolist())

    if rank == 0:
        pairs = args.batch * world * args.steps
        res = {"metric": METRIC, "value": round(pairs / dt, 2), "unit": "pairs/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
               "config": {"workload": "BASELINE.json configs[1]: ViT-B/16 MAE enc/dec + SR head + reference BERT (6L/6H/1536, vocab 30000) "
                                      "+ context fusion; full train step (fwd+bwd+grad-norm+AdamW, dropout on)",
                          "pairs_per_gpu": args.batch, "global_batch": args.batch * world,
                          "image": "448^2 -> 224^2 encoder input" + (", uint8 grayscale crops normalised on the device" if args.image_u8 else ""),
                          "seq_len": args.seq, "mask_ratio": 0.75, "accum_iter": 1, "parallelism": "dp%d" % world,
                          "last_losses_mim_res_mlm": [round(x, 5) for x in losses]},
               **({"loss_scale": {"mode": "dynamic (GradScaler: init 65536, x2 after 2000 clean steps, x0.5 and a skipped step on inf / nan)",
                                  "scale": scaler.get_scale(), "skipped_steps": scaler.skipped_steps}} if scaler.dynamic else {}),
               "input": "resident in HBM when the timed region starts (`value`, `ms_per_step`, `step_ms`); the PCIe-inclusive rate of the same "
                        "step is `host_inclusive_*` (pinned host batch -> HBM inside that timed region: every timed step issues the copy of the batch "
                        "two steps ahead on a copy stream; steady-state pipeline primed by 3 untimed steps)",
               "step_ms": value_steps,
               "resident_pairs_per_s": round(pairs / dt, 2), "resident_ms_per_step": round(1e3 * dt / args.steps, 3),   # = value (kept for the A/B tools)
               "value_median_pairs_per_s": round(1e3 * args.batch * world / value_steps["median"], 2) if value_steps else None,
               "host_inclusive_pairs_per_s": round(pairs / dt_host, 2), "host_inclusive_ms_per_step": round(1e3 * dt_host / args.steps, 3),
               "host_inclusive_step_ms": host_steps,
               "h2d_ms_per_step": round(h2d_ms / max(h2d_n, 1), 3), "h2d_mb_per_step": round(h2d_bytes / max(h2d_n, 1) / 1e6, 1),
               "h2d_gbps": round(h2d_bytes / max(h2d_ms, 1e-9) / 1e6, 2),
               "h2d_gbps_needed": round(h2d_bytes / max(h2d_n, 1) / (dt / args.steps) / 1e9, 2),
               "h2d_note": "event pairs around each batch's copies on the copy stream (rank 0); the copy of a batch hides under a step while "
                           "h2d_gbps > h2d_gbps_needed (bytes per batch / resident step time); below that the box's PCIe path, not the kernels, sets "
                           "host_inclusive_*",
               "fwd_only_ms": round(1e3 * dt_fwd, 3), "fwd_only_pairs_per_s": round(args.batch * world / dt_fwd, 2),
               "fwd_bwd_ms": round(1e3 * dt_fb, 3), "fwd_bwd_pairs_per_s": round(args.batch * world / dt_fb, 2)}
        # the north-star's ">= 40 % MFMA roofline on the ViT-B/16 forward+backward at bs=256/GPU" in its own scope: the image side alone
        # (model_ecamp.py:218-264,276-300), 7.154 GMAC forward per image x 2 FLOP x 3 (fwd + dgrad + wgrad) = 42.92 GFLOP per image
        # (SURVEY.md 8(d)); production stream layout (weight gradients on the side stream), inputs resident
        if args.seq in (128, 256) and half16:
            vit_tf = 42.92e9 * args.batch / dt_vit / 1e12
            res.update({"vit_fwd_bwd_ms": round(1e3 * dt_vit, 3), "vit_tflops": round(vit_tf, 2), "vit_frac": round(vit_tf / PEAK_BF16_TFLOPS, 4),
                        "vit_note": "image side only (stem, 12 encoder blocks, decoder, SR head, image losses), forward + backward, per GPU: "
                                    "42.92 GFLOP per image (SURVEY.md 8(d)) x pairs_per_gpu / vit_fwd_bwd_ms against the 2.5 PF dense bf16 peak"})
        if rccl is not None:
            res["rccl"] = rccl
        if not args.no_prof:
            ms, fl, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
            cat = 0 if half16 else 1
            lib.ecamp_prof_collect(cat, ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(n))
            ams, afl, an = ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
            lib.ecamp_prof_collect(2, ctypes.byref(ams), ctypes.byref(afl), ctypes.byref(an))
            ach = fl.value / (ms.value * 1e-3) / 1e12 if ms.value > 0 else 0.0
            peak = PEAK_BF16_TFLOPS if half16 else 157.3   # the dense f16 peak equals the bf16 one
            # HBM-side traffic per GEMM launch: not measurable from inside this process -- taken from the newest committed PMC passes of
            # this very command (profiles/rNN_pmc_traffic.json: FETCH_SIZE x2 per the gfx950 correction + WRITE_SIZE), and only when that
            # file is stamped with the hash of the GEMM sources this library was built from; otherwise null (never last round's number)
            traffic, traffic_src = None, None
            if args.dtype == "bf16" and args.batch == 256 and args.seq == 128:
                import glob
                from ecamp_amd.build import gemm_source_hash
                cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_traffic.json")))
                if cands:
                    tj = cands[-1]
                    try:
                        rec = json.load(open(tj))
                        if rec.get("gemm_source_sha256") == gemm_source_hash():
                            traffic = round(rec["gemm_traffic_bytes_per_launch"])
                            traffic_src = "profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, bytes per GEMM launch; stamped with this build's GEMM source hash)" % os.path.basename(tj)
                        else:
                            traffic_src = ("profiles/%s was collected on other GEMM sources (hash mismatch): not quoted -- re-run tools/pmc_traffic.sh"
                                           % os.path.basename(tj))
                    except Exception:
                        traffic = None
            res["roofline"] = {"bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                               "traffic": traffic, "traffic_source": traffic_src,
                               "kernel": "gemm_bf16_q8_kernel + gemm_bf16_q16_kernel + gemm_bf16_kernel (bf16 GEMM family)" if args.dtype == "bf16" else "the same three kernels built for IEEE half (libecamp_hip_f16.so)" if half16 else "gemm_f32_kernel",
                               "launches_per_step": n.value // max(prof_steps, 1),
                               "avg_launch_us": round(1e3 * ms.value / max(n.value, 1), 2),
                               "algorithmic_gflop_per_launch": round(fl.value / max(n.value, 1) / 1e9, 3),
                               "gemm_ms_per_step": round(ms.value / prof_steps, 3),
                               "attention_ms_per_step": round(ams.value / prof_steps, 3),
                               "serialized_ms_per_step": round(serial_ms, 3),
                               "note": "achieved = sum(2MNK) over every GEMM launch / sum of their HIP-event durations, taken in a "
                                       "serialized pass of the same step (%d steps, wgrad GEMMs and the image-decoder branch on the main stream); `value` is timed "
                                       "on the production path where wgrad GEMMs overlap the dgrad chain on a side stream and the image decoder runs "
                                       "beside the report side on a branch stream" % prof_steps}
            # whole-step view with SURVEY.md 8(d)'s algorithmic FLOPs per pair
            gflop_pair = 88.99 if args.seq == 128 else 136.64
            res["roofline"]["whole_step_tflops"] = round(gflop_pair * 1e9 * args.batch / (dt / args.steps) / 1e12, 2)
        if not args.no_cpu_baseline and world == 1:  # the CPU leg is reported at N=1 only; at N>1 the other ranks would sit waiting for it
            try:
                res["cpu_baseline"] = cpu_baseline(args.seq)
            except Exception as e:  # the baseline is reporting only; never lose the GPU number over it
                res["cpu_baseline"] = {"value": None, "unit": "pairs/s", "cores": os.cpu_count(), "kind": "port", "sample": "failed: %r" % (e,)}
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
