"""The data-parallel exchange over RCCL with as many ranks as this box has GPUs (reference:
ECAMP/Pre-training/main_pretrain.py:247-250 DDP wrap, util/misc.py:242-247 all_reduce_mean).

On a box with >= 2 visible GPUs the tests ARM THEMSELVES with two ranks, one GPU each -- the production path of
ecamp_amd/parallel.py (in-place all-reduce of arena slices on a side stream), the switches that only matter at N > 1
(ECAMP_BUCKETWISE_ADAMW, ECAMP_DDP_Q8_BWD_GRID: off by default until this file has passed with two ranks) and `bench.py --gpus 2`.
On the one-GPU boxes of this pool the same worker runs as ONE rank with the collectives forced on: that proves the plumbing (fresh
child process, RCCL group, streams, events, the scaler's two optimizer paths), not the exchange.  The ranks are fresh child processes
(never an exec from this GPU-initialised process: subprocess forks first)."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_GPUS = torch.cuda.device_count()          # counting devices does not initialise the GPU
WORLD = 2 if N_GPUS >= 2 else 1


def _free_port():
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    return port


def _run_ranks(tmp_path, tag, world=WORLD, **env):
    out = os.path.join(tmp_path, "rccl_%s.pt" % tag)
    port = _free_port()
    e = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), **{k: str(v) for k, v in env.items()})
    worker = os.path.join(ROOT, "tests", "_rccl_worker.py")
    procs = [subprocess.Popen([sys.executable, worker, str(r), str(world), str(port), out], cwd=ROOT, env=e) for r in range(world)]
    import time
    t_end = time.time() + 900
    try:   # poll ALL ranks: when one dies the others sit in an RCCL collective for ever -- stop them at once instead of after the timeout
        while True:
            rcs = [p.poll() for p in procs]
            if all(rc is not None for rc in rcs) or any(rc not in (None, 0) for rc in rcs) or time.time() > t_end:
                break
            time.sleep(0.2)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
                p.wait()
    assert rcs == [0] * world, "rank exit codes %s (None = still running when another rank failed or the 900 s limit passed)" % (rcs,)
    return torch.load(out, map_location="cpu")


@pytest.fixture(scope="module")
def single(dev):
    """ONE process on the whole B=8 recipe batch: gradient arena, norm and parameters after one AdamW step."""
    from ecamp_amd import optim
    from ecamp_amd.module import model_ecamp as me
    from ecamp_amd.util import misc
    from oracle import ecamp_oracle as orc
    from oracle import recipe
    cfg = orc.cfg_tiny()
    batch = recipe.recipe_batch(cfg, 8, 64, seed=5)
    noise = recipe.recipe_noise(8, cfg.num_patches, seed=5)
    model = me.ecamp_tiny(compute_dtype=torch.float32)
    model.load_state_dict(recipe.recipe_state(cfg, seed=0))
    model.to(dev).eval()
    model.prepare()
    opt = optim.FusedAdamW(optim.add_weight_decay(model, 0.05), lr=1e-3, betas=(0.9, 0.95))
    opt.zero_grad()
    sum(model(batch, noise=noise)).backward()
    model.arena.flush_fresh()
    torch.cuda.synchronize()
    g = model.arena.flat_g.detach().cpu().clone()
    norm = float(misc.get_grad_norm_(model.parameters()))
    opt.step()
    torch.cuda.synchronize()
    return {"flat_g": g, "flat_p": model.arena.flat_p.detach().cpu().clone(), "norm": norm}


@pytest.mark.gpu
@pytest.mark.parametrize("bucketwise,grid", [(0, 0), (1, 0), (1, 1 << 20)])
def test_rccl_ranks_equal_single_process(dev, tmp_path, single, bucketwise, grid):
    """(i) WORLD ranks of B=8/WORLD leave the single-process gradient arena (<= 1e-5) and post-step parameters; every rank holds the
    same arena; (ii) the optimizer step behind the all-reduces -- one pass, or bucket by bucket (ECAMP_BUCKETWISE_ADAMW=1), with the
    data-gradient GEMMs one tile per workgroup (ECAMP_DDP_Q8_BWD_GRID) -- equals a one-pass AdamW on the same reduced arena BIT FOR BIT."""
    got = _run_ranks(tmp_path, "f32_%d_%d" % (bucketwise, grid), ECAMP_BUCKETWISE_ADAMW=bucketwise, ECAMP_DDP_Q8_BWD_GRID=grid)
    assert got["world"] == WORLD and got["backend"] == "nccl" and got["buckets"] > 8
    eg = float((got["flat_g"] - single["flat_g"]).abs().max() / single["flat_g"].abs().max())
    ep = float((got["flat_p"] - single["flat_p"]).abs().max() / single["flat_p"].abs().max())
    print("  RCCL %d rank(s), bucketwise %d, q8_bwd_grid %d: grad arena rel %.2e, params rel %.2e, norm %.6f vs %.6f, bucketwise steps %d"
          % (WORLD, bucketwise, grid, eg, ep, got["norm"], single["norm"], got["bucketwise_steps"]))
    assert got["ranks_agree"]
    assert got["replay_equal"], "the update behind the all-reduces is not the one-pass AdamW on the same gradients"
    assert got["bucketwise_steps"] == bucketwise
    assert eg <= 1e-5, eg
    assert ep <= 1e-4, ep     # Adam's first step is lr * sign-like: compared at lr resolution (see test_ddp_two_ranks_equal_single_process)
    # (one pass: the fused kernel's own f32 sum of squares; bucket by bucket: one f32 partial per bucket launch, summed in another order)
    assert abs(got["norm"] - single["norm"]) <= (1e-3 if bucketwise else 1e-5) * single["norm"]
    assert abs(got["replay_norm"] - got["norm"]) <= 1e-3 * got["norm"]


@pytest.mark.gpu
def test_rccl_bf16_exchange_within_rounding(dev, tmp_path, single):
    """(iii) the optional bf16 payload (half the bytes on the links): the arena agrees with the f32 single-process one to bf16 rounding."""
    got = _run_ranks(tmp_path, "bf16", ECAMP_DDP_GRAD_DTYPE="bf16")
    eg = float((got["flat_g"] - single["flat_g"]).abs().max() / single["flat_g"].abs().max())
    print("  RCCL %d rank(s), bf16 exchange: grad arena rel %.2e" % (WORLD, eg))
    assert got["ranks_agree"] and got["replay_equal"]
    assert 1e-6 < eg <= 1e-2, eg
    assert abs(got["norm"] - single["norm"]) <= 5e-3 * single["norm"]


@pytest.mark.gpu
@pytest.mark.skipif(N_GPUS < 2, reason="needs two visible GPUs (this pool has one per box): arms itself where they exist")
def test_bench_two_gpus_prints_an_rccl_record(dev):
    """(iv) `python bench.py --gpus 2` self-launches two ranks and its JSON line carries the `rccl` record of the exchange."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2", "--no-cpu-baseline"],
                       cwd=ROOT, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-2000:]
    rec = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert rec["n_gpus"] == 2 and rec["scaling"] == "weak" and rec["config"]["global_batch"] == 512
    rc = rec["rccl"]
    assert rc["world"] == 2 and rc["backend"] == "nccl" and rc["allreduce_ms_per_step"] > 0
    assert rc["channels"].get("channels_reported") is not None, rc["channels"]
    print("  bench --gpus 2: %.0f pairs/s, %.2f ms/step, all-reduce %.2f ms/step on %s channels"
          % (rec["value"], rec["ms_per_step"], rc["allreduce_ms_per_step"], rc["channels"]["channels_reported"]))
