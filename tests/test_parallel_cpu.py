"""not gpu: the N>1 path on CPU -- 2 ranks over gloo.  The bucketed gradient reducer must average the flat gradient
arena exactly like dist.all_reduce / world, whatever order the backward stages report their parameters in, must not
wait for statically-unused parameters, and must skip communication on accumulation micro-steps."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ecamp_amd.parallel import GradReducer
        sizes = [64 * 3, 64, 64 * 10, 64 * 2, 64 * 7, 64]
        offs = [sum(sizes[:i]) for i in range(len(sizes))]
        n = sum(sizes)
        g = torch.arange(n, dtype=torch.float32) * (rank + 1)
        expect = torch.arange(n, dtype=torch.float32) * (sum(range(1, world + 1)) / world)
        red = GradReducer(g, offs, sizes, unused=[3], bucket_mb=64 * 8 * 4 / 2 ** 20)  # ~512-element buckets
        assert len(red.buckets) >= 2 and sum(hi - lo for lo, hi, _ in red.buckets) == n
        # backward order: last registered first, in two calls; slot 3 is "unused" and never reported
        red.mark_ready([5, 4])
        red.mark_ready([2, 1, 0])
        red.finalize()
        ok1 = torch.allclose(g, expect)
        # accumulation micro-step: no communication
        g2 = torch.full((n,), float(rank + 1))
        red2 = GradReducer(g2, offs, sizes, unused=[3], bucket_mb=1.0)
        red2.enabled = False
        red2.mark_ready([5, 4, 2, 1, 0])
        red2.finalize()
        ok2 = bool((g2 == rank + 1).all())
        red2.enabled = True
        red2.mark_ready([0, 1, 2, 4, 5])  # arbitrary order
        red2.finalize()
        ok3 = torch.allclose(g2, torch.full((n,), sum(range(1, world + 1)) / world))
        # bf16 gradient exchange (grad_dtype=torch.bfloat16): a bucket travels as bf16 (half the bytes) and lands back in the f32 arena
        # within bf16 rounding of the exact mean (contract: 1e-2 relative); the f32 default stays EXACTLY today's in-place path
        torch.manual_seed(3)
        base = torch.randn(n) * 3.0
        g3 = base * (rank + 1)
        g3_f32 = g3.clone()
        red3 = GradReducer(g3, offs, sizes, unused=[3], bucket_mb=64 * 8 * 4 / 2 ** 20, grad_dtype=torch.bfloat16)
        red3.mark_ready([5, 4, 2, 1, 0])
        red3.finalize()
        red3f = GradReducer(g3_f32, offs, sizes, unused=[3], bucket_mb=64 * 8 * 4 / 2 ** 20, grad_dtype=torch.float32)
        assert red3f.grad_dtype is None and red3f.payload_bytes() == 2 * red3.payload_bytes()
        red3f.mark_ready([5, 4, 2, 1, 0])
        red3f.finalize()
        exact = base * (sum(range(1, world + 1)) / world)
        ok5 = bool(torch.equal(g3_f32, exact)) and bool(((g3 - exact).norm() / exact.norm()) < 1e-2) and not torch.equal(g3, g3_f32) \
            and g3.dtype == torch.float32 and red3.pack.dtype == torch.bfloat16
        # lazy logging all-reduce used by train_one_epoch
        from ecamp_amd.util import misc
        r = misc.all_reduce_mean(torch.tensor([1.0 * rank, 2.0, 3.0]))
        ok4 = torch.allclose(r, torch.tensor([(world - 1) / 2.0, 2.0, 3.0]))
        q.put((rank, ok1, ok2, ok3, ok4, ok5))
    finally:
        dist.destroy_process_group()


def test_bucketed_reducer_two_ranks_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in ps:
        p.join(timeout=60)
    assert sorted(r[0] for r in res) == [0, 1]
    for r in res:
        assert all(r[1:]), r


def test_bucket_plan_closes_before_a_large_tensor_and_keeps_the_tail_small():
    """The bucket plan alone (no process group): buckets tile the arena without gaps, in reverse registration order; a tensor larger
    than the cap gets a bucket of its own instead of holding the small parameters registered after it; the parameters registered
    first (finished last by backward) sit in buckets of at most the tail cap (or one tensor)."""
    from ecamp_amd.parallel import GradReducer
    mib = 2 ** 20 // 4
    sizes = [mib // 2, mib // 2, mib, mib, 3 * mib, 3 * mib, 40 * mib, mib // 4, mib // 4, 90 * mib, mib // 8, mib // 8]
    offs = [sum(sizes[:i]) for i in range(len(sizes))]
    class _Flat:   # only numel() / element_size() / is_cuda are read by the plan
        is_cuda = False
        def numel(self): return sum(sizes)
        def element_size(self): return 4
    red = GradReducer(_Flat(), offs, sizes, unused=[7], bucket_mb=32.0, tail_bucket_mb=2.0, tail_span_mb=4.0)
    # tiling, order
    assert red.buckets[0][1] == sum(sizes) and red.buckets[-1][0] == 0
    for (lo, hi, sl), (lo2, hi2, sl2) in zip(red.buckets, red.buckets[1:]):
        assert lo == hi2 and lo < hi and min(sl) > max(sl2)
    assert sorted(s for _, _, sl in red.buckets for s in sl) == list(range(len(sizes)))
    by_slot = {s: (lo, hi, sl) for lo, hi, sl in red.buckets for s in sl}
    assert by_slot[9][2] == [9] and by_slot[11][2] == [11, 10]       # the 90 MiB tensor alone; the two small ones after it together
    assert by_slot[6][2] == [6]                                      # 40 MiB > cap: alone as well
    assert by_slot[8][2] == [8, 7]
    # slots 0..3 start inside the first 4 MiB of the arena: tail buckets of <= 2 MiB
    for s in (0, 1, 2, 3):
        lo, hi, sl = by_slot[s]
        assert hi - lo <= 2 * mib
    assert by_slot[0][2] == [1, 0] and by_slot[2][2] == [3, 2]
    assert by_slot[4][2] == [4] and by_slot[5][2] == [5]             # slot 4 starts inside the tail span: it does not join slot 5
    assert red.pending == [sum(1 for s in sl if s != 7) for _, _, sl in red.buckets]
    # the old call (no tail arguments) still gives plain capped buckets
    red2 = GradReducer(_Flat(), offs, sizes, bucket_mb=64.0)
    assert sum(hi - lo for lo, hi, _ in red2.buckets) == sum(sizes) and all(hi - lo <= 64 * mib or len(sl) == 1 for lo, hi, sl in red2.buckets)


def test_rccl_env_defaults_cap_the_channels_and_respect_explicit_settings():
    """bench.py / init_distributed_mode setdefault() RCCL's channel cap (below the co-tenancy cliff of
    profiles/r03_cotenant_ab.txt); a user's explicit NCCL_MAX_NCHANNELS wins; the `rccl` record reports what is in force and parses the
    channel count out of an RCCL INFO log when there is one."""
    import tempfile
    from ecamp_amd.parallel import RCCL_ENV_DEFAULTS, rccl_env_defaults, rccl_env_record
    env = rccl_env_defaults({})
    assert env["NCCL_MAX_NCHANNELS"] == RCCL_ENV_DEFAULTS["NCCL_MAX_NCHANNELS"] and 0 < int(env["NCCL_MAX_NCHANNELS"]) < 96
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert rccl_env_defaults({"NCCL_MAX_NCHANNELS": "8"})["NCCL_MAX_NCHANNELS"] == "8"
    with tempfile.NamedTemporaryFile("w", suffix=".log", delete=False) as f:
        f.write("host:1:1 [0] NCCL INFO Channel 00/24 : 0 1 2 3\nhost:1:1 [0] NCCL INFO 24 coll channels, 24 collnet channels, 0 nvls channels, 32 p2p channels\n")
    old = os.environ.get("NCCL_MAX_NCHANNELS")
    os.environ["NCCL_MAX_NCHANNELS"] = "32"
    try:
        rec = rccl_env_record(f.name)
    finally:
        if old is None:
            del os.environ["NCCL_MAX_NCHANNELS"]
        else:
            os.environ["NCCL_MAX_NCHANNELS"] = old
        os.unlink(f.name)
    assert rec["channels_reported"] == 24 and rec["max_channels_cap"] == 32 and rec["env"]["NCCL_MAX_NCHANNELS"] == "32" and rec["channel_lines"]
