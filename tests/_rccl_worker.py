"""Child process of tests/test_rccl_gpu.py: one data-parallel rank over RCCL (backend "nccl"), one GPU per rank -- the PRODUCTION
exchange path of ecamp_amd/parallel.py (in-place all-reduce of arena slices on the communication stream, events back to the compute
stream), which the gloo-on-one-GPU test (tests/_ddp_worker.py) does not take.  Replaces the reference's torch DDP wrap
(ECAMP/Pre-training/main_pretrain.py:247-250) and its per-step gradient all-reduce (util/misc.py:242-247).

    _rccl_worker.py <rank> <world> <port> <out.pt>        env: ECAMP_BUCKETWISE_ADAMW, ECAMP_DDP_Q8_BWD_GRID, ECAMP_DDP_GRAD_DTYPE

One optimizer step through the loss scaler on the rank's B/world pairs of the recipe batch (tiny config, f32 parity mode, eval so that
no rank-dependent dropout stream enters).  Rank 0 saves the reduced gradient arena, the updated parameters and moments, the gradient
norm, and `replay_equal`: whether a ONE-PASS AdamW from the saved pre-step state on the same reduced arena reproduces parameters,
moments and bf16 shadow bit for bit (the check that the bucket-by-bucket update behind the all-reduces is the same update)."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    from ecamp_amd.parallel import DistributedDataParallel, rccl_env_defaults
    rccl_env_defaults()
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    import datetime
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=datetime.timedelta(seconds=300))
    from ecamp_amd import hip_ops, optim
    from ecamp_amd.module import model_ecamp as me
    from ecamp_amd.util.misc import NativeScalerWithGradNormCount
    from oracle import ecamp_oracle as orc
    from oracle import recipe
    cfg = orc.cfg_tiny()
    B, S = 8, 64
    state = recipe.recipe_state(cfg, seed=0)
    batch = recipe.recipe_batch(cfg, B, S, seed=5)
    noise = recipe.recipe_noise(B, cfg.num_patches, seed=5)
    model = me.ecamp_tiny(compute_dtype=torch.float32)
    if rank == 0:
        model.load_state_dict(state)   # the other ranks keep their random init: the wrapper's broadcast must overwrite it
    model.to(dev).eval()
    A = model.prepare()
    net = DistributedDataParallel(model, bucket_cap_mb=2.0, tail_bucket_mb=0.5, tail_span_mb=2.0, force_comm=(world == 1),
                                  grad_dtype=torch.bfloat16 if os.environ.get("ECAMP_DDP_GRAD_DTYPE") == "bf16" else None)
    opt = optim.FusedAdamW(optim.add_weight_decay(model, 0.05), lr=1e-3, betas=(0.9, 0.95))
    scaler = NativeScalerWithGradNormCount()
    opt.zero_grad()
    opt._bind()
    per = B // world
    lo = rank * per
    sub = {k: v[lo:lo + per] for k, v in batch.items()}
    before = (A.flat_p.clone(), opt._m.clone(), opt._v.clone())
    mim, res, mlm = net(sub, noise=noise[lo:lo + per])
    norm = scaler(mim + res + mlm, opt, parameters=model.parameters(), update_grad=True)
    torch.cuda.synchronize()
    # the update does not modify the gradient arena: replay it in one pass from the saved state
    p, m, v = before
    s = torch.zeros(1, device=dev)
    p16 = torch.empty_like(A.flat_p16) if A.flat_p16 is not None else None   # (f32 parity mode keeps no bf16 shadow)
    g0 = opt.param_groups[0]
    hip_ops.adamw_grouped(p, A.flat_g, m, v, p16, opt._table, [g["lr"] for g in opt.param_groups], [g["weight_decay"] for g in opt.param_groups],
                          g0["betas"][0], g0["betas"][1], g0["eps"], 1, 1.0, s)
    torch.cuda.synchronize()
    used = opt._table.repeat_interleave(64) < 8
    replay_equal = bool(torch.equal(p, A.flat_p) and torch.equal(m, opt._m) and torch.equal(v, opt._v)
                        and (p16 is None or torch.equal(p16[used], A.flat_p16[used])))
    # every rank must hold the same reduced arena and the same parameters
    chk = torch.stack([A.flat_g.double().sum(), A.flat_p.double().sum()])
    lo_, hi_ = chk.clone(), chk.clone()
    dist.all_reduce(lo_, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi_, op=dist.ReduceOp.MAX)
    if rank == 0:
        torch.save({"flat_g": A.flat_g.detach().cpu(), "flat_p": A.flat_p.detach().cpu(), "m": opt._m.detach().cpu(), "norm": float(norm),
                    "replay_equal": replay_equal, "replay_norm": float(s.sqrt()), "bucketwise_steps": int(opt.bucketwise_steps),
                    "buckets": len(net.reducer.buckets), "ranks_agree": bool(torch.equal(lo_, hi_)), "world": world,
                    "backend": dist.get_backend()}, out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
