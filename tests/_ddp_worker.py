"""Child process of tests/test_engine_gpu.py::test_ddp_two_ranks_equal_single_process: one data-parallel rank of TWO sharing cuda:0
(gloo process group on device tensors -- RCCL refuses duplicate devices; the reducer stages its buckets through the host then).
Runs `accum` micro-steps of B/(2*accum) pairs each (no_sync on all but the last), one optimizer step, and saves the reduced gradient
arena and the updated parameters."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, port, accum, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), sys.argv[5]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ecamp_amd import optim
    from ecamp_amd.module import model_ecamp as me
    from ecamp_amd.parallel import DistributedDataParallel
    from ecamp_amd.util.misc import NativeScalerWithGradNormCount
    from oracle import ecamp_oracle as orc
    from oracle import recipe
    dev = torch.device("cuda:0")
    cfg = orc.cfg_tiny()
    B, S = 8, 64
    state = recipe.recipe_state(cfg, seed=0)
    batch = recipe.recipe_batch(cfg, B, S, seed=5)
    noise = recipe.recipe_noise(B, cfg.num_patches, seed=5)
    model = me.ecamp_tiny(compute_dtype=torch.float32)
    if rank == 0:
        model.load_state_dict(state)   # rank 1 keeps its random init: the wrapper's broadcast must overwrite it
    model.to(dev).eval()               # dropout off (its streams differ per rank); gradients still flow
    net = DistributedDataParallel(model, grad_dtype=torch.bfloat16 if os.environ.get("ECAMP_DDP_GRAD_DTYPE") == "bf16" else None)
    opt = optim.FusedAdamW(optim.add_weight_decay(model, 0.05), lr=1e-3, betas=(0.9, 0.95))
    dyn = os.environ.get("ECAMP_TEST_LOSS_SCALE") == "dynamic"    # GradScaler semantics, decided on the device, behind the all-reduce
    scaler = NativeScalerWithGradNormCount(dynamic=dyn)
    per = B // world
    mb = per // accum
    opt.zero_grad()
    for a in range(accum):
        lo = rank * per + a * mb
        sub = {k: v[lo:lo + mb] for k, v in batch.items()}
        net.set_grad_sync(a == accum - 1)
        mim, res, mlm = net(sub, noise=noise[lo:lo + mb])
        loss = (mim + res + mlm) / accum
        if a < accum - 1:
            scaler(loss, opt, parameters=model.parameters(), update_grad=False)
        elif dyn:
            norm = scaler(loss, opt, parameters=model.parameters(), update_grad=True)
            assert scaler.last_step_fused and scaler.skipped_steps == 0 and opt.steps_taken == 1
            torch.cuda.synchronize()
            flat_g = model.arena.flat_g.detach().cpu().clone() / 65536.0     # p.grad keeps the scaled values until zero_grad()
        else:
            loss.backward()
            arena = model.arena
            arena.flush_fresh()
            torch.cuda.synchronize()
            flat_g = arena.flat_g.detach().cpu().clone()
            from ecamp_amd.util import misc
            norm = misc.get_grad_norm_(model.parameters())
            opt.step()
    torch.cuda.synchronize()
    if rank == 0:
        torch.save({"flat_g": flat_g, "flat_p": model.arena.flat_p.detach().cpu(), "norm": float(norm)}, out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
