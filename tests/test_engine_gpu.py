"""-m gpu: the driver level -- BASELINE.json configs[0] (ViT-Tiny/16 + 2-layer BERT, 32 synthetic 448^2 image-report pairs,
mask_ratio 0.75, 1+ epochs) through `ecamp_amd.main_pretrain` with the reference's command-line flags, checkpoint format
round trip and optimizer-state compatibility with torch.optim.AdamW."""
import argparse
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _args(tmp, extra=()):
    from ecamp_amd.main_pretrain import get_args_parser
    argv = ["--model", "ecamp_tiny", "--batch_size", "8", "--accum_iter", "2", "--epochs", "3", "--warmup_epochs", "1", "--max_epoch", "4",
            "--lr", "5e-4", "--weight_decay", "0.05", "--mask_ratio", "0.75", "--norm_pix_loss", "--num_workers", "0",
            "--output_dir", str(tmp), "--data_path", str(tmp), "--synthetic", "--synthetic_len", "32", "--max_caption_length", "64", "--print_freq", "2",
            "--compute_dtype", "bf16"] + list(extra)
    return argparse.ArgumentParser(parents=[get_args_parser()]).parse_args(argv)


def test_main_pretrain_tiny_three_epochs(dev, tmp_path):
    from ecamp_amd import main_pretrain
    args = _args(tmp_path)
    main_pretrain.main(args)
    lines = open(os.path.join(tmp_path, "log.txt")).read().strip().split("\n")
    assert lines[0] == "ecamp_pretrain"
    stats = [json.loads(l) for l in lines[1:]]
    assert [s["epoch"] for s in stats] == [0, 1, 2]
    for s in stats:
        for k in ("train_mim_loss", "train_res_loss", "train_mlm_loss", "train_lr"):
            assert k in s and s[k] == s[k]  # present and not NaN
    assert stats[-1]["train_mlm_loss"] < stats[0]["train_mlm_loss"], "MLM loss should fall within 3 epochs on 32 memorisable pairs"
    assert stats[-1]["train_mim_loss"] < stats[0]["train_mim_loss"]
    ck = torch.load(os.path.join(tmp_path, "checkpoint-0.pth"), map_location="cpu", weights_only=False)  # cadence: epoch 0 is saved
    assert set(ck.keys()) == {"model", "optimizer", "epoch", "scaler", "args"} and ck["epoch"] == 0
    assert os.path.exists(os.path.join(tmp_path, "config.yaml"))


def test_main_pretrain_with_uint8_images(dev, tmp_path):
    """`--image_u8`: the synthetic dataset hands over uint8 grayscale crops, the DataLoader / pinned memory / prefetcher carry one byte
    per pixel, and the epoch trains like the f32 schema (finite, falling losses)."""
    from ecamp_amd import main_pretrain
    args = _args(tmp_path, ["--image_u8", "--epochs", "2"])
    main_pretrain.main(args)
    lines = open(os.path.join(tmp_path, "log.txt")).read().strip().split("\n")
    stats = [json.loads(l) for l in lines[1:]]
    assert [s["epoch"] for s in stats] == [0, 1]
    assert all(s[k] == s[k] for s in stats for k in ("train_mim_loss", "train_res_loss", "train_mlm_loss"))
    assert stats[-1]["train_mlm_loss"] < stats[0]["train_mlm_loss"]


def test_main_pretrain_amp_fp16(dev, tmp_path):
    """`--amp fp16` through the command line: IEEE-half activations (libecamp_hip_f16.so) + the dynamic loss scaler, accumulation 2,
    checkpoints whose `scaler` entry is GradScaler's state -- losses finite and falling like the bf16 run's, no step skipped at 65536."""
    from ecamp_amd import _lib, main_pretrain
    args = _args(tmp_path, ["--amp", "fp16", "--epochs", "2"])
    try:
        main_pretrain.main(args)
        assert _lib.half() == "f16" and args.compute_dtype == "fp16" and args.loss_scale == "dynamic"
    finally:
        _lib.set_half("bf16")
    lines = open(os.path.join(tmp_path, "log.txt")).read().strip().split("\n")
    stats = [json.loads(l) for l in lines[1:]]
    assert [s["epoch"] for s in stats] == [0, 1]
    assert all(s[k] == s[k] for s in stats for k in ("train_mim_loss", "train_res_loss", "train_mlm_loss"))
    assert stats[-1]["train_mlm_loss"] < stats[0]["train_mlm_loss"] and stats[-1]["train_mim_loss"] < stats[0]["train_mim_loss"]
    ck = torch.load(os.path.join(tmp_path, "checkpoint-0.pth"), map_location="cpu", weights_only=False)
    assert ck["scaler"]["scale"] == 65536.0 and ck["scaler"]["_growth_tracker"] == 2 and ck["scaler"]["growth_interval"] == 2000
    assert all(v.dtype == torch.float32 for v in ck["model"].values() if v.is_floating_point())     # masters are f32 in every mode
    assert float(ck["optimizer"]["state"][0]["step"]) == 2.0


def test_checkpoint_round_trip_and_torch_adamw_compat(dev, tmp_path):
    from ecamp_amd import optim
    from ecamp_amd.data import synthetic_batch
    from ecamp_amd.module import model_ecamp as me
    from ecamp_amd.util import misc
    torch.manual_seed(1)
    model = me.ecamp_tiny(compute_dtype=torch.float32).to(dev)
    model.prepare()
    opt = optim.FusedAdamW(optim.add_weight_decay(model, 0.05), lr=1e-3, betas=(0.9, 0.95))
    scaler = misc.NativeScalerWithGradNormCount()
    batch = synthetic_batch(4, 32, 448, seed=3)
    model.eval()
    opt.zero_grad()
    scaler(sum(model(batch)), opt, parameters=model.parameters(), update_grad=True)
    args = argparse.Namespace(output_dir=str(tmp_path), resume="")
    misc.save_model(args=args, epoch=7, model=model, model_without_ddp=model, optimizer=opt, loss_scaler=scaler)
    path = os.path.join(tmp_path, "checkpoint-7.pth")
    ck = torch.load(path, map_location="cpu", weights_only=False)
    # (a) the optimizer entry is a valid torch.optim.AdamW state dict for the same parameter grouping
    twin = me.ecamp_tiny(compute_dtype=torch.float32)
    twin.load_state_dict(ck["model"])
    topt = torch.optim.AdamW(optim.add_weight_decay(twin, 0.05), lr=1e-3, betas=(0.9, 0.95))
    topt.load_state_dict(ck["optimizer"])
    st = topt.state[twin.blocks[0].attn.qkv.weight]
    assert float(st["step"]) == 1.0 and st["exp_avg"].abs().sum() > 0
    # (b) resume into a fresh HIP model + optimizer: the next step is identical to continuing the original
    m2 = me.ecamp_tiny(compute_dtype=torch.float32).to(dev)
    m2.prepare()
    o2 = optim.FusedAdamW(optim.add_weight_decay(m2, 0.05), lr=1e-3, betas=(0.9, 0.95))
    args2 = argparse.Namespace(resume="./ECAMP_ckpt.pth", start_epoch=0)
    os.symlink(path, "./ECAMP_ckpt.pth") if not os.path.exists("./ECAMP_ckpt.pth") else None
    try:
        misc.load_model(args=args2, model_without_ddp=m2, optimizer=o2, loss_scaler=misc.NativeScalerWithGradNormCount())
    finally:
        os.remove("./ECAMP_ckpt.pth")
    assert args2.start_epoch == 8
    m2.eval()
    for mdl, op in ((model, opt), (m2, o2)):
        op.zero_grad()
        scaler(sum(mdl(batch, noise=torch.linspace(0, 1, 196).repeat(4, 1))), op, parameters=mdl.parameters(), update_grad=True)
    a, b = model.blocks[3].mlp.fc1.weight.detach().float().cpu(), m2.blocks[3].mlp.fc1.weight.detach().float().cpu()
    assert torch.allclose(a, b, rtol=1e-5, atol=1e-7)
    # (c) a plain MAE encoder checkpoint (subset of keys) loads by key intersection (misc.py:322-329)
    mae = {k: v for k, v in ck["model"].items() if k.startswith(("patch_embed", "blocks", "norm", "cls_token", "pos_embed"))}
    torch.save({"model": mae}, os.path.join(tmp_path, "mae.pth"))
    m3 = me.ecamp_tiny(compute_dtype=torch.float32).to(dev)
    m3.prepare()
    misc.load_model(args=argparse.Namespace(resume=os.path.join(tmp_path, "mae.pth")), model_without_ddp=m3, optimizer=None, loss_scaler=None)
    assert torch.equal(m3.blocks[0].norm1.weight.cpu(), ck["model"]["blocks.0.norm1.weight"])


def test_resume_in_fp16_continues_the_scaler_and_the_step_count(dev, tmp_path):
    """`--amp fp16` across a checkpoint: three optimizer attempts (the second one overflows: a report weight of 3e38 makes the scaled loss inf),
    save, load into fresh model / optimizer / scaler objects, three more steps -- against the same six steps without the interruption.
    What must survive: the loss scale after its backoff, the growth tracker, the skipped-step count's effect on AdamW's bias corrections
    (`step` = steps actually TAKEN), the f32 masters and moments.  Parameters agree to the atomics' summation order."""
    import argparse
    from ecamp_amd import optim
    from ecamp_amd.data import synthetic_batch
    from ecamp_amd.module import model_ecamp as me
    from ecamp_amd.util import misc
    batches = [synthetic_batch(4, 32, 448, seed=30 + i) for i in range(6)]
    batches[1]["weights"] = batches[1]["weights"].clone()
    batches[1]["weights"][0, 0] = 3e38
    noise = torch.linspace(0, 1, 196).repeat(4, 1)

    def fresh():
        torch.manual_seed(1)
        m = me.ecamp_tiny(compute_dtype=torch.float16).to(dev)
        m.prepare()
        m.eval()
        return m, optim.FusedAdamW(optim.add_weight_decay(m, 0.05), lr=1e-3, betas=(0.9, 0.95)), misc.NativeScalerWithGradNormCount(dynamic=True, growth_interval=2)

    def steps(m, o, sc, lo, hi):
        for i in range(lo, hi):
            o.zero_grad()
            sc(sum(m(batches[i], noise=noise)), o, parameters=m.parameters(), update_grad=True)

    m1, o1, s1 = fresh()
    steps(m1, o1, s1, 0, 6)
    m2, o2, s2 = fresh()
    steps(m2, o2, s2, 0, 3)
    assert s2.skipped_steps == 1 and o2.steps_taken == 2 and s2.get_scale() == 32768.0 and s2.state_dict()["_growth_tracker"] == 1   # clean, overflow (x 0.5), clean
    args = argparse.Namespace(output_dir=str(tmp_path), resume="")
    misc.save_model(args=args, epoch=0, model=m2, model_without_ddp=m2, optimizer=o2, loss_scaler=s2)
    ck = torch.load(os.path.join(tmp_path, "checkpoint-0.pth"), map_location="cpu", weights_only=False)
    assert float(ck["optimizer"]["state"][0]["step"]) == 2.0 and ck["scaler"]["scale"] == 32768.0 and ck["scaler"]["_growth_tracker"] == 1
    m3, o3, s3 = fresh()
    with torch.no_grad():
        for p in m3.parameters():
            p.add_(1.0)     # whatever the fresh objects hold must be overwritten
    link = "./ECAMP_ckpt_fp16.pth"
    os.symlink(os.path.join(tmp_path, "checkpoint-0.pth"), link) if not os.path.exists(link) else None
    try:
        misc.load_model(args=argparse.Namespace(resume=link, start_epoch=0), model_without_ddp=m3, optimizer=o3, loss_scaler=s3)
    finally:
        os.remove(link)
    m3.eval()
    steps(m3, o3, s3, 3, 6)
    assert o3.steps_taken == o1.steps_taken == 5 and s3.state_dict() == s1.state_dict(), (o3.steps_taken, o1.steps_taken, s3.state_dict(), s1.state_dict())
    # the yardstick is a second uninterrupted run: AdamW moves an element by ~lr x m / sqrt(v), and where a gradient is rounding noise the
    # summation order of the f32 atomics decides its sign -- two identical runs already differ by a fraction of a learning-rate step there
    m4, o4, s4 = fresh()
    steps(m4, o4, s4, 0, 6)
    for n in ("blocks.3.mlp.fc1.weight", "bert_encoder.model.cls.predictions.decoder.weight", "decoder_pred.bias", "bert_encoder.model.bert.embeddings.LayerNorm.weight"):
        a, b, c = (dict(m.named_parameters())[n].detach().float().cpu() for m in (m1, m3, m4))
        d_resume, d_twin = (a - b).norm().item() / a.norm().item(), (a - c).norm().item() / a.norm().item()
        print("  %-60s resumed vs uninterrupted %.2e, two uninterrupted runs %.2e" % (n, d_resume, d_twin))
        assert torch.isfinite(b).all() and d_resume <= 3 * d_twin + 1e-6, (n, d_resume, d_twin)
    from ecamp_amd import _lib
    _lib.set_half("bf16")


def test_ddp_wrapper_runs_rccl_on_the_side_stream(dev):
    """One-rank RCCL group with the collectives forced on: the bucketed all-reduce (AVG over one rank = identity) runs on the side
    HIP stream behind the backward stages, and the gradients equal those of the unwrapped model."""
    import os

    import torch.distributed as dist
    from ecamp_amd.data import synthetic_batch
    from ecamp_amd.module import model_ecamp as me
    from ecamp_amd.parallel import DistributedDataParallel
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        batch = synthetic_batch(2, 32, 448, seed=3, device=dev)
        noise = torch.rand(2, 196, generator=torch.Generator().manual_seed(0))
        grads = []
        for wrap in (False, True):
            torch.manual_seed(0)
            model = me.ecamp_tiny(compute_dtype=torch.float32).to(dev)
            model.eval()
            net = DistributedDataParallel(model, bucket_cap_mb=8.0, force_comm=True) if wrap else model
            out = net(batch, noise=noise)
            sum(out).backward()
            torch.cuda.synchronize()
            if wrap:
                assert len(net.reducer.buckets) > 4 and net.reducer.use_avg
            grads.append(model.arena.flat_g.clone())
        assert torch.isfinite(grads[1]).all()
        assert float((grads[0] - grads[1]).abs().max()) <= 1e-6 * float(grads[0].abs().max())
    finally:
        if created:
            dist.destroy_process_group()


def test_bench_prints_one_contract_line(dev):
    """bench.py's output contract: exactly ONE JSON line on stdout with the driver's keys, the `roofline` object measured from
    HIP events, and `cpu_baseline` only when asked for (skipped here to keep the test short)."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "8", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600, cwd=root)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    r = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline"):
        assert k in r, k
    assert r["unit"] == "pairs/s" and r["n_gpus"] == 1 and r["steps"] == 2 and r["warmup"] == 1 and r["higher_is_better"] is True
    assert r["scaling"] == "weak" and r["vs_baseline"] is None and r["dtype"] == "bf16" and r["data"] == "synthetic"
    assert "workload" in r["config"] and "model" not in r["config"]
    assert abs(r["value"] - 8 * 1e3 / r["ms_per_step"]) < 1e-2 * r["value"]
    rf = r["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and rf["peak"] == 2500.0 and 0 < rf["frac"] < 1
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert rf["traffic"] is None      # the committed PMC passes are for the B=256 workload only
    # the north-star's own scope: the image side alone, forward + backward, 42.92 GFLOP per image
    assert r["vit_fwd_bwd_ms"] > 0 and abs(r["vit_tflops"] - 42.92e9 * 8 / (r["vit_fwd_bwd_ms"] * 1e-3) / 1e12) < 1e-2 * r["vit_tflops"] + 0.01
    assert abs(r["vit_frac"] - r["vit_tflops"] / 2500.0) < 1e-3 and r["vit_fwd_bwd_ms"] < r["fwd_bwd_ms"]
    assert "cpu_baseline" not in r
    # `value` is the resident rate (inputs in HBM before the timed region); the PCIe-inclusive leg and its explanation ride beside it
    for k in ("step_ms", "host_inclusive_pairs_per_s", "host_inclusive_ms_per_step", "host_inclusive_step_ms", "h2d_ms_per_step", "h2d_gbps",
              "h2d_gbps_needed", "value_median_pairs_per_s"):
        assert k in r, k
    assert r["step_ms"]["min"] <= r["step_ms"]["median"] <= r["step_ms"]["p90"] <= r["step_ms"]["max"]
    assert r["h2d_gbps"] > 0 and r["h2d_mb_per_step"] > 8 * 3 * 448 * 448 * 4 / 1e6


def test_bench_driver_protocol_value_is_explained_by_its_own_record(dev):
    """VERDICT r5 item 1.  The driver's command (`--steps 20 --warmup 5`, the B = 256 workload): `value` (inputs resident in HBM when the
    timed region starts) must be a steady number -- no step of the timed region more than 10 % over the median, mean = median -- and the
    PCIe-inclusive leg must either sit within 3 % of it or carry the measured copy bandwidth that explains the gap (the copy of a
    616 MB batch hides under a step only while the box's host -> HBM path delivers more than bytes / step time).  A slow step must be
    EXPLAINED by the record (the host took long to queue it, or the allocator / garbage collector ran) and must not repeat: a run with one
    is repeated once -- the first GPU processes of a fresh box have shown single steps of 2x-10x the median (profiles/r06_host_lead_and_stalls.txt)
    -- and the second run has to be clean."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    torch.cuda.empty_cache()   # the child needs ~60 GB of the card this process may hold in its allocator's cache

    def run():
        p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-prof"],
                           capture_output=True, text=True, timeout=900, cwd=root)
        assert p.returncode == 0, p.stderr[-2000:]
        r = json.loads([l for l in p.stdout.splitlines() if l.strip()][-1])
        st, hs = r["step_ms"], r["host_inclusive_step_ms"]
        print("  value %.0f pairs/s (%.2f ms/step; steps min %.2f median %.2f p90 %.2f max %.2f) | host-inclusive %.0f pairs/s (%.2f ms/step; "
              "steps min %.2f median %.2f max %.2f) | h2d %.1f ms per batch = %.1f GB/s, needed %.1f GB/s"
              % (r["value"], r["ms_per_step"], st["min"], st["median"], st["p90"], st["max"], r["host_inclusive_pairs_per_s"],
                 r["host_inclusive_ms_per_step"], hs["min"], hs["median"], hs["max"], r["h2d_ms_per_step"], r["h2d_gbps"], r["h2d_gbps_needed"]))
        print("  value leg: host ms to queue a step %s, runtime counters %s, steps %s" % (st.get("host_queue_ms"), st.get("runtime"), st.get("sequence")))
        return r, st, hs

    r, st, hs = run()
    if st["max"] > 1.10 * st["median"]:
        hq, rt = st["host_queue_ms"], st["runtime"]
        explained = hq["of_slowest_step"] > 1.5 * hq["median"] or hq["max"] > 2.0 * hq["median"] or rt["device_allocs"] > 16 or rt["gc_gen2"] > 0 or rt["alloc_retries"] > 0
        print("  a step of %.1f ms against a median of %.2f: %s by the record; running once more" % (st["max"], st["median"], "explained" if explained else "NOT explained"))
        assert explained, st
        r, st, hs = run()
    assert st["max"] <= 1.10 * st["median"], st                      # no stall inside the timed region of `value`
    assert abs(r["ms_per_step"] - st["median"]) <= 0.03 * st["median"], (r["ms_per_step"], st)   # mean = median: nothing hides in the mean
    gap = r["host_inclusive_ms_per_step"] / r["ms_per_step"] - 1.0
    assert gap <= 0.03 or r["h2d_gbps"] < 1.25 * r["h2d_gbps_needed"], (gap, r["h2d_gbps"], r["h2d_gbps_needed"], hs)


def test_lazy_zero_grad_matches_memset_and_flushes_unwritten_weights(dev):
    """FusedAdamW.zero_grad() zeroes only the atomically-accumulated gradients; weight matrices are overwritten by the first
    weight-gradient GEMM of the next backward.  (1) with accum_iter=2, the gradients after the overwriting micro-step and after the
    accumulating one, and the gradient norm, equal the eager memset path (to the run-to-run noise of the atomic sums); (2) a weight
    whose module did not run in a window reads as zero once flushed."""
    import ecamp_amd.arena as arena_mod
    from ecamp_amd import optim
    from ecamp_amd.module import model_ecamp as me
    from ecamp_amd.util.misc import NativeScalerWithGradNormCount
    from oracle import ecamp_oracle as orc
    from oracle import recipe
    cfg = orc.cfg_tiny()
    state = recipe.recipe_state(cfg, seed=0)
    batches = [recipe.recipe_batch(cfg, 4, 128, seed=s) for s in range(2)]
    noise = recipe.recipe_noise(4, cfg.num_patches, seed=0)
    out = {}
    for lazy in (True, False):
        arena_mod.LAZY_ZERO = lazy
        torch.manual_seed(0)
        model = me.ecamp_tiny(compute_dtype=torch.bfloat16)
        model.load_state_dict(state, strict=True)
        model.to(dev).eval()
        model.prepare()
        opt = optim.FusedAdamW(optim.add_weight_decay(model, 0.05), lr=1e-3, betas=(0.9, 0.95))
        scaler = NativeScalerWithGradNormCount()
        opt.zero_grad()
        snaps = []
        for it in range(3):
            mim, res, mlm = model(batches[it % 2], mask_ratio=0.75, noise=noise)
            ((mim + res + mlm) / 2).backward()
            if it == 0:     # the first backward teaches the arena which weights are GEMM-written; no parameter update in this test,
                opt.zero_grad()   # so both modes see identical activations: this zero_grad() is the first lazy one
                continue
            opt.flush_grads()     # it = 1 overwrites, it = 2 accumulates
            snaps.append({k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None})
        from ecamp_amd.util import misc
        out[lazy] = (snaps, misc.get_grad_norm_(model.parameters()).item())
        if lazy:
            A = model.arena
            assert len(A._gemm_written) > 50 and A._zero_flags is not None
            assert int(A._zero_flags.sum().item()) < A._zero_flags.numel() // 3   # the scatter-added word-embedding table is most of the rest
            # (2) a window in which the report side does not run: its weights keep old values until flushed, then read zero
            opt.zero_grad()
            lat, mask, ids_restore, ids_keep = model.image_encoder(orc.bicubic_resize(batches[0]["image"], cfg.img_size), 0.75, noise=noise)
            lat.float().sum().backward()
            w = model.bert_encoder.model.bert.encoder.layer[0].output.dense.weight
            opt.flush_grads()
            assert w.grad.abs().max().item() == 0.0
            assert model.blocks[0].mlp.fc1.weight.grad.abs().max().item() > 0.0
    arena_mod.LAZY_ZERO = True
    assert abs(out[True][1] - out[False][1]) <= 1e-4 * out[False][1], (out[True][1], out[False][1])
    for a, b in zip(out[True][0], out[False][0]):
        for k, g in a.items():
            if k.endswith("key.bias"):      # true gradient is zero (softmax shift invariance): only rounding noise lives there
                continue
            if g.dim() == 2 and k.endswith(".weight") and "embeddings" not in k:
                assert torch.equal(g, b[k]), k          # deterministic GEMMs on identical activations: bit for bit
            else:                                       # tokens, biases, LayerNorm, embedding rows: atomic sums, order varies
                assert (g - b[k]).abs().max().item() <= 5e-3 * b[k].abs().max().item() + 1e-9, k


@pytest.mark.parametrize("accum,branches,gdt", [(1, 0, "f32"), (2, 0, "f32"), (1, 1, "f32"), (1, 0, "bf16"), (2, 0, "f32+dynamic-loss-scale")])
def test_ddp_two_ranks_equal_single_process(dev, tmp_path, accum, branches, gdt):
    """SURVEY.md section 4, "distributed without a cluster" (main_pretrain.py:247-250): two data-parallel ranks of B=4 (accum 1) or
    2 x B=2 with no_sync on the first micro-step (accum 2) must leave, after the bucketed all-reduce, the SAME gradient arena and
    the same parameters after one AdamW step as ONE process on the concatenated B=8 batch.  Both ranks share cuda:0 (this pool
    has one GPU per box), so the process group is gloo on device tensors; rank 1 starts from a different initialisation, which the
    wrapper's parameter broadcast must overwrite.  fp32 parity mode, tiny config, recipe inputs.  branches=1: the ranks run with
    ECAMP_OVERLAP_BRANCHES=1 (image decoder and report side on two streams, forward and backward): a bucket reported from a
    branch-stream node must still wait for the main stream's share of its gradients.  gdt = "bf16": the optional bf16
    gradient exchange (round 4: half the bytes on the links) -- the arena then agrees to bf16 rounding (contract 1e-2, measured ~3e-3).
    "f32+dynamic-loss-scale": the ranks run the reference's GradScaler (loss x 65536 in every micro-step, the overflow check over the
    ALL-REDUCED arena, skip / un-scale decided on the device) and must land on the same gradients, norm and parameters."""
    import socket
    import subprocess
    import sys
    from ecamp_amd import optim
    from ecamp_amd.module import model_ecamp as me
    from ecamp_amd.util import misc
    from oracle import ecamp_oracle as orc
    from oracle import recipe
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = orc.cfg_tiny()
    B, S = 8, 64
    batch = recipe.recipe_batch(cfg, B, S, seed=5)
    noise = recipe.recipe_noise(B, cfg.num_patches, seed=5)
    model = me.ecamp_tiny(compute_dtype=torch.float32)
    model.load_state_dict(recipe.recipe_state(cfg, seed=0))
    model.to(dev).eval()
    model.prepare()
    opt = optim.FusedAdamW(optim.add_weight_decay(model, 0.05), lr=1e-3, betas=(0.9, 0.95))
    opt.zero_grad()
    sum(model(batch, noise=noise)).backward()
    model.arena.flush_fresh()
    torch.cuda.synchronize()
    ref_g = model.arena.flat_g.detach().cpu().clone()
    ref_norm = float(misc.get_grad_norm_(model.parameters()))
    opt.step()
    torch.cuda.synchronize()
    ref_p = model.arena.flat_p.detach().cpu().clone()

    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    out = os.path.join(tmp_path, "ddp_rank0.pt")
    worker = os.path.join(root, "tests", "_ddp_worker.py")
    dyn = gdt.endswith("dynamic-loss-scale")
    gdt = gdt.split("+")[0]
    env = dict(os.environ, ECAMP_OVERLAP_BRANCHES=str(branches), ECAMP_DDP_GRAD_DTYPE=gdt, ECAMP_TEST_LOSS_SCALE="dynamic" if dyn else "none")
    procs = [subprocess.Popen([sys.executable, worker, str(r), "2", str(port), str(accum), out], cwd=root, env=env) for r in range(2)]
    rcs = [p.wait(timeout=600) for p in procs]
    assert rcs == [0, 0], rcs
    got = torch.load(out, map_location="cpu")
    eg = float((got["flat_g"] - ref_g).abs().max() / ref_g.abs().max())
    ep = float((got["flat_p"] - ref_p).abs().max() / ref_p.abs().max())
    print("  DDP 2 ranks (accum %d) vs single process: grad arena rel %.2e, params after AdamW rel %.2e, norm %.6f vs %.6f"
          % (accum, eg, ep, got["norm"], ref_norm))
    if gdt == "bf16":
        assert 1e-6 < eg <= 1e-2, eg      # really went through bf16, and no further than its rounding
        assert abs(got["norm"] - ref_norm) <= 5e-3 * ref_norm
        return
    assert eg <= 1e-5, eg
    # the first AdamW step is lr * g / (|g| + eps): where a gradient element is ~0 a 1e-7 difference moves the update by O(lr), so the
    # parameters are compared at lr-scale resolution (lr = 1e-3, |p|max ~ 1); the gradient arena above is the contract
    assert ep <= 1e-4, ep
    assert abs(got["norm"] - ref_norm) <= 1e-5 * ref_norm


def test_optimizer_refuses_unreduced_gradients(dev):
    """A reducer that saw gradients but was never finalised (no autograd callback, no scaler call) must stop the optimizer."""
    from ecamp_amd import optim
    from ecamp_amd.module import model_ecamp as me
    from ecamp_amd.parallel import GradReducer
    model = me.ecamp_tiny(compute_dtype=torch.float32).to(dev)
    A = model.prepare()
    opt = optim.FusedAdamW(optim.add_weight_decay(model, 0.05), lr=1e-3)
    A.reducer = GradReducer(A.flat_g, A.offsets, A.sizes, A.unused, force_comm=False)
    A.reducer.mark_ready([0])          # outside a backward pass: no callback can be queued
    with pytest.raises(RuntimeError, match="never"):
        opt.step()
    A.reducer.finalize()
    opt.step()


def test_device_prefetcher_delivers_every_batch_intact(dev):
    """ecamp_amd.data.DevicePrefetcher (defines bench.py's `value`): eight DISTINCT host batches through its three staging slots with a
    slow consumer arrive intact and in order; an early break with a copy in flight, followed by allocations that recycle the slots'
    memory, corrupts nothing; an exception inside the consumer leaves the iterator closed cleanly."""
    from ecamp_amd.data import DevicePrefetcher
    g = torch.Generator().manual_seed(0)
    host = [{"image": torch.randn(4, 3, 448, 448, generator=g), "ids": torch.randint(0, 30000, (4, 128), generator=g), "tag": i} for i in range(8)]
    seen = []
    for i, b in enumerate(DevicePrefetcher(host, dev)):
        assert b["tag"] == i and b["image"].is_cuda and b["ids"].is_cuda
        # slow consumer: a long kernel chain reads the batch well after the host has moved on to staging the next ones
        acc = b["image"].clone()
        for _ in range(20):
            acc = acc * 1.0001 + 0.0
        seen.append((b["image"].clone(), b["ids"].clone(), acc))
    torch.cuda.synchronize()
    assert len(seen) == 8
    for i, (im, ids, _) in enumerate(seen):
        assert torch.equal(im.cpu(), host[i]["image"]) and torch.equal(ids.cpu(), host[i]["ids"]), i
    # early break while batch 2's copy may be in flight, then recycle memory on the compute stream
    it = iter(DevicePrefetcher(host, dev))
    first = next(it)["image"].clone()
    second = next(it)["image"]
    keep = second.clone()
    it.close()
    junk = [torch.full((4, 3, 448, 448), float(k), device=dev) for k in range(6)]
    torch.cuda.synchronize()
    assert torch.equal(first.cpu(), host[0]["image"]) and torch.equal(keep.cpu(), host[1]["image"])
    assert all(float(j.flatten()[0]) == float(k) and float(j.flatten()[-1]) == float(k) for k, j in enumerate(junk))
    with pytest.raises(RuntimeError, match="consumer failed"):
        for b in DevicePrefetcher(host, dev):
            raise RuntimeError("consumer failed")
    torch.cuda.synchronize()


@pytest.mark.gpu
@pytest.mark.parametrize("branches", [1, 0])
def test_every_gradient_bucket_is_final_when_its_collective_may_start(dev, branches):
    """The data-parallel contract of `arena.ready()`, checked on ONE GPU under the real stream concurrency (weight-gradient side
    stream, image-decoder branch stream): a bucket's all-reduce is queued behind every stream that writes gradients at the moment
    its last parameter is reported -- so nothing may write into the bucket afterwards.  A 1-rank reducer (no communication) runs a
    hook in the collective's place that copies the bucket on the communication stream; after backward every copy must equal the
    final gradient arena bit for bit.  ViT-B / reference BERT at B=32 (the streams really run apart), second backward pass (the
    weight matrices are in overwrite mode), 4 MiB buckets (every layer boundary is a bucket boundary somewhere)."""
    from ecamp_amd import hip_ops, optim
    from ecamp_amd.data import synthetic_batch
    from ecamp_amd.module import model_ecamp as me
    from ecamp_amd.parallel import GradReducer
    old = hip_ops.OVERLAP_BRANCHES
    hip_ops.OVERLAP_BRANCHES = bool(branches)
    try:
        torch.manual_seed(0)
        model = me.ecamp(compute_dtype=torch.bfloat16).to(dev)
        model.train()
        A = model.prepare()
        opt = optim.FusedAdamW(optim.add_weight_decay(model, 0.05), lr=1e-4, betas=(0.9, 0.95))
        red = GradReducer(A.flat_g, A.offsets, A.sizes, A.unused, bucket_mb=4.0, tail_bucket_mb=1.0)
        A.reducer, A.on_ready = red, red.mark_ready
        assert red.world == 1 and len(red.buckets) > 100
        batch = synthetic_batch(32, 128, 448, seed=1, device=dev)
        snaps = []
        order = []

        def hook(lo, hi, slots):
            snaps.append((lo, hi, A.flat_g[lo:hi].clone()))     # on the communication stream, where the all-reduce would read it
            order.append(lo)

        for it in range(2):
            snaps.clear(); order.clear()
            red.main_stream = torch.cuda.current_stream()
            red.after_bucket = hook
            mim, res, mlm = model(batch)
            (mim + res + mlm).backward()
            red.after_bucket = None
            torch.cuda.synchronize()
            assert len(snaps) == len(red.buckets) and not red.dirty
            early = sum(1 for lo, hi, _ in snaps[:len(snaps) // 2] if lo > A.total // 3)
            assert early > 10                                  # buckets really were launched during backward, in backward's order
            if it == 0:
                opt.zero_grad()                                # lazy from now on: weight matrices are overwritten, not zeroed
        bad = []
        for lo, hi, snap in snaps:
            if not torch.equal(snap, A.flat_g[lo:hi]):
                names = [A.names[i] for i in range(len(A.names)) if lo <= A.offsets[i] < hi]
                d = (snap - A.flat_g[lo:hi]).abs()
                bad.append((names[0], names[-1], float(d.max()), int((d > 0).sum())))
        assert not bad, "written after their bucket was released: %s" % bad[:5]
    finally:
        hip_ops.OVERLAP_BRANCHES = old


@pytest.mark.gpu
def test_bucketwise_adamw_behind_the_allreduces_equals_the_plain_step(dev, monkeypatch):
    """Data-parallel path of the loss scaler: with a reducer that communicates (here a 1-rank RCCL group with the collectives forced
    on), the end-of-backward callback leaves the buckets' completion events to FusedAdamW, which updates bucket by bucket, each
    slice behind its own all-reduce.  Checked step by step on the SAME gradients (two training runs cannot be compared bit for bit:
    atomically summed gradients differ in their last bits run to run and Adam's first steps amplify that): after every scaler call
    the gradient arena, which the update does not modify, is fed to a one-pass AdamW starting from the saved pre-step parameters
    and moments -- parameters, moments, bf16 shadow and the gradient norm must come out the same, bit for bit.  Three optimizer
    steps, the second one over two accumulation micro-steps with no_sync on the first."""
    import os

    import torch.distributed as dist
    from ecamp_amd import hip_ops, optim
    from ecamp_amd.module import model_ecamp as me
    from ecamp_amd.parallel import DistributedDataParallel
    from ecamp_amd.util import misc
    from oracle import ecamp_oracle as orc
    from oracle import recipe
    monkeypatch.setattr(misc, "_BUCKETWISE_ADAMW", True)   # opt-in since round 5 (ecamp_amd/parallel.py ddp_defaults)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29534")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        cfg = orc.cfg_tiny()
        state = recipe.recipe_state(cfg, seed=0)
        batches = [recipe.recipe_batch(cfg, 4, 128, seed=s) for s in range(4)]
        noise = recipe.recipe_noise(4, cfg.num_patches, seed=0)
        plan = [(0, True, 1.0), (1, False, 0.5), (2, True, 0.5), (3, True, 1.0)]    # (batch, update_grad, loss scale)
        torch.manual_seed(0)
        model = me.ecamp_tiny(compute_dtype=torch.bfloat16)
        model.load_state_dict(state, strict=True)
        model.to(dev).eval()
        A = model.prepare()
        net = DistributedDataParallel(model, bucket_cap_mb=2.0, tail_bucket_mb=0.5, tail_span_mb=2.0, force_comm=True)
        opt = optim.FusedAdamW(optim.add_weight_decay(model, 0.05), lr=1e-3, betas=(0.9, 0.95))
        scaler = misc.NativeScalerWithGradNormCount()
        opt.zero_grad()
        opt._bind()
        g0 = opt.param_groups[0]
        steps = 0
        for bi, upd, sc in plan:
            net.set_grad_sync(upd)
            before = (A.flat_p.clone(), opt._m.clone(), opt._v.clone())
            mim, res, mlm = net(batches[bi], mask_ratio=0.75, noise=noise)
            n = scaler((mim + res + mlm) * sc, opt, parameters=model.parameters(), update_grad=upd)
            if not upd:
                assert n is None and torch.equal(A.flat_p, before[0])
                continue
            steps += 1
            torch.cuda.synchronize()
            assert opt.bucketwise_steps == steps and net.reducer.bucket_events is None and not net.reducer.lazy
            p, m, v = before
            s = torch.zeros(1, device=dev)
            p16 = torch.empty_like(A.flat_p16)
            hip_ops.adamw_grouped(p, A.flat_g, m, v, p16, opt._table, [g["lr"] for g in opt.param_groups], [g["weight_decay"] for g in opt.param_groups],
                                  g0["betas"][0], g0["betas"][1], g0["eps"], steps, 1.0, s)
            torch.cuda.synchronize()
            assert torch.equal(p, A.flat_p) and torch.equal(m, opt._m) and torch.equal(v, opt._v)
            used = opt._table.repeat_interleave(64) < 8          # the pooler is never updated (and its shadow never rewritten)
            assert torch.equal(p16[used], A.flat_p16[used])
            assert abs(float(s.sqrt()) - float(n)) <= 1e-3 * float(n)     # f32 sum of squares: one launch against one per bucket
            opt.zero_grad()
        assert steps == 3 and len(net.reducer.buckets) > 8
    finally:
        if created:
            dist.destroy_process_group()
