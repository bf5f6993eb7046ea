"""not gpu: host-side logic of the drop-in surface -- state-dict keys, parameter groups, LR schedule, sin-cos table,
meters, the synthetic dataset schema, the run.sh command line, and the refusal to run without the HIP device."""
import argparse
import math
import os
import types

import numpy as np
import pytest
import torch

from ecamp_amd import optim
from ecamp_amd.data import SyntheticContextBertDataset, synthetic_batch
from ecamp_amd.module import model_ecamp as me
from ecamp_amd.module.bert_config import BertConfig
from ecamp_amd.util import lr_sched, misc
from ecamp_amd.util.pos_embed import get_2d_sincos_pos_embed
from oracle import ecamp_oracle as orc

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def tiny():
    torch.manual_seed(0)
    return me.ecamp_tiny()


def test_state_dict_keys_shapes_and_order_match_the_reference(tiny):
    ref = orc.param_shapes(orc.cfg_tiny())
    sd = tiny.state_dict()
    assert list(sd.keys()) == list(ref.keys())
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(ref[k][0]), k
    assert not tiny.pos_embed.requires_grad and not tiny.decoder_pos_embed.requires_grad
    pr = tiny.bert_encoder.model.cls.predictions
    assert pr.decoder.bias is pr.bias  # one Parameter under two names, as transformers 4.42.4
    n_train = sum(p.numel() for p in tiny.parameters() if p.requires_grad)
    n_ref = sum(int(np.prod(ref[k][0])) for k in orc.trainable_names(orc.cfg_tiny()))
    assert n_train == n_ref


def test_base_model_has_the_reference_inventory():
    ref = orc.param_shapes(orc.cfg_base())
    assert len(ref) == 350
    n_train = sum(int(np.prod(ref[k][0])) for k in orc.trainable_names(orc.cfg_base()))
    assert abs(n_train / 1e6 - 183.17) < 0.4  # 183.17 M trainable (one fewer bias under the 4.42.4 tie)


def test_weight_decay_groups_match_reference(tiny):
    g = np.load(os.path.join(GOLD, "tiny_b4_s128.npz"))
    groups = optim.add_weight_decay(tiny, 0.05)
    names = {id(p): n for n, p in tiny.named_parameters()}
    assert sorted(names[id(p)] for p in groups[0]["params"]) == sorted(g["wd/no_decay"])
    assert sorted(names[id(p)] for p in groups[1]["params"]) == sorted(g["wd/decay"])
    assert groups[0]["weight_decay"] == 0.0 and groups[1]["weight_decay"] == 0.05
    assert "cls_token" in g["wd/decay"] and "mask_token" in g["wd/decay"]  # 3-D tokens ARE decayed


def test_lr_schedule_matches_reference_table():
    g = np.load(os.path.join(GOLD, "tiny_b4_s128.npz"))
    args = types.SimpleNamespace(lr=1.5e-4, min_lr=0.0, warmup_epochs=40, max_epoch=200)
    opt = types.SimpleNamespace(param_groups=[{"lr": 0.0}, {"lr": 0.0, "lr_scale": 0.5}])
    got = [lr_sched.adjust_learning_rate(opt, float(e), args) for e in g["lr/epochs"]]
    assert np.allclose(got, g["lr/values"], rtol=1e-14, atol=0)
    assert opt.param_groups[1]["lr"] == got[-1] * 0.5


def test_sincos_table_matches_reference_digest():
    from oracle.make_golden import digest
    g = np.load(os.path.join(GOLD, "base_b2_s128.npz"))
    for dim, key in ((768, "pos_embed"), (512, "decoder_pos_embed")):
        t = torch.from_numpy(get_2d_sincos_pos_embed(dim, 14, cls_token=True)).float().unsqueeze(0)
        nm, s = digest(t)
        assert np.abs(s - g["tab/%s/s" % key]).max() < 1e-7 and abs(nm[0] - g["tab/%s/nm" % key][0]) < 1e-4


def test_init_distributions_follow_the_reference(tiny):
    # xavier_uniform on every nn.Linear incl. BERT's and the 30000-way decoder (model_ecamp.py:124-135)
    w = tiny.bert_encoder.model.cls.predictions.decoder.weight
    bound = (6.0 / (w.shape[0] + w.shape[1])) ** 0.5
    assert w.abs().max().item() <= bound * 1.0001 and w.abs().max().item() > 0.9 * bound
    assert (tiny.norm.weight == 1).all() and (tiny.norm.bias == 0).all()
    emb = tiny.bert_encoder.model.bert.embeddings.word_embeddings.weight
    assert (emb[0] == 0).all() and abs(emb[1:].std().item() - 0.02) < 2e-3  # HF init, PAD row zero
    assert abs(tiny.cls_token.std().item() - 0.02) < 0.01


def test_no_cpu_fallback(tiny):
    from ecamp_amd._lib import EcampHipError
    with pytest.raises(EcampHipError):
        tiny(synthetic_batch(1, 32, 448))
    from ecamp_amd import hip_ops
    with pytest.raises(EcampHipError):
        hip_ops.layernorm_fwd(torch.zeros(4, 8), torch.ones(8), torch.zeros(8), 1e-6)


def test_old_fusion_layer_key_alias(tiny):
    sd = {k.replace("context_fusion_layer", "cross_attn_layer"): v.clone() for k, v in tiny.state_dict().items()}
    m2 = me.ecamp_tiny()
    m2.load_state_dict(sd, strict=True)
    assert torch.equal(m2.bert_encoder.model.bert.context_fusion_layer.gap_mlp.weight, tiny.bert_encoder.model.bert.context_fusion_layer.gap_mlp.weight)


def test_bad_configs_are_rejected():
    with pytest.raises(ValueError):
        me.ECAMP(embed_dim=100, num_heads=3)
    with pytest.raises(ValueError):
        BertConfig(hidden_act="relu")
    with pytest.raises(ValueError):
        me.ECAMP(compute_dtype=torch.float64)
    with pytest.raises(ValueError, match="patch_size 16"):
        me.ECAMP(img_size=128, patch_size=8)     # the SR head's kernels are built for patch 16: said at construction, not at the first forward


def test_synthetic_dataset_schema():
    b = synthetic_batch(3, 64, 448, seed=1)
    assert b["image"].shape == (3, 3, 448, 448) and b["image"].dtype == torch.float32
    for k in ("ids", "labels", "attention_mask", "type_ids"):
        assert b[k].shape == (3, 64) and b[k].dtype == torch.int64
    assert (b["labels"][:, 0] == 2).all() and (b["ids"][b["ids"] != b["labels"]] == 3).all()
    assert ((b["labels"] == 0) == (b["attention_mask"] == 0)).all()
    assert b["weights"].shape == (3, 64) and b["column"].shape == (3,) and int(b["column"].max()) <= 2
    ds = SyntheticContextBertDataset(length=4, max_caption_length=32)
    one = ds.collate_fn([ds[0]])
    assert one["ids"].shape == (1, 32)  # no .squeeze() bug at B == 1 (pretrain_datasets.py:218-225)


def test_meters_accept_tensors_lazily():
    ml = misc.MetricLogger(delimiter="  ")
    for i in range(5):
        ml.update(loss=torch.tensor(float(i)), lr=0.1)
    assert ml.loss.global_avg == pytest.approx(2.0) and ml.lr.value == 0.1 and ml.loss.median == pytest.approx(2.0)
    out = list(ml.log_every(range(3), 2, "hdr"))
    assert out == [0, 1, 2]


def test_run_sh_command_line_parses():
    """The flags of the repo-root run.sh (the reference's run.sh:3-16, value for value) parse with the driver's argument parser."""
    import os, re, shlex
    from ecamp_amd.main_pretrain import get_args_parser
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "run.sh")).read()
    cmd = text[text.index("-m ecamp_amd.main_pretrain") + len("-m ecamp_amd.main_pretrain"):].replace("\\\n", " ")
    argv = [t for t in shlex.split(cmd) if t != "$@"]
    want = "--num_workers 16 --accum_iter 8 --batch_size 256 --model ecamp --norm_pix_loss --mask_ratio 0.75 --epochs 120 " \
           "--warmup_epochs 40 --lr 1.5e-4 --weight_decay 0.05 --resume ./dataset/mae_vit_base.pth --data_path ./dataset/ " \
           "--output_dir ../output/ --description".split() + ["ECAMP pretraining"]
    assert argv == want, argv
    a = argparse.ArgumentParser(parents=[get_args_parser()]).parse_args(argv)
    assert not a.synthetic   # a missing CSV is an error unless --synthetic is given
    assert a.accum_iter == 8 and a.batch_size == 256 and a.lr == 1.5e-4 and a.max_epoch == 200 and a.norm_pix_loss
    assert "ecamp" in me.__dict__ and callable(me.__dict__[a.model])


def test_grad_norm_reference_formula_on_cpu_tensors():
    ps = [torch.nn.Parameter(torch.randn(5, 3)), torch.nn.Parameter(torch.randn(7))]
    for p in ps:
        p.grad = torch.randn_like(p)
    n = misc.get_grad_norm_(ps)
    assert n.item() == pytest.approx(torch.cat([p.grad.flatten() for p in ps]).norm().item(), rel=1e-6)


def test_bench_refuses_to_run_without_a_gpu():
    """The product path has no CPU fallback: on a box without an MI355X bench.py stops with a message instead of timing anything."""
    import os, subprocess, sys
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("a GPU is present")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    assert "no CPU fallback" in (p.stderr + p.stdout)
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]
    # --gpus 2 without RANK on a box with fewer than two GPUs: the self-launcher says so and launches nothing (no hang in RCCL init)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True,
                       timeout=600, env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")})
    assert p.returncode != 0 and "--gpus 2 but this box has 0 visible GPU(s)" in p.stderr
    # a launcher that provides a different world size is an error, not a silent single-GPU run
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0"], capture_output=True, text=True,
                       timeout=300, env=dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0"))
    assert p.returncode != 0 and "WORLD_SIZE=1" in (p.stderr + p.stdout)


def test_main_pretrain_needs_the_dataset_or_an_explicit_synthetic_flag(tmp_path):
    """a mistyped --data_path must not silently train on noise."""
    from ecamp_amd import main_pretrain
    args = argparse.ArgumentParser(parents=[main_pretrain.get_args_parser()]).parse_args(
        ["--lr", "1e-4", "--data_path", str(tmp_path), "--output_dir", str(tmp_path)])
    with pytest.raises(FileNotFoundError, match="--synthetic"):
        main_pretrain.main(args)


def test_bench_self_launcher_stops_the_other_ranks_when_one_dies():
    """bench.py's self-launcher (the `python bench.py --gpus N` path of the driver's 8-GPU run): a rank that exits non-zero must not
    leave its siblings blocked in a collective -- the parent terminates them, names the failed rank, shows its stderr and returns its
    code within seconds.  CPU stub children: rank 1 dies with code 3, rank 0 would sleep for ten minutes."""
    import importlib.util, os, sys, time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    stub = ("import os, sys, time\n"
            "r = int(os.environ['RANK']); assert os.environ['WORLD_SIZE'] == '2' and os.environ['MASTER_ADDR'] == '127.0.0.1'\n"
            "assert int(os.environ['OMP_NUM_THREADS']) >= 1\n"
            "if r == 1:\n"
            "    print('stub rank 1: simulated failure', file=sys.stderr); sys.exit(3)\n"
            "time.sleep(600)\n")
    t0 = time.time()
    rc = bench.launch_ranks([sys.executable, "-c", stub], 2, check_devices=False)
    assert rc == 3 and time.time() - t0 < 30
    ok = "import os, sys; sys.exit(0)"
    assert bench.launch_ranks([sys.executable, "-c", ok], 2, check_devices=False) == 0


def test_meters_match_the_reference_meters():
    """`SmoothedValue` / `MetricLogger` (util/misc.py:24-167) against values captured from the REFERENCE's own classes
    (oracle/make_golden_meters.py -> tests/golden/meters.npz) on a fixed series fed the way train_one_epoch feeds them -- here through
    the lazy path (0-d tensors read back only when a statistic is asked for): median, avg, global_avg, max, value of every meter and
    the formatted log line, after every update."""
    import os
    import numpy as np
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "meters.npz"), allow_pickle=False)
    vals, lr, want, lines, lens = g["values"], g["lr"], g["stats"], g["lines"], g["line_lens"]
    names = [str(n) for n in g["names"]]
    ml = misc.MetricLogger(delimiter="  ")
    ml.add_meter("lr", misc.SmoothedValue(window_size=1, fmt="{value:.6f}"))
    for i in range(len(vals)):
        ml.update(mim_loss=torch.tensor(vals[i, 0]), res_loss=float(vals[i, 1]), mlm_loss=torch.tensor(vals[i, 2]))
        ml.update(lr=float(lr[i]))
        if i % 3 == 0 or i == len(vals) - 1:   # reading only every third step leaves tensors pending in between (the lazy path)
            for j, k in enumerate(names):
                m = ml.meters[k]
                got = [m.median, m.avg, m.global_avg, m.max, m.value]
                assert got == pytest.approx(list(want[i, j]), rel=1e-12, abs=0), (i, k, got, want[i, j])
            assert str(ml) == bytes(lines[i, :lens[i]]).decode(), i


def test_roofline_traffic_is_quoted_only_for_the_sources_it_was_measured_on():
    """bench.py takes `roofline.traffic` from the newest committed PMC passes (profiles/rNN_pmc_traffic.json); tools/pmc_traffic.py
    stamps that file with the hash of the GEMM sources it was collected on, and bench.py quotes it only when the hash equals this
    tree's -- a stale file yields null, never last round's number."""
    import glob, json
    from ecamp_amd.build import gemm_source_hash
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = gemm_source_hash()
    assert len(h) == 64 and h == gemm_source_hash()
    src = open(os.path.join(root, "bench.py")).read()
    assert "gemm_source_hash()" in src and "r04_pmc_traffic.json" not in src
    assert "gemm_source_sha256" in open(os.path.join(root, "tools", "pmc_traffic.py")).read()
    for f in glob.glob(os.path.join(root, "profiles", "r0[1-4]_pmc_traffic.json")):   # the unstamped files of rounds 1-4 can never match
        assert "gemm_source_sha256" not in json.load(open(f))



def test_dynamic_loss_scaler_follows_torch_grad_scaler():
    """`NativeScalerWithGradNormCount(dynamic=True)` -- the reference's `torch.cuda.amp.GradScaler()` (util/misc.py:251-271) -- against torch's
    own GradScaler on the CPU, same model, same batches, an overflow injected on steps 2 and 3 and accumulation over two micro-steps: scale
    and growth tracker after every update, which steps were skipped, the returned gradient norm (nan / inf on overflow) and the parameters
    after every step are equal; the state dict round-trips and a non-dynamic scaler ignores a checkpoint's scale."""
    import copy
    from ecamp_amd.util.misc import NativeScalerWithGradNormCount
    torch.manual_seed(0)
    m_ref = torch.nn.Sequential(torch.nn.Linear(6, 8), torch.nn.GELU(), torch.nn.Linear(8, 3))
    m_our = copy.deepcopy(m_ref)
    o_ref = torch.optim.AdamW(m_ref.parameters(), lr=0.05, betas=(0.9, 0.95))
    o_our = torch.optim.AdamW(m_our.parameters(), lr=0.05, betas=(0.9, 0.95))
    ref = torch.amp.GradScaler("cpu", init_scale=65536.0, growth_interval=3)
    our = NativeScalerWithGradNormCount(dynamic=True, growth_interval=3)
    assert our.state_dict() == ref.state_dict()
    g = torch.Generator().manual_seed(1)
    skipped = []
    for step in range(9):
        for micro in range(2):
            x = torch.randn(5, 6, generator=g)
            boom = float("inf") if (step in (2, 3) and micro == 1) else 1.0
            upd = micro == 1
            l_ref = m_ref(x).pow(2).mean() * boom / 2
            ref.scale(l_ref).backward()
            if upd:
                ref.unscale_(o_ref)
                n_ref = torch.norm(torch.stack([torch.norm(p.grad.detach(), 2.0) for p in m_ref.parameters()]), 2.0)
                ref.step(o_ref)
                ref.update()
                o_ref.zero_grad()
            l_our = m_our(x).pow(2).mean() * boom / 2
            n_our = our(l_our, o_our, parameters=m_our.parameters(), update_grad=upd)
            if upd:
                o_our.zero_grad()
                assert our.state_dict() == ref.state_dict(), (step, our.state_dict(), ref.state_dict())
                assert (math.isnan(float(n_our)) and math.isnan(float(n_ref))) or float(n_our) == float(n_ref), (step, float(n_our), float(n_ref))
                skipped.append(our.last_found_inf)
                for a, b in zip(m_our.parameters(), m_ref.parameters()):
                    assert torch.equal(a, b), step
            else:
                assert n_our is None
    assert skipped == [False, False, True, True, False, False, False, False, False] and our.skipped_steps == 2
    assert our.get_scale() == ref.get_scale() != 65536.0
    again = NativeScalerWithGradNormCount(dynamic=True)
    again.load_state_dict(ref.state_dict())
    assert again.state_dict() == ref.state_dict()
    plain = NativeScalerWithGradNormCount()
    plain.load_state_dict(ref.state_dict())
    assert plain.get_scale() == 1.0 and set(plain.state_dict()) == set(ref.state_dict())
