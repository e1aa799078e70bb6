"""CPU-only suite (runs in the build container and on any box): the oracle against the committed golden vectors,
the host logic (config, containers, assignment solver, gradient reducer), and the C-ABI library's exports.
No HIP compute is launched here."""
import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
from oracle import tuber_oracle as O                                  # noqa: E402
from tubelet_transformer_amd import lib, synth                        # noqa: E402
from tubelet_transformer_amd.config import get_cfg_defaults, load_cfg  # noqa: E402
from tubelet_transformer_amd.misc import NestedTensor, nested_tensor_from_tensor_list  # noqa: E402
from tubelet_transformer_amd.tuber import build_model                 # noqa: E402

CFGS = ["TubeR_CSN152_AVA21", "TubeR_CSN50_AVA21", "Tuber_CSN152_JHMDB", "TubeR_CSN152_AVA22"]


def cfg_of(name):
    return load_cfg(os.path.join(ROOT, "configuration", name + ".yaml"))


# ---------------------------------------------------------------- oracle vs golden ----------------------------
def _flat(out):
    d = {k: v.detach().numpy() for k, v in out.items() if k != "aux_outputs"}
    for i, a in enumerate(out.get("aux_outputs", [])):
        for k, v in a.items():
            d["aux%d.%s" % (i, k)] = v.detach().numpy()
    return d


@pytest.mark.parametrize("name,yaml_name,sizes", [
    ("csn50_ava21_decode_eval", "TubeR_CSN50_AVA21", [(64, 96)]),
    ("csn152_ava21_avg_eval_ragged", "TubeR_CSN152_AVA21", [(64, 96), (48, 80)]),
    ("csn152_jhmdb_eval", "Tuber_CSN152_JHMDB", [(64, 64)]),
])
def test_oracle_forward_matches_reference_golden(golden_dir, name, yaml_name, sizes):
    gold = np.load(os.path.join(golden_dir, name + ".npz"))
    cfg = cfg_of(yaml_name)
    model, _, _ = build_model(cfg)
    synth.load_name_hashed(model)
    state = {k: v.clone() for k, v in model.state_dict().items()}
    clips = (synth.synthetic_clips(len(sizes), 32, sizes[0][0], sizes[0][1], seed=1234) if len(set(sizes)) == 1
             else synth.synthetic_clips(len(sizes), 32, 0, 0, seed=1234, sizes=sizes))
    with torch.no_grad():
        out = O.tuber_forward(state, cfg, clips, train=False)
    got = _flat(out)
    worst = max(float(np.abs(got[k] - gold[k]).max()) for k in got)
    assert worst <= 1e-5, worst
    tsz = torch.as_tensor(gold["post.target_sizes"])
    pp = O.post_process(cfg, {k: v for k, v in out.items() if k != "aux_outputs"}, tsz)
    for a, k in zip(pp, ("post.scores", "post.boxes", "post.out_b")):
        assert np.abs(a - gold[k]).max() <= 1e-5 * max(1.0, float(np.abs(gold[k]).max()))


def test_spread_fixture_is_decidable_for_the_oracle_and_its_bf16_rounded_execution(golden_dir):
    """The non-degenerate ("spread") train fixture (synth.SPREAD_GAINS + residual_gain 0.05 + structured clips; seeds chosen by
    oracle/gen_golden.py: spread_search): the fp32 oracle AND a bf16-rounded execution of it reproduce the reference's Hungarian
    assignment on every (decoder layer, clip) -- the property the plain name-hashed fixtures lack (a rounded run flips 12 / 12 there)
    -- the queries are spread (boxes >= 0.05), and the golden's stored cost matrices / margins are the oracle's."""
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    from parity_util import run_oracle, matcher_problems, assignment_margin
    name, yaml_name, sizes = "csn50_ava21_decode_train_spread", "TubeR_CSN50_AVA21", [(64, 64), (64, 64)]
    gold = np.load(os.path.join(golden_dir, name + ".npz"))
    cfg = cfg_of(yaml_name)
    model, _, _ = build_model(cfg)
    synth.load_name_hashed(model, residual_gain=0.05, spread=True)
    synth.zero_dropout(model)
    state = {k: v.detach().clone() for k, v in model.state_dict().items()}
    clips = synth.structured_clips(2, 32, sizes[0][0], sizes[0][1], seed=int(gold["clip_seed"]))
    targets = synth.synthetic_targets(2, "ava", cfg.CONFIG.DATA.NUM_CLASSES, seed=int(gold["target_seed"]), hw=sizes[0],
                                      boxes_per_clip=[int(v) for v in gold["boxes_per_clip"]])
    assert float(gold["box_spread"].max()) >= 0.05
    worst = float("inf")
    for rounded in (False, True):
        out, _ = run_oracle(cfg, state, clips, train=True, rounded=rounded)
        for li, per in enumerate(matcher_problems(cfg, out, targets)):
            for b, (C, (qi, ti)) in enumerate(per):
                assert np.array_equal(qi, gold["match.%d.%d.src" % (li, b)]) and np.array_equal(ti, gold["match.%d.%d.tgt" % (li, b)]), (rounded, li, b)
                if not rounded:
                    assert np.abs(C - gold["cost.%d.%d" % (li, b)]).max() <= 1e-4
                    m = assignment_margin(C, (qi, ti))[0]
                    assert abs(m - float(gold["margin.%d.%d" % (li, b)])) <= 1e-3
                    worst = min(worst, float(gold["ratio.%d.%d" % (li, b)]))
    assert worst >= 2.0          # every problem's gap is at least twice what the rounded oracle's noise moves it


@pytest.mark.parametrize("name,yaml_name", [("criterion_ava", "TubeR_CSN152_AVA21"), ("criterion_jhmdb", "Tuber_CSN152_JHMDB")])
def test_oracle_criterion_matches_reference_golden(golden_dir, name, yaml_name):
    gold = np.load(os.path.join(golden_dir, name + ".npz"))
    cfg = cfg_of(yaml_name)
    ava = cfg.CONFIG.DATA.DATASET_NAME == "ava"
    targets = synth.synthetic_targets(3, "ava" if ava else "jhmdb", cfg.CONFIG.DATA.NUM_CLASSES, seed=int(gold["seed"]) + 1,
                                      boxes_per_clip=[1, 4, 2] if ava else None)
    keys = ("pred_logits", "pred_boxes", "pred_logits_b")
    outs = {k: torch.as_tensor(gold["in." + k]) for k in keys}
    outs["aux_outputs"] = [{k: torch.as_tensor(gold["in.aux%d.%s" % (i, k)]) for k in keys} for i in range(5)]
    ld, idx = O.set_criterion(cfg, outs, targets)
    for k in gold.files:
        if k.startswith("loss."):
            assert abs(float(ld[k[5:]]) - float(gold[k])) <= 1e-5 * max(1.0, abs(float(gold[k]))), k
    for li, per in enumerate(idx):
        for b, (i, j) in enumerate(per):
            assert np.array_equal(i.numpy(), gold["match.%d.%d.src" % (li, b)])
    assert abs(float(O.total_loss(cfg, ld)) - float(gold["total_loss"])) <= 1e-4


# ---------------------------------------------------------------- boundary ----------------------------------------
@pytest.mark.parametrize("name", CFGS)
def test_state_dict_and_weight_dict_match_the_reference(golden_dir, name):
    ref = json.load(open(os.path.join(golden_dir, "reference_state_dicts.json")))[name]
    model, crit, post = build_model(cfg_of(name))
    mine = [[k, list(v.shape)] for k, v in model.state_dict().items()]
    assert mine == ref["model"]                      # names, shapes AND order (DDP / optimizer param-group order)
    assert [[k, list(v.shape)] for k, v in crit.state_dict().items()] == ref["criterion"]
    assert {k: float(v) for k, v in crit.weight_dict.items()} == ref["weight_dict"]
    assert set(post) == {"bbox"}
    names = [n for n, _ in model.named_parameters()]
    assert any("backbone" in n for n in names) and any("class_embed" in n for n in names) and any("query_embed" in n for n in names)


def test_drop_in_import_paths():
    from models.tuber_ava import build_model as b1
    from models.tuber_jhmdb import build_model as b2
    from pipelines.video_action_recognition_config import get_cfg_defaults as g
    from utils.misc import NestedTensor as N2
    assert b1 is build_model and b2 is build_model and g is get_cfg_defaults and N2 is NestedTensor


def test_product_path_fails_loudly_without_a_gpu():
    model, _, _ = build_model(cfg_of("TubeR_CSN152_AVA21"))
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        model(torch.zeros(1, 3, 32, 32, 32))
    with pytest.raises(RuntimeError):
        model.backbone.body(torch.zeros(1, 3, 32, 32, 32))


def test_config_loader_yacs_semantics(tmp_path):
    cfg = cfg_of("TubeR_CSN50_AVA21")
    assert isinstance(cfg.CONFIG.TRAIN.LR, float) and cfg.CONFIG.TRAIN.LR == 1e-4          # '1e-4' literal_eval'd like yacs
    assert cfg.CONFIG.MODEL.BACKBONE_NAME == "CSN-50" and cfg.CONFIG.MODEL.TEMPORAL_DS_STRATEGY == "decode"
    c2 = cfg.clone()
    c2.CONFIG.MODEL.QUERY_NUM = 3
    assert cfg.CONFIG.MODEL.QUERY_NUM == 15
    c2.CONFIG.NEW_KEY = 5                                                                  # CONFIG nodes are new_allowed
    with pytest.raises(KeyError):
        d = get_cfg_defaults()
        d.merge_from_other_cfg({"DDP_CONFIG": {"NOT_A_KEY": 1}})
    cfg.freeze()
    with pytest.raises(AttributeError):
        cfg.CONFIG.EVAL_ONLY = True
    p = tmp_path / "c.yaml"
    p.write_text(cfg.dump())
    again = get_cfg_defaults()
    again.merge_from_file(str(p))
    assert again.CONFIG.MODEL.QUERY_NUM == 15


def test_nested_tensor_padding_and_mask():
    a, b = torch.ones(3, 4, 5, 7), 2 * torch.ones(3, 4, 6, 3)
    nt = nested_tensor_from_tensor_list([a, b])
    assert nt.tensors.shape == (2, 3, 4, 6, 7) and nt.mask.shape == (2, 6, 7)
    assert not nt.mask[0, :5, :7].any() and nt.mask[0, 5:].all()
    assert not nt.mask[1, :6, :3].any() and nt.mask[1, :, 3:].all()
    assert float(nt.tensors[1, :, :, :, 3:].abs().sum()) == 0
    o_t, o_m = O.nested_from_list([a, b])
    assert torch.equal(o_t, nt.tensors) and torch.equal(o_m, nt.mask)
    with pytest.raises(ValueError):
        nested_tensor_from_tensor_list([torch.zeros(3, 3)])


# ---------------------------------------------------------------- C ABI --------------------------------------------
def test_library_exports_every_symbol_the_header_declares():
    protos = lib.header_prototypes()
    assert len(protos) >= 50
    L = lib.load()                      # binds every prototype; AttributeError = header/library drift
    for _, name, _ in protos:
        assert hasattr(L, name), name
    # ... and the header declares every extern "C" entry point of the sources (a stale header = a KeyError at the first lib.call of the new one)
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_header", os.path.join(ROOT, "tubelet_transformer_amd", "csrc", "gen_header.py"))
    gh = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gh)
    declared = {name for _, name, _ in protos}
    defined = {name for _, _, name, _ in gh.prototypes()}
    assert defined == declared, "include/tuber_hip.h is stale: run python tubelet_transformer_amd/csrc/gen_header.py (%s)" % sorted(defined ^ declared)
    # pure host helpers may be called without a GPU
    cfg = lib.query("tuber_gemm_nt_cfg", 348160, 64, 256)
    assert cfg == 13                                     # 64x64 tiles, two-tile prefetch
    assert lib.query("tuber_gemm_nt_stat_rows", 348160, 64) == 348160 // 64
    assert lib.query("tuber_gemm_nt_cfg", 30, 256, 256) == 13
    assert lib.query("tuber_gemm_nt_cfg", 348160, 256, 64) == 7      # 64x128
    assert lib.query("tuber_gemm_nt_cfg", 16896, 2048, 512) == 0     # 128x128: the class-branch FFN


def test_lsap_matches_scipy_including_ties():
    from scipy.optimize import linear_sum_assignment
    from tubelet_transformer_amd.criterion import _lsap
    rng = np.random.default_rng(0)
    for trial in range(3000):
        nr, nc = int(rng.integers(1, 17)), int(rng.integers(1, 8))
        mode = trial % 4
        if mode == 0:
            c = rng.standard_normal((nr, nc))
        elif mode == 1:
            c = rng.integers(0, 3, (nr, nc)).astype(float)
        elif mode == 2:                       # constant across targets, like the AVA class cost (matcher.py:72)
            c = np.repeat(rng.integers(0, 4, (nr, 1)).astype(float), nc, 1)
        else:
            c = rng.integers(0, 2, (nr, nc)).astype(float) + np.repeat(rng.integers(0, 3, (nr, 1)).astype(float), nc, 1)
        if trial % 7 == 0:
            c = np.ascontiguousarray(c.T)
        a, b = linear_sum_assignment(c)
        i, j = _lsap(c)
        assert np.array_equal(a, i) and np.array_equal(b, j)
    i, j = _lsap(np.zeros((5, 0)))
    assert len(i) == 0 and len(j) == 0


# ---------------------------------------------------------------- data-parallel reducer (gloo, 2 processes) ------------
def _reducer_worker(rank, world, port, q):
    import torch.distributed as dist
    from tubelet_transformer_amd.ddp import FlatGradReducer, broadcast_parameters
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    class M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.a = torch.nn.Linear(100, 70)
            self.query_embed = torch.nn.Embedding(15, 64)
            self.b = torch.nn.Linear(70, 33)
    torch.manual_seed(rank)
    m = M()
    names = [n for n, _ in m.named_parameters()]
    store = type("S", (), {})()
    store.module, store.names, store.params = m, names, [p for _, p in m.named_parameters()]
    off, store.offsets = 0, {}
    for n, p in zip(names, store.params):
        store.offsets[n] = off
        off += (p.numel() + 63) // 64 * 64
    store.total = off
    store.flat = torch.randn(off)
    store.gflat = torch.zeros(off)
    broadcast_parameters(store)
    red = FlatGradReducer(store, min_bucket=64)
    g = torch.Generator().manual_seed(100 + rank)
    local = torch.randn(off, generator=g)
    red.begin()
    # backward fills the flat buffer from its high end down; notify after each "stage"
    cuts = [store.offsets["b.weight"], store.offsets["query_embed.weight"], store.offsets["a.bias"], 0]
    hi = off
    for c in cuts:
        store.gflat[c:hi] = local[c:hi]
        if c:
            red.notify(c)
        hi = c
    red.finish()
    q.put((rank, store.flat.numpy().copy(), store.gflat.numpy().copy(), local.numpy().copy()))   # by value, not shared memory
    dist.destroy_process_group()


def test_flat_gradient_reducer_gloo_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_reducer_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
    (_, f0, g0, l0), (_, f1, g1, l1) = res
    assert np.array_equal(f0, f1)                                # parameters broadcast from rank 0
    mean = (l0 + l1) / 2
    assert np.allclose(g0, mean, atol=1e-6) and np.allclose(g1, mean, atol=1e-6)   # every slice reduced exactly once


# ---------------------------------------------------------------- checkpoints -----------------------
def test_checkpoint_roundtrip_with_ddp_prefix(tmp_path):
    from tubelet_transformer_amd import checkpoint as ck
    cfg = cfg_of("TubeR_CSN50_AVA21")
    cfg.CONFIG.MODEL.BACKBONE_NAME = "CSN-TEST"
    cfg.CONFIG.LOG.BASE_PATH, cfg.CONFIG.LOG.EXP_NAME = str(tmp_path), "exp"
    m1, _, _ = build_model(cfg)
    synth.load_name_hashed(m1, salt=3)
    path = ck.save_checkpoint(cfg, 7, m1, 0.0, None, None)
    saved = torch.load(path, weights_only=False)
    assert all(k.startswith("module.") for k in saved["model"]) and saved["epoch"] == 7      # interchangeable with the reference
    m2, _, _ = build_model(cfg)
    cfg.CONFIG.MODEL.PRETRAINED_PATH = path
    ck.load_model(m2, cfg)
    for (k1, v1), (k2, v2) in zip(m1.state_dict().items(), m2.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2), k1
    # DETR initialisation: transformer.*, bbox_embed.*, first QUERY_NUM query rows from a DDP-saved (module.-prefixed) file
    detr = {"model": {"module." + k: v.clone() for k, v in m1.state_dict().items() if k.startswith(("transformer.", "bbox_embed."))}}
    detr["model"]["module.query_embed.weight"] = torch.arange(100 * 256, dtype=torch.float32).view(100, 256)
    p2 = str(tmp_path / "detr.pth")
    torch.save(detr, p2)
    m3, _, _ = build_model(cfg)
    ck.load_detr_weights(m3, p2, cfg)
    assert torch.equal(m3.query_embed.weight, detr["model"]["module.query_embed.weight"][:15])
    assert torch.equal(m3.transformer.decoder.norm.weight, m1.transformer.decoder.norm.weight)


def test_csn_mat_loader_maps_caffe2_names_and_freezes(tmp_path):
    import scipy.io as sio
    from tubelet_transformer_amd import checkpoint as ck
    from tubelet_transformer_amd.backbone import ResNeXt
    body = ResNeXt([2, 2, 2, 2])
    rng = np.random.default_rng(0)
    mat, count = {}, 0

    def bn(name, c):
        for sfx in ("_s", "_b", "_rm", "_riv"):
            mat[name + sfx] = rng.standard_normal((1, c)).astype(np.float32)
    mat["conv1_w"] = rng.standard_normal((64, 3, 3, 7, 7)).astype(np.float32)
    bn("conv1_spatbn_relu", 64)
    for stage in (body.layer1, body.layer2, body.layer3, body.layer4):
        for blk in stage:
            for j, conv in ((1, blk.conv1), (3, blk.conv3), (4, blk.conv4)):
                mat["comp_%d_conv_%d_w" % (count, j)] = rng.standard_normal(tuple(conv.weight.shape)).astype(np.float32)
                bn("comp_%d_spatbn_%d" % (count, j), conv.weight.shape[0])
            if blk.down_sample is not None:
                mat["shortcut_projection_%d_w" % count] = rng.standard_normal(tuple(blk.down_sample[0].weight.shape)).astype(np.float32)
                bn("shortcut_projection_%d_spatbn" % count, blk.down_sample[0].weight.shape[0])
            count += 1
    path = str(tmp_path / "csn.mat")
    sio.savemat(path, mat)
    ck.load_csn_mat(body, path, "CSN-TEST", verbose=False)
    assert np.allclose(body.conv1.weight.detach().numpy(), mat["conv1_w"])
    assert np.allclose(body.layer3[1].conv3.weight.detach().numpy(), mat["comp_5_conv_3_w"])
    assert np.allclose(body.layer2[0].down_sample[1].running_var.numpy(), mat["shortcut_projection_2_spatbn_riv"].reshape(-1))
    assert np.allclose(body.layer4[1].bn4.bias.detach().numpy(), mat["comp_7_spatbn_4_b"].reshape(-1))
    # tune_point 4 (build_CSN): stem, layer1, layer2 frozen; layer3, layer4 trainable (ir_CSN_152.py:251-254,301-303)
    assert not body.conv1.weight.requires_grad and not body.layer2[1].bn3.weight.requires_grad
    assert body.layer3[0].conv1.weight.requires_grad and body.layer4[1].bn4.bias.requires_grad


def test_frame_map_matches_reference_evaluator_golden(tmp_path):
    """row N2: the numpy frame-mAP against the reference's own evaluator (evaluates/evaluate_ava.py over the vendored PASCAL
    evaluator) on synthetic result files with score ties, duplicate detections, invalid boxes, frames without ground truth and
    classes without ground truth (oracle/gen_eval_golden.py -> tests/golden/frame_map_case.json): identical to the last bit."""
    import json
    from tubelet_transformer_amd.evaluation import FrameMAP, write_result_files
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "frame_map_case.json")))
    (tmp_path / "GT_0.txt").write_text("\n".join(g["gt_lines"]) + "\n")
    (tmp_path / "0.txt").write_text("\n".join(g["det_lines"]) + "\n")
    ev = FrameMAP(g["class_num"])
    ev.load_gt([str(tmp_path / "GT_0.txt")])
    ev.load_detections([str(tmp_path / "0.txt")])
    mAP, per_class = ev.evaluate()
    assert mAP == g["mAP"]
    for k, v in g["per_class_ap"].items():
        c = int(k[1:])
        if v is None:
            assert c not in per_class                     # classes without ground truth do not enter the mean
        else:
            assert per_class[c] == v, (k, per_class[c], v)
    # the writer produces lines the parser (and the reference's ``line.split(' [')`` parser) reads back unchanged
    K = g["class_num"]
    rng = np.random.default_rng(1)
    det_ids = ["a_1", "a_1", "b_2"]
    boxes, sc, bn = rng.uniform(0, 100, (3, 4)), rng.uniform(0, 1, (3, K)), rng.uniform(0, 1, (3, 3))
    gt_b, gt_l = rng.uniform(0, 100, (2, 6)), (rng.uniform(0, 1, (2, K)) > 0.5).astype(np.float64)
    dp, gp = write_result_files(str(tmp_path), "res", 0, det_ids, boxes, sc, bn, ["a_1", "b_2"], gt_b, gt_l)
    lines = open(dp).read().splitlines()
    assert len(lines) == 3 and lines[0].startswith("a_1 [") and lines[0].endswith("]")
    vals = [float(x) for x in lines[2].split(" [")[1].split("]")[0].split(",")]
    assert np.array_equal(np.asarray(vals), np.concatenate([boxes[2], sc[2], bn[2]]))
    assert len(open(gp).read().splitlines()) == 2


def test_frame_map_ucf_matches_reference_evaluator(tmp_path):
    """FrameMAPUCF == the reference's STDetectionEvaluaterUCF (evaluates/evaluate_ucf.py) on JHMDB-style result files with a tiny
    (< 10 px^2) ground-truth box, frames without ground truth, 'no object on top' detections and score ties: same bits."""
    from tubelet_transformer_amd.evaluation import FrameMAPUCF
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "frame_map_ucf_case.json")))
    gp, dp = str(tmp_path / "GT_0.txt"), str(tmp_path / "0.txt")
    open(gp, "w").write("\n".join(g["gt_lines"]) + "\n")
    open(dp, "w").write("\n".join(g["det_lines"]) + "\n")
    ev = FrameMAPUCF(class_num=g["class_num"])
    ev.load_gt([gp])
    ev.load_detections([dp])
    mAP, per_class = ev.evaluate()
    assert mAP == g["mAP"]
    want = [v for v in g["per_class_ap"].values()]
    for cls in range(1, 25):
        assert per_class.get(cls) == want[cls - 1], cls


def _launch_main(cfg):
    import torch.distributed as dist
    t = torch.ones(1) * (cfg.DDP_CONFIG.GPU_WORLD_RANK + 1)
    if cfg.DDP_CONFIG.DISTRIBUTED:
        dist.all_reduce(t)
    open(os.path.join(cfg.CONFIG.LOG.BASE_PATH, "rank%d.txt" % cfg.DDP_CONFIG.GPU_WORLD_RANK), "w").write(
        "%d %d %d" % (int(t), cfg.DDP_CONFIG.GPU, cfg.DDP_CONFIG.GPU_WORLD_SIZE))
    if cfg.DDP_CONFIG.DISTRIBUTED:
        dist.destroy_process_group()


def test_spawn_workers_and_reference_call_forms(tmp_path):
    """pipelines/launch.py:spawn_workers(main, cfg) with the reference's argument list: one worker process per device, process
    group from DDP_CONFIG (gloo here, two CPU workers), main(cfg) with GPU / GPU_WORLD_RANK / GPU_WORLD_SIZE filled in; and the
    signatures the reference scripts call: deploy_model(model, cfg, is_tuber=True), validate_tuber_ucf_detection(cfg, model,
    criterion, postprocessors, data_loader, epoch, writer)."""
    import inspect
    from pipelines.launch import spawn_workers
    from utils.model_utils import deploy_model, load_detr_weights, load_model, save_checkpoint  # noqa: F401
    from utils.video_action_recognition import train_tuber_detection, validate_tuber_detection, validate_tuber_ucf_detection  # noqa: F401
    assert list(inspect.signature(deploy_model).parameters)[:3] == ["model", "cfg", "is_tuber"]
    assert list(inspect.signature(validate_tuber_ucf_detection).parameters)[:7] == ["cfg", "model", "criterion", "postprocessors", "data_loader", "epoch", "writer"]
    assert list(inspect.signature(spawn_workers).parameters)[:2] == ["main", "cfg"]
    cfg = cfg_of("TubeR_CSN50_AVA21")
    cfg.CONFIG.LOG.BASE_PATH = str(tmp_path)
    cfg.DDP_CONFIG.DISTRIBUTED = True
    cfg.DDP_CONFIG.DIST_BACKEND = "gloo"
    cfg.DDP_CONFIG.DIST_URL = "tcp://127.0.0.1:%d" % (23000 + os.getpid() % 2000)
    cfg.DDP_CONFIG.WORLD_SIZE, cfg.DDP_CONFIG.WORLD_RANK = 1, 0
    spawn_workers(_launch_main, cfg, nprocs=2)
    got = sorted(open(str(tmp_path / ("rank%d.txt" % r))).read() for r in range(2))
    assert got == ["3 0 2", "3 1 2"], got
    cfg.DDP_CONFIG.DISTRIBUTED = False
    cfg.DDP_CONFIG.GPU, cfg.DDP_CONFIG.GPU_WORLD_RANK = 0, 0
    spawn_workers(_launch_main, cfg, nprocs=1)
    assert open(str(tmp_path / "rank0.txt")).read().split()[0] == "1"


# ---------------------------------------------------------------- N1 pinned against the REFERENCE's own loaders --------
@pytest.mark.parametrize("case", ["csn152", "csn50"])
def test_weight_import_matches_the_reference_loaders(golden_dir, tmp_path, case):
    """tests/golden/weight_import.json holds what the reference's build_CSN / load_weights (ir_CSN_152.py:213-318, ir_CSN_50.py),
    load_model (utils/model_utils.py:66-95) and load_detr_weights (:10-36) produce from the seeded files of tests/weight_files.py
    (generated by oracle/gen_weight_import_golden.py with the unmodified reference).  checkpoint.py must write the same bytes into
    the same tensors, leave the same tensors untouched, and leave the same requires_grad pattern behind."""
    import weight_files as WF
    from tubelet_transformer_amd import checkpoint as ck
    from tubelet_transformer_amd.tuber import build_model
    gold = json.load(open(os.path.join(golden_dir, "weight_import.json")))
    g = gold[case + "_mat"]
    cfg = load_cfg(os.path.join(ROOT, "configuration", g["yaml"]))
    # Caffe2 .mat through build_model (CONFIG.MODEL.PRETRAINED) -- block offsets, _riv -> running_var, freeze pattern
    cfg.CONFIG.MODEL.PRETRAINED = True
    cfg.CONFIG.MODEL.PRETRAIN_BACKBONE_DIR = WF.write_csn_mat(str(tmp_path / "w.mat"), g["backbone"], g["seed"])
    model, _, _ = build_model(cfg)
    snap = WF.snapshot(model)
    for k, (c, rg) in g["body"].items():
        assert snap[k][0] == c, "%s: bytes differ from what the reference's load_weights wrote" % k
    assert {n: bool(p.requires_grad) for n, p in model.named_parameters()} == g["requires_grad"]
    # TubeR checkpoint saved from a DDP model
    cfg.CONFIG.MODEL.PRETRAINED = False
    model, _, _ = build_model(cfg)
    gc = gold[case + "_ckpt"]
    cfg.CONFIG.MODEL.PRETRAINED_PATH = WF.write_tuber_checkpoint(str(tmp_path / "c.pth"), model.state_dict(), gc["seed"])
    before = WF.snapshot(model)
    ck.load_model(model, cfg)
    after = WF.snapshot(model)
    assert len(after) == gc["total"]
    assert {k: v for k, v in after.items() if before[k] != v} == {k: v for k, v in gc["changed"].items()}
    # DETR initialisation files: 'module.' loads transformer / bbox_embed / sliced query_embed, 'detr.' loads nothing
    for prefix in ("module", "detr"):
        gd = gold["%s_detr_%s" % (case, prefix)]
        model, _, _ = build_model(cfg)
        path = WF.write_detr_checkpoint(str(tmp_path / ("d_%s.pth" % prefix)), model.state_dict(), gd["seed"], prefix)
        before = WF.snapshot(model)
        ck.load_detr_weights(model, path, cfg)
        after = WF.snapshot(model)
        assert {k: v for k, v in after.items() if before[k] != v} == gd["changed"], prefix
        assert (len(gd["changed"]) > 0) == (prefix == "module")


def test_bench_roofline_bookkeeping_matches_trace_names():
    """bench.py picks the dominant kernel family from the committed rocprofv3 kernel-trace summary and looks its PMC numbers up by name:
    the trace carries trailing template parameters the C-ABI bookkeeping does not know (a GEMM family was silently skipped for that
    reason in round 3) -- the committed files must resolve for a GEMM key, a plain key and a templated non-GEMM key."""
    import glob
    import importlib.util
    root = os.path.join(os.path.dirname(__file__), "..")
    spec = importlib.util.spec_from_file_location("tuber_bench", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    files = sorted(f for f in glob.glob(os.path.join(root, "profiles", "r*_kernel_trace_stats.txt")) if "_cfg" not in f and "_freeze_" not in f)
    assert files, "no committed kernel-trace summary"
    one = lambda k: {k: {"bytes": 1, "ms": 1.0, "launches": 1, "flops": 0}}
    for key in ("gemm_nt_kernel<64,128,1,4,2,0,7,3>", "gemm_nt_kernel<96,64,2,2,2,0,1,4>", "bn_bwd_fa_kernel", "gemm_tn3_group_kernel", "conv1_bwd_kernel<1>", "dwconv_tile_bwd_both_kernel"):
        assert bench.dominant_from_trace(one(key), True) == key, key
    assert bench.dominant_from_trace(one("no_such_kernel"), True) is None
    assert bench.dominant_from_trace(one("bn_bwd_fa_kernel"), False) is None          # other configs: no committed trace to rank by
    # every launcher the step calls has a bytes / flops entry or is deliberately uncounted ("...*")
    k, by, fl = bench.alg_cost("tuber_entry_conv_fwd", [None] * 11 + [6400])
    assert k == "entry_conv_kernel" and by == 2 * 6400 * 384 and fl == 2 * 6400 * 64 * 320


def test_weight_gradient_gemm_routing_heuristics():
    """host-side tile / slab choice of the dW GEMMs (csrc/gemm.hip: tn_big, tn_slabs_wanted) for the shapes the measurements in
    DESIGN.md section 3 were taken on -- a regression guard for the routing, callable without a GPU"""
    q = lib.query
    # layer3 conv1 / conv4 (M = 5632): 128 x 128 tiles, 2 slabs of 2816 rows (round 5, groups of 16 problems; 4 slabs of 1408 before)
    assert q("tuber_gemm_tn_tile", 5632, 1024, 256) == 128 and q("tuber_gemm_tn_slabs", 5632, 1024, 256) == 2
    assert q("tuber_gemm_tn_tile", 5632, 256, 1024) == 128
    # layer4 (M = 2816): big tiles, 2 slabs
    assert q("tuber_gemm_tn_tile", 2816, 2048, 512) == 128 and q("tuber_gemm_tn_slabs", 2816, 2048, 512) == 2
    # class-branch FFN pair (M = 16896): big tiles with 8 slabs of 2112 rows; its small linears stay on 64 x 64
    assert q("tuber_gemm_tn_tile", 16896, 256, 2048) == 128 and q("tuber_gemm_tn_slabs", 16896, 256, 2048) == 8
    assert q("tuber_gemm_tn_tile", 16896, 2048, 512) == 128 and q("tuber_gemm_tn_slabs", 16896, 2048, 512) == 8
    assert q("tuber_gemm_tn_tile", 16896, 256, 256) == 64
    # short M (encoder FFN, layer4's down-sample projection over 704 rows): 64 x 64 tiles, groupable, bias gradient fusable
    assert q("tuber_gemm_tn_tile", 704, 2048, 256) == 64
    assert q("tuber_gemm_tn_tile", 704, 2048, 1024) == 64 and q("tuber_gemm_tn_fuses_bias", 704, 2048, 1024, 2048, 1024) in (1, 2)
    # layer1 / layer2 (N or K below 128 or not a multiple of it): small tiles
    assert q("tuber_gemm_tn_tile", 348160, 64, 256) == 64 and q("tuber_gemm_tn_tile", 87040, 128, 512) == 64
    # the fused layer1 kernels' shape gates
    assert q("tuber_conv4_bwd_supported", 256, 64) == 1 and q("tuber_conv4_bwd_supported", 512, 128) == 0
    assert q("tuber_conv1_bwd_supported", 256, 64) == 1 and q("tuber_conv1_bwd_supported", 64, 64) == 0
    assert q("tuber_blockout_conv1_supported", 256, 128) == 1 and q("tuber_entry_conv_supported", 64, 64, 256) == 1


# ---------------------------------------------------------------- the reference's entry scripts import unchanged ---------------------------
_IMPORT_BLOCK_PROBE = r'''
import sys, types
repo, ref, script = sys.argv[1], sys.argv[2], sys.argv[3]
sys.path[:0] = [repo, ref]                       # INTEGRATION.md section A: this repository first, the reference checkout behind it
def mod(name, **attrs):
    m = types.ModuleType(name); m.__dict__.update(attrs); m.__path__ = []; sys.modules[name] = m; return m
# third-party packages the image lacks (nothing of the reference is stubbed)
mod("tensorboardX", SummaryWriter=object)
mod("timm"); mod("timm.scheduler")
mod("timm.scheduler.cosine_lr", CosineLRScheduler=object); mod("timm.scheduler.step_lr", StepLRScheduler=object)
mod("timm.scheduler.scheduler", Scheduler=object)
mod("yacs"); mod("yacs.config", CfgNode=dict)
mod("cv2")
import torch
tv = mod("torchvision", __version__="0.15.0")
mod("torchvision.transforms"); mod("torchvision.transforms.functional")
mod("torchvision.ops"); mod("torchvision.ops.boxes", box_area=None); mod("torchvision.ops.misc", interpolate=None)
mod("torchvision.models"); mod("torchvision.models._utils", IntermediateLayerGetter=object)
mod("torchvision.models.video"); mod("torchvision.models.video.resnet", VideoResNet=object)
block = []
for line in open(ref + "/" + script):
    if line.startswith("def "):
        break
    block.append(line)
ns = {}
exec(compile("".join(block), script, "exec"), ns)
import tubelet_transformer_amd.tuber as T, tubelet_transformer_amd.training as TR, tubelet_transformer_amd.evaluation as EV
import tubelet_transformer_amd.launch as L, tubelet_transformer_amd.config as C, tubelet_transformer_amd.checkpoint as CK
assert ns["build_model"] is T.build_model
assert ns["deploy_model"] is TR.deploy_model and ns["load_model"] is CK.load_model
assert ns["spawn_workers"] is L.spawn_workers and ns["get_cfg_defaults"] is C.get_cfg_defaults
if "train_tuber_detection" in ns: assert ns["train_tuber_detection"] is TR.train_tuber_detection
if "validate_tuber_detection" in ns: assert ns["validate_tuber_detection"] is EV.validate_tuber_detection
if "validate_tuber_ucf_detection" in ns: assert ns["validate_tuber_ucf_detection"] is EV.validate_tuber_ucf_detection
# the sub-modules this repository does NOT provide come from the reference checkout
assert ns["build_log_dir"].__code__.co_filename.startswith(ref), ns["build_log_dir"].__code__.co_filename
assert ns["build_dataloader"].__code__.co_filename.startswith(ref)
if "build_scheduler" in ns: assert ns["build_scheduler"].__code__.co_filename.startswith(ref)
print("IMPORT-BLOCK-OK", script, sorted(k for k in ns if not k.startswith("__")))
'''


@pytest.mark.parametrize("script", ["train_tuber_ava.py", "eval_tuber_ava.py", "train_tuber_jhmdb.py", "eval_tuber_jhmdb.py"])
def test_reference_entry_scripts_import_with_this_repo_first_on_the_path(script):
    """VERDICT r03 missing #1: with the repository first on ``PYTHONPATH`` the reference's scripts must get past their import block
    (``train_tuber_ava.py:1-16`` etc.): ``utils.utils``, ``utils.lr_scheduler``, ``datasets.*`` come from the reference checkout, the hot
    path (``build_model``, ``deploy_model``, ``train_tuber_detection`` ...) from this repository.  Build container only."""
    import subprocess
    ref = "/root/reference"
    if not os.path.isfile(os.path.join(ref, script)):
        pytest.skip("reference checkout absent (GPU box)")
    r = subprocess.run([sys.executable, "-c", _IMPORT_BLOCK_PROBE, ROOT, ref, script], capture_output=True, text=True, timeout=600,
                       cwd="/tmp", env={k: v for k, v in os.environ.items() if k != "PYTHONPATH"})
    assert r.returncode == 0 and "IMPORT-BLOCK-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_deferred_reduce_overwrites_only_windows_zero_grad_has_cleared():
    """engine.DeferredReduce (host logic, round 6): the head entry of a gradient window carries mode bit 3 (tuber_multi_reduce: out = sum, the
    window is not read) only when zero_grad ran and nothing has been reduced into that window since; every other case keeps out += sum --
    gradients cleared some other way, a second reduction into the same window (shared parameter in a later flush, gradient accumulation
    over two backward passes), TUBER_AB=no_fresh_reduce."""
    from tubelet_transformer_amd import ab
    from tubelet_transformer_amd.engine import DeferredReduce
    d = DeferredReduce(torch.device("cpu"))
    A, B, P = 0x1000, 0x9000, 0x100000
    d.add(P, A, 4096, 4096, 2, 0)
    assert d.entries[0][5] & 8 == 0, "nobody called zero_grad: accumulate"
    d.entries, d.outs, d.heads = [], {}, []
    d.fresh = set()                                             # what ParamStore.zero_grad does
    d.add(P, A, 4096, 4096, 2, 0)
    d.add(P, A, 4096, 4096, 2, 0)                               # a second use of the same parameter in the same flush: chained behind the head
    d.add(P, B, 128, 128, 4, 1)
    heads = [d.entries[h] for h in d.heads]
    assert [e[1] for e in heads] == [A, B] and all(e[5] & 8 for e in heads)
    assert d.entries[0][7] == 1 and d.entries[1][5] & 8 == 0, "the chained contribution accumulates on the head's sum"
    assert (heads[0][5] & 7, heads[1][5] & 7) == (2, 1), "the float4 form of mode 0 and the tree form keep their meaning under the flag"
    d.entries, d.outs, d.heads = [], {}, []                     # (a flush)
    d.add(P, A, 4096, 4096, 2, 0)                               # the same window again before the next zero_grad: accumulate
    assert d.entries[0][5] & 8 == 0
    d.entries, d.outs, d.heads = [], {}, []
    d.fresh = set()
    with ab.override("no_fresh_reduce"):
        d.add(P, A, 4096, 4096, 2, 0)
    assert d.entries[0][5] & 8 == 0
