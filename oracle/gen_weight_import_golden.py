"""Pin row N1 (checkpoint / pretrained-weight import) against the REFERENCE's own loaders -- build container only:

    python oracle/gen_weight_import_golden.py        ->  tests/golden/weight_import.json

Seeded weight files (tests/weight_files.py) are fed to the UNMODIFIED reference code (oracle/ref_import.py):
  * ``build_model(cfg)`` with ``PRETRAINED: True`` -> ``build_CSN`` -> ``load_weights`` (models/backbones/ir_CSN_152.py:213-318,
    ir_CSN_50.py: block offsets [0,3,11,47] / [0,3,7,13], ``_riv`` -> running_var, the tune_point = 4 freeze pattern);
  * ``load_model`` (utils/model_utils.py:66-95) on a model whose state_dict keys carry ``module.`` (the reference wraps in DDP);
  * ``load_detr_weights`` (utils/model_utils.py:10-36: the ``k.split('.')[1]`` rule, query_embed row slicing) with a ``module.``- and a
    ``detr.``-prefixed file.
What is stored: per tensor of the resulting model, crc32 of its bytes and ``requires_grad`` -- for the tensors the loader wrote; the
others (random initialisation) are recorded as "untouched" by comparing with a snapshot taken before the load.
"""
import contextlib
import io
import json
import os
import sys
import tempfile

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import ref_import                                          # noqa: E402
import weight_files as WF                                              # noqa: E402

CASES = {"csn152": ("TubeR_CSN152_AVA21.yaml", "CSN-152", 11), "csn50": ("TubeR_CSN50_AVA21.yaml", "CSN-50", 12)}


class Wrapped(torch.nn.Module):
    """what DistributedDataParallel does to the key names: everything under ``module.``"""

    def __init__(self, m):
        super().__init__()
        self.module = m


def strip(sd):
    return {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}


def changed(before, after):
    """{name: [crc, requires_grad]} for entries the load changed (or whose requires_grad it changed); the rest is 'untouched'"""
    return {k: v for k, v in after.items() if before.get(k) != v}


def main():
    assert ref_import.available()
    out = {}
    tmp = tempfile.mkdtemp()
    for case, (yaml_name, bname, seed) in CASES.items():
        # ---- Caffe2 .mat through the reference's build_model / build_CSN / load_weights --------------------------------------------
        mat = WF.write_csn_mat(os.path.join(tmp, case + ".mat"), bname, seed)
        cfg = ref_import.ref_cfg(yaml_name)
        cfg.CONFIG.MODEL.PRETRAINED = True
        cfg.CONFIG.MODEL.PRETRAIN_BACKBONE_DIR = mat
        model, _, _ = ref_import.build_reference(cfg)
        snap = WF.snapshot(model)
        body = {k: v for k, v in snap.items() if k.startswith("backbone.body.") and "out_fc" not in k and "num_batches_tracked" not in k}
        out[case + "_mat"] = {"seed": seed, "backbone": bname, "yaml": yaml_name, "body": body,
                              "requires_grad": {n: bool(p.requires_grad) for n, p in model.named_parameters()}}
        print(case, ".mat: body tensors", len(body), "frozen params", sum(1 for v in out[case + "_mat"]["requires_grad"].values() if not v))
        # ---- TubeR checkpoint (module. prefix) through the reference's load_model ------------------------------------------------
        cfg = ref_import.ref_cfg(yaml_name)
        model, _, _ = ref_import.build_reference(cfg)
        ck = WF.write_tuber_checkpoint(os.path.join(tmp, case + "_ckpt.pth"), model.state_dict(), seed + 100)
        cfg.CONFIG.MODEL.PRETRAINED_PATH = ck
        cfg.DDP_CONFIG.GPU = None
        before = WF.snapshot(model)
        with ref_import.reference_on_path():
            from utils.model_utils import load_model, load_detr_weights
            with contextlib.redirect_stdout(io.StringIO()):
                load_model(Wrapped(model), cfg)
            after = WF.snapshot(model)
            out[case + "_ckpt"] = {"seed": seed + 100, "changed": changed(before, after), "total": len(after)}
            print(case, "checkpoint: tensors written", len(out[case + "_ckpt"]["changed"]), "of", len(after))
            # ---- DETR files through the reference's load_detr_weights --------------------------------------------------------------
            for prefix in ("module", "detr"):
                model, _, _ = ref_import.build_reference(cfg)
                dp = WF.write_detr_checkpoint(os.path.join(tmp, "%s_detr_%s.pth" % (case, prefix)), model.state_dict(), seed + 200, prefix)
                before = WF.snapshot(model)
                with contextlib.redirect_stdout(io.StringIO()):
                    load_detr_weights(Wrapped(model), dp, cfg)
                after = WF.snapshot(model)
                out["%s_detr_%s" % (case, prefix)] = {"seed": seed + 200, "changed": changed(before, after), "total": len(after)}
                print(case, "detr file with prefix %r: tensors written" % prefix, len(out["%s_detr_%s" % (case, prefix)]["changed"]))
    path = os.path.join(ROOT, "tests", "golden", "weight_import.json")
    json.dump(out, open(path, "w"), indent=0, sort_keys=True)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
