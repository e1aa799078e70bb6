"""A/B switches of the product path, behind ONE parsed list: ``TUBER_AB=name[,name...]``.

Every entry turns one fused / grouped form of the hot path back into the separate kernels it replaced, so that a claim in DESIGN.md
("the fused conv4 backward is worth 0.2 ms") can be re-measured on the same build.  The alternative paths are part of the shipped
library, so they are covered by the GPU suite the driver runs: ``tests/test_training_gpu.py::test_every_ab_switch_...`` is
parametrised over ``KNOWN`` and holds every switch to the default path's gradients.  Unknown names are an error, not ignored.
Measured-and-rejected paths are not kept behind switches; they are deleted ("Measured and rejected")."""
import contextlib
import os

KNOWN = {
    "no_wgrad_groups": "one tuber_gemm_tn launch per weight-gradient GEMM instead of the grouped launches (engine.WgradQueue)",
    "immediate_reduce": "second-stage reductions of the weight-gradient partials per call instead of the deferred tuber_multi_reduce",
    "no_join_fusion": "stand-alone block_out_bwd instead of the join backward in the conv1 data-gradient GEMM's epilogue",
    "no_fresh_reduce": "tuber_multi_reduce always reads the gradient window it reduces into (out += sum) instead of overwriting the windows zero_grad has just cleared",
    "no_join_mask": "the join backward reads the lower block's output y for its ReLU mask instead of the bit field tuber_block_out_fwd_mask writes (bit-identical gradients)",
    "no_strided_join_fusion": "gemm + rows_scatter_add + block_out_bwd at the stage boundaries instead of the join GEMM with the strided residual",
    "no_ds_join_fusion": "stand-alone block_out_bwd for the first block of every stage (the join kernels take identity-block joins only)",
    "no_bn_bwd_fa": "BatchNorm backward as finalize + apply launches everywhere (no one-launch form)",
    "no_bn_bwd_fa_after_reduce": "no one-launch BatchNorm backward behind the first-stage row reduction (layer1 / layer2)",
    "no_bn1_in_dw_fwd": "stand-alone tuber_bn_finalize for bn1 instead of finalising it inside the stride-1 depthwise forward kernel",
    "no_bn3_in_dw": "stand-alone bn_bwd_fa for bn3 instead of forming it inside the depthwise backward kernels",
    "no_dw_bwd_one_launch": "depthwise data and weight gradient of a stride-1 block as two launches",
    "no_conv4_bwd_fused": "layer1's conv4 backward on the separate BatchNorm / GEMM kernels",
    "no_conv1_bwd_fused": "layer1's conv1 backward on the separate BatchNorm / GEMM kernels",
    "no_blockout_conv1": "layer1's residual join and the next conv1 as two launches",
    "no_entry_conv": "conv1 and the projection conv of layer1's first block as two GEMMs",
    "no_proj_bwd_fused": "layer1's projection-shortcut backward on the separate kernels",
    "no_stem_bn_in_wgrad": "stand-alone bn_bwd_apply for the stem instead of forming it inside the stem weight-gradient kernel",
    "dw_register_tiled": "the register-tiled depthwise kernels (used for the strided blocks) for every block",
    "no_ln_bwd_fusion": "LayerNorm backward and the data-gradient GEMM of the linear in front of it as two launches (no tuber_ln_bwd_dx)",
    "no_in_proj_dx2": "the few-row data gradients of the decoder (in-projection: x and query_pos; linear1) as tuber_gemm_nt launches instead of tuber_rows_dx2",
    "no_decoder_coop": "the DETR decoder stack as its ~80 separate launches instead of the one cooperative launch (csrc/decoder_coop.hip)",
    "eager_step": "train_tuber_detection without the captured hipGraph step",
    "eval_bf16_stream": "eval forward with the residual streams (block outputs, LayerNorm outputs) stored in bf16 like the training path, instead of the fp32 "
                        "streams of the eval precision mode (round 6; also TUBER_EVAL_PRECISION=bf16_stream)",
    "no_eval_conv4_join": "eval precision mode with conv4 and the residual join of an identity block as two launches (tuber_gemm_nt + tuber_block_out_fwd_f32) "
                          "instead of the join in the GEMM epilogue (tuber_gemm_nt_bn_out; bit-identical outputs)",
    "eval_bf16_decoder": "eval precision mode without the fp32 decoder / heads (csrc/eval_f32.hip): the decoder on the bf16 launch chain with fp32 LayerNorm streams only",
}


def _parse(text):
    names = {t.strip().lower() for t in (text or "").replace(";", ",").split(",") if t.strip()}
    bad = sorted(names - set(KNOWN))
    if bad:
        raise ValueError("TUBER_AB: unknown switch(es) %s; known: %s" % (bad, sorted(KNOWN)))
    return names


_active = _parse(os.environ.get("TUBER_AB"))


def on(name):
    """is the A/B switch ``name`` set?  (read at call time: tests flip switches inside one process)"""
    if name not in KNOWN:
        raise KeyError(name)
    return name in _active


def active():
    return sorted(_active)


def eval_fp32_stream():
    """eval precision mode (default): under ``model.eval()`` the residual streams stay fp32 between the blocks / layers"""
    return not on("eval_bf16_stream") and os.environ.get("TUBER_EVAL_PRECISION", "fp32_stream") != "bf16_stream"


@contextlib.contextmanager
def override(*names):
    """run a block with exactly ``names`` set (tests)"""
    global _active
    old = _active
    _active = _parse(",".join(names))
    try:
        yield
    finally:
        _active = old
