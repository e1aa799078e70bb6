"""Regenerate include/tuber_hip.h from the extern "C" definitions in csrc/*.hip.

The prototypes are extracted from the sources (so header and library cannot drift); the
per-function documentation -- what reference op each entry point replaces (file:line under
/root/reference) -- lives in DOC below.  Run:  python tubelet_transformer_amd/csrc/gen_header.py
"""
import glob
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))

DOC = {
    "tuber_gemm_nt_addproj": "packed attention in-projection with the positional embedding folded in: C = f(A).B^T + bias with f(A) = A + A2 for the output "
                             "columns [0, add_ncols) (q / k rows of in_proj_weight: with_pos_embed, models/transformer/transformer.py:150-159,215-240) and "
                             "f(A) = A for the v rows -- one GEMM instead of an add kernel and two GEMM launches. add_ncols % 128 == 0.",
    "tuber_bn_bwd_fa": "tuber_bn_bwd_finalize + tuber_bn_bwd_apply in ONE launch for short partial lists (layer3 / layer4 of the CSN body): every workgroup "
                       "derives the coefficients of its 128-channel strip from the R partial rows (fp64) and applies dx = cA*dz + cB*x + cC to its rows; "
                       "dgamma / dbeta accumulated by the first row chunk (NULL: frozen BatchNorm). autograd of nn.BatchNorm3d (ir_CSN_152.py:46,56,64,154).",
    "tuber_bn_bwd_fa_max_rows": "largest R tuber_bn_bwd_fa accepts.",
    "tuber_ln_bwd_dx": "tuber_layernorm_bwd (partial rows only) + the data-gradient tuber_gemm_nt of the linear layer in front of the LayerNorm in ONE launch: "
                       "autograd of tgt = norm(tgt + dropout(sublayer)) where the sublayer ends in out_proj / linear2 (models/transformer/transformer.py:160-167,229-247). "
                       "Every workgroup runs the LayerNorm backward of its 32 rows itself and multiplies the bf16 result from LDS with W on MFMA; the column-0 workgroups "
                       "store dx / dxd / the dgamma, dbeta partial rows ([tuber_ln_bwd_dx_blocks(M)][2E]) as the two-launch path does. dy2: a second gradient contribution (replaces "
                       "a tuber_axpby launch) or NULL; res: an existing gradient of the linear's input to add; cm / alpha: the ReLU / Dropout mask of that input.",
    "tuber_rows_dx2": "both data gradients of a packed attention in-projection with the positional embedding folded in (tuber_gemm_nt_addproj; "
                      "models/transformer/transformer.py:150-159,215-240) in ONE launch for few rows: dx = g.W over all N columns (+ res), dpos = g[:, :Nq].W[:Nq] "
                      "stored as a prefix of the same reduction. Replaces two tuber_gemm_nt launches per decoder in-projection.",
    "tuber_criterion_scale": "backward of tuber_criterion_loss's loss table in one launch: the stored per-layer gradients scaled by the incoming gradient of the [L][4] "
                             "losses (autograd of the weighted sum of loss terms, models/criterion.py:169-206 + train_tuber_ava.py loss weighting).",
    "tuber_weighted_sum": "sum_i a[i] w[i] on the device in a fixed order, or its backward gout * w: the weighted total of the loss terms "
                          "(pipelines/video_action_recognition.py:147) without ATen launches.",
    "tuber_ln_bwd_dx_blocks": "partial rows tuber_ln_bwd_dx writes for M rows (32 rows per block).",
    "tuber_ln_bwd_dx_pays": "1 where tape.py uses tuber_ln_bwd_dx instead of the two launches: M <= 64 rows at any width, or Kin <= 256 (measured; the encoder's linear2 loses).",
    "tuber_ln_bwd_dx_supported": "1 when tuber_ln_bwd_dx handles a LayerNorm of width E in front of a linear with Kin inputs (E = 256, Kin % 64 = 0).",
    "tuber_bn_bwd_fa_rows": "rows per workgroup tuber_bn_bwd_fa takes for (M, C): 64, 128 or 176 -- the grid that sits at or just under a multiple of the 256 CUs.",
    "tuber_bn_bwd_fa_rows_set": "test hook: rows per THREAD of tuber_bn_bwd_fa forced to 4, 8 or 11 (0 = the heuristic), so that tests cover every instantiation at every shape.",
    "tuber_class_error": "class_error of the matched queries of one decoder layer, on the device: 100 - exact-set accuracy (AVA, utils/misc.py:497-518 "
                         "via models/criterion.py:76-78) or top-1 accuracy (JHMDB, utils/misc.py:521-539 via criterion.py:258-260).",
    "tuber_targets_pack": "the padded [B, Tmax] target layout of one batch in ONE launch: boxes (column 0 = key-frame index dropped, models/detr/matcher.py:64, "
                          "models/criterion.py:106), labels (multi-hot rows for AVA, class ids for JHMDB) and the per-clip counts from B per-clip device tensors "
                          "whose pointers travel by value (HOST arrays boxes[B], labels[B], sizes[B]); zero padding included. Replaces the two memsets + two sliced "
                          "copies per clip the captured step issued before every replay.",
    "tuber_targets_pack_max": "largest B tuber_targets_pack accepts.",
    "tuber_decoder_coop_fwd": "the DETR decoder stack forward (TransformerDecoder.forward / TransformerDecoderLayer.forward_post, models/transformer/transformer.py:99-128,"
                              "218-249) as ONE cooperative launch: 16 workgroups on one XCD split every weight matrix by output columns, exchange the <= 32 x 256 activations "
                              "through that XCD's L2 and meet at 8 XCD-local barriers per layer. layer_ptrs: HOST array, tuber_decoder_coop_ptrs_per_layer() device pointers per "
                              "layer (6 bf16 weights: self in-proj, self out-proj, cross q rows, cross out-proj, linear1, linear2; their 6 fp32 biases; norm1 / norm2 / norm3 weight, "
                              "bias; the layer's packed memory projection [(memory + pos) W_k | memory W_v]; then the 21 tensors the launch chain saves for its backward: qkv, o1, lse1, "
                              "a1, y1, xhat1, rstd1, q, o2, lse2, a2, y2, xhat2, rstd2, h, f2, y3, xhat3, rstd3, xhatN, rstdN). layer_salts: HOST array, 6 dropout salts per layer. "
                              "sync: 4 zeroed 32-bit device words, left zero by a clean run; sync[2] != 0 afterwards = a barrier timed out (a workgroup was not co-resident): the kernel has "
                              "overwritten hs with NaN, so the step's loss / gradient norm are NaN and tuber_adamw_segment skips the update.",
    "tuber_block_out_fwd_f32": "tuber_block_out_fwd for the eval precision mode: y = relu(bn4(c4) + shortcut) (models/backbones/ir_CSN_152.py:84-90) with the residual "
                               "stream kept in fp32 between the bottlenecks -- reads the previous block's fp32 output, writes the bf16 GEMM operand AND the fp32 stream.",
    "tuber_layernorm_fwd_f32": "tuber_layernorm_fwd for the eval precision mode: LayerNorm(x + res) (models/transformer/transformer.py:160-167,229-247, post-norm) with the "
                               "residual stream in fp32 from LayerNorm to LayerNorm; writes the bf16 GEMM operand and (optionally) the fp32 stream.",
    "tuber_bn_eval_affine_multi": "tuber_bn_eval_affine (eval-mode nn.BatchNorm3d: scale = gamma / sqrt(running_var + eps), shift = beta - running_mean * scale; "
                                  "models/backbones/ir_CSN_152.py:46,56,64,154 under model.eval()) for every BatchNorm of the body in ONE launch over a device table of "
                                  "{gamma, beta, running_mean, running_var, scale, shift, C, unused} rows.",
    "tuber_gemm_nt_bn_out": "EVAL forward of a bottleneck's tail in ONE launch: conv4 on relu(bn3(c3)) + the eval-mode bn4 + the residual join + ReLU in the GEMM epilogue "
                            "(models/backbones/ir_CSN_152.py:62-64,84-90 under model.eval(): a BatchNorm is a constant affine map there, so the join does not wait for "
                            "statistics). Writes y as bf16 (next block's GEMM operand) and as fp32 (the residual stream of the eval precision mode); c4 never reaches HBM. "
                            "Replaces tuber_gemm_nt(amode 1) + tuber_block_out_fwd_f32 for identity blocks.",
    "tuber_block_out_fwd_mask": "tuber_block_out_fwd (y = relu(bn4(c4) + shortcut), models/backbones/ir_CSN_152.py:84-90) that also writes the ReLU mask of y as a bit field "
                                "([M][C / 8] bytes, bit e of byte (m, c / 8) = y[m][c + e] > 0) for the join backward (tuber_gemm_nt_join_mask).",
    "tuber_gemm_nt_join_mask": "tuber_gemm_nt_join (conv1 data gradient of bottleneck i+1 + the join backward of bottleneck i: autograd of ir_CSN_152.py:72,86-89) with the ReLU mask "
                               "[y > 0] read from the bit field of tuber_block_out_fwd_mask instead of y itself: identical results, M*N/8 bytes instead of 2*M*N.",
    "tuber_gemm_nt_join_ds_mask": "tuber_gemm_nt_join_ds with the ReLU mask read from the bit field of tuber_block_out_fwd_mask instead of y (identical results).",
    "tuber_gemm_nt_join_strided_mask": "tuber_gemm_nt_join_strided with the ReLU mask read from the bit field of tuber_block_out_fwd_mask / tuber_blockout_conv1_fwd_mask instead of y (identical results).",
    "tuber_blockout_conv1_fwd_mask": "tuber_blockout_conv1_fwd that also writes the ReLU mask of y as a bit field ([M][32] bytes) for the join backward across the stage boundary.",
    "tuber_linear_f32": "fp32 linear layer of the eval precision mode: y = act((x [+ add]) . W^T + bias) on the fp32 master weights -- the decoder's nn.Linear / packed "
                        "in-projections (models/transformer/transformer.py:218-249, with_pos_embed as the add operand) and the box / actor heads (models/tuber_ava.py:121-125,142; "
                        "MLP models/criterion.py:485-497) under model.eval().",
    "tuber_linear_f32_batched": "tuber_linear_f32 for nbatch weight sets over ONE input in one launch (set z: W + z * w_stride, bias + z * bias_stride, y + z * y_stride, strides in elements): "
                                "the memory-side K / V projections of all decoder layers (models/transformer/transformer.py:232-237), which do not depend on the decoder state.",
    "tuber_attention_f32": "fp32 multi-head attention core (head dimension 32) of the eval precision mode: the decoder's self- and cross-attention "
                           "(nn.MultiheadAttention, transformer.py:218-240) with fp32 scores, softmax and values.",
    "tuber_flag_signal": "software ordering edge between two HIP streams, producer side: *flag += 1 (release, agent scope) once everything enqueued on the stream "
                         "before it has completed; capturable as the last node of a graph part. With tuber_flag_wait it replaces the hipEventRecord / "
                         "hipStreamWaitEvent pair between the backward stream and the gradient exchange's stream (DistributedDataParallel's reducer, "
                         "utils/model_utils.py:47-52): a pending cross-stream event wait slows every kernel of a launch-bound graph by ~1.3 us on this runtime.",
    "tuber_flag_wait": "consumer side: a one-wave kernel that polls *flag until it reaches `expect` (wrap-around safe), in front of the collective on the "
                       "exchange's stream; gives up after ~2 s, sets *err = 1 and lets the stream go on (the caller checks the word where it synchronises).",
    "tuber_decoder_coop_supported": "1 when tuber_decoder_coop_fwd takes this decoder (d_model 256, 8 heads, FFN 2048, batch * 8 == 16 attention units, batch * queries <= 32, <= 6 layers).",
    "tuber_decoder_coop_ptrs_per_layer": "device pointers per layer in tuber_decoder_coop_fwd's layer_ptrs (40).",
    "tuber_mask_resize": "F.interpolate(mask[None].float(), size=(h, w)).to(torch.bool)[0] (models/backbone_builder.py:85-86): nearest-neighbour resize of the "
                         "clip padding mask to the feature grid = the transformer's key-padding mask, ATen's source-index rule.",
    "tuber_gemm_nt_join": "conv1 data gradient of one bottleneck fused with the join backward of the bottleneck below it: dz = (A.B^T + R) * [Y > 0] "
                          "plus the BatchNorm-backward partial rows (sum dz, sum dz*Cm) per 64 output rows (tuber_gemm_nt_stat_rows) -- "
                          "tuber_gemm_nt(epi 0, +R) followed by tuber_block_out_bwd without dx reaching HBM (autograd of "
                          "models/backbones/ir_CSN_152.py:72,86-89 across a block boundary). Y = the lower block's output, Cm = its raw conv4 output; R may be NULL.",
    "tuber_gemm_nt_join_strided": "tuber_gemm_nt_join at a STAGE boundary: R is the data gradient of the upper stage's strided projection shortcut (one row per sampled "
                                  "position, n*To*Ho*Wo rows) and is added to the output rows (n, t, h, w) with t % st == h % ss == w % ss == 0 inside the epilogue -- replaces "
                                  "tuber_gemm_nt + tuber_rows_scatter_add + tuber_block_out_bwd (autograd of models/backbones/ir_CSN_152.py:72,86-89,155-161).",
    "tuber_gemm_nt_join_ds": "tuber_gemm_nt_join below a stage's FIRST block: Cd = the raw output of that block's projection shortcut, stat2 = the rows sum dz*cd of the "
                             "shortcut BatchNorm's backward (tuber_block_out_bwd's third statistics buffer). autograd of models/backbones/ir_CSN_152.py:86-89,155-161.",
    "tuber_gemm_tn_group": "n <= tuber_gemm_tn_group_max() weight-gradient GEMMs (each exactly one tuber_gemm_tn: dW = G^T f(A) of a 1x1x1 conv, "
                           "autograd of models/backbones/ir_CSN_152.py:41,58,155-161) in ONE launch; args_host = HOST array of struct TuberGemmTNArgs "
                           "{const void* G; long ldg; const void* A; long lda; float* partial; float* out; int accumulate, M, N, K, amode, gather, "
                           "To, Ho, Wo, Ti, Hi, Wi, st, ss; const float* a_scale; const float* a_shift; float* bias_grad; const void* A2; long lda2;} (A2: optional addend, A := A + A2) with the meaning of the tuber_gemm_tn arguments. "
                           "Transpose-read kernel shapes only (N, K, ld multiples of 8, 64x64 tiles); several slabs need accumulate = 2 (the caller reduces them).",
    "tuber_gemm_nt_wsk_tile_rows": "rows per tile of the wave-split-K form tuber_gemm_nt takes for a plain-A (M, N, K): 0 (not taken), 64, or 96 (shapes whose "
                                   "64-row tiling has more workgroups than the chip has CUs while the 96-row one does not: the layer3 / layer4 long-K convs).",
    "tuber_gemm_nt_96_set": "EXPERIMENT hook (round 6): 0 switches the 96-row tiles of the regular tuber_gemm_nt pipeline off (64-row tiles for layer2's conv1 forward / conv4 data gradient).",
    "tuber_gemm_nt_wsk96_set": "EXPERIMENT hook: 0 switches the 96-row wave-split-K tiles of tuber_gemm_nt off (64-row tiles everywhere).",
    "tuber_gemm_tn_args_bytes": "sizeof(struct TuberGemmTNArgs) as compiled (host-side layout check).",
    "tuber_gemm_tn_group_max": "largest n tuber_gemm_tn_group accepts (the argument blocks travel by value in the kernel argument segment).",
    "tuber_comm_version": "RCCL bound at run time (dlopen librccl.so.1: the instance the host process already loaded, else ROCm's): its version code, "
                          "or < 0 when it cannot be loaded. Replaces the NCCL process group the reference's DistributedDataParallel wrapper uses "
                          "(utils/model_utils.py:43-52, pipelines/launch.py:44-49).",
    "tuber_comm_etimedout": "the (negative) return code of tuber_comm_init_timeout when a rank never reached the bootstrap (fatal for the job).",
    "tuber_comm_last_error": "message of the last failing tuber_comm_* call.",
    "tuber_comm_unique_id": "ncclGetUniqueId: rank 0 fills the 128-byte rendez-vous id (HOST memory) that every rank hands to tuber_comm_init "
                            "(shipped over any side channel: torch.distributed store, MPI, a file).",
    "tuber_comm_init": "ncclCommInitRank on HIP device `device` (collective over all `nranks` processes, one per GPU); *comm_out receives the communicator.",
    "tuber_comm_allreduce_sum": "in-place sum over all ranks of buf[0..count) (dtype 0 = fp32, 1 = bf16) enqueued on `stream` -- the gradient all-reduce DDP's "
                                "reducer issues per bucket (torch/nn/parallel/distributed.py via utils/model_utils.py:46-52), here on windows of the flat "
                                "gradient buffer with no bucket copies; stream-ordered, capturable into a hipGraph.",
    "tuber_comm_allreduce_sum_multi": "the same for n windows (HOST arrays ptrs[n], counts[n]) as ONE RCCL group.",
    "tuber_comm_destroy": "ncclCommDestroy.",
    "tuber_conv1_bwd_fused": "backward of the bottleneck's first pointwise conv through bn1, for the wide-activation stage (256-channel block input, P = 64: layer1), as ONE persistent kernel: "
                             "dc1 = cA*dz1 + cB*c1 + cC lives only in LDS; dx = dc1 . W1 + R -- with Cm given, the residual join of the identity block below: "
                             "dz = dx * [X > 0] plus the statistics rows of tuber_gemm_nt_join --; dW1 = dc1^T . X from the SAME X tile (one fp32 slab per workgroup). "
                             "With Cd (the raw output of the lower block's projection shortcut) the join is that of a stage's FIRST block: st2 receives the rows sum dz*cd of the shortcut BatchNorm's backward "
                             "(tuber_block_out_bwd's third statistics buffer). Replaces tuber_bn_bwd_apply + tuber_gemm_nt / tuber_gemm_nt_join (+ tuber_block_out_bwd) + tuber_gemm_tn. autograd of models/backbones/ir_CSN_152.py:72-74,84-90.",
    "tuber_conv1_bwd_slabs": "workgroups = fp32 weight-gradient slabs tuber_conv1_bwd_fused produces for M rows.",
    "tuber_conv1_bwd_supported": "1 for the (block-input channels, P) the fused conv1 backward is built for.",
    "tuber_blockout_conv1_fwd": "residual join of one bottleneck + the first pointwise conv of the NEXT one as ONE persistent kernel (256-channel block output: layer1, and layer1 -> layer2): "
                                "y = relu(bn4(c4) + shortcut) exactly as tuber_block_out_fwd writes it, kept in LDS per 64-row tile and multiplied with the next conv1 weight from there "
                                "(c1 + the partial statistics rows of tuber_gemm_nt epi 1) -- y is written once and not read back. models/backbones/ir_CSN_152.py:84-90 then :72-74.",
    "tuber_blockout_conv1_supported": "1 for the (block-output channels, next conv1 output channels) the fused forward kernel is built for.",
    "tuber_stem_conv_bwd_weight_bn": "tuber_stem_conv_bwd_weight with the stem BatchNorm's backward apply folded into its gradient operand: takes the masked gradient of "
                                     "bn's output (tuber_stem_pool_bwd), the raw conv output and cA / cB / cC of tuber_bn_bwd_finalize; G = bf16(cA*dz0 + cB*c0 + cC) is formed "
                                     "while the tile is parked in LDS. Replaces tuber_bn_bwd_apply + tuber_stem_conv_bwd_weight (autograd of ir_CSN_152.py:131-133).",
    "tuber_entry_conv_fwd": "layer1's first bottleneck: conv1 (64 -> 64) and the projection-shortcut conv (64 -> 256) from ONE pass over the block input "
                            "x [M, 64] (persistent 64-row tiles, both weight matrices resident in LDS), with the per-64-row statistics rows of both outputs "
                            "(what tuber_gemm_nt epi 1 writes; NULL in eval mode). models/backbones/ir_CSN_152.py:72-74 and :86-87.",
    "tuber_entry_conv_supported": "1 for the (block input channels, conv1 output channels, projection output channels) the fused entry kernel is built for.",
    "tuber_conv4_bwd_fused": "backward of the bottleneck's second pointwise conv through bn4, for the wide-activation stage (C4 = 256, P = 64: layer1), as ONE persistent kernel: "
                             "dc4 = cA*dz + cB*c4 + cC (bn4 backward apply; coefficients from tuber_bn_bwd_finalize) is formed per 64-row tile in LDS and feeds BOTH the data gradient "
                             "dz3 = (dc4 . W4) * [bn3(c3) > 0] (+ the per-tile statistics rows tuber_gemm_nt epi 2 writes) and the weight gradient dW4 = dc4^T . relu(bn3(c3)) "
                             "(one fp32 slab per workgroup: tuber_conv4_bwd_slabs(M)) -- dc4 never reaches HBM: 2 passes over the [M, C4] tensors instead of 5 "
                             "(tuber_bn_bwd_fa + tuber_gemm_nt + tuber_gemm_tn). autograd of models/backbones/ir_CSN_152.py:58-64,78-84. "
                             "sc3 = sh3 = NULL selects the projection-shortcut form (down_sample conv + BatchNorm of a stage's first block, ir_CSN_152.py:86-87): "
                             "c4 = the projection's output, c3 = the block input, w4t = the projection weight transposed; no mask, no activation, no statistics rows.",
    "tuber_conv4_bwd_slabs": "workgroups = fp32 weight-gradient slabs tuber_conv4_bwd_fused produces for M rows.",
    "tuber_conv4_bwd_supported": "1 for the (C4, P) the fused conv4 backward is built for.",
    "tuber_dwconv_tile_bwd_data_bn": "tuber_dwconv_tile_bwd_data with the BatchNorm backward of bn3 (autograd of nn.BatchNorm3d, ir_CSN_152.py:56,76-77) folded into its "
                                     "gradient operand: takes bn3's masked output gradient dzu, bn3's input xu (= c3) and the R <= 128 partial rows (sum dz, sum dz*x) "
                                     "instead of a finished dc3; every workgroup derives cA / cB / cC of its 64 channels (fp64) and forms dc3 = cA*dzu + cB*xu + cC "
                                     "in fp32 while staging. dgamma / dbeta of bn3 are accumulated (+=) unless NULL. Replaces tuber_bn_bwd_fa + tuber_dwconv_tile_bwd_data.",
    "tuber_dwconv_tile_bwd_both_bn": "data gradient AND weight gradient of one stride-1 depthwise conv (bn3's backward folded in) in ONE launch and ONE pass over the "
                                     "operands: both are sums over the same (p, p - off) position pairs, so the ring of dc3 staged for the data gradient also feeds "
                                     "dW[off] = sum_p relu(bn1(x))[p] * dc3[p - off] -- 4 tensor passes (read dzu, xu, x; write dz). dz / dgamma / dbeta bit-identical to "
                                     "tuber_dwconv_tile_bwd_data_bn, statistics rows and weight gradient equal up to fp32 summation order; partial = "
                                     "tuber_dwconv_tile_blocks(...) blocks of [27][C], reduced by the caller. autograd of models/backbones/ir_CSN_152.py:48-56,74-77.",
    "tuber_dwconv_tile_bwd_weight_bn": "tuber_dwconv_tile_bwd_weight with the same fold: the output-position gradient is formed from (dzu, xu, partial rows) on load.",
    "tuber_comm_init_timeout": "tuber_comm_init with a deadline: the bootstrap (a collective) runs on a helper thread and the call returns -3 with a message naming "
                               "the waiting rank when its peers have not arrived after timeout_ms -- a dead rank fails the job loudly instead of hanging it "
                               "(what torch.distributed's process-group timeout does for the reference, pipelines/launch.py:44-49). timeout_ms <= 0: no deadline.",
    "tuber_comm_count": "ncclCommCount / ncclCommUserRank of the communicator: the world size and rank RCCL itself sees (rank_out may be NULL).",
    "tuber_cast_bf16_f32_scale": "dst = float(src) * scale: expands a bf16-compressed, all-reduced gradient window back into the fp32 gradient buffer and averages it in the same pass.",
    "tuber_frames_resize": "PIL.Image.resize((nw, nh)) of every decoded frame (datasets/ava_frame.py:146-150; jhmdb_frame.py alike): Pillow's 8-bit two-pass "
                           "fixed-point bicubic (libImaging/Resample.c), packed RGB uint8 [nimg][H][W][3] -> [nimg][Ho][Wo][3], bit-exact.",
    "tuber_clip_prepare": "hflip + crop + ColorJitter + ToTensor/Normalize (datasets/video_transforms.py:69-85,20-66,333-369,308-322; pipeline "
                          "datasets/ava_frame.py:158-176) and the batch collate of utils/misc.py:279-282,367-425 in one pass: uint8 frames -> "
                          "fp32 [N][3][T][Hmax][Wmax] zero padded + bool mask [N][Hmax][Wmax]. One TuberClipDesc per clip (56 bytes: long long src_off; "
                          "int H, W, y1, x1, h, w, flip, jitter, hue, sat, val, pad).",
    "tuber_clip_desc_bytes": "sizeof(TuberClipDesc) as the library was compiled (host-side layout check).",
    "tuber_multi_reduce": "every deferred second-stage reduction of one backward pass in one launch: the weight-gradient launchers called with "
                          "accumulate = 2 (tuber_gemm_tn, tuber_dwconv_*_bwd_weight, tuber_colsum, tuber_layernorm_bwd) leave their partials in place; "
                          "table = MultiReduceEntry[] {const float* P; float* out; long n, stride; int S, mode, C, next}: out[j] += sum_s P[s*stride+j] "
                          "(mode 0 / 2: per element / per float4, s ascending; mode 1: 32-way tree; C > 0: depthwise [S][27][C] -> [C][27]; next: chained contribution to the same out); blk = int2[] (head entry, block in entry), "
                          "1024 / 4096 / 32 elements per block in mode 0 / 2 / 1. Same summation order as the immediate kernels.",
    "tuber_multi_reduce_entry_bytes": "sizeof(MultiReduceEntry) as compiled (host-side layout check).",
    "tuber_temporal_max_fwd": "nn.MaxPool3d((T,1,1)) over the T frames of the backbone features, TEMPORAL_DS_STRATEGY 'max' (models/backbone_builder.py:45-47,73): "
                              "rows (b,t,p) of E bf16 -> rows (b,p) + the uint8 index of the first maximal frame (arg may be NULL in eval).",
    "tuber_temporal_max_bwd": "backward of tuber_temporal_max_fwd: the gradient goes to the frame that held the maximum, zeros elsewhere.",
    "tuber_gemm_nt": "C[M,N] = f(A)[M,K] . B[N,K]^T on MFMA bf16. Replaces every nn.Conv3d(k=1) of the CSN bottlenecks "
                     "(models/backbones/ir_CSN_152.py:41,58,155-161), input_proj/class_proj (models/tuber_ava.py:57-58) and every "
                     "nn.Linear / packed in-projection (models/transformer/transformer.py:159-165,227-245; transformer_layers.py:81-94; "
                     "tuber_ava.py:65-71), plus their data gradients (B = W^T). amode 1 fuses relu(x*a_scale[k]+a_shift[k]) "
                     "(BatchNorm apply, ir_CSN_152.py:72-79) into the A load; gather=1 reads A rows through the strided "
                     "(n,t*st,h*ss,w*ss) map of the down_sample conv (:155-161). epi 0: +bias, +R, ReLU, bf16|fp32 out; "
                     "epi 1: bf16 out + per-column partial (sum, sum^2) rows for training-mode BatchNorm; epi 2: out = acc*[Cm*m_scale+m_shift>0] "
                     "+ partial (sum dz, sum dz*Cm) rows for BatchNorm backward (stat rows optional: with m_scale NULL it is the ReLU / ReLU+Dropout "
                     "backward mask from the saved activation). Accumulators are scaled by alpha first; epi 0 can end with Dropout(drop_p) "
                     "keyed by (seed, salt, m*N+n) -- FFN linear1 (transformer.py:160-162). K % 64 == 0.",
    "tuber_gemm_nt_cfg": "tile configuration tuber_gemm_nt uses for (M,N,K): 13 = 64x64 (two-tile prefetch), 7 = 64x128, 0 = 128x128 (plain epilogue only); 2 / 12 / 17 = rejected A/B variants, built only with -DTUBER_AB_VARIANTS.",
    "tuber_gemm_tn_tile": "output tile edge of the transpose-read weight-gradient kernel for (M, N, K): 128 (mid-M backbone shapes, N and K multiples of 128) or 64.",
    "tuber_gemm_nt_has_cfg": "1 when tuber_gemm_nt_set_cfg(cfg) names a tile configuration this library was built with.",
    "tuber_gemm_nt_stat_rows": "rows of partial statistics tuber_gemm_nt(epi 1|2) writes for (M,N).",
    "tuber_gemm_tn": "dW[N,K] (+)= sum_m G[m,N]^T . f(A)[m,K]: weight gradient of the same convs / linears (autograd of the ops above); "
                     "split over M into fp32 slabs `partial` [tuber_gemm_tn_slabs][N][K], then reduced deterministically.",
    "tuber_gemm_tn_fuses_bias": "can tuber_gemm_tn also produce the bias gradient for this shape: 0 no; 1 yes, accumulated into bias_grad[N] directly (single slab); 2 yes, bias_grad receives one partial row per slab ([tuber_gemm_tn_slabs][N], reduced by the caller).",
    "tuber_gemm_tn_slabs": "number of slabs (size of `partial` / (N*K)) tuber_gemm_tn uses.",
    "tuber_dwconv_fwd": "depthwise Conv3d(C,C,3,groups=C,stride=(st,ss,ss),padding=1) on NDHWC bf16, ResNeXtBottleneck.conv3 "
                        "(ir_CSN_152.py:48-51) with relu(bn1(.)) fused on load (sc/sh may be NULL) and bn3 partial statistics on store.",
    "tuber_dwconv_bwd_data": "input gradient of conv3 fused with the backward of relu(bn1(.)): dz = da*[x*sc+sh>0] + partial (sum dz, sum dz*x).",
    "tuber_dwconv_bwd_weight": "weight gradient of conv3 ([C][27] fp32), activation relu(bn1(x)) recomputed on load.",
    "tuber_dwconv_fwd_stat_rows": "partial-stat rows written by tuber_dwconv_fwd.",
    "tuber_dwconv_bwd_data_stat_rows": "partial-stat rows written by tuber_dwconv_bwd_data.",
    "tuber_dwconv_bwd_weight_blocks": "blocks (size of `partial` / (27*C)) used by tuber_dwconv_bwd_weight.",
    "tuber_dwconv_tile_fwd_bn": "tuber_bn_finalize + tuber_dwconv_tile_fwd in ONE launch: the training-mode BatchNorm in front of conv3 (bn1, ir_CSN_152.py:46-51) is "
                                "finalised inside the conv -- every workgroup derives scale / shift of its 64 channels from the producing conv's R partial rows "
                                "(pst0 = sum x, pst1 = sum x^2 over `count` rows; fp64 sums, the arithmetic of tuber_bn_finalize expression for expression), the first "
                                "workgroup of each channel group writes scale / shift / mean / invstd [C] and updates rmean / rvar / nbt (NULL: none). Other arguments as tuber_dwconv_tile_fwd.",
    "tuber_dwconv_tile_fwd": "stride-1 conv3 (ir_CSN_152.py:48-51) with the input planes staged once per workgroup in an LDS ring (activated on the way in): "
                             "same contract as tuber_dwconv_fwd for st = ss = 1; partial-stat rows = tuber_dwconv_tile_blocks.",
    "tuber_dwconv_tile_bwd_data": "LDS-staged data gradient of the stride-1 conv3, fused with the backward of relu(bn1(.)) like tuber_dwconv_bwd_data.",
    "tuber_dwconv_tile_bwd_weight": "LDS-staged weight gradient of the stride-1 conv3 (activation recomputed while staging).",
    "tuber_dwconv_tile_blocks": "workgroups along x of the LDS-staged depthwise kernels = partial-stat rows / weight-gradient partial blocks.",
    "tuber_dwconv_tile_wgrad_blocks": "partial blocks (size of `partial` / (27*C)) of tuber_dwconv_tile_bwd_weight.",
    "tuber_dw_wgrad_reduce": "dw[c][tap] (+)= sum_r partial[r][tap][c]: second stage of the depthwise weight gradient.",
    "tuber_bn_finalize": "training-mode nn.BatchNorm3d(eps=1e-3, momentum=0.1) statistics (ir_CSN_152.py:15-16,46,56,64,119,154): partial rows -> "
                         "mean/invstd, scale=gamma*invstd, shift=beta-mean*scale, running_mean/var (unbiased) and num_batches_tracked update.",
    "tuber_stat_rows_reduce": "first stage for long partial-statistics lists (R > 512 rows: layer1): [R][C] x2 -> [tuber_stat_rows_reduced(R)][C] x2.",
    "tuber_stat_rows_reduced": "rows left by tuber_stat_rows_reduce (R itself when no first stage is needed).",
    "tuber_bn_eval_affine": "eval-mode BatchNorm3d folded to scale/shift from the running statistics.",
    "tuber_bn_bwd_finalize": "BatchNorm backward coefficients: dx = cA*dz + cB*x + cC, dgamma = sum dz*xhat, dbeta = sum dz.",
    "tuber_bn_bwd_apply": "dx = cA*dz + cB*x + cC (BatchNorm backward apply), bf16 [M,C].",
    "tuber_block_out_fwd": "bottleneck join y = relu(bn4(c4) + shortcut) (ir_CSN_152.py:81-90); shortcut = res or bn_ds(res) when rs/rh given.",
    "tuber_block_out_bwd": "backward of the join: dz = dy*[y>0] and the partial statistics of bn4 (and of the down_sample BN).",
    "tuber_relu_bn_bwd_reduce": "dz = g*[x*sc+sh>0] + partial (sum dz, sum dz*x): backward of relu(bn(x)) when not fused elsewhere.",
    "tuber_rowblock_count": "partial-stat rows written by the row-blocked reduce kernels for M rows.",
    "tuber_layernorm_fwd": "y = LayerNorm(Dropout_p(x) (+ res)) over E in {256, 2048}, eps 1e-5 (nn.LayerNorm, transformer.py:163-167,229-247,116-123; "
                           "transformer_layers.py:84,91,96,437-445); saves xhat (bf16) and rstd for backward.",
    "tuber_layernorm_bwd": "backward of tuber_layernorm_fwd: dx (gradient of res), dxd (gradient of x through the regenerated dropout mask) and dgamma/dbeta via block partials.",
    "tuber_layernorm_bwd_blocks": "blocks used by tuber_layernorm_bwd (partial = 2*blocks*E floats).",
    "tuber_reduce_rows": "out[c] (+)= sum_r P[r][c].",
    "tuber_colsum_blocks": "row blocks (size of `partial` / C) used by tuber_colsum.",
    "tuber_colsum": "bias gradient: out[c] (+)= sum_m g[m][c] for bf16 g.",
    "tuber_gemm_nt_set_cfg": "tuning hook: force tile configuration cfg for every later tuber_gemm_nt (-1 = automatic choice).",
    "tuber_stem_conv_fwd": "stem Conv3d(3,64,(3,7,7),s=(1,2,2),p=(1,3,3)) (ir_CSN_152.py:109-115) as an implicit MFMA GEMM from the fp32 NCDHW clip to NDHWC bf16, "
                           "with the partial statistics of the following BatchNorm; Wp = tuber_stem_pack_weight(conv1.weight).",
    "tuber_stem_conv_bwd_weight": "weight gradient of the stem conv ([64][441] fp32) as an implicit MFMA GEMM (no patch matrix in HBM).",
    "tuber_stem_conv_blocks": "persistent grid size of the stem conv forward = its partial-stat rows.",
    "tuber_stem_conv_wgrad_blocks": "workgroups (= [512][64] fp32 partial slabs) of tuber_stem_conv_bwd_weight.",
    "tuber_stem_pack_weight": "conv1.weight [64][441] fp32 -> [64][512] bf16 with k' = (c,kt,kh)*8 + kw (zero padded).",
    "tuber_stem_pool_fwd": "relu(bn1(.)) + MaxPool3d((1,3,3),s=(1,2,2),p=(0,1,1)) (ir_CSN_152.py:119-122) on NDHWC bf16, C=64; saves the argmax tap.",
    "tuber_stem_pool_bwd": "backward of the pool + relu(bn1(.)): dz and the BN-backward partial statistics.",
    "tuber_stem_pool_bwd_stat_rows": "partial-stat rows written by tuber_stem_pool_bwd.",
    "tuber_attn_fwd": "softmax(scale*QK^T + key_padding_mask)[dropout] V for 32-wide heads, operands read in place through strided token maps "
                      "{ld,sL,s1,s2,B2}: row(l,b) = l*sL + (b/B2)*s1 + (b%B2)*s2. Core of nn.MultiheadAttention "
                      "(transformer.py:159,227,237; tuber_ava.py:138; transformer_layers.py:81,88) and of the hand-rolled MHA (:156-167,306-366).",
    "tuber_attn_bwd": "gradients dQ, dK, dV of tuber_attn_fwd (P recomputed from the saved log-sum-exp).",
    "tuber_attn_wide_fwd": "single-query, 256-wide-head attention of the LSTR pooling decoder (TEMPORAL_DS_STRATEGY decode; "
                           "backbone_builder.py:74-78, transformer_layers.py:156-167,306-366): q [NQ,2048], kv [rows,4096]=[k|v], T<=8 slots per pixel.",
    "tuber_attn_wide_bwd": "gradients dq [NQ,2048] and dkv [rows,4096] of tuber_attn_wide_fwd.",
    "tuber_lsap": "rectangular linear sum assignment on HOST doubles; restates scipy.optimize.linear_sum_assignment (call sites "
                  "models/detr/matcher.py:80, matcher_ucf.py:82) incl. its tie-breaking.",
    "tuber_lsap_device": "the same assignment (scipy.optimize.linear_sum_assignment semantics incl. tie-breaking; matcher.py:80, matcher_ucf.py:82) for all "
                         "(decoder layer, clip) problems at once ON THE DEVICE, one thread per problem: match[l][b][t] = query of target t.",
    "tuber_grad_norm_clip_coef": "global L2 norm of the flat gradient buffer and the clip coefficient min(1, max_norm/(norm+1e-6)), left on the device: "
                                 "torch.nn.utils.clip_grad_norm_(model.parameters(), 0.1) (utils/video_action_recognition.py:153).  A non-finite norm "
                                 "yields coefficient -1 = skip the update and leaves the step count alone: the reference stops BEFORE optimizer.step() on a "
                                 "non-finite loss (video_action_recognition.py:195-198).",
    "tuber_adamw_segment": "AdamW update (torch.optim.AdamW semantics: decoupled decay, bias correction) of one contiguous flat segment with the "
                           "clip coefficient applied to the gradient on the fly (video_action_recognition.py:154; groups train_tuber_ava.py:41-58).",
    "tuber_scale_f32": "x *= coef: gradient averaging after the RCCL all-reduce (DistributedDataParallel's mean, utils/model_utils.py:47-49).",
    "tuber_criterion_cost": "Hungarian matching cost C = w_bbox*L1 - w_giou*GIoU - w_class*p of every decoder layer at once, [L,B,Q,Tmax] fp32 "
                            "(HungarianMatcher.forward, models/detr/matcher.py:61-76 / matcher_ucf.py:61-78; generalized_box_iou utils/box_ops.py:41-65).",
    "tuber_criterion_loss": "all loss terms of all layers and their gradients w.r.t. logits / actor logits / boxes in one launch: AVA 3-way weighted CE + "
                            "weighted BCE (models/criterion.py:42-81), JHMDB (C+1)-way CE (:237-262), L1 + GIoU box losses (:97-117).",
    "tuber_cast_f32_bf16": "fp32 -> bf16 copy (bf16 shadow of the fp32 master weights).",
    "tuber_cast_bf16_f32": "bf16 -> fp32 copy.",
    "tuber_cast_transpose": "W[R][C] fp32 -> W^T[C][ldt] bf16 (B operand of the data-gradient GEMM).",
    "tuber_rows_gather_sum": "out[(a,b,c)] = mul * sum_d in[a*sa+b*sb+c*sc+d*sd] over rows of E bf16: temporal AvgPool3d((4,1,1)) "
                             "(backbone_builder.py:44,73), its backward (broadcast), the x6 replication of src_c (tuber_ava.py:133) and its backward (sum).",
    "tuber_axpby": "out = alpha*a + beta*b (bf16): with_pos_embed adds (transformer.py:150-151), gradient accumulation.",
    "tuber_dropout": "y = keep ? x/(1-p) : 0 with a stateless hash RNG keyed by (seed, index); applying it to a gradient with the same seed is the backward.",
    "tuber_sigmoid_fwd": "boxes = sigmoid(bbox_embed(hs)) (tuber_ava.py:142).",
    "tuber_sigmoid_bwd": "dx = dy*y*(1-y).",
    "tuber_relu_mask": "dx = dy*[h>0] (ReLU backward from the saved activation; FFN and MLP hidden layers).",
    "tuber_multi_transpose_bf16": "every GEMM weight W[R][C] -> W^T[C][ldt] (bf16 shadow of the parameters -> bf16: the B operand of the data-gradient GEMMs, autograd of "
                                  "nn.Conv3d(k=1) / nn.Linear) in one launch over a device table {src_off,dst_off,R,C,ldt,tile_begin,tiles_x,pad} of 64 x 64 tiles.",
    "tuber_cast_pad_rows": "src[R][C] fp32 -> dst[R][ldd] bf16 with zero-filled pad columns (stem 441->448 taps, head gradients).",
    "tuber_rows_scatter_add": "dst[map(m)] += src[m] over the strided (n,t*st,h*ss,w*ss) row map: input gradient of a strided down_sample conv (ir_CSN_152.py:155-161).",
    "tuber_posenc": "PositionEmbeddingSine_3D (models/transformer/position_encoding.py:32-72) of a (B,T,H,W) padding mask, token-major bf16.",
}

HEADER = '''/* tuber_hip.h -- C ABI of libtuber_hip.so: the MI355X (gfx950) kernels behind the TubeR forward/backward path.
 *
 * GENERATED by tubelet_transformer_amd/csrc/gen_header.py from the extern "C" definitions in csrc/*.hip.
 *
 * The reference (amazon-science/tubelet-transformer) has no FFI or operator-plugin boundary: its hot path is stock
 * torch.nn calls (SURVEY.md section 8b).  This ABI is therefore build-defined: one launcher per kernel family, each
 * citing the reference call site(s) whose arithmetic it replaces.  Conventions:
 *   - plain pointers to DEVICE memory + explicit sizes / leading dimensions (in elements); no torch types;
 *   - activations are row-major [rows, channels] bf16 (NDHWC / token-major), master weights and statistics fp32;
 *   - the callee never allocates: outputs, saved tensors and workspaces (`partial`, `st0`, ...) are caller-owned,
 *     sized with the *_rows / *_slabs / *_blocks helper calls;
 *   - every launcher enqueues on `stream` and returns 0, a negative argument-check code (TUBER_EINVAL = -1) or the
 *     positive hipError_t of a failed launch; nothing throws across the ABI.
 */
#ifndef TUBER_HIP_H
#define TUBER_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* hipStream_t;

/* one clip of tuber_clip_prepare (device memory, 56 bytes) */
typedef struct TuberClipDesc {
    long long src_off;   /* byte offset of the clip's first frame in `frames` (uint8 [T][H][W][3]) */
    int H, W;            /* frame size */
    int y1, x1, h, w;    /* crop window in the (flipped) frame: output (y,x) reads (y1+y, x1+x) */
    int flip, jitter;    /* horizontal flip before the crop; HSV colour jitter on/off */
    int hue, sat, val;   /* ColorJitter shifts (hue in OpenCV half-degrees) */
    int pad_;
} TuberClipDesc;

'''
FOOTER = '''
#ifdef __cplusplus
}
#endif
#endif /* TUBER_HIP_H */
'''


def prototypes():
    out = []
    for f in sorted(glob.glob(os.path.join(HERE, "*.hip")) + glob.glob(os.path.join(HERE, "*.cpp"))):
        s = open(f).read()
        for m in re.finditer(r'^(int|long|const char\*) (tuber_\w+)\(([^)]*)\)\s*\{', s, re.M):
            out.append((os.path.basename(f), m.group(1), m.group(2), " ".join(m.group(3).split())))
    return out


def wrap(text, width=116, indent=" * "):
    words, lines, cur = text.split(), [], ""
    for w in words:
        if len(cur) + len(w) + 1 > width:
            lines.append(cur)
            cur = w
        else:
            cur = (cur + " " + w).strip()
    lines.append(cur)
    return "\n".join(indent + l for l in lines)


def main():
    parts = [HEADER]
    last = None
    for f, ret, name, args in prototypes():
        if f != last:
            parts.append("/* ---- %s ---- */\n" % f)
            last = f
        parts.append("/*\n%s\n */\n" % wrap(DOC.get(name, "(undocumented)")))
        parts.append("%s %s(%s);\n\n" % (ret, name, args if args else "void"))
    parts.append(FOOTER)
    os.makedirs(os.path.join(ROOT, "include"), exist_ok=True)
    open(os.path.join(ROOT, "include", "tuber_hip.h"), "w").write("".join(parts))
    missing = [n for _, _, n, _ in prototypes() if n not in DOC]
    if missing:
        print("undocumented:", missing)


if __name__ == "__main__":
    main()
