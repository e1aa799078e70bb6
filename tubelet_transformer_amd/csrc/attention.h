// Argument block shared by the attention kernels (attention.hip: scalar flash kernels; attention_mfma.hip: MFMA kernels).
#pragma once
#include "common.h"

struct TokMap { long ld; long sL, s1, s2; int B2; };

struct AttnArgs {
    const bf16* Q; TokMap mq;
    const bf16* K; TokMap mk;
    const bf16* V; TokMap mv;
    bf16* O; TokMap mo;
    float* lse;                   // [B][H][Lq]
    const uint8_t* kpm;           // [B][Lk] (1 = padded key) or null
    int B, H, Lq, Lk;
    float scale, pdrop; uint32_t thresh; const uint64_t* seed_ptr; uint64_t salt;
    // backward
    const bf16* dO; TokMap mdo;
    bf16* dQ; TokMap mdq;
    bf16* dK; TokMap mdk;
    bf16* dV; TokMap mdv;
    float* delta;                 // [B][H][Lq]
};

// MFMA path (attention_mfma.hip), taken for Lq >= 32
extern "C" __attribute__((visibility("hidden"))) void tuber_attn_mfma_fwd_launch(const void* args, hipStream_t stream);
extern "C" __attribute__((visibility("hidden"))) void tuber_attn_mfma_bwd_launch(const void* args, hipStream_t stream);
