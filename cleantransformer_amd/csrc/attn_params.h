// Kernel-argument block shared by the attention kernels (attention.hip: 16x16 MFMA general path; attention_w32.hip: the
// 32x32 MFMA fast path for the training shapes).
#pragma once
#include "common.h"

struct AttnP {
    const void *q, *k, *v, *o, *d_o;
    void *out, *dq, *dk, *dv;
    float *stat_m, *stat_l, *delta;
    const float *slopes, *kpos, *add_mask;
    const int32_t *kvalid, *first_valid;
    int64_t B, nh, Sq, Sk, hd;
    int64_t q_bs, q_hs, q_rs, k_bs, k_hs, k_rs, v_bs, v_hs, v_rs, o_bs, o_hs, o_rs;
    int64_t am_b, am_h, am_q, am_k;
    float scale;
    uint32_t drop_thr, drop_seed;  // attention-probability dropout: keep(b,h,q,k) = hash32(counter ^ seed) >= thr (0 = off)
    float drop_scale;        // 1 / (1 - p)
    float future_fill;       // score of a (query, key) pair in the causal future whose key may be attended: FINFO_MIN (Bloom masked_fill) or GPT's -1e4
    int causal, off, vec_ok;
    int dbg;                 // timing experiments only (CTMI_ATTN_DBG): 1 = no steady-state global loads, 2 = no LDS restage
};

// fast path (attention_w32.hip): returns 1 when it launched the kernels, 0 when the problem is outside its envelope (the
// caller then takes the general kernels), < 0 never.
int ctmi_attn32_fwd(const AttnP& p, hipStream_t st, int f16 = 0);      // f16: IEEE-half operands (round 5)
int ctmi_attn32_bwd(const AttnP& p, hipStream_t st, int f16 = 0);
