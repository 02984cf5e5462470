// One Bloom block per call: the launch sequences of modeling_bloom.py:142-159 (forward) and of its hand-derived
// backward, issued from C++ so that the Python host makes ONE call per block and direction instead of 7 / 17+ kernel-level
// calls (the round-1 profile had the host enqueue at 26-35 ms per 43 ms step — above the target step time).
// Host-side orchestration only: every arithmetic step is one of the kernels in gemm.hip / attention.hip / elementwise.hip.
//
// Backward, round 5 (bf16, aligned geometry; CTMI_WGRAD_GROUP != 0): the data-gradient chain below runs on the main stream unchanged; the FOUR
// weight gradients and the column sums of du / dqkv (db1, dbqkv) are ONE grouped launch on the side stream (ctmi_wgrad_grouped, csrc/gemm.hip),
// forked after the attention backward — when the last of their operands, dqkv, exists — so it runs under the rest of this block's chain and
// the next block's.  No split-K slabs of whole gradients, no column-sum launches; one small second launch adds the K-halves of the 128 tiles
// of the last partial round.  The per-product form below remains for fp32 (parity mode)
// and shapes outside the grouped kernel's tiling.
//
// Backward (pre-LN form; the post-LN switch only moves the residual gradients):
//   main stream                                         side stream (weight / bias gradients; optional)
//   du   = (dout W2) * gelu'(u)                          dw2 = dout^T g
//   dln2 = du W1                                         dw1 = du^T ln2,  partial column sums of du   (db1)
//   dh1  = LN2'(dln2) + dout   [+ partial rows of dln2_w, dln2_b, colsum(dout) = db2, colsum(dh1) = dbd]
//   datt = dh1 Wd                                        dwd = dh1^T att
//   dqkv = attention'(datt)
//   dln1 = dqkv Wqkv                                     dwqkv = dqkv^T ln1,  partial column sums of dqkv (dbqkv)
//   dx   = LN1'(dln1) + dh1    [+ partial rows of dln1_w, dln1_b]
//   join;  ONE reduce launch turns all partial rows into the 8 small gradient vectors.
#include "common.h"
#ifndef CTMI_BLOCK_TAIL
#define CTMI_BLOCK_TAIL 1      // 0: the sum of the K-halves right behind the grouped launch (round 5), for A/B builds
#endif
#include <math.h>
#include <stdlib.h>
#include <mutex>

int ctmi_ln_bwd_parts_internal(const void* dy, const void* x, const float* w, const float* mean, const float* rstd,
                               const void* dres, void* dx, float* ws, int64_t rows, int64_t cols, int dtype, int want_sums,
                               int* nparts, int* ns, hipStream_t st);
int ctmi_colsum_parts_internal(const void* x, int64_t ld, float* ws, int64_t M, int64_t N, int dtype, int* parts_out, hipStream_t st);
bool ctmi_wgrad_grouped_ok(const ctmi_wgrad_problem* pr, int n, int64_t T, int dtype);      // csrc/gemm.hip
int ctmi_wgrad_grouped_ex(const ctmi_wgrad_problem* pr, int n, int64_t T, int dtype, void* workspace, int64_t workspace_bytes, void* stream, bool defer_reduce);
int ctmi_wgrad_tail(const ctmi_reduce_job* jobs, int count, hipStream_t st);      // the pending sum of K-halves + these jobs in one launch (csrc/gemm.hip)
void ctmi_wgrad_pending_clear();
bool ctmi_wgrad_pending();

#ifndef CTMI_BLOCK_GELUG
#define CTMI_BLOCK_GELUG 0      // 1: forward saves gelu'(u) (GELUG) and the backward multiplies (MUL); 0: save u, DGELU epilogue.
                                // Same-box A/B (3 interleaved bench runs each): 43.05 vs 43.04 ms/step — the dGELU arithmetic is not
                                // what the [T,4H] data-gradient epilogue costs; the default stays with the reference's saved tensor.
#endif
static inline int64_t al256(int64_t b) { return (b + 255) / 256 * 256; }

extern "C" int64_t ctmi_bloom_block_layout(int64_t B, int64_t S, int64_t H, int64_t nh, int dtype, int64_t* offs) {
    const int64_t T = B * S, e = dtype == CTMI_F32 ? 4 : 2;
    int64_t o = 0;
    auto put = [&](int slot, int64_t bytes) { if (offs) offs[slot] = o; o += al256(bytes); };
    put(CTMI_BLK_LN1, T * H * e);
    put(CTMI_BLK_MEAN1, T * 4);
    put(CTMI_BLK_RSTD1, T * 4);
    put(CTMI_BLK_QKV, T * 3 * H * e);
    put(CTMI_BLK_ATT, T * H * e);
    put(CTMI_BLK_STAT_M, B * nh * S * 4);
    put(CTMI_BLK_STAT_L, B * nh * S * 4);
    put(CTMI_BLK_H1, T * H * e);
    put(CTMI_BLK_MEAN2, T * 4);
    put(CTMI_BLK_RSTD2, T * 4);
    put(CTMI_BLK_LN2, T * H * e);
    put(CTMI_BLK_U, T * 4 * H * e);
    put(CTMI_BLK_G, T * 4 * H * e);
    put(CTMI_BLK_OUT, T * H * e);
    return o;
}

namespace {
struct Slab {
    char* base; int64_t off[CTMI_BLK_NSLOTS];
    template <typename P = void> P* at(int slot) const { return reinterpret_cast<P*>(base + off[slot]); }
};

int check_block(const ctmi_bloom_block* b, const char* who) {
    CTMI_REQUIRE(b != nullptr, "%s: null block descriptor", who);
    CTMI_REQUIRE(b->B > 0 && b->S > 0 && b->H > 0 && b->nh > 0 && b->H % b->nh == 0, "%s: bad geometry B=%lld S=%lld H=%lld nh=%lld", who,
                 (long long)b->B, (long long)b->S, (long long)b->H, (long long)b->nh);
    CTMI_REQUIRE(b->dtype == CTMI_F32 || b->dtype == CTMI_BF16 || b->dtype == CTMI_F16, "%s: unsupported dtype %d", who, b->dtype);
    CTMI_REQUIRE(b->ln1_w && b->ln1_b && b->wqkv && b->bqkv && b->wd && b->bd && b->ln2_w && b->ln2_b && b->w1 && b->b1 && b->w2 && b->b2,
                 "%s: null parameter pointer", who);
    CTMI_REQUIRE(b->x && b->slab, "%s: null activation pointer", who);
    CTMI_REQUIRE((b->slopes == nullptr) == (b->kpos == nullptr), "%s: slopes and kpos go together", who);
    CTMI_REQUIRE(b->kvalid == nullptr || b->first_valid != nullptr, "%s: kvalid needs first_valid", who);
    return CTMI_OK;
}

ctmi_attn_desc fused_qkv_desc(const ctmi_bloom_block* b) {
    // fused QKV activation: head-interleaved [B,S,nh,3,hd] (modeling_bloom.py:81-82) or GPT-2's q | k | v [B,S,3,nh,hd]
    // (modeling_gpt.py:69-72); merged-head context [B,S,nh*hd]
    const int64_t H = b->H, hd = H / b->nh, S = b->S;
    const bool blocked = (b->flags & CTMI_BLK_QKV_BLOCKED) != 0;
    ctmi_attn_desc d = {};
    d.B = b->B; d.nh = b->nh; d.Sq = S; d.Sk = S; d.hd = hd;
    d.q_bs = d.k_bs = d.v_bs = S * 3 * H; d.q_hs = d.k_hs = d.v_hs = blocked ? hd : 3 * hd; d.q_rs = d.k_rs = d.v_rs = 3 * H;
    d.o_bs = S * H; d.o_hs = hd; d.o_rs = H;
    d.scale = b->attn_scale != 0.0f ? b->attn_scale : 1.0f / sqrtf((float)hd);
    d.causal = S > 1 ? 1 : 0;
    d.future_fill = b->future_fill;
    return d;
}
// element offset of the k (which = 1) / v (which = 2) part of row 0 inside the fused activation
int64_t qkv_part(const ctmi_bloom_block* b, int which) {
    const int64_t H = b->H, hd = H / b->nh;
    return (b->flags & CTMI_BLK_QKV_BLOCKED) ? which * H : which * hd;
}

// y[T,N] = epi(x[T,K] W[N,K]^T + bias) (+ residual); w_in_out: the weight is stored [K,N] (GPT-2's Conv1D, modeling_gpt.py:32-46) and
// is read K-major — no transposed copy of it exists
int linear_fwd(const void* x, const void* w, void* y, int64_t T, int64_t N, int64_t K, const float* bias, const void* residual,
               int epi, void* aux_out, int dtype, hipStream_t st, bool w_in_out = false) {
    return ctmi_gemm(x, K, 0, w, w_in_out ? N : K, w_in_out ? 1 : 0, y, N, T, N, K, 1.0f, 0, bias, residual, epi, nullptr, aux_out, 0, dtype,
                     nullptr, 0, st);
}
// dx[T,Kin] = epi(dy[T,Nout] W[Nout,Kin]) (+ residual); w_in_out: W stored [Kin,Nout], read as the row-major B operand
int linear_dgrad(const void* dy, const void* w, void* dx, int64_t T, int64_t Nout, int64_t Kin, int epi, const void* aux_in,
                 const void* residual, int dtype, void* ws, int64_t ws_bytes, hipStream_t st, bool w_in_out = false) {
    const bool plain = epi == CTMI_EPI_NONE && residual == nullptr;
    return ctmi_gemm(dy, Nout, 0, w, w_in_out ? Nout : Kin, w_in_out ? 0 : 1, dx, Kin, T, Kin, Nout, 1.0f, 0, nullptr, residual, epi, aux_in,
                     nullptr, 0, dtype, plain ? ws : nullptr, plain ? ws_bytes : 0, st);
}
// dW[Nout,Kin] (fp32) = dy[T,Nout]^T x[T,Kin]   — or, in_out: dW[Kin,Nout] = x^T dy (a Conv1D weight is stored [in,out])
int linear_wgrad(const void* dy, const void* x, float* dw, int64_t T, int64_t Nout, int64_t Kin, int dtype, void* ws, int64_t ws_bytes,
                 hipStream_t st, bool in_out = false) {
    { static int dbg = -1; if (dbg < 0) { const char* e = getenv("CTMI_BLOCK_DBG"); dbg = e ? atoi(e) : 0; }
      if (dbg & 1) return CTMI_OK; }                                    // timing experiments only: 1 = skip the layer weight-gradient GEMMs
    if (in_out)
        return ctmi_gemm(x, Kin, 1, dy, Nout, 1, dw, Nout, Kin, Nout, T, 1.0f, 0, nullptr, nullptr, CTMI_EPI_NONE, nullptr, nullptr, 1, dtype,
                         ws, ws_bytes, st);
    return ctmi_gemm(dy, Nout, 1, x, Kin, 1, dw, Kin, Nout, Kin, T, 1.0f, 0, nullptr, nullptr, CTMI_EPI_NONE, nullptr, nullptr, 1, dtype,
                     ws, ws_bytes, st);
}

// fork/join events (no timing).  An event may be re-recorded while an earlier wait on it is still queued: the wait refers to
// the record that preceded it.  One small pool per device, created on first use.
struct EventPool {
    static constexpr int N = 16;
    hipEvent_t ev[N]; int next = 0; bool ok = false;
};
std::mutex g_ev_mu;
EventPool g_pools[16];
hipEvent_t next_event() {
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(g_ev_mu);
    EventPool& P = g_pools[dev & 15];
    if (!P.ok) {
        for (int i = 0; i < EventPool::N; ++i) (void)hipEventCreateWithFlags(&P.ev[i], hipEventDisableTiming);
        P.ok = true;
    }
    hipEvent_t e = P.ev[P.next];
    P.next = (P.next + 1) % EventPool::N;
    return e;
}
#define CTMI_HIP_OK(call, what) do { hipError_t e__ = (call); if (e__ != hipSuccess) { \
    ctmi_set_error("%s: %s", what, hipGetErrorString(e__)); return CTMI_ERR_LAUNCH; } } while (0)
#define RC(call) do { int rc__ = (call); if (rc__ != CTMI_OK) return rc__; } while (0)
}  // namespace

extern "C" int ctmi_bloom_block_fwd(const ctmi_bloom_block* b, void* stream) {
    RC(check_block(b, "bloom_block_fwd"));
    hipStream_t st = as_stream(stream);
    const int64_t T = b->B * b->S, H = b->H, hd = H / b->nh;
    const int dt = b->dtype, e = dt == CTMI_F32 ? 4 : 2;
    Slab s; s.base = reinterpret_cast<char*>(b->slab);
    ctmi_bloom_block_layout(b->B, b->S, H, b->nh, dt, s.off);
    const bool post = b->post_ln_res != 0;
    const bool w_io = (b->flags & CTMI_BLK_W_IN_OUT) != 0;

    RC(ctmi_layernorm_fwd(b->x, b->ln1_w, b->ln1_b, s.at(CTMI_BLK_LN1), s.at<float>(CTMI_BLK_MEAN1), s.at<float>(CTMI_BLK_RSTD1), T, H, b->eps, dt, st));
    RC(linear_fwd(s.at(CTMI_BLK_LN1), b->wqkv, s.at(CTMI_BLK_QKV), T, 3 * H, H, b->bqkv, nullptr, CTMI_EPI_NONE, nullptr, dt, st, w_io));
    const ctmi_attn_desc d = fused_qkv_desc(b);
    char* qkv = s.at<char>(CTMI_BLK_QKV);
    RC(ctmi_attn_fwd(qkv, qkv + qkv_part(b, 1) * e, qkv + qkv_part(b, 2) * e, s.at(CTMI_BLK_ATT), s.at<float>(CTMI_BLK_STAT_M), s.at<float>(CTMI_BLK_STAT_L),
                     b->slopes, b->kpos, b->kvalid, b->first_valid, nullptr, &d, dt, st));
    RC(linear_fwd(s.at(CTMI_BLK_ATT), b->wd, s.at(CTMI_BLK_H1), T, H, H, b->bd, post ? s.at(CTMI_BLK_LN1) : b->x, CTMI_EPI_NONE, nullptr, dt, st, w_io));
    RC(ctmi_layernorm_fwd(s.at(CTMI_BLK_H1), b->ln2_w, b->ln2_b, s.at(CTMI_BLK_LN2), s.at<float>(CTMI_BLK_MEAN2), s.at<float>(CTMI_BLK_RSTD2), T, H, b->eps, dt, st));
    RC(linear_fwd(s.at(CTMI_BLK_LN2), b->w1, s.at(CTMI_BLK_G), T, 4 * H, H, b->b1, nullptr, CTMI_BLOCK_GELUG ? CTMI_EPI_GELUG : CTMI_EPI_GELU, s.at(CTMI_BLK_U), dt, st, w_io));
    RC(linear_fwd(s.at(CTMI_BLK_G), b->w2, s.at(CTMI_BLK_OUT), T, H, 4 * H, b->b2, post ? s.at(CTMI_BLK_LN2) : s.at(CTMI_BLK_H1), CTMI_EPI_NONE, nullptr, dt, st, w_io));
    return CTMI_OK;
}

namespace {
enum BwdSlot { W_DU = 0, W_DLN2, W_DH1, W_DATT, W_DQKV, W_DLN1, W_DELTA, W_LNP2, W_LNP1, W_CS_DU, W_CS_DQKV, W_CS_DOUT, W_CS_DH1, W_NSLOTS };
int64_t bwd_layout(int64_t B, int64_t S, int64_t H, int64_t nh, int dtype, int64_t* offs) {
    const int64_t T = B * S, e = dtype == CTMI_F32 ? 4 : 2;
    int64_t o = 0;
    auto put = [&](int slot, int64_t bytes) { if (offs) offs[slot] = o; o += al256(bytes); };
    put(W_DU, T * 4 * H * e);
    put(W_DLN2, T * H * e);
    put(W_DH1, T * H * e);
    put(W_DATT, T * H * e);
    put(W_DQKV, T * 3 * H * e);
    put(W_DLN1, T * H * e);
    put(W_DELTA, B * nh * S * 4);
    put(W_LNP2, ctmi_layernorm_bwd_ws(T, H) * 4);
    put(W_LNP1, ctmi_layernorm_bwd_ws(T, H) * 4);
    put(W_CS_DU, ctmi_colsum_ws(T, 4 * H) * 4);
    put(W_CS_DQKV, ctmi_colsum_ws(T, 3 * H) * 4);
    put(W_CS_DOUT, ctmi_colsum_ws(T, H) * 4);
    put(W_CS_DH1, ctmi_colsum_ws(T, H) * 4);
    return o;
}
}  // namespace

// does ctmi_bloom_block_bwd take the grouped weight-gradient launch at this geometry?  (Then the whole backward of a block is best issued on ONE
// stream: the grouped launch fills the 256 CUs by itself — profiles/r05_wgrad_grouped.txt — and callers keep side_stream NULL.)
extern "C" int ctmi_bloom_block_wgrad_grouped(int64_t B, int64_t S, int64_t H, int dtype, int flags) {
    if (B <= 0 || S <= 0 || H <= 0) return 0;
    { const char* e = getenv("CTMI_BLOCK_DBG"); if (e && (atoi(e) & 1)) return 0; }
    const int io = (flags & CTMI_BLK_WGRAD_IN_OUT) ? 1 : 0;
    void* const al = reinterpret_cast<void*>(uintptr_t(256));                          // stands for any 16-byte aligned pointer
    float* const fa = reinterpret_cast<float*>(al);
    const ctmi_wgrad_problem wp[4] = {{al, al, fa, nullptr, H, 4 * H, io, 0}, {al, al, fa, io ? nullptr : fa, 4 * H, H, io, 0},
                                      {al, al, fa, nullptr, H, H, io, 0}, {al, al, fa, io ? nullptr : fa, 3 * H, H, io, 0}};
    return ctmi_wgrad_grouped_ok(wp, 4, B * S, dtype) ? 1 : 0;
}

extern "C" int64_t ctmi_bloom_block_bwd_ws(int64_t B, int64_t S, int64_t H, int64_t nh, int dtype) {
    return bwd_layout(B, S, H, nh, dtype, nullptr);
}

extern "C" int ctmi_bloom_block_bwd(const ctmi_bloom_block* b, const ctmi_bloom_block_grads* gr, void* stream) {
    RC(check_block(b, "bloom_block_bwd"));
    CTMI_REQUIRE(gr != nullptr && gr->dout && gr->dx && gr->ws, "bloom_block_bwd: null gradient pointer / workspace");
    CTMI_REQUIRE(gr->dln1_w && gr->dln1_b && gr->dwqkv && gr->dbqkv && gr->dwd && gr->dbd && gr->dln2_w && gr->dln2_b && gr->dw1 && gr->db1 &&
                 gr->dw2 && gr->db2, "bloom_block_bwd: null parameter-gradient pointer");
    const int64_t T = b->B * b->S, H = b->H, hd = H / b->nh;
    const int dt = b->dtype, e = dt == CTMI_F32 ? 4 : 2;
    int64_t woff[W_NSLOTS];
    const int64_t need = bwd_layout(b->B, b->S, H, b->nh, dt, woff);
    CTMI_REQUIRE(gr->ws_bytes >= need, "bloom_block_bwd: workspace too small (%lld < %lld bytes)", (long long)gr->ws_bytes, (long long)need);
    Slab s; s.base = reinterpret_cast<char*>(b->slab);
    ctmi_bloom_block_layout(b->B, b->S, H, b->nh, dt, s.off);
    char* wsb = reinterpret_cast<char*>(gr->ws);
    auto W = [&](int slot) { return reinterpret_cast<void*>(wsb + woff[slot]); };
    auto WF = [&](int slot) { return reinterpret_cast<float*>(wsb + woff[slot]); };
    const bool post = b->post_ln_res != 0;
    const bool wio = (b->flags & CTMI_BLK_WGRAD_IN_OUT) != 0;
    const bool w_io = (b->flags & CTMI_BLK_W_IN_OUT) != 0;
    hipStream_t main_st = as_stream(stream);
    hipStream_t side = gr->side_stream ? as_stream(gr->side_stream) : nullptr;
    const bool two = side != nullptr && side != main_st;
    hipStream_t pst = two ? side : main_st;                              // stream of the parameter-gradient work
    void* pws = two ? gr->side_splitk_ws : gr->splitk_ws;
    const int64_t pws_bytes = two ? gr->side_splitk_ws_bytes : gr->splitk_ws_bytes;
    // the side stream picks up after everything enqueued on the main stream so far (the producer of the operands just named)
    auto fork = [&]() -> int {
        if (!two) return CTMI_OK;
        hipEvent_t ev = next_event();
        CTMI_HIP_OK(hipEventRecord(ev, main_st), "bloom_block_bwd: event record");
        CTMI_HIP_OK(hipStreamWaitEvent(side, ev, 0), "bloom_block_bwd: stream wait");
        return CTMI_OK;
    };

    ctmi_wgrad_pending_clear();                                          // (a previous call of this thread that failed half-way)
    ctmi_reduce_job jobs[CTMI_REDUCE_MAX_JOBS];
    int nj = 0;
    auto job = [&](const float* src, int64_t stride, int nparts, float* dst, int64_t n) {
        ctmi_reduce_job& J = jobs[nj++];
        J.src = src; J.dst = dst; J.n = n; J.part_stride = stride; J.nparts = nparts; J.accumulate = 0; J.alpha = 1.0f; J.pad_ = 0;
    };
    auto colsum_job = [&](const void* x, int64_t N, int slot, float* dst) -> int {   // partial rows on the parameter stream
        int parts = 0;
        RC(ctmi_colsum_parts_internal(x, N, WF(slot), T, N, dt, &parts, pst));
        job(WF(slot), N, parts, dst, N);
        return CTMI_OK;
    };

    const void* dout = gr->dout;
    static int blk_dbg = -1;
    if (blk_dbg < 0) { const char* e = getenv("CTMI_BLOCK_DBG"); blk_dbg = e ? atoi(e) : 0; }
    // the four weight gradients as one grouped launch?  (list order = split order: the tiles of the last partial round — dwd, dwqkv at
    // Bloom-560M — are the ones cut in two along T)
    ctmi_wgrad_problem wp[4] = {
        {dout, s.at(CTMI_BLK_G), gr->dw2, nullptr, H, 4 * H, wio ? 1 : 0, 0},
        {W(W_DU), s.at(CTMI_BLK_LN2), gr->dw1, gr->db1, 4 * H, H, wio ? 1 : 0, 0},        // (round 6: the column sums of du / dqkv ride in the grouped launch for [in,out] weights too — there as sums of its B operand)
        {W(W_DH1), s.at(CTMI_BLK_ATT), gr->dwd, nullptr, H, H, wio ? 1 : 0, 0},
        {W(W_DQKV), s.at(CTMI_BLK_LN1), gr->dwqkv, gr->dbqkv, 3 * H, H, wio ? 1 : 0, 0},
    };
    const bool grouped = !(blk_dbg & 1) && ctmi_wgrad_grouped_ok(wp, 4, T, dt);
    // The launches run inside a lambda so that a failure half-way still reaches the join below: the side stream may already hold
    // work that reads the caller's buffers, and the caller (who frees them on error) only orders against the main stream.
    const int rc_launch = [&]() -> int {
    if (grouped) {
        RC(linear_dgrad(dout, b->w2, W(W_DU), T, H, 4 * H, CTMI_BLOCK_GELUG ? CTMI_EPI_MUL : CTMI_EPI_DGELU, s.at(CTMI_BLK_U), nullptr, dt, nullptr, 0, main_st, w_io));
        RC(linear_dgrad(W(W_DU), b->w1, W(W_DLN2), T, 4 * H, H, CTMI_EPI_NONE, nullptr, post ? dout : nullptr, dt, gr->splitk_ws, gr->splitk_ws_bytes, main_st, w_io));
        int np2 = 0, ns2 = 2;
        RC(ctmi_ln_bwd_parts_internal(W(W_DLN2), s.at(CTMI_BLK_H1), b->ln2_w, s.at<float>(CTMI_BLK_MEAN2), s.at<float>(CTMI_BLK_RSTD2),
                                      post ? nullptr : dout, W(W_DH1), WF(W_LNP2), T, H, dt, 1, &np2, &ns2, main_st));
        job(WF(W_LNP2), ns2 * H, np2, gr->dln2_w, H);
        job(WF(W_LNP2) + H, ns2 * H, np2, gr->dln2_b, H);
        RC(linear_dgrad(W(W_DH1), b->wd, W(W_DATT), T, H, H, CTMI_EPI_NONE, nullptr, nullptr, dt, gr->splitk_ws, gr->splitk_ws_bytes, main_st, w_io));
        const ctmi_attn_desc d = fused_qkv_desc(b);
        char* qkv = s.at<char>(CTMI_BLK_QKV);
        char* dqkv = reinterpret_cast<char*>(W(W_DQKV));
        RC(ctmi_attn_bwd(qkv, qkv + qkv_part(b, 1) * e, qkv + qkv_part(b, 2) * e, s.at(CTMI_BLK_ATT), W(W_DATT), s.at<float>(CTMI_BLK_STAT_M), s.at<float>(CTMI_BLK_STAT_L),
                         dqkv, dqkv + qkv_part(b, 1) * e, dqkv + qkv_part(b, 2) * e, WF(W_DELTA), b->slopes, b->kpos, b->kvalid, b->first_valid, nullptr, &d, dt, main_st));
        // every operand of the four weight gradients exists now
        RC(fork());
        if (ns2 == 4 && !post) job(WF(W_LNP2) + 2 * H, ns2 * H, np2, gr->db2, H);
        else RC(colsum_job(dout, H, W_CS_DOUT, gr->db2));
        if (ns2 == 4) job(WF(W_LNP2) + 3 * H, ns2 * H, np2, gr->dbd, H);
        else RC(colsum_job(W(W_DH1), H, W_CS_DH1, gr->dbd));
        // one stream (the default of the grouped path): the sum of the K-halves waits for the block's last launch and shares it with the
        // partial-row reductions (ctmi_wgrad_tail below) — the data gradient in between then gets no split-K workspace: the slabs of the halves live there
        const bool tail = !two && CTMI_BLOCK_TAIL;
        RC(ctmi_wgrad_grouped_ex(wp, 4, T, dt, pws, pws_bytes, pst, tail));
        const bool pend = ctmi_wgrad_pending();
        RC(linear_dgrad(dqkv, b->wqkv, W(W_DLN1), T, 3 * H, H, CTMI_EPI_NONE, nullptr, post ? W(W_DH1) : nullptr, dt, pend ? nullptr : gr->splitk_ws, pend ? 0 : gr->splitk_ws_bytes, main_st, w_io));
        int np1 = 0, ns1 = 2;
        RC(ctmi_ln_bwd_parts_internal(W(W_DLN1), b->x, b->ln1_w, s.at<float>(CTMI_BLK_MEAN1), s.at<float>(CTMI_BLK_RSTD1),
                                      post ? nullptr : W(W_DH1), gr->dx, WF(W_LNP1), T, H, dt, 0, &np1, &ns1, main_st));
        job(WF(W_LNP1), ns1 * H, np1, gr->dln1_w, H);
        job(WF(W_LNP1) + H, ns1 * H, np1, gr->dln1_b, H);
        return CTMI_OK;
    }
    // ---- MLP: out = res2 + W2 gelu(W1 ln2 + b1) + b2
    RC(fork());
    RC(linear_wgrad(dout, s.at(CTMI_BLK_G), gr->dw2, T, H, 4 * H, dt, pws, pws_bytes, pst, wio));
    RC(linear_dgrad(dout, b->w2, W(W_DU), T, H, 4 * H, CTMI_BLOCK_GELUG ? CTMI_EPI_MUL : CTMI_EPI_DGELU, s.at(CTMI_BLK_U), nullptr, dt, nullptr, 0, main_st, w_io));        // modeling_bloom.py:348-363 fused
    RC(fork());
    RC(linear_wgrad(W(W_DU), s.at(CTMI_BLK_LN2), gr->dw1, T, 4 * H, H, dt, pws, pws_bytes, pst, wio));
    RC(colsum_job(W(W_DU), 4 * H, W_CS_DU, gr->db1));
    RC(linear_dgrad(W(W_DU), b->w1, W(W_DLN2), T, 4 * H, H, CTMI_EPI_NONE, nullptr, post ? dout : nullptr, dt, gr->splitk_ws, gr->splitk_ws_bytes, main_st, w_io));
    int np2 = 0, ns2 = 2;
    RC(ctmi_ln_bwd_parts_internal(W(W_DLN2), s.at(CTMI_BLK_H1), b->ln2_w, s.at<float>(CTMI_BLK_MEAN2), s.at<float>(CTMI_BLK_RSTD2),
                                  post ? nullptr : dout, W(W_DH1), WF(W_LNP2), T, H, dt, 1, &np2, &ns2, main_st));
    job(WF(W_LNP2), ns2 * H, np2, gr->dln2_w, H);
    job(WF(W_LNP2) + H, ns2 * H, np2, gr->dln2_b, H);
    // ---- attention: h1 = res1 + Wd att + bd
    RC(fork());
    // bias gradients of the two [T,H]-output Linears: column sums of dout (4h->h) and of dh1 (dense)
    if (ns2 == 4 && !post) job(WF(W_LNP2) + 2 * H, ns2 * H, np2, gr->db2, H);
    else RC(colsum_job(dout, H, W_CS_DOUT, gr->db2));
    if (ns2 == 4) job(WF(W_LNP2) + 3 * H, ns2 * H, np2, gr->dbd, H);
    else RC(colsum_job(W(W_DH1), H, W_CS_DH1, gr->dbd));
    RC(linear_wgrad(W(W_DH1), s.at(CTMI_BLK_ATT), gr->dwd, T, H, H, dt, pws, pws_bytes, pst, wio));
    RC(linear_dgrad(W(W_DH1), b->wd, W(W_DATT), T, H, H, CTMI_EPI_NONE, nullptr, nullptr, dt, gr->splitk_ws, gr->splitk_ws_bytes, main_st, w_io));
    const ctmi_attn_desc d = fused_qkv_desc(b);
    char* qkv = s.at<char>(CTMI_BLK_QKV);
    char* dqkv = reinterpret_cast<char*>(W(W_DQKV));
    RC(ctmi_attn_bwd(qkv, qkv + qkv_part(b, 1) * e, qkv + qkv_part(b, 2) * e, s.at(CTMI_BLK_ATT), W(W_DATT), s.at<float>(CTMI_BLK_STAT_M), s.at<float>(CTMI_BLK_STAT_L),
                     dqkv, dqkv + qkv_part(b, 1) * e, dqkv + qkv_part(b, 2) * e, WF(W_DELTA), b->slopes, b->kpos, b->kvalid, b->first_valid, nullptr, &d, dt, main_st));
    RC(fork());
    RC(linear_wgrad(dqkv, s.at(CTMI_BLK_LN1), gr->dwqkv, T, 3 * H, H, dt, pws, pws_bytes, pst, wio));
    RC(colsum_job(dqkv, 3 * H, W_CS_DQKV, gr->dbqkv));
    RC(linear_dgrad(dqkv, b->wqkv, W(W_DLN1), T, 3 * H, H, CTMI_EPI_NONE, nullptr, post ? W(W_DH1) : nullptr, dt, gr->splitk_ws, gr->splitk_ws_bytes, main_st, w_io));
    int np1 = 0, ns1 = 2;
    RC(ctmi_ln_bwd_parts_internal(W(W_DLN1), b->x, b->ln1_w, s.at<float>(CTMI_BLK_MEAN1), s.at<float>(CTMI_BLK_RSTD1),
                                  post ? nullptr : W(W_DH1), gr->dx, WF(W_LNP1), T, H, dt, 0, &np1, &ns1, main_st));
    job(WF(W_LNP1), ns1 * H, np1, gr->dln1_w, H);
    job(WF(W_LNP1) + H, ns1 * H, np1, gr->dln1_b, H);
    return CTMI_OK;
    }();
    // ---- join (also after a failed launch): everything downstream (autograd accumulation, hooks, optimizer, the caller's frees)
    // is ordered on the main stream
    // defer_join (round 4): the caller orders everything that consumes the parameter gradients behind the side stream itself (one join at
    // the end of the whole backward pass) and keeps this call's scratch alive until then; the partial-row reductions then run at the end of
    // the side stream's queue and the main stream goes straight on to the next block — the last weight gradient of a block no longer
    // holds up the first data gradient of the next (measured with the join simply dropped: -0.8 ms per step).
    // (a launch that failed half-way takes the JOINED path below even in deferred mode: the side stream may already hold work that reads dout,
    // the slab and ws, and the caller — who raises before its own record_stream / slot-event bookkeeping — only orders against the main stream)
    if (two && gr->defer_join && rc_launch == CTMI_OK) {
        RC(fork());                                                             // the LayerNorm partial rows are written on the main stream
        RC(ctmi_reduce_jobs(jobs, nj, side));
        return CTMI_OK;
    }
    if (two) {
        hipEvent_t ev = next_event();
        const hipError_t e1 = hipEventRecord(ev, side);
        const hipError_t e2 = e1 == hipSuccess ? hipStreamWaitEvent(main_st, ev, 0) : e1;
        if (e2 != hipSuccess && rc_launch == CTMI_OK) { ctmi_set_error("bloom_block_bwd: joining the side stream: %s", hipGetErrorString(e2)); return CTMI_ERR_LAUNCH; }
    }
    if (rc_launch != CTMI_OK) { ctmi_wgrad_pending_clear(); return rc_launch; }
    RC(ctmi_wgrad_tail(jobs, nj, main_st));
    return CTMI_OK;
}
