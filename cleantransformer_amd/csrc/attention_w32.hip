// Fast path of the fused attention for the TRAINING shapes (bf16, causal, Sq == Sk a multiple of 64, head_dim 64 or 128, no
// additive mask, no probability dropout): same contract and same numerics policy as attention.hip (which keeps every other
// case), re-tiled for matrix density.
//   reference: modeling_bloom.py:84-116 (BloomAttentionLayer core: baddbmm(alibi, q, k^T, alpha = 1/sqrt(hd)) -> masked_fill(finfo.min)
//              -> fp32 softmax -> bmm with v), modeling_gpt.py:76-101 (the same with the -1e4 causal replacement)
//
// Why a second set of kernels.  attention.hip gives each wave 16 own rows and a 64-row workgroup one K/V stage: 16 MFMAs
// (16x16x32) per ~160 VALU + ~60 SALU instructions and one LDS restage per 64 rows — measured 17 % matrix-pipe busy.  Here
//   * a workgroup owns 256 rows (8 waves x 32) — or 128 (4 waves) where 32 own rows need > 256 registers — and ONE K/V (or
//     Q/dO) stage feeds all of them: 4x less staging traffic per flop, and the stage arrives by LDS-DMA
//     (global_load_lds_dwordx4, 1 KiB per wave-instruction) into a 3-deep ring — no staging registers, no ds_write, one
//     s_barrier per tile, counted vmcnt;
//   * v_mfma_f32_32x32x16_bf16 with the wave's own row in the accumulator COLUMN (lane & 31): a lane owns one query (or
//     key) row, softmax statistics are per-lane scalars, and a row's 32 scores of a key block sit in just two lanes
//     (l, l + 32), so the row max / sum cross lanes once per tile (v_permlane32_swap);
//   * the probabilities leave the first MFMA already in the B-operand layout of the second one: the k-slot <-> key
//     assignment of the second contraction is CHOSEN to be the accumulator's (keys (r&3) + 8*(r>>2) + 4*(lane>>5)), and the
//     transposed A operand (V^T, K^T, Q^T, dO^T) is gathered to match with ds_read_b64_tr_b16 from the same row-major tile;
//   * LDS tiles are lane-linear DMA images; bank conflicts are removed by XOR-ing the 16-byte chunk index with row bits
//     (on the SOURCE address of the DMA and on every fragment read), one permutation that is conflict-free for both the
//     ds_read_b128 row fragments and the transposed reads;
//   * softmax in the log2 domain (scale * log2(e) folded into the score FMA, exp2 directly), per-key bias
//     (ALiBi slope * position, finfo.min for padding keys) precomputed once per workgroup into LDS;
//   * causal work balance: query blocks are issued longest-first (LPT), and inside a workgroup the two waves that share a
//     SIMD own row blocks (i, 7 - i), so every SIMD sees the same number of un-skipped diagonal tiles.
// Masked scores take finfo.min exactly like the reference's masked_fill (all-masked rows become uniform over ALL keys);
// row statistics are published in natural units (m, l) exactly as attention.hip does, so forward / backward kernels of the
// two files can be mixed.
#include "common.h"
#include "attn_params.h"
#include <stdlib.h>
#include <atomic>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8_w32 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_w32 __attribute__((ext_vector_type(8)));
typedef short short4_w32 __attribute__((ext_vector_type(4)));

// forward: 4-wave workgroups (128 query rows), three per CU at head_dim 64 (168 registers, 3 waves per SIMD from workgroups that no
// barrier couples); K/V ring depth 2 (40 KiB of LDS per workgroup at S = 1024: three fit)
#ifndef CTMI_W32_FWD_NW
#define CTMI_W32_FWD_NW 4
#endif
#ifndef CTMI_W32_FWD_NST
#define CTMI_W32_FWD_NST 2
#endif
// backward, head_dim 64: 4-wave workgroups too — dQ three per CU (168 registers), dK/dV two per CU (256 registers)
#ifndef CTMI_W32_BWD_NW
#define CTMI_W32_BWD_NW 4
#endif
#ifndef CTMI_W32_BWD_NST
#define CTMI_W32_BWD_NST 2
#endif
// forward: log2 of the growth of a row maximum that is tolerated before O / l are rescaled (0 = rescale whenever a maximum moved)
#ifndef CTMI_W32_DEFER_MAX
#define CTMI_W32_DEFER_MAX 0
#endif
// -DCTMI_W32_TIMING=1 (tools/ variant builds only): the forward kernel accumulates s_memtime deltas per wave — matrix segments,
// vector segments, barrier waits after each — and dumps them over stat_l (which is then garbage): tools/attn_w32_timing.py
#ifndef CTMI_W32_TIMING
#define CTMI_W32_TIMING 0
#endif
#if CTMI_W32_TIMING
#define W32_TICK(acc) do { const uint64_t now__ = __builtin_amdgcn_s_memtime(); acc += (uint32_t)(now__ - tlast); tlast = now__; } while (0)
#else
#define W32_TICK(acc) do { } while (0)
#endif
// forward: which segment raises its wave priority (0 none, 1 the matrix segment, 2 the vector segment)
#ifndef CTMI_W32_PRIO
#define CTMI_W32_PRIO 1
#endif
#define LOG2E_F 1.4426950408889634f
#define LN2_F 0.6931471805599453f

namespace {

// F16 (round 5): the same kernels on IEEE-half operands — v_mfma_f32_32x32x16_f16, half conversions of P / dS and of the outputs; everything
// between the matrix instructions (scores, statistics, masks as C operands) is fp32 and does not change
template <bool F16> __device__ __forceinline__ f32x16 mfma32(short8 a, short8 b, f32x16 c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_w32, a), __builtin_bit_cast(f16x8_w32, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_w32, a), __builtin_bit_cast(bf16x8_w32, b), c, 0, 0, 0);
}
template <bool F16> __device__ __forceinline__ uint32_t pk2(float lo, float hi) { if constexpr (F16) return pack_h2(lo, hi); else return pack_bf2(lo, hi); }
template <bool F16> __device__ __forceinline__ float el2f(short x) {
    if constexpr (F16) { f16_t h; h.v = (uint16_t)x; return h2f(h); } else return bf2f((bf16_t)x);
}
__device__ __forceinline__ float max3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// value of the partner lane (l ^ 32) combined with the own one: v_permlane32_swap exchanges the upper 32 lanes of its first
// operand with the lower 32 of its second, so with both operands = v every lane ends up holding {own, partner} in {a, b}
// (lower half) or {partner, own} (upper half)
__device__ __forceinline__ float pair_max(float v) {
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return max3(a, a, b);
}
__device__ __forceinline__ float pair_sum(float v) {
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}
__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_dst /* wave-uniform LDS byte address */) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() {
    static_assert(N == 0 || N == 2 || N == 4 || N == 8, "counted waits are spelled out");
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
}
// Pin a value loaded by an ordinary global load BEFORE the streaming loop: hipcc places the vmcnt wait of a load at its first use,
// and a first use inside the loop is a `s_waitcnt vmcnt(0)` executed every iteration — it drains the LDS-DMA queue (the tile
// issued a moment earlier) and serialises the whole pipeline on L2/HBM latency (measured: the loop ran at 1/4 of its speed).
__device__ __forceinline__ void pin(short8& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void pin(float& v) { asm volatile("" : "+v"(v)); }
template <typename V> __device__ __forceinline__ V ldg1(const void* p) {
    typedef const __attribute__((address_space(1))) V* gptr;
    return *(gptr)(p);
}
__device__ __forceinline__ uint2 lds_tr16(const unsigned char* p) {
    typedef __attribute__((address_space(3))) short4_w32 lds_v4;
    return __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(unsigned)(size_t)p));
}
template <bool F16> __device__ __forceinline__ short8 pack8(float a0, float a1, float a2, float a3, float a4, float a5, float a6, float a7) {
    return __builtin_bit_cast(short8, make_uint4(pk2<F16>(a0, a1), pk2<F16>(a2, a3), pk2<F16>(a4, a5), pk2<F16>(a6, a7)));
}

// One streamed operand tile: 64 rows x HD bf16, row-major, rows of ROWB bytes = CPR 16-byte chunks, chunk c of row r stored at
// chunk position c ^ g(r).
//   ds_read_b128 row fragments (32 consecutive rows, one logical chunk): the hardware serves a wave in 16-lane groups
//   {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31} (+32); g must send the 16 rows of a group to 16 distinct 16-byte slots of the
//   256-byte bank row.  HD = 64 (128-byte rows, two rows per bank row): g = bits 1..3 of r, permuted; HD = 128: g = bits 0..3.
//   ds_read_b64_tr_b16 (32 lanes: 4 consecutive rows x 64 contiguous bytes): the 4 rows must land in 4 different 64-byte
//   quarters of the bank row: HD = 64: row bit 0 picks the half of the bank row, row bit 1 must flip chunk bit 2;
//   HD = 128: row bits 0..1 must drive chunk bits 2..3.  Both g below satisfy both.
template <int HD, int NW>
struct WT {
    static constexpr int ROWB = HD * 2, CPR = HD / 8, TILE = 64 * ROWB;
    static constexpr int NPC = TILE / 1024 / NW;                       // DMA wave-instructions per wave and operand tile
    static_assert(NPC * NW * 1024 == TILE, "tile must split into 1-KiB pieces over the waves");
    static __device__ __forceinline__ int g(int r) {
        return HD == 64 ? ((((r >> 1) & 1) << 2) | ((r >> 2) & 3)) : (((r & 3) << 2) | ((r >> 2) & 3));
    }
    static __device__ __forceinline__ int off(int r, int c) { return r * ROWB + ((c ^ g(r)) << 4); }
    // global source of this lane's 16 bytes of DMA piece j of wave `wid` (row0 = first row of the tile NOT included)
    static __device__ __forceinline__ const bf16_t* src(const bf16_t* base, int64_t rs, int wid, int j, int lane) {
        const int P = (wid * NPC + j) * 64 + lane;
        const int row = P / CPR, cph = P % CPR;
        return base + (int64_t)row * rs + ((cph ^ g(row)) << 3);
    }
    // A operand, rows = tile rows: lane (l & 31) = row, k = 8 consecutive head-dim elements of 16-byte chunk `c`
    static __device__ __forceinline__ short8 fragA(const unsigned char* tile, int row, int c) {
        return *reinterpret_cast<const short8*>(tile + off(row, c));
    }
    // A operand of the TRANSPOSED tile: m = head-dim index db*32 + (l & 31), k-slot (hi = l >> 5, j = 0..7) = tile row
    // rbase + 4*hi + (j & 3) + 8*(j >> 2) — the accumulator's row order, see the file header.
    static __device__ __forceinline__ short8 fragT(const unsigned char* tile, int rbase, int db, int lane) {
        const int i = lane & 15, gi = lane >> 4;
        const int row = rbase + 4 * (gi >> 1) + (i >> 2);
        const int c = db * 4 + 2 * (gi & 1) + ((i >> 1) & 1);
        const int byte = (i & 1) * 8;
        const uint2 lo = lds_tr16(tile + off(row, c) + byte);
        const uint2 hi = lds_tr16(tile + off(row + 8, c) + byte);
        return __builtin_bit_cast(short8, make_uint4(lo.x, lo.y, hi.x, hi.y));
    }
};

// row block of wave `wid`: the two waves of a SIMD (w, w + 4) take row blocks (i, 7 - i)
template <int NW> __device__ __forceinline__ int row_block(int wid) { return NW == 8 ? (wid < 4 ? wid : 11 - wid) : wid; }

// X^T accumulators (acc[db][r] = X[own row = lane & 31][d = db*32 + (r&3) + 8*(r>>2) + 4*(lane>>5)]) -> bf16 rows in HBM, through a
// wave-private LDS patch so that every store instruction writes whole contiguous row pieces (16 bytes per lane).
template <int HD, bool F16>
__device__ __forceinline__ void store_tile32(const f32x16 (&acc)[HD / 32], float mul, bf16_t* g0, int64_t rs, unsigned char* scr, int lane) {
    constexpr int PITCH = HD * 2 + 16, CPR = HD / 8;
    const int l32 = lane & 31, hi = lane >> 5;
#pragma unroll
    for (int db = 0; db < HD / 32; ++db)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint2 pk = make_uint2(pk2<F16>(acc[db][4 * j] * mul, acc[db][4 * j + 1] * mul), pk2<F16>(acc[db][4 * j + 2] * mul, acc[db][4 * j + 3] * mul));
            *reinterpret_cast<uint2*>(scr + l32 * PITCH + (db * 32 + 8 * j + 4 * hi) * 2) = pk;
        }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int it = 0; it < 32 * CPR / 64; ++it) {
        const int slot = it * 64 + lane, row = slot / CPR, ch = slot % CPR;
        const uint4 v = *reinterpret_cast<const uint4*>(scr + row * PITCH + ch * 16);
        st_wt16(g0 + (int64_t)row * rs + ch * 8, v);                        // (write-through: common.h)
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// ------------------------------------------------------------------------------------------------ forward
// x[i] = (cc_i > thr && x[i] > finfo.min) ? fill : x[i] for four elements — the fill that REPLACES a future score by a value padding keys do
// not take (GPT-2's -1e4: padding keys keep finfo.min).  Compares go into SGPR pairs (hipcc's own v_cmp (VCC) / v_cndmask pairs need a wait
// state between a compare and ITS select and are emitted back to back).  A padding key's raw score IS
// finfo.min (its bias absorbs the dot product exactly), so x[i] = (cc_i > thr && x[i] > finfo.min) ? fill : x[i] needs no second look
// at the key-bias row — which hipcc would otherwise keep live across the score MFMAs (32 registers and 16 copies per tile).
template <int R0, int C0>
__device__ __forceinline__ void fill4_future_keepmin(f32x16& v, int thr, float fill) {
    constexpr int C1 = C0 + 1, C2 = C0 + 2, C3 = C0 + 3;
    static_assert(C0 >= 0 && C3 <= 64, "inline integer constants");
    float a = v[R0], b = v[R0 + 1], c = v[R0 + 2], d = v[R0 + 3];
    const float fmin = FINFO_MIN;
    uint64_t m0, m1, m2, m3, n0, n1, n2, n3;
    asm volatile("v_cmp_lt_i32_e64 %4, %12, %15\n\t"
                 "v_cmp_lt_i32_e64 %5, %12, %16\n\t"
                 "v_cmp_lt_i32_e64 %6, %12, %17\n\t"
                 "v_cmp_lt_i32_e64 %7, %12, %18\n\t"
                 "v_cmp_lt_f32_e64 %8, %14, %0\n\t"
                 "v_cmp_lt_f32_e64 %9, %14, %1\n\t"
                 "v_cmp_lt_f32_e64 %10, %14, %2\n\t"
                 "v_cmp_lt_f32_e64 %11, %14, %3\n\t"
                 "s_and_b64 %4, %4, %8\n\t"
                 "s_and_b64 %5, %5, %9\n\t"
                 "s_and_b64 %6, %6, %10\n\t"
                 "s_and_b64 %7, %7, %11\n\t"
                 "v_cndmask_b32_e64 %0, %0, %13, %4\n\t"
                 "v_cndmask_b32_e64 %1, %1, %13, %5\n\t"
                 "v_cndmask_b32_e64 %2, %2, %13, %6\n\t"
                 "v_cndmask_b32_e64 %3, %3, %13, %7"
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3), "=&s"(n0), "=&s"(n1), "=&s"(n2), "=&s"(n3)
                 : "v"(thr), "v"(fill), "v"(fmin), "n"(C0), "n"(C1), "n"(C2), "n"(C3) : "scc");
    v[R0] = a; v[R0 + 1] = b; v[R0 + 2] = c; v[R0 + 3] = d;
}

// The hazard recognizer of hipcc does not look into inline asm: an asm VALU instruction (max3, the fill helpers) that reads an MFMA
// result gets none of the wait states a compiler-emitted reader would.  With the max tree directly behind the last score MFMA the
// row maximum was taken over the partial sums of the earlier K-slices (caught by the statistics check of tools/attn_w32_check.py).
// 19 wait states cover an MFMA of this shape; the fences keep the scheduler from moving asm readers above it.  Needed wherever
// the FIRST reader of an accumulator is an asm statement.
__device__ __forceinline__ void mfma_results_fence() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 2" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// Schedule.  Measured on this chip (profiles/r03_valu_issue_probe.txt, tools/attn_w32_timing.py): ONE wave issues at most one VALU
// instruction per ~5 cycles while a SIMD retires one per ~2.35 when two or more of its waves are in vector code; packed fp32
// (v_pk_*_f32) runs at 12.9 cycles per instruction beside a wave that issues MFMAs (scalar fp32 VALU is untouched by it); a
// matrix/vector ping-pong of the two waves of a SIMD therefore runs its vector segment at a third of the SIMD's VALU rate (built
// and measured: slower than the plain loop).  So: a plain loop, one barrier per tile, and THREE workgroups of four waves per CU
// (head_dim 64: 150 - 164 registers) that no barrier couples — so that while one wave waits for its MFMAs the others issue VALU;
// every fp32 operation scalar (this file is built with -fno-slp-vectorize: hipcc would pack adjacent adds and multiplies); the
// instruction count per score cut to the bone:
//   * the per-key bias enters as the C operand of the first score MFMA (no bias add) — and so does the finfo.min of the causal
//     future on the diagonal tile (no select after the MFMAs: at the merge of a masked and an unmasked path hipcc copies the whole
//     accumulator out, one v_mov per score on every tile),
//   * scores stay in units of 1/scale ("raw") and p = exp2(raw*c - max*c), c = scale*log2(e): ONE fma + ONE exp2 per score.
template <int HD, int NW, bool REPLACE, bool F16 = false>       // REPLACE: future scores are REPLACED by a value other than finfo.min (GPT-2's -1e4); F16: IEEE-half operands
__global__ __launch_bounds__(NW * 64, NW == 4 ? (HD == 64 ? 3 : 2) : NW / 4) void attn32_fwd_kernel(AttnP p) {
    using W = WT<HD, NW>;
    constexpr int NDS = HD / 16, NDB = HD / 32, TILE = W::TILE, STAGE = 2 * TILE, NPC = W::NPC, RPB = 32 * NW, NST = CTMI_W32_FWD_NST;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* kbS = reinterpret_cast<float*>(smem + NST * STAGE);              // per-key bias / scale  (finfo.min: padding key)
    const int tid = threadIdx.x, lane = tid & 63, l32 = lane & 31, hi = lane >> 5;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rbw = row_block<NW>(wid);
    const int BH = (int)(p.B * p.nh), nqb = (int)((p.Sq + RPB - 1) / RPB);
    const int vid = blockIdx.x;
    const int qb = nqb - 1 - vid / BH, bh = vid % BH;                        // longest query blocks first (LPT)
    const int64_t h = bh % p.nh, b = bh / p.nh;
    const int q0 = qb * RPB, q0w = q0 + 32 * rbw;
    const bool active = q0w < (int)p.Sq;
    // a query row whose whole causal window is padding sees only masked keys -> uniform over ALL keys
    const bool allk = p.kvalid != nullptr && p.first_valid[b] > q0;
    const int kv_end = allk ? (int)p.Sk : min((int)p.Sk, q0 + RPB);
    const int ntiles = kv_end / 64;
    const int my_last = !active ? -1 : (allk ? ntiles - 1 : min(ntiles - 1, (q0w + 31) / 64));
    const bf16_t* qp = reinterpret_cast<const bf16_t*>(p.q) + b * p.q_bs + h * p.q_hs;
    const bf16_t* kp = reinterpret_cast<const bf16_t*>(p.k) + b * p.k_bs + h * p.k_hs;
    const bf16_t* vp = reinterpret_cast<const bf16_t*>(p.v) + b * p.v_bs + h * p.v_hs;

    const bf16_t* pk[NPC];
    const bf16_t* pv[NPC];
#pragma unroll
    for (int j = 0; j < NPC; ++j) { pk[j] = W::src(kp, p.k_rs, wid, j, lane); pv[j] = W::src(vp, p.v_rs, wid, j, lane); }
    const unsigned lds0 = (unsigned)(size_t)smem;
    const int64_t kstep = 64 * p.k_rs, vstep = 64 * p.v_rs;
    auto issue = [&](int stage) {
        const unsigned d = lds0 + stage * STAGE + wid * NPC * 1024;
#pragma unroll
        for (int j = 0; j < NPC; ++j) { dma16(pk[j], d + j * 1024); pk[j] += kstep; }
#pragma unroll
        for (int j = 0; j < NPC; ++j) { dma16(pv[j], d + TILE + j * 1024); pv[j] += vstep; }
    };
    issue(0);
    if (NST >= 3 && ntiles > 1) issue(1);

    const float slope_r = p.slopes ? p.slopes[h] / p.scale : 0.f;
    for (int key = tid; key < kv_end; key += NW * 64) {
        const int valid = p.kvalid != nullptr ? (int)p.kvalid[b * p.Sk + key] : 1;
        const float pos = p.kpos != nullptr ? p.kpos[b * p.Sk + key] : 0.f;
        kbS[key] = valid != 0 ? slope_r * pos : FINFO_MIN;
    }
    short8 qf[NDS];
#pragma unroll
    for (int ds = 0; ds < NDS; ++ds) {
        qf[ds] = short8{0, 0, 0, 0, 0, 0, 0, 0};
        if (active) qf[ds] = ldg1<short8>(qp + (int64_t)(q0w + l32) * p.q_rs + ds * 16 + hi * 8);
    }
    __syncthreads();
#pragma unroll
    for (int ds = 0; ds < NDS; ++ds) pin(qf[ds]);

    f32x16 o[NDB];
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
    float m = -INFINITY, lsum = 0.f;                                         // m in raw units
    const float c = p.scale * LOG2E_F;
    const float ffr = p.future_fill <= FINFO_MIN ? FINFO_MIN : p.future_fill / p.scale;
#if CTMI_W32_TIMING
    uint32_t tX = 0, tY = 0, tBX = 0, tBY = 0, tPro = 0;
    uint64_t tlast = __builtin_amdgcn_s_memtime();
    const uint64_t tbeg = tlast;
#endif
    int st = 0;
    for (int t = 0; t < ntiles; ++t) {
        // ring of NST stages, NST - 1 tiles in flight: tile t + NST - 1 goes out right after the barrier that retires tile t - 1
        if (NST >= 3 && t + 1 < ntiles) wait_vm<2 * NPC>(); else wait_vm<0>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        W32_TICK(tY);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        W32_TICK(tBX);
        if (t + NST - 1 < ntiles) issue(st == 0 ? NST - 1 : st - 1);
        if (t <= my_last) {
            const unsigned char* ks = smem + st * STAGE;
            const unsigned char* vs = ks + TILE;
            const int kv0 = t * 64;
            f32x16 x[2];
            const bool diag = kv0 + 63 > q0w;                              // the tile holds (query, key) pairs in the causal future
            const int thr = q0w + l32 - kv0 - 4 * hi;                       // key offset cc = kk*32 + 8*(r>>2) + (r&3) is in the future iff cc > thr
            // masked_fill(finfo.min) of the future keys rides in the C operand too: finfo.min + q.k = finfo.min.  (A select AFTER the
            // MFMAs on the diagonal tiles only makes hipcc copy all 32 scores of EVERY tile out of the accumulator tuples where the two
            // paths merge — one v_mov per score on a loop bound by vector issue.  Before the MFMAs both paths define the tuple afresh.)
            if (!REPLACE && diag) {
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const f32x4 kb4 = *reinterpret_cast<const f32x4*>(kbS + kv0 + kk * 32 + 8 * j + 4 * hi);
#pragma unroll
                        for (int e = 0; e < 4; ++e) x[kk][4 * j + e] = (kk * 32 + 8 * j + e > thr) ? FINFO_MIN : kb4[e];
                    }
            } else {
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const f32x4 kb4 = *reinterpret_cast<const f32x4*>(kbS + kv0 + kk * 32 + 8 * j + 4 * hi);
                        x[kk][4 * j] = kb4[0]; x[kk][4 * j + 1] = kb4[1]; x[kk][4 * j + 2] = kb4[2]; x[kk][4 * j + 3] = kb4[3];
                    }
            }
#pragma unroll
            for (int ds = 0; ds < NDS; ++ds) {
                x[0] = mfma32<F16>(W::fragA(ks, l32, ds * 2 + hi), qf[ds], x[0]);   // raw score = q.k + bias/scale (padding: finfo.min absorbs the dot)
                x[1] = mfma32<F16>(W::fragA(ks, 32 + l32, ds * 2 + hi), qf[ds], x[1]);
            }
            mfma_results_fence();
            if (REPLACE && diag) {                                           // GPT-2's -1e4 replacement: padding keys keep finfo.min
#define W32_FILL(kk, j) fill4_future_keepmin<4 * j, kk * 32 + 8 * j>(x[kk], thr, ffr)
                W32_FILL(0, 0); W32_FILL(0, 1); W32_FILL(0, 2); W32_FILL(0, 3); W32_FILL(1, 0); W32_FILL(1, 1); W32_FILL(1, 2); W32_FILL(1, 3);
#undef W32_FILL
            }
            float mx0 = max3(x[0][0], x[0][1], x[0][2]), mx1 = max3(x[0][8], x[0][9], x[0][10]);
            float mx2 = max3(x[1][0], x[1][1], x[1][2]), mx3 = max3(x[1][8], x[1][9], x[1][10]);
            mx0 = max3(mx0, x[0][3], x[0][4]); mx1 = max3(mx1, x[0][11], x[0][12]); mx2 = max3(mx2, x[1][3], x[1][4]); mx3 = max3(mx3, x[1][11], x[1][12]);
            mx0 = max3(mx0, x[0][5], x[0][6]); mx1 = max3(mx1, x[0][13], x[0][14]); mx2 = max3(mx2, x[1][5], x[1][6]); mx3 = max3(mx3, x[1][13], x[1][14]);
            mx0 = max3(mx0, x[0][7], x[0][15]); mx2 = max3(mx2, x[1][7], x[1][15]);
            float mx = max3(mx0, mx1, mx2);
            mx = max3(mx, mx, mx3);
            mx = pair_max(mx);
            float m_new = max3(m, m, mx);                                    // finite: every tile holds >= 1 existing key
            if (CTMI_W32_DEFER_MAX > 0) {
                // keep the old reference while no row maximum of the wave grew by more than 2^DEFER: the probabilities of this tile are
                // then bounded by 2^DEFER instead of 1 (bf16 is a floating format: no precision is lost), O and l stay in the old scale
                if (!__any((mx - m) * c > (float)CTMI_W32_DEFER_MAX)) m_new = m;
            }
            const float alpha = __builtin_amdgcn_exp2f((m - m_new) * c);
            const float mc = m_new * c;
            if (allk) {
                // a row may be all-masked here: its scores AND its maximum are finfo.min and the exponent must be exactly 0 — the two
                // products are formed separately (identical roundings cancel); the fused form below would leave the rounding error
                // of finfo.min * c, which is astronomically large
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float pr;
                        asm("v_mul_f32 %0, %1, %2\n\tv_sub_f32 %0, %0, %3" : "=&v"(pr) : "v"(x[kk][r]), "v"(c), "v"(mc));
                        x[kk][r] = __builtin_amdgcn_exp2f(pr);
                    }
            } else {
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int r = 0; r < 16; ++r) x[kk][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(x[kk][r], c, -mc));
            }
            // row sums as plain fp32 adds in four chains (v_dot2c_f32_bf16 against (1, 1) on the packed values was built and measured:
            // 16 instead of 32 instructions, but 41.7 vs 38.8 us — the dot instruction costs several issue slots)
            float rs[4] = {x[0][0], x[0][1], x[1][0], x[1][1]};
#pragma unroll
            for (int r = 2; r < 16; r += 2) { rs[0] += x[0][r]; rs[1] += x[0][r + 1]; rs[2] += x[1][r]; rs[3] += x[1][r + 1]; }
            short8 pb[2][2];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int s = 0; s < 2; ++s)
                    pb[kk][s] = pack8<F16>(x[kk][8 * s], x[kk][8 * s + 1], x[kk][8 * s + 2], x[kk][8 * s + 3],
                                      x[kk][8 * s + 4], x[kk][8 * s + 5], x[kk][8 * s + 6], x[kk][8 * s + 7]);
            lsum = __builtin_fmaf(lsum, alpha, (rs[0] + rs[1]) + (rs[2] + rs[3]));
            if (__any(m_new > m)) {                                          // wave-uniform: rescale only when some row max moved
#pragma unroll
                for (int db = 0; db < NDB; ++db)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
            }
            m = m_new;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int db = 0; db < NDB; ++db) o[db] = mfma32<F16>(W::fragT(vs, kk * 32 + 16 * s, db, lane), pb[kk][s], o[db]);
        }
        W32_TICK(tX);
        st = st == NST - 1 ? 0 : st + 1;
    }
    lsum = pair_sum(lsum);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                           // every wave is done with the ring: reuse it as store patches
    if (active) {
        bf16_t* op = reinterpret_cast<bf16_t*>(p.out) + b * p.o_bs + h * p.o_hs + (int64_t)q0w * p.o_rs;
        store_tile32<HD, F16>(o, 1.0f / lsum, op, p.o_rs, smem + wid * 32 * (HD * 2 + 16), lane);
        if (hi == 0) {
            const int64_t srow = (b * p.nh + h) * p.Sq + q0w + l32;
            p.stat_m[srow] = m <= FINFO_MIN ? FINFO_MIN : m * p.scale;
            p.stat_l[srow] = lsum;
#if CTMI_W32_TIMING
            const uint32_t tot = (uint32_t)(__builtin_amdgcn_s_memtime() - tbeg);
            const uint32_t v[8] = {tX, tY, tBX, tBY, tPro, tot, (uint32_t)ntiles, (uint32_t)(my_last + 1)};
            if (l32 < 8) p.stat_l[srow] = (float)v[l32];
#endif
        }
    }
}

// ------------------------------------------------------------------------------------------------ backward: dQ
// own rows = queries.  Per key tile: S^T = K Q^T, dP^T = V dO^T (both with the own query in the accumulator column),
// P = exp2(s2 - m2) / l, dS^T = P (dP^T - delta), masked entries dS = 0;  dQ^T += K^T dS^T.  Also forms delta = rowsum(dO * O).
template <int HD, int NW, bool F16 = false>
__global__ __launch_bounds__(NW * 64, NW == 4 ? (HD == 64 ? 3 : 2) : NW / 4) void attn32_dq_kernel(AttnP p) {
    using W = WT<HD, NW>;
    constexpr int NDS = HD / 16, NDB = HD / 32, TILE = W::TILE, STAGE = 2 * TILE, NPC = W::NPC, RPB = 32 * NW, NST = CTMI_W32_BWD_NST;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* kbS = reinterpret_cast<float*>(smem + NST * STAGE);
    const int tid = threadIdx.x, lane = tid & 63, l32 = lane & 31, hi = lane >> 5;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rbw = row_block<NW>(wid);
    const int BH = (int)(p.B * p.nh), nqb = (int)((p.Sq + RPB - 1) / RPB);
    const int vid = blockIdx.x;
    const int qb = nqb - 1 - vid / BH, bh = vid % BH;
    const int64_t h = bh % p.nh, b = bh / p.nh;
    const int q0 = qb * RPB, q0w = q0 + 32 * rbw;
    const bool active = q0w < (int)p.Sq;
    const int kv_end = min((int)p.Sk, q0 + RPB);                             // masked entries have dS = 0: nothing beyond the diagonal
    const int ntiles = kv_end / 64;
    const int my_last = min(ntiles - 1, (q0w + 31) / 64);
    const bf16_t* qp = reinterpret_cast<const bf16_t*>(p.q) + b * p.q_bs + h * p.q_hs;
    const bf16_t* kp = reinterpret_cast<const bf16_t*>(p.k) + b * p.k_bs + h * p.k_hs;
    const bf16_t* vp = reinterpret_cast<const bf16_t*>(p.v) + b * p.v_bs + h * p.v_hs;
    const bf16_t* gp = reinterpret_cast<const bf16_t*>(p.d_o) + b * p.o_bs + h * p.o_hs;
    const bf16_t* op = reinterpret_cast<const bf16_t*>(p.o) + b * p.o_bs + h * p.o_hs;

    const bf16_t* pk[NPC];
    const bf16_t* pv[NPC];
#pragma unroll
    for (int j = 0; j < NPC; ++j) { pk[j] = W::src(kp, p.k_rs, wid, j, lane); pv[j] = W::src(vp, p.v_rs, wid, j, lane); }
    const unsigned lds0 = (unsigned)(size_t)smem;
    const int64_t kstep = 64 * p.k_rs, vstep = 64 * p.v_rs;
    auto issue = [&](int stage) {
        const unsigned d = lds0 + stage * STAGE + wid * NPC * 1024;
#pragma unroll
        for (int j = 0; j < NPC; ++j) { dma16(pk[j], d + j * 1024); pk[j] += kstep; }
#pragma unroll
        for (int j = 0; j < NPC; ++j) { dma16(pv[j], d + TILE + j * 1024); pv[j] += vstep; }
    };
    issue(0);
    if (NST >= 3 && ntiles > 1) issue(1);

    // per-key bias in units of 1/scale ("raw", as in the forward): it enters as the C operand of the first score MFMA, a padding key's
    // finfo.min absorbs the dot product exactly
    const float slope_r = p.slopes ? p.slopes[h] / p.scale : 0.f;
    for (int key = tid; key < kv_end; key += NW * 64) {
        const int valid = p.kvalid != nullptr ? (int)p.kvalid[b * p.Sk + key] : 1;
        const float pos = p.kpos != nullptr ? p.kpos[b * p.Sk + key] : 0.f;
        kbS[key] = valid != 0 ? slope_r * pos : FINFO_MIN;
    }
    const float c = p.scale * LOG2E_F;
    short8 qf[NDS], gf[NDS];
    float dl = 0.f, nm2l = 0.f;
    const int64_t srow = (b * p.nh + h) * p.Sq + q0w + l32;
    if (active) {
        const int64_t ro = (int64_t)(q0w + l32) * p.o_rs;
#pragma unroll
        for (int ds = 0; ds < NDS; ++ds) {
            qf[ds] = ldg1<short8>(qp + (int64_t)(q0w + l32) * p.q_rs + ds * 16 + hi * 8);
            gf[ds] = ldg1<short8>(gp + ro + ds * 16 + hi * 8);
            const short8 of = ldg1<short8>(op + ro + ds * 16 + hi * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) dl += el2f<F16>(gf[ds][j]) * el2f<F16>(of[j]);
        }
        // P = exp2(raw * c - m2 - log2 l): 1/l folded into the exponent.  An all-masked row (finfo.min statistic) has only finfo.min
        // scores in this kernel, P = 0 whatever the finite offset.
        const float mm = p.stat_m[srow];
        nm2l = mm <= FINFO_MIN ? 0.f : -(mm * LOG2E_F + __builtin_amdgcn_logf(p.stat_l[srow]));
    } else {
#pragma unroll
        for (int ds = 0; ds < NDS; ++ds) { qf[ds] = short8{0, 0, 0, 0, 0, 0, 0, 0}; gf[ds] = qf[ds]; }
    }
    dl = pair_sum(dl);
    if (active && hi == 0) p.delta[srow] = dl;
    __syncthreads();
#pragma unroll
    for (int ds = 0; ds < NDS; ++ds) { pin(qf[ds]); pin(gf[ds]); }
    pin(dl); pin(nm2l);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                        // the delta store as well: nothing of the prologue may be pending in the loop

    f32x16 dq[NDB];
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[db][r] = 0.f;
    int st = 0;
    for (int t = 0; t < ntiles; ++t) {
        if (NST >= 3 && t + 1 < ntiles) wait_vm<2 * NPC>(); else wait_vm<0>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (t + NST - 1 < ntiles) issue(st == 0 ? NST - 1 : st - 1);
        if (active && t <= my_last) {
            const unsigned char* ks = smem + st * STAGE;
            const unsigned char* vs = ks + TILE;
            const int kv0 = t * 64;
            // NK halves of the 64-key tile are in flight together: both at head_dim 64 (four independent MFMA chains); ONE at head_dim 128
            // (round 5: 32 accumulator registers less — with the 64 of dQ^T and the 64 of the own-row fragments that is what keeps two waves per SIMD)
            constexpr int NK = HD == 128 ? 1 : 2;
            const int thr = q0w + l32 - kv0 - 4 * hi;                       // key offset cc usable iff cc <= thr (not in the causal future)
#pragma unroll
            for (int k0h = 0; k0h < 2; k0h += NK) {
            f32x16 x[NK], y[NK];
#pragma unroll
            for (int kk = 0; kk < NK; ++kk)
#pragma unroll
                for (int r = 0; r < 16; ++r) y[kk][r] = 0.f;
            // A masked pair must give dS = 0.  A padding key does by itself (bias finfo.min -> P = exp2(-huge) = 0); the causal future of
            // the diagonal tile gets finfo.min through the same C operand (a select after the MFMAs costs a register copy per score on
            // every tile, see the forward).  An all-masked query row (LEFT padding) then has P = 0 everywhere: its exponent offset is 0.
            if (kv0 + 63 > q0w) {
#pragma unroll
                for (int kk = 0; kk < NK; ++kk)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const f32x4 kb4 = *reinterpret_cast<const f32x4*>(kbS + kv0 + (k0h + kk) * 32 + 8 * j + 4 * hi);
#pragma unroll
                        for (int e = 0; e < 4; ++e) x[kk][4 * j + e] = ((k0h + kk) * 32 + 8 * j + e > thr) ? FINFO_MIN : kb4[e];
                    }
            } else {
#pragma unroll
                for (int kk = 0; kk < NK; ++kk)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const f32x4 kb4 = *reinterpret_cast<const f32x4*>(kbS + kv0 + (k0h + kk) * 32 + 8 * j + 4 * hi);
                        x[kk][4 * j] = kb4[0]; x[kk][4 * j + 1] = kb4[1]; x[kk][4 * j + 2] = kb4[2]; x[kk][4 * j + 3] = kb4[3];
                    }
            }
#pragma unroll
            for (int ds = 0; ds < NDS; ++ds) {
#pragma unroll
                for (int kk = 0; kk < NK; ++kk) x[kk] = mfma32<F16>(W::fragA(ks, (k0h + kk) * 32 + l32, ds * 2 + hi), qf[ds], x[kk]);
#pragma unroll
                for (int kk = 0; kk < NK; ++kk) y[kk] = mfma32<F16>(W::fragA(vs, (k0h + kk) * 32 + l32, ds * 2 + hi), gf[ds], y[kk]);
            }
            // per score: one fma, one exp2, one sub, one mul
#pragma unroll
            for (int kk = 0; kk < NK; ++kk)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float pr = __builtin_amdgcn_exp2f(__builtin_fmaf(x[kk][r], c, nm2l));
                    y[kk][r] = pr * (y[kk][r] - dl);
                }
#pragma unroll
            for (int kk = 0; kk < NK; ++kk)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const short8 db8 = pack8<F16>(y[kk][8 * s], y[kk][8 * s + 1], y[kk][8 * s + 2], y[kk][8 * s + 3],
                                             y[kk][8 * s + 4], y[kk][8 * s + 5], y[kk][8 * s + 6], y[kk][8 * s + 7]);
#pragma unroll
                    for (int db = 0; db < NDB; ++db) dq[db] = mfma32<F16>(W::fragT(ks, (k0h + kk) * 32 + 16 * s, db, lane), db8, dq[db]);
                }
            }
        }
        st = st == NST - 1 ? 0 : st + 1;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (active) {
        bf16_t* dqp = reinterpret_cast<bf16_t*>(p.dq) + b * p.q_bs + h * p.q_hs + (int64_t)q0w * p.q_rs;
        store_tile32<HD, F16>(dq, p.scale, dqp, p.q_rs, smem + wid * 32 * (HD * 2 + 16), lane);
    }
}

// ------------------------------------------------------------------------------------------------ backward: dK, dV
// own rows = keys.  Per query tile: S = Q K^T and dP = dO V^T (own key in the accumulator column), P = exp2(s2 - m2[q]) / l[q],
// dS = P (dP - delta[q]); dV^T += dO^T P, dK^T += Q^T dS.  Masked entries (padding key, causal future): P keeps the fill value's
// probability (non-zero only in all-masked rows, which are uniform), dS = 0.
// MODE (round 5, head_dim 128): 0 = dK and dV in one pass (head_dim 64: 254 registers, two waves per SIMD); 1 = dV only, 2 = dK only — at head_dim 128
// the two 32-row accumulators (128 registers) plus the K and V fragments (64) left ONE wave per SIMD and the general 16-row kernels were faster
// (797 vs 915 us at B=4 S=2048 nh=32); split, the dV pass needs no V fragments, no dP and no delta, the dK pass no dV accumulator, both fit two
// waves per SIMD in registers — and in LDS, which was the second limit: see lds_qg() — and the scores are recomputed once more (the kernel is
// bound by vector issue and occupancy, not by its matrix work): 653 us.
template <int HD, int NW, bool F16 = false, int MODE = 0>
__global__ __launch_bounds__(NW * 64, (NW == 4 && (HD == 64 || MODE != 0)) ? 2 : NW / 4) void attn32_dkdv_kernel(AttnP p) {
    constexpr bool DO_DV = MODE != 2, DO_DK = MODE != 1;
    using W = WT<HD, NW>;
    constexpr int NDS = HD / 16, NDB = HD / 32, TILE = W::TILE, STAGE = 2 * TILE, NPC = W::NPC, RPB = 32 * NW, NST = CTMI_W32_BWD_NST;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // per-query statistics, see lds_qg(): [Sq] score offset | [Sq] delta (not in the dV pass, which has no dS)
    float* m2S = reinterpret_cast<float*>(smem + NST * STAGE);
    float* dlS = m2S + p.Sq;
    const int tid = threadIdx.x, lane = tid & 63, l32 = lane & 31, hi = lane >> 5;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rbw = row_block<NW>(wid);
    const int BH = (int)(p.B * p.nh);
    const int vid = blockIdx.x;
    const int kblk = vid / BH, bh = vid % BH;                                // early key blocks (most query tiles) first
    const int64_t h = bh % p.nh, b = bh / p.nh;
    const float* slG = p.stat_l + (b * p.nh + h) * p.Sq;                     // exact path (left padding): l is read from L2, not staged
    const int k0 = kblk * RPB, k0w = k0 + 32 * rbw;
    const bool active = k0w < (int)p.Sk;
    // all-masked query rows (LEFT padding) are uniform over ALL keys -> they reach every key block
    const bool allq = p.kvalid != nullptr && p.first_valid[b] > 0;
    const int qt_end = (int)(p.Sq / 64);
    const int qt_begin = allq ? 0 : min(k0 / 64, qt_end);
    const int my_first = allq ? 0 : k0w / 64;
    const int nt = qt_end - qt_begin;
    const bf16_t* qp = reinterpret_cast<const bf16_t*>(p.q) + b * p.q_bs + h * p.q_hs + (int64_t)qt_begin * 64 * p.q_rs;
    const bf16_t* kp = reinterpret_cast<const bf16_t*>(p.k) + b * p.k_bs + h * p.k_hs;
    const bf16_t* vp = reinterpret_cast<const bf16_t*>(p.v) + b * p.v_bs + h * p.v_hs;
    const bf16_t* gp = reinterpret_cast<const bf16_t*>(p.d_o) + b * p.o_bs + h * p.o_hs + (int64_t)qt_begin * 64 * p.o_rs;

    const bf16_t* pq[NPC];
    const bf16_t* pg[NPC];
#pragma unroll
    for (int j = 0; j < NPC; ++j) { pq[j] = W::src(qp, p.q_rs, wid, j, lane); pg[j] = W::src(gp, p.o_rs, wid, j, lane); }
    const unsigned lds0 = (unsigned)(size_t)smem;
    const int64_t qstep = 64 * p.q_rs, gstep = 64 * p.o_rs;
    auto issue = [&](int stage) {
        const unsigned d = lds0 + stage * STAGE + wid * NPC * 1024;
#pragma unroll
        for (int j = 0; j < NPC; ++j) { dma16(pq[j], d + j * 1024); pq[j] += qstep; }
#pragma unroll
        for (int j = 0; j < NPC; ++j) { dma16(pg[j], d + TILE + j * 1024); pg[j] += gstep; }
    };
    if (nt > 0) issue(0);
    if (NST >= 3 && nt > 1) issue(1);

    // row statistics of every query this workgroup will stream, converted once (m in log2 units, 1/l, delta): three floats per row
    // in LDS for the lifetime of the workgroup instead of a per-tile restage
    {
        const float* sm = p.stat_m + (b * p.nh + h) * p.Sq;
        const float* sl = p.stat_l + (b * p.nh + h) * p.Sq;
        const float* sd = p.delta + (b * p.nh + h) * p.Sq;
        const float rc = 1.0f / (p.scale * LOG2E_F);
        for (int q = qt_begin * 64 + tid; q < (int)p.Sq; q += NW * 64) {
            const float mm = sm[q], ll = sl[q];
            if (allq) {                                                      // exact path (all-masked query rows exist in this batch row)
                m2S[q] = mm <= FINFO_MIN ? FINFO_MIN : mm * LOG2E_F;
                if constexpr (DO_DK) dlS[q] = sd[q];
            } else {
                // fast path: both per-query terms enter as the C operands of the tile's first MFMAs — the score accumulator starts at
                // -(m2 + log2 l) / c (raw units), the dP accumulator at -delta — and the loop is P = exp2(fma(raw, c, bias)), dS = P * dPd
                m2S[q] = -(mm * LOG2E_F + __builtin_amdgcn_logf(ll)) * rc;
                if constexpr (DO_DK) dlS[q] = -sd[q];
            }
        }
    }

    short8 kf[NDS], vf[NDS];
    float kb_lane = 0.f;
    bool key_pad = false;
    if (active) {
        const int64_t key = k0w + l32;
#pragma unroll
        for (int ds = 0; ds < NDS; ++ds) {
            kf[ds] = ldg1<short8>(kp + key * p.k_rs + ds * 16 + hi * 8);
            if constexpr (DO_DK) vf[ds] = ldg1<short8>(vp + key * p.v_rs + ds * 16 + hi * 8);
            else vf[ds] = short8{0, 0, 0, 0, 0, 0, 0, 0};
        }
        const float slope2 = p.slopes ? p.slopes[h] * LOG2E_F : 0.f;
        const int valid = p.kvalid != nullptr ? (int)p.kvalid[b * p.Sk + key] : 1;
        const float pos = p.kpos != nullptr ? p.kpos[b * p.Sk + key] : 0.f;
        key_pad = valid == 0;
        kb_lane = key_pad ? FINFO_MIN : slope2 * pos;
    } else {
#pragma unroll
        for (int ds = 0; ds < NDS; ++ds) { kf[ds] = short8{0, 0, 0, 0, 0, 0, 0, 0}; vf[ds] = kf[ds]; }
    }
    const float ff2 = p.future_fill <= FINFO_MIN ? FINFO_MIN : p.future_fill * LOG2E_F;
    float lane_fill = key_pad ? FINFO_MIN : ff2;                            // what a masked score of this lane's key is replaced by
    __syncthreads();
#pragma unroll
    for (int ds = 0; ds < NDS; ++ds) { pin(kf[ds]); if constexpr (DO_DK) pin(vf[ds]); }
    pin(kb_lane); pin(lane_fill);

    f32x16 dk[NDB], dv[NDB];
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[db][r] = 0.f; dv[db][r] = 0.f; }
    const float c = p.scale * LOG2E_F;
    int st = 0;
    for (int i = 0; i < nt; ++i) {
        const int t = qt_begin + i;
        if (NST >= 3 && i + 1 < nt) wait_vm<2 * NPC>(); else wait_vm<0>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (i + NST - 1 < nt) issue(st == 0 ? NST - 1 : st - 1);
        if (active && t >= my_first) {
            const unsigned char* qs = smem + st * STAGE;
            const unsigned char* gs = qs + TILE;
            // NQ halves of the 64-query tile are in flight together: both at head_dim 64; ONE in the dK pass at head_dim 128 (32 accumulator registers less)
            constexpr int NQ = (HD == 128 && MODE == 2) ? 1 : 2;
            // masked(cc) for the query at offset cc = qq*32 + 8*j + e of this lane's group: padding key, or query index < key index
            const int thr = key_pad ? 0x7fffffff : (k0w + l32 - t * 64 - 4 * hi);
            const bool diag = t * 64 < k0w + 31;                            // some (query, key) pair of the tile is in the causal future
#pragma unroll
            for (int q0h = 0; q0h < 2; q0h += NQ) {
            f32x16 x[NQ], y[NQ];
            if (allq) {
#pragma unroll
                for (int qq = 0; qq < NQ; ++qq)
#pragma unroll
                    for (int r = 0; r < 16; ++r) { x[qq][r] = 0.f; y[qq][r] = 0.f; }
            } else {
#pragma unroll
                for (int qq = 0; qq < NQ; ++qq)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int qi = t * 64 + (q0h + qq) * 32 + 8 * j + 4 * hi;
                        const f32x4 a4 = *reinterpret_cast<const f32x4*>(m2S + qi);
                        f32x4 d4 = {0.f, 0.f, 0.f, 0.f};
                        if constexpr (DO_DK) d4 = *reinterpret_cast<const f32x4*>(dlS + qi);
                        if (diag) {                                            // masked pair: score finfo.min -> P = 0, dS = 0
#pragma unroll
                            for (int e = 0; e < 4; ++e) x[qq][4 * j + e] = ((q0h + qq) * 32 + 8 * j + e < thr) ? FINFO_MIN : a4[e];
                        } else {
                            x[qq][4 * j] = a4[0]; x[qq][4 * j + 1] = a4[1]; x[qq][4 * j + 2] = a4[2]; x[qq][4 * j + 3] = a4[3];
                        }
                        y[qq][4 * j] = d4[0]; y[qq][4 * j + 1] = d4[1]; y[qq][4 * j + 2] = d4[2]; y[qq][4 * j + 3] = d4[3];
                    }
            }
#pragma unroll
            for (int ds = 0; ds < NDS; ++ds) {
#pragma unroll
                for (int qq = 0; qq < NQ; ++qq) x[qq] = mfma32<F16>(W::fragA(qs, (q0h + qq) * 32 + l32, ds * 2 + hi), kf[ds], x[qq]);
                if constexpr (DO_DK) {                                          // dP = dO V^T: only dS (hence dK) needs it
#pragma unroll
                    for (int qq = 0; qq < NQ; ++qq) y[qq] = mfma32<F16>(W::fragA(gs, (q0h + qq) * 32 + l32, ds * 2 + hi), vf[ds], y[qq]);
                }
            }
            if (!allq) {
                // every query row has an unmasked key, so a masked pair has P = 0 and dS = 0: a padding key gets there by itself (its
                // bias is finfo.min), the causal future through the C operand above.  Per score: one fma, one exp2, one mul.
#pragma unroll
                for (int qq = 0; qq < NQ; ++qq)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float pr = __builtin_amdgcn_exp2f(__builtin_fmaf(x[qq][r], c, kb_lane));
                        x[qq][r] = pr;
                        y[qq][r] = pr * y[qq][r];
                    }
            } else {
#pragma unroll
            for (int qq = 0; qq < NQ; ++qq)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int qi = t * 64 + (q0h + qq) * 32 + 8 * j + 4 * hi;
                    const f32x4 mm = *reinterpret_cast<const f32x4*>(m2S + qi);
                    const f32x4 l4 = *reinterpret_cast<const f32x4*>(slG + qi);
                    f32x4 dl4 = {0.f, 0.f, 0.f, 0.f};
                    if constexpr (DO_DK) dl4 = *reinterpret_cast<const f32x4*>(dlS + qi);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int r = 4 * j + e;
                        const bool msk = ((q0h + qq) * 32 + 8 * j + e) < thr;
                        const float s2 = msk ? lane_fill : __builtin_fmaf(x[qq][r], c, kb_lane);
                        const float pr = __builtin_amdgcn_exp2f(s2 - mm[e]) * (1.0f / l4[e]);
                        const float d = msk ? 0.f : pr * (y[qq][r] - dl4[e]);
                        x[qq][r] = pr;
                        y[qq][r] = d;
                    }
                }
            }
#pragma unroll
            for (int qq = 0; qq < NQ; ++qq)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const short8 pb = pack8<F16>(x[qq][8 * s], x[qq][8 * s + 1], x[qq][8 * s + 2], x[qq][8 * s + 3],
                                            x[qq][8 * s + 4], x[qq][8 * s + 5], x[qq][8 * s + 6], x[qq][8 * s + 7]);
                    const short8 sb = pack8<F16>(y[qq][8 * s], y[qq][8 * s + 1], y[qq][8 * s + 2], y[qq][8 * s + 3],
                                            y[qq][8 * s + 4], y[qq][8 * s + 5], y[qq][8 * s + 6], y[qq][8 * s + 7]);
#pragma unroll
                    for (int db = 0; db < NDB; ++db) {
                        if constexpr (DO_DV) dv[db] = mfma32<F16>(W::fragT(gs, (q0h + qq) * 32 + 16 * s, db, lane), pb, dv[db]);
                        if constexpr (DO_DK) dk[db] = mfma32<F16>(W::fragT(qs, (q0h + qq) * 32 + 16 * s, db, lane), sb, dk[db]);
                    }
                }
            }
        }
        st = st == NST - 1 ? 0 : st + 1;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (active) {
        if (key_pad) {                                                       // the mask-free tiles leave p = 0 there; a padding key's dK is 0 by definition
#pragma unroll
            for (int db = 0; db < NDB; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) dk[db][r] = 0.f;
        }
        bf16_t* dkp = reinterpret_cast<bf16_t*>(p.dk) + b * p.k_bs + h * p.k_hs + (int64_t)k0w * p.k_rs;
        bf16_t* dvp = reinterpret_cast<bf16_t*>(p.dv) + b * p.v_bs + h * p.v_hs + (int64_t)k0w * p.v_rs;
        unsigned char* scr = smem + wid * 32 * (HD * 2 + 16);
        if constexpr (DO_DK) store_tile32<HD, F16>(dk, p.scale, dkp, p.k_rs, scr, lane);
        if constexpr (DO_DV) store_tile32<HD, F16>(dv, 1.0f, dvp, p.v_rs, scr, lane);
    }
}

// bit 0: forward, bit 1: backward (ctmi_attn_set_path; CTMI_ATTN_W32 gives the initial value)
std::atomic<int> g_w32_mask{-1};
int w32_mask() {
    int v = g_w32_mask.load(std::memory_order_relaxed);
    if (v < 0) { const char* e = getenv("CTMI_ATTN_W32"); v = e ? (atoi(e) & 3) : 3; g_w32_mask.store(v, std::memory_order_relaxed); }
    return v;
}
bool w32_ok(const AttnP& p) {
    return p.add_mask == nullptr && p.drop_thr == 0 && p.vec_ok && (p.hd == 64 || p.hd == 128) && p.causal &&
           p.Sq == p.Sk && p.Sk % 64 == 0 && p.Sk >= 64 && p.Sk <= 4096 && p.B * p.nh < (1 << 20);
}
template <typename K>
void launch32(K kern, int64_t grid, int threads, size_t lds, hipStream_t st, const AttnP& p) {
    ctmi_dyn_lds(reinterpret_cast<const void*>(kern), lds);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(threads), lds, st, p);
}
template <int HD, int NW> size_t lds_kv(const AttnP& p, int nst = 3) { return nst * (size_t)(2 * WT<HD, NW>::TILE) + 4 * (size_t)p.Sk; }
// the key-owned kernel keeps per-query statistics of the whole sequence in LDS: the score offset and delta, 8 bytes per query (4 in the
// dV pass, which has no dS); the exact path of left-padded batch rows reads l from L2.  At head_dim 128, S = 2048 that is 64 + 16 KiB:
// two workgroups per CU (with 1/l staged too it was 88 KiB: one, and the kernel ran at one wave per SIMD whatever its register count).
template <int HD, int NW> size_t lds_qg(const AttnP& p, int nst = 3, int mode = 0) {
    return nst * (size_t)(2 * WT<HD, NW>::TILE) + (mode == 1 ? 4 : 8) * (size_t)p.Sq;
}

}  // namespace

extern "C" int ctmi_attn_set_path(int mask) {
    const int prev = w32_mask();
    if (mask >= 0) g_w32_mask.store(mask & 3, std::memory_order_relaxed);
    return prev;
}

int ctmi_attn32_fwd(const AttnP& p, hipStream_t st, int f16) {
    if (!w32_ok(p) || !(w32_mask() & 1)) return 0;
    const int64_t BH = p.B * p.nh;
    constexpr int NW = CTMI_W32_FWD_NW, RPB = 32 * NW;
    const bool replace = p.future_fill > FINFO_MIN;
    const int64_t grid = ((p.Sq + RPB - 1) / RPB) * BH;
    if (f16) {
        if (p.hd == 64) {
            if (replace) launch32(&attn32_fwd_kernel<64, NW, true, true>, grid, 64 * NW, lds_kv<64, NW>(p, CTMI_W32_FWD_NST), st, p);
            else launch32(&attn32_fwd_kernel<64, NW, false, true>, grid, 64 * NW, lds_kv<64, NW>(p, CTMI_W32_FWD_NST), st, p);
        } else {
            if (replace) launch32(&attn32_fwd_kernel<128, NW, true, true>, grid, 64 * NW, lds_kv<128, NW>(p, CTMI_W32_FWD_NST), st, p);
            else launch32(&attn32_fwd_kernel<128, NW, false, true>, grid, 64 * NW, lds_kv<128, NW>(p, CTMI_W32_FWD_NST), st, p);
        }
    } else if (p.hd == 64) {
        if (replace) launch32(&attn32_fwd_kernel<64, NW, true>, grid, 64 * NW, lds_kv<64, NW>(p, CTMI_W32_FWD_NST), st, p);
        else launch32(&attn32_fwd_kernel<64, NW, false>, grid, 64 * NW, lds_kv<64, NW>(p, CTMI_W32_FWD_NST), st, p);
    } else {
        if (replace) launch32(&attn32_fwd_kernel<128, NW, true>, grid, 64 * NW, lds_kv<128, NW>(p, CTMI_W32_FWD_NST), st, p);
        else launch32(&attn32_fwd_kernel<128, NW, false>, grid, 64 * NW, lds_kv<128, NW>(p, CTMI_W32_FWD_NST), st, p);
    }
    return 1;
}

int ctmi_attn32_bwd(const AttnP& p, hipStream_t st, int f16) {
    // head_dim 128: dK^T and dV^T together are 128 accumulator registers, which leaves one wave per SIMD (256 VGPRs + 230 AGPRs used as
    // spill space; round 4: 915 us against the general kernels' 797 at B=4 S=2048 nh=32). Round 5 splits the key-owned kernel in two
    // passes (MODE 1: dV, MODE 2: dK) that each fit two waves per SIMD; the price is one more score recompute (8 matmul units against the
    // general kernels' 7; the algorithm has 5). Measured per kernel, same shape: dQ 228 us (904 TF/s executed), dV 237, dK 323 —
    // 775-788 us together against 797 (profiles/r05_attention_paths.txt).
    if (!w32_ok(p) || !(w32_mask() & 2)) return 0;
    const int64_t BH = p.B * p.nh;
    // dQ first: it also publishes delta = rowsum(dO * O) for the dK/dV kernel (same stream: ordered)
    constexpr int NW = CTMI_W32_BWD_NW, RPB = 32 * NW, NST = CTMI_W32_BWD_NST;
    if (p.hd == 128) {                                                           // round 5: dQ, then dV and dK in two passes (see MODE)
        const int64_t gq = ((p.Sq + RPB - 1) / RPB) * BH, gk = ((p.Sk + RPB - 1) / RPB) * BH;
        if (f16) {
            launch32(&attn32_dq_kernel<128, NW, true>, gq, 64 * NW, lds_kv<128, NW>(p, NST), st, p);
            launch32(&attn32_dkdv_kernel<128, NW, true, 1>, gk, 64 * NW, lds_qg<128, NW>(p, NST, 1), st, p);
            launch32(&attn32_dkdv_kernel<128, NW, true, 2>, gk, 64 * NW, lds_qg<128, NW>(p, NST), st, p);
        } else {
            launch32(&attn32_dq_kernel<128, NW, false>, gq, 64 * NW, lds_kv<128, NW>(p, NST), st, p);
            launch32(&attn32_dkdv_kernel<128, NW, false, 1>, gk, 64 * NW, lds_qg<128, NW>(p, NST, 1), st, p);
            launch32(&attn32_dkdv_kernel<128, NW, false, 2>, gk, 64 * NW, lds_qg<128, NW>(p, NST), st, p);
        }
        return 1;
    }
    if (f16) {
        launch32(&attn32_dq_kernel<64, NW, true>, ((p.Sq + RPB - 1) / RPB) * BH, 64 * NW, lds_kv<64, NW>(p, NST), st, p);
        launch32(&attn32_dkdv_kernel<64, NW, true>, ((p.Sk + RPB - 1) / RPB) * BH, 64 * NW, lds_qg<64, NW>(p, NST), st, p);
        return 1;
    }
    launch32(&attn32_dq_kernel<64, NW>, ((p.Sq + RPB - 1) / RPB) * BH, 64 * NW, lds_kv<64, NW>(p, NST), st, p);
    launch32(&attn32_dkdv_kernel<64, NW>, ((p.Sk + RPB - 1) / RPB) * BH, 64 * NW, lds_qg<64, NW>(p, NST), st, p);
    return 1;
}
