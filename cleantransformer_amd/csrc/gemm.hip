// MFMA GEMM family for gfx950 (CDNA4).
//   C[M,N] = epilogue(alpha * sum_k A(m,k) B(k,n)),  fp32 accumulate in the matrix cores.
// bf16 operands -> v_mfma_f32_16x16x32_bf16; fp32 operands -> v_mfma_f32_16x16x4_f32 (exact fp32, parity mode).
// One 256-thread workgroup (4 wavefronts, 2x2) owns a 128x128 output tile; each wave a 64x64 sub-tile held as
// 4x4 16x16 accumulators.  Operand tiles are staged HBM -> registers (16-byte loads) -> LDS with the global
// loads of tile t+1 in flight under the MFMAs of tile t, LDS double-buffered, one barrier per K-step.
// An operand may be stored K-major ([K][rows]); its fragments are then read transposed from LDS
// (ds_read_b64_tr_b16 for bf16), so the same kernel serves forward (x W^T), dgrad (dy W) and wgrad (dy^T x)
// without any transposed copies in HBM.
// The MFMA is issued with the operands swapped (D' = B A^T) so that every lane ends up with 4 *consecutive
// output columns* of one row: bias / residual / aux loads and the C store are 8-16 byte vector accesses.
// FAST instantiations (16-byte aligned operands, contiguous dimension a multiple of the vector width) have a
// branch-free main loop: unconditional vector loads from clamped addresses + selects; the generic instantiation
// handles any shape/alignment element-wise.  Small-tile-count problems (weight gradients) are split along K into
// fp32 slabs in a caller-provided workspace and reduced deterministically by a second kernel.
#include "common.h"
#include "reduce_jobs.h"
#include <type_traits>
#include "mma.h"
#include "prof.h"
#include <atomic>
// translation-unit split (see the note above ctmi_gemm_bf16_nt below)
#ifndef CTMI_GEMM_PART
#define CTMI_GEMM_PART (-1)
#endif
#ifndef CTMI_GELU_AUX_WT
#define CTMI_GELU_AUX_WT 0      // 1: the GELU pre-activation written through (sc1) instead of non-temporally — measured slower in the step, kept for A/B builds
#endif
#ifndef CTMI_SIDE_PRE8
#define CTMI_SIDE_PRE8 1      // round 6: the 256-row ping-pong tile prefetches its epilogue's side input too (0: rounds 3-5 — such epilogues forced onto the 128-row tile)
#endif
#define CTMI_GEMM_HAS(p) (CTMI_GEMM_PART == -1 || CTMI_GEMM_PART == (p))

// LDS-DMA ring depth and epilogue re-layout of the bf16 fast path (gemm_glds_kernel below).
//   free-running tiles (4 waves): 3 stages.
//   256-row ping-pong tile (8 waves): 4 stages of 32 KiB; logits-sized outputs re-layout through 4 x 8 KiB per-wave LDS patches (128-byte row
//     segments per store instruction), every other forward-layout output takes the XLANE instantiation: cross-lane re-layout, no patches.
//   128-row ping-pong tile: cross-lane re-layout (v_permlane16_swap), no patches, and TWO 32-k ring stages per phase (K2): 16 fragment reads +
//     6 DMA pieces | 32 MFMAs over a 6-stage ring of 24 KiB stages (144 KiB).  A phase costs ~730 cycles whatever it contains against 272
//     cycles of MFMAs per stage: most of it is latency (LDS round trip, DMA issue, two barriers) that a phase pays once.
// Everything that was measured and NOT adopted in rounds 1-4 (per-phase wave priorities, a fifth ring stage, DMA pieces before the fragment
// reads or inside the MFMA phase, a 4-step unrolled trip, the side-input tile through LDS or requested early, the timing / ablation
// instrumentation) lives as patches under tools/experiments/ (README.md there has the numbers), not as #if branches in this file.
constexpr bool glds_k2(bool pp, int wm) { return pp && wm == 4; }
constexpr int glds_ring(bool pp, int wm, bool xlane = false) { return !pp ? 3 : (glds_k2(pp, wm) ? 6 : 4); }
constexpr int glds_patch_bytes(bool pp, int wm, bool xlane = false) { return (pp && !xlane && wm != 4) ? 4 * 8192 : 0; }


template <typename T> struct Tile;
template <> struct Tile<bf16_t> { static constexpr int BM = 128, BN = 128, BK = 64, PADK = 8, PADR = 8; };
template <> struct Tile<f16_t>  { static constexpr int BM = 128, BN = 128, BK = 64, PADK = 8, PADR = 8; };     // (fp16: the register-staged kernel only)
template <> struct Tile<float>  { static constexpr int BM = 128, BN = 128, BK = 16, PADK = 4, PADR = 4; };

typedef short short4_t __attribute__((ext_vector_type(4)));
// ds_read_b64_tr_b16 through the compiler builtin (hipcc then tracks its lgkmcnt itself)
__device__ __forceinline__ uint2 lds_read_tr_b16(const void* p) {
    typedef __attribute__((address_space(3))) short4_t lds_v4;
    const short4_t r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(unsigned)(size_t)p);
    return __builtin_bit_cast(uint2, r);
}

// LDS image of one operand tile.  !KMAJOR: [ROWS][BK+PADK] (k contiguous).  KMAJOR: [BK][ROWS+PADR] (rows contiguous).
template <typename T, bool KMAJOR, int ROWS>
struct OpTile {
    static constexpr int BK = Tile<T>::BK;
    static constexpr int VEC = 16 / sizeof(T);
    static constexpr int PITCH = KMAJOR ? (ROWS + Tile<T>::PADR) : (BK + Tile<T>::PADK);
    static constexpr int LINES = KMAJOR ? BK : ROWS;
    static constexpr int CONTIG = KMAJOR ? ROWS : BK;
    static constexpr int CPL = CONTIG / VEC;                  // 16-byte chunks per line
    static constexpr int NCH = LINES * CPL / 256;             // chunks per thread
    static constexpr int ELEMS = LINES * PITCH;
    static_assert(LINES * CPL % 256 == 0, "tile must split evenly over 256 threads");

    // ---- generic (any shape / alignment): element-wise guarded loads
    static __device__ __forceinline__ void load_gen(uint4 (&regs)[NCH], const T* __restrict__ g, int64_t ld, int64_t row0,
                                                    int64_t k0, int64_t row_lim, int64_t k_lim, int tid) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int id = tid + 256 * i;
            const int line = id / CPL, c = (id % CPL) * VEC;
            const int64_t gline = (KMAJOR ? k0 : row0) + line;
            const int64_t gcol = (KMAJOR ? row0 : k0) + c;
            const int64_t line_lim = KMAJOR ? k_lim : row_lim;
            const int64_t col_lim = KMAJOR ? row_lim : k_lim;
            const T* p = g + gline * ld + gcol;
            T tmp[VEC];
#pragma unroll
            for (int j = 0; j < VEC; ++j) tmp[j] = (gline < line_lim && gcol + j < col_lim) ? p[j] : T{};
            regs[i] = *reinterpret_cast<const uint4*>(tmp);
        }
    }

    // ---- fast path: per-thread base pointers (clamped in-bounds) + validity bits, set up once per block
    struct Ptrs { const T* p[NCH]; };
    static __device__ __forceinline__ void init(Ptrs& P, const T* __restrict__ g, int64_t ld, int64_t row0, int64_t row_lim, int tid) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int id = tid + 256 * i;
            const int line = id / CPL, c = (id % CPL) * VEC;
            if (!KMAJOR) {
                const int64_t row = row0 + line;
                P.p[i] = g + (row < row_lim ? row : row_lim - 1) * ld + c;
            } else {
                const int64_t col = row0 + c;                            // row_lim % VEC == 0 on this path: chunk is all-in or all-out
                P.p[i] = g + (int64_t)line * ld + (col < row_lim ? col : 0);
            }
        }
    }
    // k0 = first k of the tile.  Rows (or K-major columns) outside the matrix were clamped to valid addresses by
    // init(): they load real-but-irrelevant data that only reaches accumulator rows/columns the epilogue never
    // stores, so the steady-state loads carry NO predicate (a select here makes hipcc re-introduce exec-masked
    // loads with a vmcnt(0) at every join).  Only the TAIL tile (reaches beyond K) zero-fills, because k >= K must
    // contribute exactly 0 to valid outputs (K % VEC == 0 for !KMAJOR on this path).
    template <bool TAIL>
    static __device__ __forceinline__ void load_fast(uint4 (&regs)[NCH], const Ptrs& P, int64_t ld, int64_t k0, int64_t K, int tid) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            if (!TAIL) {
                regs[i] = *reinterpret_cast<const uint4*>(P.p[i] + (KMAJOR ? k0 * ld : k0));
            } else {
                const int id = tid + 256 * i;
                const int line = id / CPL, c = (id % CPL) * VEC;
                const bool kin = KMAJOR ? ((k0 + line) < K) : ((k0 + c) < K);
                const int64_t kk = kin ? k0 : (KMAJOR ? (K - 1 - line) : (K - VEC - c));
                const uint4 v = *reinterpret_cast<const uint4*>(P.p[i] + (KMAJOR ? kk * ld : kk));
                regs[i] = kin ? v : make_uint4(0u, 0u, 0u, 0u);
            }
        }
    }

    static __device__ __forceinline__ void store(const uint4 (&regs)[NCH], T* __restrict__ tile, int tid) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int id = tid + 256 * i;
            const int line = id / CPL, c = (id % CPL) * VEC;
            *reinterpret_cast<uint4*>(tile + line * PITCH + c) = regs[i];
        }
    }
    // MFMA operand fragment for the 16 tile rows r16..r16+15 (lane's row = r16 + (lane&15)) at k offset
    // kofs = kk*K + (lane>>4)*KL: KL consecutive k.
    static __device__ __forceinline__ typename Mma<T>::Frag frag(const T* __restrict__ tile, int r16, int lane, int kofs);
};

template <> __device__ __forceinline__ short8 OpTile<bf16_t, false, 128>::frag(const bf16_t* __restrict__ tile, int r16, int lane, int kofs) {
    return *reinterpret_cast<const short8*>(tile + (r16 + (lane & 15)) * PITCH + kofs);
}
// K-major image [k][rows]: hardware transpose read.  Lane i of a 16-lane group supplies the address of k-row
// kofs + (i>>2), columns r16 + 4*(i&3) .. +3 and receives column r16+i for k = kofs .. kofs+3 (probe-verified layout).
template <> __device__ __forceinline__ short8 OpTile<bf16_t, true, 128>::frag(const bf16_t* __restrict__ tile, int r16, int lane, int kofs) {
    const int i = lane & 15;
    const bf16_t* p = tile + (kofs + (i >> 2)) * PITCH + r16 + 4 * (i & 3);
    const uint2 lo = lds_read_tr_b16(p);
    const uint2 hi = lds_read_tr_b16(p + 4 * PITCH);
    return __builtin_bit_cast(short8, make_uint4(lo.x, lo.y, hi.x, hi.y));
}
template <> __device__ __forceinline__ short8 OpTile<f16_t, false, 128>::frag(const f16_t* __restrict__ tile, int r16, int lane, int kofs) {
    return *reinterpret_cast<const short8*>(tile + (r16 + (lane & 15)) * PITCH + kofs);
}
template <> __device__ __forceinline__ short8 OpTile<f16_t, true, 128>::frag(const f16_t* __restrict__ tile, int r16, int lane, int kofs) {
    const int i = lane & 15;
    const f16_t* p = tile + (kofs + (i >> 2)) * PITCH + r16 + 4 * (i & 3);
    const uint2 lo = lds_read_tr_b16(p);
    const uint2 hi = lds_read_tr_b16(p + 4 * PITCH);
    return __builtin_bit_cast(short8, make_uint4(lo.x, lo.y, hi.x, hi.y));
}
template <> __device__ __forceinline__ float OpTile<float, false, 128>::frag(const float* __restrict__ tile, int r16, int lane, int kofs) {
    return tile[(r16 + (lane & 15)) * PITCH + kofs];
}
template <> __device__ __forceinline__ float OpTile<float, true, 128>::frag(const float* __restrict__ tile, int r16, int lane, int kofs) {
    return tile[kofs * PITCH + r16 + (lane & 15)];
}

struct GemmArgs {
    const void* A; const void* B; void* C;
    int64_t lda, ldb, ldc, M, N, K;
    float alpha; int beta;
    const float* bias; const void* residual; const void* aux_in; void* aux_out;
    int tiles_m, tiles_n, vec_c, vec8;
    int nt_c;                                              // C is far larger than the 256 MiB Infinity Cache: write it non-temporally
    int splits; int64_t k_per_split; float* slabs;        // split-K: partial products go to slabs[s][M][N] (fp32)
    int splitk_ok; float* ws; int64_t ws_bytes;           // split-K permission + caller workspace (the launch path decides)
};

// Grouped weight gradients (round 5): ONE persistent launch computes the weight gradients of several Linears — dW_p = dy_p^T x_p, every operand
// K-major with K = T rows — from a host-built list of work items over 128x256 output tiles.  Whole tiles accumulate over all of K and are stored;
// the tiles of the last, partial round of the 256 CUs are cut in two along K: each half stores its partial tile (and partial column sums) into a
// slab of the caller's workspace and a small second launch adds the two halves in a fixed order (deterministic).  (The first version ADDED the
// halves into zeroed memory with fp32 hardware atomics — two commutative contributions, also deterministic: 32 768 scattered 4-byte atomics per
// half tile cost 70 us per block, profiles/r05_wgrad_grouped.txt.)  The column sums of dy (the bias gradient) ride in the first tile column of
// a problem: ONE extra MFMA per wave and ring stage against a fragment of ones — each of the four waves of a row group starts its A fragments at
// a different 16-row block (i -> (i + wc) & 3), so wave wc's fragment 0 is block wc and the four waves cover the 64 rows between them.
constexpr int WG_MAXP = 4;
struct GroupProb { const void* A; const void* B; float* C; float* cs; float* part; float* cspart; int64_t lda, ldb, ldc, M, N, csn; };   // csn: length of cs (M: sums of the A operand; N: of the B operand, [in,out] weights)
    // C[M,N] = A^T B, A = [K][M], B = [K][N]; cs[M] = column sums of A (or null); part = [2][M][N] / cspart = [2][M] partial results of the K-halves
struct GroupedArgs : GemmArgs { GroupProb p[WG_MAXP]; const int4* items; int nitems; };
// work item (int4): x = problem | WG_PART | WG_HALF1 | WG_COLSUM, y = first row, z = first column of the tile, w = first K-step | K-steps << 16
constexpr int WG_PART = 16, WG_COLSUM = 32, WG_HALF1 = 64, WG_COLSUMB = 128;      // WG_COLSUMB (round 6): the column sums of the B operand (dy of an [in,out] weight), tiles of the first tile ROW
template <bool GRP, int WM, bool BSUM = false> struct GrpRegs { };
template <int WM> struct GrpRegs<true, WM, false> { f32x4 accs; bool cs_on; };      // column-sum accumulator of this wave's 16-row block, and whether the tile has one
template <int WM> struct GrpRegs<true, WM, true> { f32x4 accs; bool cs_on; f32x4 accb[4]; bool csb_on; };   // + the sums of the wave's four 16-column blocks of the B operand
#define WG_ONES8 (std::is_same<T, f16_t>::value ? short8{0x3C00, 0x3C00, 0x3C00, 0x3C00, 0x3C00, 0x3C00, 0x3C00, 0x3C00} \
                                                : short8{0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80})   /* eight 1.0 in the operand type (half: 0x3C00, bf16: 0x3F80) */

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <typename TO> __device__ __forceinline__ void store4(TO* p, const float* v);
template <> __device__ __forceinline__ void store4<float>(float* p, const float* v) { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }
template <> __device__ __forceinline__ void store4<bf16_t>(bf16_t* p, const float* v) { *reinterpret_cast<uint2*>(p) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])); }
template <> __device__ __forceinline__ void store4<f16_t>(f16_t* p, const float* v) { *reinterpret_cast<uint2*>(p) = make_uint2(pack_h2(v[0], v[1]), pack_h2(v[2], v[3])); }
template <typename TO> __device__ __forceinline__ void load4(const TO* p, float* v);
template <> __device__ __forceinline__ void load4<float>(const float* p, float* v) { float4 t = *reinterpret_cast<const float4*>(p); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
template <> __device__ __forceinline__ void load4<bf16_t>(const bf16_t* p, float* v) {
    uint2 t = *reinterpret_cast<const uint2*>(p);
    v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u); v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
}

template <> __device__ __forceinline__ void load4<f16_t>(const f16_t* p, float* v) {
    const uint2 t = *reinterpret_cast<const uint2*>(p);
    unpack_h2(t.x, v[0], v[1]); unpack_h2(t.y, v[2], v[3]);
}

template <typename T, typename TO, bool AK, bool BKM, int EPI, bool FAST>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs g) {
    using TA = OpTile<T, AK, Tile<T>::BM>;
    using TB = OpTile<T, BKM, Tile<T>::BN>;
    constexpr int BM = Tile<T>::BM, BN = Tile<T>::BN, BK = Tile<T>::BK;
    constexpr int MK = Mma<T>::K, KL = Mma<T>::KL;
    constexpr int STAGE = TA::ELEMS + TB::ELEMS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* smem = reinterpret_cast<T*>(smem_raw);

    // Block -> (tile, K-split).  XCD-aware: the dispatcher places block b on XCD b%8; give each XCD a contiguous run
    // of tile ids (bijective for any grid size).  Inside the run tiles are ordered in groups of 8 M-tiles x all
    // N-tiles, M fastest, so the ~64 blocks resident on one XCD share 8 A-panels and 8 B-panels in its 4 MiB L2.
    const int ntile = g.tiles_m * g.tiles_n;
    const int nblk = ntile * g.splits;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, q = nblk >> 3, r8 = nblk & 7;
    const int vid_all = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (bid >> 3);
    const int split = vid_all / ntile, vid = vid_all - split * ntile;
    constexpr int GM = 8;
    const int group = vid / (GM * g.tiles_n), first_m = group * GM;
    const int gm = min(GM, g.tiles_m - first_m);
    const int in_group = vid - group * GM * g.tiles_n;
    const int tm = first_m + in_group % gm, tn = in_group / gm;
    const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;
    const int64_t kbeg = (int64_t)split * g.k_per_split;
    const int64_t kend = min(g.K, kbeg + g.k_per_split);

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1;
    const T* A = reinterpret_cast<const T*>(g.A);
    const T* B = reinterpret_cast<const T*>(g.B);

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nt = (int)((kend - kbeg + BK - 1) / BK);
    const bool ktail = ((kend - kbeg) % BK) != 0;
    uint4 ra[TA::NCH], rb[TB::NCH];
    typename TA::Ptrs pa;
    typename TB::Ptrs pb;
    if (FAST) {
        TA::init(pa, A, g.lda, m0, g.M, tid);
        TB::init(pb, B, g.ldb, n0, g.N, tid);
    }
    auto load_tile = [&](int t) {
        const int64_t k0 = kbeg + (int64_t)t * BK;
        if (FAST) {
            if (ktail && t == nt - 1) { TA::template load_fast<true>(ra, pa, g.lda, k0, kend, tid); TB::template load_fast<true>(rb, pb, g.ldb, k0, kend, tid); }
            else { TA::template load_fast<false>(ra, pa, g.lda, k0, kend, tid); TB::template load_fast<false>(rb, pb, g.ldb, k0, kend, tid); }
        } else {
            TA::load_gen(ra, A, g.lda, m0, k0, g.M, kend, tid);
            TB::load_gen(rb, B, g.ldb, n0, k0, g.N, kend, tid);
        }
    };
    load_tile(0);
    TA::store(ra, smem, tid);
    TB::store(rb, smem + TA::ELEMS, tid);
    __syncthreads();

    for (int t = 0; t < nt; ++t) {
        const int cur = t & 1;
        if (t + 1 < nt) load_tile(t + 1);
        const T* as = smem + cur * STAGE;
        const T* bs = as + TA::ELEMS;
#pragma unroll
        for (int kk = 0; kk < BK / MK; ++kk) {
            const int kofs = kk * MK + (lane >> 4) * KL;
            typename Mma<T>::Frag af[4], bf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = TA::frag(as, wr * 64 + i * 16, lane, kofs);
#pragma unroll
            for (int j = 0; j < 4; ++j) bf[j] = TB::frag(bs, wc * 64 + j * 16, lane, kofs);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = Mma<T>::mma(bf[j], af[i], acc[i][j]);   // D'[n][m]
        }
        if (t + 1 < nt) {
            T* nx = smem + (cur ^ 1) * STAGE;
            TA::store(ra, nx, tid);
            TB::store(rb, nx + TA::ELEMS, tid);
        }
        __syncthreads();
    }

    // epilogue: lane holds C[m][n..n+3], m = m0+wr*64+i*16+(lane&15), n = n0+wc*64+j*16+(lane>>4)*4
    if (g.splits > 1) {
        float* S = g.slabs + (int64_t)split * g.M * g.N;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t m = m0 + wr * 64 + i * 16 + (lane & 15);
            if (m >= g.M) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int64_t n = n0 + wc * 64 + j * 16 + (lane >> 4) * 4;
                if (n >= g.N) continue;
                float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                if (n + 4 <= g.N && (g.N & 3) == 0) store4<float>(S + m * g.N + n, v);
                else { for (int r = 0; r < 4; ++r) if (n + r < g.N) S[m * g.N + n + r] = v[r]; }
            }
        }
        return;
    }
    TO* C = reinterpret_cast<TO*>(g.C);
    const T* R = reinterpret_cast<const T*>(g.residual);
    const T* AUXI = reinterpret_cast<const T*>(g.aux_in);
    T* AUXO = reinterpret_cast<T*>(g.aux_out);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t m = m0 + wr * 64 + i * 16 + (lane & 15);
        if (m >= g.M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t n = n0 + wc * 64 + j * 16 + (lane >> 4) * 4;
            if (n >= g.N) continue;
            const bool full = (n + 4 <= g.N) && g.vec_c;
            const int64_t off = m * g.ldc + n;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = g.alpha * acc[i][j][r];
            if (g.bias != nullptr) {
                if (full) { const float4 bb = *reinterpret_cast<const float4*>(g.bias + n); v[0] += bb.x; v[1] += bb.y; v[2] += bb.z; v[3] += bb.w; }
                else { for (int r = 0; r < 4; ++r) if (n + r < g.N) v[r] += g.bias[n + r]; }
            }
            if (EPI == CTMI_EPI_GELU || EPI == CTMI_EPI_GELUG) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = Cvt<T>::to_f(Cvt<T>::from_f(v[r]));       // GELU sees the stored value
                float ax[4];                                                                    // what the backward wants: x, or gelu'(x)
#pragma unroll
                for (int r = 0; r < 4; ++r) ax[r] = (EPI == CTMI_EPI_GELUG) ? gelu_tanh_grad_f(v[r]) : v[r];
                if (full) store4<T>(AUXO + off, ax);
                else { for (int r = 0; r < 4; ++r) if (n + r < g.N) AUXO[off + r] = Cvt<T>::from_f(ax[r]); }
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = gelu_tanh_f(v[r]);
            } else if (EPI == CTMI_EPI_RELU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
            } else if (EPI == CTMI_EPI_DGELU || EPI == CTMI_EPI_DRELU || EPI == CTMI_EPI_MUL) {
                float u[4] = {0.f, 0.f, 0.f, 0.f};
                if (full) load4<T>(AUXI + off, u);
                else { for (int r = 0; r < 4; ++r) if (n + r < g.N) u[r] = Cvt<T>::to_f(AUXI[off + r]); }
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = (EPI == CTMI_EPI_DGELU) ? v[r] * gelu_tanh_grad_f(u[r]) : (EPI == CTMI_EPI_MUL ? v[r] * u[r] : (u[r] > 0.f ? v[r] : 0.f));
            }
            if (R != nullptr) {
                float u[4] = {0.f, 0.f, 0.f, 0.f};
                if (full) load4<T>(R + off, u);
                else { for (int r = 0; r < 4; ++r) if (n + r < g.N) u[r] = Cvt<T>::to_f(R[off + r]); }
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += u[r];
            }
            if (g.beta) {
                float u[4] = {0.f, 0.f, 0.f, 0.f};
                if (full) load4<TO>(C + off, u);
                else { for (int r = 0; r < 4; ++r) if (n + r < g.N) u[r] = Cvt<TO>::to_f(C[off + r]); }
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += u[r];
            }
            if (full) store4<TO>(C + off, v);
            else { for (int r = 0; r < 4; ++r) if (n + r < g.N) C[off + r] = Cvt<TO>::from_f(v[r]); }
        }
    }
}

// =================================================================================================================
// bf16 fast path v2: LDS-DMA staging (global_load_lds_dwordx4), 3-stage ring, big wave tiles.
//   * workgroup = 4 waves (2x2) on a (WM*32) x 128 output tile; each wave (WM*16) x 64 -> WM x 4 accumulators.
//     WM = 8 (256x128) halves the LDS fragment traffic per MFMA of the 128x128 tile; WM = 4 serves small problems.
//   * BK = 32 (one MFMA K-step per stage).  A stage is the raw tile, UNPADDED, written by the LDS-DMA engine: each
//     wave-instruction drops 64 x 16 B = 1 KiB at a wave-uniform LDS base (M0) + lane*16, so the LDS image is linear
//     in the order of the lanes; bank conflicts are removed by permuting which 16-byte GLOBAL chunk each lane fetches
//     (XOR swizzle on the source address) and applying the same permutation when fragments are read.
//   * no VGPR staging, no ds_write: the LDS write port (79 B/clk for ds_write_b128) was the v1 bottleneck.
//   * 3 stages in flight: loads of tile t+2 are issued right after the barrier that retires tile t-1; each wave
//     waits only for its own oldest tile with a counted s_waitcnt vmcnt(N) (never 0 in steady state) and one
//     s_barrier per K-step orders LDS-DMA writes against the other waves' reads.  The DMA instructions are inline
//     asm, so hipcc neither counts nor drains them.
// Requires: bf16, 16-byte aligned operands, K % 32 == 0 (per split).  Anything else takes the v1 kernel above.
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst /* wave-uniform LDS byte address */) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ int swz4(int key) { return (0x78 >> (2 * key)) & 3; }          // {0,2,3,1}: see GTile<false>
__device__ __forceinline__ int kkey(int k) { return (k & 3) | (((k >> 3) & 1) << 2); }

template <bool KMAJOR, int ROWS>
struct GTile {
    static constexpr int BK = 32;
    static constexpr int BYTES = ROWS * BK * 2;
    static constexpr int NINSTR = BYTES / 1024;             // 1-KiB wave-instructions per stage
    static constexpr int PER_WAVE = NINSTR / 4;
    static constexpr int CPR = ROWS / 8;                    // KMAJOR: 16-byte chunks per k-row
    static constexpr int ROWB = ROWS * 2;                   // KMAJOR: bytes per k-row

    // global source of LDS 16-byte slot P of the stage (k offset NOT included).
    //  !KMAJOR image: [row][4 chunks of 8 k] (64 B per row).  slot (row, c') holds global chunk c' ^ swz4((row>>2)&3):
    //   a ds_read_b128 lane group (16 rows x one k-chunk, in the hardware's 4+4+8 lane grouping) then covers all 16
    //   16-byte slots of the 256-byte bank row exactly once.
    //  KMAJOR image: [k][ROWS] (rows contiguous).  32-byte blocks of a k-row are XOR-ed with kkey(k) (3 bits), so the
    //   8 k-rows touched by one half-wave of ds_read_b64_tr_b16 land in 8 different 32-byte bank groups.
    template <typename E> static __device__ __forceinline__ const E* src(const E* __restrict__ g, int64_t ld, int64_t row0, int64_t row_lim, int P) {
        if (!KMAJOR) {
            const int row = P >> 2, cp = P & 3;
            const int c = cp ^ swz4((row >> 2) & 3);
            const int64_t grow = min(row0 + row, row_lim - 1);
            return g + grow * ld + c * 8;
        } else {
            const int k = P / CPR, cp = P % CPR;
            const int b32p = cp >> 1;
            const int b32 = (b32p & ~7) | ((b32p ^ kkey(k)) & 7);
            const int64_t col = row0 + (((b32 << 1) | (cp & 1)) * 8);
            return g + (int64_t)k * ld + (col < row_lim ? col : 0);
        }
    }
    static __device__ __forceinline__ short8 frag(const unsigned char* __restrict__ tile, int r16, int lane) {
        const int i = lane & 15, g = lane >> 4;
        if (!KMAJOR) {
            const int r = r16 + i;
            const int cpp = g ^ swz4((r >> 2) & 3);
            return *reinterpret_cast<const short8*>(tile + r * 64 + cpp * 16);
        } else {
            const int k = g * 8 + (i >> 2);
            const int b32 = r16 >> 4;
            const int b32s = (b32 & ~7) | ((b32 ^ kkey(k)) & 7);
            const bf16_t* p = reinterpret_cast<const bf16_t*>(tile + k * ROWB + b32s * 32 + 8 * (i & 3));
            const uint2 lo = lds_read_tr_b16(p);
            const uint2 hi = lds_read_tr_b16(p + 4 * ROWS);                 // kkey(k+4) == kkey(k)
            return __builtin_bit_cast(short8, make_uint4(lo.x, lo.y, hi.x, hi.y));
        }
    }
};

template <typename TO, bool AK, bool BKM, int EPI, int WM, int WGN, bool PP, bool RES, bool XLANE, bool GRP, typename GA, typename TE = bf16_t, bool BSUM = false>
__device__ __forceinline__ void glds_body(const GA g) {   // (by value: through a reference hipcc kept a 16-byte piece of the kernel arguments in scratch memory, reloaded in every epilogue)
    using T = TE;                                                           // bf16_t (the measured path) or f16_t (round 5: same schedules, v_mfma_f32_16x16x32_f16)
    static_assert(!GRP || (PP && WM == 4 && WGN == 4 && AK && BKM && EPI == CTMI_EPI_NONE && sizeof(TO) == 4), "grouped launches: weight gradients on the 128x256 ping-pong tile");
    constexpr int NW = 2 * WGN;                                             // waves: 2 along M x WGN along N
    constexpr int BM = WM * 32, BN = WGN * 64, BK = 32, NST = glds_ring(PP, WM, XLANE);
    using TA = GTile<AK, BM>;
    using TB = GTile<BKM, BN>;
    constexpr int STAGE = TA::BYTES + TB::BYTES;
    // LDS ring: the A tiles of all stages, then the B tiles of all stages — every fragment read is then ONE lane-constant base register per
    // operand plus a 16-bit immediate (stage * tile bytes + fragment row offset < 64 KiB), also across the stages of an unrolled trip
    constexpr int AOFF = 0, BOFF = NST * TA::BYTES;
    constexpr int PA = TA::NINSTR / NW, PB = TB::NINSTR / NW;               // DMA instructions per wave per stage
    constexpr int LOADS = PA + PB;
    static_assert(PP ? (LOADS == 3 || LOADS == 4) : (LOADS == 4 || LOADS == 6), "vmcnt immediates below assume these DMA piece counts");
    constexpr bool K2 = glds_k2(PP, WM);                                      // two ring stages per phase
    constexpr int FILL = K2 ? NST - 2 : NST - 1;                              // stages in flight between K-steps (K2: two slots stay free for the pair to issue)
    constexpr int LAND = K2 ? 2 : 1;                                          // stages that have LANDED when a K-step starts (a pair reads two): every counted wait keeps that
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];

    const int tiles_m = (int)((g.M + BM - 1) / BM);
    const int tiles_n = (int)((g.N + BN - 1) / BN);
    const int ntile = tiles_m * tiles_n;
    int nwork_;
    if constexpr (GRP) nwork_ = g.nitems; else nwork_ = ntile * g.splits;
    const int nwork = nwork_;                                                 // work items: (split, output tile) — or the grouped launch's list
    const int bid = blockIdx.x, G = gridDim.x;                                // G == nwork, or a multiple of 8 (persistent)
    // (weight-gradient layout on the 256-row tile = the LM head's [V,H] gradient, four tile columns wide: groups of ONE tile row put the four
    // workgroups that share a dlogits panel next to each other in the order — 3.55 vs 3.61 ms, profiles/r03_gemm_tile_sweep.txt; the
    // forward wants 4: 3.85 vs 4.32 ms)
    constexpr int GM = (WM == 8) ? ((AK && BKM) ? 1 : 4) : 8;
    // work item w -> (split, tile origin).  Items w, w+8, w+16, ... run on one XCD (workgroup b lands on XCD b % 8), so
    // each XCD gets a CONTIGUOUS range of the grouped tile order and its private L2 sees the operand panels reused.
    auto decode = [&](int w, int64_t& m0, int64_t& n0, int& split) {
        const int xcd = w & 7, q = nwork >> 3, r8 = nwork & 7;
        const int vid_all = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (w >> 3);
        split = vid_all / ntile;
        const int vid = vid_all - split * ntile;
        const int group = vid / (GM * tiles_n), first_m = group * GM;
        const int gm = min(GM, tiles_m - first_m);
        const int in_group = vid - group * GM * tiles_n;
        m0 = (int64_t)(first_m + in_group % gm) * BM;
        n0 = (int64_t)(in_group / gm) * BN;
    };

    const int tid = threadIdx.x, lane = tid & 63, lane_ = lane;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid / WGN, wc = wid % WGN;
    const T* A = reinterpret_cast<const T*>(g.A);
    const T* B = reinterpret_cast<const T*>(g.B);
    const int64_t astep_c = AK ? (int64_t)BK * g.lda : BK, bstep_c = BKM ? (int64_t)BK * g.ldb : BK;
    int64_t astep_g = 0, bstep_g = 0;                                         // grouped launches: the K-step strides of the DMA stream's current problem
    const unsigned lds0 = (unsigned)(size_t)smem_raw;

    // ---- issue side: the DMA stream runs ahead of the MFMA stream by two K-steps and crosses work-item boundaries,
    // so the first stages of the next output tile land while this tile's last K-steps and its epilogue execute.
    const T* pa[PA];
    const T* pb[PB];
    int wi = bid, ti = 0, nti = 0;                                            // issue-side work item, K-step, K-steps
    auto setup_issue = [&]() {
        if constexpr (GRP) {
            const int4 it = g.items[wi];
            const GroupProb& P = g.p[it.x & (WG_MAXP - 1)];
            const int64_t kbeg = (int64_t)(it.w & 0xffff) * BK;
            nti = it.w >> 16; ti = 0;
            astep_g = (int64_t)BK * P.lda; bstep_g = (int64_t)BK * P.ldb;
#pragma unroll
            for (int j = 0; j < PA; ++j) pa[j] = TA::src(reinterpret_cast<const T*>(P.A), P.lda, it.y, P.M, (wid * PA + j) * 64 + lane) + kbeg * P.lda;
#pragma unroll
            for (int j = 0; j < PB; ++j) pb[j] = TB::src(reinterpret_cast<const T*>(P.B), P.ldb, it.z, P.N, (wid * PB + j) * 64 + lane) + kbeg * P.ldb;
            return;
        }
        int64_t m0, n0; int split;
        decode(wi, m0, n0, split);
        const int64_t kbeg = (int64_t)split * g.k_per_split;
        const int64_t kend = min(g.K, kbeg + g.k_per_split);
        nti = (int)((kend - kbeg) / BK); ti = 0;
#pragma unroll
        for (int j = 0; j < PA; ++j) pa[j] = TA::src(A, g.lda, m0, g.M, (wid * PA + j) * 64 + lane) + (AK ? kbeg * g.lda : kbeg);
#pragma unroll
        for (int j = 0; j < PB; ++j) pb[j] = TB::src(B, g.ldb, n0, g.N, (wid * PB + j) * 64 + lane) + (BKM ? kbeg * g.ldb : kbeg);
    };
    // one DMA piece (j-th of this wave's LOADS per stage) of the stage being issued; pointers advance
    auto issue_one = [&](int stage_buf, int j) {
        if (j < PA) { glds16(pa[j], lds0 + AOFF + stage_buf * TA::BYTES + (wid * PA + j) * 1024); pa[j] += (GRP ? astep_g : astep_c); }
        else { const int jb = j - PA; glds16(pb[jb], lds0 + BOFF + stage_buf * TB::BYTES + (wid * PB + jb) * 1024); pb[jb] += (GRP ? bstep_g : bstep_c); }
    };
    // all pieces of one stage in ONE statement: M0 saved/restored once, the second piece of each operand reached by
    // bumping M0 (the ping-pong schedule issues a stage back-to-back, so the SALU traffic around each DMA matters)
    auto issue_stage = [&](int stage_buf) {
        const unsigned da = lds0 + AOFF + stage_buf * TA::BYTES + wid * PA * 1024, db = lds0 + BOFF + stage_buf * TB::BYTES + wid * PB * 1024;
        unsigned keep;
        static_assert(!PP || (PB == 2 && (PA == 1 || PA == 2)), "issue_stage handles 1-2 A pieces and 2 B pieces per wave");
        if constexpr (!PP) { (void)da; (void)db; (void)keep; }
        else if constexpr (PA == 2) {
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
                         "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\t"
                         "s_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, off\n\t"
                         "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(pa[0]), "v"(pa[1]), "v"(pb[0]), "v"(pb[1]), "s"(da), "s"(db) : "memory", "scc");
        } else {
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
                         "s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\t"
                         "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(pa[0]), "v"(pb[0]), "v"(pb[1]), "s"(da), "s"(db) : "memory", "scc");
        }
#pragma unroll
        for (int j = 0; j < PA; ++j) pa[j] += (GRP ? astep_g : astep_c);
#pragma unroll
        for (int j = 0; j < PB; ++j) pb[j] += (GRP ? bstep_g : bstep_c);
    };
    auto stage_issued = [&]() {                                               // bookkeeping after a whole stage went out
        if (++ti == nti) { wi += G; if (wi < nwork) setup_issue(); }
    };


    f32x4 acc[WM][4];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // grouped weight gradients: column sums of the A operand (the bias gradient) of this tile, accumulated by MFMAs against a fragment of ones
    // (state of the grouped instantiation only: the other kernels do not even declare it)
    static_assert(!BSUM || GRP, "B-operand column sums: grouped launches only");
    GrpRegs<GRP, WM, BSUM> gs;
    if constexpr (GRP) { gs.accs = f32x4{0.f, 0.f, 0.f, 0.f}; gs.cs_on = false; }
    if constexpr (BSUM) {
        gs.csb_on = false;
#pragma unroll
        for (int j = 0; j < 4; ++j) gs.accb[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // first row (inside the wave's WM*16 rows) of A fragment i: grouped launches rotate the blocks by the wave's column index
    auto arow = [&](int i) { return GRP ? ((i + wc) & (WM - 1)) * 16 : i * 16; };

    auto epilogue = [&](const int64_t m0, const int64_t n0, const int split) {
        // the lane id is laundered through an empty asm so none of the address arithmetic below is loop-invariant for
        // hipcc: hoisted out of the persistent tile loop it stayed live across the MFMA loop, spilled, and every reload
        // (a scratch load: vmcnt) then waited for the stores in front of it
        int lane = lane_;
        asm volatile("" : "+v"(lane));
    // epilogue (same contract as v1): lane holds C[m][n..n+3], m = m0+wr*WM*16+i*16+(lane&15), n = n0+wc*64+j*16+(lane>>4)*4
    if constexpr (GRP) {
        // `split` carries the work item's x word: problem, WG_PART (a K-half: the result goes to the partial slabs), WG_HALF1, WG_COLSUM.
        // Tiles of a grouped launch are interior by construction.  The cross-lane re-layout of the 128-row tile (v_permlane16_swap of column
        // tiles j, j+1: 8 consecutive columns per lane); accumulator row i holds block (i + wc) & 3 of the wave's rows (see arow).
        const GroupProb& P = g.p[split & (WG_MAXP - 1)];
        const bool part = (split & WG_PART) != 0, half1 = (split & WG_HALF1) != 0;
        float* Cg = part ? P.part + (half1 ? P.M * P.N : 0) : P.C;
        const int64_t ldc = P.ldc;
        const int64_t mw = m0 + wr * (WM * 16), nw = n0 + wc * 64;
        const int q = lane >> 4;
        float* Cl = Cg + (mw + (lane & 15)) * ldc + nw + 16 * (q & 1) + 8 * (q >> 1);
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int jp = 0; jp < 2; ++jp) {
                const f32x4 a = acc[i][2 * jp], b = acc[i][2 * jp + 1];
                float a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3], b0 = b[0], b1 = b[1], b2 = b[2], b3 = b[3];
                asm volatile("s_nop 1\n\t"
                             "v_permlane16_swap_b32 %0, %4\n\t"
                             "v_permlane16_swap_b32 %1, %5\n\t"
                             "v_permlane16_swap_b32 %2, %6\n\t"
                             "v_permlane16_swap_b32 %3, %7"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3));
                float* d = Cl + (int64_t)arow(i) * ldc + 32 * jp;
                st_wt16(d, f32x4{a0, a1, a2, a3});
                st_wt16(d + 4, f32x4{b0, b1, b2, b3});
            }
        if (gs.cs_on) {
            // gs.accs: D'[n][m] = sum_k 1 * A[k][m] in every n for the wave's block arow(0): lanes 0-15 hold its 16 rows in element 0
            if ((lane >> 4) == 0) {
                float* d = (part ? P.cspart + (half1 ? P.csn : 0) : P.cs) + mw + arow(0) + lane;
                *d = gs.accs[0];
            }
        }
        if constexpr (BSUM) {
            // gs.accb[j]: D[m][n] = sum_k 1 * B[k][n] in every m: the lanes of accumulator row 0 hold columns 4 (lane >> 4) .. + 3 of block j
            if (gs.csb_on && (lane & 15) == 0) {
                float* d = (part ? P.cspart + (half1 ? P.csn : 0) : P.cs) + nw + (lane >> 4) * 4;
#pragma unroll
                for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(d + j * 16) = gs.accb[j];
            }
        }
        return;
    }
    if (g.splits > 1) {
        float* S = g.slabs + (int64_t)split * g.M * g.N;
#pragma unroll
        for (int i = 0; i < WM; ++i) {
            const int64_t m = m0 + wr * (WM * 16) + i * 16 + (lane & 15);
            if (m >= g.M) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int64_t n = n0 + wc * 64 + j * 16 + (lane >> 4) * 4;
                if (n >= g.N) continue;
                float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                if (n + 4 <= g.N && (g.N & 3) == 0) store4<float>(S + m * g.N + n, v);
                else { for (int r = 0; r < 4; ++r) if (n + r < g.N) S[m * g.N + n + r] = v[r]; }
            }
        }
        return;
    }
    TO* C = reinterpret_cast<TO*>(g.C);
    const T* R = reinterpret_cast<const T*>(g.residual);
    const T* AUXI = reinterpret_cast<const T*>(g.aux_in);
    T* AUXO = reinterpret_cast<T*>(g.aux_out);
    // Interior tiles with aligned pointers (the common case) take a straight-line epilogue: no per-lane bounds test, so
    // hipcc emits no exec-masked branches around the vector loads/stores; edge tiles take the guarded path.
    const bool interior = g.vec_c && (m0 + BM <= g.M) && (n0 + BN <= g.N);
    if constexpr (PP) {
        // LDS-shuffled epilogue: the MFMA accumulator layout (lane = one row, 4 columns) makes 8-byte stores that touch
        // 16 rows per instruction; measured on the LM-head forward those stores cost 26 % of the kernel.  Each wave
        // instead passes its tile through a private 8 KiB LDS patch as fp32, 32 rows at a time (XOR-swizzled 16-byte
        // chunks, conflict-free both ways), and reads it back with 8 lanes per row: residual / GELU-input / C loads
        // and all stores become 16-byte row-contiguous accesses (8 full 128-byte lines per bf16 store instruction).
        // Kernels with the cross-lane epilogue take the fast path only in its PLAIN form (no inline residual load, no beta): with the
        // non-plain variants compiled into the same kernel, hipcc's wait-count pass carried a pending vector-memory event around the
        // persistent tile loop and put an `s_waitcnt vmcnt(0)` at the head of the STEADY K-LOOP — every K-step drained the whole LDS-DMA
        // ring, in every 128-row and every XLANE ping-pong kernel (~190 of the step's layer-GEMM launches; found in round 4, see
        // tools/kernel_isa_scan.py, which now checks every instantiation).  The combinations this excludes (a residual on a kernel that
        // was not instantiated for it, beta with a bf16 output) do not occur on the training path; they take the guarded epilogue below.
        constexpr bool USE_DIRECT = WM == 4 || XLANE;
        constexpr bool RES_IN_KERNEL = RES && WM == 4 && !(EPI == CTMI_EPI_DGELU || EPI == CTMI_EPI_DRELU || EPI == CTMI_EPI_MUL);   // == PRE_RES below
        const bool plain_tile = (RES_IN_KERNEL || R == nullptr) && !g.beta;
        if (interior && g.vec8 && (!USE_DIRECT || plain_tile)) {
            unsigned char* scr = smem_raw + NST * STAGE + wc * 8192;          // the two row groups never overlap in time
            const int64_t mw = m0 + wr * (WM * 16), nw = n0 + wc * 64;
            // (round 4: in the cross-lane epilogue the bias switch is a compile-time variant too, with the loads inside the variant that uses them)
            auto load_bias4 = [&](f32x4 (&b4)[4]) {
#pragma unroll
                for (int j = 0; j < 4; ++j) b4[j] = *reinterpret_cast<const f32x4*>(g.bias + nw + j * 16 + (lane >> 4) * 4);
            };
            // the side input of a pass (GELU-derivative input, or the residual rows when the kernel is instantiated with
            // RES) is fetched one pass ahead, into the registers the previous pass's accumulators just freed: loaded
            // where it is used, each of the 16 row groups of a tile would expose a full global-load latency
            // (only the 128-row tiles have the registers for it: with 128 accumulators live hipcc spills the prefetch)
            constexpr bool PRE_AUX = (EPI == CTMI_EPI_DGELU || EPI == CTMI_EPI_DRELU || EPI == CTMI_EPI_MUL) && (WM == 4 || CTMI_SIDE_PRE8);
            constexpr bool PRE_RES = RES && !PRE_AUX && WM == 4;
            constexpr bool PRE = PRE_AUX || PRE_RES;
            const T* side = PRE_AUX ? AUXI : R;
            // per-lane element offset of (row lane>>3, column chunk lane&7) of the wave tile; rows advance by 8 * ldc
            const int64_t off0 = (mw + (lane >> 3)) * g.ldc + nw + (lane & 7) * 8;
            const int64_t row8 = 8 * g.ldc;
            // One 16-byte row piece (8 consecutive columns in v[]) through the epilogue proper: activation (+ pre-activation out) or
            // activation derivative, residual, beta, conversion and the store.  Shared by both re-layouts below; always inlined
            // (v[] lives in registers).
            auto finish = [&](auto plain_c, auto nt_flag, float (&v)[8], const int64_t off, const uint4& pre_it) __attribute__((always_inline)) {
                constexpr bool PLAIN = decltype(plain_c)::value;            // no inline residual load, no beta
                constexpr bool NT = decltype(nt_flag)::value;
                if (EPI == CTMI_EPI_GELU) {
                    const uint4 tb = pack16<T>(v);
                    // the pre-activation is kept for the BACKWARD only (a whole forward and half a backward away): CTMI_GELU_AUX_NT writes it
                    // non-temporally, so that the activation next to it — the A operand of the very next GEMM — is what stays in the caches
                    // (same box, interleaved: forward chain of 24 blocks 6.16 / 6.26 -> 6.07 / 6.05 ms, step 37.01-37.35 -> 36.83-37.08 ms)
                    if constexpr (sizeof(T) == 2 && CTMI_GELU_AUX_WT) st_wt16(AUXO + off, tb);                 // (-DCTMI_GELU_AUX_WT=1, A/B builds only: written through like the tile itself — measured +0.2 ms per step against the non-temporal store below, profiles/r06_boundary_dirty.txt)
                    else if constexpr (sizeof(T) == 2) __builtin_nontemporal_store(__builtin_bit_cast(u32x4, tb), reinterpret_cast<u32x4*>(AUXO + off));   // (the builtin, not asm: hipcc's hazard recognizer does not see into asm, and a first asm version — no wait state between the 16-byte store and the next write of its data registers — stored garbage in a few rows; tests/test_gpu_ops.py::test_gemm_at_the_step_shapes_sampled_vs_fp64 caught it)
                    else *reinterpret_cast<uint4*>(AUXO + off) = tb;
                    unpack16<T>(tb, v);
#pragma unroll
                    for (int r = 0; r < 8; r += 2) { const f32x2 y = gelu_tanh_pk(f32x2{v[r], v[r + 1]}); v[r] = y[0]; v[r + 1] = y[1]; }
                } else if (EPI == CTMI_EPI_GELUG) {
                    // forward of the MLP activation that ALSO leaves gelu'(x) for the backward (instead of x): the logistic factor
                    // s is shared, so the derivative costs ~5 more packed ops here and the 4h->h data-gradient GEMM's epilogue
                    // shrinks to one multiply per element (CTMI_EPI_MUL) — no exp / rcp on that kernel's critical tail
                    const uint4 tb = pack16<T>(v);
                    unpack16<T>(tb, v);                                                     // the activation sees the value as stored before
                    float gd[8];
#pragma unroll
                    for (int r = 0; r < 8; r += 2) {
                        const f32x2 x = f32x2{v[r], v[r + 1]}, x2 = x * x;
                        const f32x2 sg = gelu_sigma_pk(x, x2);
                        const f32x2 d = sg * (x * (1.0f - sg) * (x2 * 0.21406445f + 1.5957691f) + 1.0f);
                        const f32x2 y = x * sg;
                        gd[r] = d[0]; gd[r + 1] = d[1]; v[r] = y[0]; v[r + 1] = y[1];
                    }
                    st_wt16(AUXO + off, pack16<T>(gd));
                } else if (EPI == CTMI_EPI_RELU) {
#pragma unroll
                    for (int r = 0; r < 8; ++r) v[r] = fmaxf(v[r], 0.f);
                } else if (EPI == CTMI_EPI_DGELU || EPI == CTMI_EPI_DRELU || EPI == CTMI_EPI_MUL) {
                    float u[8];
                    if constexpr (PRE_AUX) unpack16<T>(pre_it, u);
                    else unpack16<T>(*reinterpret_cast<const uint4*>(AUXI + off), u);
                    if (EPI == CTMI_EPI_MUL) {
#pragma unroll
                        for (int r = 0; r < 8; ++r) v[r] *= u[r];
                    } else if (EPI == CTMI_EPI_DGELU) {
#pragma unroll
                        for (int r = 0; r < 8; r += 2) { const f32x2 y = f32x2{v[r], v[r + 1]} * gelu_tanh_grad_pk(f32x2{u[r], u[r + 1]}); v[r] = y[0]; v[r + 1] = y[1]; }
                    } else {
#pragma unroll
                        for (int r = 0; r < 8; ++r) v[r] = u[r] > 0.f ? v[r] : 0.f;
                    }
                }
                if constexpr (PRE_RES) {
                    float u[8];
                    unpack16<T>(pre_it, u);
#pragma unroll
                    for (int r = 0; r < 8; ++r) v[r] += u[r];
                } else if constexpr (!PLAIN) {
                    if (R != nullptr) {
                        float u[8];
                        unpack16<T>(*reinterpret_cast<const uint4*>(R + off), u);
#pragma unroll
                        for (int r = 0; r < 8; ++r) v[r] += u[r];
                    }
                }
                if constexpr (sizeof(TO) == 4) {
                    float* Cf = reinterpret_cast<float*>(C) + off;
                    if constexpr (!PLAIN) {
                        if (g.beta) {
                            const f32x4 c0v = *reinterpret_cast<const f32x4*>(Cf), c1v = *reinterpret_cast<const f32x4*>(Cf + 4);
#pragma unroll
                            for (int r = 0; r < 4; ++r) { v[r] += c0v[r]; v[4 + r] += c1v[r]; }
                        }
                    }
                    st_wt16(Cf, f32x4{v[0], v[1], v[2], v[3]});
                    st_wt16(Cf + 4, f32x4{v[4], v[5], v[6], v[7]});
                } else {
                    if constexpr (!PLAIN) {
                        if (g.beta) {
                            float u[8];
                            unpack16<T>(*reinterpret_cast<const uint4*>(C + off), u);
#pragma unroll
                            for (int r = 0; r < 8; ++r) v[r] += u[r];
                        }
                    }
                    const uint4 pk = pack16<T>(v);
                    // logits-sized outputs (>> the 256 MiB Infinity Cache) are written non-temporally so they do
                    // not push the operand panels out of L2 (asm: hipcc would merge a plain and a nontemporal store
                    // to one address into one plain store)
                    if (NT && g.nt_c) asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" :: "v"(C + off), "v"(__builtin_bit_cast(u32x4, pk)) : "memory");
                    else st_wt16(C + off, pk);                                // (write-through: common.h — the next kernel's boundary does not wait for this tile's lines)
                }
            };
            // The runtime-uniform switches (residual / beta / non-temporal) are lifted OUT of the unrolled body into
            // compile-time variants: as branches inside it they cut every 8-row step into its own basic block, so hipcc
            // could not batch the LDS reads and each step exposed an LDS round trip plus a 64-bit multiply for its address.
            auto shuffle = [&](auto plain_c, auto nt_flag, auto unit_c) {
                constexpr bool UNIT = decltype(unit_c)::value;              // alpha == 1 and no bias (the logits): the accumulators go to the patch as they are
                f32x4 bias4[4];
                if constexpr (!UNIT) { if (g.bias != nullptr) load_bias4(bias4); }
                // 256-row tile (round 6): ONE slot — the piece of pass p + 1 is requested into the register quad that `finish` has just consumed for
                // pass p (a pass of lead time all the same, 16 registers instead of 32: with two slots this instantiation needed 260)
                constexpr bool ONE_SLOT = PRE && WM == 8;
                uint4 pre[ONE_SLOT ? 1 : 2][PRE ? 4 : 1];
                auto prefetch = [&](int p, int slot) {
                    if constexpr (PRE) {
#pragma unroll
                        for (int it = 0; it < 4; ++it) pre[slot][it] = *reinterpret_cast<const uint4*>(side + off0 + (p * 4 + it) * row8);
                    }
                };
                prefetch(0, 0);
#pragma unroll
                for (int p = 0; p < WM / 2; ++p) {
#pragma unroll
                    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            f32x4 v = acc[p * 2 + ii][j];
                            if constexpr (!UNIT) {
                                v *= g.alpha;
                                if (g.bias != nullptr) v += bias4[j];
                            }
                            const int row = ii * 16 + (lane & 15), c = j * 4 + (lane >> 4);
                            *reinterpret_cast<f32x4*>(scr + row * 256 + ((c ^ (row & 15)) << 4)) = v;
                        }
                    __builtin_amdgcn_sched_barrier(0);                         // the next pass's inputs go into the registers these accumulators just freed
                    if (!ONE_SLOT && p + 1 < WM / 2) prefetch(p + 1, (p + 1) & 1);
                    constexpr int RB = (WM == 8) ? 2 : 4;                       // 8-row steps read back together (register budget)
#pragma unroll
                    for (int ib = 0; ib < 4; ib += RB) {
                    f32x4 lo[RB], hi[RB];
#pragma unroll
                    for (int k = 0; k < RB; ++k) {
                        const int row = (ib + k) * 8 + (lane >> 3), c0 = (lane & 7) * 2;
                        lo[k] = *reinterpret_cast<const f32x4*>(scr + row * 256 + ((c0 ^ (row & 15)) << 4));
                        hi[k] = *reinterpret_cast<const f32x4*>(scr + row * 256 + (((c0 ^ (row & 15)) ^ 1) << 4));
                    }
#pragma unroll
                    for (int k = 0; k < RB; ++k) {
                        const int it = ib + k;
                        float v[8] = {lo[k][0], lo[k][1], lo[k][2], lo[k][3], hi[k][0], hi[k][1], hi[k][2], hi[k][3]};
                        const int64_t off = off0 + (p * 4 + it) * row8;
                        finish(plain_c, nt_flag, v, off, pre[ONE_SLOT ? 0 : (p & 1)][PRE ? it : 0]);
                        if constexpr (ONE_SLOT) { if (p + 1 < WM / 2) pre[0][it] = *reinterpret_cast<const uint4*>(side + off0 + ((p + 1) * 4 + it) * row8); }
                    }
                    }
                    __builtin_amdgcn_sched_barrier(0);                         // one pass at a time: keeps the live set at acc + one pass
                }
            };
            // Cross-lane variant of the same re-layout: v_permlane16_swap exchanges the odd 16-lane rows of one register
            // with the even rows of another, so swapping the accumulators of column tiles j and j+1 element by element
            // leaves every lane with 8 consecutive columns of its row (lane group q: columns 32*jp + 16*(q&1) + 8*(q>>1))
            // — 16-byte bf16 stores / side-input loads, 64 contiguous bytes per row and instruction, with no LDS round
            // trip (4 swaps per 8 values instead of 2 ds_write_b128 + 2 ds_read_b128 and their waits).
            auto direct = [&](auto plain_c, auto nt_flag, auto bias_c) {
                constexpr bool BIAS = decltype(bias_c)::value;
                f32x4 bias4[4];
                if constexpr (BIAS) load_bias4(bias4);
                const int q = lane >> 4;
                const int64_t offL = (mw + (lane & 15)) * g.ldc + nw + 16 * (q & 1) + 8 * (q >> 1);
                const int64_t row16 = 16 * g.ldc;
                uint4 pre[2][PRE ? 4 : 1];
                auto prefetch = [&](int p, int slot) {
                    if constexpr (PRE) {
#pragma unroll
                        for (int it = 0; it < 4; ++it) pre[slot][it] = *reinterpret_cast<const uint4*>(side + offL + (p * 2 + (it >> 1)) * row16 + 32 * (it & 1));
                    }
                };
                prefetch(0, 0);
#pragma unroll
                for (int p = 0; p < WM / 2; ++p) {
                    if (p + 1 < WM / 2) prefetch(p + 1, (p + 1) & 1);
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const int i = p * 2 + (it >> 1), jp = it & 1;
                        f32x4 a = acc[i][2 * jp] * g.alpha, b = acc[i][2 * jp + 1] * g.alpha;
                        if constexpr (BIAS) { a += bias4[2 * jp]; b += bias4[2 * jp + 1]; }
                        // (inline asm, not __builtin_amdgcn_permlane16_swap: in the 256-row instantiations hipcc merged the four
                        // swaps of a register quad into one and replicated its result — caught by the forced-tile parity test.
                        // One statement per quad; the leading s_nop 1 is the "VALU write -> v_permlane read" hazard pad.)
                        float a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3], b0 = b[0], b1 = b[1], b2 = b[2], b3 = b[3];
                        asm volatile("s_nop 1\n\t"
                                     "v_permlane16_swap_b32 %0, %4\n\t"
                                     "v_permlane16_swap_b32 %1, %5\n\t"
                                     "v_permlane16_swap_b32 %2, %6\n\t"
                                     "v_permlane16_swap_b32 %3, %7"
                                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3));
                        float v[8] = {a0, a1, a2, a3, b0, b1, b2, b3};
                        const int64_t off = offL + i * row16 + 32 * jp;
                        finish(plain_c, nt_flag, v, off, pre[p & 1][PRE ? it : 0]);
                    }
                    __builtin_amdgcn_sched_barrier(0);                         // one pass at a time: keeps the live set at acc + one pass
                }
            };
            const bool plain = (PRE_RES || R == nullptr) && !g.beta;
            constexpr bool CAN_NT = sizeof(TO) == 2 && EPI == CTMI_EPI_NONE && !RES;
            if constexpr (USE_DIRECT) {
                static_assert(RES_IN_KERNEL == PRE_RES, "the plain test in front of the fast path must match the variant compiled here");
                if (g.bias != nullptr) direct(std::true_type{}, std::integral_constant<bool, CAN_NT>{}, std::true_type{});
                else direct(std::true_type{}, std::integral_constant<bool, CAN_NT>{}, std::false_type{});
            } else {
                if (plain && CAN_NT && g.alpha == 1.0f && g.bias == nullptr) shuffle(std::true_type{}, std::integral_constant<bool, CAN_NT>{}, std::true_type{});
                else if (plain) shuffle(std::true_type{}, std::integral_constant<bool, CAN_NT>{}, std::false_type{});
                else shuffle(std::false_type{}, std::false_type{}, std::false_type{});
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // patch reads retired before the other row group may write it
            }
            return;
        }
    }
    if (interior) {
#pragma unroll
        for (int i = 0; i < WM; ++i) {
            const int64_t m = m0 + wr * (WM * 16) + i * 16 + (lane & 15);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int64_t n = n0 + wc * 64 + j * 16 + (lane >> 4) * 4;
                const int64_t off = m * g.ldc + n;
                f32x4 v = acc[i][j] * g.alpha;
                if (g.bias != nullptr) v += *reinterpret_cast<const f32x4*>(g.bias + n);
                if (EPI == CTMI_EPI_GELU || EPI == CTMI_EPI_GELUG) {
                    float t[4] = {v[0], v[1], v[2], v[3]}, ax[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) { t[r] = Cvt<T>::to_f(Cvt<T>::from_f(t[r])); ax[r] = (EPI == CTMI_EPI_GELUG) ? gelu_tanh_grad_f(t[r]) : t[r]; }
                    store4<T>(AUXO + off, ax);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = gelu_tanh_f(t[r]);
                } else if (EPI == CTMI_EPI_RELU) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
                } else if (EPI == CTMI_EPI_DGELU || EPI == CTMI_EPI_DRELU || EPI == CTMI_EPI_MUL) {
                    float u[4];
                    load4<T>(AUXI + off, u);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = (EPI == CTMI_EPI_DGELU) ? v[r] * gelu_tanh_grad_f(u[r]) : (EPI == CTMI_EPI_MUL ? v[r] * u[r] : (u[r] > 0.f ? v[r] : 0.f));
                }
                if (R != nullptr) { float u[4]; load4<T>(R + off, u); v += f32x4{u[0], u[1], u[2], u[3]}; }
                if (g.beta) { float u[4]; load4<TO>(C + off, u); v += f32x4{u[0], u[1], u[2], u[3]}; }
                const float o[4] = {v[0], v[1], v[2], v[3]};
                store4<TO>(C + off, o);
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < WM; ++i) {
        const int64_t m = m0 + wr * (WM * 16) + i * 16 + (lane & 15);
        if (m >= g.M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t n = n0 + wc * 64 + j * 16 + (lane >> 4) * 4;
            if (n >= g.N) continue;
            const bool full = (n + 4 <= g.N) && g.vec_c;
            const int64_t off = m * g.ldc + n;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = g.alpha * acc[i][j][r];
            if (g.bias != nullptr) {
                if (full) { const float4 bb = *reinterpret_cast<const float4*>(g.bias + n); v[0] += bb.x; v[1] += bb.y; v[2] += bb.z; v[3] += bb.w; }
                else { for (int r = 0; r < 4; ++r) if (n + r < g.N) v[r] += g.bias[n + r]; }
            }
            if (EPI == CTMI_EPI_GELU || EPI == CTMI_EPI_GELUG) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = Cvt<T>::to_f(Cvt<T>::from_f(v[r]));
                float ax[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) ax[r] = (EPI == CTMI_EPI_GELUG) ? gelu_tanh_grad_f(v[r]) : v[r];
                if (full) store4<T>(AUXO + off, ax);
                else { for (int r = 0; r < 4; ++r) if (n + r < g.N) AUXO[off + r] = Cvt<T>::from_f(ax[r]); }
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = gelu_tanh_f(v[r]);
            } else if (EPI == CTMI_EPI_RELU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
            } else if (EPI == CTMI_EPI_DGELU || EPI == CTMI_EPI_DRELU || EPI == CTMI_EPI_MUL) {
                float u[4] = {0.f, 0.f, 0.f, 0.f};
                if (full) load4<T>(AUXI + off, u);
                else { for (int r = 0; r < 4; ++r) if (n + r < g.N) u[r] = Cvt<T>::to_f(AUXI[off + r]); }
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = (EPI == CTMI_EPI_DGELU) ? v[r] * gelu_tanh_grad_f(u[r]) : (EPI == CTMI_EPI_MUL ? v[r] * u[r] : (u[r] > 0.f ? v[r] : 0.f));
            }
            if (R != nullptr) {
                float u[4] = {0.f, 0.f, 0.f, 0.f};
                if (full) load4<T>(R + off, u);
                else { for (int r = 0; r < 4; ++r) if (n + r < g.N) u[r] = Cvt<T>::to_f(R[off + r]); }
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += u[r];
            }
            if (g.beta) {
                float u[4] = {0.f, 0.f, 0.f, 0.f};
                if (full) load4<TO>(C + off, u);
                else { for (int r = 0; r < 4; ++r) if (n + r < g.N) u[r] = Cvt<TO>::to_f(C[off + r]); }
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += u[r];
            }
            if (full) store4<TO>(C + off, v);
            else { for (int r = 0; r < 4; ++r) if (n + r < g.N) C[off + r] = Cvt<TO>::from_f(v[r]); }
        }
    }
    };

    setup_issue();
    int inflight = 0;                                                         // stages issued and not yet consumed
    int rd = 0, wrb = 0;                                                      // ring positions: read stage, next write stage
    if constexpr (PP) {
        // ---- ping-pong schedule (8 waves, one workgroup per CU).  Every SIMD holds one wave of row-group wr = 0 and one
        // of wr = 1; group 1 runs one barrier behind group 0, so while one group's 32 MFMAs own the matrix pipe the
        // other group reads its fragments from LDS and issues its LDS-DMA pieces.  Two barriers per K-step:
        //   phase A: ds_read fragments of stage c | DMA-issue stage c+NST-1 | wait own pieces of stage c+1 | lgkmcnt(0)
        //   phase B: MFMA
        // RAW: a stage is read one full K-step after every wave's counted wait for it (the lagging group's wait
        // precedes the barrier the leading group passes before reading).  WAR: stage c+NST-1 reuses the slot of stage c-1,
        // whose last reads (lagging group, phase A of c-1) were retired by lgkmcnt(0) before the barrier in between.
        // Wave priorities are static: the later-dispatched row group (waves 4-7) runs at priority 1 for the whole kernel, no per-phase flips.
        auto wait_stages = [&](int n) {                                       // allow n younger stages to stay in flight
            static_assert(LOADS == 3 || LOADS == 4, "counted waits are spelled out for 3 or 4 DMA instructions per stage");
            static_assert(NST <= 6, "wait_stages covers rings of up to 6 stages");
            if (n >= 4 && NST >= 6) { if (LOADS == 4) asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); }
            else if (n >= 3 && NST >= 5) { if (LOADS == 4) asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); }
            else if (n >= 2) { if (LOADS == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
            else if (n == 1) { if (LOADS == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); }
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        };
#pragma unroll 1
        for (int s = 0; s < FILL && wi < nwork; ++s) {
            issue_stage(wrb);
            stage_issued(); ++inflight; wrb = wrb == NST - 1 ? 0 : wrb + 1;
        }
        wait_stages(inflight - LAND);
        __builtin_amdgcn_s_barrier();
        if (wr == 1) __builtin_amdgcn_s_barrier();                            // stagger the two row groups by one phase
        if (wr == 1) __builtin_amdgcn_s_setprio(1);
        int cw = bid, tc = 0, ntc;
        int64_t m0, n0; int split;
        auto next_tile = [&]() {
            if constexpr (GRP) {
                const int4 it = g.items[cw];
                m0 = it.y; n0 = it.z; split = it.x; ntc = it.w >> 16;
                gs.cs_on = (it.x & WG_COLSUM) != 0;
                if constexpr (BSUM) gs.csb_on = (it.x & WG_COLSUMB) != 0 && wr == 0;      // (both row groups hold the same B columns: one of them sums)
            } else {
                decode(cw, m0, n0, split);
                ntc = (int)((min(g.K, (int64_t)(split + 1) * g.k_per_split) - (int64_t)split * g.k_per_split) / BK);
            }
        };
        next_tile();
        for (;;) {
            bool tile_done = false;
            // STEADY inner loops — as many K-steps as fit before either side (the MFMA side's tile, the DMA side's work item, FILL stages
            // ahead) reaches a boundary — with one trip counter and none of the generic step's per-step checks (more work? item switch? ring
            // fill?).  The generic step below stays for ring fill / drain.  (A loop around them since round 4: after the DMA stream's switch
            // to the next work item the remaining K-steps of THIS tile run steady too.)
            while (!tile_done && inflight == FILL && wi < nwork) {
                if constexpr (K2) {
                    // pairs of K-steps while both sides (this tile, the DMA stream's work item) have two left: stages rd, rd+1 are read, the pair
                    // rd+4, rd+5 goes into the two free slots, the counted wait leaves exactly that pair in flight (rd+2, rd+3 have landed for
                    // the next pair), 32 MFMAs: first all sixteen accumulators with stage rd, then again with rd+1 (no back-to-back dependency)
                    const int npair = min(ntc - tc, nti - ti) >> 1;
#pragma unroll 1
                    for (int n = npair; n > 0; --n) {
                        const int rd1 = rd == NST - 1 ? 0 : rd + 1;
                        short8 af0[WM], bf0[4], af1[WM], bf1[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) bf0[j] = TB::frag(smem_raw + BOFF + rd * TB::BYTES, wc * 64 + j * 16, lane);
#pragma unroll
                        for (int i = 0; i < WM; ++i) af0[i] = TA::frag(smem_raw + AOFF + rd * TA::BYTES, wr * (WM * 16) + arow(i), lane);
#pragma unroll
                        for (int j = 0; j < 4; ++j) bf1[j] = TB::frag(smem_raw + BOFF + rd1 * TB::BYTES, wc * 64 + j * 16, lane);
#pragma unroll
                        for (int i = 0; i < WM; ++i) af1[i] = TA::frag(smem_raw + AOFF + rd1 * TA::BYTES, wr * (WM * 16) + arow(i), lane);
                        issue_stage(wrb);
                        wrb = wrb == NST - 1 ? 0 : wrb + 1;
                        issue_stage(wrb);
                        wrb = wrb == NST - 1 ? 0 : wrb + 1;
                        wait_stages(2);
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_sched_barrier(0);
                        __builtin_amdgcn_s_barrier();
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int i = 0; i < WM; ++i)
#pragma unroll
                            for (int j = 0; j < 4; ++j) acc[i][j] = Mma<T>::mma(bf0[j], af0[i], acc[i][j]);
#pragma unroll
                        for (int i = 0; i < WM; ++i)
#pragma unroll
                            for (int j = 0; j < 4; ++j) acc[i][j] = Mma<T>::mma(bf1[j], af1[i], acc[i][j]);
                        if constexpr (GRP) {
                            if (gs.cs_on) {
                                gs.accs = Mma<T>::mma(WG_ONES8, af0[0], gs.accs);
                                gs.accs = Mma<T>::mma(WG_ONES8, af1[0], gs.accs);
                            }
                            if constexpr (BSUM) {
                                if (gs.csb_on) {
#pragma unroll
                                    for (int j = 0; j < 4; ++j) gs.accb[j] = Mma<T>::mma(bf0[j], WG_ONES8, gs.accb[j]);
#pragma unroll
                                    for (int j = 0; j < 4; ++j) gs.accb[j] = Mma<T>::mma(bf1[j], WG_ONES8, gs.accb[j]);
                                }
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        __builtin_amdgcn_s_barrier();
                        __builtin_amdgcn_sched_barrier(0);
                        rd = rd1 == NST - 1 ? 0 : rd1 + 1;
                    }
                    ti += 2 * npair;
                    tc += 2 * npair;
                    if (ti == nti) { wi += G; if (wi < nwork) setup_issue(); }
                    if (tc == ntc) { tile_done = true; break; }
                    if (npair > 0) continue;                                  // re-evaluate: the next run may again hold pairs
                }
                const int nsteady = min(ntc - tc, nti - ti);                  // >= 1 on both sides here
#pragma unroll 1
                for (int n = nsteady; n > 0; --n) {
                    const unsigned char* as = smem_raw + AOFF + rd * TA::BYTES;
                    const unsigned char* bs = smem_raw + BOFF + rd * TB::BYTES;
                    short8 af[WM], bf[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) bf[j] = TB::frag(bs, wc * 64 + j * 16, lane);
#pragma unroll
                    for (int i = 0; i < WM; ++i) af[i] = TA::frag(as, wr * (WM * 16) + arow(i), lane);
                    issue_stage(wrb);
                    wait_stages(FILL - LAND);                                 // (one stage consumed, one issued: FILL - LAND of the FILL younger ones may fly)
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < WM; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[i][j] = Mma<T>::mma(bf[j], af[i], acc[i][j]);
                    if constexpr (GRP) {
                        if (gs.cs_on) gs.accs = Mma<T>::mma(WG_ONES8, af[0], gs.accs);
                        if constexpr (BSUM) {
                            if (gs.csb_on) {
#pragma unroll
                                for (int j = 0; j < 4; ++j) gs.accb[j] = Mma<T>::mma(bf[j], WG_ONES8, gs.accb[j]);
                            }
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr ((NST & (NST - 1)) == 0) { rd = (rd + 1) & (NST - 1); wrb = (wrb + 1) & (NST - 1); }
                    else { rd = rd == NST - 1 ? 0 : rd + 1; wrb = wrb == NST - 1 ? 0 : wrb + 1; }
                }
                ti += nsteady;
                tc += nsteady;
                if (ti == nti) { wi += G; if (wi < nwork) setup_issue(); }
                if (tc == ntc) tile_done = true;
            }
            if (!tile_done) {
            // generic step: ring fill / drain
            const unsigned char* as = smem_raw + AOFF + rd * TA::BYTES;
            const unsigned char* bs = smem_raw + BOFF + rd * TB::BYTES;
            short8 af[WM], bf[4];
            const bool more = wi < nwork;
#pragma unroll
            for (int j = 0; j < 4; ++j) bf[j] = TB::frag(bs, wc * 64 + j * 16, lane);
#pragma unroll
            for (int i = 0; i < WM; ++i) af[i] = TA::frag(as, wr * (WM * 16) + arow(i), lane);
            if (more) {
                issue_stage(wrb);
                stage_issued(); ++inflight; wrb = wrb == NST - 1 ? 0 : wrb + 1;
            }
            wait_stages(inflight - 1 - LAND);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = Mma<T>::mma(bf[j], af[i], acc[i][j]);
            if constexpr (GRP) {
                if (gs.cs_on) gs.accs = Mma<T>::mma(WG_ONES8, af[0], gs.accs);
                if constexpr (BSUM) {
                    if (gs.csb_on) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) gs.accb[j] = Mma<T>::mma(bf[j], WG_ONES8, gs.accb[j]);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            rd = rd == NST - 1 ? 0 : rd + 1;
            --inflight;
            if (++tc < ntc) continue;
            }
            epilogue(m0, n0, split);
            cw += G;
            if (cw >= nwork) break;
            next_tile();
            tc = 0;
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (GRP) gs.accs = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (BSUM) {
#pragma unroll
                for (int j = 0; j < 4; ++j) gs.accb[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        if (wr == 0) __builtin_amdgcn_s_barrier();
        return;
    }
    // ---- free-running schedule (4 waves, 2-3 workgroups per CU): one barrier per K-step, the DMA issues of the stage two K-steps ahead spread
    // between the MFMAs
#pragma unroll 1
    for (int s = 0; s < 2 && wi < nwork; ++s) {
#pragma unroll
        for (int j = 0; j < LOADS; ++j) issue_one(wrb, j);
        stage_issued(); ++inflight; wrb = wrb == NST - 1 ? 0 : wrb + 1;
    }
    // ---- compute side
    int cw = bid, tc = 0, ntc;
    int64_t m0, n0; int split;
    decode(cw, m0, n0, split);
    ntc = (int)((min(g.K, (int64_t)(split + 1) * g.k_per_split) - (int64_t)split * g.k_per_split) / BK);
    constexpr int EVERY = (WM * 4) / LOADS;                                   // MFMAs between two DMA issues
    for (;;) {
        // My DMA of the stage about to be consumed has landed once at most the LOADS of the next stage are still
        // outstanding (loads retire in order; the epilogue's stores, which are older than that next stage only when
        // they have to be, are covered by the same count).
        if (inflight >= 2) { if (LOADS == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                      // everyone's stage landed; the previous one is fully consumed
        const bool more = wi < nwork;
        const unsigned char* as = smem_raw + AOFF + rd * TA::BYTES;
        const unsigned char* bs = smem_raw + BOFF + rd * TB::BYTES;
        short8 af[WM], bf[4];
#pragma unroll
        for (int i = 0; i < WM; ++i) af[i] = TA::frag(as, wr * (WM * 16) + arow(i), lane);
#pragma unroll
        for (int j = 0; j < 4; ++j) bf[j] = TB::frag(bs, wc * 64 + j * 16, lane);
        __builtin_amdgcn_s_setprio(1);                                        // MFMA phase outranks the partner wave's DMA / epilogue issue
        // the DMA issues of the stage two K-steps ahead are spread between the MFMAs (an LDS-DMA issue costs ~60-180
        // cycles of the wave's issue slot; back-to-back they would stall the matrix pipe for a whole K-step)
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[i][j] = Mma<T>::mma(bf[j], af[i], acc[i][j]);
                const int idx = i * 4 + j;
                if (idx % EVERY == EVERY - 1 && idx / EVERY < LOADS) {
                    if (more) issue_one(wrb, idx / EVERY);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        __builtin_amdgcn_s_setprio(0);
        rd = rd == NST - 1 ? 0 : rd + 1;
        --inflight;
        if (more) { stage_issued(); ++inflight; wrb = wrb == NST - 1 ? 0 : wrb + 1; }
        if (++tc < ntc) continue;
        epilogue(m0, n0, split);
        cw += G;
        if (cw >= nwork) break;
        decode(cw, m0, n0, split);
        ntc = (int)((min(g.K, (int64_t)(split + 1) * g.k_per_split) - (int64_t)split * g.k_per_split) / BK);
        tc = 0;
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
}

template <typename TO, bool AK, bool BKM, int EPI, int WM, int WGN, bool PP = false, bool RES = false, bool XLANE = false>
__global__ __launch_bounds__(128 * WGN, 2) void gemm_glds_kernel(GemmArgs g) {
    glds_body<TO, AK, BKM, EPI, WM, WGN, PP, RES, XLANE, false, GemmArgs>(g);
}
template <typename TO, bool AK, bool BKM, int EPI, int WM, int WGN, bool PP = false, bool RES = false, bool XLANE = false>
__global__ __launch_bounds__(128 * WGN, 2) void gemm_glds_kernel_f16(GemmArgs g) {
    glds_body<TO, AK, BKM, EPI, WM, WGN, PP, RES, XLANE, false, GemmArgs, f16_t>(g);
}
// the grouped weight-gradient launch (see GroupedArgs): the 128x256 ping-pong tile of gemm_glds_kernel<float, true, true, 0, 4, 4, true> walking a
// host-built work list over several problems
#if CTMI_GEMM_HAS(3) || CTMI_GEMM_PART == 8
__global__ __launch_bounds__(512, 2) void gemm_wgrad_grouped_kernel(GroupedArgs g) {
    glds_body<float, true, true, CTMI_EPI_NONE, 4, 4, true, false, false, true, GroupedArgs>(g);
}
__global__ __launch_bounds__(512, 2) void gemm_wgrad_grouped_kernel_f16(GroupedArgs g) {             // IEEE-half operands (the column sums then multiply by a fragment of half ones)
    glds_body<float, true, true, CTMI_EPI_NONE, 4, 4, true, false, false, true, GroupedArgs, f16_t>(g);
}
// (round 6) the same launch with the column sums of the B operand too: bias gradients of [in,out] (Conv1D) weights, whose dy is the B operand.  Its own
// instantiation: the kernels above — the measured Bloom path — keep their code
__global__ __launch_bounds__(512, 2) void gemm_wgrad_grouped_bsum_kernel(GroupedArgs g) {
    glds_body<float, true, true, CTMI_EPI_NONE, 4, 4, true, false, false, true, GroupedArgs, bf16_t, true>(g);
}
__global__ __launch_bounds__(512, 2) void gemm_wgrad_grouped_bsum_kernel_f16(GroupedArgs g) {
    glds_body<float, true, true, CTMI_EPI_NONE, 4, 4, true, false, false, true, GroupedArgs, f16_t, true>(g);
}
// second launch of a grouped call that cut tiles in two along K: C = partial 0 + partial 1 (and the column sums), one 32-row slice of a tile per workgroup
struct WgReduceArgs { GroupProb p[WG_MAXP]; const int4* tiles; int ntiles; };           // tiles: x = problem | WG_COLSUM, y = first row, z = first column
__global__ __launch_bounds__(256) void wgrad_partials_reduce_k(WgReduceArgs z);
// (round 6) the same sum and the block's partial-row reductions (LayerNorm affine gradients, bias column sums: csrc/reduce_jobs.h) in ONE launch:
// blocks [0, 4 * ntiles) are wgrad_partials_reduce_k's, the rest are reduce_jobs_k's — one dependent launch per block backward less
__device__ __forceinline__ void wgrad_partials_block(const WgReduceArgs& z, int block) {
    const int4 it = z.tiles[block >> 2];
    const GroupProb& P = z.p[it.x & (WG_MAXP - 1)];
    const int qr = block & 3, tid = threadIdx.x;
    const int64_t half = P.M * P.N;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int id = r * 256 + tid;                                                   // 32 rows x 64 float4
        const int64_t off = (int64_t)(it.y + qr * 32 + (id >> 6)) * P.ldc + it.z + (id & 63) * 4;
        const float4 a = *reinterpret_cast<const float4*>(P.part + off), b = *reinterpret_cast<const float4*>(P.part + half + off);
        *reinterpret_cast<float4*>(P.C + off) = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    }
    if ((it.x & WG_COLSUM) && qr == 0 && tid < 128) P.cs[it.y + tid] = P.cspart[it.y + tid] + P.cspart[P.csn + it.y + tid];
    if ((it.x & WG_COLSUMB) && qr == 0) P.cs[it.z + tid] = P.cspart[it.z + tid] + P.cspart[P.csn + it.z + tid];      // 256 columns of the tile, one per thread
}
__global__ __launch_bounds__(256) void wgrad_partials_reduce_k(WgReduceArgs z) { wgrad_partials_block(z, (int)blockIdx.x); }
__global__ __launch_bounds__(256) void wgrad_tail_k(WgReduceArgs z, ReduceJobs R) {
    __shared__ float sm[4][64];
    const int nb = 4 * z.ntiles;
    if ((int)blockIdx.x < nb) wgrad_partials_block(z, (int)blockIdx.x);
    else reduce_jobs_block(R, (int)blockIdx.x - nb, sm);
}
#endif

// C[m,n] = alpha * sum_s slabs[s][m][n] (+ C_old)   — deterministic split-K reduction
template <typename TO>
__global__ __launch_bounds__(256) void splitk_reduce(const float* __restrict__ slabs, TO* __restrict__ C, int64_t ldc,
                                                     int64_t M, int64_t N, int splits, float alpha, int beta) {
    const int64_t total = M * N;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        float s = 0.f;
        for (int k = 0; k < splits; ++k) s += slabs[(int64_t)k * total + i];
        const int64_t m = i / N, n = i - m * N;
        float v = alpha * s;
        if (beta) v += Cvt<TO>::to_f(C[m * ldc + n]);
        C[m * ldc + n] = Cvt<TO>::from_f(v);
    }
}

static bool shared_mode();
static bool shared_tiles();
static int reserved_cus();

template <typename TE, typename TO, bool AK, bool BKM, int EPI, int WM, int WGN, bool PP = false, bool RES = false, bool XLANE = false>
static void glds_launch(GemmArgs& g, hipStream_t st) {
    constexpr int BM = WM * 32, BN = WGN * 64;
    const size_t lds = glds_ring(PP, WM, XLANE) * (size_t)(GTile<AK, BM>::BYTES + GTile<BKM, BN>::BYTES) + glds_patch_bytes(PP, WM, XLANE);
    const int64_t nwork = cdiv64(g.M, BM) * cdiv64(g.N, BN) * g.splits;
    // persistent launch: one resident workgroup per occupancy slot (256 CUs x workgroups that fit a CU's 160 KiB LDS),
    // each walking work items bid, bid+G, ... with its DMA stream prefetching across item boundaries
    static int persist_env = -2;                                            // CTMI_GEMM_PERSIST overrides the policy (experiments)
    if (persist_env == -2) { const char* e = getenv("CTMI_GEMM_PERSIST"); persist_env = e ? atoi(e) : -1; }
    const int persist = persist_env >= 0 ? persist_env : (shared_mode() ? 0 : 1);
    // CTMI_GEMM_RESERVE_CUS = R leaves R of the 256 CUs out of every persistent launch.  A persistent GEMM owns each CU it
    // runs on until it ends (all LDS, all VGPRs), so with R = 0 a concurrent RCCL all-reduce kernel makes no progress for
    // the length of the GEMM (up to ~4 ms for the LM head); data-parallel runs set R ~ 16 so communication streams
    // continuously under backward (bench.py does for --gpus > 1).
    const int reserve = reserved_cus();
    const int64_t per_cu = (int64_t)std::min<size_t>((size_t)(WGN == 4 ? 1 : 8), (160 * 1024) / lds);
    const int64_t slots = (256 - reserve) / 8 * 8 * per_cu;                  // multiple of 8: the XCD-aware item order needs it
    // Long tiles are launched one workgroup per tile even under the persistent policy: the prefetch across tile boundaries is worth ~2 us per
    // tile, nothing against a 256-K-step tile, and workgroups that the dispatcher starts in order stay closer together than persistent ones
    // that drift over 15 tiles each — the four column tiles that share a dlogits panel of the LM-head weight gradient then find it in their
    // XCD's L2: FETCH 13.2 -> 8.6 GB per launch, 3.47 -> 3.45 ms (profiles/r05_lmhead_wgrad_launch.txt).  CTMI_GEMM_PERSIST_KSTEPS overrides (0: no limit).
    static int persist_ksteps = -1;
    if (persist_ksteps < 0) { const char* e = getenv("CTMI_GEMM_PERSIST_KSTEPS"); persist_ksteps = e ? atoi(e) : 128; }
    const bool long_tiles = persist_ksteps > 0 && g.k_per_split / 32 > persist_ksteps;
    const unsigned grid = (unsigned)((persist && !long_tiles && nwork > slots) ? slots : nwork);
    if constexpr (std::is_same<TE, f16_t>::value) {
        auto kern = &gemm_glds_kernel_f16<TO, AK, BKM, EPI, WM, WGN, PP, RES, XLANE>;
        ctmi_dyn_lds(reinterpret_cast<const void*>(kern), lds);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(128 * WGN), lds, st, g);
    } else {
        auto kern = &gemm_glds_kernel<TO, AK, BKM, EPI, WM, WGN, PP, RES, XLANE>;
        ctmi_dyn_lds(reinterpret_cast<const void*>(kern), lds);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(128 * WGN), lds, st, g);
    }
}

// tile / split choice for the LDS-DMA path:
//   0 = 128x128 (4 waves, 3 workgroups/CU)   1 = 256x128 (4 waves, 2/CU)   2 = 256x256 (8 waves, 1/CU, free-running)
//   3 = 256x256 ping-pong (8 waves, 1/CU)    4 = 128x256 ping-pong
// Rules distilled from tools/microbench.py sweeps of every Bloom-560M shape on MI355X (profiles/r01_gemm_tile_sweep.txt):
// the ping-pong schedule wins whenever its tiles fill the 256 CUs (>= ~1.4 rounds of 256x256, or one full round of
// 128x256 for [T,1024] outputs); row-major-B dgrads with long K and the layer weight gradients (both operands K-major,
// K = T, ~512+ workgroups via deterministic split-K) stay on the free-running tiles; the very-long-K LM-head dgrad
// (K = V) takes ping-pong 256x256 with a 2-way split.
// CTMI_GEMM_SHARED=1: the GPU is shared with somebody else's long-running kernels — in practice the RCCL all-reduce of a
// data-parallel job, which holds a few dozen CUs for milliseconds under backward.  Measured with tools/contention_probe.py
// (ONE CU held by a spin kernel): the default policy loses 23 % of the step, because a persistent launch sized to the CU
// count, or a ping-pong launch whose tiles fill the chip in exactly 1-2 rounds, waits a whole extra round for the CU it
// cannot get.  Shared mode therefore (a) never launches persistently (one workgroup per work item: the hardware dispatcher
// balances), (b) keeps the one-workgroup-per-CU ping-pong tiles for launches of >= 2048 items only (the LM head), with an
// 8-way split of the K = V dgrad, and (c) runs the layer GEMMs on the 128x128 / 256x128 tiles (2-3 workgroups per CU,
// which also co-reside with a small foreign workgroup).
// The policy is process state set through ctmi_set_launch_policy() (the data-parallel wrapper calls it whenever it is
// constructed or re-armed, so a model that already ran a GEMM before being wrapped still switches); the environment variables
// CTMI_GEMM_SHARED / CTMI_GEMM_RESERVE_CUS only provide the initial value.
#if CTMI_GEMM_HAS(0)
std::atomic<int> g_ctmi_policy_shared{-1}, g_ctmi_policy_reserve{-1};    // ONE copy for all translation units of this file
#else
extern std::atomic<int> g_ctmi_policy_shared, g_ctmi_policy_reserve;
#endif
// launch policy: 0 = persistent launches (single GPU), 1 = "shared": one workgroup per tile AND the co-resident free-running tiles of rounds 1-2
// for the layer GEMMs, 2 = "flow" (round 6): one workgroup per tile, but the SAME tile choice as the persistent policy (the ping-pong tiles) —
// the dispatcher flows the workgroups over whatever CUs the collectives' workgroups leave free, a tile that cannot start on a CU held by an
// RCCL channel starts on the next free one
static int shared_level() {
    int v = g_ctmi_policy_shared.load(std::memory_order_relaxed);
    if (v < 0) { const char* e = getenv("CTMI_GEMM_SHARED"); v = e ? std::max(0, std::min(2, atoi(e))) : 0; g_ctmi_policy_shared.store(v, std::memory_order_relaxed); }
    return v;
}
static bool shared_mode() { return shared_level() >= 1; }
static bool shared_tiles() { return shared_level() == 1; }
static int reserved_cus() {
    int v = g_ctmi_policy_reserve.load(std::memory_order_relaxed);
    if (v < 0) { const char* e = getenv("CTMI_GEMM_RESERVE_CUS"); v = e ? std::max(0, std::min(128, atoi(e))) : 0; g_ctmi_policy_reserve.store(v, std::memory_order_relaxed); }
    return v;
}
#if CTMI_GEMM_HAS(0)
extern "C" int ctmi_set_launch_policy(int shared, int reserve_cus) {
    CTMI_REQUIRE(reserve_cus >= 0 && reserve_cus <= 128, "set_launch_policy: reserve_cus must be in [0, 128]");
    g_ctmi_policy_shared.store(shared < 0 ? 0 : (shared > 2 ? 2 : shared), std::memory_order_relaxed);
    g_ctmi_policy_reserve.store(reserve_cus, std::memory_order_relaxed);
    return CTMI_OK;
}
extern "C" int ctmi_get_launch_policy(int* shared, int* reserve_cus) {
    if (shared) *shared = shared_level();
    if (reserve_cus) *reserve_cus = reserved_cus();
    return CTMI_OK;
}
#endif

constexpr int64_t WGRAD_ITEMS = 256;     // round-3 rule (CTMI_WGRAD_RULE=0): split-K target of the layer weight gradients, work items (tiles x splits) to aim for
// layer weight-gradient rule (CTMI_WGRAD_RULE): 0 = round 3 (128x128 / 256x128 free-running tiles, co-resident with the data gradients),
// 1 = 128x256 ping-pong unsplit, 2 (default since round 4) = the same with split-K up to CTMI_WGRAD_ITEMS4 (128) work items: at Bloom-560M the
// QKV gradient (96 tiles) splits two ways, the dense one (32 tiles) four ways, the two [4H,H] ones (128 tiles) stay whole
static int wgrad_rule() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("CTMI_WGRAD_RULE"); v = e ? std::max(0, atoi(e)) : 2; }
    return v;
}
static int64_t wgrad_items4() {
    static int64_t v = -1;
    if (v < 0) { const char* e = getenv("CTMI_WGRAD_ITEMS4"); v = e ? std::max(1, atoi(e)) : 128; }
    return v;
}
// smallest number of 256x256 output tiles for which a forward / data-gradient GEMM takes the 256-row ping-pong tile (below: 128x256 if
// that fills the chip).  350 since round 1 (384 = the QKV forward: 1.5 rounds of 256 CUs on tile 3, 3 full rounds of 768 on tile 4);
// CTMI_TILE3_MIN overrides it for sweeps — every rule of rounds 1-3 was measured with the K-loop drain of the cross-lane kernels in place.
static int64_t tile3_min() {
    static int64_t v = -1;
    if (v < 0) { const char* e = getenv("CTMI_TILE3_MIN"); v = e ? std::max(1, atoi(e)) : 350; }
    return v;
}
static void pick_tile(int64_t M, int64_t N, int64_t K, bool wgrad, bool bkm, int epi, int max_splits, int& tile, int& splits) {
    static int force = -2, force_split = -2;
    if (force == -2) { const char* e = getenv("CTMI_GEMM_TILE"); force = e ? atoi(e) : -1; }
    if (force_split == -2) { const char* e = getenv("CTMI_GEMM_SPLIT"); force_split = e ? atoi(e) : -1; }
    if (force_split >= 1) max_splits = std::min(max_splits, force_split);
    const int64_t t0 = cdiv64(M, 128) * cdiv64(N, 128), t1 = cdiv64(M, 256) * cdiv64(N, 128), t2 = cdiv64(M, 256) * cdiv64(N, 256);
    const int64_t t4 = cdiv64(M, 128) * cdiv64(N, 256);
    tile = 0; splits = 1;
    if (shared_tiles()) {
        if (wgrad) {
            if (t1 >= 1024) tile = t2 >= 2048 ? 3 : 1;
            else {
                tile = t1 >= 128 ? 1 : 0;
                const int64_t tiles = tile ? t1 : t0;
                while (splits < max_splits && tiles * splits < 512 && K / (splits * 2) >= 1024) splits *= 2;
            }
        } else if (K >= 32768 && max_splits >= 2) { tile = 3; splits = (int)std::min<int64_t>(max_splits, std::max<int64_t>(2, 1024 / std::max<int64_t>(t2, 1))); }
        else if (t2 >= 2048) tile = 3;
        else if (t1 >= 700) tile = 1;
        else tile = 0;
        if (force >= 0) tile = force;
        return;
    }
    if (wgrad) {
        static int nosplit = -1;
        if (nosplit < 0) { const char* e = getenv("CTMI_WGRAD_NOSPLIT"); nosplit = e ? atoi(e) : 0; }
        // LM head: [V,H] — and its row windows: the data-parallel path produces the tied gradient in <= 64 MiB pieces of 15 360 / 16 384
        // rows (trainer/ddp.py chunk_rows: whole rounds of the CUs the policy leaves), 240 / 256 tiles of 256x256; the layer weight
        // gradients of this geometry have <= 64 such tiles (round-3 advisor: a piece used to fall to the unsplit 128x128 rule below)
        if (t1 >= 1024 || (t2 >= 128 && M >= 8 * N)) tile = 3;                // (tall: a [rows, H] window, not a square layer gradient)
        else if (nosplit) tile = nosplit == 2 ? 4 : 3;
        else if (wgrad_rule() >= 1) {
            // round 4: with the K-loop drain of the 128-row ping-pong kernels fixed AND the side stream no longer joined after every block
            // (ctmi_bloom_block_grads.defer_join) the layer weight gradients run fastest in the step on the 128x256 ping-pong tile, unsplit:
            // 96 / 32 / 128 / 128 workgroups that take whole CUs next to the data-gradient chain instead of sharing them — 37.33-37.49 vs
            // 38.10-38.17 ms with the round-3 rule below (128x128 free-running, co-resident), same box, interleaved (profiles/r04_wgrad_rule.txt).
            // wgrad_rule() = 2 adds a split along K where that leaves fewer than CTMI_WGRAD_ITEMS4 work items: 37.30-37.38 vs 37.53-37.63 (128 items;
            // 256: 38.15-38.30, 64: 37.60-37.70).
            tile = 4;
            if (wgrad_rule() >= 2) { while (splits < max_splits && t4 * splits < wgrad_items4() && K / (splits * 2) >= 1024) splits *= 2; }
        }
        else {
            // round 3 (profiles/r03_gemm_tile_sweep.txt): a weight gradient with >= 256 tiles of 128x128 (h->4h and 4h->h) runs them
            // UNSPLIT (WGRAD_ITEMS = 256).  Alone on the GPU that is the slowest choice (602-621 TF/s, one workgroup per
            // CU; split two ways 778-805; the round-2 256x128 tiles split two ways 651-693) — in the training step it is the fastest:
            // 39.63 / 39.99 ms against 40.22 / 40.64 (split two ways) and 40.40 / 40.51 (round-2 rule), same box, interleaved.  The
            // weight gradients run on the side stream under the data-gradient chain: what counts there is how little they take from
            // the main stream (no fp32 slabs, no reduce launch, one co-resident workgroup per CU), not their own duration.
            const bool t0_fills = t0 >= 256;
            tile = t0_fills ? 0 : (t1 >= 128 ? 1 : 0);
            const int64_t tiles = tile ? t1 : t0;
            const int64_t items = WGRAD_ITEMS;
            while (splits < max_splits && tiles * splits < items && K / (splits * 2) >= 1024) splits *= 2;
        }
    } else if (K >= 32768 && max_splits >= 2 && t2 * 2 >= 192) { tile = 3; splits = 2; }
    else if (t2 >= tile3_min() || (t2 == 256 && K >= 2048)) tile = 3;             // (round 6: exactly ONE full round of 256x256 tiles with a long K — the [T,H] outputs of the Bloom-7B1 geometry, T = H = 4096:
                                                                                  // 204.2 -> 197.3 ms per step against two rounds of the 128-row tile, profiles/r06_bloom7b1_1gpu_bench.txt)
                                                                                  // (also where 256x256 tiles fill their last round badly — QKV forward: 384
                                                                                  // tiles = 1.5 rounds; onto 768 tiles of 128x256 it is +4 % alone and 0.1 ms WORSE in the step, round 4)
    else if (t4 >= 256) tile = 4;                                  // (round 3: also K-major B with long K — 919 vs 861 TF/s on the QKV data gradient)
    else if (t1 >= 700) tile = 1;
    else tile = 0;
    // epilogues that read a second [M,N] operand (activation-derivative input): rounds 3-5 only the 128-row ping-pong tile had the registers to
    // prefetch it a pass ahead (112 vs 121 us on the [T,4H] DGELU dgrad) and took these launches; round 6 the 256-row tile prefetches it through ONE
    // register slot (244 VGPRs, no scratch) and keeps them: 77 - 81 vs 88 - 90 us (profiles/r06_dgelu_side_input.txt)
    if (!CTMI_SIDE_PRE8 && (epi == CTMI_EPI_DGELU || epi == CTMI_EPI_DRELU || epi == CTMI_EPI_MUL) && tile == 3) tile = 4;
    if (force >= 0) tile = force;
}

static bool glds_enabled() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("CTMI_GEMM_GLDS"); v = (e && e[0] == '0') ? 0 : 1; }
    return v == 1;
}

template <typename T, typename TO, bool AK, bool BKM, int EPI>
static int gemm_launch(GemmArgs& g, bool fast, hipStream_t st) {
    using TA = OpTile<T, AK, Tile<T>::BM>;
    using TB = OpTile<T, BKM, Tile<T>::BN>;
    const size_t lds = 2 * (size_t)(TA::ELEMS + TB::ELEMS) * sizeof(T);
    const unsigned grid = (unsigned)(g.tiles_m * g.tiles_n * g.splits);
    if constexpr (std::is_same<T, bf16_t>::value || std::is_same<T, f16_t>::value) {   // the LDS-DMA family: bf16 and (round 5) fp16; fp32 takes the register-staged kernel below
        if (fast && glds_enabled() && g.K % 32 == 0) {
            int tile, splits;
            const int64_t slab = g.M * g.N * (int64_t)sizeof(float);
            const int max_sp = g.splitk_ok ? (int)std::max<int64_t>(1, std::min<int64_t>(8, g.ws_bytes / std::max<int64_t>(slab, 1))) : 1;
            pick_tile(g.M, g.N, g.K, AK && BKM, BKM, EPI, max_sp, tile, splits);
            g.splits = 1; g.k_per_split = g.K; g.slabs = nullptr;
            if (splits > 1) {
                const int64_t kps = cdiv64(cdiv64(g.K, splits), 64) * 64;
                const int sp = (int)cdiv64(g.K, kps);
                if (sp > 1) { g.splits = sp; g.k_per_split = kps; g.slabs = g.ws; }
            }
            constexpr bool CAN_RES = (EPI == CTMI_EPI_NONE) && !AK && sizeof(TO) == 2;      // residual-prefetching instantiations
            const bool res = CAN_RES && g.residual != nullptr;
            // forward-layout (row-major B) outputs that are not logits-sized: the 256-row tile with the cross-lane epilogue
            constexpr bool CAN_XLANE = !AK && !BKM && sizeof(TO) == 2 && (EPI == CTMI_EPI_NONE || EPI == CTMI_EPI_GELU);
            bool xl = false;
            if constexpr (CAN_XLANE) xl = tile == 3 && !g.nt_c && !res && g.vec8;
            if (xl) { if constexpr (CAN_XLANE) glds_launch<T, TO, AK, BKM, EPI, 8, 4, true, false, true>(g, st); }
            else if (tile == 3) { if constexpr (CAN_RES) { if (res) glds_launch<T, TO, AK, BKM, EPI, 8, 4, true, true>(g, st); else glds_launch<T, TO, AK, BKM, EPI, 8, 4, true>(g, st); }
                             else glds_launch<T, TO, AK, BKM, EPI, 8, 4, true>(g, st); }
            else if (tile == 4) { if constexpr (CAN_RES) { if (res) glds_launch<T, TO, AK, BKM, EPI, 4, 4, true, true>(g, st); else glds_launch<T, TO, AK, BKM, EPI, 4, 4, true>(g, st); }
                                  else glds_launch<T, TO, AK, BKM, EPI, 4, 4, true>(g, st); }
            else if (tile == 2) glds_launch<T, TO, AK, BKM, EPI, 8, 4>(g, st);
            else if (tile == 1) glds_launch<T, TO, AK, BKM, EPI, 8, 2>(g, st);
            else glds_launch<T, TO, AK, BKM, EPI, 4, 2>(g, st);
            CTMI_CHECK_LAUNCH("gemm_glds");
            if (g.splits > 1) {
                const int64_t total = g.M * g.N;
                const unsigned rg = (unsigned)std::min<int64_t>(cdiv64(total, 256), 4096);
                hipLaunchKernelGGL((splitk_reduce<TO>), dim3(rg), dim3(256), 0, st, g.slabs, reinterpret_cast<TO*>(g.C), g.ldc, g.M, g.N, g.splits, g.alpha, g.beta);
                CTMI_CHECK_LAUNCH("gemm_splitk_reduce");
            }
            return CTMI_OK;
        }
    }
    if (fast) {
        auto kern = &gemm_kernel<T, TO, AK, BKM, EPI, true>;
        ctmi_dyn_lds(reinterpret_cast<const void*>(kern), lds);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, g);
    } else {
        auto kern = &gemm_kernel<T, TO, AK, BKM, EPI, false>;
        ctmi_dyn_lds(reinterpret_cast<const void*>(kern), lds);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, g);
    }
    CTMI_CHECK_LAUNCH("gemm");
    if (g.splits > 1) {
        const int64_t total = g.M * g.N;
        const unsigned rg = (unsigned)std::min<int64_t>(cdiv64(total, 256), 4096);
        hipLaunchKernelGGL((splitk_reduce<TO>), dim3(rg), dim3(256), 0, st, g.slabs, reinterpret_cast<TO*>(g.C), g.ldc, g.M, g.N, g.splits, g.alpha, g.beta);
        CTMI_CHECK_LAUNCH("gemm_splitk_reduce");
    }
    return CTMI_OK;
}

// This file is compiled as SEVEN translation units (-DCTMI_GEMM_PART=0..6, see _build.py) so the kernel instantiations — three operand
// layouts x epilogues x five tile shapes x two 16-bit operand types — build in parallel:
//   part 0 = the C entry point (ctmi_gemm), the launch-policy state and fp32 (parity mode);
//   parts 1 / 2 / 3 = bf16 forward (NT) / data-gradient (NN) / weight-gradient (TN) families; part 3 ALSO owns the grouped weight-gradient
//     launch — ctmi_wgrad_grouped, its work-list cache, gemm_wgrad_grouped_kernel AND gemm_wgrad_grouped_kernel_f16, wgrad_partials_reduce_k
//     (everything under CTMI_GEMM_HAS(3), including the fp16 kernel: the grouped launch is one entry point for both operand types);
//   parts 4 / 5 / 6 = the fp16 (IEEE half) twins of parts 1 / 2 / 3's single-problem families (round 5).
// Undefined = everything in one unit.  (parts 8 / 9 exist only for tools/: one grouped kernel / one instantiation, for ISA inspection.)
int ctmi_gemm_bf16_nt(GemmArgs& g, int epi, bool fast, hipStream_t st);
int ctmi_gemm_bf16_nn(GemmArgs& g, int epi, bool fast, hipStream_t st);
int ctmi_gemm_bf16_tn(GemmArgs& g, int epi, bool fast, hipStream_t st);
int ctmi_gemm_f16_nt(GemmArgs& g, int epi, bool fast, hipStream_t st);
int ctmi_gemm_f16_nn(GemmArgs& g, int epi, bool fast, hipStream_t st);
int ctmi_gemm_f16_tn(GemmArgs& g, int epi, bool fast, hipStream_t st);
static int gemm_unsupported(int ak, int bk, int epi, int out_f32) {
    ctmi_set_error("gemm: unsupported combination a_kmajor=%d b_kmajor=%d epilogue=%d out_f32=%d", ak, bk, epi, out_f32);
    return CTMI_ERR_UNSUPPORTED;
}

#if CTMI_GEMM_HAS(1)
int ctmi_gemm_bf16_nt(GemmArgs& g, int epi, bool fast, hipStream_t st) {
    if (epi == CTMI_EPI_NONE) return gemm_launch<bf16_t, bf16_t, false, false, CTMI_EPI_NONE>(g, fast, st);
    if (epi == CTMI_EPI_GELU) return gemm_launch<bf16_t, bf16_t, false, false, CTMI_EPI_GELU>(g, fast, st);
    if (epi == CTMI_EPI_GELUG) return gemm_launch<bf16_t, bf16_t, false, false, CTMI_EPI_GELUG>(g, fast, st);
    if (epi == CTMI_EPI_RELU) return gemm_launch<bf16_t, bf16_t, false, false, CTMI_EPI_RELU>(g, fast, st);
    if (epi == CTMI_EPI_DGELU) return gemm_launch<bf16_t, bf16_t, false, false, CTMI_EPI_DGELU>(g, fast, st);   // data gradient through an [in,out] (Conv1D) weight
    return gemm_unsupported(0, 0, epi, 0);
}
#endif
#if CTMI_GEMM_HAS(2)
int ctmi_gemm_bf16_nn(GemmArgs& g, int epi, bool fast, hipStream_t st) {
    if (epi == CTMI_EPI_NONE) return gemm_launch<bf16_t, bf16_t, false, true, CTMI_EPI_NONE>(g, fast, st);
    if (epi == CTMI_EPI_DGELU) return gemm_launch<bf16_t, bf16_t, false, true, CTMI_EPI_DGELU>(g, fast, st);
    if (epi == CTMI_EPI_MUL) return gemm_launch<bf16_t, bf16_t, false, true, CTMI_EPI_MUL>(g, fast, st);
    if (epi == CTMI_EPI_DRELU) return gemm_launch<bf16_t, bf16_t, false, true, CTMI_EPI_DRELU>(g, fast, st);
    if (epi == CTMI_EPI_GELU) return gemm_launch<bf16_t, bf16_t, false, true, CTMI_EPI_GELU>(g, fast, st);     // forward through an [in,out] (Conv1D) weight
    return gemm_unsupported(0, 1, epi, 0);
}
#endif
#if CTMI_GEMM_HAS(3)
int ctmi_gemm_bf16_tn(GemmArgs& g, int epi, bool fast, hipStream_t st) {
    if (epi == CTMI_EPI_NONE) return gemm_launch<bf16_t, float, true, true, CTMI_EPI_NONE>(g, fast, st);
    return gemm_unsupported(1, 1, epi, 1);
}

// ---- grouped weight gradients: host side -----------------------------------------------------------------------------------------------------
#include <map>
#include <mutex>
#include <vector>
namespace {
struct WgShape { int64_t M[WG_MAXP], N[WG_MAXP]; int cs[WG_MAXP]; int n; int64_t ksteps; int slots; int split;      // cs: 0 none, 1 sums of the A operand, 2 of the B operand ([in,out] weight)
    bool operator<(const WgShape& o) const {
        if (n != o.n) return n < o.n;
        if (ksteps != o.ksteps) return ksteps < o.ksteps;
        if (slots != o.slots) return slots < o.slots;
        if (split != o.split) return split < o.split;
        for (int i = 0; i < n; ++i) { if (M[i] != o.M[i]) return M[i] < o.M[i]; if (N[i] != o.N[i]) return N[i] < o.N[i]; if (cs[i] != o.cs[i]) return cs[i] < o.cs[i]; }
        return false;
    }
};
struct WgTable { int4* dev = nullptr; int nitems = 0, nsplit = 0; unsigned split_mask = 0; };   // nsplit tiles (listed behind the items) are cut in two; split_mask: their problems
std::mutex g_wg_mu;
std::map<std::pair<int, WgShape>, WgTable> g_wg_tables;                                   // per (device, shape)

// XCD-aware position of workgroup b among L items of one round: workgroups b, b+8, ... run on one XCD and get a contiguous run of the list
inline int xcd_pos(int b, int L) {
    const int xcd = b & 7, q = L >> 3, r8 = L & 7;
    return (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (b >> 3);
}

// Work list.  Tiles of all problems in the grouped order of the single-problem kernel (groups of 8 tile rows x all tile columns, rows fastest:
// 32 consecutive tiles are a compact block that shares 8 A panels and 4 B panels in an XCD's L2).  Whole rounds of `slots` tiles run over all of K;
// the R tiles of the last, partial round are cut in two along K when both halves fit the round (2R <= slots) — `split` = 0 turns that off,
// 2 cuts EVERY tile (experiments).  `out` = the work items, then the list of cut tiles for the second launch.
void build_items(const WgShape& sh, std::vector<int4>& out, int& nitems, int& nsplit, unsigned& split_mask) {
    struct Tile { int p, m0, n0; };
    std::vector<Tile> tiles;
    for (int p = 0; p < sh.n; ++p) {
        const int tm = (int)(sh.M[p] / 128), tn = (int)(sh.N[p] / 256);
        for (int g0 = 0; g0 < tm; g0 += 8) {
            const int gm = std::min(8, tm - g0);
            for (int j = 0; j < tn; ++j)
                for (int i = 0; i < gm; ++i) tiles.push_back({p, (g0 + i) * 128, j * 256});
        }
    }
    const int nt = (int)tiles.size(), slots = sh.slots;
    const int ks = (int)sh.ksteps, h0 = ks / 2, h1 = ks - h0;
    int whole = nt;                                                                        // tiles [0, whole) run over all of K
    if (sh.split == 2 && ks >= 2) whole = 0;
    else if (sh.split == 1 && ks >= 2) { const int R = nt % slots; if (R > 0 && 2 * R <= slots) whole = nt - R; }
    split_mask = 0;
    auto csflag = [&](const Tile& t) { return sh.cs[t.p] == 1 ? (t.n0 == 0 ? WG_COLSUM : 0) : (sh.cs[t.p] == 2 ? (t.m0 == 0 ? WG_COLSUMB : 0) : 0); };
    auto item = [&](const Tile& t, int kb, int n, bool part) {
        const int cs = csflag(t);
        if (part) split_mask |= 1u << t.p;
        return make_int4(t.p | (part ? WG_PART : 0) | ((part && kb) ? WG_HALF1 : 0) | cs, t.m0, t.n0, kb | (n << 16));
    };
    std::vector<int4> lin;                                                                 // items in list order, before the XCD-aware placement
    for (int t = 0; t < whole; ++t) lin.push_back(item(tiles[t], 0, ks, false));
    // the halves: chunks of 16 tiles, first halves then second halves, so that the 32 items one XCD takes are 16 tiles x 2
    for (int t0 = whole; t0 < nt; t0 += 16) {
        const int t1 = std::min(nt, t0 + 16);
        for (int t = t0; t < t1; ++t) lin.push_back(item(tiles[t], 0, h0, true));
        for (int t = t0; t < t1; ++t) lin.push_back(item(tiles[t], h0, h1, true));
    }
    // placement: round r holds list entries [r*slots, ...); inside a round workgroup b takes the entry at its XCD-aware position
    const int total = (int)lin.size();
    out.assign(total, make_int4(0, 0, 0, 0));
    for (int r0 = 0; r0 < total; r0 += slots) {
        const int L = std::min(slots, total - r0);
        for (int b = 0; b < L; ++b) out[r0 + b] = lin[r0 + xcd_pos(b, L)];
    }
    nitems = total; nsplit = nt - whole;
    for (int t = whole; t < nt; ++t) out.push_back(make_int4(tiles[t].p | csflag(tiles[t]), tiles[t].m0, tiles[t].n0, 0));
}
}  // namespace

// CTMI_WGRAD_GROUP: 0 = off (four launches per block, split-K slabs: round 4), 1 (default) = grouped, the tiles of the last partial round cut in two
// along K (partials in the workspace + a small second launch), 2 = every tile cut in two (experiments), 3 = grouped, nothing cut
int ctmi_wgrad_group_mode() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("CTMI_WGRAD_GROUP"); v = e ? std::max(0, atoi(e)) : 1; }
    return v;
}

bool ctmi_wgrad_grouped_ok(const ctmi_wgrad_problem* pr, int n, int64_t T, int dtype) {
    if (ctmi_wgrad_group_mode() == 0 || !glds_enabled() || (dtype != CTMI_BF16 && dtype != CTMI_F16) || n < 1 || n > WG_MAXP || T < 64 || T % 32 != 0 || T / 32 > 0x7fff) return false;
    for (int i = 0; i < n; ++i) {
        const int64_t M = pr[i].in_out ? pr[i].n_in : pr[i].n_out, N = pr[i].in_out ? pr[i].n_out : pr[i].n_in;
        if (M <= 0 || N <= 0 || M % 128 != 0 || N % 256 != 0) return false;
        // gradients of >= 32 Mi elements fill the GPU with 256x256 tiles on their own (>= 512 of them), at a higher rate than this launch's
        // 128x256 tiles reach (Bloom-7B1 geometry, one GPU: 213.8 ms per step per-product vs 216.9 grouped, profiles/r05_wgrad_grouped.txt)
        if (M * N >= (32LL << 20)) return false;
        if (!pr[i].dy || !pr[i].x || !pr[i].dw) return false;
        if ((((uintptr_t)pr[i].dy) | ((uintptr_t)pr[i].x) | ((uintptr_t)pr[i].dw)) & 15) return false;
        if (pr[i].db && (((uintptr_t)pr[i].db) & 15)) return false;                        // (round 6: also with an [in,out] weight, where dy is the B operand)
    }
    return true;
}

// A grouped call made with defer_reduce leaves the sum of its K-halves PENDING (per host thread): ctmi_wgrad_tail() then launches it together with
// the caller's partial-row reductions (block.hip: one launch at the end of a block's backward instead of two).  The partial slabs live in the
// caller's workspace: nothing may overwrite it in between.
static thread_local struct { bool valid = false; WgReduceArgs z; hipStream_t st = nullptr; } g_wg_pending;
void ctmi_wgrad_pending_clear() { g_wg_pending.valid = false; }
bool ctmi_wgrad_pending() { return g_wg_pending.valid; }
int ctmi_wgrad_tail(const ctmi_reduce_job* jobs, int count, hipStream_t st) {
    if (!g_wg_pending.valid) return ctmi_reduce_jobs(jobs, count, st);
    g_wg_pending.valid = false;
    if (st != g_wg_pending.st || count > CTMI_REDUCE_MAX_JOBS || count <= 0) {           // not combinable: the two launches of rounds 5
        hipLaunchKernelGGL(wgrad_partials_reduce_k, dim3(4 * g_wg_pending.z.ntiles), dim3(256), 0, g_wg_pending.st, g_wg_pending.z);
        CTMI_CHECK_LAUNCH("wgrad_grouped_reduce");
        return ctmi_reduce_jobs(jobs, count, st);
    }
    ProfScope prof__(CTMI_PROF_REDUCE, st);
    ReduceJobs R;
    const int chunks = reduce_jobs_pack(jobs, count, R);
    if (chunks < 0) return CTMI_ERR_ARG;
    hipLaunchKernelGGL(wgrad_tail_k, dim3((unsigned)(4 * g_wg_pending.z.ntiles + chunks)), dim3(256), 0, st, g_wg_pending.z, R);
    CTMI_CHECK_LAUNCH("wgrad_tail");
    return CTMI_OK;
}
int ctmi_wgrad_grouped_ex(const ctmi_wgrad_problem* pr, int n, int64_t T, int dtype, void* workspace, int64_t workspace_bytes, void* stream, bool defer_reduce);
extern "C" int ctmi_wgrad_grouped(const ctmi_wgrad_problem* pr, int n, int64_t T, int dtype, void* workspace, int64_t workspace_bytes, void* stream) {
    return ctmi_wgrad_grouped_ex(pr, n, T, dtype, workspace, workspace_bytes, stream, false);
}
int ctmi_wgrad_grouped_ex(const ctmi_wgrad_problem* pr, int n, int64_t T, int dtype, void* workspace, int64_t workspace_bytes, void* stream, bool defer_reduce) {
    CTMI_REQUIRE(pr != nullptr, "wgrad_grouped: null problem list");
    if (!ctmi_wgrad_grouped_ok(pr, n, T, dtype)) {
        ctmi_set_error("wgrad_grouped: unsupported problem set (bf16 / fp16, <= %d problems, rows a multiple of 128 and columns of 256 of every gradient, T %% 32 == 0, "
                       "16-byte aligned operands and bias gradients; CTMI_WGRAD_GROUP != 0)", WG_MAXP);
        return CTMI_ERR_UNSUPPORTED;
    }
    hipStream_t st = as_stream(stream);
    ProfScope prof__(CTMI_PROF_GEMM_WGRAD, st);
    const int persist = shared_mode() ? 0 : 1;
    const int slots = (256 - reserved_cus()) / 8 * 8;
    WgShape sh = {};
    sh.n = n; sh.ksteps = T / 32; sh.slots = slots;
    const int mode = ctmi_wgrad_group_mode();
    sh.split = mode == 3 ? 0 : (mode == 2 ? 2 : 1);
    // the K-halves need 2 x (M N + M) floats per problem (the slabs are carved for every problem of the call, cut or not: the carve below must not
    // depend on which tiles the work list cuts): without that much workspace nothing is cut — the result is the same, the last partial round then
    // leaves half the CUs idle; said ONCE, because it changes the speed silently (round-5 advisor)
    {
        int64_t need = 0;
        for (int i = 0; i < n; ++i) { const int64_t M = pr[i].in_out ? pr[i].n_in : pr[i].n_out, N = pr[i].in_out ? pr[i].n_out : pr[i].n_in; need += 2 * (M * N + std::max(M, N)) * 4 + 512; }
        if (sh.split != 0 && (workspace == nullptr || workspace_bytes < need || (((uintptr_t)workspace) & 15))) {
            static std::atomic<bool> said{false};
            if (!said.exchange(true))
                fprintf(stderr, "[ctmi355] wgrad_grouped: workspace of %lld bytes (need %lld, 16-byte aligned) — the tiles of the last partial round are NOT cut "
                                "in K-halves (same results, lower speed)\n", (long long)workspace_bytes, (long long)need);
            sh.split = 0;
        }
    }
    GroupedArgs g = {};
    bool bsum = false;                                                                      // some problem wants the column sums of its B operand
    g.M = 128; g.N = 256; g.K = T; g.k_per_split = T; g.splits = 1; g.alpha = 1.0f;       // (the single-problem fields are not read by a grouped launch)
    for (int i = 0; i < n; ++i) {
        const bool io = pr[i].in_out != 0;
        GroupProb& P = g.p[i];
        P.A = io ? pr[i].x : pr[i].dy; P.B = io ? pr[i].dy : pr[i].x; P.C = pr[i].dw; P.cs = pr[i].db;
        P.M = io ? pr[i].n_in : pr[i].n_out; P.N = io ? pr[i].n_out : pr[i].n_in;
        P.lda = P.M; P.ldb = P.N; P.ldc = P.N;
        P.csn = io ? P.N : P.M;
        sh.M[i] = P.M; sh.N[i] = P.N; sh.cs[i] = P.cs == nullptr ? 0 : (io ? 2 : 1);
        bsum = bsum || (io && P.cs != nullptr);
    }
    int dev = 0;
    (void)hipGetDevice(&dev);
    WgTable tab;
    {
        std::lock_guard<std::mutex> lk(g_wg_mu);
        auto key = std::make_pair(dev, sh);
        auto it = g_wg_tables.find(key);
        if (it == g_wg_tables.end()) {
            if (g_wg_tables.size() >= 64) {
                // bounded cache (round-5 advisor): a process that walks through many geometries does not keep every work list for ever.  Lists may
                // be read by launches in flight, so the purge waits for the device first — a once-in-64-geometries event
                (void)hipDeviceSynchronize();
                for (auto& kv : g_wg_tables) (void)hipFree(kv.second.dev);
                g_wg_tables.clear();
            }
            std::vector<int4> items;
            WgTable t;
            build_items(sh, items, t.nitems, t.nsplit, t.split_mask);
            if (hipMalloc(&t.dev, items.size() * sizeof(int4)) != hipSuccess) { ctmi_set_error("wgrad_grouped: cannot allocate the work list"); return CTMI_ERR_LAUNCH; }
            // synchronous copy, once per geometry and device: the list is read by every later launch on any stream
            if (hipMemcpy(t.dev, items.data(), items.size() * sizeof(int4), hipMemcpyHostToDevice) != hipSuccess) { ctmi_set_error("wgrad_grouped: cannot upload the work list"); return CTMI_ERR_LAUNCH; }
            it = g_wg_tables.emplace(key, t).first;
        }
        tab = it->second;
    }
    g.items = tab.dev; g.nitems = tab.nitems;
    if (tab.nsplit) {                                                                     // partial slabs of the cut problems, carved out of the workspace
        char* w = reinterpret_cast<char*>(workspace);
        for (int i = 0; i < n; ++i) {
            g.p[i].part = reinterpret_cast<float*>(w); w += 2 * g.p[i].M * g.p[i].N * 4;
            g.p[i].cspart = reinterpret_cast<float*>(w); w += (2 * g.p[i].csn * 4 + 255) / 256 * 256;
        }
    }
    constexpr size_t lds = 6 * (size_t)(GTile<true, 128>::BYTES + GTile<true, 256>::BYTES);
    const unsigned grid = (unsigned)((persist && tab.nitems > slots) ? slots : tab.nitems);
    auto kern = bsum ? (dtype == CTMI_F16 ? &gemm_wgrad_grouped_bsum_kernel_f16 : &gemm_wgrad_grouped_bsum_kernel)
                     : (dtype == CTMI_F16 ? &gemm_wgrad_grouped_kernel_f16 : &gemm_wgrad_grouped_kernel);
    ctmi_dyn_lds(reinterpret_cast<const void*>(kern), lds);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, g);
    CTMI_CHECK_LAUNCH("wgrad_grouped");
    if (tab.nsplit) {
        WgReduceArgs z = {};
        for (int i = 0; i < n; ++i) z.p[i] = g.p[i];
        z.tiles = tab.dev + tab.nitems; z.ntiles = tab.nsplit;
        if (defer_reduce) { g_wg_pending.valid = true; g_wg_pending.z = z; g_wg_pending.st = st; return CTMI_OK; }
        hipLaunchKernelGGL(wgrad_partials_reduce_k, dim3(4 * tab.nsplit), dim3(256), 0, st, z);
        CTMI_CHECK_LAUNCH("wgrad_grouped_reduce");
    }
    return CTMI_OK;
}
#endif

#if CTMI_GEMM_HAS(4)
int ctmi_gemm_f16_nt(GemmArgs& g, int epi, bool fast, hipStream_t st) {
    using T = f16_t;
    if (epi == CTMI_EPI_NONE) return gemm_launch<T, T, false, false, CTMI_EPI_NONE>(g, fast, st);
    if (epi == CTMI_EPI_GELU) return gemm_launch<T, T, false, false, CTMI_EPI_GELU>(g, fast, st);
    if (epi == CTMI_EPI_GELUG) return gemm_launch<T, T, false, false, CTMI_EPI_GELUG>(g, fast, st);
    if (epi == CTMI_EPI_RELU) return gemm_launch<T, T, false, false, CTMI_EPI_RELU>(g, fast, st);
    if (epi == CTMI_EPI_DGELU) return gemm_launch<T, T, false, false, CTMI_EPI_DGELU>(g, fast, st);
    return gemm_unsupported(0, 0, epi, 0);
}
#endif
#if CTMI_GEMM_HAS(5)
int ctmi_gemm_f16_nn(GemmArgs& g, int epi, bool fast, hipStream_t st) {
    using T = f16_t;
    if (epi == CTMI_EPI_NONE) return gemm_launch<T, T, false, true, CTMI_EPI_NONE>(g, fast, st);
    if (epi == CTMI_EPI_DGELU) return gemm_launch<T, T, false, true, CTMI_EPI_DGELU>(g, fast, st);
    if (epi == CTMI_EPI_MUL) return gemm_launch<T, T, false, true, CTMI_EPI_MUL>(g, fast, st);
    if (epi == CTMI_EPI_DRELU) return gemm_launch<T, T, false, true, CTMI_EPI_DRELU>(g, fast, st);
    if (epi == CTMI_EPI_GELU) return gemm_launch<T, T, false, true, CTMI_EPI_GELU>(g, fast, st);
    return gemm_unsupported(0, 1, epi, 0);
}
#endif
#if CTMI_GEMM_HAS(6)
int ctmi_gemm_f16_tn(GemmArgs& g, int epi, bool fast, hipStream_t st) {
    if (epi == CTMI_EPI_NONE) return gemm_launch<f16_t, float, true, true, CTMI_EPI_NONE>(g, fast, st);
    return gemm_unsupported(1, 1, epi, 1);
}
#endif

#if CTMI_GEMM_HAS(0)
static int gemm_dispatch_bf16(GemmArgs& g, int ak, int bk, int epi, int out_f32, bool fast, hipStream_t st) {
    if (!ak && !bk && !out_f32) return ctmi_gemm_bf16_nt(g, epi, fast, st);
    if (!ak && bk && !out_f32) return ctmi_gemm_bf16_nn(g, epi, fast, st);
    if (ak && bk && out_f32) return ctmi_gemm_bf16_tn(g, epi, fast, st);
    return gemm_unsupported(ak, bk, epi, out_f32);
}

// fp16 storage (round 5): the same layouts, epilogues and tile families as bf16 (translation-unit parts 4 / 5 / 6)
static int gemm_dispatch_f16(GemmArgs& g, int ak, int bk, int epi, int out_f32, bool fast, hipStream_t st) {
    if (!ak && !bk && !out_f32) return ctmi_gemm_f16_nt(g, epi, fast, st);
    if (!ak && bk && !out_f32) return ctmi_gemm_f16_nn(g, epi, fast, st);
    if (ak && bk && out_f32) return ctmi_gemm_f16_tn(g, epi, fast, st);
    return gemm_unsupported(ak, bk, epi, out_f32);
}

static int gemm_dispatch_f32(GemmArgs& g, int ak, int bk, int epi, bool fast, hipStream_t st) {   // fp32 storage: output is fp32 either way
    using T = float;
    if (!ak && !bk) {
        if (epi == CTMI_EPI_NONE) return gemm_launch<T, float, false, false, CTMI_EPI_NONE>(g, fast, st);
        if (epi == CTMI_EPI_GELU) return gemm_launch<T, float, false, false, CTMI_EPI_GELU>(g, fast, st);
        if (epi == CTMI_EPI_GELUG) return gemm_launch<T, float, false, false, CTMI_EPI_GELUG>(g, fast, st);
        if (epi == CTMI_EPI_RELU) return gemm_launch<T, float, false, false, CTMI_EPI_RELU>(g, fast, st);
        if (epi == CTMI_EPI_DGELU) return gemm_launch<T, float, false, false, CTMI_EPI_DGELU>(g, fast, st);
    }
    if (!ak && bk) {
        if (epi == CTMI_EPI_NONE) return gemm_launch<T, float, false, true, CTMI_EPI_NONE>(g, fast, st);
        if (epi == CTMI_EPI_DGELU) return gemm_launch<T, float, false, true, CTMI_EPI_DGELU>(g, fast, st);
        if (epi == CTMI_EPI_MUL) return gemm_launch<T, float, false, true, CTMI_EPI_MUL>(g, fast, st);
        if (epi == CTMI_EPI_DRELU) return gemm_launch<T, float, false, true, CTMI_EPI_DRELU>(g, fast, st);
        if (epi == CTMI_EPI_GELU) return gemm_launch<T, float, false, true, CTMI_EPI_GELU>(g, fast, st);
    }
    if (ak && bk && epi == CTMI_EPI_NONE) return gemm_launch<T, float, true, true, CTMI_EPI_NONE>(g, fast, st);
    return gemm_unsupported(ak, bk, epi, 1);
}

extern "C" int ctmi_gemm(const void* A, int64_t lda, int a_kmajor, const void* B, int64_t ldb, int b_kmajor,
                         void* C, int64_t ldc, int64_t M, int64_t N, int64_t K,
                         float alpha, int beta, const float* bias, const void* residual, int epilogue,
                         const void* aux_in, void* aux_out, int out_f32, int dtype,
                         void* workspace, int64_t workspace_bytes, void* stream) {
    CTMI_REQUIRE(A && B && C, "gemm: null operand");
    CTMI_REQUIRE(M > 0 && N > 0 && K > 0, "gemm: bad shape M=%lld N=%lld K=%lld", (long long)M, (long long)N, (long long)K);
    CTMI_REQUIRE(lda >= (a_kmajor ? M : K) && ldb >= (b_kmajor ? N : K) && ldc >= N, "gemm: leading dimension too small");
    CTMI_REQUIRE((epilogue != CTMI_EPI_GELU && epilogue != CTMI_EPI_GELUG) || aux_out, "gemm: GELU epilogues need aux_out");
    CTMI_REQUIRE((epilogue != CTMI_EPI_DGELU && epilogue != CTMI_EPI_DRELU && epilogue != CTMI_EPI_MUL) || aux_in, "gemm: dGELU/dReLU/MUL epilogues need aux_in");
    CTMI_REQUIRE(dtype == CTMI_F32 || dtype == CTMI_BF16 || dtype == CTMI_F16, "gemm: unsupported dtype %d", dtype);
    // profile class: a vocabulary-sized dimension marks the tied head's three products; otherwise by operand layout
    ProfScope prof__((M >= 65536 || N >= 65536 || K >= 65536) ? CTMI_PROF_LM_HEAD
                     : (a_kmajor && b_kmajor ? CTMI_PROF_GEMM_WGRAD : ((!a_kmajor && b_kmajor) ? CTMI_PROF_GEMM_DGRAD : CTMI_PROF_GEMM_FWD)), as_stream(stream));
    const int es = dtype == CTMI_F32 ? 4 : 2;
    const int vec = 16 / es;
    const int bkt = dtype == CTMI_F32 ? Tile<float>::BK : Tile<bf16_t>::BK;
    GemmArgs g;
    g.A = A; g.B = B; g.C = C; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.M = M; g.N = N; g.K = K;
    g.alpha = alpha; g.beta = beta; g.bias = bias; g.residual = residual; g.aux_in = aux_in; g.aux_out = aux_out;
    g.tiles_m = (int)cdiv64(M, 128); g.tiles_n = (int)cdiv64(N, 128);
    const int64_t ntile = (int64_t)g.tiles_m * g.tiles_n;
    CTMI_REQUIRE(ntile < (1LL << 27), "gemm: too many tiles");
    const bool vec_a = (lda % vec == 0) && ((((uintptr_t)A) & 15) == 0) && ((a_kmajor ? M : K) % vec == 0);
    const bool vec_b = (ldb % vec == 0) && ((((uintptr_t)B) & 15) == 0) && ((b_kmajor ? N : K) % vec == 0);
    const bool fast = vec_a && vec_b;
    // C-side vector accesses are 4 elements wide (8 B bf16 / 16 B fp32)
    auto al = [](const void* p, int bytes) { return p == nullptr || ((((uintptr_t)p) & (bytes - 1)) == 0); };
    const int cbytes = (out_f32 || dtype == CTMI_F32) ? 16 : 8;
    g.vec_c = (ldc % 4 == 0) && al(C, cbytes) && al(residual, 4 * es) && al(aux_in, 4 * es) && al(aux_out, 4 * es) && al(bias, 16);
    static int nt_on = -1;
    if (nt_on < 0) { const char* e = getenv("CTMI_GEMM_NT"); nt_on = e ? atoi(e) : 1; }
    g.nt_c = nt_on && (M * N * (int64_t)((out_f32 || dtype == CTMI_F32) ? 4 : 2) > (1LL << 30)) ? 1 : 0;
    g.vec8 = g.vec_c && (ldc % 8 == 0) && al(C, 16) && al(residual, 16) && al(aux_in, 16) && al(aux_out, 16);   // 8-element rows (LDS-shuffled epilogue)
    // split-K: only for plain accumulations (weight gradients) that would leave most of the 256 CUs idle
    g.splits = 1; g.k_per_split = K; g.slabs = nullptr;
    g.splitk_ok = (workspace != nullptr && epilogue == CTMI_EPI_NONE && bias == nullptr && residual == nullptr) ? 1 : 0;
    g.ws = reinterpret_cast<float*>(workspace); g.ws_bytes = workspace_bytes;
    if (workspace != nullptr && epilogue == CTMI_EPI_NONE && bias == nullptr && residual == nullptr && ntile < 384 && K >= 8 * bkt) {
        int64_t want = std::min<int64_t>(8, cdiv64(512, ntile));
        want = std::min<int64_t>(want, K / (4 * bkt));
        want = std::min<int64_t>(want, workspace_bytes / (int64_t)(M * N * sizeof(float)));
        if (want > 1) {
            const int64_t kps = cdiv64(cdiv64(K, want), bkt) * bkt;
            g.splits = (int)cdiv64(K, kps); g.k_per_split = kps; g.slabs = reinterpret_cast<float*>(workspace);
            if (g.splits <= 1) { g.splits = 1; g.k_per_split = K; g.slabs = nullptr; }
        }
    }
    if (dtype == CTMI_F32) return gemm_dispatch_f32(g, a_kmajor, b_kmajor, epilogue, fast, as_stream(stream));
    if (dtype == CTMI_F16) return gemm_dispatch_f16(g, a_kmajor, b_kmajor, epilogue, out_f32, fast, as_stream(stream));
    return gemm_dispatch_bf16(g, a_kmajor, b_kmajor, epilogue, out_f32, fast, as_stream(stream));
}
#endif  // CTMI_GEMM_HAS(0)

// -DCTMI_GEMM_PART=9 (tools only): nothing but ONE explicit instantiation, chosen with -DCTMI_ONE_KERNEL="...": a seconds-long compile for
// reading the ISA of a single kernel (tools/kernel_isa_scan.py)
// -DCTMI_GEMM_PART=8 (tools only): the grouped weight-gradient kernel alone
#if CTMI_GEMM_PART == 9
#ifndef CTMI_ONE_KERNEL
#define CTMI_ONE_KERNEL bf16_t, false, true, 0, 4, 4, true, false, false
#endif
template __global__ void gemm_glds_kernel<CTMI_ONE_KERNEL>(GemmArgs);
#endif
