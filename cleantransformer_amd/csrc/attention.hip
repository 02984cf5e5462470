// Fused (flash-style) attention for gfx950: ALiBi + causal + key-padding mask (+ optional additive mask),
// forward and backward, never materialising the [S,S] score matrix.
//   reference: modeling_bloom.py:84-116 (BloomAttentionLayer core), transformer.py:40-57 (AttentionLayer core)
//
// Layout idea (all three kernels): a 256-thread workgroup handles 64 "own" rows (queries in fwd/dq, keys in
// dk/dv), 16 per wavefront, and streams the other sequence dimension through LDS in tiles of 64.  Every MFMA is
// issued so that the wave's own row index sits in the accumulator COLUMN (lane & 15): each lane then owns exactly
// one query (or key) row, the softmax statistics are per-lane scalars (2 shuffles per tile instead of 16), and the
// probabilities / dS values come out of the first MFMA already in operand layout for the second one — no LDS
// round trip for P.  The operand that must be contracted over the streamed dimension (V, K, Q, dO) is read from the SAME
// row-major LDS tile with the hardware transpose read (ds_read_b64_tr_b16): no transposed copy is staged anywhere.
// Masked scores take finfo(float).min like the reference's masked_fill, so all-masked rows become uniform.
#include "common.h"
#include "mma.h"
#include <type_traits>

// A/B knobs for tools/ variant builds (defaults = the adopted configuration)
#ifndef CTMI_ATTN_SADDR
#define CTMI_ATTN_SADDR 1        // FAST kernels: stream the K/V tiles through a scalar tile origin + constant per-lane offsets (same-box A/B: forward 281 -> 285-289 TF/s, backward 267 -> 272)
#endif
#ifndef CTMI_ATTN_SADDR_DKDV
#define CTMI_ATTN_SADDR_DKDV 0
#endif
#ifndef CTMI_ATTN_FASTBODY
#define CTMI_ATTN_FASTBODY 1     // backward kernels: mask-free loop body for tiles that cannot contain a masked score (A/B: slower while it costs a wave per SIMD: 207 VGPRs)
#endif
#ifndef CTMI_ATTN_SWAPRED
#define CTMI_ATTN_SWAPRED 1      // forward kernel: row-max across the 4 lane groups by v_permlane{16,32}_swap instead of ds_bpermute
#endif
#ifndef CTMI_ATTN_MINW
#define CTMI_ATTN_MINW 2         // __launch_bounds__ minimum waves per SIMD of the three streaming kernels (register cap = 512 / MINW)
#endif
#ifndef CTMI_ATTN_NBUF
#define CTMI_ATTN_NBUF 0         // 0: two LDS stages whenever they fit (one barrier per tile); 1: force one stage (more workgroups per CU)
#endif

#ifndef CTMI_ATTN_DBG_BUILD
#define CTMI_ATTN_DBG_BUILD 0    // 1: timing-ablation switches driven by CTMI_ATTN_DBG exist (A/B builds only); 0: compiled out
#endif
#if CTMI_ATTN_DBG_BUILD
#define ATTN_DBG(p) ((p).dbg)
#else
#define ATTN_DBG(p) 0
#endif
#include "attn_params.h"
#include "prof.h"

// 64 x HDP tile staging helpers (256 threads).  Thread -> (row = id / CPR, 16-byte chunk = id % CPR): the CPR lanes of
// one row read one contiguous run of HBM (coalesced: a head row of hd=64 bf16 is exactly one 128-byte line) and write one
// padded row-major LDS row (pitch HDP + 8 elements: ds_write_b128 / ds_read_b128 / ds_read_b64_tr_b16 all spread over banks).
// 16 bytes of tile data in flight: a first-class vector, NOT the uint4 struct — struct copies become memcpy intrinsics that
// SROA left in private memory (scratch stores + vmcnt(0) after every tile load) once the scalar gather path was compiled out
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// Loads through pointers that hipcc cannot prove to be global (loop-carried / selected pointers derived from the by-value
// kernel-argument struct) are emitted as flat_load, and a FLAT access counts on LGKM_CNT as well as VM_CNT: every
// `s_waitcnt lgkmcnt(0)` in front of an MFMA (placed there for the LDS fragment reads) then also waited for the tile prefetch
// issued just before it — the software pipeline was serialised on HBM/L2 latency in all three kernels.  These helpers cast to
// the global address space explicitly, so the loads are global_load_* and only the vmcnt wait at their use sees them.
template <typename V> __device__ __forceinline__ V ldg_as1(const void* p) {
    typedef const __attribute__((address_space(1))) V* gptr;
    return *(gptr)(p);
}
template <typename T, int HDP>
struct AT {
    static constexpr int VEC = 16 / sizeof(T);
    static constexpr int CPR = HDP / VEC;
    static constexpr int NCH = (64 * CPR) / 256;
    // row-major pitch.  bf16: HDP + 16 elements (160 B at hd = 64, 288 B at hd = 128): with the hardware's ds_read_b128 lane
    // groups {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31} the 16 row fragments of a group then fall on 16 distinct 16-byte
    // slots of the 256-byte bank row, and the 8 rows of a ds_read_b64_tr_b16 half-wave on 8 distinct 32-byte slots; the
    // previous HDP + 8 pitch (144 B) made 7 of 16 lanes 2-way conflict (SQ_LDS_BANK_CONFLICT = 34 % of SQ_LDS_IDX_ACTIVE).
    static constexpr int PRM = HDP + (sizeof(T) == 2 ? 16 : VEC);
    static constexpr int RM_ELEMS = 64 * PRM;
    static_assert((64 * CPR) % 256 == 0, "tile must split over 256 threads");

    // `fast` (wave-uniform: 16-byte aligned strides and hd == HDP): unconditional vector loads; rows beyond the
    // tensor are clamped to its last row — they carry finite data whose probabilities / dS are forced to zero by the
    // callers, so no predicate (and no exec-masked load + vmcnt(0) join) is needed in the steady state.
    static __device__ __forceinline__ void load(u32x4 (&regs)[NCH], const T* __restrict__ base, int64_t rs, int64_t row0,
                                                int64_t nrows, int hd, bool fast, int tid) {
        if (fast) {
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const int id = tid + 256 * i;
                const int64_t grow = min(row0 + id / CPR, nrows - 1);
                regs[i] = ldg_as1<u32x4>(base + grow * rs + (id % CPR) * VEC);
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int id = tid + 256 * i;
            const int row = id / CPR, c = (id % CPR) * VEC;
            const int64_t grow = row0 + row;
            const T* p = base + grow * rs + c;
            // assembled as packed 32-bit words: gathered element-wise into a T[VEC] array, hipcc kept one 16-bit value per
            // register for the tile data of BOTH paths and re-packed it with v_perm_b32 before every LDS store
            uint32_t wd[4] = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const T e = (grow < nrows && c + j < hd) ? p[j] : T{};
                if constexpr (sizeof(T) == 2) wd[j >> 1] |= (uint32_t)__builtin_bit_cast(uint16_t, e) << (16 * (j & 1));
                else wd[j] = __builtin_bit_cast(uint32_t, e);
            }
            regs[i] = u32x4{wd[0], wd[1], wd[2], wd[3]};
        }
    }
    // Streamed tiles: per-thread row pointers advance by one tile (64 rows) per call, so the steady state issues NCH plain
    // 16-byte loads and NCH 64-bit adds; the clamped / generic address computation of `load` (64-bit multiplies, min) runs
    // only for a tile that is not entirely inside the tensor.  (PMC: 43 % of the forward kernel's VALU instructions were
    // integer ops, most of them this per-tile address arithmetic.)
    static __device__ __forceinline__ void stream_init(const T* (&ptrs)[NCH], const T* __restrict__ base, int64_t rs, int64_t row0, int tid) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int id = tid + 256 * i;
            ptrs[i] = base + (row0 + id / CPR) * rs + (id % CPR) * VEC;
        }
    }
    static __device__ __forceinline__ void stream_load(u32x4 (&regs)[NCH], const T* (&ptrs)[NCH], const T* __restrict__ base, int64_t rs,
                                                       int64_t row0, int64_t nrows, int hd, bool fast, int tid) {
        if (fast) {
            // only the ADDRESSES differ between an interior tile and one that crosses the end of the tensor (rows clamped to
            // the last one): the loads themselves stay outside the branch.  (With the loads inside both arms hipcc merged
            // the arms through a private-memory copy of `regs` — scratch stores with a vmcnt(0) after every load.)
            const T* src[NCH];
#pragma unroll
            for (int i = 0; i < NCH; ++i) src[i] = ptrs[i];
            if (row0 + 64 > nrows) {
#pragma unroll
                for (int i = 0; i < NCH; ++i) {
                    const int id = tid + 256 * i;
                    src[i] = base + min(row0 + id / CPR, nrows - 1) * rs + (id % CPR) * VEC;
                }
            }
#pragma unroll
            for (int i = 0; i < NCH; ++i) regs[i] = ldg_as1<u32x4>(src[i]);
        } else {
            load(regs, base, rs, row0, nrows, hd, false, tid);
        }
#pragma unroll
        for (int i = 0; i < NCH; ++i) ptrs[i] += 64 * rs;
    }
    // Offset form of the same stream (CTMI_ATTN_SADDR): the tile origin is wave-uniform, so it lives in SGPRs (advanced by scalar
    // adds) and each lane keeps only a constant 32-bit byte offset — the loads become `global_load_dwordx4 v, v_off, s[base:base+1]`
    // and the NCH 64-bit VALU pointer bumps (+ the pointer copies hipcc made in front of every load) per tile and operand disappear.
    static __device__ __forceinline__ void stream_init_off(uint32_t (&offs)[NCH], int64_t rs, int tid) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int id = tid + 256 * i;
            offs[i] = (uint32_t)(((int64_t)(id / CPR) * rs + (id % CPR) * VEC) * (int64_t)sizeof(T));
        }
    }
    static __device__ __forceinline__ void stream_load_off(u32x4 (&regs)[NCH], const uint32_t (&offs)[NCH], const T* __restrict__ base, int64_t rs,
                                                           int64_t row0, int64_t nrows, int tid) {
        // base + row0 * rs: scalar arithmetic (all wave-uniform)
        const char* origin = reinterpret_cast<const char*>(base + row0 * rs);
        uint32_t o[NCH];
#pragma unroll
        for (int i = 0; i < NCH; ++i) o[i] = offs[i];
        if (row0 + 64 > nrows) {                                             // the one tile that crosses the end: rows clamped to the last one
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const int id = tid + 256 * i;
                const int64_t r = min((int64_t)(id / CPR), nrows - 1 - row0);
                o[i] = (uint32_t)((r * rs + (id % CPR) * VEC) * (int64_t)sizeof(T));
            }
        }
#pragma unroll
        for (int i = 0; i < NCH; ++i) regs[i] = ldg_as1<u32x4>(origin + o[i]);
    }
    static __device__ __forceinline__ void store_rm(const u32x4 (&regs)[NCH], T* __restrict__ tile, int tid) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int id = tid + 256 * i;
            *reinterpret_cast<u32x4*>(tile + (id / CPR) * PRM + (id % CPR) * VEC) = regs[i];
        }
    }
};

// fragment of a row-major [64][HDP] LDS tile: KL consecutive head-dim elements of `row`
template <typename T, int HDP> __device__ __forceinline__ typename Mma<T>::Frag frag_rm(const T* __restrict__ tile, int row, int kofs);
template <> __device__ __forceinline__ short8 frag_rm<bf16_t, 32>(const bf16_t* __restrict__ t, int row, int kofs) { return *reinterpret_cast<const short8*>(t + row * AT<bf16_t, 32>::PRM + kofs); }
template <> __device__ __forceinline__ short8 frag_rm<bf16_t, 64>(const bf16_t* __restrict__ t, int row, int kofs) { return *reinterpret_cast<const short8*>(t + row * AT<bf16_t, 64>::PRM + kofs); }
template <> __device__ __forceinline__ short8 frag_rm<bf16_t, 128>(const bf16_t* __restrict__ t, int row, int kofs) { return *reinterpret_cast<const short8*>(t + row * AT<bf16_t, 128>::PRM + kofs); }
template <> __device__ __forceinline__ short8 frag_rm<f16_t, 32>(const f16_t* __restrict__ t, int row, int kofs) { return *reinterpret_cast<const short8*>(t + row * AT<f16_t, 32>::PRM + kofs); }
template <> __device__ __forceinline__ short8 frag_rm<f16_t, 64>(const f16_t* __restrict__ t, int row, int kofs) { return *reinterpret_cast<const short8*>(t + row * AT<f16_t, 64>::PRM + kofs); }
template <> __device__ __forceinline__ short8 frag_rm<f16_t, 128>(const f16_t* __restrict__ t, int row, int kofs) { return *reinterpret_cast<const short8*>(t + row * AT<f16_t, 128>::PRM + kofs); }
template <> __device__ __forceinline__ float frag_rm<float, 32>(const float* __restrict__ t, int row, int kofs) { return t[row * AT<float, 32>::PRM + kofs]; }
template <> __device__ __forceinline__ float frag_rm<float, 64>(const float* __restrict__ t, int row, int kofs) { return t[row * AT<float, 64>::PRM + kofs]; }
template <> __device__ __forceinline__ float frag_rm<float, 128>(const float* __restrict__ t, int row, int kofs) { return t[row * AT<float, 128>::PRM + kofs]; }

// fragment straight from HBM: KL consecutive elements of one row (row == nullptr -> zeros)
// FAST: hd == HDP and 16-byte aligned rows (known at compile time) — only the vector load exists
template <typename T, bool FAST = false> __device__ __forceinline__ typename Mma<T>::Frag frag_global(const T* row, int kofs, int hd, bool vec_ok);
template <> __device__ __forceinline__ short8 frag_global<bf16_t, true>(const bf16_t* row, int kofs, int, bool) {
    short8 f = {0, 0, 0, 0, 0, 0, 0, 0};
    if (row != nullptr) f = ldg_as1<short8>(row + kofs);
    return f;
}
template <> __device__ __forceinline__ short8 frag_global<bf16_t, false>(const bf16_t* row, int kofs, int hd, bool vec_ok) {
    short8 f = {0, 0, 0, 0, 0, 0, 0, 0};
    if (row == nullptr) return f;
    if (kofs + 8 <= hd && vec_ok) return *reinterpret_cast<const short8*>(row + kofs);
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = (kofs + j < hd) ? (short)row[kofs + j] : (short)0;
    return f;
}
template <> __device__ __forceinline__ short8 frag_global<f16_t, true>(const f16_t* row, int kofs, int, bool) {
    short8 f = {0, 0, 0, 0, 0, 0, 0, 0};
    if (row != nullptr) f = ldg_as1<short8>(row + kofs);
    return f;
}
template <> __device__ __forceinline__ short8 frag_global<f16_t, false>(const f16_t* row, int kofs, int hd, bool vec_ok) {
    short8 f = {0, 0, 0, 0, 0, 0, 0, 0};
    if (row == nullptr) return f;
    if (kofs + 8 <= hd && vec_ok) return *reinterpret_cast<const short8*>(row + kofs);
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = (kofs + j < hd) ? (short)row[kofs + j].v : (short)0;
    return f;
}
template <> __device__ __forceinline__ float frag_global<float, false>(const float* row, int kofs, int hd, bool) {
    return (row != nullptr && kofs < hd) ? row[kofs] : 0.f;
}
template <> __device__ __forceinline__ float frag_global<float, true>(const float* row, int kofs, int, bool) {
    return row != nullptr ? row[kofs] : 0.f;
}

// x[nt][r] = sum_d tile[nt*16 + (g*4+r)][d] * own[d]   (tile rows stream, lane's own row from registers)
template <typename T, int HDP>
__device__ __forceinline__ void dot_tile(f32x4 (&x)[4], const T* __restrict__ rm_tile,
                                         const typename Mma<T>::Frag (&own)[HDP / Mma<T>::K], int lane) {
    constexpr int MK = Mma<T>::K, KL = Mma<T>::KL;
    const int g = lane >> 4, li = lane & 15;
    // k-step outer, key group inner: consecutive MFMAs go to four DIFFERENT accumulators, so the second k-step of a group
    // issues four matrix instructions after the first instead of right behind it (a dependent MFMA pair stalls for the whole
    // pipeline latency; hipcc kept the source order and padded it with s_nop)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) x[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < HDP / MK; ++kk)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
            x[nt] = Mma<T>::mma(frag_rm<T, HDP>(rm_tile, nt * 16 + li, kk * MK + g * KL), own[kk], x[nt]);
}

// acc[dt][r] += sum_{j in tile} tile[j][dt*16 + g*4 + r] * x(j)      with x in accumulator layout:
// x[nt][r] belongs to streamed row j = nt*16 + g*4 + r of the lane's own column.  `rm_tile` is the ROW-MAJOR
// [64][HDP] LDS image (the same one dot_tile reads): the operand that must be contracted over the streamed dimension
// (V, K, Q, dO) is fetched with ds_read_b64_tr_b16 — lane i of a 16-lane group addresses row R0 + (i>>2), columns
// C0 + 4*(i&3).. and receives column C0 + i for rows R0..R0+3 (probe-verified) — so no transposed copy is ever staged.
typedef short short4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint2 lds_tr_b16(const void* p) {
    typedef __attribute__((address_space(3))) short4_t lds_v4;
    return __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(unsigned)(size_t)p));
}
template <typename T, int HDP>
__device__ __forceinline__ void contract64(f32x4 (&acc)[HDP / 16], const T* __restrict__ rm_tile, const f32x4 (&x)[4], int lane) {
    constexpr int PRM = AT<T, HDP>::PRM;
    const int g = lane >> 4, li = lane & 15;
    if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const f32x4 lo = x[2 * ks], hi = x[2 * ks + 1];
            uint4 pk = make_uint4(pack2<T>(lo[0], lo[1]), pack2<T>(lo[2], lo[3]), pack2<T>(hi[0], hi[1]), pack2<T>(hi[2], hi[3]));
            const short8 b = __builtin_bit_cast(short8, pk);
            const T* base = rm_tile + (ks * 32 + g * 4 + (li >> 2)) * PRM + 4 * (li & 3);
#pragma unroll
            for (int dt = 0; dt < HDP / 16; ++dt) {
                const uint2 a0 = lds_tr_b16(base + dt * 16);                     // rows ks*32 + g*4 .. +3
                const uint2 a1 = lds_tr_b16(base + 16 * PRM + dt * 16);          // rows ks*32 + 16 + g*4 .. +3
                const short8 a = __builtin_bit_cast(short8, make_uint4(a0.x, a0.y, a1.x, a1.y));
                acc[dt] = Mma<T>::mma(a, b, acc[dt]);
            }
        }
    } else {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float b = x[nt][r];
#pragma unroll
                for (int dt = 0; dt < HDP / 16; ++dt)
                    acc[dt] = Mma<float>::mma(rm_tile[(nt * 16 + g * 4 + r) * PRM + dt * 16 + li], b, acc[dt]);
            }
    }
}

// XCD-aware block order.  The dispatcher places workgroup b on XCD b % 8 and each XCD has a private 4 MiB L2.  Give every XCD
// a contiguous run of logical block ids (bijective for any grid size) so that all the 64-row blocks of one (batch, head) —
// which stream the same K/V (or Q/dO) tiles — run on the same XCD and hit its L2 instead of re-fetching from HBM/MALL.
// (Ablation: with no steady-state global loads the forward kernel ran 1.6x faster; the loads were L2 misses.)
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int xcd = bid & 7, q = nblk >> 3, r8 = nblk & 7;
    return (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (bid >> 3);
}

// 3-input max in ONE instruction.  (fmaxf on MFMA results makes hipcc insert a canonicalising v_max x,x per operand.)
__device__ __forceinline__ float max3f(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// max over the 4 lane groups (lanes l, l^16, l^32, l^48) of a per-lane value, without the LDS crossbar: v_permlane16_swap
// exchanges the odd 16-lane rows of its first operand with the even rows of its second, v_permlane32_swap the upper 32 lanes of
// the first with the lower 32 of the second; applied to two copies of v, max(a, b) is the xor-16 / xor-32 butterfly step.
// (__shfl_xor compiles to ds_bpermute_b32: ~6 integer VALU instructions for the lane index — recomputed every tile — plus an
// LDS round trip and an lgkmcnt(0) in the middle of the softmax dependency chain, twice per tile.)
__device__ __forceinline__ float group_max4(float v) {
#if CTMI_ATTN_SWAPRED
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    float c = max3f(a, a, b), d = c;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(c), "+v"(d));
    return max3f(c, c, d);
#else
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    return fmaxf(v, __shfl_xor(v, 32, 64));
#endif
}

// sum over the 4 lane groups, same idiom
__device__ __forceinline__ float group_sum4(float v) {
#if CTMI_ATTN_SWAPRED
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    float c = a + b, d = c;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(c), "+v"(d));
    return c + d;
#else
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
#endif
}
// dot product of two operand fragments (the lane's KL consecutive head-dim elements of two rows)
template <typename T> __device__ __forceinline__ float frag_dot(short8 a, short8 b) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        T x, y;
        if constexpr (std::is_same<T, f16_t>::value) { x.v = (uint16_t)a[j]; y.v = (uint16_t)b[j]; }
        else { x = (T)a[j]; y = (T)b[j]; }
        s += Cvt<T>::to_f(x) * Cvt<T>::to_f(y);
    }
    return s;
}
template <typename T> __device__ __forceinline__ float frag_dot(float a, float b) { return a * b; }

// Per-key additive bias staged once per tile (one float per key):
//    slope * ALiBi position            for a key that may be attended            (modeling_bloom.py:328-330)
//    FINFO_MIN                          for a padding key (attention_mask == 0): fma(dot, scale, FINFO_MIN) == FINFO_MIN
//                                       exactly, i.e. the reference's masked_fill value, with no compare/select
//    -inf                               for a key index beyond Sk (does not exist: probability exactly 0)
__device__ __forceinline__ float key_bias(const AttnP& p, int64_t b, int64_t key, float slope) {
    if (key >= p.Sk) return -INFINITY;
    if (p.kvalid != nullptr && p.kvalid[b * p.Sk + key] == 0) return FINFO_MIN;
    return p.kpos != nullptr ? slope * p.kpos[b * p.Sk + key] : 0.0f;
}
// The same, split in two for the streaming loops: `load` issues the (unconditional, index-clamped) loads while the next
// tile is being requested, `value` turns them into the bias when the tile is published to LDS at the end of the iteration.
// (key_bias's early returns made dependent loads with a vmcnt(0) each, which also drained the tile prefetch issued just
// before — in the one wave all the others then wait for at the barrier.)
struct KeyBiasRaw {
    int valid; float pos;
    __device__ __forceinline__ void load(const AttnP& p, int64_t b, int64_t key) {
        const int64_t kc = b * p.Sk + min(key, p.Sk - 1);
        valid = p.kvalid != nullptr ? (int)p.kvalid[kc] : 1;
        pos = p.kpos != nullptr ? p.kpos[kc] : 0.0f;
    }
    __device__ __forceinline__ float value(const AttnP& p, int64_t key, float slope) const {
        const float r = valid != 0 ? slope * pos : FINFO_MIN;
        return key < p.Sk ? r : -INFINITY;
    }
};
// raw score; AM (additive mask, transformer.py:43-45) is a compile-time switch; am_base points at element (b,h,0,0)
template <bool AM>
__device__ __forceinline__ float score_raw(const AttnP& p, float dot, float kb, int q, int key, const float* __restrict__ am_base) {
    float s = fmaf(dot, p.scale, kb);
    if (AM) {
        const int qc = min(q, (int)p.Sq - 1), kc = min(key, (int)p.Sk - 1);
        s += am_base[(int64_t)qc * p.am_q + (int64_t)kc * p.am_k];
    }
    return s;
}

template <typename T, int HDP, bool FAST = false>
__device__ __forceinline__ void store_own_row(T* rowp, const f32x4 (&acc)[HDP / 16], float mul, int hd, int g, bool vec_ok) {
#pragma unroll
    for (int dt = 0; dt < HDP / 16; ++dt) {
        const int d = dt * 16 + g * 4;
        if (!FAST && d >= hd) continue;
        float v[4] = {acc[dt][0] * mul, acc[dt][1] * mul, acc[dt][2] * mul, acc[dt][3] * mul};
        if (FAST || (d + 4 <= hd && vec_ok)) {
            if constexpr (sizeof(T) == 2) *reinterpret_cast<uint2*>(rowp + d) = make_uint2(pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3]));
            else *reinterpret_cast<float4*>(rowp + d) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
            for (int r = 0; r < 4; ++r) if (d + r < hd) rowp[d + r] = Cvt<T>::from_f(v[r]);
        }
    }
}

// ------------------------------------------------------------------------------------------------ forward
// LDS stage sizes (bytes) and buffering depth: two stages (one barrier per tile) whenever they fit in 160 KiB with room
// for >= 2 workgroups per CU, else one stage (two barriers per tile).
template <typename T, int HDP> struct FwdStage { static constexpr int BYTES = 2 * AT<T, HDP>::RM_ELEMS * (int)sizeof(T) + 256; };
template <typename T, int HDP> struct DkdvStage { static constexpr int BYTES = 2 * AT<T, HDP>::RM_ELEMS * (int)sizeof(T) + 768; };
template <typename T, int HDP> struct DqStage { static constexpr int BYTES = 2 * AT<T, HDP>::RM_ELEMS * (int)sizeof(T) + 256; };
constexpr int nbuf_for(int stage_bytes) { return CTMI_ATTN_NBUF ? CTMI_ATTN_NBUF : (2 * stage_bytes <= 80 * 1024 ? 2 : 1); }

template <typename T, int HDP, bool AM, bool FAST, bool DROP = false>
__global__ __launch_bounds__(256, CTMI_ATTN_MINW) void attn_fwd_kernel(AttnP p) {
    using A = AT<T, HDP>;
    constexpr int MK = Mma<T>::K, KL = Mma<T>::KL, NKK = HDP / MK, NDT = HDP / 16;
    constexpr int STAGE = FwdStage<T, HDP>::BYTES, NBUF = nbuf_for(STAGE);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    int cur = 0;
    auto KS = [&](int s_) { return reinterpret_cast<T*>(smem_raw + s_ * STAGE); };
    auto VS = [&](int s_) { return KS(s_) + A::RM_ELEMS; };
    auto KB = [&](int s_) { return reinterpret_cast<float*>(VS(s_) + A::RM_ELEMS); };   // [64] per-key digest of the staged tile
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, g = lane >> 4, li = lane & 15;
    // FAST (compile time): 16-byte aligned strides and hd == HDP — the element-wise gather / guarded store paths are not even
    // compiled into the kernel the training shapes use (their mere presence cost registers, see AT::load)
    const bool fast = FAST || (p.vec_ok && p.hd == HDP);
    const int hd_ = FAST ? HDP : (int)p.hd;
    const bool vok = FAST || p.vec_ok;
    const int nqb = (int)((p.Sq + 63) / 64);
    const int vid = xcd_remap(blockIdx.x, gridDim.x);
    const int qb = nqb - 1 - (vid % nqb);                                   // longest (latest) query blocks first
    const int64_t bh = vid / nqb, h = bh % p.nh, b = bh / p.nh;
    const int64_t q0 = (int64_t)qb * 64, my_q = q0 + wid * 16 + li;
    const T* qp = reinterpret_cast<const T*>(p.q) + b * p.q_bs + h * p.q_hs;
    const T* kp = reinterpret_cast<const T*>(p.k) + b * p.k_bs + h * p.k_hs;
    const T* vp = reinterpret_cast<const T*>(p.v) + b * p.v_bs + h * p.v_hs;
    const float slope = p.slopes ? p.slopes[h] : 0.f;

    typename Mma<T>::Frag qf[NKK];
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk)
        qf[kk] = frag_global<T, FAST>(my_q < p.Sq ? qp + my_q * p.q_rs : nullptr, kk * MK + g * KL, hd_, vok);

    int64_t kv_end = p.Sk;
    if (p.causal) {
        kv_end = min(p.Sk, q0 + 63 + p.off + 1);
        if (kv_end < 1) kv_end = 1;
        // a query row whose whole causal window is padding sees only masked keys -> uniform over ALL keys
        if (p.kvalid != nullptr && p.first_valid[b] > q0 + p.off) kv_end = p.Sk;
    }
    const int ntiles = (int)((kv_end + 63) / 64);

    float m = -INFINITY, lsum = 0.f;
    f32x4 acc[NDT];
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) acc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};

    u32x4 rk[A::NCH], rv[A::NCH];
    float rkb = 0.f;
    KeyBiasRaw kbr{1, 0.f};
    A::load(rk, kp, p.k_rs, 0, p.Sk, hd_, fast, tid);
    A::load(rv, vp, p.v_rs, 0, p.Sk, hd_, fast, tid);
    const T* pk[A::NCH];
    const T* pv[A::NCH];
    uint32_t ok[A::NCH], ov[A::NCH];
    constexpr bool SADDR = CTMI_ATTN_SADDR && FAST;
    if constexpr (SADDR) { A::stream_init_off(ok, p.k_rs, tid); A::stream_init_off(ov, p.v_rs, tid); }
    else { A::stream_init(pk, kp, p.k_rs, 64, tid); A::stream_init(pv, vp, p.v_rs, 64, tid); }
    if (tid < 64) rkb = key_bias(p, b, tid, slope);
    A::store_rm(rk, KS(0), tid);
    A::store_rm(rv, VS(0), tid);
    if (tid < 64) KB(0)[tid] = rkb;
    __syncthreads();
    const int q_eff = my_q < p.Sq ? (int)my_q : 0;
    const float* am_base = AM ? p.add_mask + b * p.am_b + h * p.am_h : nullptr;

    for (int t = 0; t < ntiles; ++t) {
        if (t + 1 < ntiles && !(ATTN_DBG(p) & 1)) {
            if (tid < 64) kbr.load(p, b, (int64_t)(t + 1) * 64 + tid);
            if constexpr (SADDR) {
                A::stream_load_off(rk, ok, kp, p.k_rs, (int64_t)(t + 1) * 64, p.Sk, tid);
                A::stream_load_off(rv, ov, vp, p.v_rs, (int64_t)(t + 1) * 64, p.Sk, tid);
            } else {
                A::stream_load(rk, pk, kp, p.k_rs, (int64_t)(t + 1) * 64, p.Sk, hd_, fast, tid);
                A::stream_load(rv, pv, vp, p.v_rs, (int64_t)(t + 1) * 64, p.Sk, hd_, fast, tid);
            }
        }
        f32x4 x[4];
        const float* kbs = KB(cur);
        dot_tile<T, HDP>(x, KS(cur), qf, lane);                              // x[nt][r] = q . k[key]
        const int kv0 = t * 64;
        const bool diag = p.causal && (kv0 + 63 > (int)q0 + p.off);          // only tiles crossing the diagonal need the causal test
        // softmax arithmetic on 4-wide vectors so that hipcc emits packed fp32 ops (v_pk_fma/add/mul_f32): this kernel
        // is bound by VALU issue (hd = 64 gives only 16 MFMAs per ~300 VALU instructions per tile), not by the matrix pipe
        float mx = -INFINITY;
        // three passes over the 4 key groups instead of one: with the (tile-uniform) causal branch inside the group loop every
        // group was its own basic block, so the four key-bias LDS reads could not be batched and each exposed its latency
        f32x4 kb4[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) kb4[nt] = *reinterpret_cast<const f32x4*>(kbs + nt * 16 + g * 4);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            f32x4 s4 = x[nt] * p.scale + kb4[nt];                             // fma(dot, scale, key bias): padding -> finfo.min, no key -> -inf
            if (AM) {
#pragma unroll
                for (int r = 0; r < 4; ++r) s4[r] = score_raw<true>(p, x[nt][r], kb4[nt][r], q_eff, kv0 + nt * 16 + g * 4 + r, am_base);
            }
            x[nt] = s4;
        }
        if (diag) {
            const int thr = q_eff + (int)p.off - kv0 - g * 4;                 // key offset c = nt*16 + r is in the causal future iff c > thr
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    x[nt][r] = (nt * 16 + r > thr) ? (kb4[nt][r] > FINFO_MIN ? p.future_fill : kb4[nt][r]) : x[nt][r];   // future: fill (padding stays finfo.min, no key stays -inf)
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) mx = max3f(max3f(mx, x[nt][0], x[nt][1]), x[nt][2], x[nt][3]);
        mx = group_max4(mx);
        const float m_new = fmaxf(m, mx);                                   // finite: every tile holds >= 1 real key
        const float alpha = __expf(m - m_new);
        f32x4 rs4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const f32x4 e4 = (x[nt] - m_new) * 1.4426950408889634f;           // (s - m) first: finfo.min - finfo.min must be 0, not inf - inf
            f32x4 p4;
#pragma unroll
            for (int r = 0; r < 4; ++r) p4[r] = __builtin_amdgcn_exp2f(e4[r]);
            x[nt] = p4;
            rs4 += p4;
        }
        const float rs = (rs4[0] + rs4[1]) + (rs4[2] + rs4[3]);
        lsum = lsum * alpha + rs;
        if constexpr (DROP) {
            // torch.nn.Dropout on the NORMALISED probabilities (modeling_bloom.py:111, modeling_gpt.py:96, transformer.py:46-47): the
            // row sum above stays undropped, the values that meet V are masked and scaled by 1/(1-p)
            const uint32_t c0 = (uint32_t)(((b * p.nh + h) * p.Sq + q_eff) * p.Sk + kv0 + g * 4);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    x[nt][r] = ctmi_keep_hash(c0 + (uint32_t)(nt * 16 + r), p.drop_seed) >= p.drop_thr ? x[nt][r] * p.drop_scale : 0.f;
        }
        if (__any(m_new > m)) {                                              // wave-uniform: rescale only when some row max moved
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) acc[dt] *= alpha;
        }
        m = m_new;
        contract64<T, HDP>(acc, VS(cur), x, lane);                          // acc[dt][r] = O^T[d][my_q]
        if (NBUF == 1) __syncthreads();
        if (t + 1 < ntiles && !(ATTN_DBG(p) & 2)) {
            const int nx = NBUF == 2 ? cur ^ 1 : 0;
            A::store_rm(rk, KS(nx), tid);
            A::store_rm(rv, VS(nx), tid);
            if (tid < 64) KB(nx)[tid] = kbr.value(p, (int64_t)(t + 1) * 64 + tid, slope);
            cur = nx;
        }
        if (!(ATTN_DBG(p) & 4)) __syncthreads();
    }
    lsum += __shfl_xor(lsum, 16, 64);
    lsum += __shfl_xor(lsum, 32, 64);
    if (my_q < p.Sq) {
        T* op = reinterpret_cast<T*>(p.out) + b * p.o_bs + h * p.o_hs + my_q * p.o_rs;
        store_own_row<T, HDP, FAST>(op, acc, 1.0f / lsum, hd_, g, vok);
        if (g == 0) {
            p.stat_m[(b * p.nh + h) * p.Sq + my_q] = m;
            p.stat_l[(b * p.nh + h) * p.Sq + my_q] = lsum;
        }
    }
}

// ------------------------------------------------------------------------------------------------ backward: dK, dV
// own rows = keys.  Per query tile: S = Q K^T and dP = dO V^T (own key in the accumulator column), P = exp(S-m)/l,
// dS = P (dP - delta); dV^T += dO^T P, dK^T += Q^T dS  (Q, dO staged both row-major and transposed).
template <typename T, int HDP, bool AM, bool FAST, bool DROP = false>
__global__ __launch_bounds__(256, CTMI_ATTN_MINW) void attn_bwd_dkdv_kernel(AttnP p) {
    using A = AT<T, HDP>;
    constexpr int MK = Mma<T>::K, KL = Mma<T>::KL, NKK = HDP / MK, NDT = HDP / 16;
    constexpr int STAGE = DkdvStage<T, HDP>::BYTES, NBUF = nbuf_for(STAGE);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    int cur = 0;
    auto QS = [&](int s_) { return reinterpret_cast<T*>(smem_raw + s_ * STAGE); };
    auto GS = [&](int s_) { return QS(s_) + A::RM_ELEMS; };
    auto ST = [&](int s_) { return reinterpret_cast<float*>(GS(s_) + A::RM_ELEMS); };   // [3][64]: m, 1/l, delta
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, g = lane >> 4, li = lane & 15;
    // FAST (compile time): 16-byte aligned strides and hd == HDP — the element-wise gather / guarded store paths are not even
    // compiled into the kernel the training shapes use (their mere presence cost registers, see AT::load)
    const bool fast = FAST || (p.vec_ok && p.hd == HDP);
    const int hd_ = FAST ? HDP : (int)p.hd;
    const bool vok = FAST || p.vec_ok;
    const int nkb = (int)((p.Sk + 63) / 64);
    const int vid = xcd_remap(blockIdx.x, gridDim.x);
    const int kb = vid % nkb;                                                // early key blocks (most work) first
    const int64_t bh = vid / nkb, h = bh % p.nh, b = bh / p.nh;
    const int64_t k0 = (int64_t)kb * 64, my_k = k0 + wid * 16 + li;
    const T* qp = reinterpret_cast<const T*>(p.q) + b * p.q_bs + h * p.q_hs;
    const T* kp = reinterpret_cast<const T*>(p.k) + b * p.k_bs + h * p.k_hs;
    const T* vp = reinterpret_cast<const T*>(p.v) + b * p.v_bs + h * p.v_hs;
    const T* gp = reinterpret_cast<const T*>(p.d_o) + b * p.o_bs + h * p.o_hs;
    const float slope = p.slopes ? p.slopes[h] : 0.f;

    typename Mma<T>::Frag kf[NKK], vf[NKK];
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) {
        kf[kk] = frag_global<T, FAST>(my_k < p.Sk ? kp + my_k * p.k_rs : nullptr, kk * MK + g * KL, hd_, vok);
        vf[kk] = frag_global<T, FAST>(my_k < p.Sk ? vp + my_k * p.v_rs : nullptr, kk * MK + g * KL, hd_, vok);
    }
    f32x4 dk[NDT], dv[NDT];
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) { dk[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    const float my_kb = key_bias(p, b, my_k, slope);                         // -inf: key row does not exist
    const bool key_live = my_kb > -INFINITY;
    const bool key_pad = my_kb <= FINFO_MIN;
    const float lane_fill = key_pad ? FINFO_MIN : p.future_fill;            // what a masked score of this lane's key is replaced by
    const uint32_t drop_c0 = (uint32_t)((b * p.nh + h) * p.Sq * p.Sk + my_k);                  // dropout counter of (q, my_k) = drop_c0 + q * Sk
    const float* am_base = AM ? p.add_mask + b * p.am_b + h * p.am_h : nullptr;

    int qt_begin = 0;
    if (p.causal) {
        const int64_t first_q = k0 - p.off;                                  // rows q >= first_q can see key k0
        qt_begin = first_q > 0 ? (int)(first_q / 64) : 0;
        // all-masked query rows (q + off < first_valid) are uniform over ALL keys -> they reach every key block
        if (p.kvalid != nullptr && p.first_valid[b] - p.off > 0) qt_begin = 0;
    }
    const int qt_end = (int)((p.Sq + 63) / 64);
    const float* sm = p.stat_m + (b * p.nh + h) * p.Sq;
    const float* sl = p.stat_l + (b * p.nh + h) * p.Sq;
    const float* sd = p.delta + (b * p.nh + h) * p.Sq;

    u32x4 rq[A::NCH], rg[A::NCH];
    float rstat = 0.f;
    // raw load now (index clamped, no arithmetic on the value: nothing forces a wait next to the tile prefetch), the
    // reciprocal / zeroing when the stats are published to LDS
    const float* sbase = (tid >> 6) == 0 ? sm : ((tid >> 6) == 1 ? sl : sd);
    auto load_stats = [&](int t) {
        const int64_t q = (int64_t)t * 64 + (tid & 63);
        if (tid < 192) rstat = ldg_as1<float>(sbase + min(q, p.Sq - 1));
    };
    auto stat_value = [&](int t) {
        const int64_t q = (int64_t)t * 64 + (tid & 63);
        const float v = (tid >> 6) == 1 ? 1.0f / rstat : rstat;
        return q < p.Sq ? v : 0.f;
    };
    const T* pq[A::NCH];
    const T* pg[A::NCH];
    uint32_t oq[A::NCH], og[A::NCH];
    constexpr bool SADDR = CTMI_ATTN_SADDR_DKDV && FAST;          // (off: with it hipcc allocates 170 instead of 163 VGPRs here — the third wave per SIMD)
    if (qt_begin < qt_end) {
        A::load(rq, qp, p.q_rs, (int64_t)qt_begin * 64, p.Sq, hd_, fast, tid);
        A::load(rg, gp, p.o_rs, (int64_t)qt_begin * 64, p.Sq, hd_, fast, tid);
        if constexpr (SADDR) { A::stream_init_off(oq, p.q_rs, tid); A::stream_init_off(og, p.o_rs, tid); }
        else { A::stream_init(pq, qp, p.q_rs, (int64_t)(qt_begin + 1) * 64, tid); A::stream_init(pg, gp, p.o_rs, (int64_t)(qt_begin + 1) * 64, tid); }
        load_stats(qt_begin);
        A::store_rm(rq, QS(0), tid);
        A::store_rm(rg, GS(0), tid);
        if (tid < 192) ST(0)[tid] = stat_value(qt_begin);
    }
    __syncthreads();

    // A tile can hold a masked score only if it crosses the causal diagonal (query block == key block in training) or reaches
    // beyond Sq; every other tile takes the mask-free body (MASKED = false): no per-element compares / selects at all — padding
    // keys still get P through their FINFO_MIN bias (uniform rows) and their dK column is zeroed once, after the loop; rows
    // q >= Sq exist only in a tile that reaches beyond Sq.  The exception is a batch row with LEFT padding: its leading queries
    // see only masked keys (uniform over ALL keys), so every tile of such a row keeps the general body.
    const bool general = AM || !CTMI_ATTN_FASTBODY || (p.kvalid != nullptr && p.first_valid[b] - p.off > 0);
    auto body = [&](auto masked_c, const int t) __attribute__((always_inline)) {
        constexpr bool MASKED = decltype(masked_c)::value;
        if (t + 1 < qt_end && !(ATTN_DBG(p) & 1)) {
            load_stats(t + 1);
            if constexpr (SADDR) {
                A::stream_load_off(rq, oq, qp, p.q_rs, (int64_t)(t + 1) * 64, p.Sq, tid);
                A::stream_load_off(rg, og, gp, p.o_rs, (int64_t)(t + 1) * 64, p.Sq, tid);
            } else {
                A::stream_load(rq, pq, qp, p.q_rs, (int64_t)(t + 1) * 64, p.Sq, hd_, fast, tid);
                A::stream_load(rg, pg, gp, p.o_rs, (int64_t)(t + 1) * 64, p.Sq, hd_, fast, tid);
            }
        }
        f32x4 x[4], y[4];
        const float* st = ST(cur);
        dot_tile<T, HDP>(x, QS(cur), kf, lane);                              // x[nt][r] = q[qi] . k[my_k]
        dot_tile<T, HDP>(y, GS(cur), vf, lane);                              // y[nt][r] = dO[qi] . v[my_k]
        // The per-element masks of the general body are one integer compare each against per-tile, per-lane thresholds on the
        // compile-time offset c = nt*16 + r of the query inside the tile (query index = qbase + c):
        //   masked(c) = padding key | (causal & my_k > q + off)   <=>  c < thr_m
        //   valid(c)  = key row exists & q < Sq                   <=>  c < thr_v
        const int qbase = t * 64 + g * 4;
        const int thr_m = key_pad ? 0x7fffffff : (p.causal ? (int)my_k - (int)p.off - qbase : (int)0x80000000);
        const int thr_v = key_live ? (int)p.Sq - qbase : (int)0x80000000;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const f32x4 mm = *reinterpret_cast<const f32x4*>(st + nt * 16 + g * 4);
            const f32x4 il = *reinterpret_cast<const f32x4*>(st + 64 + nt * 16 + g * 4);
            const f32x4 dl = *reinterpret_cast<const f32x4*>(st + 128 + nt * 16 + g * 4);
            f32x4 s4 = x[nt] * p.scale + my_kb;                                   // packed fma
            if (AM) {
#pragma unroll
                for (int r = 0; r < 4; ++r) s4[r] = score_raw<true>(p, x[nt][r], my_kb, qbase + nt * 16 + r, (int)my_k, am_base);
            }
            bool msk[4] = {false, false, false, false};
            if constexpr (MASKED) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    msk[r] = (nt * 16 + r) < thr_m;
                    s4[r] = msk[r] ? lane_fill : s4[r];
                }
            }
            const f32x4 e4 = (s4 - mm) * 1.4426950408889634f;                     // (s - m) first (finfo.min - finfo.min = 0)
            f32x4 p4;
#pragma unroll
            for (int r = 0; r < 4; ++r) p4[r] = __builtin_amdgcn_exp2f(e4[r]);
            p4 = p4 * il;
            if constexpr (MASKED) {
#pragma unroll
                for (int r = 0; r < 4; ++r) p4[r] = ((nt * 16 + r) < thr_v) ? p4[r] : 0.f;
            }
            f32x4 pv4 = p4, dp4 = y[nt];
            if constexpr (DROP) {                                                 // O = dropout(P) V:  dV uses dropout(P), dP = mask/(1-p) * (dO V^T)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool keep = ctmi_keep_hash(drop_c0 + (uint32_t)(qbase + nt * 16 + r) * (uint32_t)p.Sk, p.drop_seed) >= p.drop_thr;
                    pv4[r] = keep ? p4[r] * p.drop_scale : 0.f;
                    dp4[r] = keep ? dp4[r] * p.drop_scale : 0.f;
                }
            }
            f32x4 d4 = p4 * (dp4 - dl);                                           // already 0 where the row / key does not exist
            if constexpr (MASKED) {
#pragma unroll
                for (int r = 0; r < 4; ++r) d4[r] = msk[r] ? 0.f : d4[r];        // masked entries: P kept (uniform rows), dS = 0
            }
            x[nt] = pv4;
            y[nt] = d4;
        }
        contract64<T, HDP>(dv, GS(cur), x, lane);                            // dV^T[d][my_k] += sum_q dO[q][d] P[q][my_k]
        contract64<T, HDP>(dk, QS(cur), y, lane);                            // dK^T[d][my_k] += sum_q Q[q][d] dS[q][my_k]
        if (NBUF == 1) __syncthreads();
        if (t + 1 < qt_end && !(ATTN_DBG(p) & 2)) {
            const int nx = NBUF == 2 ? cur ^ 1 : 0;
            A::store_rm(rq, QS(nx), tid);
            A::store_rm(rg, GS(nx), tid);
            if (tid < 192) ST(nx)[tid] = stat_value(t + 1);
            cur = nx;
        }
        __syncthreads();
    };
    // Three loops rather than one loop with a per-tile branch: masked tiles are a prefix (the tiles that cross the causal
    // diagonal) and a suffix (a tile reaching beyond Sq) of the query range, and with both bodies inlined into ONE loop hipcc
    // kept the invariants of both alive (207 VGPRs: two waves per SIMD instead of three, slower despite 35 % fewer VALU ops).
    int t = qt_begin;
    int t_plain_beg = qt_end, t_plain_end = qt_end;                          // [t_plain_beg, t_plain_end): tiles that cannot hold a masked score
    if (!general) {
        // first tile with 64*t + off >= k0 + 63 (does not cross the diagonal); tiles entirely inside Sq
        int64_t first_clear = p.causal ? (k0 + 63 - p.off + 63) / 64 : 0;
        if (first_clear < qt_begin) first_clear = qt_begin;
        t_plain_beg = (int)min<int64_t>(first_clear, qt_end);
        t_plain_end = (int)min<int64_t>(p.Sq / 64, qt_end);
        if (t_plain_end < t_plain_beg) t_plain_end = t_plain_beg;
    }
    for (; t < t_plain_beg; ++t) body(std::true_type{}, t);
    for (; t < t_plain_end; ++t) body(std::false_type{}, t);
    for (; t < qt_end; ++t) body(std::true_type{}, t);
    if (!general && key_pad) {                                               // the mask-free tiles left dS != 0 in a padding key's column
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) dk[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (my_k < p.Sk) {
        T* dkp = reinterpret_cast<T*>(p.dk) + b * p.k_bs + h * p.k_hs + my_k * p.k_rs;
        T* dvp = reinterpret_cast<T*>(p.dv) + b * p.v_bs + h * p.v_hs + my_k * p.v_rs;
        store_own_row<T, HDP, FAST>(dkp, dk, p.scale, hd_, g, vok);
        store_own_row<T, HDP, FAST>(dvp, dv, 1.0f, hd_, g, vok);
    }
}

// ------------------------------------------------------------------------------------------------ backward: dQ
// own rows = queries.  Per key tile: S^T = K Q^T, dP^T = V dO^T, dS^T = P^T (dP^T - delta); dQ^T += K^T dS^T.
template <typename T, int HDP, bool AM, bool FAST, bool DROP = false>
__global__ __launch_bounds__(256, CTMI_ATTN_MINW) void attn_bwd_dq_kernel(AttnP p) {
    using A = AT<T, HDP>;
    constexpr int MK = Mma<T>::K, KL = Mma<T>::KL, NKK = HDP / MK, NDT = HDP / 16;
    constexpr int STAGE = DqStage<T, HDP>::BYTES, NBUF = nbuf_for(STAGE);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    int cur = 0;
    auto KS = [&](int s_) { return reinterpret_cast<T*>(smem_raw + s_ * STAGE); };
    auto VS = [&](int s_) { return KS(s_) + A::RM_ELEMS; };
    auto KB = [&](int s_) { return reinterpret_cast<float*>(VS(s_) + A::RM_ELEMS); };
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, g = lane >> 4, li = lane & 15;
    // FAST (compile time): 16-byte aligned strides and hd == HDP — the element-wise gather / guarded store paths are not even
    // compiled into the kernel the training shapes use (their mere presence cost registers, see AT::load)
    const bool fast = FAST || (p.vec_ok && p.hd == HDP);
    const int hd_ = FAST ? HDP : (int)p.hd;
    const bool vok = FAST || p.vec_ok;
    const int nqb = (int)((p.Sq + 63) / 64);
    const int vid = xcd_remap(blockIdx.x, gridDim.x);
    const int qb = nqb - 1 - (vid % nqb);
    const int64_t bh = vid / nqb, h = bh % p.nh, b = bh / p.nh;
    const int64_t q0 = (int64_t)qb * 64, my_q = q0 + wid * 16 + li;
    const bool live = my_q < p.Sq;
    const T* qp = reinterpret_cast<const T*>(p.q) + b * p.q_bs + h * p.q_hs;
    const T* kp = reinterpret_cast<const T*>(p.k) + b * p.k_bs + h * p.k_hs;
    const T* vp = reinterpret_cast<const T*>(p.v) + b * p.v_bs + h * p.v_hs;
    const T* gp = reinterpret_cast<const T*>(p.d_o) + b * p.o_bs + h * p.o_hs;
    const float slope = p.slopes ? p.slopes[h] : 0.f;

    typename Mma<T>::Frag qf[NKK], gf[NKK];
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) {
        qf[kk] = frag_global<T, FAST>(live ? qp + my_q * p.q_rs : nullptr, kk * MK + g * KL, hd_, vok);
        gf[kk] = frag_global<T, FAST>(live ? gp + my_q * p.o_rs : nullptr, kk * MK + g * KL, hd_, vok);
    }
    const int64_t srow = (b * p.nh + h) * p.Sq + my_q;
    const float m = live ? p.stat_m[srow] : 0.f;
    const float il = live ? 1.0f / p.stat_l[srow] : 0.f;
    // delta = rowsum(dO * O) of the lane's own query row, formed here from the dO fragments the kernel holds anyway plus the
    // matching O fragments (the 4 lane groups own the 4 quarters of every 32-wide k-step), and published for the dK/dV kernel,
    // which runs after this one: the separate attn_delta launch (one per layer and step) is gone.
    float dl = 0.f;
    {
        const T* orow = reinterpret_cast<const T*>(p.o) + b * p.o_bs + h * p.o_hs + my_q * p.o_rs;
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk)
            dl += frag_dot<T>(frag_global<T, FAST>(live ? orow : nullptr, kk * MK + g * KL, hd_, vok), gf[kk]);
        dl = group_sum4(dl);
        if (live && g == 0) p.delta[srow] = dl;
    }
    const float* am_base = AM ? p.add_mask + b * p.am_b + h * p.am_h : nullptr;
    const int q_eff = live ? (int)my_q : 0;

    int64_t kv_end = p.Sk;
    if (p.causal) { kv_end = min(p.Sk, q0 + 63 + p.off + 1); if (kv_end < 1) kv_end = 1; }   // masked entries have dS = 0
    const int ntiles = (int)((kv_end + 63) / 64);

    f32x4 dq[NDT];
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};

    u32x4 rk[A::NCH], rv[A::NCH];
    float rkb = 0.f;
    KeyBiasRaw kbr{1, 0.f};
    A::load(rk, kp, p.k_rs, 0, p.Sk, hd_, fast, tid);
    A::load(rv, vp, p.v_rs, 0, p.Sk, hd_, fast, tid);
    const T* pk[A::NCH];
    const T* pv[A::NCH];
    uint32_t ok[A::NCH], ov[A::NCH];
    constexpr bool SADDR = CTMI_ATTN_SADDR && FAST;
    if constexpr (SADDR) { A::stream_init_off(ok, p.k_rs, tid); A::stream_init_off(ov, p.v_rs, tid); }
    else { A::stream_init(pk, kp, p.k_rs, 64, tid); A::stream_init(pv, vp, p.v_rs, 64, tid); }
    if (tid < 64) rkb = key_bias(p, b, tid, slope);
    A::store_rm(rk, KS(0), tid);
    A::store_rm(rv, VS(0), tid);
    if (tid < 64) KB(0)[tid] = rkb;
    __syncthreads();

    // Mask-free body for key tiles entirely in the causal past of all 64 queries (all but the last tile of a query block):
    // padding keys carry the FINFO_MIN bias, so P — and with it dS — is exactly 0 for every row that sees a real key; that is
    // every row unless the batch row has LEFT padding (leading queries see only masked keys: P uniform, dS must still be 0) or
    // the query block reaches beyond Sq (dead rows: 1/l = 0 but exp of an un-shifted score may overflow) — those keep the general body.
    const bool general = AM || !CTMI_ATTN_FASTBODY || (p.kvalid != nullptr && p.first_valid[b] - p.off > 0) || (q0 + 64 > p.Sq);
    auto body = [&](auto masked_c, const int t) __attribute__((always_inline)) {
        constexpr bool MASKED = decltype(masked_c)::value;
        if (t + 1 < ntiles && !(ATTN_DBG(p) & 1)) {
            if (tid < 64) kbr.load(p, b, (int64_t)(t + 1) * 64 + tid);
            if constexpr (SADDR) {
                A::stream_load_off(rk, ok, kp, p.k_rs, (int64_t)(t + 1) * 64, p.Sk, tid);
                A::stream_load_off(rv, ov, vp, p.v_rs, (int64_t)(t + 1) * 64, p.Sk, tid);
            } else {
                A::stream_load(rk, pk, kp, p.k_rs, (int64_t)(t + 1) * 64, p.Sk, hd_, fast, tid);
                A::stream_load(rv, pv, vp, p.v_rs, (int64_t)(t + 1) * 64, p.Sk, hd_, fast, tid);
            }
        }
        f32x4 x[4], y[4];
        const float* kbs = KB(cur);
        dot_tile<T, HDP>(x, KS(cur), qf, lane);
        dot_tile<T, HDP>(y, VS(cur), gf, lane);
        const int kv0 = t * 64;
        // usable(c) for the key at offset c = nt*16 + r of this lane's 4-key group: row alive & not in the causal future
        const int thr_u = live ? (p.causal ? q_eff + (int)p.off - kv0 - g * 4 : 0x7fffffff) : (int)0x80000000;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const f32x4 kb4 = *reinterpret_cast<const f32x4*>(kbs + nt * 16 + g * 4);
            const int k0t = kv0 + nt * 16 + g * 4;
            f32x4 s4 = x[nt] * p.scale + kb4;
            if (AM) {
#pragma unroll
                for (int r = 0; r < 4; ++r) s4[r] = score_raw<true>(p, x[nt][r], kb4[r], q_eff, k0t + r, am_base);
            }
            const f32x4 e4 = (s4 - m) * 1.4426950408889634f;
            f32x4 p4;
#pragma unroll
            for (int r = 0; r < 4; ++r) p4[r] = __builtin_amdgcn_exp2f(e4[r]);
            f32x4 dp4 = y[nt];
            if constexpr (DROP) {
                const uint32_t c0 = (uint32_t)(((b * p.nh + h) * p.Sq + q_eff) * p.Sk + k0t);
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    dp4[r] = ctmi_keep_hash(c0 + (uint32_t)r, p.drop_seed) >= p.drop_thr ? dp4[r] * p.drop_scale : 0.f;
            }
            f32x4 d4 = (p4 * il) * (dp4 - dl);
            if constexpr (MASKED) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool use = (kb4[r] > FINFO_MIN) & ((nt * 16 + r) <= thr_u);   // padding / missing / future keys, dead rows: dS = 0
                    d4[r] = use ? d4[r] : 0.f;
                }
            }
            y[nt] = d4;
        }
        contract64<T, HDP>(dq, KS(cur), y, lane);                            // dQ^T[d][my_q] += sum_key K[key][d] dS[my_q][key]
        if (NBUF == 1) __syncthreads();
        if (t + 1 < ntiles && !(ATTN_DBG(p) & 2)) {
            const int nx = NBUF == 2 ? cur ^ 1 : 0;
            A::store_rm(rk, KS(nx), tid);
            A::store_rm(rv, VS(nx), tid);
            if (tid < 64) KB(nx)[tid] = kbr.value(p, (int64_t)(t + 1) * 64 + tid, slope);
            cur = nx;
        }
        __syncthreads();
    };
    // two loops (see the dK/dV kernel): key tiles entirely in the causal past of all 64 queries and inside Sk first, the rest after
    int n_plain = 0;
    if (!general) {
        // tile t is clear iff t*64 + 63 <= q0 + off (no future pair) and (t+1)*64 <= Sk
        int64_t lim = p.causal ? (q0 + p.off - 63 + 64) / 64 : ntiles;       // number of t with t*64 + 63 <= q0 + off  (floor((q0+off-63)/64) + 1)
        if (p.causal && q0 + p.off - 63 < 0) lim = 0;
        n_plain = (int)max<int64_t>(0, min<int64_t>(min<int64_t>(lim, p.Sk / 64), ntiles));
    }
    int t = 0;
    for (; t < n_plain; ++t) body(std::false_type{}, t);
    for (; t < ntiles; ++t) body(std::true_type{}, t);
    if (live) {
        T* dqp = reinterpret_cast<T*>(p.dq) + b * p.q_bs + h * p.q_hs + my_q * p.q_rs;
        store_own_row<T, HDP, FAST>(dqp, dq, p.scale, hd_, g, vok);
    }
}

// ------------------------------------------------------------------------------------------------ host side
static int fill_params(AttnP& p, const ctmi_attn_desc* d, int dtype, const char* who) {
    CTMI_REQUIRE(d != nullptr, "%s: null desc", who);
    CTMI_REQUIRE(d->B > 0 && d->nh > 0 && d->Sq > 0 && d->Sk > 0 && d->hd > 0 && d->hd <= 128, "%s: bad shape (hd must be <= 128)", who);
    CTMI_REQUIRE(dtype == CTMI_F32 || dtype == CTMI_BF16 || dtype == CTMI_F16, "%s: unsupported dtype %d", who, dtype);
    p.B = d->B; p.nh = d->nh; p.Sq = d->Sq; p.Sk = d->Sk; p.hd = d->hd;
    p.q_bs = d->q_bs; p.q_hs = d->q_hs; p.q_rs = d->q_rs; p.k_bs = d->k_bs; p.k_hs = d->k_hs; p.k_rs = d->k_rs;
    p.v_bs = d->v_bs; p.v_hs = d->v_hs; p.v_rs = d->v_rs; p.o_bs = d->o_bs; p.o_hs = d->o_hs; p.o_rs = d->o_rs;
    p.am_b = d->am_b; p.am_h = d->am_h; p.am_q = d->am_q; p.am_k = d->am_k;
    p.scale = d->scale; p.causal = d->causal; p.off = (int)(d->Sk - d->Sq);
    CTMI_REQUIRE(d->dropout_p >= 0.0f && d->dropout_p < 1.0f, "%s: dropout_p must be in [0, 1)", who);
    p.drop_thr = ctmi_drop_threshold(d->dropout_p); p.drop_seed = d->dropout_seed;
    p.drop_scale = 1.0f / (1.0f - d->dropout_p);
    p.future_fill = d->future_fill == 0.0f ? FINFO_MIN : d->future_fill;
    CTMI_REQUIRE(p.future_fill < 0.0f && p.future_fill >= FINFO_MIN, "%s: future_fill must be 0 (= finfo.min) or a negative finite value", who);
    { static int dbg = -1; if (dbg < 0) { const char* e = getenv("CTMI_ATTN_DBG"); dbg = e ? atoi(e) : 0; } p.dbg = dbg; }
    const int vec = dtype == CTMI_F32 ? 4 : 8;
    auto ok = [&](const void* ptr, int64_t a, int64_t b2, int64_t c) {
        return ptr == nullptr || (((((uintptr_t)ptr) & 15) == 0) && a % vec == 0 && b2 % vec == 0 && c % vec == 0);
    };
    p.vec_ok = ok(p.q, p.q_bs, p.q_hs, p.q_rs) && ok(p.k, p.k_bs, p.k_hs, p.k_rs) && ok(p.v, p.v_bs, p.v_hs, p.v_rs) &&
               ok(p.out, p.o_bs, p.o_hs, p.o_rs) && ok(p.o, p.o_bs, p.o_hs, p.o_rs) && ok(p.d_o, p.o_bs, p.o_hs, p.o_rs) &&
               ok(p.dq, p.q_bs, p.q_hs, p.q_rs) && ok(p.dk, p.k_bs, p.k_hs, p.k_rs) && ok(p.dv, p.v_bs, p.v_hs, p.v_rs);
    return CTMI_OK;
}

template <typename K>
static void launch_k(K kern, int64_t grid, size_t lds, hipStream_t st, AttnP& p) {
    ctmi_dyn_lds(reinterpret_cast<const void*>(kern), lds);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, st, p);
}

template <typename T, int HDP>
static int fwd_launch(AttnP& p, hipStream_t st) {
    using A = AT<T, HDP>;
    const size_t lds = (size_t)FwdStage<T, HDP>::BYTES * nbuf_for(FwdStage<T, HDP>::BYTES);
    const int64_t grid = ((p.Sq + 63) / 64) * p.B * p.nh;
    if (p.drop_thr != 0) {                                             // dropout instantiations: general kernels only (not the measured path)
            if (p.add_mask) launch_k(&attn_fwd_kernel<T, HDP, true, false, true>, grid, lds, st, p);
            else launch_k(&attn_fwd_kernel<T, HDP, false, false, true>, grid, lds, st, p);
        } else if (p.add_mask) launch_k(&attn_fwd_kernel<T, HDP, true, false>, grid, lds, st, p);
        else if (p.vec_ok && p.hd == HDP) launch_k(&attn_fwd_kernel<T, HDP, false, true>, grid, lds, st, p);
        else launch_k(&attn_fwd_kernel<T, HDP, false, false>, grid, lds, st, p);
    CTMI_CHECK_LAUNCH("attn_fwd");
    return CTMI_OK;
}
template <typename T, int HDP>
static int bwd_launch(AttnP& p, hipStream_t st) {
    using A = AT<T, HDP>;
    {   // dQ first: it also publishes delta = rowsum(dO * O) for the dK/dV kernel below (same stream: ordered)
        const size_t lds = (size_t)DqStage<T, HDP>::BYTES * nbuf_for(DqStage<T, HDP>::BYTES);
        const int64_t grid = ((p.Sq + 63) / 64) * p.B * p.nh;
        if (p.drop_thr != 0) {                                             // dropout instantiations: general kernels only (not the measured path)
            if (p.add_mask) launch_k(&attn_bwd_dq_kernel<T, HDP, true, false, true>, grid, lds, st, p);
            else launch_k(&attn_bwd_dq_kernel<T, HDP, false, false, true>, grid, lds, st, p);
        } else if (p.add_mask) launch_k(&attn_bwd_dq_kernel<T, HDP, true, false>, grid, lds, st, p);
        else if (p.vec_ok && p.hd == HDP) launch_k(&attn_bwd_dq_kernel<T, HDP, false, true>, grid, lds, st, p);
        else launch_k(&attn_bwd_dq_kernel<T, HDP, false, false>, grid, lds, st, p);
        CTMI_CHECK_LAUNCH("attn_bwd_dq");
    }
    {
        const size_t lds = (size_t)DkdvStage<T, HDP>::BYTES * nbuf_for(DkdvStage<T, HDP>::BYTES);
        const int64_t grid = ((p.Sk + 63) / 64) * p.B * p.nh;
        if (p.drop_thr != 0) {                                             // dropout instantiations: general kernels only (not the measured path)
            if (p.add_mask) launch_k(&attn_bwd_dkdv_kernel<T, HDP, true, false, true>, grid, lds, st, p);
            else launch_k(&attn_bwd_dkdv_kernel<T, HDP, false, false, true>, grid, lds, st, p);
        } else if (p.add_mask) launch_k(&attn_bwd_dkdv_kernel<T, HDP, true, false>, grid, lds, st, p);
        else if (p.vec_ok && p.hd == HDP) launch_k(&attn_bwd_dkdv_kernel<T, HDP, false, true>, grid, lds, st, p);
        else launch_k(&attn_bwd_dkdv_kernel<T, HDP, false, false>, grid, lds, st, p);
        CTMI_CHECK_LAUNCH("attn_bwd_dkdv");
    }
    return CTMI_OK;
}

#define HDP_DISPATCH(FN, T) \
    (p.hd <= 32 ? FN<T, 32>(p, st) : (p.hd <= 64 ? FN<T, 64>(p, st) : FN<T, 128>(p, st)))

extern "C" int ctmi_attn_fwd(const void* q, const void* k, const void* v, void* o, float* stat_m, float* stat_l,
                             const float* slopes, const float* kpos, const int32_t* kvalid, const int32_t* first_valid,
                             const float* add_mask, const ctmi_attn_desc* desc, int dtype, void* stream) {
    CTMI_REQUIRE(q && k && v && o && stat_m && stat_l, "attn_fwd: null pointer");
    CTMI_REQUIRE((slopes == nullptr) == (kpos == nullptr), "attn_fwd: slopes and kpos go together");
    CTMI_REQUIRE(kvalid == nullptr || first_valid != nullptr, "attn_fwd: kvalid needs first_valid");
    AttnP p = {};
    p.q = q; p.k = k; p.v = v; p.out = o; p.stat_m = stat_m; p.stat_l = stat_l;
    p.slopes = slopes; p.kpos = kpos; p.kvalid = kvalid; p.first_valid = first_valid; p.add_mask = add_mask;
    int rc = fill_params(p, desc, dtype, "attn_fwd");
    if (rc != CTMI_OK) return rc;
    hipStream_t st = as_stream(stream);
    ProfScope prof__(CTMI_PROF_ATTN_FWD, st);
    if (dtype == CTMI_F32) return HDP_DISPATCH(fwd_launch, float);
    if (dtype == CTMI_F16) {                                                             // fp16: the same two kernel families
        if (ctmi_attn32_fwd(p, st, 1)) { CTMI_CHECK_LAUNCH("attn32_fwd"); return CTMI_OK; }
        return HDP_DISPATCH(fwd_launch, f16_t);
    }
    if (ctmi_attn32_fwd(p, st)) { CTMI_CHECK_LAUNCH("attn32_fwd"); return CTMI_OK; }   // training shapes: attention_w32.hip
    return HDP_DISPATCH(fwd_launch, bf16_t);
}

extern "C" int ctmi_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* d_o,
                             const float* stat_m, const float* stat_l, void* dq, void* dk, void* dv, float* delta,
                             const float* slopes, const float* kpos, const int32_t* kvalid, const int32_t* first_valid,
                             const float* add_mask, const ctmi_attn_desc* desc, int dtype, void* stream) {
    CTMI_REQUIRE(q && k && v && o && d_o && stat_m && stat_l && dq && dk && dv && delta, "attn_bwd: null pointer");
    CTMI_REQUIRE((slopes == nullptr) == (kpos == nullptr), "attn_bwd: slopes and kpos go together");
    CTMI_REQUIRE(kvalid == nullptr || first_valid != nullptr, "attn_bwd: kvalid needs first_valid");
    AttnP p = {};
    p.q = q; p.k = k; p.v = v; p.o = o; p.d_o = d_o; p.dq = dq; p.dk = dk; p.dv = dv;
    p.stat_m = const_cast<float*>(stat_m); p.stat_l = const_cast<float*>(stat_l); p.delta = delta;
    p.slopes = slopes; p.kpos = kpos; p.kvalid = kvalid; p.first_valid = first_valid; p.add_mask = add_mask;
    int rc = fill_params(p, desc, dtype, "attn_bwd");
    if (rc != CTMI_OK) return rc;
    hipStream_t st = as_stream(stream);
    ProfScope prof__(CTMI_PROF_ATTN_BWD, st);
    if (dtype == CTMI_F32) return HDP_DISPATCH(bwd_launch, float);
    if (dtype == CTMI_F16) {
        if (ctmi_attn32_bwd(p, st, 1)) { CTMI_CHECK_LAUNCH("attn32_bwd"); return CTMI_OK; }
        return HDP_DISPATCH(bwd_launch, f16_t);
    }
    if (ctmi_attn32_bwd(p, st)) { CTMI_CHECK_LAUNCH("attn32_bwd"); return CTMI_OK; }   // training shapes: attention_w32.hip
    return HDP_DISPATCH(bwd_launch, bf16_t);
}
