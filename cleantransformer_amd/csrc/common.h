// Shared device/host helpers for the ctmi355 kernels (gfx950 / CDNA4 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/ctmi355.h"

typedef uint16_t bf16_t;                                   // raw bfloat16 bits
// IEEE half (round 5: compute_dtype "fp16" — the autocast dtype of the reference's published DDP launch, ft_bloom_DDP.py:107-128 / scripts/ft_bloom_DDP.sh:11).
// A distinct C++ type (bf16_t is a plain uint16_t) so that every kernel template gets its own instantiation: the LDS-DMA GEMM family, the grouped
// weight gradients and the 128-row attention kernels have fp16 twins of their bf16 forms (same schedules, the f16 MFMA), like every other kernel.
struct f16_t { uint16_t v; };
typedef short  short8 __attribute__((ext_vector_type(8)));   // 8 x bf16 MFMA operand (4 VGPRs)
typedef short  short4v __attribute__((ext_vector_type(4)));
typedef float  f32x4 __attribute__((ext_vector_type(4)));    // 16x16 MFMA accumulator

#define WAVE 64
#define FINFO_MIN (-3.4028234663852886e+38f)                // torch.finfo(torch.float32).min

// ---------------------------------------------------------------- error plumbing (host)
void ctmi_set_error(const char* fmt, ...);
#define CTMI_REQUIRE(cond, ...) do { if (!(cond)) { ctmi_set_error(__VA_ARGS__); return CTMI_ERR_ARG; } } while (0)
// Dynamic-LDS opt-in of a kernel (every launch that asks for more than the 64 KiB default).  A refusal — a request beyond the 160 KiB of a CU, a
// kernel whose static + dynamic LDS no longer fits — is not dropped: it is remembered for this thread and the next CTMI_CHECK_LAUNCH returns
// CTMI_ERR_LAUNCH with the attribute call's own message, whatever the launch behind it reported (round-5 verdict: the return was discarded).
void ctmi_dyn_lds(const void* kern, size_t lds_bytes);
bool ctmi_take_attr_error();                               // true once after a refused ctmi_dyn_lds (the message is in ctmi_last_error)
#define CTMI_CHECK_LAUNCH(name) do { hipError_t e__ = hipGetLastError(); if (ctmi_take_attr_error()) return CTMI_ERR_LAUNCH; if (e__ != hipSuccess) { \
    ctmi_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); return CTMI_ERR_LAUNCH; } } while (0)

// ---------------------------------------------------------------- scalar conversions
__device__ __forceinline__ float bf2f(bf16_t x) { return __uint_as_float(((uint32_t)x) << 16); }
// fp32 -> bf16, round-to-nearest-even: the native conversion, which hipcc lowers to gfx950's v_cvt_pk_bf16_f32
typedef __bf16 bf16x2_native __attribute__((ext_vector_type(2)));
typedef float f32x2_native __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
    const f32x2_native v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_native));
}

__device__ __forceinline__ float h2f(f16_t x) { return (float)__builtin_bit_cast(_Float16, x.v); }
__device__ __forceinline__ f16_t f2h(float f) { f16_t r; r.v = __builtin_bit_cast(uint16_t, (_Float16)f); return r; }     // round-to-nearest-even, overflow -> inf (what a loss scaler looks for)
typedef _Float16 f16x2_native __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_h2(float lo, float hi) {
    const f32x2_native v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2_native));
}
__device__ __forceinline__ void unpack_h2(uint32_t w, float& lo, float& hi) {
    const f16x2_native h = __builtin_bit_cast(f16x2_native, w);
    lo = (float)h[0]; hi = (float)h[1];
}

template <typename T> struct Cvt;
template <> struct Cvt<float>  { static __device__ __forceinline__ float to_f(float x) { return x; }
                                 static __device__ __forceinline__ float from_f(float x) { return x; } };
template <> struct Cvt<bf16_t> { static __device__ __forceinline__ float to_f(bf16_t x) { return bf2f(x); }
                                 static __device__ __forceinline__ bf16_t from_f(float x) { return f2bf(x); } };
template <> struct Cvt<f16_t>  { static __device__ __forceinline__ float to_f(f16_t x) { return h2f(x); }
                                 static __device__ __forceinline__ f16_t from_f(float x) { return f2h(x); } };

// 16-byte vector of T: VEC = 16/sizeof(T) elements
template <typename T> struct Vec16 { static constexpr int N = 16 / sizeof(T); uint4 raw; };

template <typename T> __device__ __forceinline__ void unpack16(const uint4& r, float* out);
template <> __device__ __forceinline__ void unpack16<float>(const uint4& r, float* o) {
    o[0] = __uint_as_float(r.x); o[1] = __uint_as_float(r.y); o[2] = __uint_as_float(r.z); o[3] = __uint_as_float(r.w);
}
template <> __device__ __forceinline__ void unpack16<bf16_t>(const uint4& r, float* o) {
    o[0] = __uint_as_float(r.x << 16); o[1] = __uint_as_float(r.x & 0xffff0000u);
    o[2] = __uint_as_float(r.y << 16); o[3] = __uint_as_float(r.y & 0xffff0000u);
    o[4] = __uint_as_float(r.z << 16); o[5] = __uint_as_float(r.z & 0xffff0000u);
    o[6] = __uint_as_float(r.w << 16); o[7] = __uint_as_float(r.w & 0xffff0000u);
}
template <> __device__ __forceinline__ void unpack16<f16_t>(const uint4& r, float* o) {
    unpack_h2(r.x, o[0], o[1]); unpack_h2(r.y, o[2], o[3]); unpack_h2(r.z, o[4], o[5]); unpack_h2(r.w, o[6], o[7]);
}
template <typename T> __device__ __forceinline__ uint4 pack16(const float* in);
template <> __device__ __forceinline__ uint4 pack16<float>(const float* i) {
    return make_uint4(__float_as_uint(i[0]), __float_as_uint(i[1]), __float_as_uint(i[2]), __float_as_uint(i[3]));
}
template <> __device__ __forceinline__ uint4 pack16<bf16_t>(const float* i) {
    return make_uint4(pack_bf2(i[0], i[1]), pack_bf2(i[2], i[3]), pack_bf2(i[4], i[5]), pack_bf2(i[6], i[7]));
}
template <> __device__ __forceinline__ uint4 pack16<f16_t>(const float* i) {
    return make_uint4(pack_h2(i[0], i[1]), pack_h2(i[2], i[3]), pack_h2(i[4], i[5]), pack_h2(i[6], i[7]));
}
// two 16-bit elements of T from two floats (the 4- and 8-byte store helpers of the kernels)
template <typename T> __device__ __forceinline__ uint32_t pack2(float lo, float hi);
template <> __device__ __forceinline__ uint32_t pack2<bf16_t>(float lo, float hi) { return pack_bf2(lo, hi); }
template <> __device__ __forceinline__ uint32_t pack2<f16_t>(float lo, float hi) { return pack_h2(lo, hi); }

// ---------------------------------------------------------------- write-through stores of large outputs
// The eight XCD L2s are write-back and not coherent with each other, so a kernel's dirty lines are flushed at its END — on the critical path of the
// dependent launch behind it.  tools/probes/boundary_dirty.hip (profiles/r06_boundary_dirty.txt): a writer that leaves 16 - 64 MiB behind costs
// the boundary 4.0 - 4.2 us with plain stores and 1.2 - 2.2 us with `sc1` (write-through: the bytes go out to the memory side while the kernel still
// computes, the line is dropped from the XCD's L2 — the consumers of a [T, *] activation run on all eight XCDs and read it from the Infinity Cache
// either way).  Kernels whose outputs are streamed once and read by the NEXT kernels (GEMM epilogues, LayerNorm, attention) store through these.
// The asm ends in `s_nop 1`: hipcc's hazard recognizer does not see into asm, and a 16-byte store needs a wait state before its data registers may
// be overwritten (cdna_hip_programming.md 5.7 item 1).  -DCTMI_ST_WT=0 restores plain stores (A/B builds).
#ifndef CTMI_ST_WT
#define CTMI_ST_WT 1
#endif
typedef uint32_t ctmi_u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t ctmi_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void st_wt16(void* p, const uint4& v) {
#if CTMI_ST_WT
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(ctmi_u32x4{v.x, v.y, v.z, v.w}) : "memory");
#else
    *reinterpret_cast<uint4*>(p) = v;
#endif
}
// fp32 outputs (weight gradients: 32-byte row pieces per lane, two store instructions per 128-byte line) stay write-back: written through, every
// half-filled line went out twice — the grouped weight-gradient launch 213 -> 232 us (profiles/r06_boundary_dirty.txt)
#ifndef CTMI_ST_WT_F32
#define CTMI_ST_WT_F32 0
#endif
__device__ __forceinline__ void st_wt16(void* p, const f32x4& v) {
#if CTMI_ST_WT && CTMI_ST_WT_F32
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(v) : "memory");
#else
    *reinterpret_cast<f32x4*>(p) = v;
#endif
}
__device__ __forceinline__ void st_wt8(void* p, const uint2& v) {
#if CTMI_ST_WT
    asm volatile("global_store_dwordx2 %0, %1, off sc1\n\ts_nop 0" :: "v"(p), "v"(ctmi_u32x2{v.x, v.y}) : "memory");
#else
    *reinterpret_cast<uint2*>(p) = v;
#endif
}

// ---------------------------------------------------------------- wave / block reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---------------------------------------------------------------- math
// GELU(tanh) through the logistic form: 0.5 * (1 + tanh(a)) == sigma(2a), a = 0.79788456 x (1 + 0.044715 x^2), so
//   gelu(x)  = x * s                                   (modeling_bloom.py:344)
//   gelu'(x) = s * (1 + x (1 - s) (2*0.79788456 + 2*0.1070322243 x^2))   (modeling_bloom.py:360-362, same algebra)
// with s = 1 / (1 + 2^z), z = -x (c1 + c2 x^2), c1 = 2*0.79788456*log2(e), c2 = 0.044715 c1: one v_exp_f32 and one
// v_rcp_f32 per element (abs error ~1e-7) and about half the VALU work of the tanh form; the f32x2 versions compile to
// v_pk_mul/fma/add_f32.  The GEMM epilogues that apply them are not hidden behind MFMA work, so their length matters.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 gelu_sigma_pk(f32x2 x, f32x2 x2) {
    const f32x2 z = -x * (x2 * 0.10294324f + 2.3022082f);
    const f32x2 d = f32x2{__builtin_amdgcn_exp2f(z[0]), __builtin_amdgcn_exp2f(z[1])} + 1.0f;
    return f32x2{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
}
__device__ __forceinline__ f32x2 gelu_tanh_pk(f32x2 x) { return x * gelu_sigma_pk(x, x * x); }
__device__ __forceinline__ f32x2 gelu_tanh_grad_pk(f32x2 x) {
    const f32x2 x2 = x * x;
    const f32x2 s = gelu_sigma_pk(x, x2);
    return s * (x * (1.0f - s) * (x2 * 0.21406445f + 1.5957691f) + 1.0f);
}
__device__ __forceinline__ float gelu_tanh_f(float x) { return gelu_tanh_pk(f32x2{x, x})[0]; }
__device__ __forceinline__ float gelu_tanh_grad_f(float x) { return gelu_tanh_grad_pk(f32x2{x, x})[0]; }

// ---------------------------------------------------------------- dropout: counter-based keep mask
// keep(i) = keep_hash(i, seed) >= thr with thr = p * 2^32: a pure function of the element's 32-bit counter and a per-call seed,
// so the backward regenerates the mask instead of storing it (no [S,S] or [T,H] mask tensor ever exists).  hash32 = the "lowbias32"
// integer finaliser (two multiplies, three xor-shifts); seeds come well mixed from the host (cleantransformer_amd/rng.py).
// keep_hash is KEYED by the seed in two places: hash32(hash32(i ^ seed) + key(seed)).  With the xor alone every mask of a run would be
// an xor-relabelled window of ONE 2^32-long bit sequence (two sites whose seeds differ only in high bits draw permutations of the same
// mask); the additive key of the second round makes two seeds two different functions of the counter.
__host__ __device__ __forceinline__ uint32_t ctmi_hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x21f0aaadu; x ^= x >> 15; x *= 0x735a2d97u; x ^= x >> 15;
    return x;
}
__host__ __device__ __forceinline__ uint32_t ctmi_keep_key(uint32_t seed) { return seed * 0x9E3779B1u + 0x7F4A7C15u; }
__host__ __device__ __forceinline__ uint32_t ctmi_keep_hash(uint32_t counter, uint32_t seed) {
    return ctmi_hash32(ctmi_hash32(counter ^ seed) + ctmi_keep_key(seed));
}
static inline uint32_t ctmi_drop_threshold(float p) {                  // p in [0, 1): P(hash < thr) = p to 2^-32
    const double t = (double)p * 4294967296.0;
    return t <= 0.0 ? 0u : (t >= 4294967295.0 ? 4294967295u : (uint32_t)t);
}

static inline hipStream_t as_stream(void* s) { return (hipStream_t)s; }
static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
