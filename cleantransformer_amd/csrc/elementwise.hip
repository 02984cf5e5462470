// HBM-bound kernels of the SFT hot path: LayerNorm, cross entropy, embedding, optimizers, small utilities.
// gfx950: one wave (64 lanes) per row for LayerNorm, one 256-thread block per vocabulary row for CE,
// 16-byte loads everywhere, fp32 statistics, wavefront shuffles for the reductions.
#include "common.h"
#include "reduce_jobs.h"
#include "prof.h"
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

// CTMI_LN_BWD_PREFETCH: ln_bwd_vec keeps the loads of its next row in flight while it reduces and stores the current one (round 4).
#ifndef CTMI_LN_BWD_PREFETCH
#define CTMI_LN_BWD_PREFETCH 1
#endif

// ------------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void ctmi_set_error(const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
}
extern "C" const char* ctmi_last_error(void) { return g_err; }
static thread_local bool g_attr_err = false;
void ctmi_dyn_lds(const void* kern, size_t lds_bytes) {
    const hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        ctmi_set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize = %zu bytes) refused: %s", lds_bytes, hipGetErrorString(e));
        g_attr_err = true;
    }
}
bool ctmi_take_attr_error() { const bool r = g_attr_err; g_attr_err = false; return r; }
extern "C" int ctmi_abi_version(void) { return CTMI_ABI_VERSION; }

static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// ------------------------------------------------------------------------------------------------
// LayerNorm  (transformer.py:71-89)
// ------------------------------------------------------------------------------------------------
// Fast path: one wave per row, the row cached in registers (MAXV 16-byte vectors per lane).
template <typename T, int MAXV>
__global__ __launch_bounds__(256) void ln_fwd_vec(const T* __restrict__ x, const float* __restrict__ w,
                                                  const float* __restrict__ b, T* __restrict__ y,
                                                  float* __restrict__ mean_o, float* __restrict__ rstd_o,
                                                  int64_t rows, int cols, float eps) {
    constexpr int VEC = 16 / sizeof(T);
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    const float inv_n = 1.0f / (float)cols;
    for (int64_t row = wave0; row < rows; row += nwaves) {
        const T* xr = x + row * cols;
        float vals[MAXV][VEC];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = (i * 64 + lane) * VEC;
            if (c < cols) {
                uint4 r = *reinterpret_cast<const uint4*>(xr + c);
                unpack16<T>(r, vals[i]);
#pragma unroll
                for (int j = 0; j < VEC; ++j) s += vals[i][j];
            }
        }
        const float mean = wave_sum(s) * inv_n;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = (i * 64 + lane) * VEC;
            if (c < cols) {
#pragma unroll
                for (int j = 0; j < VEC; ++j) { float d = vals[i][j] - mean; q += d * d; }
            }
        }
        const float var = wave_sum(q) * inv_n;
        const float rstd = 1.0f / sqrtf(var + eps);
        if (lane == 0) { mean_o[row] = mean; rstd_o[row] = rstd; }
        T* yr = y + row * cols;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = (i * 64 + lane) * VEC;
            if (c < cols) {
                float o[VEC];
#pragma unroll
                for (int j = 0; j < VEC; ++j) o[j] = w[c + j] * ((vals[i][j] - mean) * rstd) + b[c + j];
                st_wt16(yr + c, pack16<T>(o));                           // (write-through: common.h)
            }
        }
    }
}

// Generic path (any cols / alignment): scalar loads, the row is re-read (L1/L2 hits).
template <typename T>
__global__ __launch_bounds__(256) void ln_fwd_gen(const T* __restrict__ x, const float* __restrict__ w,
                                                  const float* __restrict__ b, T* __restrict__ y,
                                                  float* __restrict__ mean_o, float* __restrict__ rstd_o,
                                                  int64_t rows, int64_t cols, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    const float inv_n = 1.0f / (float)cols;
    for (int64_t row = wave0; row < rows; row += nwaves) {
        const T* xr = x + row * cols;
        float s = 0.f;
        for (int64_t c = lane; c < cols; c += 64) s += Cvt<T>::to_f(xr[c]);
        const float mean = wave_sum(s) * inv_n;
        float q = 0.f;
        for (int64_t c = lane; c < cols; c += 64) { float d = Cvt<T>::to_f(xr[c]) - mean; q += d * d; }
        const float rstd = 1.0f / sqrtf(wave_sum(q) * inv_n + eps);
        if (lane == 0) { mean_o[row] = mean; rstd_o[row] = rstd; }
        T* yr = y + row * cols;
        for (int64_t c = lane; c < cols; c += 64)
            yr[c] = Cvt<T>::from_f(w[c] * ((Cvt<T>::to_f(xr[c]) - mean) * rstd) + b[c]);
    }
}

template <typename T>
static int ln_fwd_launch(const void* x, const float* w, const float* b, void* y, float* mean, float* rstd,
                         int64_t rows, int64_t cols, float eps, hipStream_t st) {
    constexpr int VEC = 16 / sizeof(T);
    const bool vec_ok = (cols % VEC == 0) && aligned16(x) && aligned16(y);
    int grid = (int)std::min<int64_t>(cdiv64(rows, 4), 2048);
    if (grid < 1) grid = 1;
    const T* xp = (const T*)x; T* yp = (T*)y;
#define LN_FWD_CASE(MV) if (vec_ok && cols <= (int64_t)MV * 64 * VEC) { \
        hipLaunchKernelGGL((ln_fwd_vec<T, MV>), dim3(grid), dim3(256), 0, st, xp, w, b, yp, mean, rstd, rows, (int)cols, eps); \
        CTMI_CHECK_LAUNCH("layernorm_fwd"); return CTMI_OK; }
    LN_FWD_CASE(1) LN_FWD_CASE(2) LN_FWD_CASE(4) LN_FWD_CASE(8)
#undef LN_FWD_CASE
    hipLaunchKernelGGL((ln_fwd_gen<T>), dim3(grid), dim3(256), 0, st, xp, w, b, yp, mean, rstd, rows, cols, eps);
    CTMI_CHECK_LAUNCH("layernorm_fwd");
    return CTMI_OK;
}

extern "C" int ctmi_layernorm_fwd(const void* x, const float* w, const float* b, void* y, float* mean, float* rstd,
                                  int64_t rows, int64_t cols, float eps, int dtype, void* stream) {
    ProfScope prof__(CTMI_PROF_LAYERNORM, as_stream(stream));
    CTMI_REQUIRE(x && w && b && y && mean && rstd, "layernorm_fwd: null pointer");
    CTMI_REQUIRE(rows >= 0 && cols > 0, "layernorm_fwd: bad shape rows=%lld cols=%lld", (long long)rows, (long long)cols);
    if (rows == 0) return CTMI_OK;
    if (dtype == CTMI_F32) return ln_fwd_launch<float>(x, w, b, y, mean, rstd, rows, cols, eps, as_stream(stream));
    if (dtype == CTMI_BF16) return ln_fwd_launch<bf16_t>(x, w, b, y, mean, rstd, rows, cols, eps, as_stream(stream));
    if (dtype == CTMI_F16) return ln_fwd_launch<f16_t>(x, w, b, y, mean, rstd, rows, cols, eps, as_stream(stream));
    ctmi_set_error("layernorm_fwd: unsupported dtype %d", dtype);
    return CTMI_ERR_UNSUPPORTED;
}

// Backward.  dx = rstd * (g - mean(g) - xhat*mean(g*xhat)),  g = dy*w, xhat = (x-mean)*rstd.
// Each wave keeps per-lane partial sums of dw = sum dy*xhat and db = sum dy for its rows; a block combines its
// 4 waves through LDS and writes one partial row to ws[blockIdx][2][cols]; ln_bwd_reduce sums the partial rows.
// waves per workgroup in ln_bwd_vec: 8 for rows up to 64*VEC*2 elements, fewer for wider rows so the per-wave dw/db partials
// ([waves][2][cols] fp32 in LDS) stay within 64 KiB
static constexpr int lnb_waves(int maxv) { return maxv <= 2 ? 8 : (16 / maxv); }
// NS = 2: partial rows {dw, db};  NS = 4: additionally {colsum(dres), colsum(dx)} — the bias gradients of the two Linear layers
// whose output gradients this kernel already reads (dres) and writes (dx), so no separate column-sum pass over them exists.
// FULL: cols == MAXV * 64 * VEC exactly (no column guards) and RESM = 0 / 1: `dres` known absent / present at compile time — a loop body without
// control flow, in which hipcc counts its vector-memory operations exactly: with the guards its wait at the loop head was `vmcnt(0)`, i.e. the
// acknowledgement of the previous row's STORES was waited for before the next loads were issued.  RESM = -1: decided at run time (any width).
template <typename T, int MAXV, int NS = 2, bool FULL = false, int RESM = -1>
__global__ __launch_bounds__(64 * lnb_waves(MAXV)) void ln_bwd_vec(const T* __restrict__ dy, const T* __restrict__ x,
                                                  const float* __restrict__ w, const float* __restrict__ mean_i,
                                                  const float* __restrict__ rstd_i, const T* __restrict__ dres,
                                                  T* __restrict__ dx, float* __restrict__ ws, int64_t rows, int cols) {
    constexpr int VEC = 16 / sizeof(T);
    const bool has_res = RESM < 0 ? dres != nullptr : RESM == 1;
    constexpr int LNB_WAVES = lnb_waves(MAXV);
    constexpr int NX = NS == 4 ? MAXV : 1;                             // extra accumulators exist only for NS = 4
    float ar[NX][VEC], ax[NX][VEC];
#pragma unroll
    for (int i = 0; i < NX; ++i)
#pragma unroll
        for (int j = 0; j < VEC; ++j) { ar[i][j] = 0.f; ax[i][j] = 0.f; }
    extern __shared__ __attribute__((aligned(16))) float lds[];      // [LNB_WAVES][2][cols]
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int64_t wave0 = (int64_t)blockIdx.x * LNB_WAVES + wid;
    const int64_t nwaves = (int64_t)gridDim.x * LNB_WAVES;
    const float inv_n = 1.0f / (float)cols;
    float aw[MAXV][VEC], ab[MAXV][VEC], wv[MAXV][VEC];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = (i * 64 + lane) * VEC;
#pragma unroll
        for (int j = 0; j < VEC; ++j) { aw[i][j] = 0.f; ab[i][j] = 0.f; wv[i][j] = (FULL || c < cols) ? w[c + j] : 0.f; }
    }
    // all three input rows are requested before anything is reduced (the residual-gradient row is only needed after the two wave
    // reductions, and loaded there it exposed a full HBM latency per row) — and, round 4, one row AHEAD: the loads of the wave's next row
    // are in flight while this row is reduced and stored (with 8 waves per CU a wave that waits for its own row leaves the memory pipe
    // idle half the time: 21.7 us for 64 MiB = 2.9 TB/s).  The row after the last is clamped to the last (loaded twice, used once).
    auto load_row = [&](int64_t r, uint4 (&xq)[MAXV], uint4 (&gq)[MAXV], uint4 (&rq)[MAXV], float& mu, float& rs) {
        mu = mean_i[r]; rs = rstd_i[r];
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = (i * 64 + lane) * VEC;
            if (FULL || c < cols) {
                xq[i] = *reinterpret_cast<const uint4*>(x + r * cols + c);
                gq[i] = *reinterpret_cast<const uint4*>(dy + r * cols + c);
                if (has_res) rq[i] = *reinterpret_cast<const uint4*>(dres + r * cols + c);
            }
        }
    };
    // one row: the two wave reductions, dx (+ residual gradient), the per-lane partial sums.  `valid` = false (the clamped row after the last,
    // second half of a two-row trip): nothing is stored and dy counts as zero, so the partial sums do not move.
    auto process = [&](const uint4 (&xraw)[MAXV], const uint4 (&graw)[MAXV], const uint4 (&rraw)[MAXV], float mean, float rstd, int64_t row, bool valid) {
        float xh[MAXV][VEC], g[MAXV][VEC];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = (i * 64 + lane) * VEC;
            if (FULL || c < cols) {
                float xv[VEC], dv[VEC];
                unpack16<T>(xraw[i], xv);
                unpack16<T>(graw[i], dv);
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    if (!valid) dv[j] = 0.f;
                    xh[i][j] = (xv[j] - mean) * rstd;
                    g[i][j] = dv[j] * wv[i][j];
                    s1 += g[i][j];
                    s2 += g[i][j] * xh[i][j];
                    aw[i][j] += dv[j] * xh[i][j];
                    ab[i][j] += dv[j];
                }
            }
        }
        s1 = wave_sum(s1) * inv_n;
        s2 = wave_sum(s2) * inv_n;
        T* dxr = dx + row * cols;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = (i * 64 + lane) * VEC;
            if (FULL || c < cols) {
                float o[VEC];
#pragma unroll
                for (int j = 0; j < VEC; ++j) o[j] = rstd * (g[i][j] - s1 - xh[i][j] * s2);
                if (has_res) {
                    float r[VEC];
                    unpack16<T>(rraw[i], r);
#pragma unroll
                    for (int j = 0; j < VEC; ++j) { if (!valid) r[j] = 0.f; o[j] += r[j]; if (NS == 4) ar[NS == 4 ? i : 0][j] += r[j]; }
                }
                const uint4 pk = pack16<T>(o);
                if (valid) st_wt16(dxr + c, pk);
                if (NS == 4) {                                           // column sums of dx as STORED (what the weight-gradient GEMM reads)
                    float orr[VEC];
                    unpack16<T>(pk, orr);
#pragma unroll
                    for (int j = 0; j < VEC; ++j) ax[NS == 4 ? i : 0][j] += valid ? orr[j] : 0.f;
                }
            }
        }
    };
    // (NS = 4 keeps one row at a time: with 32 more accumulators the two-set form needs all 256 registers and spills)
    if (CTMI_LN_BWD_PREFETCH && NS == 2) {
        // two rows per trip over two register sets (A, B) — no copies between them: a copy of the prefetched row into "current" registers is
        // where hipcc placed its wait, i.e. at the bottom of the trip that had just issued the loads
        uint4 xa[MAXV], ga[MAXV], ra[MAXV], xb[MAXV], gb[MAXV], rb[MAXV];
        float mean_a = 0.f, rstd_a = 0.f, mean_b = 0.f, rstd_b = 0.f;
        if (wave0 < rows) load_row(wave0, xa, ga, ra, mean_a, rstd_a);
        for (int64_t row = wave0; row < rows; row += 2 * nwaves) {
            const int64_t r1 = row + nwaves, r1c = min(r1, rows - 1);
            load_row(r1c, xb, gb, rb, mean_b, rstd_b);
            process(xa, ga, ra, mean_a, rstd_a, row, true);
            load_row(min(row + 2 * nwaves, rows - 1), xa, ga, ra, mean_a, rstd_a);
            process(xb, gb, rb, mean_b, rstd_b, r1c, r1 < rows);
        }
    } else {
        for (int64_t row = wave0; row < rows; row += nwaves) {
            uint4 xraw[MAXV], graw[MAXV], rraw[MAXV];
            float mean, rstd;
            load_row(row, xraw, graw, rraw, mean, rstd);
            process(xraw, graw, rraw, mean, rstd, row, true);
        }
    }
    // block combine, two partial rows per pass through the [LNB_WAVES][2][cols] LDS patch
    float* out = ws + (int64_t)blockIdx.x * NS * cols;
#pragma unroll
    for (int pass = 0; pass < NS / 2; ++pass) {
        if (pass) __syncthreads();
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = (i * 64 + lane) * VEC;
            if (FULL || c < cols) {
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    lds[(wid * 2 + 0) * cols + c + j] = pass == 0 ? aw[i][j] : ar[NS == 4 ? i : 0][j];
                    lds[(wid * 2 + 1) * cols + c + j] = pass == 0 ? ab[i][j] : ax[NS == 4 ? i : 0][j];
                }
            }
        }
        __syncthreads();
        for (int c = threadIdx.x; c < 2 * cols; c += 64 * LNB_WAVES) {
            const int which = c / cols, cc = c - which * cols;
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < LNB_WAVES; ++k) t += lds[(k * 2 + which) * cols + cc];
            out[pass * 2 * cols + c] = t;
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void ln_bwd_gen(const T* __restrict__ dy, const T* __restrict__ x,
                                                  const float* __restrict__ w, const float* __restrict__ mean_i,
                                                  const float* __restrict__ rstd_i, const T* __restrict__ dres,
                                                  T* __restrict__ dx, int64_t rows, int64_t cols) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    const float inv_n = 1.0f / (float)cols;
    for (int64_t row = wave0; row < rows; row += nwaves) {
        const float mean = mean_i[row], rstd = rstd_i[row];
        const T* xr = x + row * cols;
        const T* gr = dy + row * cols;
        float s1 = 0.f, s2 = 0.f;
        for (int64_t c = lane; c < cols; c += 64) {
            float g = Cvt<T>::to_f(gr[c]) * w[c];
            s1 += g; s2 += g * ((Cvt<T>::to_f(xr[c]) - mean) * rstd);
        }
        s1 = wave_sum(s1) * inv_n; s2 = wave_sum(s2) * inv_n;
        for (int64_t c = lane; c < cols; c += 64) {
            float xh = (Cvt<T>::to_f(xr[c]) - mean) * rstd;
            float o = rstd * (Cvt<T>::to_f(gr[c]) * w[c] - s1 - xh * s2);
            if (dres != nullptr) o += Cvt<T>::to_f(dres[row * cols + c]);
            dx[row * cols + c] = Cvt<T>::from_f(o);
        }
    }
}

// generic dw/db partials: block b handles rows [b*chunk, (b+1)*chunk); thread per column.
template <typename T>
__global__ __launch_bounds__(256) void ln_bwd_wb_gen(const T* __restrict__ dy, const T* __restrict__ x,
                                                     const float* __restrict__ mean_i, const float* __restrict__ rstd_i,
                                                     float* __restrict__ ws, int64_t rows, int64_t cols, int64_t chunk) {
    const int64_t r0 = (int64_t)blockIdx.x * chunk, r1 = min(rows, r0 + chunk);
    float* out = ws + (int64_t)blockIdx.x * 2 * cols;
    for (int64_t c = threadIdx.x; c < cols; c += 256) {
        float aw = 0.f, ab = 0.f;
        for (int64_t r = r0; r < r1; ++r) {
            float d = Cvt<T>::to_f(dy[r * cols + c]);
            aw += d * ((Cvt<T>::to_f(x[r * cols + c]) - mean_i[r]) * rstd_i[r]);
            ab += d;
        }
        out[c] = aw; out[cols + c] = ab;
    }
}

// Wide rows (more than 4 x 64 16-byte chunks: hidden sizes above 2048 in bf16): one 4-wave workgroup per row, each wave
// owning every 4th 1-KiB chunk, so a lane keeps MVW chunks of x / dy / the dw,db partials in registers instead of 8 (the
// one-wave-per-row kernel needs all 256 VGPRs there and leaves one wave per SIMD).  The two row sums cross the waves through
// LDS (double-buffered by row parity: one barrier per row).  Partials: ws[block][2][cols], no cross-wave combine.
template <typename T, int MVW>
__global__ __launch_bounds__(256) void ln_bwd_wide(const T* __restrict__ dy, const T* __restrict__ x, const float* __restrict__ w,
                                                   const float* __restrict__ mean_i, const float* __restrict__ rstd_i,
                                                   const T* __restrict__ dres, T* __restrict__ dx, float* __restrict__ ws,
                                                   int64_t rows, int cols) {
    constexpr int VEC = 16 / sizeof(T);
    __shared__ float sm[2][2][4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const float inv_n = 1.0f / (float)cols;
    float aw[MVW][VEC], ab[MVW][VEC], wvv[MVW][VEC];
    int col[MVW];
#pragma unroll
    for (int i = 0; i < MVW; ++i) {
        col[i] = ((i * 4 + wv) * 64 + lane) * VEC;
#pragma unroll
        for (int j = 0; j < VEC; ++j) { aw[i][j] = 0.f; ab[i][j] = 0.f; wvv[i][j] = (col[i] < cols) ? w[col[i] + j] : 0.f; }
    }
    int par = 0;
    for (int64_t row = blockIdx.x; row < rows; row += gridDim.x, par ^= 1) {
        const float mean = mean_i[row], rstd = rstd_i[row];
        uint4 xraw[MVW], graw[MVW], rraw[MVW];
#pragma unroll
        for (int i = 0; i < MVW; ++i)
            if (col[i] < cols) {
                xraw[i] = *reinterpret_cast<const uint4*>(x + row * cols + col[i]);
                graw[i] = *reinterpret_cast<const uint4*>(dy + row * cols + col[i]);
                if (dres != nullptr) rraw[i] = *reinterpret_cast<const uint4*>(dres + row * cols + col[i]);
            }
        float xh[MVW][VEC], g[MVW][VEC];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < MVW; ++i) {
            if (col[i] < cols) {
                float xv[VEC], dv[VEC];
                unpack16<T>(xraw[i], xv);
                unpack16<T>(graw[i], dv);
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    xh[i][j] = (xv[j] - mean) * rstd;
                    g[i][j] = dv[j] * wvv[i][j];
                    s1 += g[i][j];
                    s2 += g[i][j] * xh[i][j];
                    aw[i][j] += dv[j] * xh[i][j];
                    ab[i][j] += dv[j];
                }
            }
        }
        s1 = wave_sum(s1);
        s2 = wave_sum(s2);
        if (lane == 0) { sm[par][0][wv] = s1; sm[par][1][wv] = s2; }
        __syncthreads();
        s1 = (sm[par][0][0] + sm[par][0][1] + sm[par][0][2] + sm[par][0][3]) * inv_n;
        s2 = (sm[par][1][0] + sm[par][1][1] + sm[par][1][2] + sm[par][1][3]) * inv_n;
#pragma unroll
        for (int i = 0; i < MVW; ++i) {
            if (col[i] < cols) {
                float o[VEC];
#pragma unroll
                for (int j = 0; j < VEC; ++j) o[j] = rstd * (g[i][j] - s1 - xh[i][j] * s2);
                if (dres != nullptr) {
                    float r[VEC];
                    unpack16<T>(rraw[i], r);
#pragma unroll
                    for (int j = 0; j < VEC; ++j) o[j] += r[j];
                }
                *reinterpret_cast<uint4*>(dx + row * cols + col[i]) = pack16<T>(o);
            }
        }
    }
    float* out = ws + (int64_t)blockIdx.x * 2 * cols;
#pragma unroll
    for (int i = 0; i < MVW; ++i)
        if (col[i] < cols) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) { out[col[i] + j] = aw[i][j]; out[cols + col[i] + j] = ab[i][j]; }
        }
}

__global__ __launch_bounds__(256) void ln_bwd_reduce(const float* __restrict__ ws, float* __restrict__ dw,
                                                     float* __restrict__ db, int nparts, int64_t cols, int accumulate) {
    // 64 columns x 4 row-slices per block; blockIdx.y selects dw / db
    __shared__ float sm[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int64_t c = (int64_t)blockIdx.x * 64 + tx;
    const int which = blockIdx.y;
    float t = 0.f;
    if (c < cols) {
#pragma unroll 8
        for (int p = ty; p < nparts; p += 4) t += ws[((int64_t)p * 2 + which) * cols + c];
    }
    sm[ty][tx] = t;
    __syncthreads();
    if (ty == 0 && c < cols) {
        float r = sm[0][tx] + sm[1][tx] + sm[2][tx] + sm[3][tx];
        float* o = which ? db : dw;
        o[c] = accumulate ? o[c] + r : r;
    }
}
// the same for wide rows (cols % 256 == 0): a lane sums 4 adjacent columns with 16-byte loads, so a wave reads 1 KiB of each
// partial row instead of 256 B (the partial rows are 2*cols*4 bytes apart: 256-byte pieces of them ran at ~40 GB/s)
__global__ __launch_bounds__(256) void ln_bwd_reduce4(const float* __restrict__ ws, float* __restrict__ dw,
                                                      float* __restrict__ db, int nparts, int64_t cols, int accumulate) {
    __shared__ float4 sm[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int64_t c = ((int64_t)blockIdx.x * 64 + tx) * 4;
    const int which = blockIdx.y;
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
    for (int p = ty; p < nparts; p += 4) {
        const float4 v = *reinterpret_cast<const float4*>(ws + ((int64_t)p * 2 + which) * cols + c);
        t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
    }
    sm[ty][tx] = t;
    __syncthreads();
    if (ty == 0) {
        float4 r = sm[0][tx];
#pragma unroll
        for (int k = 1; k < 4; ++k) { r.x += sm[k][tx].x; r.y += sm[k][tx].y; r.z += sm[k][tx].z; r.w += sm[k][tx].w; }
        float* o = (which ? db : dw) + c;
        if (accumulate) { r.x += o[0]; r.y += o[1]; r.z += o[2]; r.w += o[3]; }
        o[0] = r.x; o[1] = r.y; o[2] = r.z; o[3] = r.w;
    }
}

#ifndef CTMI_LN_BWD_BLOCKS
#define CTMI_LN_BWD_BLOCKS 256      // one workgroup per CU: same-box A/B vs 512: 31.0 -> 21.4 us at [8192,1024] bf16 (half the end-of-kernel LDS combines and partial rows)
#endif
static const int LN_BWD_MAX_BLOCKS = 512;                               // workspace sizing (fixed); the launch uses CTMI_LN_BWD_BLOCKS <= this
extern "C" int64_t ctmi_layernorm_bwd_ws(int64_t rows, int64_t cols) {
    (void)rows;
    return (int64_t)LN_BWD_MAX_BLOCKS * 4 * cols;          // up to 4 partial rows per block (dw, db, colsum(dres), colsum(dx))
}

// Main pass only: dx (+ dres) and the per-block partial rows ws[part][NS][cols] (NS = 2: dw, db; NS = 4: + colsum(dres),
// colsum(dx) when `want_sums` and the row width has the vector kernel).  The caller reduces the partial rows
// (ln_bwd_reduce* below, or one ctmi_reduce_jobs launch for a whole transformer block).
template <typename T>
static int ln_bwd_parts(const void* dy, const void* x, const float* w, const float* mean, const float* rstd,
                        const void* dres, void* dx, float* ws, int64_t rows, int64_t cols, int want_sums,
                        int* nparts_out, int* ns_out, hipStream_t st) {
    constexpr int VEC = 16 / sizeof(T);
    const bool vec_ok = (cols % VEC == 0) && aligned16(x) && aligned16(dy) && aligned16(dx) &&
                        (dres == nullptr || aligned16(dres)) && cols <= 8LL * 64 * VEC;
    int nparts, ns = 2;
    if (vec_ok) {
        const int mv = cols <= 64 * VEC ? 1 : (cols <= 2 * 64 * VEC ? 2 : (cols <= 4 * 64 * VEC ? 4 : 8));
        const int nw = lnb_waves(mv);
        int grid = (int)std::min<int64_t>(cdiv64(rows, nw), CTMI_LN_BWD_BLOCKS);
        if (mv >= 4) {                                                  // wide rows: one workgroup per row, columns split over 4 waves
            grid = (int)std::min<int64_t>(rows, LN_BWD_MAX_BLOCKS);
            if (mv == 4) hipLaunchKernelGGL((ln_bwd_wide<T, 1>), dim3(grid), dim3(256), 0, st, (const T*)dy, (const T*)x, w, mean, rstd, (const T*)dres, (T*)dx, ws, rows, (int)cols);
            else hipLaunchKernelGGL((ln_bwd_wide<T, 2>), dim3(grid), dim3(256), 0, st, (const T*)dy, (const T*)x, w, mean, rstd, (const T*)dres, (T*)dx, ws, rows, (int)cols);
        }
        nparts = grid;
        size_t lds = (size_t)cols * 2 * nw * sizeof(float);
#define LN_BWD_LAUNCH(MV, NSV, FL, RM) { \
            ctmi_dyn_lds(reinterpret_cast<const void*>(&ln_bwd_vec<T, MV, NSV, FL, RM>), lds); \
            hipLaunchKernelGGL((ln_bwd_vec<T, MV, NSV, FL, RM>), dim3(grid), dim3(64 * lnb_waves(MV)), lds, st, (const T*)dy, (const T*)x, w, mean, rstd, \
                               (const T*)dres, (T*)dx, ws, rows, (int)cols); }
#define LN_BWD_CASE(MV, NSV) { \
            if (cols == (int64_t)MV * 64 * VEC) { if (dres != nullptr) LN_BWD_LAUNCH(MV, NSV, true, 1) else LN_BWD_LAUNCH(MV, NSV, true, 0) } \
            else LN_BWD_LAUNCH(MV, NSV, false, -1) }
        if (mv >= 4) { (void)lds; }
        else if (want_sums) { ns = 4; if (mv == 1) LN_BWD_CASE(1, 4) else LN_BWD_CASE(2, 4) }
        else { if (mv == 1) LN_BWD_CASE(1, 2) else LN_BWD_CASE(2, 2) }
#undef LN_BWD_LAUNCH
#undef LN_BWD_CASE
        CTMI_CHECK_LAUNCH("layernorm_bwd");
    } else {
        int grid = (int)std::min<int64_t>(cdiv64(rows, 4), 2048);
        hipLaunchKernelGGL((ln_bwd_gen<T>), dim3(grid), dim3(256), 0, st, (const T*)dy, (const T*)x, w, mean, rstd,
                           (const T*)dres, (T*)dx, rows, cols);
        CTMI_CHECK_LAUNCH("layernorm_bwd");
        nparts = (int)std::min<int64_t>(rows, LN_BWD_MAX_BLOCKS);
        int64_t chunk = cdiv64(rows, nparts);
        nparts = (int)cdiv64(rows, chunk);
        hipLaunchKernelGGL((ln_bwd_wb_gen<T>), dim3(nparts), dim3(256), 0, st, (const T*)dy, (const T*)x, mean, rstd, ws, rows, cols, chunk);
        CTMI_CHECK_LAUNCH("layernorm_bwd_wb");
    }
    *nparts_out = nparts; *ns_out = ns;
    return CTMI_OK;
}

template <typename T>
static int ln_bwd_launch(const void* dy, const void* x, const float* w, const float* mean, const float* rstd,
                         const void* dres, void* dx, float* dw, float* db, int accumulate, float* ws,
                         int64_t rows, int64_t cols, hipStream_t st) {
    int nparts = 0, ns = 2;
    int rc = ln_bwd_parts<T>(dy, x, w, mean, rstd, dres, dx, ws, rows, cols, 0, &nparts, &ns, st);
    if (rc != CTMI_OK) return rc;
    if (cols % 256 == 0 && cols >= 2048 && aligned16(ws))
        hipLaunchKernelGGL(ln_bwd_reduce4, dim3((unsigned)(cols / 256), 2), dim3(256), 0, st, ws, dw, db, nparts, cols, accumulate);
    else
        hipLaunchKernelGGL(ln_bwd_reduce, dim3((unsigned)cdiv64(cols, 64), 2), dim3(256), 0, st, ws, dw, db, nparts, cols, accumulate);
    CTMI_CHECK_LAUNCH("layernorm_bwd_reduce");
    return CTMI_OK;
}

// internal (block.hip): main pass + partial rows only
int ctmi_ln_bwd_parts_internal(const void* dy, const void* x, const float* w, const float* mean, const float* rstd,
                               const void* dres, void* dx, float* ws, int64_t rows, int64_t cols, int dtype, int want_sums,
                               int* nparts, int* ns, hipStream_t st) {
    ProfScope prof__(CTMI_PROF_LAYERNORM, st);
    if (dtype == CTMI_F32) return ln_bwd_parts<float>(dy, x, w, mean, rstd, dres, dx, ws, rows, cols, want_sums, nparts, ns, st);
    if (dtype == CTMI_BF16) return ln_bwd_parts<bf16_t>(dy, x, w, mean, rstd, dres, dx, ws, rows, cols, want_sums, nparts, ns, st);
    if (dtype == CTMI_F16) return ln_bwd_parts<f16_t>(dy, x, w, mean, rstd, dres, dx, ws, rows, cols, want_sums, nparts, ns, st);
    ctmi_set_error("layernorm_bwd: unsupported dtype %d", dtype);
    return CTMI_ERR_UNSUPPORTED;
}

extern "C" int ctmi_layernorm_bwd(const void* dy, const void* x, const float* w, const float* mean, const float* rstd,
                                  const void* dres, void* dx, float* dw, float* db, int accumulate, float* ws,
                                  int64_t rows, int64_t cols, int dtype, void* stream) {
    ProfScope prof__(CTMI_PROF_LAYERNORM, as_stream(stream));
    CTMI_REQUIRE(dy && x && w && mean && rstd && dx && dw && db && ws, "layernorm_bwd: null pointer");
    CTMI_REQUIRE(rows > 0 && cols > 0, "layernorm_bwd: bad shape");
    if (dtype == CTMI_F32) return ln_bwd_launch<float>(dy, x, w, mean, rstd, dres, dx, dw, db, accumulate, ws, rows, cols, as_stream(stream));
    if (dtype == CTMI_BF16) return ln_bwd_launch<bf16_t>(dy, x, w, mean, rstd, dres, dx, dw, db, accumulate, ws, rows, cols, as_stream(stream));
    if (dtype == CTMI_F16) return ln_bwd_launch<f16_t>(dy, x, w, mean, rstd, dres, dx, dw, db, accumulate, ws, rows, cols, as_stream(stream));
    ctmi_set_error("layernorm_bwd: unsupported dtype %d", dtype);
    return CTMI_ERR_UNSUPPORTED;
}

// ------------------------------------------------------------------------------------------------
// column sum (bias gradients)
// ------------------------------------------------------------------------------------------------
static const int COLSUM_PARTS = 128;
extern "C" int64_t ctmi_colsum_ws(int64_t M, int64_t N) { (void)M; return (int64_t)COLSUM_PARTS * N; }

// grid (ceil(N/ (64*VEC)), parts): each block sums a slab of rows for 64*VEC columns, 4 waves over rows.
template <typename T>
__global__ __launch_bounds__(256) void colsum_part(const T* __restrict__ x, int64_t ld, float* __restrict__ ws,
                                                   int64_t M, int64_t N, int64_t rows_per_part, int vec_ok) {
    constexpr int VEC = 16 / sizeof(T);
    __shared__ float sm[4][64 * VEC];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int64_t c0 = ((int64_t)blockIdx.x * 64 + lane) * VEC;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_part, r1 = min(M, r0 + rows_per_part);
    float acc[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
    if (vec_ok) {
        if (c0 < N) {
            int64_t r = r0 + wid;
            for (; r + 12 < r1; r += 16) {                                   // 4 independent 16-byte loads in flight per lane
                uint4 q[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) q[u] = *reinterpret_cast<const uint4*>(x + (r + 4 * u) * ld + c0);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    float v[VEC];
                    unpack16<T>(q[u], v);
#pragma unroll
                    for (int j = 0; j < VEC; ++j) acc[j] += v[j];
                }
            }
            for (; r < r1; r += 4) {
                float v[VEC];
                unpack16<T>(*reinterpret_cast<const uint4*>(x + r * ld + c0), v);
#pragma unroll
                for (int j = 0; j < VEC; ++j) acc[j] += v[j];
            }
        }
    } else {
        for (int64_t r = r0 + wid; r < r1; r += 4)
#pragma unroll
            for (int j = 0; j < VEC; ++j) if (c0 + j < N) acc[j] += Cvt<T>::to_f(x[r * ld + c0 + j]);
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) sm[wid][lane * VEC + j] = acc[j];
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * VEC; i += 256) {
        const int64_t c = (int64_t)blockIdx.x * 64 * VEC + i;
        if (c < N) ws[(int64_t)blockIdx.y * N + c] = sm[0][i] + sm[1][i] + sm[2][i] + sm[3][i];
    }
}
// 64 columns x 4 part-slices per block, LDS combine (the parts loop is unrolled so its loads pipeline)
__global__ __launch_bounds__(256) void colsum_final(const float* __restrict__ ws, float* __restrict__ out, int parts,
                                                    int64_t N, int accumulate) {
    __shared__ float sm[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int64_t c = (int64_t)blockIdx.x * 64 + tx;
    float t = 0.f;
    if (c < N) {
#pragma unroll 8
        for (int p = ty; p < parts; p += 4) t += ws[(int64_t)p * N + c];
    }
    sm[ty][tx] = t;
    __syncthreads();
    if (ty == 0 && c < N) {
        const float r = sm[0][tx] + sm[1][tx] + sm[2][tx] + sm[3][tx];
        out[c] = accumulate ? out[c] + r : r;
    }
}

// partial rows only: ws[part][N]; *parts_out partial rows
int ctmi_colsum_parts_internal(const void* x, int64_t ld, float* ws, int64_t M, int64_t N, int dtype, int* parts_out, hipStream_t st) {
    ProfScope prof__(CTMI_PROF_REDUCE, st);
    const int64_t xblocks = cdiv64(N, 64 * (dtype == CTMI_F32 ? 4 : 8));
    // ~2048 workgroups (8 waves per SIMD) so the row streams cover the HBM latency; >= 16 rows per part
    int parts = (int)std::min<int64_t>(std::max<int64_t>(16, std::min<int64_t>(COLSUM_PARTS, cdiv64(2048, xblocks))), cdiv64(M, 16));
    int64_t rpp = cdiv64(M, parts);
    parts = (int)cdiv64(M, rpp);
    if (dtype == CTMI_F32) {
        int vec_ok = (N % 4 == 0) && (ld % 4 == 0) && aligned16(x);
        hipLaunchKernelGGL((colsum_part<float>), dim3((unsigned)cdiv64(N, 64 * 4), parts), dim3(256), 0, st, (const float*)x, ld, ws, M, N, rpp, vec_ok);
    } else if (dtype == CTMI_BF16) {
        int vec_ok = (N % 8 == 0) && (ld % 8 == 0) && aligned16(x);
        hipLaunchKernelGGL((colsum_part<bf16_t>), dim3((unsigned)cdiv64(N, 64 * 8), parts), dim3(256), 0, st, (const bf16_t*)x, ld, ws, M, N, rpp, vec_ok);
    } else if (dtype == CTMI_F16) {
        int vec_ok = (N % 8 == 0) && (ld % 8 == 0) && aligned16(x);
        hipLaunchKernelGGL((colsum_part<f16_t>), dim3((unsigned)cdiv64(N, 64 * 8), parts), dim3(256), 0, st, (const f16_t*)x, ld, ws, M, N, rpp, vec_ok);
    } else { ctmi_set_error("colsum: unsupported dtype %d", dtype); return CTMI_ERR_UNSUPPORTED; }
    CTMI_CHECK_LAUNCH("colsum_part");
    *parts_out = parts;
    return CTMI_OK;
}

extern "C" int ctmi_colsum(const void* x, int64_t ld, float* out, int accumulate, float* ws, int64_t M, int64_t N, int dtype, void* stream) {
    CTMI_REQUIRE(x && out && ws && M > 0 && N > 0, "colsum: bad args");
    hipStream_t st = as_stream(stream);
    int parts = 0;
    int rc = ctmi_colsum_parts_internal(x, ld, ws, M, N, dtype, &parts, st);
    if (rc != CTMI_OK) return rc;
    hipLaunchKernelGGL(colsum_final, dim3((unsigned)cdiv64(N, 64)), dim3(256), 0, st, ws, out, parts, N, accumulate);
    CTMI_CHECK_LAUNCH("colsum_final");
    return CTMI_OK;
}

// ------------------------------------------------------------------------------------------------
// activation dropout (hidden / embedding / residual-branch dropout of the reference's modules)
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void dropout_k(const T* __restrict__ x, const T* __restrict__ res, T* __restrict__ y, int64_t n,
                                                 uint32_t thr, uint32_t seed, float scale, int vec_ok) {
    constexpr int VEC = 16 / sizeof(T);
    const int64_t nv = vec_ok ? n / VEC : 0;
    for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < nv; v += (int64_t)gridDim.x * 256) {
        float a[VEC], r[VEC];
        unpack16<T>(*reinterpret_cast<const uint4*>(x + v * VEC), a);
        if (res != nullptr) unpack16<T>(*reinterpret_cast<const uint4*>(res + v * VEC), r);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const bool keep = ctmi_keep_hash((uint32_t)(v * VEC + j), seed) >= thr;
            a[j] = keep ? a[j] * scale : 0.f;
            if (res != nullptr) a[j] += r[j];
        }
        *reinterpret_cast<uint4*>(y + v * VEC) = pack16<T>(a);
    }
    for (int64_t i = nv * VEC + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const bool keep = ctmi_keep_hash((uint32_t)i, seed) >= thr;
        float a = keep ? Cvt<T>::to_f(x[i]) * scale : 0.f;
        if (res != nullptr) a += Cvt<T>::to_f(res[i]);
        y[i] = Cvt<T>::from_f(a);
    }
}
extern "C" uint32_t ctmi_dropout_hash(uint32_t x) { return ctmi_hash32(x); }
extern "C" uint32_t ctmi_dropout_keep_hash(uint32_t counter, uint32_t seed) { return ctmi_keep_hash(counter, seed); }
extern "C" uint32_t ctmi_dropout_threshold(float p) { return ctmi_drop_threshold(p); }
extern "C" int ctmi_dropout(const void* x, const void* residual, void* y, int64_t n, float p, uint32_t seed, int dtype, void* stream) {
    ProfScope prof__(CTMI_PROF_OTHER, as_stream(stream));
    CTMI_REQUIRE(x && y && n >= 0, "dropout: bad args");
    CTMI_REQUIRE(p >= 0.0f && p < 1.0f, "dropout: p must be in [0, 1)");
    if (n == 0) return CTMI_OK;
    const uint32_t thr = ctmi_drop_threshold(p);
    const float scale = 1.0f / (1.0f - p);
    const unsigned grid = (unsigned)std::min<int64_t>(cdiv64(n, 256 * 8), 8192);
    const int vec_ok = aligned16(x) && aligned16(y) && (residual == nullptr || aligned16(residual));
    if (dtype == CTMI_F32) hipLaunchKernelGGL((dropout_k<float>), dim3(grid), dim3(256), 0, as_stream(stream), (const float*)x, (const float*)residual, (float*)y, n, thr, seed, scale, vec_ok);
    else if (dtype == CTMI_BF16) hipLaunchKernelGGL((dropout_k<bf16_t>), dim3(grid), dim3(256), 0, as_stream(stream), (const bf16_t*)x, (const bf16_t*)residual, (bf16_t*)y, n, thr, seed, scale, vec_ok);
    else if (dtype == CTMI_F16) hipLaunchKernelGGL((dropout_k<f16_t>), dim3(grid), dim3(256), 0, as_stream(stream), (const f16_t*)x, (const f16_t*)residual, (f16_t*)y, n, thr, seed, scale, vec_ok);
    else { ctmi_set_error("dropout: unsupported dtype %d", dtype); return CTMI_ERR_UNSUPPORTED; }
    CTMI_CHECK_LAUNCH("dropout");
    return CTMI_OK;
}

// ------------------------------------------------------------------------------------------------
// multi-job partial-row reduction: dst[c] (+)= alpha * sum_p src[p * part_stride + c], c < n, for up to CTMI_REDUCE_MAX_JOBS
// independent jobs in ONE launch (fixed summation order: deterministic).  A transformer block's backward leaves the partial
// rows of its two LayerNorm backward passes and its bias column sums in workspaces and reduces them all here, instead of one
// small "final" launch per vector (what used to be 96 colsum_final + 50 ln_bwd_reduce launches per Bloom-560M step).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void reduce_jobs_k(ReduceJobs R) {
    __shared__ float sm[4][64];
    reduce_jobs_block(R, (int)blockIdx.x, sm);                            // (csrc/reduce_jobs.h: shared with gemm.hip's wgrad_tail_k)
}

extern "C" int ctmi_reduce_jobs(const ctmi_reduce_job* jobs, int count, void* stream) {
    ProfScope prof__(CTMI_PROF_REDUCE, as_stream(stream));
    CTMI_REQUIRE(jobs != nullptr && count >= 0, "reduce_jobs: bad args");
    hipStream_t st = as_stream(stream);
    for (int base = 0; base < count; base += CTMI_REDUCE_MAX_JOBS) {
        ReduceJobs R;
        const int chunks = reduce_jobs_pack(jobs + base, std::min(CTMI_REDUCE_MAX_JOBS, count - base), R);
        if (chunks < 0) return CTMI_ERR_ARG;
        hipLaunchKernelGGL(reduce_jobs_k, dim3((unsigned)chunks), dim3(256), 0, st, R);
        CTMI_CHECK_LAUNCH("reduce_jobs");
    }
    return CTMI_OK;
}

// ------------------------------------------------------------------------------------------------
// embedding gather / scatter-add   (modeling_bloom.py:190)
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void embed_fwd_k(const T* __restrict__ table, const int64_t* __restrict__ ids,
                                                   T* __restrict__ out, int64_t n, int64_t H, int64_t V, int vec_ok,
                                                   int32_t* err_flag) {
    constexpr int VEC = 16 / sizeof(T);
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    for (int64_t t = wave0; t < n; t += (int64_t)gridDim.x * 4) {
        int64_t id = ids[t];
        if (id < 0 || id >= V) { if (lane == 0 && err_flag) atomicExch(err_flag, 1); id = 0; }
        const T* src = table + id * H;
        T* dst = out + t * H;
        if (vec_ok) {
            for (int64_t c = (int64_t)lane * VEC; c < H; c += 64 * VEC)
                *reinterpret_cast<uint4*>(dst + c) = *reinterpret_cast<const uint4*>(src + c);
        } else {
            for (int64_t c = lane; c < H; c += 64) dst[c] = src[c];
        }
    }
}
template <typename T>
__global__ __launch_bounds__(256) void embed_bwd_k(const T* __restrict__ dout, const int64_t* __restrict__ ids,
                                                   float* __restrict__ dtable, int64_t n, int64_t H, int64_t V, float scale) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    for (int64_t t = wave0; t < n; t += (int64_t)gridDim.x * 4) {
        const int64_t id = ids[t];
        if (id < 0 || id >= V) continue;
        float* dst = dtable + id * H;
        const T* src = dout + t * H;
        for (int64_t c = lane; c < H; c += 64) unsafeAtomicAdd(dst + c, scale * Cvt<T>::to_f(src[c]));
    }
}

extern "C" int ctmi_embed_fwd(const void* table, const int64_t* ids, void* out, int64_t n, int64_t H, int64_t V,
                              int dtype, int32_t* err_flag, void* stream) {
    ProfScope prof__(CTMI_PROF_OTHER, as_stream(stream));
    CTMI_REQUIRE(table && ids && out && n >= 0 && H > 0 && V > 0, "embed_fwd: bad args");
    if (n == 0) return CTMI_OK;
    int grid = (int)std::min<int64_t>(cdiv64(n, 4), 4096);
    if (dtype == CTMI_F32) {
        int vec_ok = (H % 4 == 0) && aligned16(table) && aligned16(out);
        hipLaunchKernelGGL((embed_fwd_k<float>), dim3(grid), dim3(256), 0, as_stream(stream), (const float*)table, ids, (float*)out, n, H, V, vec_ok, err_flag);
    } else if (dtype == CTMI_BF16) {
        int vec_ok = (H % 8 == 0) && aligned16(table) && aligned16(out);
        hipLaunchKernelGGL((embed_fwd_k<bf16_t>), dim3(grid), dim3(256), 0, as_stream(stream), (const bf16_t*)table, ids, (bf16_t*)out, n, H, V, vec_ok, err_flag);
    } else if (dtype == CTMI_F16) {
        int vec_ok = (H % 8 == 0) && aligned16(table) && aligned16(out);
        hipLaunchKernelGGL((embed_fwd_k<f16_t>), dim3(grid), dim3(256), 0, as_stream(stream), (const f16_t*)table, ids, (f16_t*)out, n, H, V, vec_ok, err_flag);
    } else { ctmi_set_error("embed_fwd: unsupported dtype %d", dtype); return CTMI_ERR_UNSUPPORTED; }
    CTMI_CHECK_LAUNCH("embed_fwd");
    return CTMI_OK;
}
extern "C" int ctmi_embed_bwd(const void* dout, const int64_t* ids, float* dtable, int64_t n, int64_t H, int64_t V,
                              int dtype, float scale, void* stream) {
    ProfScope prof__(CTMI_PROF_OTHER, as_stream(stream));
    CTMI_REQUIRE(dout && ids && dtable && n >= 0 && H > 0 && V > 0, "embed_bwd: bad args");
    if (n == 0) return CTMI_OK;
    int grid = (int)std::min<int64_t>(cdiv64(n, 4), 4096);
    if (dtype == CTMI_F32) hipLaunchKernelGGL((embed_bwd_k<float>), dim3(grid), dim3(256), 0, as_stream(stream), (const float*)dout, ids, dtable, n, H, V, scale);
    else if (dtype == CTMI_BF16) hipLaunchKernelGGL((embed_bwd_k<bf16_t>), dim3(grid), dim3(256), 0, as_stream(stream), (const bf16_t*)dout, ids, dtable, n, H, V, scale);
    else if (dtype == CTMI_F16) hipLaunchKernelGGL((embed_bwd_k<f16_t>), dim3(grid), dim3(256), 0, as_stream(stream), (const f16_t*)dout, ids, dtable, n, H, V, scale);
    else { ctmi_set_error("embed_bwd: unsupported dtype %d", dtype); return CTMI_ERR_UNSUPPORTED; }
    CTMI_CHECK_LAUNCH("embed_bwd");
    return CTMI_OK;
}

// ------------------------------------------------------------------------------------------------
// cross entropy  (modeling_bloom.py:224-230; loss.py:29-49 in its numerically stable form)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int64_t ce_target(const int64_t* labels, int64_t row, int64_t seq, int64_t shift, int64_t ignore) {
    const int64_t b = row / seq, s = row - b * seq;
    if (s + shift >= seq) return -1;
    const int64_t t = labels[b * seq + s + shift];
    return (t == ignore) ? -1 : t;
}

// combine two (max, sum) online-softmax states
__device__ __forceinline__ void lse_merge(float& m, float& s, float m2, float s2) {
    const float mn = fmaxf(m, m2);
    s = s * __expf(m - mn) + s2 * __expf(m2 - mn);
    m = mn;
}

// Streaming (non-temporal) 16-byte accesses for single-pass kernels whose data is far larger than the 256 MiB Infinity Cache
// (the optimizer walks 16.8 GB per step, the loss backward 8.2 GB): no line is ever reused, so none should be kept.
typedef float f32x4_nt __attribute__((ext_vector_type(4)));
typedef unsigned u32x4_nt __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ldg_stream(const float* p) {
    const f32x4_nt v = __builtin_nontemporal_load(reinterpret_cast<const f32x4_nt*>(p));
    return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void stg_stream(float* p, const float4& v) {
    __builtin_nontemporal_store(f32x4_nt{v.x, v.y, v.z, v.w}, reinterpret_cast<f32x4_nt*>(p));
}

__device__ __forceinline__ uint4 ldg_stream16(const void* p) {
    const u32x4_nt v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_nt*>(p));
    return make_uint4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void stg_stream16(void* p, const uint4& v) {
    __builtin_nontemporal_store(u32x4_nt{v.x, v.y, v.z, v.w}, reinterpret_cast<u32x4_nt*>(p));
}

template <typename T>
__global__ __launch_bounds__(256) void ce_fwd_k(const T* __restrict__ logits, int64_t ld, const int64_t* __restrict__ labels,
                                                float* __restrict__ row_lse, float* __restrict__ row_loss,
                                                int64_t C, int64_t seq, int64_t shift, int64_t ignore, int vec_ok) {
    constexpr int VEC = 16 / sizeof(T);
    const int64_t row = blockIdx.x;
    const T* x = logits + row * ld;
    float m = -INFINITY, s = 0.f;
    if (vec_ok) {
        const int64_t Cv = C - C % VEC;                                 // odd class counts (GPT-2: 50257): vector head + scalar tail
        for (int64_t c = Cv + threadIdx.x; c < C; c += 256) {
            const float v = Cvt<T>::to_f(x[c]);
            const float mn = fmaxf(m, v);
            s = s * __expf(m - mn) + __expf(v - mn);
            m = mn;
        }
        for (int64_t c = (int64_t)threadIdx.x * VEC; c < Cv; c += 256 * VEC) {
            float v[VEC];
            unpack16<T>(*reinterpret_cast<const uint4*>(x + c), v);
            float vm = v[0];
#pragma unroll
            for (int j = 1; j < VEC; ++j) vm = fmaxf(vm, v[j]);
            const float mn = fmaxf(m, vm);
            float add = 0.f;
#pragma unroll
            for (int j = 0; j < VEC; ++j) add += __expf(v[j] - mn);
            s = s * __expf(m - mn) + add;
            m = mn;
        }
    } else {
        for (int64_t c = threadIdx.x; c < C; c += 256) {
            const float v = Cvt<T>::to_f(x[c]);
            const float mn = fmaxf(m, v);
            s = s * __expf(m - mn) + __expf(v - mn);
            m = mn;
        }
    }
    // threads that saw nothing hold (m=-inf, s=0): __expf(-inf - mn) = 0 keeps them neutral as long as mn is finite
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float m2 = __shfl_xor(m, o, 64), s2 = __shfl_xor(s, o, 64);
        const float mn = fmaxf(m, m2);
        const float a = (m == -INFINITY) ? 0.f : s * __expf(m - mn);
        const float b = (m2 == -INFINITY) ? 0.f : s2 * __expf(m2 - mn);
        s = a + b; m = mn;
    }
    __shared__ float sm_m[4], sm_s[4];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) { sm_m[wid] = m; sm_s[wid] = s; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float M = sm_m[0], S = sm_s[0];
        for (int k = 1; k < 4; ++k) {
            const float m2 = sm_m[k], s2 = sm_s[k];
            const float mn = fmaxf(M, m2);
            const float a = (M == -INFINITY) ? 0.f : S * __expf(M - mn);
            const float b = (m2 == -INFINITY) ? 0.f : s2 * __expf(m2 - mn);
            S = a + b; M = mn;
        }
        const float lse = M + logf(S);
        row_lse[row] = lse;
        const int64_t t = ce_target(labels, row, seq, shift, ignore);
        row_loss[row] = (t >= 0 && t < C) ? (lse - Cvt<T>::to_f(x[t])) : -1.0f;     // -1 marks "no loss"
    }
}

// single block: loss_out[0] = sum(valid row_loss)/denom ; loss_out[1] = 1/denom
__global__ __launch_bounds__(1024) void ce_finalize_k(const float* __restrict__ row_loss, float* __restrict__ loss_out,
                                                      int64_t N, int denom_mode, int64_t denom_rows) {
    __shared__ double sm_s[16];
    __shared__ unsigned long long sm_c[16];
    double s = 0.0; unsigned long long cnt = 0;
    for (int64_t r = threadIdx.x; r < N; r += 1024) {
        const float l = row_loss[r];
        if (l >= 0.f) { s += (double)l; cnt += 1; }
    }
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); cnt += __shfl_xor(cnt, o, 64); }
    if ((threadIdx.x & 63) == 0) { sm_s[threadIdx.x >> 6] = s; sm_c[threadIdx.x >> 6] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double S = 0.0; unsigned long long Cn = 0;
        for (int k = 0; k < 16; ++k) { S += sm_s[k]; Cn += sm_c[k]; }
        double denom = denom_mode == 0 ? (double)Cn : (denom_mode == 1 ? (double)denom_rows : 1.0);
        loss_out[0] = (float)(S / denom);                   // 0/0 -> NaN exactly as torch's mean over zero rows
        loss_out[1] = (float)(1.0 / denom);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void ce_bwd_k(const T* __restrict__ logits, int64_t ld, const int64_t* __restrict__ labels,
                                                const float* __restrict__ row_lse, const float* __restrict__ loss_out,
                                                const float* __restrict__ gout, T* __restrict__ dlogits, int64_t ldd,
                                                int64_t C, int64_t seq, int64_t shift, int64_t ignore, int vec_ok) {
    constexpr int VEC = 16 / sizeof(T);
    const int64_t row = blockIdx.x;
    const T* x = logits + row * ld;
    T* d = dlogits + row * ldd;
    const int64_t t = ce_target(labels, row, seq, shift, ignore);
    const bool live = (t >= 0 && t < C);
    const float coef = live ? (gout ? gout[0] : 1.0f) * loss_out[1] : 0.f;
    const float lse = row_lse[row];
    if (vec_ok) {
        const int64_t Cv = C - C % VEC;
        for (int64_t c = Cv + threadIdx.x; c < C; c += 256)
            d[c] = Cvt<T>::from_f(live ? (__expf(Cvt<T>::to_f(x[c]) - lse) - ((c == t) ? 1.0f : 0.0f)) * coef : 0.f);
        for (int64_t c = (int64_t)threadIdx.x * VEC; c < Cv; c += 256 * VEC) {
            float o[VEC];
            if (live) {
                float v[VEC];
                unpack16<T>(ldg_stream16(x + c), v);                      // logits: last use
#pragma unroll
                for (int j = 0; j < VEC; ++j) o[j] = (__expf(v[j] - lse) - ((c + j == t) ? 1.0f : 0.0f)) * coef;
            } else {
#pragma unroll
                for (int j = 0; j < VEC; ++j) o[j] = 0.f;
            }
            stg_stream16(d + c, pack16<T>(o));                            // 4.1 GB of dlogits: far beyond any cache
        }
    } else {
        for (int64_t c = threadIdx.x; c < C; c += 256)
            d[c] = Cvt<T>::from_f(live ? (__expf(Cvt<T>::to_f(x[c]) - lse) - ((c == t) ? 1.0f : 0.0f)) * coef : 0.f);
    }
}

extern "C" int ctmi_ce_fwd(const void* logits, int64_t ld, const int64_t* labels, float* row_lse, float* row_loss,
                           float* loss_out, int64_t N, int64_t C, int64_t seq, int64_t shift, int64_t ignore_index,
                           int denom_mode, int64_t denom_rows, int dtype, void* stream) {
    ProfScope prof__(CTMI_PROF_LOSS, as_stream(stream));
    CTMI_REQUIRE(logits && labels && row_lse && row_loss && loss_out, "ce_fwd: null pointer");
    CTMI_REQUIRE(N > 0 && C > 0 && seq > 0 && N % seq == 0 && shift >= 0 && ld >= C, "ce_fwd: bad shape N=%lld C=%lld seq=%lld", (long long)N, (long long)C, (long long)seq);
    hipStream_t st = as_stream(stream);
    if (dtype == CTMI_F32) {
        int vec_ok = (ld % 4 == 0) && aligned16(logits);
        hipLaunchKernelGGL((ce_fwd_k<float>), dim3((unsigned)N), dim3(256), 0, st, (const float*)logits, ld, labels, row_lse, row_loss, C, seq, shift, ignore_index, vec_ok);
    } else if (dtype == CTMI_BF16) {
        int vec_ok = (ld % 8 == 0) && aligned16(logits);
        hipLaunchKernelGGL((ce_fwd_k<bf16_t>), dim3((unsigned)N), dim3(256), 0, st, (const bf16_t*)logits, ld, labels, row_lse, row_loss, C, seq, shift, ignore_index, vec_ok);
    } else if (dtype == CTMI_F16) {
        int vec_ok = (ld % 8 == 0) && aligned16(logits);
        hipLaunchKernelGGL((ce_fwd_k<f16_t>), dim3((unsigned)N), dim3(256), 0, st, (const f16_t*)logits, ld, labels, row_lse, row_loss, C, seq, shift, ignore_index, vec_ok);
    } else { ctmi_set_error("ce_fwd: unsupported dtype %d", dtype); return CTMI_ERR_UNSUPPORTED; }
    CTMI_CHECK_LAUNCH("ce_fwd");
    hipLaunchKernelGGL(ce_finalize_k, dim3(1), dim3(1024), 0, st, row_loss, loss_out, N, denom_mode, denom_rows);
    CTMI_CHECK_LAUNCH("ce_finalize");
    return CTMI_OK;
}

extern "C" int ctmi_ce_bwd(const void* logits, int64_t ld, const int64_t* labels, const float* row_lse, const float* loss_out,
                           const float* gout, void* dlogits, int64_t ldd, int64_t N, int64_t C, int64_t seq, int64_t shift,
                           int64_t ignore_index, int dtype, void* stream) {
    ProfScope prof__(CTMI_PROF_LOSS, as_stream(stream));
    CTMI_REQUIRE(logits && labels && row_lse && loss_out && dlogits, "ce_bwd: null pointer");
    CTMI_REQUIRE(N > 0 && C > 0 && seq > 0 && N % seq == 0 && ld >= C && ldd >= C, "ce_bwd: bad shape");
    hipStream_t st = as_stream(stream);
    if (dtype == CTMI_F32) {
        int vec_ok = (ld % 4 == 0) && (ldd % 4 == 0) && aligned16(logits) && aligned16(dlogits);
        hipLaunchKernelGGL((ce_bwd_k<float>), dim3((unsigned)N), dim3(256), 0, st, (const float*)logits, ld, labels, row_lse, loss_out, gout, (float*)dlogits, ldd, C, seq, shift, ignore_index, vec_ok);
    } else if (dtype == CTMI_BF16) {
        int vec_ok = (ld % 8 == 0) && (ldd % 8 == 0) && aligned16(logits) && aligned16(dlogits);
        hipLaunchKernelGGL((ce_bwd_k<bf16_t>), dim3((unsigned)N), dim3(256), 0, st, (const bf16_t*)logits, ld, labels, row_lse, loss_out, gout, (bf16_t*)dlogits, ldd, C, seq, shift, ignore_index, vec_ok);
    } else if (dtype == CTMI_F16) {
        int vec_ok = (ld % 8 == 0) && (ldd % 8 == 0) && aligned16(logits) && aligned16(dlogits);
        hipLaunchKernelGGL((ce_bwd_k<f16_t>), dim3((unsigned)N), dim3(256), 0, st, (const f16_t*)logits, ld, labels, row_lse, loss_out, gout, (f16_t*)dlogits, ldd, C, seq, shift, ignore_index, vec_ok);
    } else { ctmi_set_error("ce_bwd: unsupported dtype %d", dtype); return CTMI_ERR_UNSUPPORTED; }
    CTMI_CHECK_LAUNCH("ce_bwd");
    return CTMI_OK;
}

// ---- training path: loss AND its gradient in one pass over the logits (modeling_bloom.py:224-230 followed by loss.backward()).
// The two-kernel form reads the 4.1 GB of bf16 logits twice from HBM (statistics, then gradient).  Here ONE 1024-thread
// workgroup owns a row: pass 1 streams the row (502 KB at V = 250880) and forms its log-sum-exp, pass 2 walks the SAME row
// back to front — the most recently fetched lines first — and writes dlogits.  One workgroup per CU (the dynamic LDS request
// enforces it) keeps 256 rows = 128 MiB in flight, inside the 256 MiB Infinity Cache, so the second read does not go to HBM.
// The gradient is written for an upstream gradient of 1 (loss.backward()); ctmi_scale_if rescales it in the rare other case.
// 1/denom is needed before any row is finished, so it is counted from the labels alone by ce_count_k first.
__global__ __launch_bounds__(1024) void ce_count_k(const int64_t* __restrict__ labels, float* __restrict__ loss_out, int64_t N, int64_t C,
                                                   int64_t seq, int64_t shift, int64_t ignore, int denom_mode, int64_t denom_rows) {
    __shared__ unsigned long long sm_c[16];
    unsigned long long cnt = 0;
    for (int64_t r = threadIdx.x; r < N; r += 1024) {
        const int64_t t = ce_target(labels, r, seq, shift, ignore);
        cnt += (t >= 0 && t < C) ? 1 : 0;
    }
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
    if ((threadIdx.x & 63) == 0) sm_c[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long Cn = 0;
        for (int k = 0; k < 16; ++k) Cn += sm_c[k];
        const double denom = denom_mode == 0 ? (double)Cn : (denom_mode == 1 ? (double)denom_rows : 1.0);
        loss_out[1] = (float)(1.0 / denom);
    }
}

template <typename T>
__global__ __launch_bounds__(1024) void ce_fused_k(const T* __restrict__ logits, int64_t ld, const int64_t* __restrict__ labels,
                                                   float* __restrict__ row_lse, float* __restrict__ row_loss,
                                                   const float* __restrict__ loss_out, T* __restrict__ dlogits, int64_t ldd,
                                                   int64_t C, int64_t seq, int64_t shift, int64_t ignore,
                                                   float gfac, const float* __restrict__ gdev) {
    constexpr int VEC = 16 / sizeof(T);
    constexpr int NT = 1024, UNR = 4;
    __shared__ float sm_m[16], sm_s[16];
    const int64_t row = blockIdx.x;
    const T* x = logits + row * ld;
    T* d = dlogits + row * ldd;
    const int64_t Cv = C - C % VEC;                                     // vector body (rows are 16-byte aligned: checked by the host)
    const int64_t nvec = Cv / VEC;                                      // 16-byte chunks in the row
    float m = -INFINITY, s = 0.f;
    // ---- pass 1: online (max, sum exp), UNR independent 16-byte loads in flight per lane
    for (int64_t c = Cv + threadIdx.x; c < C; c += NT) {
        const float v = Cvt<T>::to_f(x[c]);
        const float mn = fmaxf(m, v);
        s = s * __expf(m - mn) + __expf(v - mn);
        m = mn;
    }
    int64_t i = threadIdx.x;
    for (; i + (UNR - 1) * NT < nvec; i += UNR * NT) {
        uint4 raw[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) raw[u] = *reinterpret_cast<const uint4*>(x + (i + u * NT) * VEC);
        float vm = -INFINITY;
        float v[UNR][VEC];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            unpack16<T>(raw[u], v[u]);
#pragma unroll
            for (int j = 0; j < VEC; ++j) vm = fmaxf(vm, v[u][j]);
        }
        const float mn = fmaxf(m, vm);
        float add = 0.f;
#pragma unroll
        for (int u = 0; u < UNR; ++u)
#pragma unroll
            for (int j = 0; j < VEC; ++j) add += __expf(v[u][j] - mn);
        s = s * __expf(m - mn) + add;
        m = mn;
    }
    for (; i < nvec; i += NT) {
        float v[VEC];
        unpack16<T>(*reinterpret_cast<const uint4*>(x + i * VEC), v);
        float vm = v[0];
#pragma unroll
        for (int j = 1; j < VEC; ++j) vm = fmaxf(vm, v[j]);
        const float mn = fmaxf(m, vm);
        float add = 0.f;
#pragma unroll
        for (int j = 0; j < VEC; ++j) add += __expf(v[j] - mn);
        s = s * __expf(m - mn) + add;
        m = mn;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float m2 = __shfl_xor(m, o, 64), s2 = __shfl_xor(s, o, 64);
        const float mn = fmaxf(m, m2);
        const float a = (m == -INFINITY) ? 0.f : s * __expf(m - mn);
        const float b = (m2 == -INFINITY) ? 0.f : s2 * __expf(m2 - mn);
        s = a + b; m = mn;
    }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) { sm_m[wid] = m; sm_s[wid] = s; }
    __syncthreads();
    float M = sm_m[0], S = sm_s[0];                                     // every thread folds the 16 wave states in the same order
#pragma unroll
    for (int k = 1; k < 16; ++k) {
        const float m2 = sm_m[k], s2 = sm_s[k];
        const float mn = fmaxf(M, m2);
        const float a = (M == -INFINITY) ? 0.f : S * __expf(M - mn);
        const float b = (m2 == -INFINITY) ? 0.f : s2 * __expf(m2 - mn);
        S = a + b; M = mn;
    }
    const float lse = M + logf(S);
    const int64_t t = ce_target(labels, row, seq, shift, ignore);
    const bool live = (t >= 0 && t < C);
    if (threadIdx.x == 0) {
        row_lse[row] = lse;
        row_loss[row] = live ? (lse - Cvt<T>::to_f(x[t])) : -1.0f;       // -1 marks "no loss"
    }
    // ---- pass 2: gradient, back to front (the tail of the row is the freshest in the caches)
    // the upstream gradient the caller EXPECTS (1/accumulation steps, a loss scale: gfac * gdev[0]) is folded in here, in fp32, before the one
    // rounding to the storage type; the backward then rescales only if the actual upstream gradient differs (ctmi_scale_if)
    const float coef = live ? loss_out[1] * (gdev != nullptr ? gfac * gdev[0] : gfac) : 0.f;
    for (int64_t c = Cv + threadIdx.x; c < C; c += NT)
        d[c] = Cvt<T>::from_f(live ? (__expf(Cvt<T>::to_f(x[c]) - lse) - ((c == t) ? 1.0f : 0.0f)) * coef : 0.f);
    if (!live) {
        const uint4 z = make_uint4(0u, 0u, 0u, 0u);
        for (int64_t k = threadIdx.x; k < nvec; k += NT) stg_stream16(d + k * VEC, z);
        return;
    }
    int64_t k = nvec - 1 - threadIdx.x;
    for (; k - (UNR - 1) * NT >= 0; k -= UNR * NT) {
        uint4 raw[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) raw[u] = ldg_stream16(x + (k - u * NT) * VEC);       // logits: last use
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int64_t c = (k - u * NT) * VEC;
            float v[VEC], o[VEC];
            unpack16<T>(raw[u], v);
#pragma unroll
            for (int j = 0; j < VEC; ++j) o[j] = (__expf(v[j] - lse) - ((c + j == t) ? 1.0f : 0.0f)) * coef;
            stg_stream16(d + c, pack16<T>(o));
        }
    }
    for (; k >= 0; k -= NT) {
        const int64_t c = k * VEC;
        float v[VEC], o[VEC];
        unpack16<T>(ldg_stream16(x + c), v);
#pragma unroll
        for (int j = 0; j < VEC; ++j) o[j] = (__expf(v[j] - lse) - ((c + j == t) ? 1.0f : 0.0f)) * coef;
        stg_stream16(d + c, pack16<T>(o));
    }
}

extern "C" int ctmi_ce_fwd_bwd(const void* logits, int64_t ld, const int64_t* labels, float* row_lse, float* row_loss,
                               float* loss_out, void* dlogits, int64_t ldd, int64_t N, int64_t C, int64_t seq, int64_t shift,
                               int64_t ignore_index, int denom_mode, int64_t denom_rows, float grad_factor, const float* grad_factor_dev,
                               int dtype, void* stream) {
    ProfScope prof__(CTMI_PROF_LOSS, as_stream(stream));
    CTMI_REQUIRE(logits && labels && row_lse && row_loss && loss_out && dlogits, "ce_fwd_bwd: null pointer");
    CTMI_REQUIRE(N > 0 && C > 0 && seq > 0 && N % seq == 0 && shift >= 0 && ld >= C && ldd >= C, "ce_fwd_bwd: bad shape N=%lld C=%lld seq=%lld", (long long)N, (long long)C, (long long)seq);
    CTMI_REQUIRE(dtype == CTMI_F32 || dtype == CTMI_BF16 || dtype == CTMI_F16, "ce_fwd_bwd: unsupported dtype %d", dtype);
    const int vec = dtype == CTMI_F32 ? 4 : 8;
    CTMI_REQUIRE(ld % vec == 0 && ldd % vec == 0 && aligned16(logits) && aligned16(dlogits),
                 "ce_fwd_bwd: rows must be 16-byte aligned (use ctmi_ce_fwd + ctmi_ce_bwd otherwise)");
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(ce_count_k, dim3(1), dim3(1024), 0, st, labels, loss_out, N, C, seq, shift, ignore_index, denom_mode, denom_rows);
    CTMI_CHECK_LAUNCH("ce_count");
    const size_t lds_hold = 96 * 1024;                                   // one workgroup per CU: 256 rows in flight (see above)
    if (dtype == CTMI_F32) {
        auto kern = &ce_fused_k<float>;
        ctmi_dyn_lds(reinterpret_cast<const void*>(kern), lds_hold);
        hipLaunchKernelGGL(kern, dim3((unsigned)N), dim3(1024), lds_hold, st, (const float*)logits, ld, labels, row_lse, row_loss, loss_out, (float*)dlogits, ldd, C, seq, shift, ignore_index, grad_factor, grad_factor_dev);
    } else if (dtype == CTMI_F16) {
        auto kern = &ce_fused_k<f16_t>;
        ctmi_dyn_lds(reinterpret_cast<const void*>(kern), lds_hold);
        hipLaunchKernelGGL(kern, dim3((unsigned)N), dim3(1024), lds_hold, st, (const f16_t*)logits, ld, labels, row_lse, row_loss, loss_out, (f16_t*)dlogits, ldd, C, seq, shift, ignore_index, grad_factor, grad_factor_dev);
    } else {
        auto kern = &ce_fused_k<bf16_t>;
        ctmi_dyn_lds(reinterpret_cast<const void*>(kern), lds_hold);
        hipLaunchKernelGGL(kern, dim3((unsigned)N), dim3(1024), lds_hold, st, (const bf16_t*)logits, ld, labels, row_lse, row_loss, loss_out, (bf16_t*)dlogits, ldd, C, seq, shift, ignore_index, grad_factor, grad_factor_dev);
    }
    CTMI_CHECK_LAUNCH("ce_fused");
    hipLaunchKernelGGL(ce_finalize_k, dim3(1), dim3(1024), 0, st, row_loss, loss_out, N, denom_mode, denom_rows);
    CTMI_CHECK_LAUNCH("ce_finalize");
    return CTMI_OK;
}

// x *= s[0] / (applied * applied_dev[0]) unless that ratio is exactly 1 (then every workgroup returns after two scalar loads): the backward of
// the fused loss, whose gradient was written for the EXPECTED upstream gradient applied * applied_dev[0] (ctmi_ce_fwd_bwd's grad_factor pair;
// 1 by default).  g_scale_if_passes counts the calls that really rescaled (tests assert "no second pass over [T,V]" with it).
static __device__ unsigned long long g_scale_if_passes;
template <typename T>
__global__ __launch_bounds__(256) void scale_if_k(T* __restrict__ x, int64_t ld, int64_t rows, int64_t cols, const float* __restrict__ sp,
                                                  float applied, const float* __restrict__ applied_dev) {
    const float want = sp[0], have = applied_dev != nullptr ? applied * applied_dev[0] : applied;
    if (want == have) return;
    const float s = want / have;
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&g_scale_if_passes, 1ULL);
    for (int64_t r = blockIdx.x; r < rows; r += gridDim.x)
        for (int64_t c = threadIdx.x; c < cols; c += 256) x[r * ld + c] = Cvt<T>::from_f(Cvt<T>::to_f(x[r * ld + c]) * s);
}
extern "C" int ctmi_scale_if(void* x, int64_t ld, int64_t rows, int64_t cols, const float* s_dev, float applied, const float* applied_dev,
                             int dtype, void* stream) {
    ProfScope prof__(CTMI_PROF_LOSS, as_stream(stream));
    CTMI_REQUIRE(x && s_dev && rows > 0 && cols > 0 && ld >= cols, "scale_if: bad args");
    CTMI_REQUIRE(applied != 0.f, "scale_if: the factor already applied must not be 0");
    const unsigned grid = (unsigned)std::min<int64_t>(rows, 4096);
    if (dtype == CTMI_F32) hipLaunchKernelGGL((scale_if_k<float>), dim3(grid), dim3(256), 0, as_stream(stream), (float*)x, ld, rows, cols, s_dev, applied, applied_dev);
    else if (dtype == CTMI_BF16) hipLaunchKernelGGL((scale_if_k<bf16_t>), dim3(grid), dim3(256), 0, as_stream(stream), (bf16_t*)x, ld, rows, cols, s_dev, applied, applied_dev);
    else if (dtype == CTMI_F16) hipLaunchKernelGGL((scale_if_k<f16_t>), dim3(grid), dim3(256), 0, as_stream(stream), (f16_t*)x, ld, rows, cols, s_dev, applied, applied_dev);
    else { ctmi_set_error("scale_if: unsupported dtype %d", dtype); return CTMI_ERR_UNSUPPORTED; }
    CTMI_CHECK_LAUNCH("scale_if");
    return CTMI_OK;
}
extern "C" int64_t ctmi_scale_if_passes(void) {
    unsigned long long v = 0;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_scale_if_passes), sizeof(v), 0, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return (int64_t)v;
}

// ---- probability targets (the second branch of loss.py:43-46): loss = -sum_{n,c} t[n,c] * log_softmax(x)[n,c]
//      = sum_n ( lse_n * sum_c t[n,c] - sum_c t[n,c] x[n,c] );   dx[n,c] = (softmax[n,c] * sum_c t[n,c] - t[n,c]) * g / denom.
// One workgroup per row, fp32 statistics; the row sums of t are kept for the backward.  Not on the SFT path (general widths,
// scalar loads): the kernels exist so that the module is complete, not to be fast.
template <typename T>
__global__ __launch_bounds__(256) void ce_soft_fwd_k(const T* __restrict__ logits, int64_t ld, const float* __restrict__ target, int64_t ldt,
                                                     float* __restrict__ row_lse, float* __restrict__ row_tsum,
                                                     float* __restrict__ row_loss, int64_t C) {
    const int64_t row = blockIdx.x;
    const T* x = logits + row * ld;
    const float* t = target + row * ldt;
    float m = -INFINITY;
    for (int64_t c = threadIdx.x; c < C; c += 256) m = fmaxf(m, Cvt<T>::to_f(x[c]));
    __shared__ float red[3][4];
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) red[0][threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3]));
    __syncthreads();
    float s = 0.f, ts = 0.f, tx = 0.f;
    for (int64_t c = threadIdx.x; c < C; c += 256) {
        const float v = Cvt<T>::to_f(x[c]), tv = t[c];
        s += expf(v - m); ts += tv; tx += tv * v;
    }
    s = wave_sum(s); ts = wave_sum(ts); tx = wave_sum(tx);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s; red[1][threadIdx.x >> 6] = ts; red[2][threadIdx.x >> 6] = tx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float S = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        const float TS = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
        const float TX = (red[2][0] + red[2][1]) + (red[2][2] + red[2][3]);
        const float lse = m + logf(S);
        row_lse[row] = lse; row_tsum[row] = TS; row_loss[row] = lse * TS - TX;
    }
}
// single block: loss_out[0] = sum(row_loss)/denom over ALL rows (a soft-target row loss may be negative), loss_out[1] = 1/denom
__global__ __launch_bounds__(1024) void ce_soft_finalize_k(const float* __restrict__ row_loss, float* __restrict__ loss_out, int64_t N, double denom) {
    __shared__ double sm_s[16];
    double s = 0.0;
    for (int64_t r = threadIdx.x; r < N; r += 1024) s += (double)row_loss[r];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) sm_s[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double S = 0.0;
        for (int k = 0; k < 16; ++k) S += sm_s[k];
        loss_out[0] = (float)(S / denom); loss_out[1] = (float)(1.0 / denom);
    }
}
template <typename T>
__global__ __launch_bounds__(256) void ce_soft_bwd_k(const T* __restrict__ logits, int64_t ld, const float* __restrict__ target, int64_t ldt,
                                                     const float* __restrict__ row_lse, const float* __restrict__ row_tsum,
                                                     const float* __restrict__ loss_out, const float* __restrict__ gout,
                                                     T* __restrict__ dlogits, int64_t ldd, int64_t C) {
    const int64_t row = blockIdx.x;
    const float coef = (gout ? gout[0] : 1.0f) * loss_out[1], lse = row_lse[row], ts = row_tsum[row];
    for (int64_t c = threadIdx.x; c < C; c += 256)
        dlogits[row * ldd + c] = Cvt<T>::from_f((expf(Cvt<T>::to_f(logits[row * ld + c]) - lse) * ts - target[row * ldt + c]) * coef);
}
extern "C" int ctmi_ce_soft_fwd(const void* logits, int64_t ld, const float* target, int64_t ldt, float* row_lse, float* row_tsum,
                                float* row_loss, float* loss_out, int64_t N, int64_t C, int denom_mode, int64_t denom_rows, int dtype,
                                void* stream) {
    ProfScope prof__(CTMI_PROF_LOSS, as_stream(stream));
    CTMI_REQUIRE(logits && target && row_lse && row_tsum && row_loss && loss_out, "ce_soft_fwd: null pointer");
    CTMI_REQUIRE(N > 0 && C > 0 && ld >= C && ldt >= C && (denom_mode == 1 || denom_mode == 2), "ce_soft_fwd: bad args");
    hipStream_t st = as_stream(stream);
    if (dtype == CTMI_F32) hipLaunchKernelGGL((ce_soft_fwd_k<float>), dim3((unsigned)N), dim3(256), 0, st, (const float*)logits, ld, target, ldt, row_lse, row_tsum, row_loss, C);
    else if (dtype == CTMI_BF16) hipLaunchKernelGGL((ce_soft_fwd_k<bf16_t>), dim3((unsigned)N), dim3(256), 0, st, (const bf16_t*)logits, ld, target, ldt, row_lse, row_tsum, row_loss, C);
    else if (dtype == CTMI_F16) hipLaunchKernelGGL((ce_soft_fwd_k<f16_t>), dim3((unsigned)N), dim3(256), 0, st, (const f16_t*)logits, ld, target, ldt, row_lse, row_tsum, row_loss, C);
    else { ctmi_set_error("ce_soft_fwd: unsupported dtype %d", dtype); return CTMI_ERR_UNSUPPORTED; }
    CTMI_CHECK_LAUNCH("ce_soft_fwd");
    hipLaunchKernelGGL(ce_soft_finalize_k, dim3(1), dim3(1024), 0, st, row_loss, loss_out, N, denom_mode == 1 ? (double)denom_rows : 1.0);
    CTMI_CHECK_LAUNCH("ce_soft_finalize");
    return CTMI_OK;
}
extern "C" int ctmi_ce_soft_bwd(const void* logits, int64_t ld, const float* target, int64_t ldt, const float* row_lse,
                                const float* row_tsum, const float* loss_out, const float* gout, void* dlogits, int64_t ldd,
                                int64_t N, int64_t C, int dtype, void* stream) {
    ProfScope prof__(CTMI_PROF_LOSS, as_stream(stream));
    CTMI_REQUIRE(logits && target && row_lse && row_tsum && loss_out && dlogits, "ce_soft_bwd: null pointer");
    CTMI_REQUIRE(N > 0 && C > 0 && ld >= C && ldt >= C && ldd >= C, "ce_soft_bwd: bad shape");
    hipStream_t st = as_stream(stream);
    if (dtype == CTMI_F32) hipLaunchKernelGGL((ce_soft_bwd_k<float>), dim3((unsigned)N), dim3(256), 0, st, (const float*)logits, ld, target, ldt, row_lse, row_tsum, loss_out, gout, (float*)dlogits, ldd, C);
    else if (dtype == CTMI_BF16) hipLaunchKernelGGL((ce_soft_bwd_k<bf16_t>), dim3((unsigned)N), dim3(256), 0, st, (const bf16_t*)logits, ld, target, ldt, row_lse, row_tsum, loss_out, gout, (bf16_t*)dlogits, ldd, C);
    else if (dtype == CTMI_F16) hipLaunchKernelGGL((ce_soft_bwd_k<f16_t>), dim3((unsigned)N), dim3(256), 0, st, (const f16_t*)logits, ld, target, ldt, row_lse, row_tsum, loss_out, gout, (f16_t*)dlogits, ldd, C);
    else { ctmi_set_error("ce_soft_bwd: unsupported dtype %d", dtype); return CTMI_ERR_UNSUPPORTED; }
    CTMI_CHECK_LAUNCH("ce_soft_bwd");
    return CTMI_OK;
}

// ------------------------------------------------------------------------------------------------
// fused multi-tensor AdamW / SGD   (optimizer.py:53-97, 12-50; torch.optim.AdamW semantics at ft_bloom.py:70)
// 28 B/param of HBM traffic (+2 B when the bf16 shadow is written, +4 B when the L2 form writes the grad back).
// ------------------------------------------------------------------------------------------------
struct MTPack {
    float* p[CTMI_MT_MAX];
    float* g[CTMI_MT_MAX];
    float* m[CTMI_MT_MAX];
    float* v[CTMI_MT_MAX];
    bf16_t* shadow[CTMI_MT_MAX];
    int64_t n[CTMI_MT_MAX];
};
struct AdamHyper { float lr, b1, b2, eps, wd, bc1, bc2, sqrt_bc2, gscale; int decoupled, mutate_grad, shadow_f16; };

__device__ __forceinline__ void adam_one(float& p, float& g, float& m, float& v, const AdamHyper& h) {
    // no FMA contraction: every launch form, and the vector and scalar paths of one launch, round every product and sum the same way — the
    // update of an element does not depend on which kernel (or which lane of it) happened to process it
#pragma clang fp contract(off)
    g *= h.gscale;
    if (h.decoupled) { p *= (1.0f - h.lr * h.wd); } else { g += h.wd * p; }
    m = h.b1 * m + (1.0f - h.b1) * g;
    v = h.b2 * v + (1.0f - h.b2) * g * g;
    if (h.decoupled) p -= (h.lr / h.bc1) * (m / (sqrtf(v) / h.sqrt_bc2 + h.eps));
    else             p -= h.lr * (m / h.bc1) / (sqrtf(v / h.bc2) + h.eps);
}

__global__ __launch_bounds__(256) void adamw_mt_k(MTPack pk, AdamHyper h) {
    const int ti = blockIdx.y;
    const int64_t n = pk.n[ti];
    float* __restrict__ p = pk.p[ti]; float* __restrict__ g = pk.g[ti];
    float* __restrict__ m = pk.m[ti]; float* __restrict__ v = pk.v[ti];
    bf16_t* __restrict__ sh = pk.shadow[ti];
    const bool vec = ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0) && (sh == nullptr || (((uintptr_t)sh) & 7) == 0);
    const int64_t n4 = vec ? (n / 4) : 0;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        float4 P = ldg_stream(p + 4 * i), G = ldg_stream(g + 4 * i);
        float4 M = ldg_stream(m + 4 * i), V = ldg_stream(v + 4 * i);
        adam_one(P.x, G.x, M.x, V.x, h); adam_one(P.y, G.y, M.y, V.y, h);
        adam_one(P.z, G.z, M.z, V.z, h); adam_one(P.w, G.w, M.w, V.w, h);
        stg_stream(p + 4 * i, P); stg_stream(m + 4 * i, M); stg_stream(v + 4 * i, V);
        if (h.mutate_grad) stg_stream(g + 4 * i, G);
        // the operand copy of the weights in the compute dtype (bf16, or IEEE half since round 5), written in the same pass
        if (sh) reinterpret_cast<uint2*>(sh)[i] = h.shadow_f16 ? make_uint2(pack_h2(P.x, P.y), pack_h2(P.z, P.w)) : make_uint2(pack_bf2(P.x, P.y), pack_bf2(P.z, P.w));
    }
    for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        float P = p[i], G = g[i], M = m[i], V = v[i];
        adam_one(P, G, M, V, h);
        p[i] = P; m[i] = M; v[i] = V;
        if (h.mutate_grad) g[i] = G;
        if (sh) sh[i] = h.shadow_f16 ? f2h(P).v : f2bf(P);
    }
}

// Chunk-balanced form (round 6): up to CTMI_MT_FLAT tensors per launch, a 1-D grid of one 16 Ki-element chunk per workgroup (64 KiB of every
// fp32 stream; 4 x 16-byte loads per stream and thread in flight).  The (x = stride loop, y = tensor) grid above gives every tensor of a pack
// the workgroups of the LARGEST one — a pack of the 257 M-element tied table and 23 small vectors launches 24 x 2048 workgroups of which 23 x
// ~2040 exit at once — and the 294 parameters of Bloom-560M took 13 launches; here it is 5, every workgroup with the same amount of work.
#define CTMI_MT_FLAT 64
constexpr int ADAM_CHUNK = 16384;                          // elements per workgroup: 256 threads x 4 float4 x 4
struct MTFlat {
    float* p[CTMI_MT_FLAT]; float* g[CTMI_MT_FLAT]; float* m[CTMI_MT_FLAT]; float* v[CTMI_MT_FLAT];
    bf16_t* shadow[CTMI_MT_FLAT];
    int64_t n[CTMI_MT_FLAT];
    int first[CTMI_MT_FLAT + 1];                           // first chunk of tensor i in this launch's grid
    int count;
};
// DEV: the hyper-parameters (with this step's bias corrections) are read from device memory — the form a captured hipGraph replays: the launch's
// arguments are frozen at capture, the record behind `hd` is rewritten before every replay (ctmi_adamw_set_hyper)
template <bool DEV>
__global__ __launch_bounds__(256) void adamw_flat_k(MTFlat pk, AdamHyper hv, const AdamHyper* __restrict__ hd) {
    AdamHyper h;
    if constexpr (DEV) h = *hd; else h = hv;
    // tensor of this chunk: binary search over <= 64 prefix entries (scalar registers: blockIdx is uniform)
    int lo = 0, hi = pk.count;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if ((int)blockIdx.x >= pk.first[mid]) lo = mid; else hi = mid; }
    const int ti = lo;
    const int64_t n = pk.n[ti];
    float* __restrict__ p = pk.p[ti]; float* __restrict__ g = pk.g[ti];
    float* __restrict__ m = pk.m[ti]; float* __restrict__ v = pk.v[ti];
    bf16_t* __restrict__ sh = pk.shadow[ti];
    const int64_t e0 = (int64_t)((int)blockIdx.x - pk.first[ti]) * ADAM_CHUNK;
    const int64_t e1 = e0 + ADAM_CHUNK < n ? e0 + ADAM_CHUNK : n;
    const bool vec = ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0) && (sh == nullptr || (((uintptr_t)sh) & 7) == 0);
    if (vec && e1 - e0 == ADAM_CHUNK) {
        // full chunk: four passes of 4 float4 per stream and thread, the sixteen loads of a pass issued before its first use
#pragma unroll 1
        for (int pass = 0; pass < ADAM_CHUNK / 4096; ++pass) {
        const int64_t i0 = e0 / 4 + pass * 1024 + threadIdx.x;
        float4 P[4], G[4], M[4], V[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { P[u] = ldg_stream(p + 4 * (i0 + 256 * u)); G[u] = ldg_stream(g + 4 * (i0 + 256 * u)); }
#pragma unroll
        for (int u = 0; u < 4; ++u) { M[u] = ldg_stream(m + 4 * (i0 + 256 * u)); V[u] = ldg_stream(v + 4 * (i0 + 256 * u)); }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t i = i0 + 256 * u;
            adam_one(P[u].x, G[u].x, M[u].x, V[u].x, h); adam_one(P[u].y, G[u].y, M[u].y, V[u].y, h);
            adam_one(P[u].z, G[u].z, M[u].z, V[u].z, h); adam_one(P[u].w, G[u].w, M[u].w, V[u].w, h);
            stg_stream(p + 4 * i, P[u]); stg_stream(m + 4 * i, M[u]); stg_stream(v + 4 * i, V[u]);
            if (h.mutate_grad) stg_stream(g + 4 * i, G[u]);
            if (sh) reinterpret_cast<uint2*>(sh)[i] = h.shadow_f16 ? make_uint2(pack_h2(P[u].x, P[u].y), pack_h2(P[u].z, P[u].w))
                                                                   : make_uint2(pack_bf2(P[u].x, P[u].y), pack_bf2(P[u].z, P[u].w));
        }
        }
        return;
    }
    // last chunk of a tensor / unaligned tensor: the vector part of what is left, then the scalar tail
    const int64_t nv = vec ? (e1 - e0) / 4 : 0;
    for (int64_t k = threadIdx.x; k < nv; k += 256) {
        const int64_t i = e0 / 4 + k;
        float4 P = ldg_stream(p + 4 * i), G = ldg_stream(g + 4 * i), M = ldg_stream(m + 4 * i), V = ldg_stream(v + 4 * i);
        adam_one(P.x, G.x, M.x, V.x, h); adam_one(P.y, G.y, M.y, V.y, h);
        adam_one(P.z, G.z, M.z, V.z, h); adam_one(P.w, G.w, M.w, V.w, h);
        stg_stream(p + 4 * i, P); stg_stream(m + 4 * i, M); stg_stream(v + 4 * i, V);
        if (h.mutate_grad) stg_stream(g + 4 * i, G);
        if (sh) reinterpret_cast<uint2*>(sh)[i] = h.shadow_f16 ? make_uint2(pack_h2(P.x, P.y), pack_h2(P.z, P.w)) : make_uint2(pack_bf2(P.x, P.y), pack_bf2(P.z, P.w));
    }
    for (int64_t i = e0 + nv * 4 + threadIdx.x; i < e1; i += 256) {
        float P = p[i], G = g[i], M = m[i], V = v[i];
        adam_one(P, G, M, V, h);
        p[i] = P; m[i] = M; v[i] = V;
        if (h.mutate_grad) g[i] = G;
        if (sh) sh[i] = h.shadow_f16 ? f2h(P).v : f2bf(P);
    }
}
static int adamw_form() {                                   // CTMI_ADAMW_FLAT=0: the (stride loop, tensor) grid of rounds 1-5, for A/B runs
    static int f = -1;
    if (f < 0) { const char* e = getenv("CTMI_ADAMW_FLAT"); f = e ? atoi(e) : 1; }
    return f;
}

static int mt_grid_x(const int64_t* n, int count) {
    int64_t mx = 1;
    for (int i = 0; i < count; ++i) mx = std::max(mx, n[i]);
    return (int)std::min<int64_t>(cdiv64(mx, 256 * 4 * 4), 2048);
}

static AdamHyper adam_hyper(float lr, float beta1, float beta2, float eps, float weight_decay, int step, int decoupled, int mutate_grad, float grad_scale) {
    AdamHyper h;
    h.lr = lr; h.b1 = beta1; h.b2 = beta2; h.eps = eps; h.wd = weight_decay;
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    h.bc1 = (float)bc1; h.bc2 = (float)bc2; h.sqrt_bc2 = (float)sqrt(bc2);
    h.shadow_f16 = (mutate_grad & CTMI_OPT_SHADOW_F16) ? 1 : 0;
    mutate_grad &= 1;
    h.gscale = grad_scale; h.decoupled = decoupled; h.mutate_grad = (mutate_grad && !decoupled && weight_decay != 0.f) || (mutate_grad && grad_scale != 1.0f);
    return h;
}
__global__ void adam_set_hyper_k(AdamHyper h, AdamHyper* dst) { if (threadIdx.x == 0 && blockIdx.x == 0) *dst = h; }

// one launch per <= CTMI_MT_FLAT tensors; hd != nullptr: the hyper-parameters come from the device record
static int adamw_flat_launch(float* const* p, float* const* g, float* const* m, float* const* v, void* const* shadow, const int64_t* n, int count,
                             const AdamHyper& h, const AdamHyper* hd, hipStream_t st) {
        for (int base = 0; base < count; ) {
            MTFlat pk;
            int c = 0;
            int64_t chunks = 0;
            while (base + c < count && c < CTMI_MT_FLAT) {
                const int i = base + c;
                CTMI_REQUIRE(p[i] && g[i] && m[i] && v[i] && n[i] >= 0, "adamw_step: null tensor %d", i);
                const int64_t ch = cdiv64(n[i], ADAM_CHUNK);
                if (chunks + ch > (int64_t)0x7fffffff) break;                       // (a launch's grid; never reached below 3.5e13 elements)
                pk.p[c] = p[i]; pk.g[c] = g[i]; pk.m[c] = m[i]; pk.v[c] = v[i];
                pk.shadow[c] = shadow ? (bf16_t*)shadow[i] : nullptr; pk.n[c] = n[i];
                pk.first[c] = (int)chunks;
                chunks += ch;
                ++c;
            }
            CTMI_REQUIRE(c > 0, "adamw_step: tensor %d is too large for one launch", base);
            pk.first[c] = (int)chunks; pk.count = c;
            for (int i = c; i < CTMI_MT_FLAT; ++i) { pk.p[i] = pk.g[i] = pk.m[i] = pk.v[i] = nullptr; pk.shadow[i] = nullptr; pk.n[i] = 0; pk.first[i + 1] = (int)chunks; }
            if (chunks > 0) {
                if (hd != nullptr) hipLaunchKernelGGL(adamw_flat_k<true>, dim3((unsigned)chunks), dim3(256), 0, st, pk, h, hd);
                else hipLaunchKernelGGL(adamw_flat_k<false>, dim3((unsigned)chunks), dim3(256), 0, st, pk, h, hd);
                CTMI_CHECK_LAUNCH("adamw_step");
            }
            base += c;
        }
        return CTMI_OK;
}

extern "C" int ctmi_adamw_set_hyper(void* hyper_dev, float lr, float beta1, float beta2, float eps, float weight_decay, int step, int decoupled,
                                    int mutate_grad, float grad_scale, void* stream) {
    CTMI_REQUIRE(hyper_dev != nullptr && step >= 1 && (((uintptr_t)hyper_dev) & 3) == 0, "adamw_set_hyper: bad args (step must be >= 1)");
    const AdamHyper h = adam_hyper(lr, beta1, beta2, eps, weight_decay, step, decoupled, mutate_grad, grad_scale);
    hipLaunchKernelGGL(adam_set_hyper_k, dim3(1), dim3(64), 0, as_stream(stream), h, reinterpret_cast<AdamHyper*>(hyper_dev));
    CTMI_CHECK_LAUNCH("adamw_set_hyper");
    return CTMI_OK;
}
extern "C" int ctmi_adamw_step_dev(float* const* p, float* const* g, float* const* m, float* const* v, void* const* shadow,
                                   const int64_t* n, int count, const void* hyper_dev, void* stream) {
    ProfScope prof__(CTMI_PROF_OPTIMIZER, as_stream(stream));
    CTMI_REQUIRE(p && g && m && v && n && count >= 0 && hyper_dev != nullptr, "adamw_step_dev: bad args");
    const AdamHyper unused = {};
    return adamw_flat_launch(p, g, m, v, shadow, n, count, unused, reinterpret_cast<const AdamHyper*>(hyper_dev), as_stream(stream));
}

extern "C" int ctmi_adamw_step(float* const* p, float* const* g, float* const* m, float* const* v, void* const* shadow,
                               const int64_t* n, int count, float lr, float beta1, float beta2, float eps,
                               float weight_decay, int step, int decoupled, int mutate_grad, float grad_scale, void* stream) {
    ProfScope prof__(CTMI_PROF_OPTIMIZER, as_stream(stream));
    CTMI_REQUIRE(p && g && m && v && n && count >= 0 && step >= 1, "adamw_step: bad args (step must be >= 1)");
    const bool legacy_grid = (mutate_grad & CTMI_OPT_LEGACY_GRID) != 0;
    const AdamHyper h = adam_hyper(lr, beta1, beta2, eps, weight_decay, step, decoupled, mutate_grad, grad_scale);
    if (adamw_form() != 0 && !legacy_grid) return adamw_flat_launch(p, g, m, v, shadow, n, count, h, nullptr, as_stream(stream));
    for (int base = 0; base < count; base += CTMI_MT_MAX) {
        const int c = std::min(CTMI_MT_MAX, count - base);
        MTPack pk;
        for (int i = 0; i < c; ++i) {
            CTMI_REQUIRE(p[base + i] && g[base + i] && m[base + i] && v[base + i] && n[base + i] >= 0, "adamw_step: null tensor %d", base + i);
            pk.p[i] = p[base + i]; pk.g[i] = g[base + i]; pk.m[i] = m[base + i]; pk.v[i] = v[base + i];
            pk.shadow[i] = shadow ? (bf16_t*)shadow[base + i] : nullptr; pk.n[i] = n[base + i];
        }
        hipLaunchKernelGGL(adamw_mt_k, dim3(mt_grid_x(n + base, c), c), dim3(256), 0, as_stream(stream), pk, h);
        CTMI_CHECK_LAUNCH("adamw_step");
    }
    return CTMI_OK;
}

// ------------------------------------------------------------------------------------------------ loss scaling (GradScaler)
// torch.cuda.amp.GradScaler semantics (ft_bloom_DDP.py:107-128: scaler.scale(loss).backward(); scaler.step(opt); scaler.update()).
// state[0] = scale, state[1] = growth tracker, state[2] = found_inf (0/1) — one device-resident record, so scale() / unscale /
// update never read the scale on the host.  Unscale is in place (the caller can inspect true gradients after step(), as with
// torch) and multi-tensor: one launch per <= 24 gradients instead of one per parameter.
__global__ __launch_bounds__(256) void amp_unscale_k(MTPack pk, float* __restrict__ state) {
    const int ti = blockIdx.y;
    const int64_t n = pk.n[ti];
    float* __restrict__ g = pk.g[ti];
    const float inv = 1.0f / state[0];
    bool bad = false;
    const bool vec = (((uintptr_t)g) & 15) == 0;
    const int64_t n4 = vec ? n / 4 : 0;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        float4 G = reinterpret_cast<float4*>(g)[i];
        G.x *= inv; G.y *= inv; G.z *= inv; G.w *= inv;
        bad |= !(isfinite(G.x) && isfinite(G.y) && isfinite(G.z) && isfinite(G.w));
        reinterpret_cast<float4*>(g)[i] = G;
    }
    for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const float G = g[i] * inv;
        bad |= !isfinite(G);
        g[i] = G;
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) state[2] = 1.0f;          // benign race: every writer stores the same value
}
extern "C" int ctmi_amp_unscale(float* const* g, const int64_t* n, int count, float* state, void* stream) {
    ProfScope prof__(CTMI_PROF_OPTIMIZER, as_stream(stream));
    CTMI_REQUIRE(g && n && state && count >= 0, "amp_unscale: bad args");
    for (int base = 0; base < count; base += CTMI_MT_MAX) {
        const int c = std::min(CTMI_MT_MAX, count - base);
        MTPack pk;
        for (int i = 0; i < c; ++i) {
            CTMI_REQUIRE(g[base + i] && n[base + i] >= 0, "amp_unscale: null tensor %d", base + i);
            pk.p[i] = nullptr; pk.g[i] = g[base + i]; pk.m[i] = nullptr; pk.v[i] = nullptr; pk.shadow[i] = nullptr; pk.n[i] = n[base + i];
        }
        hipLaunchKernelGGL(amp_unscale_k, dim3(mt_grid_x(n + base, c), c), dim3(256), 0, as_stream(stream), pk, state);
        CTMI_CHECK_LAUNCH("amp_unscale");
    }
    return CTMI_OK;
}
// torch's _amp_update_scale_: found_inf -> scale *= backoff, tracker = 0; else ++tracker, and at growth_interval scale *= growth
// (only if the grown scale is finite), tracker = 0.  Clears found_inf for the next step.
__global__ void amp_update_k(float* __restrict__ state, float growth, float backoff, int interval) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (state[2] != 0.0f) { state[0] *= backoff; state[1] = 0.0f; }
    else {
        const float t = state[1] + 1.0f;
        if ((int)t == interval) { const float ns = state[0] * growth; if (isfinite(ns)) state[0] = ns; state[1] = 0.0f; }
        else state[1] = t;
    }
    state[2] = 0.0f;
}
extern "C" int ctmi_amp_update(float* state, float growth, float backoff, int interval, void* stream) {
    ProfScope prof__(CTMI_PROF_OPTIMIZER, as_stream(stream));
    CTMI_REQUIRE(state && interval >= 1, "amp_update: bad args");
    hipLaunchKernelGGL(amp_update_k, dim3(1), dim3(64), 0, as_stream(stream), state, growth, backoff, interval);
    CTMI_CHECK_LAUNCH("amp_update");
    return CTMI_OK;
}

struct SgdHyper { float lr, momentum, dampening, wd; int first, shadow_f16; };
__global__ __launch_bounds__(256) void sgd_mt_k(MTPack pk, SgdHyper h) {
    const int ti = blockIdx.y;
    const int64_t n = pk.n[ti];
    float* __restrict__ p = pk.p[ti]; float* __restrict__ g = pk.g[ti]; float* __restrict__ buf = pk.m[ti];
    bf16_t* __restrict__ sh = pk.shadow[ti];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float P = p[i], G = g[i];
        if (h.wd != 0.f) G += h.wd * P;                         // optimizer.py:38-39
        if (buf != nullptr) {                                    // optimizer.py:41-48
            float B = h.first ? G : h.momentum * buf[i] + (1.0f - h.dampening) * G;
            buf[i] = B; G = B;
        }
        g[i] = G;                                                // the reference leaves param.grad = buf / decayed grad
        P -= h.lr * G;
        p[i] = P;
        if (sh) sh[i] = h.shadow_f16 ? f2h(P).v : f2bf(P);
    }
}
extern "C" int ctmi_sgd_step(float* const* p, float* const* g, float* const* buf, void* const* shadow, const int64_t* n,
                             int count, float lr, float momentum, float dampening, float weight_decay, int first_step, void* stream) {
    ProfScope prof__(CTMI_PROF_OPTIMIZER, as_stream(stream));
    CTMI_REQUIRE(p && g && n && count >= 0, "sgd_step: bad args");
    SgdHyper h{lr, momentum, dampening, weight_decay, first_step & 1, (first_step & CTMI_OPT_SHADOW_F16) ? 1 : 0};
    for (int base = 0; base < count; base += CTMI_MT_MAX) {
        const int c = std::min(CTMI_MT_MAX, count - base);
        MTPack pk;
        for (int i = 0; i < c; ++i) {
            pk.p[i] = p[base + i]; pk.g[i] = g[base + i]; pk.m[i] = buf ? buf[base + i] : nullptr; pk.v[i] = nullptr;
            pk.shadow[i] = shadow ? (bf16_t*)shadow[base + i] : nullptr; pk.n[i] = n[base + i];
        }
        hipLaunchKernelGGL(sgd_mt_k, dim3(mt_grid_x(n + base, c), c), dim3(256), 0, as_stream(stream), pk, h);
        CTMI_CHECK_LAUNCH("sgd_step");
    }
    return CTMI_OK;
}

// ------------------------------------------------------------------------------------------------
// utilities
// ------------------------------------------------------------------------------------------------
template <typename S, typename D>
__global__ __launch_bounds__(256) void cast_k(const S* __restrict__ src, D* __restrict__ dst, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        dst[i] = Cvt<D>::from_f(Cvt<S>::to_f(src[i]));
}
__global__ __launch_bounds__(256) void cast_f32_bf16_v4(const float4* __restrict__ src, uint2* __restrict__ dst, int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 a = src[i];
        dst[i] = make_uint2(pack_bf2(a.x, a.y), pack_bf2(a.z, a.w));
    }
}
extern "C" int ctmi_cast(const void* src, int sd, void* dst, int dd, int64_t n, void* stream) {
    ProfScope prof__(CTMI_PROF_OTHER, as_stream(stream));
    CTMI_REQUIRE(src && dst && n >= 0, "cast: bad args");
    if (n == 0) return CTMI_OK;
    hipStream_t st = as_stream(stream);
    int grid = (int)std::min<int64_t>(cdiv64(n, 256 * 4), 4096);
    if (sd == CTMI_F32 && dd == CTMI_BF16) {
        if (aligned16(src) && ((((uintptr_t)dst) & 7) == 0) && n % 4 == 0)
            hipLaunchKernelGGL(cast_f32_bf16_v4, dim3(grid), dim3(256), 0, st, (const float4*)src, (uint2*)dst, n / 4);
        else hipLaunchKernelGGL((cast_k<float, bf16_t>), dim3(grid), dim3(256), 0, st, (const float*)src, (bf16_t*)dst, n);
    } else if (sd == CTMI_BF16 && dd == CTMI_F32) hipLaunchKernelGGL((cast_k<bf16_t, float>), dim3(grid), dim3(256), 0, st, (const bf16_t*)src, (float*)dst, n);
    else if (sd == CTMI_F32 && dd == CTMI_F32) hipLaunchKernelGGL((cast_k<float, float>), dim3(grid), dim3(256), 0, st, (const float*)src, (float*)dst, n);
    else if (sd == CTMI_BF16 && dd == CTMI_BF16) hipLaunchKernelGGL((cast_k<bf16_t, bf16_t>), dim3(grid), dim3(256), 0, st, (const bf16_t*)src, (bf16_t*)dst, n);
    else if (sd == CTMI_F32 && dd == CTMI_F16) hipLaunchKernelGGL((cast_k<float, f16_t>), dim3(grid), dim3(256), 0, st, (const float*)src, (f16_t*)dst, n);
    else if (sd == CTMI_F16 && dd == CTMI_F32) hipLaunchKernelGGL((cast_k<f16_t, float>), dim3(grid), dim3(256), 0, st, (const f16_t*)src, (float*)dst, n);
    else if (sd == CTMI_F16 && dd == CTMI_F16) hipLaunchKernelGGL((cast_k<f16_t, f16_t>), dim3(grid), dim3(256), 0, st, (const f16_t*)src, (f16_t*)dst, n);
    else { ctmi_set_error("cast: unsupported dtypes %d->%d", sd, dd); return CTMI_ERR_UNSUPPORTED; }
    CTMI_CHECK_LAUNCH("cast");
    return CTMI_OK;
}

// dst[c][r] = (TD) src[r][c]   (src fp32 [R,C] row-major -> dst [C,R]): 64x64 tiles through padded LDS, coalesced both ways.
// GPT-2's Conv1D keeps its weight as [in,out] (modeling_gpt.py:32-46); the GEMM kernels read the [out,in] compute copy.
template <typename TD>
__global__ __launch_bounds__(256) void transpose_cast_k(const float* __restrict__ src, TD* __restrict__ dst, int64_t R, int64_t C) {
    __shared__ float tile[64][65];
    const int64_t r0 = (int64_t)blockIdx.y * 64, c0 = (int64_t)blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int64_t r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < R && c < C) ? src[r * C + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int64_t c = c0 + i, r = r0 + tx;
        if (c < C && r < R) dst[c * R + r] = Cvt<TD>::from_f(tile[tx][i]);
    }
}
extern "C" int ctmi_transpose_cast(const float* src, void* dst, int dst_dtype, int64_t rows, int64_t cols, void* stream) {
    ProfScope prof__(CTMI_PROF_OTHER, as_stream(stream));
    CTMI_REQUIRE(src && dst && rows > 0 && cols > 0, "transpose_cast: bad args");
    const dim3 grid((unsigned)cdiv64(cols, 64), (unsigned)cdiv64(rows, 64));
    if (dst_dtype == CTMI_F32) hipLaunchKernelGGL((transpose_cast_k<float>), grid, dim3(256), 0, as_stream(stream), src, (float*)dst, rows, cols);
    else if (dst_dtype == CTMI_BF16) hipLaunchKernelGGL((transpose_cast_k<bf16_t>), grid, dim3(256), 0, as_stream(stream), src, (bf16_t*)dst, rows, cols);
    else if (dst_dtype == CTMI_F16) hipLaunchKernelGGL((transpose_cast_k<f16_t>), grid, dim3(256), 0, as_stream(stream), src, (f16_t*)dst, rows, cols);
    else { ctmi_set_error("transpose_cast: unsupported dtype %d", dst_dtype); return CTMI_ERR_UNSUPPORTED; }
    CTMI_CHECK_LAUNCH("transpose_cast");
    return CTMI_OK;
}

__global__ __launch_bounds__(256) void sumsq_k(const float* __restrict__ x, int64_t n, double* __restrict__ out) {
    double s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) { const double v = x[i]; s += v * v; }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    __shared__ double sm[4];
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, sm[0] + sm[1] + sm[2] + sm[3]);
}
extern "C" int ctmi_sumsq(const float* x, int64_t n, double* out, int accumulate, void* stream) {
    ProfScope prof__(CTMI_PROF_OTHER, as_stream(stream));
    CTMI_REQUIRE(x && out && n >= 0, "sumsq: bad args");
    hipStream_t st = as_stream(stream);
    if (!accumulate) { if (hipMemsetAsync(out, 0, sizeof(double), st) != hipSuccess) { ctmi_set_error("sumsq: memset failed"); return CTMI_ERR_LAUNCH; } }
    if (n == 0) return CTMI_OK;
    int grid = (int)std::min<int64_t>(cdiv64(n, 256 * 8), 1024);
    hipLaunchKernelGGL(sumsq_k, dim3(grid), dim3(256), 0, st, x, n, out);
    CTMI_CHECK_LAUNCH("sumsq");
    return CTMI_OK;
}

// dst[i] = f * src[i]  (dst may alias src).  float4 body when both pointers are 16-byte aligned, scalar tail.
__global__ __launch_bounds__(256) void scale_copy_k(const float* src, float* dst, int64_t n, float s, const float* __restrict__ sd, int vec) {
    const float f = sd ? s * sd[0] : s;
    const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x, nth = (int64_t)gridDim.x * 256;
    const int64_t n4 = vec ? n / 4 : 0;
    for (int64_t i = tid; i < n4; i += nth) {
        float4 v = reinterpret_cast<const float4*>(src)[i];
        v.x *= f; v.y *= f; v.z *= f; v.w *= f;
        reinterpret_cast<float4*>(dst)[i] = v;
    }
    for (int64_t i = n4 * 4 + tid; i < n; i += nth) dst[i] = f * src[i];
}
static int scale_copy_launch(const float* src, float* dst, int64_t n, float s, const float* s_dev, void* stream, const char* what) {
    if (n == 0) return CTMI_OK;
    const int vec = ((((uintptr_t)src) | ((uintptr_t)dst)) & 15) == 0;
    int grid = (int)std::min<int64_t>(cdiv64(n, 256 * 16), 4096);
    hipLaunchKernelGGL(scale_copy_k, dim3(grid), dim3(256), 0, as_stream(stream), src, dst, n, s, s_dev, vec);
    CTMI_CHECK_LAUNCH(what);
    return CTMI_OK;
}
extern "C" int ctmi_scale(float* x, int64_t n, float s, const float* s_dev, void* stream) {
    ProfScope prof__(CTMI_PROF_OTHER, as_stream(stream));
    CTMI_REQUIRE(x && n >= 0, "scale: bad args");
    return scale_copy_launch(x, x, n, s, s_dev, stream, "scale");
}
extern "C" int ctmi_scale_copy(const float* src, float* dst, int64_t n, float s, void* stream) {
    ProfScope prof__(CTMI_PROF_OTHER, as_stream(stream));
    CTMI_REQUIRE(src && dst && n >= 0, "scale_copy: bad args");
    return scale_copy_launch(src, dst, n, s, nullptr, stream, "scale_copy");
}

template <typename T>
__global__ __launch_bounds__(256) void argmax_k(const T* __restrict__ x, int64_t ld, int64_t* __restrict__ out, int64_t cols) {
    const T* r = x + (int64_t)blockIdx.x * ld;
    float best = -INFINITY; int64_t bi = INT64_MAX;
    for (int64_t c = threadIdx.x; c < cols; c += 256) {
        const float v = Cvt<T>::to_f(r[c]);
        if (v > best || bi == INT64_MAX) { best = v; bi = c; }          // strictly greater keeps the first index
    }
    for (int o = 32; o > 0; o >>= 1) {
        const float b2 = __shfl_xor(best, o, 64); const int64_t i2 = __shfl_xor(bi, o, 64);
        if (b2 > best || (b2 == best && i2 < bi)) { best = b2; bi = i2; }
    }
    __shared__ float sb[4]; __shared__ int64_t si[4];
    if ((threadIdx.x & 63) == 0) { sb[threadIdx.x >> 6] = best; si[threadIdx.x >> 6] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 4; ++k) if (sb[k] > best || (sb[k] == best && si[k] < bi)) { best = sb[k]; bi = si[k]; }
        out[blockIdx.x] = bi;
    }
}
extern "C" int ctmi_argmax(const void* x, int64_t ld, int64_t* out, int64_t rows, int64_t cols, int dtype, void* stream) {
    CTMI_REQUIRE(x && out && rows > 0 && cols > 0 && ld >= cols, "argmax: bad args");
    if (dtype == CTMI_F32) hipLaunchKernelGGL((argmax_k<float>), dim3((unsigned)rows), dim3(256), 0, as_stream(stream), (const float*)x, ld, out, cols);
    else if (dtype == CTMI_BF16) hipLaunchKernelGGL((argmax_k<bf16_t>), dim3((unsigned)rows), dim3(256), 0, as_stream(stream), (const bf16_t*)x, ld, out, cols);
    else if (dtype == CTMI_F16) hipLaunchKernelGGL((argmax_k<f16_t>), dim3((unsigned)rows), dim3(256), 0, as_stream(stream), (const f16_t*)x, ld, out, cols);
    else { ctmi_set_error("argmax: unsupported dtype %d", dtype); return CTMI_ERR_UNSUPPORTED; }
    CTMI_CHECK_LAUNCH("argmax");
    return CTMI_OK;
}

// ------------------------------------------------------------------------------------------------ decode (beam search, samplers)
// Row statistics of log_softmax (generation_util.py:200): stats[row] = {max, log(sum exp(x - max))}, so that the consumer
// forms (x - max) - logsum exactly as torch.log_softmax does.  One workgroup per row; decode is latency-, not HBM-bound.
template <typename T>
__global__ __launch_bounds__(256) void row_lse_k(const T* __restrict__ x, int64_t ld, float* __restrict__ stats, int64_t cols) {
    const T* r = x + (int64_t)blockIdx.x * ld;
    __shared__ float red[4];
    float m = -INFINITY;
    for (int64_t c = threadIdx.x; c < cols; c += 256) m = fmaxf(m, Cvt<T>::to_f(r[c]));
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sum = 0.f;
    for (int64_t c = threadIdx.x; c < cols; c += 256) sum += expf(Cvt<T>::to_f(r[c]) - m);
    sum = wave_sum(sum);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        stats[2 * blockIdx.x] = m;
        stats[2 * blockIdx.x + 1] = logf((red[0] + red[1]) + (red[2] + red[3]));
    }
}
extern "C" int ctmi_row_lse(const void* x, int64_t ld, float* stats, int64_t rows, int64_t cols, int dtype, void* stream) {
    CTMI_REQUIRE(x && stats && rows > 0 && cols > 0 && ld >= cols, "row_lse: bad args");
    if (dtype == CTMI_F32) hipLaunchKernelGGL((row_lse_k<float>), dim3((unsigned)rows), dim3(256), 0, as_stream(stream), (const float*)x, ld, stats, cols);
    else if (dtype == CTMI_BF16) hipLaunchKernelGGL((row_lse_k<bf16_t>), dim3((unsigned)rows), dim3(256), 0, as_stream(stream), (const bf16_t*)x, ld, stats, cols);
    else if (dtype == CTMI_F16) hipLaunchKernelGGL((row_lse_k<f16_t>), dim3((unsigned)rows), dim3(256), 0, as_stream(stream), (const f16_t*)x, ld, stats, cols);
    else { ctmi_set_error("row_lse: unsupported dtype %d", dtype); return CTMI_ERR_UNSUPPORTED; }
    CTMI_CHECK_LAUNCH("row_lse");
    return CTMI_OK;
}

// Top-k over the `group` x cols candidates of one batch element (generation_util.py:199-224: scores.view(bsz, -1).topk(2*beam)):
//     score(i, v) = ((x[g*group + i][v] - max_i) - logsum_i) + add[g*group + i] * add_mul       (stats / add optional)
// Output k (value, flat index i*cols + v) pairs in descending value order, equal values by ascending flat index.
// Selection by k rounds of a lexicographic arg-max over the candidates that come after the previous pick: no per-thread
// candidate lists (dynamic register arrays would live in scratch), exact for -inf entries and duplicates; the candidate set
// (group * cols * 4 B <= a few MiB) stays in L2 between rounds.
template <typename T>
__global__ __launch_bounds__(1024) void group_topk_k(const T* __restrict__ x, int64_t ld, const float* __restrict__ stats,
                                                     const float* __restrict__ add, float add_mul, float* __restrict__ out_val,
                                                     int64_t* __restrict__ out_idx, int group, int64_t cols, int k) {
    const int64_t g = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    __shared__ float sv[16]; __shared__ int64_t si[16];
    __shared__ float pick_v; __shared__ int64_t pick_i;
    float pv = INFINITY; int64_t pi = -1;                         // previous pick: everything is "after" it
    for (int r = 0; r < k; ++r) {
        float bv = -INFINITY; int64_t bi = INT64_MAX;
        for (int i = 0; i < group; ++i) {
            const int64_t row = g * group + i;
            const T* xr = x + row * ld;
            const float mx = stats ? stats[2 * row] : 0.f, ls = stats ? stats[2 * row + 1] : 0.f;
            const float a = add ? add[row] * add_mul : 0.f;
            for (int64_t c = tid; c < cols; c += 1024) {
                float v = Cvt<T>::to_f(xr[c]);
                if (stats) v = (v - mx) - ls;
                if (add) v = v + a;
                const int64_t f = (int64_t)i * cols + c;
                const bool after = (v < pv) || (v == pv && f > pi);
                if (after && (v > bv || (v == bv && f < bi))) { bv = v; bi = f; }
            }
        }
        for (int o = 32; o > 0; o >>= 1) {
            const float v2 = __shfl_xor(bv, o, 64); const int64_t i2 = __shfl_xor(bi, o, 64);
            if (v2 > bv || (v2 == bv && i2 < bi)) { bv = v2; bi = i2; }
        }
        if (lane == 0) { sv[wave] = bv; si[wave] = bi; }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < 16; ++w) if (sv[w] > bv || (sv[w] == bv && si[w] < bi)) { bv = sv[w]; bi = si[w]; }
            pick_v = bv; pick_i = bi;
            out_val[g * k + r] = bv; out_idx[g * k + r] = bi;
        }
        __syncthreads();
        pv = pick_v; pi = pick_i;
    }
}
extern "C" int ctmi_group_topk(const void* x, int64_t ld, const float* stats, const float* add, float add_mul, float* out_val,
                               int64_t* out_idx, int64_t groups, int group, int64_t cols, int k, int dtype, void* stream) {
    CTMI_REQUIRE(x && out_val && out_idx && groups > 0 && group > 0 && cols > 0 && ld >= cols, "group_topk: bad args");
    CTMI_REQUIRE(k > 0 && (int64_t)k <= (int64_t)group * cols, "group_topk: k must be in [1, group*cols]");
    if (dtype == CTMI_F32) hipLaunchKernelGGL((group_topk_k<float>), dim3((unsigned)groups), dim3(1024), 0, as_stream(stream),
                                              (const float*)x, ld, stats, add, add_mul, out_val, out_idx, group, cols, k);
    else if (dtype == CTMI_BF16) hipLaunchKernelGGL((group_topk_k<bf16_t>), dim3((unsigned)groups), dim3(1024), 0, as_stream(stream),
                                                    (const bf16_t*)x, ld, stats, add, add_mul, out_val, out_idx, group, cols, k);
    else if (dtype == CTMI_F16) hipLaunchKernelGGL((group_topk_k<f16_t>), dim3((unsigned)groups), dim3(1024), 0, as_stream(stream),
                                                    (const f16_t*)x, ld, stats, add, add_mul, out_val, out_idx, group, cols, k);
    else { ctmi_set_error("group_topk: unsupported dtype %d", dtype); return CTMI_ERR_UNSUPPORTED; }
    CTMI_CHECK_LAUNCH("group_topk");
    return CTMI_OK;
}

// Sampler filters on fp32 scores (logits_processor.py:35-56): out = x / divisor, then entries strictly below the row's
// threshold (thr[row * thr_stride], optional) are replaced by `fill`.  A true division, as the reference performs.
__global__ __launch_bounds__(256) void scores_filter_k(const float* __restrict__ x, int64_t ld, float divisor, const float* __restrict__ thr,
                                                       int64_t thr_stride, float fill, float* __restrict__ out, int64_t ldo, int64_t cols) {
    const int64_t row = blockIdx.y;
    const float t = thr ? thr[row * thr_stride] : -INFINITY;
    for (int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x; c < cols; c += (int64_t)gridDim.x * 256) {
        float v = x[row * ld + c];
        if (divisor != 1.0f) v = __fdiv_rn(v, divisor);
        out[row * ldo + c] = (thr && v < t) ? fill : v;
    }
}
extern "C" int ctmi_scores_filter(const float* x, int64_t ld, float divisor, const float* thr, int64_t thr_stride, float fill,
                                  float* out, int64_t ldo, int64_t rows, int64_t cols, void* stream) {
    CTMI_REQUIRE(x && out && rows > 0 && rows < 65536 && cols > 0 && ld >= cols && ldo >= cols, "scores_filter: bad args");
    const unsigned gx = (unsigned)std::min<int64_t>(cdiv64(cols, 256), 1024);
    hipLaunchKernelGGL(scores_filter_k, dim3(gx, (unsigned)rows), dim3(256), 0, as_stream(stream), x, ld, divisor, thr, thr_stride, fill,
                       out, ldo, cols);
    CTMI_CHECK_LAUNCH("scores_filter");
    return CTMI_OK;
}

// attention_mask [B,S] -> ALiBi key positions (cumsum(mask)-1)*mask as fp32, validity as int32 (modeling_bloom.py:328,178)
__global__ __launch_bounds__(64) void mask_prep_k(const int64_t* __restrict__ am, float* __restrict__ kpos,
                                                  int32_t* __restrict__ kvalid, int32_t* __restrict__ first_valid, int64_t S) {
    const int64_t b = blockIdx.x;
    const int lane = threadIdx.x;
    int64_t running = 0;
    int64_t first = S;
    for (int64_t base = 0; base < S; base += 64) {
        const int64_t j = base + lane;
        const int64_t mv = (j < S) ? am[b * S + j] : 0;
        int64_t incl = mv;                                       // inclusive wave scan
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int64_t t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
        if (j < S) {
            kpos[b * S + j] = (float)((running + incl - 1) * mv);
            kvalid[b * S + j] = (mv != 0) ? 1 : 0;
            if (mv != 0 && j < first) first = j;
        }
        running += __shfl(incl, 63, 64);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const int64_t t = __shfl_xor(first, o, 64); first = t < first ? t : first; }
    if (lane == 0) first_valid[b] = (int32_t)first;
}
extern "C" int ctmi_mask_prep(const int64_t* am, float* kpos, int32_t* kvalid, int32_t* first_valid, int64_t B, int64_t S, void* stream) {
    ProfScope prof__(CTMI_PROF_OTHER, as_stream(stream));
    CTMI_REQUIRE(am && kpos && kvalid && first_valid && B > 0 && S > 0, "mask_prep: bad args");
    hipLaunchKernelGGL(mask_prep_k, dim3((unsigned)B), dim3(64), 0, as_stream(stream), am, kpos, kvalid, first_valid, S);
    CTMI_CHECK_LAUNCH("mask_prep");
    return CTMI_OK;
}
