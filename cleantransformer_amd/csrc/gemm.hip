// MFMA GEMM family for gfx950 (CDNA4).
//   C[M,N] = epilogue(alpha * sum_k A(m,k) B(k,n)),  fp32 accumulate in the matrix cores.
// bf16 operands -> v_mfma_f32_16x16x32_bf16; fp32 operands -> v_mfma_f32_16x16x4_f32 (exact fp32, parity mode).
// One 256-thread workgroup (4 wavefronts, 2x2) owns a 128x128 output tile; each wave a 64x64 sub-tile held as
// 4x4 16x16 accumulators.  Operand tiles are staged HBM -> registers (16-byte loads) -> LDS with the global
// loads of tile t+1 in flight under the MFMAs of tile t, LDS double-buffered, one barrier per K-step.
// An operand may be stored K-major ([K][rows]); its fragments are then gathered transposed from LDS, so the
// same kernel serves forward (x W^T), dgrad (dy W) and wgrad (dy^T x) without any transposed copies in HBM.
// The MFMA is issued with the operands swapped (D' = B A^T) so that every lane ends up with 4 *consecutive
// output columns* of one row: bias / residual / aux loads and the C store are 8-16 byte vector accesses.
#include "common.h"
#include "mma.h"

template <typename T> struct Tile;
template <> struct Tile<bf16_t> { static constexpr int BM = 128, BN = 128, BK = 64, PADK = 8, PADR = 8; };
template <> struct Tile<float>  { static constexpr int BM = 128, BN = 128, BK = 16, PADK = 4, PADR = 4; };

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

    // HBM -> registers.  row0: first tile row (in the M or N dimension), k0: first k of the tile.
    static __device__ __forceinline__ void load(uint4 (&regs)[NCH], const T* __restrict__ g, int64_t ld, int64_t row0,
                                                int64_t k0, int64_t row_lim, int64_t k_lim, bool vec_ok, int tid) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int id = tid + 256 * i;
            const int line = id / CPL, c = (id % CPL) * VEC;
            const int64_t gline = (KMAJOR ? k0 : row0) + line;           // index along the strided dimension
            const int64_t gcol = (KMAJOR ? row0 : k0) + c;               // index along the contiguous dimension
            const int64_t line_lim = KMAJOR ? k_lim : row_lim;
            const int64_t col_lim = KMAJOR ? row_lim : k_lim;
            const T* p = g + gline * ld + gcol;
            if (gline < line_lim && gcol + VEC <= col_lim && vec_ok) {
                regs[i] = *reinterpret_cast<const uint4*>(p);
            } else {
                T tmp[VEC];
#pragma unroll
                for (int j = 0; j < VEC; ++j) tmp[j] = (gline < line_lim && gcol + j < col_lim) ? p[j] : (T)0;
                regs[i] = *reinterpret_cast<const uint4*>(tmp);
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
    // MFMA operand fragment for tile row r (0..ROWS) at k offset kofs (= kk*K + (lane>>4)*KL): KL consecutive k.
    static __device__ __forceinline__ typename Mma<T>::Frag frag(const T* __restrict__ tile, int r, int kofs);
};

template <> __device__ __forceinline__ short8 OpTile<bf16_t, false, 128>::frag(const bf16_t* __restrict__ tile, int r, int kofs) {
    return *reinterpret_cast<const short8*>(tile + r * PITCH + kofs);
}
template <> __device__ __forceinline__ short8 OpTile<bf16_t, true, 128>::frag(const bf16_t* __restrict__ tile, int r, int kofs) {
    short8 f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = (short)tile[(kofs + j) * PITCH + r];
    return f;
}
template <> __device__ __forceinline__ float OpTile<float, false, 128>::frag(const float* __restrict__ tile, int r, int kofs) {
    return tile[r * PITCH + kofs];
}
template <> __device__ __forceinline__ float OpTile<float, true, 128>::frag(const float* __restrict__ tile, int r, int kofs) {
    return tile[kofs * PITCH + r];
}

struct GemmArgs {
    const void* A; const void* B; void* C;
    int64_t lda, ldb, ldc, M, N, K;
    float alpha; int beta;
    const float* bias; const void* residual; const void* aux_in; void* aux_out;
    int tiles_m, tiles_n, vec_a, vec_b, vec_c;
};

template <typename TO> __device__ __forceinline__ void store4(TO* p, const float* v);
template <> __device__ __forceinline__ void store4<float>(float* p, const float* v) { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }
template <> __device__ __forceinline__ void store4<bf16_t>(bf16_t* p, const float* v) { *reinterpret_cast<uint2*>(p) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])); }
template <typename TO> __device__ __forceinline__ void load4(const TO* p, float* v);
template <> __device__ __forceinline__ void load4<float>(const float* p, float* v) { float4 t = *reinterpret_cast<const float4*>(p); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
template <> __device__ __forceinline__ void load4<bf16_t>(const bf16_t* p, float* v) {
    uint2 t = *reinterpret_cast<const uint2*>(p);
    v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u); v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
}

template <typename T, typename TO, bool AK, bool BKM, int EPI>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs g) {
    using TA = OpTile<T, AK, Tile<T>::BM>;
    using TB = OpTile<T, BKM, Tile<T>::BN>;
    constexpr int BM = Tile<T>::BM, BN = Tile<T>::BN, BK = Tile<T>::BK;
    constexpr int MK = Mma<T>::K, KL = Mma<T>::KL;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* smem = reinterpret_cast<T*>(smem_raw);
    T* As[2] = { smem, smem + TA::ELEMS + TB::ELEMS };
    T* Bs[2] = { smem + TA::ELEMS, smem + 2 * TA::ELEMS + TB::ELEMS };

    // XCD-aware block -> tile map: the dispatcher places block b on XCD b%8; give each XCD a contiguous run of
    // tiles (bijective for any grid size) walking down M for a fixed weight tile, so B tiles are L2-resident.
    const int nblk = g.tiles_m * g.tiles_n;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, q = nblk >> 3, r8 = nblk & 7;
    const int vid = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (bid >> 3);
    const int tm = vid % g.tiles_m, tn = vid / g.tiles_m;
    const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1;
    const T* A = reinterpret_cast<const T*>(g.A);
    const T* B = reinterpret_cast<const T*>(g.B);

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nt = (int)((g.K + BK - 1) / BK);
    uint4 ra[TA::NCH], rb[TB::NCH];
    TA::load(ra, A, g.lda, m0, 0, g.M, g.K, g.vec_a, tid);
    TB::load(rb, B, g.ldb, n0, 0, g.N, g.K, g.vec_b, tid);
    TA::store(ra, As[0], tid);
    TB::store(rb, Bs[0], tid);
    __syncthreads();

    for (int t = 0; t < nt; ++t) {
        const int cur = t & 1;
        if (t + 1 < nt) {
            TA::load(ra, A, g.lda, m0, (int64_t)(t + 1) * BK, g.M, g.K, g.vec_a, tid);
            TB::load(rb, B, g.ldb, n0, (int64_t)(t + 1) * BK, g.N, g.K, g.vec_b, tid);
        }
        const T* as = As[cur];
        const T* bs = Bs[cur];
#pragma unroll
        for (int kk = 0; kk < BK / MK; ++kk) {
            const int kofs = kk * MK + (lane >> 4) * KL;
            typename Mma<T>::Frag af[4], bf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = TA::frag(as, wr * 64 + i * 16 + (lane & 15), kofs);
#pragma unroll
            for (int j = 0; j < 4; ++j) bf[j] = TB::frag(bs, wc * 64 + j * 16 + (lane & 15), kofs);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = Mma<T>::mma(bf[j], af[i], acc[i][j]);   // D'[n][m]
        }
        if (t + 1 < nt) {
            TA::store(ra, As[cur ^ 1], tid);
            TB::store(rb, Bs[cur ^ 1], tid);
        }
        __syncthreads();
    }

    // epilogue: lane holds C[m][n..n+3], m = m0+wr*64+i*16+(lane&15), n = n0+wc*64+j*16+(lane>>4)*4
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
#pragma unroll
                for (int r = 0; r < 4; ++r) if (n + r < g.N) v[r] += g.bias[n + r];
            }
            if (EPI == CTMI_EPI_GELU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = Cvt<T>::to_f(Cvt<T>::from_f(v[r]));       // GELU sees the stored value
                if (full) store4<T>(AUXO + off, v);
                else { for (int r = 0; r < 4; ++r) if (n + r < g.N) AUXO[off + r] = Cvt<T>::from_f(v[r]); }
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = gelu_tanh_f(v[r]);
            } else if (EPI == CTMI_EPI_RELU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
            } else if (EPI == CTMI_EPI_DGELU || EPI == CTMI_EPI_DRELU) {
                float u[4] = {0.f, 0.f, 0.f, 0.f};
                if (full) load4<T>(AUXI + off, u);
                else { for (int r = 0; r < 4; ++r) if (n + r < g.N) u[r] = Cvt<T>::to_f(AUXI[off + r]); }
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = (EPI == CTMI_EPI_DGELU) ? v[r] * gelu_tanh_grad_f(u[r]) : (u[r] > 0.f ? v[r] : 0.f);
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

template <typename T, typename TO, bool AK, bool BKM, int EPI>
static int gemm_launch(GemmArgs& g, hipStream_t st) {
    using TA = OpTile<T, AK, Tile<T>::BM>;
    using TB = OpTile<T, BKM, Tile<T>::BN>;
    const size_t lds = 2 * (size_t)(TA::ELEMS + TB::ELEMS) * sizeof(T);
    auto kern = &gemm_kernel<T, TO, AK, BKM, EPI>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3((unsigned)(g.tiles_m * g.tiles_n)), dim3(256), lds, st, g);
    CTMI_CHECK_LAUNCH("gemm");
    return CTMI_OK;
}

template <typename T>
static int gemm_dispatch(GemmArgs& g, int ak, int bk, int epi, int out_f32, hipStream_t st) {
    const bool of = out_f32 || sizeof(T) == 4;
    if (!ak && !bk && !of) {
        if (epi == CTMI_EPI_NONE) return gemm_launch<T, T, false, false, CTMI_EPI_NONE>(g, st);
        if (epi == CTMI_EPI_GELU) return gemm_launch<T, T, false, false, CTMI_EPI_GELU>(g, st);
        if (epi == CTMI_EPI_RELU) return gemm_launch<T, T, false, false, CTMI_EPI_RELU>(g, st);
    }
    if (!ak && bk && !of) {
        if (epi == CTMI_EPI_NONE) return gemm_launch<T, T, false, true, CTMI_EPI_NONE>(g, st);
        if (epi == CTMI_EPI_DGELU) return gemm_launch<T, T, false, true, CTMI_EPI_DGELU>(g, st);
        if (epi == CTMI_EPI_DRELU) return gemm_launch<T, T, false, true, CTMI_EPI_DRELU>(g, st);
    }
    if (ak && bk && of && epi == CTMI_EPI_NONE) return gemm_launch<T, float, true, true, CTMI_EPI_NONE>(g, st);
    if constexpr (sizeof(T) == 4) {                         // fp32 storage: output is fp32 either way
        if (!ak && !bk) {
            if (epi == CTMI_EPI_NONE) return gemm_launch<T, float, false, false, CTMI_EPI_NONE>(g, st);
            if (epi == CTMI_EPI_GELU) return gemm_launch<T, float, false, false, CTMI_EPI_GELU>(g, st);
            if (epi == CTMI_EPI_RELU) return gemm_launch<T, float, false, false, CTMI_EPI_RELU>(g, st);
        }
        if (!ak && bk) {
            if (epi == CTMI_EPI_NONE) return gemm_launch<T, float, false, true, CTMI_EPI_NONE>(g, st);
            if (epi == CTMI_EPI_DGELU) return gemm_launch<T, float, false, true, CTMI_EPI_DGELU>(g, st);
            if (epi == CTMI_EPI_DRELU) return gemm_launch<T, float, false, true, CTMI_EPI_DRELU>(g, st);
        }
    }
    ctmi_set_error("gemm: unsupported combination a_kmajor=%d b_kmajor=%d epilogue=%d out_f32=%d", ak, bk, epi, out_f32);
    return CTMI_ERR_UNSUPPORTED;
}

extern "C" int ctmi_gemm(const void* A, int64_t lda, int a_kmajor, const void* B, int64_t ldb, int b_kmajor,
                         void* C, int64_t ldc, int64_t M, int64_t N, int64_t K,
                         float alpha, int beta, const float* bias, const void* residual, int epilogue,
                         const void* aux_in, void* aux_out, int out_f32, int dtype, void* stream) {
    CTMI_REQUIRE(A && B && C, "gemm: null operand");
    CTMI_REQUIRE(M > 0 && N > 0 && K > 0, "gemm: bad shape M=%lld N=%lld K=%lld", (long long)M, (long long)N, (long long)K);
    CTMI_REQUIRE(lda >= (a_kmajor ? M : K) && ldb >= (b_kmajor ? N : K) && ldc >= N, "gemm: leading dimension too small");
    CTMI_REQUIRE(epilogue != CTMI_EPI_GELU || aux_out, "gemm: GELU epilogue needs aux_out");
    CTMI_REQUIRE((epilogue != CTMI_EPI_DGELU && epilogue != CTMI_EPI_DRELU) || aux_in, "gemm: dGELU/dReLU epilogue needs aux_in");
    CTMI_REQUIRE(dtype == CTMI_F32 || dtype == CTMI_BF16, "gemm: unsupported dtype %d", dtype);
    const int es = dtype == CTMI_F32 ? 4 : 2;
    const int vec = 16 / es;
    GemmArgs g;
    g.A = A; g.B = B; g.C = C; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.M = M; g.N = N; g.K = K;
    g.alpha = alpha; g.beta = beta; g.bias = bias; g.residual = residual; g.aux_in = aux_in; g.aux_out = aux_out;
    g.tiles_m = (int)cdiv64(M, 128); g.tiles_n = (int)cdiv64(N, 128);
    CTMI_REQUIRE((int64_t)g.tiles_m * g.tiles_n < (1LL << 31), "gemm: too many tiles");
    g.vec_a = (lda % vec == 0) && ((((uintptr_t)A) & 15) == 0);
    g.vec_b = (ldb % vec == 0) && ((((uintptr_t)B) & 15) == 0);
    // C-side vector accesses are 4 elements wide (8 B bf16 / 16 B fp32)
    auto al = [](const void* p, int bytes) { return p == nullptr || ((((uintptr_t)p) & (bytes - 1)) == 0); };
    const int cbytes = (out_f32 || dtype == CTMI_F32) ? 16 : 8;
    g.vec_c = (ldc % 4 == 0) && al(C, cbytes) && al(residual, 4 * es) && al(aux_in, 4 * es) && al(aux_out, 4 * es);
    if (dtype == CTMI_F32) return gemm_dispatch<float>(g, a_kmajor, b_kmajor, epilogue, out_f32, as_stream(stream));
    return gemm_dispatch<bf16_t>(g, a_kmajor, b_kmajor, epilogue, out_f32, as_stream(stream));
}
