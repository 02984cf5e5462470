// Hardware layout probes (diagnostics only; exercised by tests/test_gpu_ops.py::test_probe_*, never by the product path).
//   which = 0: ds_read_b64_tr_b16.  LDS holds u16[i] = i (4096 entries); lane l reads at byte address in[l]
//              (host-supplied, as floats) and dumps its four 16-bit results to out[l*4 .. l*4+3].
//   which = 1: v_mfma_f32_16x16x32_bf16 with A[i][k] = in[i*32+k], B[k][j] = in[512 + k*16 + j] loaded under the
//              layout assumption of mma.h; D is dumped as out[lane*4 + r].
//   which = 2: v_mfma_f32_16x16x4_f32, A[i][k] = in[i*4+k], B[k][j] = in[64 + k*16 + j]; same dump.
#include "common.h"
#include "mma.h"

__global__ void probe_kernel(int which, const float* __restrict__ in, float* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
    const int lane = threadIdx.x;
    if (which == 0) {
        for (int i = lane; i < 4096; i += 64) lds[i] = (unsigned short)i;
        __syncthreads();
        const unsigned addr = (unsigned)(size_t)lds + (unsigned)in[lane];
        uint2 r;
        asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr) : "memory");
        out[lane * 4 + 0] = (float)(r.x & 0xffffu); out[lane * 4 + 1] = (float)(r.x >> 16);
        out[lane * 4 + 2] = (float)(r.y & 0xffffu); out[lane * 4 + 3] = (float)(r.y >> 16);
    } else if (which == 1) {
        short8 a, b;
        const int i = lane & 15, g = lane >> 4;
        for (int j = 0; j < 8; ++j) {
            a[j] = (short)f2bf(in[i * 32 + g * 8 + j]);
            b[j] = (short)f2bf(in[512 + (g * 8 + j) * 16 + i]);
        }
        f32x4 c = {0.f, 0.f, 0.f, 0.f};
        c = Mma<bf16_t>::mma(a, b, c);
        for (int r = 0; r < 4; ++r) out[lane * 4 + r] = c[r];
    } else {
        const int i = lane & 15, g = lane >> 4;
        f32x4 c = {0.f, 0.f, 0.f, 0.f};
        c = Mma<float>::mma(in[i * 4 + g], in[64 + g * 16 + i], c);
        for (int r = 0; r < 4; ++r) out[lane * 4 + r] = c[r];
    }
}

// Shader-clock probe (bench.py's `timing.shader_clock_mhz_*`): every workgroup runs `iters` dependent-free bf16 MFMAs per wave (the load the
// training step puts on the chip: power is what sets the clock, the guide's "DVFS give-back"), and wave 0 of workgroup 0 brackets them with
// s_memtime (shader cycles) and s_memrealtime (the constant 100 MHz reference counter): clock = 100 MHz x d(memtime) / d(memrealtime).
__global__ __launch_bounds__(256) void clock_probe_kernel(int iters, unsigned long long* __restrict__ out) {
    short8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (short)(0x3f80 + threadIdx.x + j); b[j] = (short)(0x3f00 + 3 * threadIdx.x + j); }
    f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0, c2 = c0, c3 = c0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
        c0 = Mma<bf16_t>::mma(a, b, c0); c1 = Mma<bf16_t>::mma(a, b, c1);
        c2 = Mma<bf16_t>::mma(a, b, c2); c3 = Mma<bf16_t>::mma(a, b, c3);
    }
    asm volatile("" :: "v"(c0), "v"(c1), "v"(c2), "v"(c3));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (blockIdx.x == 0 && threadIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; }
}

extern "C" int ctmi_clock_probe(int mfma_iters, unsigned long long* out, void* stream) {
    CTMI_REQUIRE(out && mfma_iters > 0 && mfma_iters <= (1 << 22), "clock_probe: bad args");
    hipLaunchKernelGGL(clock_probe_kernel, dim3(2048), dim3(256), 0, as_stream(stream), mfma_iters, out);
    CTMI_CHECK_LAUNCH("clock_probe");
    return CTMI_OK;
}

extern "C" int ctmi_probe(int which, const float* in, float* out, void* stream) {
    CTMI_REQUIRE(in && out && which >= 0 && which <= 2, "probe: bad args");
    hipLaunchKernelGGL(probe_kernel, dim3(1), dim3(64), 0, as_stream(stream), which, in, out);
    CTMI_CHECK_LAUNCH("probe");
    return CTMI_OK;
}

// Dynamic-LDS opt-in probe (tests only): asks for `bytes` of dynamic LDS for a trivial kernel through the same ctmi_dyn_lds every product launch
// uses, and launches it.  A request beyond the CU's 160 KiB must come back as CTMI_ERR_LAUNCH with the attribute call's message — not be dropped.
__global__ void dyn_lds_probe_kernel(unsigned* __restrict__ out) {
    extern __shared__ unsigned dyn_lds_probe_smem[];
    dyn_lds_probe_smem[threadIdx.x] = threadIdx.x;
    __syncthreads();
    if (threadIdx.x == 0 && out) out[0] = dyn_lds_probe_smem[63];
}
extern "C" int ctmi_probe_dyn_lds(int64_t bytes, unsigned* out, void* stream) {
    CTMI_REQUIRE(bytes >= 256, "probe_dyn_lds: at least 256 bytes");
    ctmi_dyn_lds(reinterpret_cast<const void*>(&dyn_lds_probe_kernel), (size_t)bytes);
    hipLaunchKernelGGL(dyn_lds_probe_kernel, dim3(1), dim3(64), (size_t)bytes, as_stream(stream), out);
    CTMI_CHECK_LAUNCH("probe_dyn_lds");
    return CTMI_OK;
}
