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

extern "C" int ctmi_probe(int which, const float* in, float* out, void* stream) {
    CTMI_REQUIRE(in && out && which >= 0 && which <= 2, "probe: bad args");
    hipLaunchKernelGGL(probe_kernel, dim3(1), dim3(64), 0, as_stream(stream), which, in, out);
    CTMI_CHECK_LAUNCH("probe");
    return CTMI_OK;
}
