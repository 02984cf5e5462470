// Issue cost (shader cycles per wave64 instruction, s_memtime) of the VALU instructions the attention softmax is made of, with
// 1 or 2 waves per SIMD running the same stream, and beside a partner wave that issues back-to-back MFMAs.
// Build: hipcc --offload-arch=gfx950 -O3 tools/probes/valu_rate.hip -o /tmp/valu_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define REP16(x) x x x x x x x x x x x x x x x x
#define ITERS 64

template <int OP>
__global__ void k(float* out, uint32_t* cyc, int mfma_partner) {
    const int wid = threadIdx.x >> 6;
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b0 = 0.5f, b1 = 0.25f;
    f32x16 acc = {0};
    bf16x8 fa = {1, 1, 1, 1, 1, 1, 1, 1}, fb = fa;
    __syncthreads();
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    if (mfma_partner && wid >= 4) {                      // waves 4-7: MFMA stream (one per SIMD beside waves 0-3)
        for (int i = 0; i < ITERS; ++i) {
#pragma unroll
            for (int j = 0; j < 16; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
        }
    } else {
        for (int i = 0; i < ITERS; ++i) {
            if (OP == 0) asm volatile(REP16("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
            if (OP == 1) asm volatile(REP16("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));
            if (OP == 2) asm volatile(REP16("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4\n") : "+v"(*(double*)&a0), "+v"(*(double*)&a2), "+v"(*(double*)&a4), "+v"(*(double*)&a6) : "v"(*(double*)&b0));
            if (OP == 3) asm volatile(REP16("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n") : "+v"(*(double*)&a0), "+v"(*(double*)&a2), "+v"(*(double*)&a4), "+v"(*(double*)&a6) : "v"(*(double*)&b0));
            if (OP == 4) asm volatile(REP16("v_max3_f32 %0, %0, %4, %5\n v_max3_f32 %1, %1, %4, %5\n v_max3_f32 %2, %2, %4, %5\n v_max3_f32 %3, %3, %4, %5\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));
            if (OP == 5) asm volatile(REP16("v_cvt_pk_bf16_f32 %0, %4, %5\n v_cvt_pk_bf16_f32 %1, %4, %5\n v_cvt_pk_bf16_f32 %2, %4, %5\n v_cvt_pk_bf16_f32 %3, %4, %5\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));
            if (OP == 6) asm volatile(REP16("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));
            if (OP == 7) asm volatile(REP16("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n") : "+v"(*(double*)&a0), "+v"(*(double*)&a2), "+v"(*(double*)&a4), "+v"(*(double*)&a6) : "v"(*(double*)&b0));
            if (OP == 8) asm volatile(REP16("v_exp_f32 %0, %0\n v_fma_f32 %1, %1, %4, %5\n v_exp_f32 %2, %2\n v_fma_f32 %3, %3, %4, %5\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));
            if (OP == 9) asm volatile(REP16("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0) : "vcc");
        }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + wid] = (uint32_t)(t1 - t0);
    float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    for (int j = 0; j < 16; ++j) s += acc[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP> void run(const char* name, float* out, uint32_t* cyc) {
    for (int cfg = 0; cfg < 3; ++cfg) {
        const int threads = cfg == 0 ? 256 : 512, partner = cfg == 2;
        hipMemset(cyc, 0, 4096 * 4);
        hipLaunchKernelGGL(k<OP>, dim3(256), dim3(threads), 0, 0, out, cyc, partner);
        hipDeviceSynchronize();
        uint32_t h[4096];
        hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        const int wpb = threads / 64;
        double valu = 0, mf = 0; int nv = 0, nm = 0;
        for (int b = 0; b < 256; ++b) for (int w = 0; w < wpb; ++w) { if (partner && w >= 4) { mf += h[b * wpb + w]; ++nm; } else { valu += h[b * wpb + w]; ++nv; } }
        printf("%-16s %s: %6.2f cycles/instr", name, cfg == 0 ? "1 wave/SIMD        " : (cfg == 1 ? "2 waves/SIMD same  " : "beside MFMA partner"), valu / nv / (ITERS * 64.0));
        if (partner) printf("   (partner: %6.2f cycles/MFMA)", mf / nm / (ITERS * 16.0));
        printf("\n");
    }
}

int main() {
    float* out; uint32_t* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 4096 * 4);
    run<0>("v_exp_f32", out, cyc); run<1>("v_fma_f32", out, cyc); run<2>("v_pk_fma_f32", out, cyc); run<3>("v_pk_add_f32", out, cyc);
    run<7>("v_pk_mul_f32", out, cyc); run<4>("v_max3_f32", out, cyc); run<5>("v_cvt_pk_bf16_f32", out, cyc); run<6>("v_add_f32", out, cyc);
    run<8>("exp+fma mix", out, cyc); run<9>("v_cndmask_b32", out, cyc);
    return 0;
}
