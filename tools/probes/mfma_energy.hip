// What does a matrix instruction cost in ENERGY on this chip?  Bare MFMA streams (no memory, no LDS: operands and accumulators live in registers),
// two waves per SIMD on every CU, for ~1 s each, with the package power sampled from rocm-smi beside them and the shader clock from s_memtime /
// s_memrealtime:   bf16 16x16x32 | bf16 32x32x16 | f16 16x16x32 | f16 32x32x16   x   operands random N(0, sigma) / all zero.
// All four shapes have the same peak rate (1024 FLOP per cycle and SIMD); the step is power-limited (profiles/r04_power_samples.txt), the half-
// precision step runs 0.8 - 1.2 ms faster than the bf16 one on the same kernels (profiles/r05_dtype_step.txt): this probe asks the matrix pipe alone.
// Build: hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_energy.hip -o tools/probes/mfma_energy.bin ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <thread>
#include <atomic>
#include <vector>
#include <chrono>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short short8 __attribute__((ext_vector_type(8)));

// KIND: 0 bf16 16x16x32, 1 bf16 32x32x16, 2 f16 16x16x32, 3 f16 32x32x16.  Every wave: 8 A and 8 B fragments (different data), 16 / 8 independent accumulators
template <int KIND>
__global__ __launch_bounds__(512) void k(const short8* __restrict__ ops, float* out, unsigned long long* clk, int iters) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    short8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = ops[((blockIdx.x * 8 + w) * 8 + i) * 64 + lane]; b[i] = ops[((blockIdx.x * 8 + w) * 8 + 4 + i) * 64 + lane]; }
    unsigned long long t0 = 0, r0 = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    float sum = 0.f;
    if constexpr (KIND == 0 || KIND == 2) {
        f32x4 c[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) c[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if constexpr (KIND == 0) c[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[i]), __builtin_bit_cast(bf16x8, b[j]), c[i * 4 + j], 0, 0, 0);
                    else c[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a[i]), __builtin_bit_cast(f16x8, b[j]), c[i * 4 + j], 0, 0, 0);
                }
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) sum += c[i][0] + c[i][3];
    } else {
        f32x16 c[8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) c[i][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
            // 8 MFMAs of 32x32x16 = the FLOPs of 16 MFMAs of 16x16x32: (a_i, b_j) over a 4 x 2 grid
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if constexpr (KIND == 1) c[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i]), __builtin_bit_cast(bf16x8, b[j]), c[i * 2 + j], 0, 0, 0);
                    else c[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[i]), __builtin_bit_cast(f16x8, b[j]), c[i * 2 + j], 0, 0, 0);
                }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) sum += c[i][0] + c[i][15];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = __builtin_amdgcn_s_memtime() - t0; clk[1] = __builtin_amdgcn_s_memrealtime() - r0; }
    if (sum == 12345.678f) out[threadIdx.x] = sum;                      // keep the accumulators alive
}

static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
static uint16_t f2h(float f) { _Float16 h = (_Float16)f; uint16_t u; memcpy(&u, &h, 2); return u; }
static float gauss() { float u1 = (rand() + 1.0f) / (RAND_MAX + 2.0f), u2 = rand() / (float)RAND_MAX; return sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2); }

static double smi_power() {
    FILE* f = popen("/opt/rocm/bin/rocm-smi -d 0 --showpower 2>/dev/null", "r");
    if (!f) return -1;
    char line[512]; double w = -1;
    while (fgets(line, sizeof line, f)) { const char* p = strstr(line, "Package Power (W):"); if (p) w = atof(p + 18); }
    pclose(f);
    return w;
}

template <int KIND> void run(const char* name, const short8* d_ops, float* d_out, unsigned long long* d_clk, double idle_w) {
    const int grid = 256, iters0 = 20000;
    // calibrate
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(512), 0, 0, d_ops, d_out, d_clk, iters0);
    hipEventRecord(e0); hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(512), 0, 0, d_ops, d_out, d_clk, iters0); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms0; hipEventElapsedTime(&ms0, e0, e1);
    const int iters = (int)(iters0 * (1200.0 / ms0));                  // ~1.2 s
    std::atomic<bool> stop{false};
    std::vector<double> samples;
    std::thread th([&] { std::this_thread::sleep_for(std::chrono::milliseconds(250)); while (!stop) { double w = smi_power(); if (w > 0 && !stop) samples.push_back(w); } });
    hipEventRecord(e0); hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(512), 0, 0, d_ops, d_out, d_clk, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    stop = true; th.join();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long clk[2]; hipMemcpy(clk, d_clk, sizeof clk, hipMemcpyDeviceToHost);
    const double flop = (double)grid * 8 * iters * 16 * (2.0 * 16 * 16 * 32);
    double w = 0; int n = 0; for (size_t i = 0; i + 1 < samples.size(); ++i) { w += samples[i]; ++n; }   // (the last sample may straddle the end)
    w = n ? w / n : -1;
    printf("%-44s %7.1f ms  %7.1f TF/s  clock %5.0f MHz  %6.0f W (%2d samples)  %5.2f J/TFLOP  dynamic %5.2f\n", name, ms, flop / ms / 1e9,
           100.0 * clk[0] / clk[1], w, n, w * ms * 1e-3 / (flop / 1e12), (w - idle_w) * ms * 1e-3 / (flop / 1e12));
    fflush(stdout);
}

int main(int argc, char** argv) {
    const float sigma = argc > 1 ? atof(argv[1]) : 0.5f;
    const size_t nfrag = 256 * 8 * 8 * 64;                            // short8 fragments: per workgroup, wave, 8 fragments, lane
    std::vector<uint16_t> hb(nfrag * 8), hh(nfrag * 8), hz(nfrag * 8, 0);
    srand(7);
    for (size_t i = 0; i < hb.size(); ++i) { const float x = gauss() * sigma; hb[i] = f2bf(x); hh[i] = f2h(x); }
    short8 *d_b, *d_h, *d_z; float* d_out; unsigned long long* d_clk;
    hipMalloc(&d_b, nfrag * 16); hipMalloc(&d_h, nfrag * 16); hipMalloc(&d_z, nfrag * 16); hipMalloc(&d_out, 4096); hipMalloc(&d_clk, 64);
    hipMemcpy(d_b, hb.data(), nfrag * 16, hipMemcpyHostToDevice); hipMemcpy(d_h, hh.data(), nfrag * 16, hipMemcpyHostToDevice); hipMemcpy(d_z, hz.data(), nfrag * 16, hipMemcpyHostToDevice);
    hipDeviceSynchronize();
    std::this_thread::sleep_for(std::chrono::milliseconds(1500));
    double idle = 0; for (int i = 0; i < 4; ++i) idle += smi_power(); idle /= 4;
    printf("bare MFMA streams, 256 workgroups x 8 waves (2 per SIMD), operands N(0, %.2f) rounded to the operand type; idle package power %.0f W\n", sigma, idle);
    for (int rep = 0; rep < 2; ++rep) {
        run<0>("bf16 v_mfma_f32_16x16x32_bf16", d_b, d_out, d_clk, idle);
        run<1>("bf16 v_mfma_f32_32x32x16_bf16", d_b, d_out, d_clk, idle);
        run<2>("fp16 v_mfma_f32_16x16x32_f16", d_h, d_out, d_clk, idle);
        run<3>("fp16 v_mfma_f32_32x32x16_f16", d_h, d_out, d_clk, idle);
    }
    run<0>("bf16 16x16x32, ALL-ZERO operands", d_z, d_out, d_clk, idle);
    run<1>("bf16 32x32x16, ALL-ZERO operands", d_z, d_out, d_clk, idle);
    run<2>("fp16 16x16x32, ALL-ZERO operands", d_z, d_out, d_clk, idle);
    return 0;
}
