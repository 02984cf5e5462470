// What does one LDS-DMA piece (1 KiB per wave-instruction) cost the wave that issues it, and what does a CU sustain?
// Round-4 question behind it: the ping-pong GEMM K-loop takes ~1500-1660 cycles per K-step for 2 x 544 cycles of MFMAs; its load phase is
// 12 ds_read_b128 + 4 DMA pieces per wave, 32 pieces per CU and K-step.  Is the pole the per-wave issue cost, a per-CU rate, or the memory?
// Modes (all: 256 workgroups x 8 waves, one per CU, 128 KiB of dynamic LDS; every wave issues PIECES pieces per trip, <= 12 in flight):
//   form   0 global_load_lds_dwordx4 (64-bit VGPR address, advanced by v_lshl_add_u64)     1 buffer_load_dwordx4 ... offen lds (32-bit VGPR
//          offset, SGPR base advanced by s_add_u32 / s_addc_u32)
//   src    0 hot (every trip re-reads the same 32 KiB per workgroup: L1/L2 hits)   1 streaming (a private 2 MiB window per workgroup, L2 hits
//          after the first pass: 512 MiB total)   2 streaming, 64-byte half-lines as a row-major GEMM operand tile reads them
//   active number of waves of the workgroup that issue (the others idle at the final barrier)
//   mix    0 DMA only    1 12 x ds_read_b128 before the 4 pieces of a trip (the GEMM's load phase)    2 as 1, and waves 4-7 run MFMAs instead
// Output: shader cycles per piece per issuing wave, bytes / cycle / CU.
// Build: hipcc --offload-arch=gfx950 -O3 tools/probes/dma_issue.hip -o tools/probes/dma_issue.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short short8 __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));

#define TRIPS 256
#define PIECES 4

template <int FORM, int SRC, int MIX>
__global__ __launch_bounds__(512, 2) void k(const unsigned char* __restrict__ src, uint32_t* __restrict__ cyc, float* __restrict__ sink, int active) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = (unsigned)(size_t)lds + wid * (PIECES * 4 * 1024);                 // 4 ring slots of PIECES KiB per wave
    const size_t window = SRC == 0 ? 32768 : (SRC == 3 ? 65536 : (size_t)2 << 20);       // SRC 3 (round 5): the half-line pieces of SRC 2, re-read hot (L2 hits)
    const unsigned char* base = src + (size_t)blockIdx.x * ((size_t)2 << 20);
    // lane -> source bytes of a piece: SRC 2: 16 rows x 64 B out of 128-byte-pitch rows (half lines); else 1 KiB contiguous
    const unsigned lane_off = (SRC == 2 || SRC == 3) ? (unsigned)((lane >> 2) * 128 + (lane & 3) * 16) : (unsigned)lane * 16;
    const unsigned piece_bytes = (SRC == 2 || SRC == 3) ? 2048 : 1024;
    const unsigned char* pa[PIECES];
    unsigned voff[PIECES];
#pragma unroll
    for (int j = 0; j < PIECES; ++j) { voff[j] = (wid * PIECES + j) * piece_bytes + lane_off; pa[j] = base + voff[j]; }
    const unsigned step = 8 * PIECES * piece_bytes;                                            // what the 8 waves cover per trip
    v4i srd = {(int)(size_t)base, (int)((size_t)base >> 32), -1, 0x00020000};
    f32x4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    short8 fa = {1, 2, 3, 4, 5, 6, 7, 8}, fb = fa;
    float keep = 0.f;
    __syncthreads();
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    unsigned pos = 0;
    if (MIX == 2 && wid >= 4) {
        for (int t = 0; t < TRIPS; ++t) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc[i], 0, 0, 0);
        }
    } else if (wid < active) {
        for (int t = 0; t < TRIPS; ++t) {
            const unsigned slot = lds0 + (t & 3) * (PIECES * 1024);
            if (MIX >= 1) {
                const unsigned ra = (unsigned)(size_t)lds + 65536 + ((lane * 16 + t * 64) & 16383);
                f32x4 r[12];
#pragma unroll
                for (int i = 0; i < 12; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r[i]) : "v"(ra), "i"(i * 1024));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int i = 0; i < 12; ++i) keep += r[i][0];
            }
            if (FORM == 0) {
                asm volatile("s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off\n\t"
                             "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
                             "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\t"
                             "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, off"
                             :: "v"(pa[0]), "v"(pa[1]), "v"(pa[2]), "v"(pa[3]), "s"(slot) : "memory", "scc");
                pos += step;
                const bool wrap = pos >= window;
#pragma unroll
                for (int j = 0; j < PIECES; ++j) pa[j] = wrap ? base + voff[j] : pa[j] + step;
                if (wrap) pos = 0;
            } else {
                asm volatile("s_mov_b32 m0, %5\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %4, 0 offen lds\n\t"
                             "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %4, 0 offen lds\n\t"
                             "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %4, 0 offen lds\n\t"
                             "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %3, %4, 0 offen lds"
                             :: "v"(voff[0]), "v"(voff[1]), "v"(voff[2]), "v"(voff[3]), "s"(srd), "s"(slot) : "memory", "scc");
                pos += step;
                const bool wrap = pos >= window;
                const size_t nb = (size_t)base + (wrap ? 0 : pos);
                if (wrap) pos = 0;
                srd[0] = (int)nb; srd[1] = (int)(nb >> 32);
            }
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cyc[blockIdx.x * 8 + wid] = (uint32_t)(t1 - t0);
    __syncthreads();
    float s = keep + lds[tid * 4];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0];
    sink[blockIdx.x * 512 + tid] = s;
}

template <int FORM, int SRC, int MIX> void run(const unsigned char* src, uint32_t* cyc, float* sink, int active) {
    auto kern = &k<FORM, SRC, MIX>;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    for (int rep = 0; rep < 2; ++rep) {                                                        // second launch: L2-warm
        hipMemset(cyc, 0, 2048 * 4);
        hipLaunchKernelGGL(kern, dim3(256), dim3(512), 131072, 0, src, cyc, sink, active);
        hipDeviceSynchronize();
    }
    uint32_t h[2048];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double dma = 0, mf = 0; int nd = 0, nm = 0;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < 8; ++w) {
        if (MIX == 2 && w >= 4) { mf += h[b * 8 + w]; ++nm; } else if (w < active) { dma += h[b * 8 + w]; ++nd; }
    }
    const double per_piece = dma / nd / (TRIPS * (double)PIECES);
    const int issuing = (MIX == 2 && active > 4) ? 4 : active;
    printf("%-7s src=%-14s mix=%-22s issuing waves %d: %7.1f cycles/piece/wave  = %6.1f B/clk/CU", FORM == 0 ? "global" : "buffer",
           SRC == 0 ? "hot" : (SRC == 1 ? "stream" : (SRC == 2 ? "stream-halfline" : "hot-halfline")), MIX == 0 ? "dma only" : (MIX == 1 ? "12 ds_read + 4 dma" : "same, beside MFMA"),
           issuing, per_piece, 1024.0 * issuing / per_piece);
    if (MIX == 2) printf("   (MFMA partner: %5.1f cycles/MFMA)", mf / nm / (TRIPS * 32.0));
    printf("\n");
}

int main() {
    unsigned char* src; uint32_t* cyc; float* sink;
    hipMalloc(&src, (size_t)512 << 20); hipMemset(src, 1, (size_t)512 << 20);
    hipMalloc(&cyc, 2048 * 4); hipMalloc(&sink, 256 * 512 * 4);
    for (int active : {1, 2, 4, 8}) { run<0, 0, 0>(src, cyc, sink, active); run<1, 0, 0>(src, cyc, sink, active); }
    for (int active : {4, 8}) { run<0, 1, 0>(src, cyc, sink, active); run<1, 1, 0>(src, cyc, sink, active); run<0, 2, 0>(src, cyc, sink, active); run<1, 2, 0>(src, cyc, sink, active); }
    for (int active : {4, 8}) { run<0, 0, 1>(src, cyc, sink, active); run<1, 0, 1>(src, cyc, sink, active); run<0, 2, 1>(src, cyc, sink, active); run<1, 2, 1>(src, cyc, sink, active); }
    for (int active : {1, 2, 4, 8}) run<0, 3, 0>(src, cyc, sink, active);                    // round 5: hot half-line pieces
    for (int active : {4, 8}) run<0, 3, 1>(src, cyc, sink, active);
    run<0, 3, 2>(src, cyc, sink, 4);
    run<0, 0, 2>(src, cyc, sink, 4); run<1, 0, 2>(src, cyc, sink, 4); run<0, 2, 2>(src, cyc, sink, 4); run<1, 2, 2>(src, cyc, sink, 4);
    return 0;
}
