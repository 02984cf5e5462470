// What does a dependent kernel boundary cost when the predecessor leaves dirty lines in the XCDs' L2s — and does a write-through store policy
// move that cost into the predecessor's own (compute-covered) run time?
//
//   W<POL>(bytes, spin): 1024 workgroups x 256 threads; every thread stores 16-byte pieces of its workgroup's contiguous chunk (row-contiguous 4 KiB
//   per workgroup-instruction) and, between two stores, runs `spin` dependent FMAs — spin > 0 turns the writer into a compute-bound kernel that
//   trickles its output (what a GEMM epilogue stream looks like from the memory side).
//   POL: 0 plain, 1 sc1, 2 sc0 sc1, 3 nt
//   R(bytes): a reader of the same buffer (the dependent successor: must see the bytes -> the boundary carries the release)
//
// Reported per (policy, bytes, spin): chain of N x [W ; R(4 KiB)] minus N x the time of ONE big W covering N x bytes -> the per-boundary cost.
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/boundary_dirty.hip -o tools/probes/boundary_dirty.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int POL> __device__ __forceinline__ void st16(void* p, u32x4 v) {
    if constexpr (POL == 0) *reinterpret_cast<u32x4*>(p) = v;
    else if constexpr (POL == 1) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(v) : "memory");
    else if constexpr (POL == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" :: "v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" :: "v"(p), "v"(v) : "memory");
}

template <int POL>
__global__ __launch_bounds__(256) void W(unsigned char* buf, size_t bytes, int spin, float seed) {
    const size_t per_wg = bytes / gridDim.x;                               // multiple of 4096 by construction
    unsigned char* base = buf + (size_t)blockIdx.x * per_wg;
    float a = seed + threadIdx.x;
    for (size_t off = (size_t)threadIdx.x * 16; off < per_wg; off += 4096) {
        for (int s = 0; s < spin; ++s) a = __builtin_fmaf(a, 1.0000001f, 0.5f);
        const unsigned w = __float_as_uint(a);
        st16<POL>(base + off, u32x4{w, w + 1, w + 2, w + 3});
    }
}

__global__ __launch_bounds__(256) void R(const unsigned char* buf, size_t bytes, unsigned* out) {
    // reads 16 bytes per thread from the END of the buffer (the last bytes the writer produced)
    const u32x4 v = *reinterpret_cast<const u32x4*>(buf + bytes - 4096 + threadIdx.x * 16);
    if (v[0] == 0xdeadbeefu) out[0] = v[1];
}

template <int POL>
static float run_chain(unsigned char* buf, size_t bytes, int spin, int N, unsigned* out, hipStream_t st, bool with_reader) {
    // the chain is captured and replayed as ONE hipGraph launch, so the host's launch rate (~3.5 us per launch) is not what is measured
    hipGraph_t graph; hipGraphExec_t exec;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int i = 0; i < N; ++i) {
        hipLaunchKernelGGL(W<POL>, dim3(1024), dim3(256), 0, st, buf, bytes, spin, (float)i);
        if (with_reader) hipLaunchKernelGGL(R, dim3(1), dim3(256), 0, st, buf, bytes, out);
    }
    CK(hipStreamEndCapture(st, &graph));
    CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0, st));
        CK(hipGraphLaunch(exec, st));
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0) best = std::min(best, ms);
    }
    CK(hipGraphExecDestroy(exec)); CK(hipGraphDestroy(graph));
    return best * 1e3f / N;                                                  // us per link
}

template <int POL>
static float run_one_big(unsigned char* buf, size_t bytes, int spin, int N, hipStream_t st) {
    // the same total stores in ONE launch (N x more chunk per workgroup, wrapping inside the buffer is not possible -> the buffer is N x bytes)
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0, st));
        hipLaunchKernelGGL(W<POL>, dim3(1024), dim3(256), 0, st, buf, bytes * N, spin, 1.0f);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0) best = std::min(best, ms);
    }
    return best * 1e3f / N;
}

template <int POL>
static void sweep(const char* name, unsigned char* buf, unsigned* out, hipStream_t st) {
    const int N = 16;
    for (int spin : {0, 256, 1024}) {
        for (size_t mb : {1, 4, 16, 64}) {
            const size_t bytes = mb << 20;
            const float chain = run_chain<POL>(buf, bytes, spin, N, out, st, true);
            const float chain_nr = run_chain<POL>(buf, bytes, spin, N, out, st, false);
            const float big = run_one_big<POL>(buf, bytes, spin, N, st);
            printf("%-8s spin %3d  %3zu MiB: [W;R] link %8.2f us | [W] link %8.2f us | 1/N of one big W %8.2f us | boundary+flush = %6.2f (W;R) / %6.2f (W only)\n",
                   name, spin, mb, chain, chain_nr, big, chain - big, chain_nr - big);
        }
    }
}

int main() {
    unsigned char* buf; unsigned* out;
    const size_t cap = (size_t)16 * 64 << 20;                               // 1 GiB
    CK(hipMalloc(&buf, cap)); CK(hipMalloc(&out, 64));
    CK(hipMemset(buf, 0, cap));
    hipStream_t st; CK(hipStreamCreate(&st));
    // trivial boundary for reference
    {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < 2000; ++i) hipLaunchKernelGGL(R, dim3(1), dim3(256), 0, st, buf, (size_t)4096, out);
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep == 2) printf("trivial kernel chain: %.2f us per launch (host-bound if > ~2)\n", ms * 1e3f / 2000);
        }
    }
    sweep<0>("plain", buf, out, st);
    sweep<1>("sc1", buf, out, st);
    sweep<2>("sc0sc1", buf, out, st);
    sweep<3>("nt", buf, out, st);
    return 0;
}
