// The multi-job partial-row reduction of a transformer block's backward (see elementwise.hip: ctmi_reduce_jobs), as a device function two kernels
// share: reduce_jobs_k (elementwise.hip) and wgrad_tail_k (gemm.hip: the same jobs in ONE launch with the sum of the K-halves of a grouped
// weight-gradient call — round 6: one dependent launch per block less).
#pragma once
#include "common.h"

struct ReduceJobs { ctmi_reduce_job j[CTMI_REDUCE_MAX_JOBS]; int chunk0[CTMI_REDUCE_MAX_JOBS + 1]; int count; };

// block -> (job, 64-column chunk); 64 columns x 4 part-slices per block of 256 threads, LDS combine; fixed summation order (deterministic)
__device__ __forceinline__ void reduce_jobs_block(const ReduceJobs& R, int block, float (&sm)[4][64]) {
    int job = 0;
    while (job + 1 < R.count && block >= R.chunk0[job + 1]) ++job;
    const ctmi_reduce_job& J = R.j[job];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int64_t c = (int64_t)(block - R.chunk0[job]) * 64 + tx;
    float t = 0.f;
    if (c < J.n) {
#pragma unroll 8
        for (int p = ty; p < J.nparts; p += 4) t += J.src[(int64_t)p * J.part_stride + c];
    }
    sm[ty][tx] = t;
    __syncthreads();
    if (ty == 0 && c < J.n) {
        const float r = J.alpha * (sm[0][tx] + sm[1][tx] + sm[2][tx] + sm[3][tx]);
        J.dst[c] = J.accumulate ? J.dst[c] + r : r;
    }
}

// host: validate + lay the jobs out in a ReduceJobs record; returns the number of 64-column chunks (blocks), or -1 (ctmi_last_error set)
static inline int reduce_jobs_pack(const ctmi_reduce_job* jobs, int count, ReduceJobs& R) {
    R.count = count;
    int chunks = 0;
    for (int i = 0; i < count; ++i) {
        const ctmi_reduce_job& J = jobs[i];
        if (!(J.src && J.dst && J.n > 0 && J.nparts > 0)) { ctmi_set_error("reduce_jobs: job %d is malformed", i); return -1; }
        R.j[i] = J;
        R.chunk0[i] = chunks;
        chunks += (int)cdiv64(J.n, 64);
    }
    R.chunk0[count] = chunks;
    return chunks;
}
