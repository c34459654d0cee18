// reduce_batch.hpp — one (job, 128 | 32 outputs) block of the deferred split-K / block-partial reductions (kernels.hpp: ReduceJobs), as a
// device function: the body of reduce_batch_kernel (small_kernels.hip) and of the tail that the idle top-layer workgroups of the bf16
// backward pair launch run (lstm_bwd_bf16.hip).  256 threads per block (t = 0..255); every thread of the workgroup must make the call
// (one __syncthreads inside), `active` says whether vblock exists.  Same chains and order per output wherever it runs: bit-identical.
#pragma once
#include "kernels.hpp"

namespace shm {

__device__ __forceinline__ void reduce_batch_block(const ReduceJobs& jobs, const int vblock, const int t, const bool active, f32x4 (*red4)[32],
                                                   float (*red)[32]) {
    int j = 0;
    while (j + 1 < jobs.n && vblock >= jobs.first_block[j + 1]) ++j;
    const ReduceJob jb = jobs.job[j];
    const int o = t & 31, grp = t >> 5;
    const float* __restrict__ partial = jb.partial;
    const size_t n = jb.stride;
    const bool vec = jb.vec != 0;
    // 16 bytes per lane: a block covers 128 outputs, a partial row is read in 512-byte runs (the 4-byte form streamed the 67 MB of a
    // split-K GEMM's partials at 3.9 TB/s); same summation order per output as the scalar form
    const size_t i4 = ((size_t)(vblock - jobs.first_block[j]) * 32 + o) * 4;
    const size_t i = (size_t)(vblock - jobs.first_block[j]) * 32 + o;
    if (active && vec) {
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
        f32x4 s0 = z4, s1 = z4, s2 = z4, s3 = z4, s4 = z4, s5 = z4, s6 = z4, s7 = z4;
        if (i4 < jb.n_out) {
            auto ld = [&](int p) { return *reinterpret_cast<const f32x4*>(partial + (size_t)p * n + i4); };
            int p = grp;
            if (jb.nparts == 128) {                        // the split-K GEMMs' count: all 16 loads of the chain in flight, same adds in the same order
                const f32x4 a0 = ld(p), a1 = ld(p + 8), a2 = ld(p + 16), a3 = ld(p + 24), a4 = ld(p + 32), a5 = ld(p + 40), a6 = ld(p + 48), a7 = ld(p + 56);
                const f32x4 b0 = ld(p + 64), b1 = ld(p + 72), b2 = ld(p + 80), b3 = ld(p + 88), b4 = ld(p + 96), b5 = ld(p + 104), b6 = ld(p + 112), b7 = ld(p + 120);
                s0 += a0; s1 += a1; s2 += a2; s3 += a3; s4 += a4; s5 += a5; s6 += a6; s7 += a7;
                s0 += b0; s1 += b1; s2 += b2; s3 += b3; s4 += b4; s5 += b5; s6 += b6; s7 += b7;
                p += 128;
            }
            for (; p + 56 < jb.nparts; p += 64) {
                s0 += ld(p); s1 += ld(p + 8); s2 += ld(p + 16); s3 += ld(p + 24);
                s4 += ld(p + 32); s5 += ld(p + 40); s6 += ld(p + 48); s7 += ld(p + 56);
            }
            for (; p < jb.nparts; p += 8) s0 += ld(p);
        }
        red4[grp][o] = ((s0 + s1) + (s2 + s3)) + ((s4 + s5) + (s6 + s7));
    } else if (active) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f, s5 = 0.f, s6 = 0.f, s7 = 0.f;
        if (i < jb.n_out) {
            int p = grp;
            for (; p + 56 < jb.nparts; p += 64) {
                s0 += partial[(size_t)p * n + i];
                s1 += partial[(size_t)(p + 8) * n + i];
                s2 += partial[(size_t)(p + 16) * n + i];
                s3 += partial[(size_t)(p + 24) * n + i];
                s4 += partial[(size_t)(p + 32) * n + i];
                s5 += partial[(size_t)(p + 40) * n + i];
                s6 += partial[(size_t)(p + 48) * n + i];
                s7 += partial[(size_t)(p + 56) * n + i];
            }
            for (; p < jb.nparts; p += 8) s0 += partial[(size_t)p * n + i];
        }
        red[grp][o] = ((s0 + s1) + (s2 + s3)) + ((s4 + s5) + (s6 + s7));
    }
    __syncthreads();
    if (active && vec) {
        if (grp == 0 && i4 < jb.n_out) {
            const f32x4 v = ((red4[0][o] + red4[1][o]) + (red4[2][o] + red4[3][o])) + ((red4[4][o] + red4[5][o]) + (red4[6][o] + red4[7][o]));
            *reinterpret_cast<f32x4*>(jb.out0 + i4) = v;
            if (jb.out1) *reinterpret_cast<f32x4*>(jb.out1 + i4) = v;
        }
    } else if (active) {
        if (grp == 0 && i < jb.n_out) {
            const float v = ((red[0][o] + red[1][o]) + (red[2][o] + red[3][o])) + ((red[4][o] + red[5][o]) + (red[6][o] + red[7][o]));
            size_t dst = i;
            if (jb.tr_M > 0) { const size_t nn = i / jb.tr_M, mm = i - nn * jb.tr_M; dst = mm * (jb.n_out / jb.tr_M) + nn; }
            jb.out0[dst] = v;
            if (jb.out1) jb.out1[dst] = v;
        }
    }
}

}  // namespace shm
