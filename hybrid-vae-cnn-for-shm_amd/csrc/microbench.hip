// microbench.hip — measured ceilings of this box, quoted beside the nominal peaks in DESIGN.md ("verify on the box with a
// microbenchmark and quote the measured peak too"); gfx950.  Not part of any product path.
//   shm_peak_mfma: back-to-back independent MFMAs from every wave of the chip (bf16 16x16x32 or fp32 16x16x4), operands in registers
//   shm_peak_hbm:  streaming read (sum), write (fill) or copy of a large buffer with 16-byte accesses
#include "kernels.hpp"

namespace shm {
namespace {

template <bool BF16>
__global__ void __launch_bounds__(256) peak_mfma_kernel(int iters, float* out) {
    f32x4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float seed = (float)(threadIdx.x & 7) * 0.125f;
    bf16x8 a8, b8;
#pragma unroll
    for (int e = 0; e < 8; ++e) { a8[e] = (bf16_t)(seed + 0.01f * e); b8[e] = (bf16_t)(1.0f - seed); }
    const float af = seed + 0.5f, bfv = 1.0f - seed;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if constexpr (BF16) acc[i] = mfma_bf16_16x16x32(a8, b8, acc[i]);
            else acc[i] = mfma_f32_16x16x4(af, bfv, acc[i]);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678f) out[0] = s;                       // keeps the accumulators alive; never true
}

// mode 0: read (sum), 1: write, 2: copy
__global__ void __launch_bounds__(256) peak_hbm_kernel(const f32x4* __restrict__ src, f32x4* __restrict__ dst, size_t n16, int mode, float* out) {
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    const f32x4 fill = {1.f, 2.f, 3.f, 4.f};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        if (mode == 0) { const f32x4 v = src[i]; s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3]; }
        else if (mode == 1) dst[i] = fill;
        else dst[i] = src[i];
    }
    if (mode == 0 && s[0] + s[1] + s[2] + s[3] == 12345.678f) out[0] = s[0];
}

}  // namespace
}  // namespace shm

using namespace shm;

// TFLOP/s of `iters` x 8 independent MFMAs per wave on `waves_per_cu` x 256 CUs x ... (grid = 256 * blocks_per_cu workgroups of 4 waves)
extern "C" int shm_peak_mfma(int32_t bf16, int32_t blocks_per_cu, int32_t iters, float* scratch, double* tflops, void* stream) {
    SHM_REQUIRE(blocks_per_cu >= 1 && iters >= 1 && scratch && tflops, SHM_EINVAL, "peak_mfma: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    hipEvent_t e0, e1;
    SHM_HIP_CHECK(hipEventCreate(&e0)); SHM_HIP_CHECK(hipEventCreate(&e1));
    const int blocks = 256 * blocks_per_cu;
    for (int rep = 0; rep < 2; ++rep) {                    // first pass warms clocks / code
        SHM_HIP_CHECK(hipEventRecord(e0, s));
        if (bf16) SHM_LAUNCH((peak_mfma_kernel<true>), dim3(blocks), dim3(256), 0, s, iters, scratch);
        else SHM_LAUNCH((peak_mfma_kernel<false>), dim3(blocks), dim3(256), 0, s, iters, scratch);
        SHM_LAUNCH_CHECK();
        SHM_HIP_CHECK(hipEventRecord(e1, s));
        SHM_HIP_CHECK(hipEventSynchronize(e1));
    }
    float ms = 0.f;
    SHM_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double flop_per_mfma = bf16 ? 2.0 * 16 * 16 * 32 : 2.0 * 16 * 16 * 4;
    *tflops = (double)blocks * 4 * iters * 8 * flop_per_mfma / (ms * 1e-3) / 1e12;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return SHM_OK;
}

// GB/s of one pass over `bytes` (mode 0 read, 1 write, 2 copy: read + write counted)
extern "C" int shm_peak_hbm(const void* src, void* dst, size_t bytes, int32_t mode, float* scratch, double* gbps, void* stream) {
    SHM_REQUIRE(bytes >= 16 && scratch && gbps && mode >= 0 && mode <= 2 && (mode == 1 || src) && (mode == 0 || dst), SHM_EINVAL, "peak_hbm: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    hipEvent_t e0, e1;
    SHM_HIP_CHECK(hipEventCreate(&e0)); SHM_HIP_CHECK(hipEventCreate(&e1));
    const size_t n16 = bytes / 16;
    for (int rep = 0; rep < 2; ++rep) {
        SHM_HIP_CHECK(hipEventRecord(e0, s));
        SHM_LAUNCH(peak_hbm_kernel, dim3(256 * 16), dim3(256), 0, s, (const f32x4*)src, (f32x4*)dst, n16, mode, scratch);
        SHM_LAUNCH_CHECK();
        SHM_HIP_CHECK(hipEventRecord(e1, s));
        SHM_HIP_CHECK(hipEventSynchronize(e1));
    }
    float ms = 0.f;
    SHM_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    *gbps = (double)n16 * 16 * (mode == 2 ? 2 : 1) / (ms * 1e-3) / 1e9;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return SHM_OK;
}
