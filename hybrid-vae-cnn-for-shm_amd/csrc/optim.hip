// optim.hip — clip_grad_norm_ + Adam over flat fp32 buffers (gfx950).
//
// Replaces torch.nn.utils.clip_grad_norm_(vae.parameters(), 2.0) and torch.optim.Adam(lr, weight_decay).step()
// (reference 4DOF/Scripts/03_train_vae.py:222,270-271) with two launches over ONE flat buffer of all tensors:
//   1. per-block partial sums of squares (fixed order);
//   2. every block re-reduces the partials in the same order (so every block — and every data-parallel rank —
//      sees the bit-identical norm), derives coef = min(1, max_norm / (norm + 1e-6)) and applies
//      g = coef*g + wd*p;  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;  p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps).
#include "kernels.hpp"

namespace shm {

namespace {

constexpr int NORM_BLOCKS = 256;

__device__ __forceinline__ float block_sum_256(float v, float* red) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    const float r = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
    return r;
}

__global__ void __launch_bounds__(256) sumsq_kernel(const float* __restrict__ g, int64_t n, float scale, float* __restrict__ partial) {
    __shared__ float red[4];
    float s = 0.f;
    const int64_t n4 = n >> 2;
    const float4* g4 = reinterpret_cast<const float4*>(g);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)NORM_BLOCKS * 256) {
        float4 v = g4[i];
        v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
        s += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
    if (blockIdx.x == 0)
        for (int64_t i = (n4 << 2) + threadIdx.x; i < n; i += 256) { const float v = g[i] * scale; s += v * v; }
    s = block_sum_256(s, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

__global__ void __launch_bounds__(256) adam_kernel(int64_t n, float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, float step_size,
                                                   float inv_sqrt_bc2, float b1, float b2, float eps, float wd,
                                                   float max_norm, float grad_scale, const float* __restrict__ partial,
                                                   float* __restrict__ out_norm, const uint32_t* __restrict__ status_src,
                                                   uint32_t* __restrict__ status_dst) {
    __shared__ float red[4];
    // partial == nullptr: neither clipping nor the norm itself was asked for (the sumsq launch was skipped)
    const float tot = partial ? block_sum_256(partial[threadIdx.x], red) : 0.f;     // NORM_BLOCKS == blockDim.x
    const float norm = sqrtf(tot);
    float coef = grad_scale;
    if (max_norm > 0.f) coef *= fminf(1.0f, max_norm / (norm + 1e-6f));
    if (blockIdx.x == 0 && threadIdx.x == 0 && out_norm) *out_norm = norm;
    // the caller's per-step status word (a plan's sticky hand-off word) rides along: no 4-byte copy launch of its own
    if (blockIdx.x == 0 && threadIdx.x == 0 && status_dst) *status_dst = status_src ? *status_src : 0u;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float pi = p[i];
        const float gi = g[i] * coef + wd * pi;
        const float mi = b1 * m[i] + (1.0f - b1) * gi;
        const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        p[i] = pi - step_size * (mi / (sqrtf(vi) * inv_sqrt_bc2 + eps));
    }
}

}  // namespace
}  // namespace shm

extern "C" size_t shm_adam_scratch_bytes(int64_t) { return sizeof(float) * shm::NORM_BLOCKS; }

extern "C" int shm_adam_step(int64_t n, float* params, float* grads, float* exp_avg, float* exp_avg_sq, int32_t step,
                             float lr, float beta1, float beta2, float eps, float weight_decay, float max_norm,
                             float grad_scale, void* scratch, float* out_norm, void* stream) {
    return shm_adam_step_status(n, params, grads, exp_avg, exp_avg_sq, step, lr, beta1, beta2, eps, weight_decay, max_norm, grad_scale, scratch,
                                out_norm, nullptr, nullptr, stream);
}

extern "C" int shm_adam_step_status(int64_t n, float* params, float* grads, float* exp_avg, float* exp_avg_sq, int32_t step,
                                    float lr, float beta1, float beta2, float eps, float weight_decay, float max_norm,
                                    float grad_scale, void* scratch, float* out_norm, const uint32_t* status_src, uint32_t* status_dst,
                                    void* stream) {
    using namespace shm;
    SHM_REQUIRE(n > 0 && params && grads && exp_avg && exp_avg_sq && scratch, SHM_EINVAL, "shm_adam_step: null buffer or n <= 0");
    SHM_REQUIRE(step >= 1, SHM_EINVAL, "shm_adam_step: step is 1-based, got %d", step);
    SHM_REQUIRE((reinterpret_cast<uintptr_t>(grads) & 15) == 0, SHM_EINVAL, "shm_adam_step: grads must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    float* partial = (float*)scratch;
    // the norm costs a launch of its own: taken only when something consumes it (clipping, or the caller wants it reported)
    const bool need_norm = max_norm > 0.f || out_norm != nullptr;
    if (need_norm) {
        SHM_LAUNCH(sumsq_kernel, dim3(NORM_BLOCKS), dim3(256), 0, s, grads, n, grad_scale, partial);
        SHM_LAUNCH_CHECK();
    } else partial = nullptr;
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    const float step_size = (float)(lr / bc1);
    const float inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    const int blocks = (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
    SHM_LAUNCH(adam_kernel, dim3(blocks), dim3(256), 0, s, n, params, grads, exp_avg, exp_avg_sq, step_size, inv_sqrt_bc2,
                       beta1, beta2, eps, weight_decay, max_norm, grad_scale, partial, out_norm, status_src, status_dst);
    SHM_LAUNCH_CHECK();
    return SHM_OK;
}
