// common.hpp — shared device/host helpers for the gfx950 LSTM-VAE kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/shm_hip.h"

namespace shm {

// ---------------------------------------------------------------- error plumbing
void set_error(const char* fmt, ...);

#define SHM_HIP_CHECK(expr)                                                               \
    do {                                                                                  \
        hipError_t e__ = (expr);                                                          \
        if (e__ != hipSuccess) {                                                          \
            shm::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
            return (int)e__;                                                              \
        }                                                                                 \
    } while (0)

#define SHM_REQUIRE(cond, code, ...)                                                      \
    do {                                                                                  \
        if (!(cond)) { shm::set_error(__VA_ARGS__); return (code); }                      \
    } while (0)

// hipGetLastError() reports the last error of ANY HIP call on this host thread, including calls other libraries in the process
// (PyTorch) made earlier and never looked at.  Every launch therefore clears the slot first, so that SHM_LAUNCH_CHECK() sees this
// launch's status and nothing older.
#define SHM_LAUNCH(...)                                                                   \
    do {                                                                                  \
        (void)hipGetLastError();                                                          \
        hipLaunchKernelGGL(__VA_ARGS__);                                                  \
    } while (0)
#define SHM_LAUNCH_CHECK() SHM_HIP_CHECK(hipGetLastError())

// ---------------------------------------------------------------- MFMA wrappers (gfx950)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// BPTT record of the bf16 path: {i,f | g,o | c_{t-1}, tanh(c_t)}, six bf16 = 12 bytes per lane and step.  A struct, not
// ext_vector_type(3): that type is padded to 16 bytes in memory, which is exactly the quarter this layout saves.
struct __attribute__((aligned(4))) u32x3 {
    unsigned v[3];
    __device__ __forceinline__ unsigned& operator[](int i) { return v[i]; }
    __device__ __forceinline__ const unsigned& operator[](int i) const { return v[i]; }
};
static_assert(sizeof(u32x3) == 12, "12-byte BPTT record");

// D[16x16] += A[16x4] * B[4x16], exact f32 (v_mfma_f32_16x16x4_f32).
// lane l supplies A[row = l&15][k = l>>4] and B[k = l>>4][col = l&15]; holds D[row = 4*(l>>4)+r][col = l&15], r = 0..3.
__device__ __forceinline__ f32x4 mfma_f32_16x16x4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
// D[16x16] += A[16x32] * B[32x16], bf16 in / f32 accumulate (v_mfma_f32_16x16x32_bf16).
// lane l supplies A[row = l&15][k = 8*(l>>4)+e] and B[k = 8*(l>>4)+e][col = l&15], e = 0..7; D as above.
__device__ __forceinline__ f32x4 mfma_bf16_16x16x32(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// D[16x16] += A[16x32] * B[32x16], fp16 in / f32 accumulate (v_mfma_f32_16x16x32_f16); operand layout as the bf16 form.
typedef _Float16 f16_t;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 mfma_f16_16x16x32(f16x8 a, f16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
// D[16x16] += A[16x32] * B[32x16], fp8 (e4m3) in / f32 accumulate (v_mfma_f32_16x16x32_fp8_fp8): lane l supplies 8 consecutive k
// (one byte each, k = 8*(l>>4)+e in byte e of the 64-bit operand) of row / column l&15, as the bf16 form does with 16-bit elements.
__device__ __forceinline__ f32x4 mfma_fp8_16x16x32(long a, long b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a, b, c, 0, 0, 0);
}
// Split-precision input projections (lstm_sp.hip): W_ih ~ fp16(W) + e4m3((W - fp16(W)) * 2^15) * 2^-15 (15 significant bits in 3 bytes
// per element, against 4 for the bf16 pair).  SHM_F8_MAX clamps what goes through v_cvt_pk_fp8_f32 (well inside e4m3's 448).
#define SHM_W8_SCALE 32768.0f
#define SHM_W8_UNSCALE (1.0f / 32768.0f)
#define SHM_F8_MAX 224.0f

// f32 -> bf16 by plain casts: hipcc emits v_cvt_pk_bf16_f32 (round-to-nearest-even, NaN stays NaN)
__device__ __forceinline__ bf16x4 cvt_bf16x4(f32x4 v) { return __builtin_convertvector(v, bf16x4); }
__device__ __forceinline__ f32x4 cvt_f32x4(bf16x4 v) { return __builtin_convertvector(v, f32x4); }

// LDS image of a [16 windows][K] bf16 operand tile, quarter-major: element (j,k) lives at
//   ((k / KQ) * 16 + j) * RS + (k % KQ),   RS = KQ + pad  with RS*2/16 odd
// so that the 16 lanes of every ds_read_b128 group (fixed k-step, 16 windows across two quarters) hit 16 distinct
// 16-byte slots of the 256-byte bank row: conflict-free operand reads.
__host__ __device__ constexpr int lds_row_stride(int KQ) { return KQ + ((KQ / 8) % 2 == 0 ? 8 : 16); }

// ---------------------------------------------------------------- activation math
// Parity path (fp32): ocml expf/tanhf and IEEE division.
__device__ __forceinline__ float sigmoid_acc(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float tanh_acc(float x) { return tanhf(x); }
// Throughput path (bf16 operands): v_exp_f32 / v_rcp_f32 hardware approximations (~1 ulp each).
__device__ __forceinline__ float sigmoid_fast(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float tanh_fast(float x) {
    // tanh(x) = 1 - 2/(1+exp(2x)); saturates cleanly for |x| large (exp2 -> inf or 0)
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.8853900817779268f * x));
}

// ---------------------------------------------------------------- dropout counter RNG
// keep-decision for element idx of stream (seed, stack); identical on host (shm_dropout_keep_mask) and device.
__host__ __device__ __forceinline__ uint32_t drop_hash(uint64_t seed, uint32_t stack, uint32_t idx) {
    uint32_t x = idx * 0x9E3779B1u ^ (uint32_t)seed;
    x ^= x >> 16; x *= 0x85EBCA6Bu;
    x ^= (uint32_t)(seed >> 32) + stack * 0x632BE5ABu;
    x ^= x >> 13; x *= 0xC2B2AE35u;
    x ^= x >> 16; x *= 0x27D4EB2Fu;
    x ^= x >> 15;
    return x;
}
__host__ __device__ __forceinline__ uint32_t drop_threshold(float p) {   // keep iff hash >= threshold
    double t = (double)p * 4294967296.0;
    return t >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)t;
}

// ---------------------------------------------------------------- reparameterisation noise (counter RNG)
// eps[i] ~ N(0,1) of stream (seed, offset): Philox4x32-10 on counter (i, 0, offset_lo, offset_hi), key = seed, then Box-Muller on the
// first two words.  Stateless, so the same element gets the same draw wherever it is computed (head_fwd on the device, shm_eps_normal
// on the host for tests); replaces torch.randn_like at temporal_vae.py:62 on the fused train step (same distribution, not the same numbers).
__host__ __device__ __forceinline__ void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}
__host__ __device__ __forceinline__ float eps_normal(uint64_t seed, uint64_t offset, uint32_t idx) {
    uint32_t c[4] = {idx, 0u, (uint32_t)offset, (uint32_t)(offset >> 32)};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    const float u1 = ((float)(c[0] >> 8) + 0.5f) * (1.0f / 16777216.0f);      // (0, 1): 24 random bits, never 0
    const float u2 = ((float)(c[1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
    return sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2);
}

// Polling a progress word of an in-launch hand-off (pair launches).  The word is wave-uniform, so it is read with a SCALAR load (glc: served
// by L2, not by the scalar cache), which waits on lgkmcnt: a poll then does not drain the wave's vector-memory queue.  With a vector load
// (vmcnt retires in order) every poll waited for all the record / operand prefetches in flight, and a role that had caught up with its
// producer emptied its prefetch queue once per hand-off chunk (bf16 backward pair 0.222 -> 0.216 ms per launch with the scalar form).
// uniform_addr: the word's address in an SGPR pair (the readfirstlane builtin returns int: cast before widening, or a low half with bit
// 31 set sign-extends into the high half — a wild address).
__device__ __forceinline__ uint64_t uniform_addr(const void* p) {
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32) |
           (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
}
__device__ __forceinline__ unsigned poll_uniform(uint64_t addr_s) {
    unsigned v;
    asm volatile("s_load_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(addr_s) : "memory");
    return v;
}

__host__ __device__ __forceinline__ int ceil_div(int a, int b) { return (a + b - 1) / b; }
__host__ __device__ __forceinline__ int round_up(int a, int b) { return ceil_div(a, b) * b; }

}  // namespace shm
