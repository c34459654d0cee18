// lstm_fwd.hip — persistent LSTM layer forward, fp32 MFMA (v_mfma_f32_16x16x4_f32), gfx950.
//
// Replaces the nn.LSTM cell loop behind TemporalVAE.encode/decode
// (reference 4DOF/Scripts/Models/temporal_vae.py:53,69).  One workgroup owns 16 windows (batch rows) and walks
// all T steps; the recurrence never leaves the CU:
//   gates^T[4H x 16] = W_ih[4H x In] x_t^T + W_hh[4H x H] h_{t-1}^T + b        (MFMA, A = weights, B = activations)
//   c = sig(f) c + sig(i) tanh(g);  h = sig(o) tanh(c)                          (in the accumulator's own lanes)
// Wave w of the workgroup owns hidden units [16w, 16w+16) and the four gate tiles (i,f,g,o) of those units, so the
// cell update needs no cross-lane traffic: lane (q = lane>>4, j = lane&15) holds units 16w+4q+r (r=0..3) of window j.
// h_t goes back to LDS ([window][unit], double-buffered) as the next step's B operand; one barrier per step.
// The K index is split so that lane-quarter q owns the contiguous k-range [q*KQ, (q+1)*KQ): a single ds_read_b128
// then feeds four MFMA k-steps (the packed weights use the same split, see kernels.hpp).
#include "kernels.hpp"

namespace shm {

namespace {

constexpr int PAD = 4;   // LDS row pad (floats): <=2-way conflicts on the b128 operand reads

template <int H, bool DROP>
__global__ void __launch_bounds__(4 * H) lstm_fwd_f32_kernel(LstmFwdArgs a) {
    constexpr int NW = H / 16;            // waves per workgroup
    constexpr int NTHR = 64 * NW;
    constexpr int KQH = H / 4, S4H = KQH / 4;
    constexpr int LDH = H + PAD;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int In = a.In, T = a.T, B = a.B;
    const int Kpx = (In + 15) & ~15, KQX = Kpx >> 2, S4X = KQX >> 2, LDX = Kpx + PAD;
    float* h_lds = smem;                               // [2][16][LDH]
    float* x_lds = smem + 2 * 16 * LDH;                // [2][16][LDX]

    const int tid = threadIdx.x, lane = tid & 63, ht = tid >> 6, q = lane >> 4, j = lane & 15;
    const int tile = blockIdx.x, b0 = tile * 16, b = b0 + j;
    const bool bvalid = b < B;
    const int u0 = ht * 16 + q * 4;

    // zero both LDS regions (h_0 = 0; K padding of x stays 0 forever)
    for (int e = tid; e < 2 * 16 * (LDH + LDX); e += NTHR) smem[e] = 0.f;

    f32x4 bias4[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bias4[g] = *reinterpret_cast<const f32x4*>(a.bias + g * H + u0);

    const bool vec4 = (In & 3) == 0;
    const int n_x4 = 4 * In;              // float4 elements of one x tile (vec4 path)
    const int n_x1 = 16 * In;             // scalar elements (fallback path, e.g. D = 3)
    const uint32_t thr = drop_threshold(a.drop_p);
    const float dscale = DROP ? 1.0f / (1.0f - a.drop_p) : 1.0f;

    // staging of x_t: global -> registers (prefetch) -> LDS
    f32x4 xr4 = {0.f, 0.f, 0.f, 0.f};
    float xr1[4] = {0.f, 0.f, 0.f, 0.f};
    auto x_load = [&](int t) {
        if (vec4) {
            if (tid < n_x4) {
                const int k4 = In >> 2, jj = tid / k4, kk = (tid - jj * k4) * 4, bb = b0 + jj;
                if (bb < B) {
                    const size_t row = a.in_const ? (size_t)bb : ((size_t)bb * T + t);
                    xr4 = *reinterpret_cast<const f32x4*>(a.in + row * In + kk);
                    if (DROP) {
                        const uint32_t idx = (uint32_t)(row * In + kk);
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            xr4[e] = drop_hash(a.drop_seed, a.drop_stack, idx + e) >= thr ? xr4[e] * dscale : 0.f;
                    }
                } else {
                    xr4 = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
        } else {
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int e = tid + it * NTHR;
                float v = 0.f;
                if (e < n_x1) {
                    const int jj = e / In, kk = e - jj * In, bb = b0 + jj;
                    if (bb < B) {
                        const size_t row = a.in_const ? (size_t)bb : ((size_t)bb * T + t);
                        v = a.in[row * In + kk];
                        if (DROP) v = drop_hash(a.drop_seed, a.drop_stack, (uint32_t)(row * In + kk)) >= thr ? v * dscale : 0.f;
                    }
                }
                xr1[it] = v;
            }
        }
    };
    auto x_store = [&](int buf) {
        float* xl = x_lds + buf * 16 * LDX;
        if (vec4) {
            if (tid < n_x4) {
                const int k4 = In >> 2, jj = tid / k4, kk = (tid - jj * k4) * 4;
                *reinterpret_cast<f32x4*>(xl + jj * LDX + kk) = xr4;
            }
        } else {
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int e = tid + it * NTHR;
                if (e < n_x1) { const int jj = e / In, kk = e - jj * In; xl[jj * LDX + kk] = xr1[it]; }
            }
        }
    };

    // acc += Wpacked(ht) . lds_tile^T over one operand (S4 float4 groups per quarter)
    auto mma_operand = [&](f32x4 (&acc)[4], const float* Wp, const float* lds_row, int S4) {
        const f32x4* wp = reinterpret_cast<const f32x4*>(Wp) + (size_t)ht * S4 * 4 * 64 + lane;
        for (int s4 = 0; s4 < S4; ++s4) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(lds_row + 4 * s4);
            f32x4 w[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) w[g] = wp[(s4 * 4 + g) * 64];
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int g = 0; g < 4; ++g) acc[g] = mfma_f32_16x16x4(w[g][e], bv[e], acc[g]);
        }
    };

    __syncthreads();
    x_load(0);
    x_store(0);
    __syncthreads();

    f32x4 xproj[4];   // constant-input layer (decoder layer 0): W_ih h0 + b, computed once (temporal_vae.py:68)
    if (a.in_const) {
#pragma unroll
        for (int g = 0; g < 4; ++g) xproj[g] = bias4[g];
        mma_operand(xproj, a.Wx, x_lds + j * LDX + q * KQX, S4X);
    }

    float c[4] = {0.f, 0.f, 0.f, 0.f};
    f32x4 h4 = {0.f, 0.f, 0.f, 0.f};
    const size_t sv_stride = (size_t)NW * 64 * 4;                        // floats per saved plane (one of i,f,g,o,c)
    float* sv = a.saved ? a.saved + ((size_t)tile * T * 5) * sv_stride + ((size_t)ht * 64 + lane) * 4 : nullptr;

    for (int t = 0; t < T; ++t) {
        const int cur = t & 1;
        if (!a.in_const && t + 1 < T) x_load(t + 1);
        f32x4 acc[4];
        if (a.in_const) {
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g] = xproj[g];
        } else {
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g] = bias4[g];
            mma_operand(acc, a.Wx, x_lds + cur * 16 * LDX + j * LDX + q * KQX, S4X);
        }
        mma_operand(acc, a.Wh, h_lds + cur * 16 * LDH + j * LDH + q * KQH, S4H);

        f32x4 gi, gf, gg, go, cv;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            gi[r] = sigmoid_acc(acc[0][r]);
            gf[r] = sigmoid_acc(acc[1][r]);
            gg[r] = tanh_acc(acc[2][r]);
            go[r] = sigmoid_acc(acc[3][r]);
            c[r] = gf[r] * c[r] + gi[r] * gg[r];
            cv[r] = c[r];
            h4[r] = go[r] * tanh_acc(c[r]);
        }
        *reinterpret_cast<f32x4*>(h_lds + (cur ^ 1) * 16 * LDH + j * LDH + u0) = h4;
        if (!a.in_const && t + 1 < T) x_store(cur ^ 1);
        if (a.h_seq && bvalid) *reinterpret_cast<f32x4*>(a.h_seq + ((size_t)b * T + t) * H + u0) = h4;
        if (sv) {
            float* p = sv + (size_t)t * 5 * sv_stride;
            *reinterpret_cast<f32x4*>(p) = gi;
            *reinterpret_cast<f32x4*>(p + sv_stride) = gf;
            *reinterpret_cast<f32x4*>(p + 2 * sv_stride) = gg;
            *reinterpret_cast<f32x4*>(p + 3 * sv_stride) = go;
            *reinterpret_cast<f32x4*>(p + 4 * sv_stride) = cv;
        }
        __syncthreads();
    }
    if (a.h_last && bvalid) *reinterpret_cast<f32x4*>(a.h_last + (size_t)b * H + u0) = h4;
}

template <int H>
int launch_h(const LstmFwdArgs& a, hipStream_t s) {
    const int tiles = ceil_div(a.B, 16);
    const int Kpx = round_up(a.In, 16);
    const size_t lds = sizeof(float) * 2 * 16 * ((H + PAD) + (Kpx + PAD));
    SHM_REQUIRE(lds <= 160 * 1024, SHM_EINVAL, "lstm_fwd: %zu bytes of LDS exceed the 160 KiB of a gfx950 CU", lds);
    auto kern = (a.drop_p > 0.f) ? lstm_fwd_f32_kernel<H, true> : lstm_fwd_f32_kernel<H, false>;
    if (lds > 48 * 1024)   // opt in to large dynamic LDS (no-op where the runtime does not need it)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    SHM_LAUNCH(kern, dim3(tiles), dim3(4 * H), lds, s, a);
    SHM_LAUNCH_CHECK();
    return SHM_OK;
}

}  // namespace

int launch_lstm_fwd_f32(int H, const LstmFwdArgs& a, hipStream_t s) {
    SHM_REQUIRE(a.B > 0 && a.T > 0 && a.In > 0, SHM_EINVAL, "lstm_fwd: empty shape B=%d T=%d In=%d", a.B, a.T, a.In);
    SHM_REQUIRE(a.In <= H || a.In <= 16, SHM_EINVAL, "lstm_fwd: input width %d exceeds hidden width %d", a.In, H);
    SHM_REQUIRE((a.In & 3) == 0 || a.In <= 16, SHM_EINVAL, "lstm_fwd: input width %d must be a multiple of 4 or <= 16", a.In);
    switch (H) {
        case 32: return launch_h<32>(a, s);
        case 64: return launch_h<64>(a, s);
        case 128: return launch_h<128>(a, s);
        default: set_error("lstm_fwd: hidden_dim %d unsupported (32, 64, 128)", H); return SHM_EINVAL;
    }
}

}  // namespace shm
