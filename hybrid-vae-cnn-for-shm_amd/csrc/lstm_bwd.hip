// lstm_bwd.hip — persistent LSTM layer backward (BPTT), fp32 MFMA, gfx950.
//
// Replaces autograd's walk back through nn.LSTM for loss.backward() (reference 4DOF/Scripts/03_train_vae.py:269).
// Same ownership as the forward (lstm_fwd.hip): a workgroup owns 16 windows, wave w owns hidden units [16w,16w+16),
// lane (q,j) units 16w+4q+r of window j.  Per step, in reverse time:
//   dh   = dh_above_t + dh_rec                      dh_above: layer above (dh_seq), output layer (dy.W_out), or h_T only
//   do   = dh tanh(c);  dc += dh o (1-tanh(c)^2);  di = dc g; dg = dc i; df = dc c_{t-1};  dc <- dc f
//   dpre = (di i(1-i), df f(1-f), dg (1-g^2), do o(1-o))                    -> LDS [window][4H] and HBM (dW GEMMs)
//   dh_rec^T[H x 16]  = W_hh^T[H x 4H] dpre^T     (MFMA; result lands in the lanes that own those units)
//   dx^T   [In x 16]  = W_ih^T[In x 4H] dpre^T    (MFMA; written for the layer below, dropout mask re-applied)
// Weight gradients are deferred to gemm_tn.hip over the stored dpre sequence; sum_t dpre (bias gradient and the
// constant-input decoder layer) is accumulated in registers.
#include "kernels.hpp"

namespace shm {

namespace {

constexpr int PAD = 4;

template <int H, bool DROP>
__global__ void __launch_bounds__(4 * H) lstm_bwd_f32_kernel(LstmBwdArgs a) {
    constexpr int NW = H / 16;
    constexpr int NTHR = 64 * NW;
    constexpr int S4 = H / 4;              // float4 groups per lane quarter (quarter q <-> gate q)
    constexpr int LDP = 4 * H + PAD;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* dp_lds = smem;                  // [2][16][LDP]
    float* wo_lds = smem + 2 * 16 * LDP;   // [Dy][H]

    const int T = a.T, B = a.B, In = a.In, Dy = a.Dy;
    const int tid = threadIdx.x, lane = tid & 63, ht = tid >> 6, q = lane >> 4, j = lane & 15;
    const int tile = blockIdx.x, b0 = tile * 16, b = b0 + j;
    const bool bvalid = b < B;
    const int u0 = ht * 16 + q * 4;

    if (a.dy)
        for (int e = tid; e < Dy * H; e += NTHR) wo_lds[e] = a.Wout[e];

    const size_t sv_stride = (size_t)NW * 64 * 4;
    const float* sv = a.saved + ((size_t)tile * T * 5) * sv_stride + ((size_t)ht * 64 + lane) * 4;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    f32x4 s_i, s_f, s_g, s_o, s_c, s_cp;        // saved state of the step being processed
    auto load_saved = [&](int t) {
        const float* p = sv + (size_t)t * 5 * sv_stride;
        s_i = *reinterpret_cast<const f32x4*>(p);
        s_f = *reinterpret_cast<const f32x4*>(p + sv_stride);
        s_g = *reinterpret_cast<const f32x4*>(p + 2 * sv_stride);
        s_o = *reinterpret_cast<const f32x4*>(p + 3 * sv_stride);
        s_c = *reinterpret_cast<const f32x4*>(p + 4 * sv_stride);
        s_cp = t > 0 ? *reinterpret_cast<const f32x4*>(p - 5 * sv_stride + 4 * sv_stride) : zero4;
    };
    auto load_dh_above = [&](int t) -> f32x4 {
        f32x4 d = zero4;
        if (!bvalid) return d;
        const size_t row = (size_t)b * T + t;
        if (a.dh_seq) d = *reinterpret_cast<const f32x4*>(a.dh_seq + row * H + u0);
        if (a.dy) {
            for (int dd = 0; dd < Dy; ++dd) {
                const float dyv = a.dy[row * Dy + dd];
                const f32x4 w = *reinterpret_cast<const f32x4*>(wo_lds + dd * H + u0);
                d += dyv * w;
            }
        }
        if (a.dh_last && t == T - 1) d += *reinterpret_cast<const f32x4*>(a.dh_last + (size_t)b * H + u0);
        return d;
    };

    const uint32_t thr = drop_threshold(a.drop_p);
    const float dscale = DROP ? 1.0f / (1.0f - a.drop_p) : 1.0f;
    const int MTx = (In + 15) >> 4;

    f32x4 dc = zero4, dh_rec = zero4;
    f32x4 tsum[4] = {zero4, zero4, zero4, zero4};

    __syncthreads();                      // wo_lds visible
    load_saved(T - 1);
    f32x4 dha = load_dh_above(T - 1);

    for (int t = T - 1; t >= 0; --t) {
        const int cur = t & 1;
        float* dpl = dp_lds + cur * 16 * LDP;
        f32x4 dpre[4];
        {
            const f32x4 dh = dha + dh_rec;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float tc = tanh_acc(s_c[r]);
                const float d_o = dh[r] * tc;
                const float dcr = dc[r] + dh[r] * s_o[r] * (1.0f - tc * tc);
                const float d_i = dcr * s_g[r], d_g = dcr * s_i[r], d_f = dcr * s_cp[r];
                dc[r] = dcr * s_f[r];
                dpre[0][r] = d_i * s_i[r] * (1.0f - s_i[r]);
                dpre[1][r] = d_f * s_f[r] * (1.0f - s_f[r]);
                dpre[2][r] = d_g * (1.0f - s_g[r] * s_g[r]);
                dpre[3][r] = d_o * s_o[r] * (1.0f - s_o[r]);
            }
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            *reinterpret_cast<f32x4*>(dpl + j * LDP + g * H + u0) = dpre[g];
            tsum[g] += dpre[g];
        }
        if (bvalid && a.dpre_seq) {
            float* p = a.dpre_seq + ((size_t)b * T + t) * 4 * H + u0;
#pragma unroll
            for (int g = 0; g < 4; ++g) *reinterpret_cast<f32x4*>(p + g * H) = dpre[g];
        }
        __syncthreads();
        // prefetch the next (earlier) step's operands under the MFMA phase
        if (t > 0) { load_saved(t - 1); dha = load_dh_above(t - 1); }

        const float* brow = dpl + j * LDP + q * H;
        {
            f32x4 acc[4] = {zero4, zero4, zero4, zero4};
            const f32x4* wp = reinterpret_cast<const f32x4*>(a.WhT) + (size_t)ht * S4 * 64 + lane;
#pragma unroll 4
            for (int s4 = 0; s4 < S4; ++s4) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(brow + 4 * s4);
                const f32x4 w = wp[s4 * 64];
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] = mfma_f32_16x16x4(w[e], bv[e], acc[e]);
            }
            dh_rec = (acc[0] + acc[1]) + (acc[2] + acc[3]);
        }
        if (a.need_dx && ht < MTx) {
            f32x4 acc[4] = {zero4, zero4, zero4, zero4};
            const f32x4* wp = reinterpret_cast<const f32x4*>(a.WxT) + (size_t)ht * S4 * 64 + lane;
#pragma unroll 4
            for (int s4 = 0; s4 < S4; ++s4) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(brow + 4 * s4);
                const f32x4 w = wp[s4 * 64];
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] = mfma_f32_16x16x4(w[e], bv[e], acc[e]);
            }
            f32x4 dx = (acc[0] + acc[1]) + (acc[2] + acc[3]);
            if (bvalid) {
                const size_t base = ((size_t)b * T + t) * In + u0;
                if (DROP) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        dx[r] = drop_hash(a.drop_seed, a.drop_stack, (uint32_t)(base + r)) >= thr ? dx[r] * dscale : 0.f;
                }
                if ((In & 3) == 0) {
                    if (u0 < In) *reinterpret_cast<f32x4*>(a.dx_seq + base) = dx;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (u0 + r < In) a.dx_seq[base + r] = dx[r];
                }
            }
        }
    }
    if (bvalid && a.dpre_tsum) {
#pragma unroll
        for (int g = 0; g < 4; ++g) *reinterpret_cast<f32x4*>(a.dpre_tsum + (size_t)b * 4 * H + g * H + u0) = tsum[g];
    }
}

template <int H>
int launch_h(const LstmBwdArgs& a, hipStream_t s) {
    const int tiles = ceil_div(a.B, 16);
    const size_t lds = sizeof(float) * (2 * 16 * (4 * H + PAD) + (a.dy ? a.Dy * H : 0));
    SHM_REQUIRE(lds <= 160 * 1024, SHM_EINVAL, "lstm_bwd: %zu bytes of LDS exceed the 160 KiB of a gfx950 CU", lds);
    auto kern = (a.drop_p > 0.f && a.need_dx) ? lstm_bwd_f32_kernel<H, true> : lstm_bwd_f32_kernel<H, false>;
    if (lds > 48 * 1024)   // opt in to large dynamic LDS (no-op where the runtime does not need it)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    SHM_LAUNCH(kern, dim3(tiles), dim3(4 * H), lds, s, a);
    SHM_LAUNCH_CHECK();
    return SHM_OK;
}

}  // namespace

int launch_lstm_bwd_f32(int H, const LstmBwdArgs& a, hipStream_t s) {
    SHM_REQUIRE(a.B > 0 && a.T > 0, SHM_EINVAL, "lstm_bwd: empty shape");
    SHM_REQUIRE(a.saved != nullptr, SHM_ESTATE, "lstm_bwd: no saved forward state");
    SHM_REQUIRE(!a.need_dx || (a.WxT && a.dx_seq && a.In <= H), SHM_EINVAL, "lstm_bwd: dx requested without operands");
    SHM_REQUIRE(!a.dy || (a.Wout && a.Dy > 0 && a.Dy <= 64), SHM_EINVAL, "lstm_bwd: bad output-layer gradient");
    switch (H) {
        case 32: return launch_h<32>(a, s);
        case 64: return launch_h<64>(a, s);
        case 128: return launch_h<128>(a, s);
        default: set_error("lstm_bwd: hidden_dim %d unsupported (32, 64, 128)", H); return SHM_EINVAL;
    }
}

}  // namespace shm
