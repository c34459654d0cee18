// lstm_bwd_bf16.hip — persistent LSTM layer backward (BPTT), bf16 MFMA with fp32 accumulate; gfx950.
// Throughput twin of lstm_bwd.hip (reference: autograd through nn.LSTM, 4DOF/Scripts/03_train_vae.py:269).
//   * W_hh^T and W_ih^T fragments stay in registers for the whole sequence (64 VGPRs each at H=128);
//   * dpre_t is rounded to bf16 once and used three times: LDS image (operand of dh_rec / dx), HBM copy for the
//     deferred weight-gradient GEMM (gemm_tn_bf16.hip);
//   * gate/cell gradients, dc and dh accumulate in fp32;
//   * encoder layer 0 with <= 4 channels folds its NARROW weight gradient into the time loop (FUSE = 2:
//     dW_ih += dpre_t (x) x_t): 64 per-lane fp32 accumulators (the registers W_ih^T would have used), one 16-lane
//     shuffle reduction after the loop, per-tile partials summed in fixed order by reduce_partials.  That removes a
//     full pass over dpre_seq for a 4-column output.  (Folding dW_out into the top decoder layer the same way spills
//     at H = 128 — 256 VGPRs + scratch — so that product goes through narrow_tn in small_kernels.hip.)
#include "kernels.hpp"

namespace shm {

namespace {

template <int H, bool DROP, bool NEEDX, int FUSE>
__global__ void __launch_bounds__(4 * H) lstm_bwd_bf16_kernel(LstmBwdArgsB a) {
    constexpr int NW = H / 16, NTHR = 64 * NW;
    constexpr int KS = H / 8;                      // k-steps of 32 over K = 4H
    constexpr int RS = lds_row_stride(H);          // quarter q <-> gate q, KQ = H
    constexpr int PBUF = 4 * 16 * RS;
    extern __shared__ __attribute__((aligned(16))) bf16_t smem_b[];
    bf16_t* dp_lds = smem_b;                                               // [2][PBUF]
    float* wo_lds = reinterpret_cast<float*>(smem_b + 2 * PBUF);           // [Dy][H]

    const int T = a.T, B = a.B, In = a.In, Dy = a.Dy;
    const int tid = threadIdx.x, lane = tid & 63, ht = tid >> 6, q = lane >> 4, j = lane & 15;
    const int tile = blockIdx.x, b0 = tile * 16, b = b0 + j;
    const bool bvalid = b < B;
    const int u0 = ht * 16 + q * 4;

    if (a.dy)
        for (int e = tid; e < Dy * H; e += NTHR) wo_lds[e] = a.Wout[e];

    bf16x8 whT[KS], wxT[NEEDX ? KS : 1];
    {
        const bf16x8* p = reinterpret_cast<const bf16x8*>(a.WhT) + (size_t)ht * KS * 64 + lane;
#pragma unroll
        for (int s = 0; s < KS; ++s) whT[s] = p[s * 64];
        if constexpr (NEEDX) {
            const bf16x8* px = reinterpret_cast<const bf16x8*>(a.WxT) + (size_t)ht * KS * 64 + lane;
#pragma unroll
            for (int s = 0; s < KS; ++s) wxT[s] = px[s * 64];
        }
    }
    // fused narrow products: per-lane accumulators
    f32x4 fw[FUSE == 2 ? 16 : 1];                        // [g*4 + r] over d
#pragma unroll
    for (int i = 0; i < (FUSE == 2 ? 16 : 1); ++i) fw[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 aux = {0.f, 0.f, 0.f, 0.f};                    // x_t[0..3] of the step being processed

    const bf16_t* svg = a.sv_g + (((size_t)tile * T * NW + ht) * 64 + lane) * 16;
    const float* svc = a.sv_c + (((size_t)tile * T * NW + ht) * 64 + lane) * 4;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    f32x4 s_i, s_f, s_g, s_o, s_c, s_cp;
    auto load_saved = [&](int t) {
        const bf16_t* p = svg + (size_t)t * NW * 64 * 16;
        const bf16x8 lo = *reinterpret_cast<const bf16x8*>(p), hi = *reinterpret_cast<const bf16x8*>(p + 8);
#pragma unroll
        for (int r = 0; r < 4; ++r) { s_i[r] = (float)lo[r]; s_f[r] = (float)lo[4 + r]; s_g[r] = (float)hi[r]; s_o[r] = (float)hi[4 + r]; }
        s_c = s_cp;                                       // c_t was fetched as "c_{t-1}" one iteration ago
        s_cp = t > 0 ? *reinterpret_cast<const f32x4*>(svc + (size_t)(t - 1) * NW * 64 * 4) : zero4;
    };
    auto load_dh_above = [&](int t) -> f32x4 {
        f32x4 d = zero4;
        if constexpr (FUSE != 0) aux = zero4;
        if (!bvalid) return d;
        const size_t row = (size_t)b * T + t;
        if (a.dh_seq) d = *reinterpret_cast<const f32x4*>(a.dh_seq + row * H + u0);
        if (a.dy) {
            for (int dd = 0; dd < Dy; ++dd) {
                const float dyv = a.dy[row * Dy + dd];
                d += dyv * *reinterpret_cast<const f32x4*>(wo_lds + dd * H + u0);
            }
        }
        if constexpr (FUSE == 2) {
            for (int dd = 0; dd < a.Dx; ++dd) aux[dd & 3] = a.x[row * a.Dx + dd];
        }
        if (a.dh_last && t == T - 1) d += *reinterpret_cast<const f32x4*>(a.dh_last + (size_t)b * H + u0);
        return d;
    };

    const uint32_t thr = drop_threshold(a.drop_p);
    const float dscale = DROP ? 1.0f / (1.0f - a.drop_p) : 1.0f;
    const int MTx = (In + 15) >> 4;

    f32x4 dc = zero4, dh_rec = zero4;
    f32x4 tsum[4] = {zero4, zero4, zero4, zero4};

    __syncthreads();
    s_cp = *reinterpret_cast<const f32x4*>(svc + (size_t)(T - 1) * NW * 64 * 4);
    load_saved(T - 1);
    f32x4 dha = load_dh_above(T - 1);

    for (int t = T - 1; t >= 0; --t) {
        bf16_t* dpl = dp_lds + (t & 1) * PBUF;
        f32x4 dpre[4];
        {
            const f32x4 dh = dha + dh_rec;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float tc = tanh_fast(s_c[r]);
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
            const bf16x4 pb = cvt_bf16x4(dpre[g]);
            *reinterpret_cast<bf16x4*>(dpl + (g * 16 + j) * RS + u0) = pb;
            tsum[g] += dpre[g];
            if constexpr (FUSE == 2) {
#pragma unroll
                for (int r = 0; r < 4; ++r) fw[g * 4 + r] += dpre[g][r] * aux;      // before the prefetch overwrites aux
            }
            if (bvalid && a.dpre_seq) *reinterpret_cast<bf16x4*>(a.dpre_seq + ((size_t)b * T + t) * 4 * H + g * H + u0) = pb;
        }
        __syncthreads();
        if (t > 0) { load_saved(t - 1); dha = load_dh_above(t - 1); }

        const bf16_t* brow = dpl + (q * 16 + j) * RS;
        f32x4 acc_h = zero4, acc_x = zero4;
        if constexpr (NEEDX) {
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const bf16x8 bv = *reinterpret_cast<const bf16x8*>(brow + 8 * s);
                acc_h = mfma_bf16_16x16x32(whT[s], bv, acc_h);
                acc_x = mfma_bf16_16x16x32(wxT[s], bv, acc_x);
            }
        } else {
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const bf16x8 bv = *reinterpret_cast<const bf16x8*>(brow + 8 * s);
                acc_h = mfma_bf16_16x16x32(whT[s], bv, acc_h);
            }
        }
        dh_rec = acc_h;
        if (NEEDX && ht < MTx && bvalid) {
            f32x4 dx = acc_x;
            const size_t base = ((size_t)b * T + t) * In + u0;
            if (DROP) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    dx[r] = drop_hash(a.drop_seed, a.drop_stack, (uint32_t)(base + r)) >= thr ? dx[r] * dscale : 0.f;
            }
            if (u0 < In) *reinterpret_cast<f32x4*>(a.dx_seq + base) = dx;
        }
    }
    if (bvalid && a.dpre_tsum) {
#pragma unroll
        for (int g = 0; g < 4; ++g) *reinterpret_cast<f32x4*>(a.dpre_tsum + (size_t)b * 4 * H + g * H + u0) = tsum[g];
    }
    // fused narrow products: sum over the 16 windows of the tile (lanes of one quarter), lane j == 0 writes the partial
    if constexpr (FUSE != 0) {
        constexpr int NF = 16;
#pragma unroll
        for (int i = 0; i < NF; ++i)
#pragma unroll
            for (int off = 1; off < 16; off <<= 1)
#pragma unroll
                for (int e = 0; e < 4; ++e) fw[i][e] += __shfl_xor(fw[i][e], off, 64);
        {
            if (j == 0) {
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        for (int dd = 0; dd < a.Dx; ++dd)
                            a.fuse_w[((size_t)tile * 4 * H + g * H + u0 + r) * a.Dx + dd] = fw[g * 4 + r][dd];   // [tile][4H][D]
            }
        }
    }
}

template <int H>
int launch_h(const LstmBwdArgsB& a, hipStream_t s) {
    const int tiles = ceil_div(a.B, 16);
    const size_t lds = sizeof(bf16_t) * 2 * 4 * 16 * lds_row_stride(H) + sizeof(float) * (a.dy ? a.Dy * H : 0);
    void (*kern)(LstmBwdArgsB) = nullptr;
    const bool drop = a.drop_p > 0.f && a.need_dx;
    if (a.need_dx) kern = drop ? lstm_bwd_bf16_kernel<H, true, true, 0> : lstm_bwd_bf16_kernel<H, false, true, 0>;
    else kern = a.fuse == 2 ? lstm_bwd_bf16_kernel<H, false, false, 2> : lstm_bwd_bf16_kernel<H, false, false, 0>;
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(4 * H), lds, s, a);
    SHM_LAUNCH_CHECK();
    return SHM_OK;
}

}  // namespace

int launch_lstm_bwd_bf16(int H, const LstmBwdArgsB& a, hipStream_t s) {
    SHM_REQUIRE(a.B > 0 && a.T > 0, SHM_EINVAL, "lstm_bwd_bf16: empty shape");
    SHM_REQUIRE(a.sv_g && a.sv_c, SHM_ESTATE, "lstm_bwd_bf16: no saved forward state");
    SHM_REQUIRE(!a.need_dx || (a.WxT && a.dx_seq && a.In == H), SHM_EINVAL, "lstm_bwd_bf16: dx needs In == H and operands");
    SHM_REQUIRE(!a.dy || (a.Wout && a.Dy > 0 && a.Dy <= 64), SHM_EINVAL, "lstm_bwd_bf16: bad output-layer gradient");
    SHM_REQUIRE(a.fuse == 0 || (a.fuse == 2 && a.x && a.Dx >= 1 && a.Dx <= 4 && a.fuse_w && !a.need_dx), SHM_EINVAL,
                "lstm_bwd_bf16: fused narrow product needs <= 4 channels, no dx and its buffer");
    switch (H) {
        case 32: return launch_h<32>(a, s);
        case 64: return launch_h<64>(a, s);
        case 128: return launch_h<128>(a, s);
        default: set_error("lstm_bwd_bf16: hidden_dim %d unsupported (32, 64, 128)", H); return SHM_EINVAL;
    }
}

}  // namespace shm
