// lstm_fwd_bf16.hip — persistent LSTM layer forward, bf16 MFMA (v_mfma_f32_16x16x32_bf16), fp32 accumulate and
// fp32 cell state; gfx950.  Throughput twin of lstm_fwd.hip (same ownership: workgroup = 16 windows, wave w = hidden
// units [16w,16w+16) x gates i,f,g,o, lane (q,j) = units 16w+4q+r of window j), replacing the nn.LSTM cell loop of
// reference 4DOF/Scripts/Models/temporal_vae.py:53,69.
//
// What changes against the fp32 kernel:
//   * W_hh and W_ih live in REGISTERS for the whole sequence (64 VGPRs per matrix per wave at H=128) — the time loop
//     touches no weight memory at all;
//   * h_t returns to LDS as bf16 in a quarter-major, conflict-free image (common.hpp: lds_row_stride);
//   * sigmoid/tanh use v_exp_f32 / v_rcp_f32;
//   * saved state for BPTT: gates as bf16 (32 B/lane/step), c as fp32 (16 B/lane/step), both in fragment order;
//   * h_seq is tile-blocked row-major (row r' = (tile*T + t)*16 + j, see lstm_bwd_bf16.hip): h_t is copied out of the
//     completed LDS image one step later as 16-byte fully coalesced stores, and the next layer stages x_t = h^{l-1}_t with
//     16-byte coalesced loads of the same contiguous 16 x H block.
#include "kernels.hpp"

namespace shm {

namespace {

// XK: k-steps (of 32) of the input operand.  XF32: dynamic input is fp32 [B,T,In] with In <= 32 (encoder layer 0).
// CONSTIN: the input is one fp32 vector per window (decoder layer 0).  SAVE: keep gates / cell state for BPTT.
// The time loop is branch-free (clamped loads, unconditional stores into tile-padded buffers, T % 3 steps peeled) so that
// the compiler emits ONE counted s_waitcnt vmcnt per step instead of draining the queue (see lstm_bwd_bf16.hip).
template <int H, int XK, bool XF32, bool DROP, bool CONSTIN, bool SAVE>
__global__ void __launch_bounds__(4 * H) lstm_fwd_bf16_kernel(LstmFwdArgsB a) {
    constexpr int NW = H / 16, NTHR = 64 * NW;
    constexpr int KSH = H / 32;                 // k-steps of the recurrent operand
    constexpr int KQH = H / 4, RSH = lds_row_stride(KQH);
    constexpr int KQX = XK * 8, RSX = lds_row_stride(KQX);
    constexpr int HBUF = 4 * 16 * RSH, XBUF = 4 * 16 * RSX;       // elements per buffer
    extern __shared__ __attribute__((aligned(16))) bf16_t smem_b[];
    bf16_t* h_lds = smem_b;                     // [2][HBUF]
    bf16_t* x_lds = smem_b + 2 * HBUF;          // [2][XBUF]

    const int In = a.In, T = a.T, B = a.B;
    const int tid = threadIdx.x, lane = tid & 63, ht = tid >> 6, q = lane >> 4, j = lane & 15;
    const int tile = blockIdx.x, b0 = tile * 16, b = b0 + j;
    const bool bvalid = b < B;
    const int u0 = ht * 16 + q * 4;

    for (int e = tid; e < 2 * (HBUF + XBUF); e += NTHR) smem_b[e] = (bf16_t)0.f;

    // ---- weights -> registers (fragment order, see kernels.hpp)
    bf16x8 wh[4][KSH], wx[4][XK];
    {
        const bf16x8* p = reinterpret_cast<const bf16x8*>(a.Wh) + (size_t)ht * KSH * 4 * 64 + lane;
#pragma unroll
        for (int s = 0; s < KSH; ++s)
#pragma unroll
            for (int g = 0; g < 4; ++g) wh[g][s] = p[(s * 4 + g) * 64];
        const bf16x8* px = reinterpret_cast<const bf16x8*>(a.Wx) + (size_t)ht * XK * 4 * 64 + lane;
#pragma unroll
        for (int s = 0; s < XK; ++s)
#pragma unroll
            for (int g = 0; g < 4; ++g) wx[g][s] = px[(s * 4 + g) * 64];
    }
    f32x4 bias4[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bias4[g] = *reinterpret_cast<const f32x4*>(a.bias + g * H + u0);

    const uint32_t thr = drop_threshold(a.drop_p);
    const float dscale = DROP ? 1.0f / (1.0f - a.drop_p) : 1.0f;

    // my slot in the operand images: (window j, units u0..u0+3)
    const int my_h_off = ((u0 / KQH) * 16 + j) * RSH + (u0 % KQH);
    const int my_x_off = ((u0 / KQX) * 16 + j) * RSX + (u0 % KQX);      // only meaningful when In == H

    // ---- x staging: x_{t+1..t+3} travel in three rotating register slots (the loop below is unrolled x3 so the slot index
    // is static); a slot is committed to LDS one step before it is consumed and refilled 4 steps ahead, so the global-load
    // latency (and the in-order vmcnt queue behind this step's stores) is covered by three full steps.
    struct XSlot { bf16x8 v; float s0, s1; };
    // bf16 input (layer >= 1): the 16 x H block of step t is contiguous; thread c < 2H moves the 16-byte chunk (row jj, cols c8..c8+7)
    const int xc = tid < 2 * H ? tid : 0, xjj = xc / (H / 8), xc8 = (xc - xjj * (H / 8)) * 8;
    const int x_img_off = ((xc8 / KQX) * 16 + xjj) * RSX + (xc8 % KQX);
    // XF32 staging: element e of the 16 x In tile -> (window jj, channel kk); two elements per thread cover In <= 16 at H = 32
    const int xe0 = tid < 16 * In ? tid : 0, xe1 = tid + NTHR < 16 * In ? tid + NTHR : 0;
    const int xj0 = xe0 / In, xk0 = xe0 - xj0 * In, xj1 = xe1 / In, xk1 = xe1 - xj1 * In;
    const int xb0 = b0 + xj0 < B ? b0 + xj0 : B - 1, xb1 = b0 + xj1 < B ? b0 + xj1 : B - 1;
    auto x_fetch = [&](XSlot& X, int t) {
        if constexpr (CONSTIN) return;
        const int tc = t < T ? t : T - 1;                  // clamp: steps >= T are fetched but never committed
        if constexpr (XF32) {
            const float* xin = reinterpret_cast<const float*>(a.in);
            X.s0 = xin[((size_t)xb0 * T + tc) * In + xk0];
            X.s1 = xin[((size_t)xb1 * T + tc) * In + xk1];
        } else {
            const bf16_t* xin = reinterpret_cast<const bf16_t*>(a.in);
            X.v = *reinterpret_cast<const bf16x8*>(xin + ((size_t)tile * T + tc) * 16 * H + (size_t)xc * 8);
        }
    };
    auto x_commit = [&](XSlot& X, int t, int buf) {      // t: the step the slot holds
        bf16_t* xl = x_lds + buf * XBUF;
        if constexpr (XF32) {
            asm volatile("" : "+v"(X.s0), "+v"(X.s1));     // pin the use (see lstm_bwd_bf16.hip)
            if (tid < 16 * In) xl[((xk0 / KQX) * 16 + xj0) * RSX + (xk0 % KQX)] = (bf16_t)X.s0;
            if (tid + NTHR < 16 * In) xl[((xk1 / KQX) * 16 + xj1) * RSX + (xk1 % KQX)] = (bf16_t)X.s1;
        } else {
            asm volatile("" : "+v"(X.v));
            bf16x8 v = X.v;
            if (DROP) {     // element index = (tile-blocked row) * H + unit: identical in dx_gemm and the dW_ih GEMM staging
                const uint32_t idx = (uint32_t)((((size_t)tile * T + t) * 16 + xjj) * H + xc8);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = drop_hash(a.drop_seed, a.drop_stack, idx + e) >= thr ? (bf16_t)((float)v[e] * dscale) : (bf16_t)0.f;
            }
            if (tid < 2 * H) *reinterpret_cast<bf16x8*>(xl + x_img_off) = v;
        }
    };

    auto mma_h = [&](f32x4 (&acc)[4], const bf16_t* img) {
        const bf16_t* row = img + (q * 16 + j) * RSH;
#pragma unroll
        for (int s = 0; s < KSH; ++s) {
            const bf16x8 bv = *reinterpret_cast<const bf16x8*>(row + 8 * s);
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g] = mfma_bf16_16x16x32(wh[g][s], bv, acc[g]);
        }
    };
    auto mma_x = [&](f32x4 (&acc)[4], const bf16_t* img) {
        const bf16_t* row = img + (q * 16 + j) * RSX;
#pragma unroll
        for (int s = 0; s < XK; ++s) {
            const bf16x8 bv = *reinterpret_cast<const bf16x8*>(row + 8 * s);
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g] = mfma_bf16_16x16x32(wx[g][s], bv, acc[g]);
        }
    };

    __syncthreads();
    f32x4 xproj[4];
    if constexpr (CONSTIN) {
        // decoder layer 0: the same h0 at every step (temporal_vae.py:68) -> project once, keep in registers
        const float* h0 = reinterpret_cast<const float*>(a.in);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (bvalid) v = *reinterpret_cast<const f32x4*>(h0 + (size_t)b * H + u0);
        *reinterpret_cast<bf16x4*>(x_lds + my_x_off) = cvt_bf16x4(v);
        __syncthreads();
#pragma unroll
        for (int g = 0; g < 4; ++g) xproj[g] = bias4[g];
        mma_x(xproj, x_lds);
    } else {
        XSlot X;
        x_fetch(X, 0);
        x_commit(X, 0, 0);
        __syncthreads();
    }

    f32x4 c = {0.f, 0.f, 0.f, 0.f}, h4 = {0.f, 0.f, 0.f, 0.f};
    bf16_t* svg = SAVE ? a.sv_g + (((size_t)tile * T * NW + ht) * 64 + lane) * 16 : nullptr;
    float* svc = SAVE ? a.sv_c + (((size_t)tile * T * NW + ht) * 64 + lane) * 4 : nullptr;

    // X: slot holding x_{t+1}; committed at the end of the step and refilled with x_{t+4}
    // h_{t} image -> h_seq block (tile, t): 16-byte chunks, fully coalesced; called once the image is complete (after the barrier)
    const int hq = xc8 / KQH, h_img_off = (hq * 16 + xjj) * RSH + (xc8 % KQH);
    auto h_flush = [&](int t_done, int buf) {
        if (tid < 2 * H)
            *reinterpret_cast<bf16x8*>(a.h_seq + ((size_t)tile * T + t_done) * 16 * H + (size_t)xc * 8) =
                *reinterpret_cast<const bf16x8*>(h_lds + buf * HBUF + h_img_off);
    };
    auto step = [&](XSlot& X, int t) {
        const int cur = t & 1;
        if (t > 0) h_flush(t - 1, cur);                    // h_{t-1} sits complete in h_lds[cur]
        f32x4 acc[4];
        if constexpr (CONSTIN) {
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g] = xproj[g];
        } else {
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g] = bias4[g];
            mma_x(acc, x_lds + cur * XBUF);
        }
        mma_h(acc, h_lds + cur * HBUF);

        f32x4 gi, gf, gg, go;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            gi[r] = sigmoid_fast(acc[0][r]);
            gf[r] = sigmoid_fast(acc[1][r]);
            gg[r] = tanh_fast(acc[2][r]);
            go[r] = sigmoid_fast(acc[3][r]);
            c[r] = gf[r] * c[r] + gi[r] * gg[r];
            h4[r] = go[r] * tanh_fast(c[r]);
        }
        const bf16x4 hb = cvt_bf16x4(h4);
        *reinterpret_cast<bf16x4*>(h_lds + (cur ^ 1) * HBUF + my_h_off) = hb;
        if constexpr (!CONSTIN) {
            if (t + 1 < T) x_commit(X, t + 1, cur ^ 1);
            x_fetch(X, t + 4);
        }
        if constexpr (SAVE) {
            bf16_t* p = svg + (size_t)t * NW * 64 * 16;
            bf16x8 lo, hi;
            const bf16x4 bi = cvt_bf16x4(gi), bf = cvt_bf16x4(gf), bg = cvt_bf16x4(gg), bo = cvt_bf16x4(go);
#pragma unroll
            for (int r = 0; r < 4; ++r) { lo[r] = bi[r]; lo[4 + r] = bf[r]; hi[r] = bg[r]; hi[4 + r] = bo[r]; }
            *reinterpret_cast<bf16x8*>(p) = lo;
            *reinterpret_cast<bf16x8*>(p + 8) = hi;
            *reinterpret_cast<f32x4*>(svc + (size_t)t * NW * 64 * 4) = c;
        }
        __syncthreads();
    };
    XSlot X0, X1, X2;
    int t = 0;
    for (int k = T % 3; k > 0; --k, ++t) { x_fetch(X0, t + 1); step(X0, t); }   // peel: the main loop runs whole triples
    x_fetch(X0, t + 1); x_fetch(X1, t + 2); x_fetch(X2, t + 3);
    __builtin_amdgcn_s_waitcnt(0x0F70);                    // vmcnt(0) once: only steady-state loads cross the back-edge
    for (; t + 2 < T; t += 3) {
        step(X0, t);
        step(X1, t + 1);
        step(X2, t + 2);
    }
    h_flush(T - 1, T & 1);                                 // the last image (written into buffer (T-1)&1 ^ 1 = T&1, barrier passed)
    if (a.h_last && bvalid) *reinterpret_cast<f32x4*>(a.h_last + (size_t)b * H + u0) = h4;
}

template <int H, int XK, bool XF32, bool CONSTIN>
int launch_x(const LstmFwdArgsB& a, hipStream_t s) {
    const int tiles = ceil_div(a.B, 16);
    const size_t lds = sizeof(bf16_t) * 2 * 4 * 16 * (lds_row_stride(H / 4) + lds_row_stride(XK * 8));
    const bool drop = a.drop_p > 0.f && !XF32 && !CONSTIN, save = a.sv_g != nullptr;
    void (*kern)(LstmFwdArgsB);
    if constexpr (XF32 || CONSTIN)
        kern = save ? lstm_fwd_bf16_kernel<H, XK, XF32, false, CONSTIN, true> : lstm_fwd_bf16_kernel<H, XK, XF32, false, CONSTIN, false>;
    else
        kern = drop ? (save ? lstm_fwd_bf16_kernel<H, XK, false, true, false, true> : lstm_fwd_bf16_kernel<H, XK, false, true, false, false>)
                    : (save ? lstm_fwd_bf16_kernel<H, XK, false, false, false, true> : lstm_fwd_bf16_kernel<H, XK, false, false, false, false>);
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(4 * H), lds, s, a);
    SHM_LAUNCH_CHECK();
    return SHM_OK;
}

template <int H>
int launch_h(const LstmFwdArgsB& a, hipStream_t s) {
    SHM_REQUIRE(a.h_seq != nullptr && (a.sv_g == nullptr) == (a.sv_c == nullptr), SHM_EINVAL, "lstm_fwd_bf16: h_seq / saved buffers");
    if (a.in_f32 && !a.in_const) {
        SHM_REQUIRE(a.In <= 32 && 16 * a.In <= 8 * H, SHM_EINVAL, "lstm_fwd_bf16: fp32 input width %d too wide for H=%d", a.In, H);
        return launch_x<H, 1, true, false>(a, s);
    }
    SHM_REQUIRE(a.In == H, SHM_EINVAL, "lstm_fwd_bf16: layer input width %d must equal hidden width %d", a.In, H);
    if (a.in_const) return launch_x<H, H / 32, false, true>(a, s);
    return launch_x<H, H / 32, false, false>(a, s);
}

}  // namespace

int launch_lstm_fwd_bf16(int H, const LstmFwdArgsB& a, hipStream_t s) {
    SHM_REQUIRE(a.B > 0 && a.T > 0 && a.In > 0, SHM_EINVAL, "lstm_fwd_bf16: empty shape");
    switch (H) {
        case 32: return launch_h<32>(a, s);
        case 64: return launch_h<64>(a, s);
        case 128: return launch_h<128>(a, s);
        default: set_error("lstm_fwd_bf16: hidden_dim %d unsupported (32, 64, 128)", H); return SHM_EINVAL;
    }
}

}  // namespace shm
