// lstm_sp.hip — split-precision ("bf16x2") LSTM kernels for gfx950: the parity-grade fast path (SHM_PREC_SPLIT).
// Reference arithmetic: the fp32 nn.LSTM cell loop of 4DOF/Scripts/Models/temporal_vae.py:53,69 and its autograd BPTT
// (4DOF/Scripts/03_train_vae.py:269).  north_star asks for 1e-4 agreement with that fp32 result; bf16 operands miss it
// (weights rounded to bf16 alone move the reconstruction by 4.5e-4) and the exact-fp32 MFMA
// (v_mfma_f32_16x16x4_f32) runs at 1/16 of the bf16 rate.  Here every MFMA operand is a PAIR of bf16 values
//     x ~ hi + lo,   hi = bf16(x),   lo = bf16(x - hi)        (16 significant bits, |x - hi - lo| <= 2^-16 |x|)
// and a product is the sum of the bf16 products hi*hi + hi*lo + lo*hi (+ lo*lo), accumulated in fp32 by the MFMA itself.
// Measured on the north-star golden (emulation, then the kernels): reconstruction within 1e-6 of the fp32 reference.
//
// Decomposition: quad tiles as in lstm_fwd_bf16.hip (one workgroup = 4 windows x all T steps, wave w owns hidden units
// [16w, 16w+16) x 4 gates, one cell per lane after the DPP re-deal).  What is different:
//   * the 16-column MFMA carries the 4 windows TWICE: columns 0..3 = hi(h), columns 4..7 = lo(h) (rows 0..3 / 4..7 of the LDS
//     operand image).  One MFMA with A = hi(W) then gives hi*hi and hi*lo, a second with A = lo(W) adds lo*hi and lo*lo into
//     the same accumulator: all four cross terms for 2 MFMAs per fragment, and the re-deal adds column j+4 to column j;
//   * W_hh hi AND lo fragments live in registers (128 VGPRs at H = 128), which leaves no room for W_ih: the input
//     projection W_ih x_t + b of layers whose input is a hidden sequence is hoisted into a time-parallel kernel
//     (xproj_sp_kernel: full 16-column MFMA tiles, HBM-bound) and arrives as an fp32 stream in CELL ORDER (one f32x4
//     {i,f,g,o} per lane and step); the constant-input decoder layer 0 reads its projection once; encoder layer 0
//     (In <= 16: one 32-deep k-step) keeps its 8 input MFMAs in the loop;
//   * BPTT records are fp32: {i,f,g,o} (16 B) + {c_{t-1}, tanh c_t} (8 B) per cell and step.  Tried and not kept (round 2): 12-byte
//     records with the gates as 16-bit fixed point and tanh(c_t) recomputed in the backward — the recurrences are latency-bound
//     (~0.85 us per step), so halving their largest stream bought nothing (backward launch 0.218 -> 0.230 ms with the unpacking
//     on the dependency chain, step 2.94 vs 2.93 ms) while the worst gradient tensor moved from 7e-6 to 2.7e-5 of its max;
//   * backward: dpre_t leaves as hi / lo planes, dh_rec = W_hh^T dpre_t with the same column trick; dx = dpre W_ih of the
//     layer below is the time-parallel dx_sp_kernel, weight gradients are gemm_tn_bf16.hip run over three plane pairs.
#include "kernels.hpp"

namespace shm {

namespace {

constexpr int WPT = SHM_WPT;
static_assert(WPT == 4, "hi/lo columns 0..3 / 4..7 and the 16-row = 4 steps x 4 windows tiles assume quad tiles");

__device__ __forceinline__ bf16x2 split2(float v) {
    const bf16_t hi = (bf16_t)v;
    const bf16_t lo = (bf16_t)(v - (float)hi);
    return bf16x2{hi, lo};
}

// accumulator lanes (q, j, r) -> cell lanes (q, jj): the cell of lane jj = (window jj & 3, register jj >> 2) is
// acc[jj >> 2] of lane jj & 3 (hi columns) + acc[jj >> 2] of lane 4 + (jj & 3) (lo columns); 6 DPP moves + 1 add.
__device__ __forceinline__ float redeal_hl(const f32x4 a) {
    int v = __float_as_int(a[0]);
    v = __builtin_amdgcn_update_dpp(v, __float_as_int(a[1]), 0x114, 0xF, 0x2, false);   // row_shr:4  -> lanes 4..7   (from 0..3)
    v = __builtin_amdgcn_update_dpp(v, __float_as_int(a[2]), 0x118, 0xF, 0x4, false);   // row_shr:8  -> lanes 8..11  (from 0..3)
    v = __builtin_amdgcn_update_dpp(v, __float_as_int(a[3]), 0x11C, 0xF, 0x8, false);   // row_shr:12 -> lanes 12..15 (from 0..3)
    int w = __float_as_int(a[1]);                                                        // lanes 4..7 keep their own register 1
    w = __builtin_amdgcn_update_dpp(w, __float_as_int(a[0]), 0x104, 0xF, 0x1, false);   // row_shl:4  -> lanes 0..3   (from 4..7)
    w = __builtin_amdgcn_update_dpp(w, __float_as_int(a[2]), 0x114, 0xF, 0x4, false);   // row_shr:4  -> lanes 8..11  (from 4..7)
    w = __builtin_amdgcn_update_dpp(w, __float_as_int(a[3]), 0x118, 0xF, 0x8, false);   // row_shr:8  -> lanes 12..15 (from 4..7)
    return __int_as_float(v) + __int_as_float(w);
}

// ---------------------------------------------------------------------------------------------------------------------
// forward recurrence.  MODE 0: xproj stream, 1: xproj per window, 2: fp32 x (In <= 16) projected in the loop.
template <int H, int MODE, bool SAVE>
__global__ void __launch_bounds__(4 * H) lstm_fwd_sp_kernel(LstmFwdArgsS a) {
    constexpr int NW = H / 16, NTHR = 64 * NW;
    constexpr int KSH = H / 32, KQH = H / 4, RSH = lds_row_stride(KQH);
    constexpr int KQX = 8, RSX = lds_row_stride(KQX);              // input operand of mode 2: one k-step of 32
    constexpr int HBUF = 4 * 16 * RSH, XBUF = 4 * 16 * RSX;
    extern __shared__ __attribute__((aligned(16))) bf16_t smem_b[];
    bf16_t* h_lds = smem_b;                     // [2][HBUF]   rows 0..3 hi, 4..7 lo, 8..15 zero
    bf16_t* x_lds = smem_b + 2 * HBUF;          // [2][XBUF]   (mode 2)

    const int In = a.In > 0 ? a.In : 1, T = a.T, B = a.B, tile = blockIdx.x;
    const int t0 = a.t0, t1 = a.t1;                                 // this launch walks steps [t0, t1) (a time chunk, or the whole window)
    const int tid = threadIdx.x, lane = tid & 63, ht = tid >> 6, q = lane >> 4, j = lane & 15;
    const int cj = j & 3, cr = j >> 2, cu = ht * 16 + q * 4 + cr;  // my cell: window cj of the tile, unit cu
    const int cb = tile * WPT + cj;
    const bool cvalid = cb < B;

    for (int e = tid; e < 2 * HBUF + (MODE == 2 ? 2 * XBUF : 0); e += NTHR) smem_b[e] = (bf16_t)0.f;

    bf16x8 whi[4][KSH], wlo[4][KSH];
    {
        const size_t off = (size_t)ht * KSH * 4 * 64 + lane;
        const bf16x8* ph = reinterpret_cast<const bf16x8*>(a.Wh_hi) + off;
        const bf16x8* pl = reinterpret_cast<const bf16x8*>(a.Wh_lo) + off;
#pragma unroll
        for (int s = 0; s < KSH; ++s)
#pragma unroll
            for (int g = 0; g < 4; ++g) { whi[g][s] = ph[(s * 4 + g) * 64]; wlo[g][s] = pl[(s * 4 + g) * 64]; }
    }
    bf16x8 wxh[4], wxl[4];
    float bias_c[4] = {0.f, 0.f, 0.f, 0.f};
    if constexpr (MODE == 2) {
        const size_t off = (size_t)ht * 4 * 64 + lane;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            wxh[g] = reinterpret_cast<const bf16x8*>(a.Wx_hi)[off + g * 64];
            wxl[g] = reinterpret_cast<const bf16x8*>(a.Wx_lo)[off + g * 64];
            bias_c[g] = a.bias[g * H + cu];
        }
    }

    const int my_h_off = ((cu / KQH) * 16 + cj) * RSH + (cu % KQH);

    // h_{t} planes: the WPT x H block of a step is contiguous; thread c < NCH moves a 16-byte chunk of hi, NCH..2NCH-1 of lo
    constexpr int NCH = WPT * H / 8;
    static_assert(2 * NCH <= NTHR, "one chunk per thread");
    const int fpl = tid >= NCH ? 1 : 0, fc = tid < 2 * NCH ? tid - fpl * NCH : 0;
    const int fjj = fc / (H / 8), fc8 = (fc - fjj * (H / 8)) * 8;
    const int f_img_off = ((fc8 / KQH) * 16 + fjj + 4 * fpl) * RSH + (fc8 % KQH);
    bf16_t* const f_dst = (fpl ? a.h_lo : a.h_hi) + (size_t)tile * T * WPT * H + (size_t)fc * 8;
    auto h_flush = [&](int t_done, int buf) {
        if (tid < 2 * NCH)
            *reinterpret_cast<bf16x8*>(f_dst + (size_t)t_done * WPT * H) = *reinterpret_cast<const bf16x8*>(h_lds + buf * HBUF + f_img_off);
    };

    // per-step input of this lane: mode 0 the f32x4 projection of its cell; mode 2 one element of x_{t+1} (thread e < WPT*In)
    const int xe = tid < WPT * In ? tid : 0, xj0 = xe / In, xk0 = xe - xj0 * In;
    const int xb0 = tile * WPT + xj0 < B ? tile * WPT + xj0 : B - 1;
    const f32x4* xp = reinterpret_cast<const f32x4*>(a.xproj) + ((size_t)tile * (MODE == 1 ? 1 : T) * NW + ht) * 64 + lane;
    struct Slot { f32x4 v; };
    // input slots = prefetch distance in steps: 6 for the projection stream (16 bytes per lane and step out of HBM, ~5 us ahead), 3 for raw x
    constexpr int NS = MODE == 0 ? 8 : 3;
    auto fetch = [&](Slot& X, int t) {           // mode 0: projection of step t; mode 2: x of step t; clamped, never predicated
        const int tc = t < t1 ? t : t1 - 1;        // never reads another chunk's rows (they may be being written right now)
        if constexpr (MODE == 0) X.v = xp[(size_t)tc * NW * 64];
        else if constexpr (MODE == 2) X.v[0] = a.x[((size_t)xb0 * T + tc) * In + xk0];
    };
    auto x_commit = [&](float v, int t, int buf) {   // mode 2: x_t -> operand image (hi row / lo row) and the tile-blocked copies
        if (tid < WPT * In) {
            const bf16x2 s = split2(v);
            bf16_t* xl = x_lds + buf * XBUF + ((xk0 / KQX) * 16 + xj0) * RSX + (xk0 % KQX);
            xl[0] = s[0]; xl[4 * RSX] = s[1];
            if (a.x_hi) {
                const size_t o = (((size_t)tile * T + t) * WPT + xj0) * 16 + xk0;
                a.x_hi[o] = s[0]; a.x_lo[o] = s[1];
            }
        }
    };

    f32x4 xconst = {0.f, 0.f, 0.f, 0.f};
    if constexpr (MODE == 1) xconst = xp[0];
    if constexpr (MODE == 2) {
        __syncthreads();                          // images zeroed
        x_commit(a.x[((size_t)xb0 * T + t0) * In + xk0], t0, t0 & 1);
    }
    float c = 0.f, h = 0.f;
    float* const cst = a.c_state ? a.c_state + ((size_t)tile * NW + ht) * 64 + lane : nullptr;
    if (t0 > 0) {                                 // continue a previous chunk: c_{t0-1} from the state buffer, h_{t0-1} planes -> operand image
        if (MODE != 2) __syncthreads();           // images zeroed
        c = cst[0];
        if (tid < 2 * NCH)
            *reinterpret_cast<bf16x8*>(h_lds + (t0 & 1) * HBUF + f_img_off) = *reinterpret_cast<const bf16x8*>(f_dst + (size_t)(t0 - 1) * WPT * H);
    }
    __syncthreads();

    f32x4* sv4 = SAVE ? reinterpret_cast<f32x4*>(a.rec4) + ((size_t)tile * T * NW + ht) * 64 + lane : nullptr;
    f32x2* sv2 = SAVE ? reinterpret_cast<f32x2*>(a.rec2) + ((size_t)tile * T * NW + ht) * 64 + lane : nullptr;
    f32x4 r4_prev = {0.f, 0.f, 0.f, 0.f};
    f32x2 r2_prev = {0.f, 0.f};

    auto step = [&](Slot& X, int t) {
        const int cur = t & 1;
        if constexpr (MODE != 1) asm volatile("" : "+v"(X.v));   // pin the consumption point of this slot (see lstm_bwd_bf16.hip)
        if (t > t0) h_flush(t - 1, cur);          // h_{t-1} sits complete in h_lds[cur]
        f32x4 acc[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
        __builtin_amdgcn_s_setprio(1);
        if constexpr (MODE == 2) {
            const bf16x8 bx = *reinterpret_cast<const bf16x8*>(x_lds + cur * XBUF + (q * 16 + j) * RSX);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                acc[g] = mfma_bf16_16x16x32(wxh[g], bx, acc[g]);
                acc[g] = mfma_bf16_16x16x32(wxl[g], bx, acc[g]);
            }
        }
        {
            const bf16_t* row = h_lds + cur * HBUF + (q * 16 + j) * RSH;
#pragma unroll
            for (int s = 0; s < KSH; ++s) {
                const bf16x8 bv = *reinterpret_cast<const bf16x8*>(row + 8 * s);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    acc[g] = mfma_bf16_16x16x32(whi[g][s], bv, acc[g]);
                    acc[g] = mfma_bf16_16x16x32(wlo[g][s], bv, acc[g]);
                }
            }
        }
        __builtin_amdgcn_s_setprio(0);
        // h-independent vector work behind the MFMA issue: the record of step t-1, the staging of x_{t+1}
        if constexpr (SAVE) {
            if (t > t0) { sv4[(size_t)(t - 1) * NW * 64] = r4_prev; sv2[(size_t)(t - 1) * NW * 64] = r2_prev; }
        }
        float gate[4];
        if constexpr (MODE == 2) {
            if (t + 1 < t1) x_commit(X.v[0], t + 1, cur ^ 1);
#pragma unroll
            for (int g = 0; g < 4; ++g) gate[g] = redeal_hl(acc[g]) + bias_c[g];
            fetch(X, t + 1 + NS);
        } else if constexpr (MODE == 0) {
#pragma unroll
            for (int g = 0; g < 4; ++g) gate[g] = redeal_hl(acc[g]) + X.v[g];
            fetch(X, t + NS);
        } else {
#pragma unroll
            for (int g = 0; g < 4; ++g) gate[g] = redeal_hl(acc[g]) + xconst[g];
        }
        constexpr float L2E = 1.4426950408889634f;
        const float e_i = __builtin_amdgcn_exp2f(-L2E * gate[0]), e_f = __builtin_amdgcn_exp2f(-L2E * gate[1]);
        const float e_g = __builtin_amdgcn_exp2f(2.0f * L2E * gate[2]), e_o = __builtin_amdgcn_exp2f(-L2E * gate[3]);
        const float gi = __builtin_amdgcn_rcpf(1.0f + e_i), gf = __builtin_amdgcn_rcpf(1.0f + e_f);
        const float r_g = __builtin_amdgcn_rcpf(1.0f + e_g), go = __builtin_amdgcn_rcpf(1.0f + e_o);
        const float gg = 1.0f - 2.0f * r_g;
        const float c_prev = c;
        c = gf * c + gi * gg;
        const float e_c = __builtin_amdgcn_exp2f(2.0f * L2E * c);
        const float tc = 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + e_c);
        h = go * tc;
        {
            const bf16x2 s = split2(h);
            bf16_t* hl = h_lds + (cur ^ 1) * HBUF + my_h_off;
            hl[0] = s[0]; hl[4 * RSH] = s[1];
        }
        if constexpr (SAVE) { r4_prev = f32x4{gi, gf, gg, go}; r2_prev = f32x2{c_prev, tc}; }
        __syncthreads();
    };

    Slot X[NS];
    constexpr int LEAD = MODE == 2 ? 1 : 0;       // mode 2 slots run one step ahead (x_{t+1} is staged during step t)
    int t = t0;
    for (int k = (t1 - t0) % NS; k > 0; --k, ++t) { fetch(X[0], t + LEAD); step(X[0], t); }     // peel: the main loop runs whole groups of NS
#pragma unroll
    for (int u = 0; u < NS; ++u) fetch(X[u], t + u + LEAD);
    __builtin_amdgcn_s_waitcnt(0x0F70);           // vmcnt(0) once: only steady-state loads cross the back-edge
    for (; t + NS - 1 < t1; t += NS) {
#pragma unroll
        for (int u = 0; u < NS; ++u) step(X[u], t + u);
    }
    if constexpr (SAVE) { sv4[(size_t)(t1 - 1) * NW * 64] = r4_prev; sv2[(size_t)(t1 - 1) * NW * 64] = r2_prev; }
    h_flush(t1 - 1, t1 & 1);
    if (t1 < T) { if (cst) cst[0] = c; }
    else if (a.h_last && cvalid) a.h_last[(size_t)cb * H + cu] = h;
}

// ---------------------------------------------------------------------------------------------------------------------
// backward recurrence (BPTT).  DH: 0 nothing (encoder top layer: dh_last only), 1 dx of the layer above (fragment order),
// 2 output layer with exactly 4 channels (raw dy prefetched, product with W_out at use), 3 any other channel count.
struct PrefS { f32x4 r4; f32x2 r2; f32x4 up; };

template <int H, int DH>
__global__ void __launch_bounds__(4 * H) lstm_bwd_sp_kernel(LstmBwdArgsS a) {
    constexpr int NW = H / 16, NTHR = 64 * NW;
    constexpr int KS = H / 8;                      // k-steps of 32 over K = 4H
    constexpr int RS = lds_row_stride(H);          // quarter q <-> gate q, KQ = H
    constexpr int PBUF = 4 * 16 * RS;
    extern __shared__ __attribute__((aligned(16))) bf16_t smem_b[];
    bf16_t* dp_lds = smem_b;                                               // [2][PBUF]  rows 0..3 hi, 4..7 lo of each gate quarter
    float* wo_lds = reinterpret_cast<float*>(smem_b + 2 * PBUF);           // [Dy][H]

    const int T = a.T, B = a.B, Dy = a.Dy, tile = blockIdx.x;
    const int t0 = a.t0, t1 = a.t1;                // this launch walks steps t1-1 down to t0 (a time chunk, or the whole window)
    const int tid = threadIdx.x, lane = tid & 63, ht = tid >> 6, q = lane >> 4, j = lane & 15;
    const int cj = j & 3, cr = j >> 2, cu = ht * 16 + q * 4 + cr;
    const int cb = tile * WPT + cj;
    const bool cvalid = cb < B;
    const int bc = cvalid ? cb : B - 1;
    const float vmask = cvalid ? 1.f : 0.f;

    for (int e = tid; e < 2 * PBUF; e += NTHR) smem_b[e] = (bf16_t)0.f;
    if (DH >= 2)
        for (int e = tid; e < Dy * H; e += NTHR) wo_lds[e] = a.Wout[e];

    bf16x8 whi[KS], wlo[KS];
    {
        const size_t off = (size_t)ht * KS * 64 + lane;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            whi[s] = reinterpret_cast<const bf16x8*>(a.WhT_hi)[off + s * 64];
            wlo[s] = reinterpret_cast<const bf16x8*>(a.WhT_lo)[off + s * 64];
        }
    }

    const f32x4* sv4 = reinterpret_cast<const f32x4*>(a.rec4) + ((size_t)tile * T * NW + ht) * 64 + lane;
    const f32x2* sv2 = reinterpret_cast<const f32x2*>(a.rec2) + ((size_t)tile * T * NW + ht) * 64 + lane;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    constexpr int NSB = DH == 2 ? 4 : (DH == 1 ? 5 : 6);                         // record slots = prefetch distance in steps (24 bytes per lane and step out of HBM)
    auto fetch = [&](PrefS& P, int t) {
        const int tc = t < t0 ? t0 : t;            // never reads another chunk's rows
        P.r4 = sv4[(size_t)tc * NW * 64];
        P.r2 = sv2[(size_t)tc * NW * 64];
        const size_t row = (size_t)bc * T + tc;
        if constexpr (DH == 1) {
            const size_t rp = ((size_t)tile * T + tc) * WPT + cj;          // tile-blocked row -> fragment rp/16, lane (q, rp%16), element cr
            P.up = zero4;
            P.up[0] = a.dh_seq[(((rp >> 4) * NW + ht) * 64 + q * 16 + (rp & 15)) * 4 + cr];
        } else if constexpr (DH == 2) P.up = *reinterpret_cast<const f32x4*>(a.dy + row * 4);
        else if constexpr (DH == 3) {
            float d = 0.f;
            for (int dd = 0; dd < Dy; ++dd) d = fmaf(a.dy[row * Dy + dd], wo_lds[dd * H + cu], d);
            P.up = zero4; P.up[0] = d;
        } else P.up = zero4;
    };

    float dc = 0.f, dh_rec = 0.f;
    float tsum[4] = {0.f, 0.f, 0.f, 0.f};

    // dpre_t planes -> HBM: the WPT x 4H block of a step is contiguous; NCH 16-byte chunks of hi then NCH of lo: one per thread
    constexpr int NCH = WPT * 4 * H / 8;
    static_assert(NCH * 2 == NTHR, "one chunk per thread");
    const int fpl = tid >= NCH ? 1 : 0, fcch = tid - fpl * NCH;
    const int fjj = fcch / (4 * H / 8), fc8 = (fcch - fjj * (4 * H / 8)) * 8, fg = fc8 / H, fu = fc8 - fg * H;
    const int f_img_off = (fg * 16 + fjj + 4 * fpl) * RS + fu;
    bf16_t* const f_dst = (fpl ? a.dpre_lo : a.dpre_hi) + (size_t)tile * T * WPT * 4 * H + (size_t)fcch * 8;

    auto step = [&](PrefS& P, int t) {
        asm volatile("" : "+v"(P.r4), "+v"(P.r2), "+v"(P.up));
        bf16_t* dpl = dp_lds + (t & 1) * PBUF;
        float dha;
        if constexpr (DH == 2) {
            dha = 0.f;
#pragma unroll
            for (int dd = 0; dd < 4; ++dd) dha = fmaf(P.up[dd], wo_lds[dd * H + cu], dha);
        } else dha = P.up[0];
        const float s_i = P.r4[0], s_f = P.r4[1], s_g = P.r4[2], s_o = P.r4[3], c_prev = P.r2[0], tc = P.r2[1];
        float dpre[4];
        {
            const float dh = dha * vmask + dh_rec;
            const float d_o = dh * tc;
            const float dcr = dc + dh * s_o * (1.0f - tc * tc);
            const float d_i = dcr * s_g, d_g = dcr * s_i, d_f = dcr * c_prev;
            dc = dcr * s_f;
            dpre[0] = d_i * s_i * (1.0f - s_i);
            dpre[1] = d_f * s_f * (1.0f - s_f);
            dpre[2] = d_g * (1.0f - s_g * s_g);
            dpre[3] = d_o * s_o * (1.0f - s_o);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const bf16x2 s = split2(dpre[g]);
            dpl[(g * 16 + cj) * RS + cu] = s[0];
            dpl[(g * 16 + 4 + cj) * RS + cu] = s[1];
            tsum[g] += dpre[g];
        }
        fetch(P, t - NSB);
        __syncthreads();
        *reinterpret_cast<bf16x8*>(f_dst + (size_t)t * WPT * 4 * H) = *reinterpret_cast<const bf16x8*>(dpl + f_img_off);
        const bf16_t* brow = dpl + (q * 16 + j) * RS;
        f32x4 acc0 = zero4, acc1 = zero4;
#pragma unroll
        for (int s = 0; s < KS; s += 2) {
            const bf16x8 bv0 = *reinterpret_cast<const bf16x8*>(brow + 8 * s);
            const bf16x8 bv1 = *reinterpret_cast<const bf16x8*>(brow + 8 * s + 8);
            acc0 = mfma_bf16_16x16x32(whi[s], bv0, acc0);
            acc1 = mfma_bf16_16x16x32(whi[s + 1], bv1, acc1);
            acc0 = mfma_bf16_16x16x32(wlo[s], bv0, acc0);
            acc1 = mfma_bf16_16x16x32(wlo[s + 1], bv1, acc1);
        }
        dh_rec = redeal_hl(acc0 + acc1);
    };

    __syncthreads();
    PrefS P[NSB];
    f32x2* const bst = a.state ? reinterpret_cast<f32x2*>(a.state) + ((size_t)tile * NW + ht) * 64 + lane : nullptr;
    if (t1 < T) { const f32x2 st = bst[0]; dc = st[0]; dh_rec = st[1]; }      // continue the chunk above: (dc, dh_rec) of step t1
    else if (a.dh_last) dh_rec = vmask * a.dh_last[(size_t)bc * H + cu];
    int t = t1 - 1;
    for (int k = (t1 - t0) % NSB; k > 0; --k, --t) { fetch(P[0], t); step(P[0], t); }
#pragma unroll
    for (int u = 0; u < NSB; ++u) fetch(P[u], t - u);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    for (; t >= t0 + NSB - 1; t -= NSB) {
#pragma unroll
        for (int u = 0; u < NSB; ++u) step(P[u], t - u);
    }
    if (t0 > 0 && bst) bst[0] = f32x2{dc, dh_rec};
    if (cvalid && a.dpre_tsum) {                   // sum_t dpre: the chunk that runs first (t1 == T) writes, later chunks add
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float* d = a.dpre_tsum + (size_t)cb * 4 * H + g * H + cu;
            *d = t1 < T ? *d + tsum[g] : tsum[g];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// xproj: out[row][4H] = W_ih in[row] + b for every tile-blocked row at once (the time-parallel half of the forward), written in
// the recurrent kernel's CELL ORDER: block (row / 4), wave w, lane (q, r*4 + row % 4) holds f32x4 {i,f,g,o} of unit 16w + 4q + r.
// Workgroup = H/16 waves; wave w owns units [16w, 16w+16) x 4 gates with its W_ih hi / lo fragments in registers and walks
// 16-row tiles staged through LDS (hi image, lo image; the forward's operand layout).  Three products per fragment
// (hi*hi, lo*hi, hi*lo).  Dropout: the inter-layer mask is applied to the input while staging ((hi + lo) * scale, split again)
// and the masked planes are kept for the dW_ih GEMM.
template <int H, bool DROP>
__global__ void __launch_bounds__(4 * H) xproj_sp_kernel(XprojArgsS a, int tiles_per_wg) {
    constexpr int NW = H / 16, NTHR = 64 * NW, KS = H / 32, KQ = H / 4, RS = lds_row_stride(KQ), IMG = 4 * 16 * RS;
    __shared__ __attribute__((aligned(16))) bf16_t img[2 * 2 * IMG];      // [buf][plane][IMG]
    const int tid = threadIdx.x, lane = tid & 63, ht = tid >> 6, q = lane >> 4, j = lane & 15;
    bf16x8 whi[4][KS], wlo[4][KS];
    {
        const size_t off = (size_t)ht * KS * 4 * 64 + lane;
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                whi[g][s] = reinterpret_cast<const bf16x8*>(a.W_hi)[off + (s * 4 + g) * 64];
                wlo[g][s] = reinterpret_cast<const bf16x8*>(a.W_lo)[off + (s * 4 + g) * 64];
            }
    }
    f32x4 bias4[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bias4[g] = *reinterpret_cast<const f32x4*>(a.bias + g * H + ht * 16 + q * 4);
    const uint32_t thr = drop_threshold(a.drop_p);
    const float dscale = DROP ? 1.0f / (1.0f - a.drop_p) : 1.0f;
    // a time chunk of every window tile (cpt > 0): chunk-local 16-row tile k of the launch -> global tile (k / cpt) * seq_tiles + tile_off + k % cpt
    const int R = a.rows, n_tiles = a.cpt > 0 ? a.n_chunk_tiles : (R + 15) / 16;
    auto gtile = [&](int k) { return a.cpt > 0 ? (k / a.cpt) * a.seq_tiles + a.tile_off + (k % a.cpt) : k; };
    const int t0 = blockIdx.x * tiles_per_wg, t1 = min(n_tiles, t0 + tiles_per_wg);
    // staging: a 16-row tile = 16 x H bf16 per plane = 2H chunks of 16 B; chunk c of a plane -> (row c / (H/8), cols 8 (c % (H/8)))
    constexpr int CHP = 16 * H / 8;                       // chunks per plane (= NTHR / 2)
    static_assert(2 * CHP == NTHR, "one chunk per thread");
    const int spl = tid >= CHP ? 1 : 0, sc = tid - spl * CHP, sjj = sc / (H / 8), sc8 = (sc - sjj * (H / 8)) * 8;
    const int s_img = spl * IMG + ((sc8 / KQ) * 16 + sjj) * RS + (sc8 % KQ);
    const bf16x8 zero8 = {(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
    bf16x8 st, st2;                                       // DROP: every thread needs both planes of its chunk (threads < CHP do the work)
    auto load_tile = [&](int tl) {
        const int r = gtile(tl) * 16 + sjj;
        if constexpr (DROP) {
            st = zero8; st2 = zero8;
            if (tid < CHP && r < R) {
                st = *reinterpret_cast<const bf16x8*>(a.in_hi + (size_t)r * H + sc8);
                st2 = *reinterpret_cast<const bf16x8*>(a.in_lo + (size_t)r * H + sc8);
            }
        } else {
            st = r < R ? *reinterpret_cast<const bf16x8*>((spl ? a.in_lo : a.in_hi) + (size_t)r * H + sc8) : zero8;
        }
    };
    auto store_tile = [&](int tl, int buf) {
        if constexpr (DROP) {
            if (tid < CHP) {
                const int r = gtile(tl) * 16 + sjj;
                bf16x8 oh, ol;
                const uint32_t idx = (uint32_t)((size_t)r * H + sc8);     // element index = tile-blocked row * H + unit (host replica: shm_dropout_keep_mask_tiled)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float f = (float)st[e] + (float)st2[e];
                    const float m = drop_hash(a.drop_seed, a.drop_stack, idx + e) >= thr ? f * dscale : 0.f;
                    const bf16x2 s = split2(m);
                    oh[e] = s[0]; ol[e] = s[1];
                }
                bf16_t* d = img + buf * 2 * IMG + ((sc8 / KQ) * 16 + sjj) * RS + (sc8 % KQ);
                *reinterpret_cast<bf16x8*>(d) = oh;
                *reinterpret_cast<bf16x8*>(d + IMG) = ol;
                if (r < R) {
                    *reinterpret_cast<bf16x8*>(a.xd_hi + (size_t)r * H + sc8) = oh;
                    *reinterpret_cast<bf16x8*>(a.xd_lo + (size_t)r * H + sc8) = ol;
                }
            }
        } else {
            *reinterpret_cast<bf16x8*>(img + buf * 2 * IMG + s_img) = st;
        }
    };
    if (t0 < t1) { load_tile(t0); store_tile(t0, 0); }
    __syncthreads();
    for (int tl = t0; tl < t1; ++tl) {
        const int cur = (tl - t0) & 1;
        if (tl + 1 < t1) load_tile(tl + 1);
        const bf16_t* rh = img + cur * 2 * IMG + (q * 16 + j) * RS;
        f32x4 acc[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const bf16x8 bh = *reinterpret_cast<const bf16x8*>(rh + 8 * s);
            const bf16x8 bl = *reinterpret_cast<const bf16x8*>(rh + IMG + 8 * s);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                acc[g] = mfma_bf16_16x16x32(whi[g][s], bh, acc[g]);
                acc[g] = mfma_bf16_16x16x32(wlo[g][s], bh, acc[g]);
                acc[g] = mfma_bf16_16x16x32(whi[g][s], bl, acc[g]);
            }
        }
        const int r = gtile(tl) * 16 + j;                 // accumulator column j = tile row j
        if (r < R) {
            f32x4* dst = reinterpret_cast<f32x4*>(a.out) + ((size_t)(r >> 2) * NW + ht) * 64 + q * 16 + (r & 3);
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
                dst[rr * 4] = f32x4{acc[0][rr] + bias4[0][rr], acc[1][rr] + bias4[1][rr], acc[2][rr] + bias4[2][rr], acc[3][rr] + bias4[3][rr]};
        }
        if (tl + 1 < t1) store_tile(tl + 1, cur ^ 1);
        __syncthreads();
    }
}

// dx[r][k] = sum_m dpre[r][m] W_ih[m][k] for every row at once (the time-parallel half of BPTT), three products per fragment;
// output in MFMA fragment order (read by lstm_bwd_sp_kernel<DH = 1>); the inter-layer dropout mask of the layer input is re-applied.
template <int H, bool DROP>
__global__ void __launch_bounds__(4 * H) dx_sp_kernel(DxArgsS a, int tiles_per_wg) {
    constexpr int NW = H / 16, NTHR = 64 * NW, KS = H / 8, RS = lds_row_stride(H), PBUF = 4 * 16 * RS;
    extern __shared__ __attribute__((aligned(16))) bf16_t simg[];         // [buf][plane][PBUF]
    const int tid = threadIdx.x, lane = tid & 63, ht = tid >> 6, q = lane >> 4, j = lane & 15;
    bf16x8 whi[KS], wlo[KS];
    {
        const size_t off = (size_t)ht * KS * 64 + lane;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            whi[s] = reinterpret_cast<const bf16x8*>(a.WxT_hi)[off + s * 64];
            wlo[s] = reinterpret_cast<const bf16x8*>(a.WxT_lo)[off + s * 64];
        }
    }
    const uint32_t thr = drop_threshold(a.drop_p);
    const float dscale = DROP ? 1.0f / (1.0f - a.drop_p) : 1.0f;
    const int R = a.rows, n_tiles = a.cpt > 0 ? a.n_chunk_tiles : (R + 15) / 16;
    auto gtile = [&](int k) { return a.cpt > 0 ? (k / a.cpt) * a.seq_tiles + a.tile_off + (k % a.cpt) : k; };   // see xproj_sp_kernel
    const int t0 = blockIdx.x * tiles_per_wg, t1 = min(n_tiles, t0 + tiles_per_wg);
    constexpr int CHP = 16 * 4 * H / 8, CPT = 2 * CHP / NTHR;            // chunks per plane; = 4 chunks per thread
    static_assert(CPT * NTHR == 2 * CHP && CHP % NTHR == 0, "whole chunks per thread, a thread's chunks stay in one plane");
    const bf16x8 zero8 = {(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
    bf16x8 st[CPT];
    auto load_tile = [&](int tl) {
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const int e = tid + i * NTHR, pl = e >= CHP ? 1 : 0, cch = e - pl * CHP;
            const int jj = cch / (4 * H / 8), c8 = (cch - jj * (4 * H / 8)) * 8, r = gtile(tl) * 16 + jj;
            st[i] = r < R ? *reinterpret_cast<const bf16x8*>((pl ? a.dpre_lo : a.dpre_hi) + (size_t)r * 4 * H + c8) : zero8;
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const int e = tid + i * NTHR, pl = e >= CHP ? 1 : 0, cch = e - pl * CHP;
            const int jj = cch / (4 * H / 8), c8 = (cch - jj * (4 * H / 8)) * 8, g = c8 / H, u = c8 - g * H;
            *reinterpret_cast<bf16x8*>(simg + (buf * 2 + pl) * PBUF + (g * 16 + jj) * RS + u) = st[i];
        }
    };
    if (t0 < t1) { load_tile(t0); store_tile(0); }
    __syncthreads();
    for (int tl = t0; tl < t1; ++tl) {
        const int cur = (tl - t0) & 1;
        if (tl + 1 < t1) load_tile(tl + 1);
        const bf16_t* bh = simg + cur * 2 * PBUF + (q * 16 + j) * RS;
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KS; s += 2) {
            const bf16x8 h0 = *reinterpret_cast<const bf16x8*>(bh + 8 * s), h1 = *reinterpret_cast<const bf16x8*>(bh + 8 * s + 8);
            const bf16x8 l0 = *reinterpret_cast<const bf16x8*>(bh + PBUF + 8 * s), l1 = *reinterpret_cast<const bf16x8*>(bh + PBUF + 8 * s + 8);
            acc0 = mfma_bf16_16x16x32(whi[s], h0, acc0);
            acc1 = mfma_bf16_16x16x32(whi[s + 1], h1, acc1);
            acc0 = mfma_bf16_16x16x32(wlo[s], h0, acc0);
            acc1 = mfma_bf16_16x16x32(wlo[s + 1], h1, acc1);
            acc0 = mfma_bf16_16x16x32(whi[s], l0, acc0);
            acc1 = mfma_bf16_16x16x32(whi[s + 1], l1, acc1);
        }
        f32x4 d = acc0 + acc1;
        const int gt = gtile(tl);
        if (DROP) {
            const uint32_t base = (uint32_t)(((size_t)gt * 16 + j) * H + ht * 16 + q * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) d[e] = drop_hash(a.drop_seed, a.drop_stack, base + e) >= thr ? d[e] * dscale : 0.f;
        }
        *reinterpret_cast<f32x4*>(a.dx + (((size_t)gt * NW + ht) * 64 + lane) * 4) = d;
        if (tl + 1 < t1) store_tile(cur ^ 1);
        __syncthreads();
    }
}

__global__ void __launch_bounds__(256) split_planes_kernel(const float* __restrict__ src, size_t n_valid, size_t n, bf16_t* __restrict__ hi, bf16_t* __restrict__ lo) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const bf16x2 s = split2(i < n_valid ? src[i] : 0.f);
        hi[i] = s[0]; lo[i] = s[1];
    }
}

template <int H>
int launch_fwd_h(const LstmFwdArgsS& a, hipStream_t s) {
    const int tiles = ceil_div(a.B, WPT);
    const bool save = a.rec4 != nullptr;
    const size_t lds = sizeof(bf16_t) * 2 * 4 * 16 * (lds_row_stride(H / 4) + (a.mode == 2 ? lds_row_stride(8) : 0));
    void (*kern)(LstmFwdArgsS);
    if (a.mode == 0) kern = save ? lstm_fwd_sp_kernel<H, 0, true> : lstm_fwd_sp_kernel<H, 0, false>;
    else if (a.mode == 1) kern = save ? lstm_fwd_sp_kernel<H, 1, true> : lstm_fwd_sp_kernel<H, 1, false>;
    else kern = save ? lstm_fwd_sp_kernel<H, 2, true> : lstm_fwd_sp_kernel<H, 2, false>;
    SHM_LAUNCH(kern, dim3(tiles), dim3(4 * H), lds, s, a);
    SHM_LAUNCH_CHECK();
    return SHM_OK;
}

template <int H>
int launch_bwd_h(const LstmBwdArgsS& a, hipStream_t s) {
    const int tiles = ceil_div(a.B, WPT);
    const size_t lds = sizeof(bf16_t) * 2 * 4 * 16 * lds_row_stride(H) + sizeof(float) * (a.dy ? a.Dy * H : 0);
    void (*kern)(LstmBwdArgsS) = a.dy ? (a.Dy == 4 ? lstm_bwd_sp_kernel<H, 2> : lstm_bwd_sp_kernel<H, 3>)
                                 : a.dh_seq ? lstm_bwd_sp_kernel<H, 1> : lstm_bwd_sp_kernel<H, 0>;
    SHM_LAUNCH(kern, dim3(tiles), dim3(4 * H), lds, s, a);
    SHM_LAUNCH_CHECK();
    return SHM_OK;
}

inline int tiles_per_wg_for(int n_tiles) { return n_tiles >= 2048 ? 8 : 1; }

template <int H>
int launch_xproj_h(const XprojArgsS& a, hipStream_t s) {
    const int n_tiles = a.cpt > 0 ? a.n_chunk_tiles : ceil_div(a.rows, 16), tpw = tiles_per_wg_for(n_tiles), grid = ceil_div(n_tiles, tpw);
    if (a.drop_p > 0.f) SHM_LAUNCH((xproj_sp_kernel<H, true>), dim3(grid), dim3(4 * H), 0, s, a, tpw);
    else SHM_LAUNCH((xproj_sp_kernel<H, false>), dim3(grid), dim3(4 * H), 0, s, a, tpw);
    SHM_LAUNCH_CHECK();
    return SHM_OK;
}

template <int H>
int launch_dx_h(const DxArgsS& a, hipStream_t s) {
    const int n_tiles = a.cpt > 0 ? a.n_chunk_tiles : ceil_div(a.rows, 16), tpw = tiles_per_wg_for(n_tiles), grid = ceil_div(n_tiles, tpw);
    const size_t lds = sizeof(bf16_t) * 2 * 2 * 4 * 16 * lds_row_stride(H);
    if (a.drop_p > 0.f) {
        static bool attr = false;
        if (!attr) { SHM_HIP_CHECK(hipFuncSetAttribute((const void*)dx_sp_kernel<H, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); attr = true; }
        SHM_LAUNCH((dx_sp_kernel<H, true>), dim3(grid), dim3(4 * H), lds, s, a, tpw);
    } else {
        static bool attr = false;
        if (!attr) { SHM_HIP_CHECK(hipFuncSetAttribute((const void*)dx_sp_kernel<H, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); attr = true; }
        SHM_LAUNCH((dx_sp_kernel<H, false>), dim3(grid), dim3(4 * H), lds, s, a, tpw);
    }
    SHM_LAUNCH_CHECK();
    return SHM_OK;
}

}  // namespace

#define SHM_DISPATCH_H(fn, what)                                                                          \
    switch (H) {                                                                                          \
        case 32: return fn<32>(a, s);                                                                     \
        case 64: return fn<64>(a, s);                                                                     \
        case 128: return fn<128>(a, s);                                                                   \
        default: set_error(what ": hidden_dim %d unsupported (32, 64, 128)", H); return SHM_EINVAL;     \
    }

int launch_lstm_fwd_sp(int H, const LstmFwdArgsS& a0, hipStream_t s) {
    LstmFwdArgsS a = a0;
    if (a.t1 == 0) { a.t0 = 0; a.t1 = a.T; }                // whole window
    SHM_REQUIRE(a.B > 0 && a.T > 0 && a.t0 >= 0 && a.t0 < a.t1 && a.t1 <= a.T, SHM_EINVAL, "lstm_fwd_sp: empty shape or bad chunk [%d, %d) of %d", a.t0, a.t1, a.T);
    SHM_REQUIRE((a.t0 == 0 && a.t1 == a.T) || a.c_state, SHM_EINVAL, "lstm_fwd_sp: a time chunk needs the cell-state buffer");
    SHM_REQUIRE(a.Wh_hi && a.Wh_lo && a.h_hi && a.h_lo, SHM_EINVAL, "lstm_fwd_sp: packed weights or h planes missing");
    SHM_REQUIRE((a.rec4 != nullptr) == (a.rec2 != nullptr), SHM_EINVAL, "lstm_fwd_sp: both record planes or neither");
    if (a.mode == 2) SHM_REQUIRE(a.x && a.Wx_hi && a.Wx_lo && a.bias && a.In >= 1 && a.In <= 16, SHM_EINVAL, "lstm_fwd_sp: mode 2 needs x, W_ih, bias and In <= 16 (got %d)", a.In);
    else SHM_REQUIRE(a.xproj && (a.mode == 0 || a.mode == 1), SHM_EINVAL, "lstm_fwd_sp: modes 0/1 need the projected input");
    SHM_DISPATCH_H(launch_fwd_h, "lstm_fwd_sp")
}
int launch_lstm_bwd_sp(int H, const LstmBwdArgsS& a0, hipStream_t s) {
    LstmBwdArgsS a = a0;
    if (a.t1 == 0) { a.t0 = 0; a.t1 = a.T; }
    SHM_REQUIRE(a.B > 0 && a.T > 0 && a.t0 >= 0 && a.t0 < a.t1 && a.t1 <= a.T, SHM_EINVAL, "lstm_bwd_sp: empty shape or bad chunk [%d, %d) of %d", a.t0, a.t1, a.T);
    SHM_REQUIRE((a.t0 == 0 && a.t1 == a.T) || a.state, SHM_EINVAL, "lstm_bwd_sp: a time chunk needs the (dc, dh) state buffer");
    SHM_REQUIRE(a.rec4 && a.rec2 && a.dpre_hi && a.dpre_lo && a.WhT_hi && a.WhT_lo, SHM_ESTATE, "lstm_bwd_sp: no saved forward state / dpre planes / weights");
    SHM_REQUIRE(!a.dy || (a.Wout && a.Dy > 0 && a.Dy <= 64), SHM_EINVAL, "lstm_bwd_sp: bad output-layer gradient");
    SHM_DISPATCH_H(launch_bwd_h, "lstm_bwd_sp")
}
int launch_xproj_sp(int H, const XprojArgsS& a, hipStream_t s) {
    SHM_REQUIRE(a.rows > 0 && a.in_hi && a.in_lo && a.W_hi && a.W_lo && a.bias && a.out, SHM_EINVAL, "xproj_sp: operand missing");
    SHM_REQUIRE(!(a.drop_p > 0.f) || (a.xd_hi && a.xd_lo), SHM_EINVAL, "xproj_sp: dropout needs the masked-input planes");
    SHM_DISPATCH_H(launch_xproj_h, "xproj_sp")
}
int launch_dx_sp(int H, const DxArgsS& a, hipStream_t s) {
    SHM_REQUIRE(a.rows > 0 && a.dpre_hi && a.dpre_lo && a.WxT_hi && a.WxT_lo && a.dx, SHM_EINVAL, "dx_sp: operand missing");
    SHM_DISPATCH_H(launch_dx_h, "dx_sp")
}
int launch_split_planes(const float* src, size_t n_valid, size_t n, bf16_t* hi, bf16_t* lo, hipStream_t s) {
    if (n == 0) return SHM_OK;
    const size_t blocks = (n + 255) / 256;
    SHM_LAUNCH(split_planes_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, s, src, n_valid, n, hi, lo);
    SHM_LAUNCH_CHECK();
    return SHM_OK;
}

}  // namespace shm
