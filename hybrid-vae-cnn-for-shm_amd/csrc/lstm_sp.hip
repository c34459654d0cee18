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
#include <type_traits>
#include "kernels.hpp"
#ifndef SHM_SP_COPY_AT
#define SHM_SP_COPY_AT (KS / 2)
#endif

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
    // hi + lo first, in accumulator layout (lane j < 4 adds its own register to that of lane j + 4: one VALU add with a DPP source each),
    // then the three moves that deal registers 1..3 to lanes 4..15.  Same two addends per cell as gathering hi and lo separately
    // (6 moves + 1 add), so the sum is bit-identical; 7 vector instructions per gate instead of 9.
    float s[4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
        s[r] = a[r] + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a[r]), 0x104, 0xF, 0xF, true));   // row_shl:4
    int v = __float_as_int(s[0]);
    v = __builtin_amdgcn_update_dpp(v, __float_as_int(s[1]), 0x114, 0xF, 0x2, false);   // row_shr:4  -> lanes 4..7
    v = __builtin_amdgcn_update_dpp(v, __float_as_int(s[2]), 0x118, 0xF, 0x4, false);   // row_shr:8  -> lanes 8..11
    v = __builtin_amdgcn_update_dpp(v, __float_as_int(s[3]), 0x11C, 0xF, 0x8, false);   // row_shr:12 -> lanes 12..15
    return __int_as_float(v);
}

// ---------------------------------------------------------------------------------------------------------------------
// Operands of the input projection W_ih x + b of a layer whose input is a hidden sequence (round 3).  The input arrives as a bf16
// pair (the h planes of the layer below); the projection uses
//     W_ih ~ fp16(W) + e4m3((W - fp16(W)) 2^15) 2^-15          (pack kinds 7 / 8: 3 bytes per weight instead of 4)
//     x    = f16 hi + f16 lo (exact: the pair carries <= 17 significant bits)  and  x8 = e4m3(x) for the residual product
//     W x ~ [sum_k W8 x8] 2^-15 + sum_k (W16 xhi + W16 xlo)     in THIS order (k-step major, gates inner), bias added last
// — the canonical MFMA sequence of both xproj_sp_kernel and lstm_fwd_sp_fused, so the two are bit-identical.  Emulated on the
// north-star golden (tools/emulate_f16f8_projection.py): mu / logvar within 5e-6 of the fp32 reference, the bf16-pair form's level;
// what it buys is a W_ih that is RESIDENT in the fused kernel (fp16 part 128 KB in LDS, fp8 part 32 VGPRs).
struct XConv { f16x4 hi, lo; unsigned x8; bf16x4 dh, dl; };
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
template <bool DROP>
__device__ __forceinline__ XConv x_convert(const u32x2 raw_hi, const u32x2 raw_lo, uint64_t seed, uint32_t stack, uint32_t idx, uint32_t thr, float dscale) {
    const bf16x4 sh = __builtin_bit_cast(bf16x4, raw_hi), sl = __builtin_bit_cast(bf16x4, raw_lo);
    XConv o;
    float m[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float f = (float)sh[e] + (float)sl[e];
        m[e] = DROP ? (drop_hash(seed, stack, idx + e) >= thr ? f * dscale : 0.f) : f;   // element index = tile-blocked row * H + unit (host replica: shm_dropout_keep_mask_tiled)
        const f16_t h = (f16_t)m[e];
        o.hi[e] = h; o.lo[e] = (f16_t)(m[e] - (float)h);
        if (DROP) { const bf16x2 s2 = split2(m[e]); o.dh[e] = s2[0]; o.dl[e] = s2[1]; }
        m[e] = fminf(fmaxf(m[e], -SHM_F8_MAX), SHM_F8_MAX);
    }
    int p8 = 0;
    p8 = __builtin_amdgcn_cvt_pk_fp8_f32(m[0], m[1], p8, false);
    p8 = __builtin_amdgcn_cvt_pk_fp8_f32(m[2], m[3], p8, true);
    o.x8 = (unsigned)p8;
    return o;
}

typedef __attribute__((address_space(1))) unsigned gu32;
#define SHM_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
constexpr int DLY = 4;           // pair launch: a step's h planes are published DLY steps after they were stored
constexpr int PAIR_GROUP = 128;  // pair launch: tiles per ticket group (two workgroups per tile -> 256 CUs)
constexpr unsigned SPIN_MAX = 1u << 21;   // bounded consumer spin (a safety net: ~1 s); on expiry the err word is set

// ---------------------------------------------------------------------------------------------------------------------
// forward recurrence.  MODE 0: xproj stream, 1: xproj per window, 2: fp32 x (In <= 16) projected in the loop.
// PRODUCE (pair launch, lstm_fwd_pair_sp_kernel): the h planes are consumed by the layer-1 workgroup of the same tile inside this
// launch.  The hand-off is the sc1 form of lstm_fwd_bf16.hip (cdna_hip_programming.md section 6 Guideline 16, R1): wave 0 stores
// BOTH planes of a step write-through (buffer_store ... sc1), retires the stores of DLY steps ago with a counted s_waitcnt and
// publishes "steps < n are in memory" with one sc1 flag store.
template <int H, int MODE, bool SAVE, bool PRODUCE>
__device__ __forceinline__ void lstm_fwd_sp_body(const LstmFwdArgsS& a, const int tile, gu32* flag_out) {
    constexpr int NW = H / 16, NTHR = 64 * NW;
    constexpr int KSH = H / 32, KQH = H / 4, RSH = lds_row_stride(KQH);
    constexpr int KQX = 8, RSX = lds_row_stride(KQX);              // input operand of mode 2: one k-step of 32
    constexpr int HBUF = 4 * 16 * RSH, XBUF = 4 * 16 * RSX;
    extern __shared__ __attribute__((aligned(16))) bf16_t smem_b[];
    bf16_t* h_lds = smem_b;                     // [2][HBUF]   rows 0..3 hi, 4..7 lo, 8..15 zero
    bf16_t* x_lds = smem_b + 2 * HBUF;          // [2][XBUF]   (mode 2)

    const int In = a.In > 0 ? a.In : 1, T = a.T, B = a.B;
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
    // PRODUCE: thread c < NCH (all in wave 0) moves chunk c of BOTH planes; OPS = vector-memory instructions wave 0 issues per step (a lower bound)
    constexpr int OPS = 2 + (SAVE ? 2 : 0) + (MODE == 2 ? 1 : 0);
    const uint32_t seq_bytes = (uint32_t)T * WPT * H * 2;
    const __amdgpu_buffer_rsrc_t hi_rsrc = make_rsrc(PRODUCE ? a.h_hi + (size_t)tile * T * WPT * H : nullptr, PRODUCE ? seq_bytes : 0);
    const __amdgpu_buffer_rsrc_t lo_rsrc = make_rsrc(PRODUCE ? a.h_lo + (size_t)tile * T * WPT * H : nullptr, PRODUCE ? seq_bytes : 0);
    const int pc = tid < NCH ? tid : 0, pjj = pc / (H / 8), pc8 = (pc - pjj * (H / 8)) * 8;
    const int p_img_off = ((pc8 / KQH) * 16 + pjj) * RSH + (pc8 % KQH);
    auto h_flush = [&](int t_done, int buf) {
        if constexpr (PRODUCE) {
            if (tid < NCH) {
                const u32x4 vh = *reinterpret_cast<const u32x4*>(h_lds + buf * HBUF + p_img_off);
                const u32x4 vl = *reinterpret_cast<const u32x4*>(h_lds + buf * HBUF + p_img_off + 4 * RSH);
                const uint32_t off = (uint32_t)(t_done * WPT * H + pc * 8) * 2;
                __builtin_amdgcn_raw_buffer_store_b128(vh, hi_rsrc, off, 0, 16);
                __builtin_amdgcn_raw_buffer_store_b128(vl, lo_rsrc, off, 0, 16);
                // in-order vmcnt: with at most (DLY-1)*OPS younger operations outstanding the stores of step t_done - DLY + 1 have retired;
                // publish one step more conservatively than that
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DLY - 1) * OPS) : "memory");
                if (tid == 0 && t_done >= DLY) __hip_atomic_store(flag_out, (unsigned)(t_done - DLY + 1), SHM_RLX_AGENT);
            }
        } else {
            if (tid < 2 * NCH)
                *reinterpret_cast<bf16x8*>(f_dst + (size_t)t_done * WPT * H) = *reinterpret_cast<const bf16x8*>(h_lds + buf * HBUF + f_img_off);
        }
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
    if constexpr (PRODUCE) {
        if (tid < NCH) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (tid == 0) __hip_atomic_store(flag_out, (unsigned)T, SHM_RLX_AGENT);
        }
    }
    if (t1 < T) { if (cst) cst[0] = c; }
    else if (a.h_last && cvalid) a.h_last[(size_t)cb * H + cu] = h;
}

template <int H, int MODE, bool SAVE>
__global__ void __launch_bounds__(4 * H) lstm_fwd_sp_kernel(LstmFwdArgsS a) {
    lstm_fwd_sp_body<H, MODE, SAVE, false>(a, blockIdx.x, nullptr);
}

// ---------------------------------------------------------------------------------------------------------------------
// Fused layer-1 forward (round 3): the consumer half of lstm_fwd_pair_sp_kernel.  The input projection W_ih x_t + b of a layer whose
// input is the hidden sequence of the layer below used to be its own time-parallel launch (xproj_sp_kernel) feeding the recurrence
// through a 268 MB fp32 stream per layer; here it is formed INSIDE the recurrent workgroup, so that both layers of a stack can run
// as one layer-pipelined launch (the recurrences of layer 0 and layer 1 overlap on 2 x 128 CUs) and the stream never exists.
// What made that impossible before is the register file: W_hh hi + lo (128 VGPRs) leave no room for W_ih hi + lo.  Now
//   * W_ih hi lives in LDS (fragment order, 4H x H bf16 = 128 KB at H = 128: the CU's 160 KB hold it beside the operand images),
//   * W_ih lo is streamed from L2, one k-step (4 fragments = 16 VGPRs) per time step, issued at the top of the step,
//   * the projection is formed FOUR STEPS AT A TIME: the 16 MFMA columns carry 4 steps x 4 windows, so a k-step's three products
//     (hi*hi, lo*hi, hi*lo) cost 12 MFMAs per wave and time step instead of 48 — k-step s of the NEXT group runs during step s of
//     the current one, off the recurrence's dependency chain;
//   * the accumulated projection (column 4*s + w = step s, window w) is picked apart with DPP row shifts at its step: cell lane
//     (q, jj) takes register jj >> 2 of lane (q, 4*s + (jj & 3)).
// The MFMA sequence per output element (k-step major, products hi*hi, lo*hi, hi*lo, bias added last) is that of xproj_sp_kernel,
// and the recurrent part is that of lstm_fwd_sp_body<MODE 0>, so the results are BIT-IDENTICAL to the two-launch form
// (tests/test_gpu_vae.py::test_split_pair_launch_is_bit_identical).
// Input hand-off: the layer below (lstm_fwd_sp_body<..., PRODUCE>) stores its h planes with sc1 stores and publishes its progress;
// this workgroup polls that word and reads the planes ONLY with sc1 buffer loads, one 16-byte load per thread and group of 4 steps.
template <int S> __device__ __forceinline__ float xsel(const f32x4 x) {
    // dest lane jj of each 16-lane row <- register jj >> 2 of lane 4*S + (jj & 3): bank r (lanes 4r..4r+3) reads lanes 4S..4S+3
    auto mv = [](int old, float src, auto rc) {
        constexpr int R = decltype(rc)::value;
        constexpr int ctrl = R > S ? 0x110 + 4 * (R - S) : (R < S ? 0x100 + 4 * (S - R) : 0xE4);   // row_shr / row_shl / identity quad_perm
        return __builtin_amdgcn_update_dpp(old, __float_as_int(src), ctrl, 0xF, 1 << R, false);
    };
    int v = 0;
    v = mv(v, x[0], std::integral_constant<int, 0>{});
    v = mv(v, x[1], std::integral_constant<int, 1>{});
    v = mv(v, x[2], std::integral_constant<int, 2>{});
    v = mv(v, x[3], std::integral_constant<int, 3>{});
    return __int_as_float(v);
}

template <int H, bool DROP, bool SAVE>
__device__ __forceinline__ void lstm_fwd_sp_fused(const LstmFwdArgsS& a, const int tile, gu32* flag_in, gu32* err) {
    constexpr int NW = H / 16, NTHR = 64 * NW;
    constexpr int KSH = H / 32, KQH = H / 4, RSH = lds_row_stride(KQH);
    constexpr int HBUF = 4 * 16 * RSH;                 // bf16 elements of one operand image (16 rows x H, quarter-major)
    constexpr int WLDS = 4 * H * H;                    // W_ih hi, fragment order [wave][k-step][gate][lane][8]
    static_assert(KSH <= 4, "one k-step of the projection per time step of a 4-step group");
    extern __shared__ __attribute__((aligned(16))) bf16_t smem_b[];
    bf16_t* w_lds = smem_b;                            // W_ih fp16 part
    bf16_t* h_lds = smem_b + WLDS;                     // [2][HBUF]   rows 0..3 hi, 4..7 lo
    f16_t* x_lds = reinterpret_cast<f16_t*>(h_lds + 2 * HBUF);       // [2 planes][HBUF] f16 hi / lo; row 4*s + w = step s of the group, window w
    unsigned char* x8_lds = reinterpret_cast<unsigned char*>(x_lds + 2 * HBUF);   // [HBUF] bytes: e4m3(x), same geometry
    float* b_lds = reinterpret_cast<float*>(x8_lds + HBUF);          // [4H] b_ih + b_hh: read back once per group (a global load there would
                                                                      // wait, in order, behind every record store still in flight)

    const int T = a.T, B = a.B;
    const int NG = (T + 3) >> 2;                       // groups of 4 steps
    const int tid = threadIdx.x, lane = tid & 63, ht = tid >> 6, q = lane >> 4, j = lane & 15;
    const int cj = j & 3, cr = j >> 2, cu = ht * 16 + q * 4 + cr;
    const int cb = tile * WPT + cj;
    const bool cvalid = cb < B;

    for (int e = tid; e < 4 * HBUF + HBUF / 2; e += NTHR) h_lds[e] = (bf16_t)0.f;      // both h images, the two f16 x images, the byte image

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
    // W_ih: this wave's fragments; the fp16 part -> LDS (read back only by this wave), the fp8 residual -> registers (2 per fragment)
    const size_t wfrag = (size_t)ht * KSH * 4 * 64 + lane;
    f16x8* const wx_lds = reinterpret_cast<f16x8*>(w_lds) + wfrag;
    long wx8[4][KSH];
    {
        const f16x8* px = reinterpret_cast<const f16x8*>(a.Wx_hi) + wfrag;
        const long* p8 = reinterpret_cast<const long*>(a.Wx_lo) + wfrag;
#pragma unroll
        for (int i = 0; i < KSH * 4; ++i) wx_lds[i * 64] = px[i * 64];
#pragma unroll
        for (int s = 0; s < KSH; ++s)
#pragma unroll
            for (int g = 0; g < 4; ++g) { wx8[g][s] = p8[(s * 4 + g) * 64]; asm volatile("" : "+v"(wx8[g][s])); }   // opaque: a loop-invariant load the
    }                                                                                                                     // compiler would re-issue inside the loop
    for (int e = tid; e < 4 * H; e += NTHR) b_lds[e] = a.bias[e];
    const float* const bias_p = b_lds + ht * 16 + q * 4;

    const int my_h_off = ((cu / KQH) * 16 + cj) * RSH + (cu % KQH);
    constexpr int NCH = WPT * H / 8;
    static_assert(2 * NCH <= NTHR, "one chunk per thread");
    const int fpl = tid >= NCH ? 1 : 0, fc = tid < 2 * NCH ? tid - fpl * NCH : 0;
    const int fjj = fc / (H / 8), fc8 = (fc - fjj * (H / 8)) * 8;
    const int f_img_off = ((fc8 / KQH) * 16 + fjj + 4 * fpl) * RSH + (fc8 % KQH);
    // global streams of this workgroup go through buffer descriptors (SGPRs) + 32-bit offsets: 64-bit pointers per stream cost the
    // registers this kernel does not have
    const uint32_t seq_bytes = (uint32_t)T * WPT * H * 2;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const __amdgpu_buffer_rsrc_t out_hi = make_rsrc(a.h_hi + (size_t)tile * T * WPT * H, seq_bytes);
    const __amdgpu_buffer_rsrc_t out_lo = make_rsrc(a.h_lo + (size_t)tile * T * WPT * H, seq_bytes);
    bf16_t* const f_dst = (fpl ? a.h_lo : a.h_hi) + (size_t)tile * T * WPT * H + (size_t)fc * 8;
    auto h_flush = [&](int t_done, int buf) {
        if constexpr (NCH % 64 == 0) {                 // the plane is wave-uniform: descriptor path
            if (tid < 2 * NCH) {
                const u32x4 v = *reinterpret_cast<const u32x4*>(h_lds + buf * HBUF + f_img_off);
                if (wv >= NCH / 64) __builtin_amdgcn_raw_buffer_store_b128(v, out_lo, (uint32_t)fc * 16, (uint32_t)t_done * WPT * H * 2, 0);
                else __builtin_amdgcn_raw_buffer_store_b128(v, out_hi, (uint32_t)fc * 16, (uint32_t)t_done * WPT * H * 2, 0);
            }
        } else if (tid < 2 * NCH)
            *reinterpret_cast<bf16x8*>(f_dst + (size_t)t_done * WPT * H) = *reinterpret_cast<const bf16x8*>(h_lds + buf * HBUF + f_img_off);
    };

    // ---- staging of the layer input, a group of 4 steps (16 tile-blocked rows x H, contiguous in both planes) at a time.
    // No dropout: thread c < 2 CHP moves chunk c % CHP of plane c / CHP (wave-uniform).  Dropout: thread c < CHP needs both planes of
    // its chunk (mask on hi + lo, rescale, split again) and also keeps the masked planes for the dW_ih GEMM.
    // thread c < 2 CHP takes HALF a 16-byte chunk (4 elements) of BOTH planes: the conversion needs hi + lo of an element in one thread
    constexpr int CHP = 16 * H / 8;
    static_assert(2 * CHP == NTHR, "one half chunk per thread, every thread stages");
    const int sc = (tid >> 1) % CHP, sjj = sc / (H / 8);
    const int sc8 = (sc - sjj * (H / 8)) * 8 + 4 * (tid & 1);
    const int s_img = ((sc8 / KQH) * 16 + sjj) * RSH + (sc8 % KQH);
    const __amdgpu_buffer_rsrc_t in_hi = make_rsrc(a.in_hi + (size_t)tile * T * WPT * H, seq_bytes);
    const __amdgpu_buffer_rsrc_t in_lo = make_rsrc(a.in_lo + (size_t)tile * T * WPT * H, seq_bytes);
    const uint32_t thr = drop_threshold(a.drop_p);
    const float dscale = DROP ? 1.0f / (1.0f - a.drop_p) : 1.0f;
    int avail = 0;                                     // wave-uniform: producer steps < avail are in memory
    const uint64_t flag_in_s = uniform_addr((const void*)flag_in);
    auto wait_for = [&](int need) {
        if (avail >= need) return;
        unsigned v, spins = 0;
#pragma clang loop unroll(disable)
        for (;;) {
            v = poll_uniform(flag_in_s);                      // scalar load: no vmcnt drain (common.hpp)
            if ((int)v >= need) break;
            if (++spins > SPIN_MAX) {                  // give up: finish with garbage, flagged in the err word
                if (lane == 0) __hip_atomic_store(err, 1u + (unsigned)tile, SHM_RLX_AGENT);
                v = 0x7fffffff;
                break;
            }
            __builtin_amdgcn_s_sleep(8);
        }
        avail = (int)v;
    };
    const __amdgpu_buffer_rsrc_t xd_hi = make_rsrc(DROP ? a.xd_hi + (size_t)tile * T * WPT * H : nullptr, DROP ? seq_bytes : 0);
    const __amdgpu_buffer_rsrc_t xd_lo = make_rsrc(DROP ? a.xd_lo + (size_t)tile * T * WPT * H : nullptr, DROP ? seq_bytes : 0);
    const uint32_t s_off = (uint32_t)(sjj * H + sc8) * 2;          // this thread's half chunk inside a 16-row group block (bytes)
    struct Raw { u32x2 h, l; };
    auto x_fetch = [&](Raw& R, int G) {               // rows of steps 4G .. 4G+3 (rows of steps >= T: out of range, read as zero)
        const int need = G >= NG ? 0 : (4 * G + 4 < T ? 4 * G + 4 : T);       // G == NG: a fetch of nothing (see step)
        wait_for(need);
        const uint32_t goff = (uint32_t)(4 * G * WPT * H) * 2;
        // inline asm: the compiler does not track these two loads, so it cannot put its own (uncounted: the step bodies branch) vmcnt(0) in
        // front of their first use — x_ready() below waits with the count this kernel knows
        asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen sc1" : "=v"(R.h) : "v"(s_off), "s"(in_hi), "s"(goff) : "memory");
        asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen sc1" : "=v"(R.l) : "v"(s_off), "s"(in_lo), "s"(goff) : "memory");
    };
    // Every wave issues, after the two loads of x_fetch (top of step S = 0) and before their use (step S = 1), at least its two record
    // stores (SAVE) — waves 0 and 1 also an h-plane store.  vmcnt retires in order, so "at most 2 outstanding" covers the loads without
    // waiting for the record stores just issued (the compiler's own wait here was vmcnt(0): the stores' round trip, once per group).
    auto x_ready = [&](Raw& R, bool in_loop) {
        if (SAVE && in_loop) asm volatile("s_waitcnt vmcnt(2)" : "+v"(R.h), "+v"(R.l) : : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" : "+v"(R.h), "+v"(R.l) : : "memory");
    };
    auto x_commit = [&](Raw& R, int G, bool in_loop = true) {
        x_ready(R, in_loop);
        const size_t r = ((size_t)tile * T + 4 * G) * WPT + sjj;               // tile-blocked row
        const XConv cv = x_convert<DROP>(R.h, R.l, a.drop_seed, a.drop_stack, (uint32_t)(r * H + sc8), thr, dscale);
        *reinterpret_cast<f16x4*>(x_lds + s_img) = cv.hi;
        *reinterpret_cast<f16x4*>(x_lds + HBUF + s_img) = cv.lo;
        *reinterpret_cast<unsigned*>(x8_lds + s_img) = cv.x8;
        if constexpr (DROP) {                          // the masked planes are kept for the dW_ih GEMM (rows of steps >= T: out of range, dropped)
            const uint32_t goff = (uint32_t)(4 * G * WPT * H) * 2;
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, cv.dh), xd_hi, s_off, goff, 0);
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, cv.dl), xd_lo, s_off, goff, 0);
        }
    };

    // ---- projection of a whole group in the canonical order (see x_convert): XC[g] = [sum_s W8 x8] 2^-15 + sum_s (W16 xh + W16 xl) + bias.
    // ONE set of accumulators: the block runs at the LAST step of a group, right after that step has taken its own values out of XC,
    // beside the cell update's vector work (the MFMAs depend on nothing the step computes).
    f32x4 XC[4];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    auto xp_group = [&]() {
        const int xoff = (q * 16 + j) * RSH;
#pragma unroll
        for (int g = 0; g < 4; ++g) XC[g] = zero4;
#pragma unroll
        for (int s = 0; s < KSH; ++s) {
            const long b8 = *reinterpret_cast<const long*>(x8_lds + xoff + 8 * s);
#pragma unroll
            for (int g = 0; g < 4; ++g) XC[g] = mfma_fp8_16x16x32(wx8[g][s], b8, XC[g]);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) XC[g] = XC[g] * SHM_W8_UNSCALE;
        // the fp16 fragments of W_ih come out of LDS one fragment AHEAD of the MFMAs that use them
        f16x8 wn = wx_lds[0];
#pragma unroll
        for (int s = 0; s < KSH; ++s) {
            const f16x8 bh = *reinterpret_cast<const f16x8*>(x_lds + xoff + 8 * s);
            const f16x8 bl = *reinterpret_cast<const f16x8*>(x_lds + HBUF + xoff + 8 * s);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f16x8 wh = wn;
                if (s * 4 + g + 1 < KSH * 4) wn = wx_lds[(s * 4 + g + 1) * 64];
                XC[g] = mfma_f16_16x16x32(wh, bh, XC[g]);
                XC[g] = mfma_f16_16x16x32(wh, bl, XC[g]);
            }
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) XC[g] = XC[g] + *reinterpret_cast<const f32x4*>(bias_p + g * H);
    };

    // ---- prologue: group 0 staged and projected
    {
        Raw R0;
        x_fetch(R0, 0);
        __syncthreads();                               // images zeroed
        x_commit(R0, 0, false);
        __syncthreads();
        xp_group();
        __syncthreads();                               // the image is free for group 1
    }

    float c = 0.f, h = 0.f;
    // BPTT records: stored at the END of their own step (the two-launch form keeps them across the barrier to store them behind the next
    // step's MFMA issue: six registers this kernel needs elsewhere)
    const __amdgpu_buffer_rsrc_t rec4 = make_rsrc(SAVE ? a.rec4 + (size_t)tile * T * NW * 64 * 4 : nullptr, SAVE ? (uint32_t)T * NW * 64 * 16 : 0);
    const __amdgpu_buffer_rsrc_t rec2 = make_rsrc(SAVE ? a.rec2 + (size_t)tile * T * NW * 64 * 2 : nullptr, SAVE ? (uint32_t)T * NW * 64 * 8 : 0);
    const uint32_t rec_off = (uint32_t)tid * 16;
    Raw RX;

    // step S of group G.  S == 0: the rows of group G+1 are requested (sc1 loads); S == 1: they go into the x image (free since the
    // projection block of the group before); S == 3: group G+1 is projected.
    auto step = [&](auto sc_, const int G) {
        constexpr int S = decltype(sc_)::value;
        const int t = 4 * G + S;
        const bool more = G + 1 < NG;
        // unconditional (behind the last group: all rows out of range, nothing waited for): a fetch under `if (more)` makes RX a merge of
        // two definitions, and the copy the compiler then inserts would read the destination of a load it does not know is pending
        // (tools/lint_pending_loads.py checks the assembly for exactly that)
        if (S == 0) x_fetch(RX, G + 1);
        float gate[4];
        const int cur = t & 1;
        if (t < T) {
            if (t > 0) h_flush(t - 1, cur);
            f32x4 acc[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g] = zero4;
            // The recurrent MFMAs are issued with the step's h-independent vector work BETWEEN them (it runs in the matrix pipe's shadow):
            // this step's values picked out of XC, and at S == 1 the conversion of the next group's input rows.  A wave issues in order:
            // placed behind the MFMA block that work would wait for the pipe, in front of the barrier it delays every wave's arrival.
            bf16x8 bv[KSH];                            // the whole operand row up front: one LDS latency, not one per k-step
            {
                const bf16_t* row = h_lds + cur * HBUF + (q * 16 + j) * RSH;
#pragma unroll
                for (int s = 0; s < KSH; ++s) bv[s] = *reinterpret_cast<const bf16x8*>(row + 8 * s);
            }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int s = 0; s < KSH; ++s) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    acc[g] = mfma_bf16_16x16x32(whi[g][s], bv[s], acc[g]);
                    acc[g] = mfma_bf16_16x16x32(wlo[g][s], bv[s], acc[g]);
                }
            }
            float xs[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) xs[g] = xsel<S>(XC[g]);
            // (no branch on `more` here: one basic block, or the interleave below does not apply.  Without a next group RX is stale, the image
            // it lands in is never read and the masked-plane stores are out of range, i.e. dropped)
            if (S == 1) x_commit(RX, G + 1);
#pragma unroll
            for (int i = 0; i < 8 * KSH; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // one MFMA ...
                __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);     // ... two vector instructions
            }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < 4; ++g) gate[g] = redeal_hl(acc[g]) + xs[g];
        }                                              // (t >= T only happens in the last group: nothing left to stage)
        auto cell = [&]() {
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
                const bf16x2 s2 = split2(h);
                bf16_t* hl = h_lds + (cur ^ 1) * HBUF + my_h_off;
                hl[0] = s2[0]; hl[4 * RSH] = s2[1];
            }
            if constexpr (SAVE) {
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{gi, gf, gg, go}), rec4, rec_off, (uint32_t)t * NW * 64 * 16, 0);
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, f32x2{c_prev, tc}), rec2, rec_off >> 1, (uint32_t)t * NW * 64 * 8, 0);
            }
        };
        if (S == 3 && more) {                          // XC is dead from here on in this group: the next group's projection, with the cell
            if (t < T) {                               // update's vector work between its MFMAs
                __builtin_amdgcn_sched_barrier(0);
                xp_group();
                cell();
#pragma unroll
                for (int i = 0; i < 12 * KSH; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            } else xp_group();
        } else if (t < T) cell();
        __syncthreads();
    };
    for (int G = 0; G < NG; ++G) {
        step(std::integral_constant<int, 0>{}, G);
        step(std::integral_constant<int, 1>{}, G);
        step(std::integral_constant<int, 2>{}, G);
        step(std::integral_constant<int, 3>{}, G);
    }
    h_flush(T - 1, T & 1);
    if (a.h_last && cvalid) a.h_last[(size_t)cb * H + cu] = h;
}

// Two layers of a stack, one launch (the split-precision twin of lstm_fwd_pair_bf16_kernel).  sync: [0] ticket, flag of tile i at
// sync[16 * (1 + i)] (a 64-byte line each), all zero at launch: the previous pair launch zeroed it (the plan alternates two blocks;
// shm_vae_pack_weights zeroes the first); err: the plan's sticky status word.  Roles come from a ticket, not from blockIdx: a consumer
// only ever waits for a workgroup that drew a smaller ticket, i.e. one that is already running, so the launch cannot deadlock whatever
// the dispatch order or residency.
template <int H, int MODE0, bool DROP, bool SAVE>
__global__ void __launch_bounds__(4 * H) lstm_fwd_pair_sp_kernel(LstmFwdPairArgsS p) {
    extern __shared__ __attribute__((aligned(16))) bf16_t smem_b[];
    gu32* sync = (gu32*)p.sync;
    const int tiles = gridDim.x >> 1;
    int* role_lds = reinterpret_cast<int*>(reinterpret_cast<char*>(smem_b) + p.role_lds_off);
    if (threadIdx.x == 0) *role_lds = (int)__hip_atomic_fetch_add(sync, 1u, SHM_RLX_AGENT);
    __syncthreads();
    const int ticket = *role_lds;
    __syncthreads();
    const int grp = ticket / (2 * PAIR_GROUP), r = ticket - grp * 2 * PAIR_GROUP, base = grp * PAIR_GROUP;
    const int ntg = min(PAIR_GROUP, tiles - base);
    const bool producer = r < ntg;
    const int tile = base + (producer ? r : r - ntg);
    if (threadIdx.x == 0 && producer) {                   // nobody touches the other block in this launch
        p.sync_other[16 * (1 + tile)] = 0u;
        if (ticket == 0) p.sync_other[0] = 0u;
    }
    if (producer) lstm_fwd_sp_body<H, MODE0, SAVE, true>(p.l0, tile, sync + 16 * (1 + tile));
    else lstm_fwd_sp_fused<H, DROP, SAVE>(p.l1, tile, sync + 16 * (1 + tile), (gu32*)p.err);
}

// ---------------------------------------------------------------------------------------------------------------------
// backward recurrence (BPTT).  DH: 0 nothing (encoder top layer: dh_last only), 1 dx of the layer above (fragment order),
// 2 output layer with exactly 4 channels (raw dy prefetched, product with W_out at use), 3 any other channel count.
struct PrefS { f32x4 r4; f32x2 r2; f32x4 up; };

// PRODUCE (lstm_bwd_dx_pair_sp_kernel): the dpre planes are consumed by the dx workgroup of the same tile inside this launch.  All
// eight waves store a step's planes (one 16-byte chunk per thread) write-through (sc1); each retires its store of DLY steps ago with a
// counted s_waitcnt right after storing, and lane 0 publishes "steps >= t + DLY are in memory" behind the NEXT step's workgroup barrier
// (every storing wave has passed its wait by then: the R1 form of cdna_hip_programming.md section 6 G16, as in lstm_bwd_bf16.hip).
template <int H, int DH, bool PRODUCE>
__device__ __forceinline__ void lstm_bwd_sp_body(const LstmBwdArgsS& a, const int tile, gu32* flag_out) {
    constexpr int NW = H / 16, NTHR = 64 * NW;
    constexpr int KS = H / 8;                      // k-steps of 32 over K = 4H
    constexpr int RS = lds_row_stride(H);          // quarter q <-> gate q, KQ = H
    constexpr int PBUF = 4 * 16 * RS;
    extern __shared__ __attribute__((aligned(16))) bf16_t smem_b[];
    bf16_t* dp_lds = smem_b;                                               // [2][PBUF]  rows 0..3 hi, 4..7 lo of each gate quarter
    float* wo_lds = reinterpret_cast<float*>(smem_b + 2 * PBUF);           // [Dy][H]

    const int T = a.T, B = a.B, Dy = a.Dy;
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
    // PRODUCE: vector-memory instructions every wave issues per step: two record loads + its dpre store (a lower bound)
    constexpr int OPS = 3;
    const __amdgpu_buffer_rsrc_t out_rsrc = make_rsrc(PRODUCE ? f_dst - (size_t)fcch * 8 : nullptr, PRODUCE ? (uint32_t)T * WPT * 4 * H * 2 : 0);

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
        if constexpr (PRODUCE) {
            // every storing wave has, in step t + 1, waited until its dpre store of step t + DLY had retired
            if (tid == 0 && t + DLY <= T - 1) __hip_atomic_store(flag_out, (unsigned)(T - (t + DLY)), SHM_RLX_AGENT);
        }
        const bf16_t* brow = dpl + (q * 16 + j) * RS;
        f32x4 acc0 = zero4, acc1 = zero4;
        auto mma_range = [&](int s0, int s1) {
#pragma unroll
            for (int s = s0; s < s1; s += 2) {
                const bf16x8 bv0 = *reinterpret_cast<const bf16x8*>(brow + 8 * s);
                const bf16x8 bv1 = *reinterpret_cast<const bf16x8*>(brow + 8 * s + 8);
                acc0 = mfma_bf16_16x16x32(whi[s], bv0, acc0);
                acc1 = mfma_bf16_16x16x32(whi[s + 1], bv1, acc1);
                acc0 = mfma_bf16_16x16x32(wlo[s], bv0, acc0);
                acc1 = mfma_bf16_16x16x32(wlo[s + 1], bv1, acc1);
            }
        };
        mma_range(0, SHM_SP_COPY_AT);
        // dpre_t -> HBM out of the complete image, every thread one 16-byte chunk: placed inside the MFMA block (in front of it every wave
        // started its matrix work an LDS round trip late)
        if constexpr (PRODUCE) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(dpl + f_img_off);
            __builtin_amdgcn_raw_buffer_store_b128(v, out_rsrc, (uint32_t)(t * WPT * 4 * H + fcch * 8) * 2, 0, 16);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DLY - 1) * OPS) : "memory");   // retires the store of step t + DLY - 1
        } else
            *reinterpret_cast<bf16x8*>(f_dst + (size_t)t * WPT * 4 * H) = *reinterpret_cast<const bf16x8*>(dpl + f_img_off);
        mma_range(SHM_SP_COPY_AT, KS);
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
    if constexpr (PRODUCE) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_store(flag_out, (unsigned)T, SHM_RLX_AGENT);
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

template <int H, int DH>
__global__ void __launch_bounds__(4 * H) lstm_bwd_sp_kernel(LstmBwdArgsS a) {
    lstm_bwd_sp_body<H, DH, false>(a, blockIdx.x, nullptr);
}

// ---------------------------------------------------------------------------------------------------------------------
// (r3) Top-layer backward + dx in one launch.  The top layer's recurrence holds 128 of the 256 CUs, and dx = dpre W_ih of the layer
// below used to be a full-chip launch BEHIND it (94 us, twice per step, on the step's critical path).  Here the workgroup of tile i in the
// second half of the grid forms dx for tile i's rows while the recurrence of tile i is still running: it follows the producer's progress
// word (16 rows = 4 steps at a time, walking down in time as the recurrence does), reads the dpre planes with sc1 loads only, and runs
// dx_sp_kernel's own loop body on them — same fragments, same three products in the same order, same mask: bit-identical output.
// W_ih^T hi / lo live in registers (this role has no recurrent weights); the layer-0 recurrence stays its own launch behind this one.
template <int H, bool DROP>
__device__ __forceinline__ void dx_sp_follow(const DxArgsS& a, const int tile, const int T, gu32* flag_in, gu32* err) {
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
    const uint32_t seq_bytes = (uint32_t)T * WPT * 4 * H * 2;
    const __amdgpu_buffer_rsrc_t in_hi = make_rsrc(a.dpre_hi + (size_t)tile * T * WPT * 4 * H, seq_bytes);
    const __amdgpu_buffer_rsrc_t in_lo = make_rsrc(a.dpre_lo + (size_t)tile * T * WPT * 4 * H, seq_bytes);
    int avail = 0;                                         // wave-uniform: the producer's steps >= T - avail are in memory
    const uint64_t flag_in_s = uniform_addr((const void*)flag_in);
    auto wait_for = [&](int need) {
        if (avail >= need) return;
        unsigned v, spins = 0;
#pragma clang loop unroll(disable)
        for (;;) {
            v = poll_uniform(flag_in_s);                      // scalar load: no vmcnt drain (common.hpp)
            if ((int)v >= need) break;
            if (++spins > SPIN_MAX) {
                if (lane == 0) __hip_atomic_store(err, 0x40000001u + (unsigned)tile, SHM_RLX_AGENT);
                v = 0x7fffffff;
                break;
            }
            __builtin_amdgcn_s_sleep(8);
        }
        avail = (int)v;
    };
    constexpr int CHP = 16 * 4 * H / 8, CPT = 2 * CHP / NTHR;            // chunks per plane; = 4 chunks per thread
    static_assert(CPT * NTHR == 2 * CHP && CHP % NTHR == 0, "whole chunks per thread, a thread's chunks stay in one plane");
    const int n_loc = (T * WPT + 15) / 16;                               // 16-row tiles of this window tile (the last one may be partial)
    // two register sets: the rows of tile c - 2 are requested before tile c is multiplied (one tile of look-ahead left the role latency-bound:
    // 4.4 us per tile against the 3.5 us the recurrence takes for four steps, i.e. the launch ended 50-80 us behind its recurrence)
    struct Stage { u32x4 v[CPT]; };
    auto load_tile = [&](Stage& S, int c) {                               // rows 16c .. 16c+15 of the tile = steps 4c .. 4c+3
        if (c < 0) return;
        wait_for(T - 4 * c);
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const int e = tid + i * NTHR, pl = e >= CHP ? 1 : 0, cch = e - pl * CHP;      // pl is wave-uniform (CHP % NTHR == 0)
            const uint32_t off = (uint32_t)(c * 16 * 4 * H + cch * 8) * 2;                // rows past the window: out of range, read as zero
            S.v[i] = pl ? __builtin_amdgcn_raw_buffer_load_b128(in_lo, off, 0, 16) : __builtin_amdgcn_raw_buffer_load_b128(in_hi, off, 0, 16);
        }
    };
    auto store_tile = [&](Stage& S, int buf) {
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const int e = tid + i * NTHR, pl = e >= CHP ? 1 : 0, cch = e - pl * CHP;
            const int jj = cch / (4 * H / 8), c8 = (cch - jj * (4 * H / 8)) * 8, g = c8 / H, u = c8 - g * H;
            *reinterpret_cast<u32x4*>(simg + (buf * 2 + pl) * PBUF + (g * 16 + jj) * RS + u) = S.v[i];
        }
    };
    // tile c is multiplied out of image `cur`; Snext holds tile c - 1 (requested an iteration ago), Sfree is refilled with tile c - 2
    auto tile_step = [&](int c, int cur, Stage& Sfree, Stage& Snext) {
        load_tile(Sfree, c - 2);
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
        // global 16-row tile index: dx_sp_kernel numbers the tiles over ALL rows; a window tile starts on a 16-row boundary iff T % 4 == 0
        // (the launcher requires it for this kernel)
        const size_t gt = (size_t)tile * (T * WPT / 16) + c;
        if (DROP) {
            const uint32_t base = (uint32_t)((gt * 16 + j) * H + ht * 16 + q * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) d[e] = drop_hash(a.drop_seed, a.drop_stack, base + e) >= thr ? d[e] * dscale : 0.f;
        }
        *reinterpret_cast<f32x4*>(a.dx + ((gt * NW + ht) * 64 + lane) * 4) = d;
        if (c > 0) store_tile(Snext, cur ^ 1);
        __syncthreads();
    };
    Stage S0, S1;
    int c = n_loc - 1;
    load_tile(S0, c); load_tile(S1, c - 1);
    store_tile(S0, 0);
    __syncthreads();
    for (; c >= 1; c -= 2) {
        tile_step(c, 0, S0, S1);
        tile_step(c - 1, 1, S1, S0);
    }
    if (c == 0) tile_step(0, 0, S0, S1);
}

template <int H, int DH, bool DROP>
__global__ void __launch_bounds__(4 * H) lstm_bwd_dx_pair_sp_kernel(LstmBwdDxPairArgsS p) {
    extern __shared__ __attribute__((aligned(16))) bf16_t smem_b[];
    gu32* sync = (gu32*)p.sync;
    const int tiles = gridDim.x >> 1;
    int* role_lds = reinterpret_cast<int*>(reinterpret_cast<char*>(smem_b) + p.role_lds_off);
    if (threadIdx.x == 0) *role_lds = (int)__hip_atomic_fetch_add(sync, 1u, SHM_RLX_AGENT);
    __syncthreads();
    const int ticket = *role_lds;
    __syncthreads();
    const int grp = ticket / (2 * PAIR_GROUP), r = ticket - grp * 2 * PAIR_GROUP, base = grp * PAIR_GROUP;
    const int ntg = min(PAIR_GROUP, tiles - base);
    const bool producer = r < ntg;
    const int tile = base + (producer ? r : r - ntg);
    if (threadIdx.x == 0 && producer) {
        p.sync_other[16 * (1 + tile)] = 0u;
        if (ticket == 0) p.sync_other[0] = 0u;
    }
    if (producer) lstm_bwd_sp_body<H, DH, true>(p.top, tile, sync + 16 * (1 + tile));
    else dx_sp_follow<H, DROP>(p.dx, tile, p.top.T, sync + 16 * (1 + tile), (gu32*)p.err);
}

// ---------------------------------------------------------------------------------------------------------------------
// xproj: out[row][4H] = W_ih in[row] + b for every tile-blocked row at once (the time-parallel half of the forward), written in
// the recurrent kernel's CELL ORDER: block (row / 4), wave w, lane (q, r*4 + row % 4) holds f32x4 {i,f,g,o} of unit 16w + 4q + r.
// Workgroup = H/16 waves; wave w owns units [16w, 16w+16) x 4 gates with its W_ih fragments (fp16 part + fp8 residual, see x_convert)
// in registers and walks 16-row tiles staged through LDS as f16 hi / f16 lo / e4m3 images (the forward's operand layout).  The MFMA
// order is the canonical one of x_convert's comment — the fused layer-1 kernel (lstm_fwd_sp_fused) repeats it bit for bit.
// Dropout: the inter-layer mask is applied to the input while staging ((hi + lo) * scale) and the masked bf16 planes are kept for
// the dW_ih GEMM.  Since round 3 this launch only runs for the constant-input decoder layer 0 and when pair launches are off.
template <int H, bool DROP>
__global__ void __launch_bounds__(4 * H) xproj_sp_kernel(XprojArgsS a, int tiles_per_wg) {
    constexpr int NW = H / 16, NTHR = 64 * NW, KS = H / 32, KQ = H / 4, RS = lds_row_stride(KQ), IMG = 4 * 16 * RS;
    __shared__ __attribute__((aligned(16))) f16_t img[2 * 2 * IMG];       // [buf][plane][IMG]
    __shared__ __attribute__((aligned(16))) unsigned char img8[2 * IMG];  // [buf][IMG] bytes
    const int tid = threadIdx.x, lane = tid & 63, ht = tid >> 6, q = lane >> 4, j = lane & 15;
    f16x8 whi[4][KS];
    long w8[4][KS];
    {
        const size_t off = (size_t)ht * KS * 4 * 64 + lane;
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                whi[g][s] = reinterpret_cast<const f16x8*>(a.W_hi)[off + (s * 4 + g) * 64];
                w8[g][s] = reinterpret_cast<const long*>(a.W_lo)[off + (s * 4 + g) * 64];
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
    // staging: a 16-row tile = 16 x H bf16 per plane = CHP chunks of 16 B; thread c takes HALF a chunk (4 elements) of both planes
    constexpr int CHP = 16 * H / 8;
    static_assert(2 * CHP == NTHR, "one half chunk per thread");
    const int sc = tid >> 1, sjj = sc / (H / 8), sc8 = (sc - sjj * (H / 8)) * 8 + 4 * (tid & 1);
    const int s_img = ((sc8 / KQ) * 16 + sjj) * RS + (sc8 % KQ);
    u32x2 st_h, st_l;
    auto load_tile = [&](int tl) {
        const int r = gtile(tl) * 16 + sjj;
        st_h = u32x2{0u, 0u}; st_l = u32x2{0u, 0u};
        if (r < R) {
            st_h = *reinterpret_cast<const u32x2*>(a.in_hi + (size_t)r * H + sc8);
            st_l = *reinterpret_cast<const u32x2*>(a.in_lo + (size_t)r * H + sc8);
        }
    };
    auto store_tile = [&](int tl, int buf) {
        const int r = gtile(tl) * 16 + sjj;
        const XConv cv = x_convert<DROP>(st_h, st_l, a.drop_seed, a.drop_stack, (uint32_t)((size_t)r * H + sc8), thr, dscale);
        *reinterpret_cast<f16x4*>(img + buf * 2 * IMG + s_img) = cv.hi;
        *reinterpret_cast<f16x4*>(img + buf * 2 * IMG + IMG + s_img) = cv.lo;
        *reinterpret_cast<unsigned*>(img8 + buf * IMG + s_img) = cv.x8;
        if constexpr (DROP) {
            if (r < R) {
                *reinterpret_cast<bf16x4*>(a.xd_hi + (size_t)r * H + sc8) = cv.dh;
                *reinterpret_cast<bf16x4*>(a.xd_lo + (size_t)r * H + sc8) = cv.dl;
            }
        }
    };
    if (t0 < t1) { load_tile(t0); store_tile(t0, 0); }
    __syncthreads();
    for (int tl = t0; tl < t1; ++tl) {
        const int cur = (tl - t0) & 1;
        if (tl + 1 < t1) load_tile(tl + 1);
        const int xoff = (q * 16 + j) * RS;
        const f16_t* rh = img + cur * 2 * IMG + xoff;
        const unsigned char* r8 = img8 + cur * IMG + xoff;
        f32x4 acc[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const long b8 = *reinterpret_cast<const long*>(r8 + 8 * s);
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g] = mfma_fp8_16x16x32(w8[g][s], b8, acc[g]);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = acc[g] * SHM_W8_UNSCALE;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const f16x8 bh = *reinterpret_cast<const f16x8*>(rh + 8 * s);
            const f16x8 bl = *reinterpret_cast<const f16x8*>(rh + IMG + 8 * s);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                acc[g] = mfma_f16_16x16x32(whi[g][s], bh, acc[g]);
                acc[g] = mfma_f16_16x16x32(whi[g][s], bl, acc[g]);
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

// The same projection on bf16 PAIRS (W_ih hi / lo as pack kinds 0 / 5, three products hi*hi, lo*hi, hi*lo: round 2's arithmetic, 16
// significant bits) for the one layer whose projection is NOT fused into a recurrent workgroup: the constant-input decoder layer 0
// (rows = windows, one projection per window, no dropout).  Its own small launch, so residency is not a constraint and the 15-bit W_ih
// would only cost accuracy — and this projection enters EVERY step of that layer with the same error (emulated: recon 9.4e-7 -> 5.7e-7).
template <int H>
__global__ void __launch_bounds__(4 * H) xproj_pair_sp_kernel(XprojArgsS a) {
    constexpr int NW = H / 16, NTHR = 64 * NW, KS = H / 32, KQ = H / 4, RS = lds_row_stride(KQ), IMG = 4 * 16 * RS;
    __shared__ __attribute__((aligned(16))) bf16_t img[2 * IMG];          // [plane][IMG]
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
    const int R = a.rows, tl = blockIdx.x;                               // one 16-row tile per workgroup
    constexpr int CHP = 16 * H / 8;                                       // 16-byte chunks per plane (= NTHR / 2)
    static_assert(2 * CHP == NTHR, "one chunk per thread");
    const int spl = tid >= CHP ? 1 : 0, sc = tid - spl * CHP, sjj = sc / (H / 8), sc8 = (sc - sjj * (H / 8)) * 8;
    const bf16x8 zero8 = {(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
    {
        const int r = tl * 16 + sjj;
        *reinterpret_cast<bf16x8*>(img + spl * IMG + ((sc8 / KQ) * 16 + sjj) * RS + (sc8 % KQ)) =
            r < R ? *reinterpret_cast<const bf16x8*>((spl ? a.in_lo : a.in_hi) + (size_t)r * H + sc8) : zero8;
    }
    __syncthreads();
    const bf16_t* rh = img + (q * 16 + j) * RS;
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
    const int r = tl * 16 + j;                                            // accumulator column j = tile row j
    if (r < R) {
        f32x4* dst = reinterpret_cast<f32x4*>(a.out) + ((size_t)(r >> 2) * NW + ht) * 64 + q * 16 + (r & 3);
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
            dst[rr * 4] = f32x4{acc[0][rr] + bias4[0][rr], acc[1][rr] + bias4[1][rr], acc[2][rr] + bias4[2][rr], acc[3][rr] + bias4[3][rr]};
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

template <int H>
int launch_fwd_pair_h(const LstmFwdPairArgsS& p0, hipStream_t s) {
    LstmFwdPairArgsS p = p0;
    const LstmFwdArgsS &a = p.l0, &b = p.l1;
    const int tiles = ceil_div(a.B, WPT);
    const bool save = a.rec4 != nullptr, drop = b.drop_p > 0.f;
    const size_t lds_cons = sizeof(bf16_t) * ((size_t)4 * H * H + 4 * 4 * 16 * lds_row_stride(H / 4)) + 4 * 16 * lds_row_stride(H / 4) + sizeof(float) * 4 * H;
    const size_t lds_prod = sizeof(bf16_t) * 2 * 4 * 16 * (lds_row_stride(H / 4) + (a.mode == 2 ? lds_row_stride(8) : 0));
    const size_t lds = lds_cons > lds_prod ? lds_cons : lds_prod;
    p.role_lds_off = (int)lds;
    void (*kern)(LstmFwdPairArgsS);
    if (a.mode == 2) kern = drop ? (save ? lstm_fwd_pair_sp_kernel<H, 2, true, true> : lstm_fwd_pair_sp_kernel<H, 2, true, false>)
                                 : (save ? lstm_fwd_pair_sp_kernel<H, 2, false, true> : lstm_fwd_pair_sp_kernel<H, 2, false, false>);
    else kern = drop ? (save ? lstm_fwd_pair_sp_kernel<H, 1, true, true> : lstm_fwd_pair_sp_kernel<H, 1, true, false>)
                     : (save ? lstm_fwd_pair_sp_kernel<H, 1, false, true> : lstm_fwd_pair_sp_kernel<H, 1, false, false>);
    // the consumer's images + W_ih hi exceed the default 64 KB of dynamic LDS: raise the limit once per kernel instance
    static const void* raised[32] = {nullptr};
    bool done = false;
    for (const void* k : raised) done = done || k == (const void*)kern;
    if (!done) {
        SHM_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds + 16)));
        for (auto& k : raised) if (!k) { k = (const void*)kern; break; }
    }
    SHM_LAUNCH(kern, dim3(2 * tiles), dim3(4 * H), lds + 16, s, p);
    SHM_LAUNCH_CHECK();
    return SHM_OK;
}

template <int H>
int launch_bwd_dx_pair_h(const LstmBwdDxPairArgsS& p0, hipStream_t s) {
    LstmBwdDxPairArgsS p = p0;
    const LstmBwdArgsS& a = p.top;
    const int tiles = ceil_div(a.B, WPT);
    const size_t lds_top = sizeof(bf16_t) * 2 * 4 * 16 * lds_row_stride(H) + sizeof(float) * (a.dy ? a.Dy * H : 0);
    const size_t lds_dx = sizeof(bf16_t) * 2 * 2 * 4 * 16 * lds_row_stride(H);
    const size_t lds = lds_top > lds_dx ? lds_top : lds_dx;
    p.role_lds_off = (int)lds;
    const bool drop = p.dx.drop_p > 0.f;
    void (*kern)(LstmBwdDxPairArgsS);
    if (a.dy) kern = a.Dy == 4 ? (drop ? lstm_bwd_dx_pair_sp_kernel<H, 2, true> : lstm_bwd_dx_pair_sp_kernel<H, 2, false>)
                               : (drop ? lstm_bwd_dx_pair_sp_kernel<H, 3, true> : lstm_bwd_dx_pair_sp_kernel<H, 3, false>);
    else kern = drop ? lstm_bwd_dx_pair_sp_kernel<H, 0, true> : lstm_bwd_dx_pair_sp_kernel<H, 0, false>;
    static const void* raised[8] = {nullptr};
    bool done = false;
    for (const void* k : raised) done = done || k == (const void*)kern;
    if (!done) {
        SHM_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds + 16)));
        for (auto& k : raised) if (!k) { k = (const void*)kern; break; }
    }
    SHM_LAUNCH(kern, dim3(2 * tiles), dim3(4 * H), lds + 16, s, p);
    SHM_LAUNCH_CHECK();
    return SHM_OK;
}

inline int tiles_per_wg_for(int n_tiles) { return n_tiles >= 2048 ? 8 : 1; }

template <int H>
int launch_xproj_h(const XprojArgsS& a, hipStream_t s) {
    if (a.pair_bf16) {                                    // bf16-pair operands (constant-input layer): one 16-row tile per workgroup
        SHM_LAUNCH((xproj_pair_sp_kernel<H>), dim3(ceil_div(a.rows, 16)), dim3(4 * H), 0, s, a);
        SHM_LAUNCH_CHECK();
        return SHM_OK;
    }
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
int launch_lstm_fwd_pair_sp(int H, const LstmFwdPairArgsS& a, hipStream_t s) {
    const LstmFwdArgsS &l0 = a.l0, &l1 = a.l1;
    SHM_REQUIRE(l0.B > 0 && l0.T > 0 && l0.B == l1.B && l0.T == l1.T, SHM_EINVAL, "lstm_fwd_pair_sp: empty shape or shape mismatch");
    SHM_REQUIRE(a.sync && a.sync_other && a.sync != a.sync_other && a.err, SHM_EINVAL, "lstm_fwd_pair_sp: sync blocks or status word missing");
    SHM_REQUIRE(l0.Wh_hi && l0.Wh_lo && l0.h_hi && l0.h_lo && l1.Wh_hi && l1.Wh_lo && l1.h_hi && l1.h_lo, SHM_EINVAL, "lstm_fwd_pair_sp: packed weights or h planes missing");
    SHM_REQUIRE((l0.rec4 != nullptr) == (l0.rec2 != nullptr) && (l1.rec4 != nullptr) == (l1.rec2 != nullptr) && (l0.rec4 != nullptr) == (l1.rec4 != nullptr),
                SHM_EINVAL, "lstm_fwd_pair_sp: both layers save both record planes, or neither");
    if (l0.mode == 2) SHM_REQUIRE(l0.x && l0.Wx_hi && l0.Wx_lo && l0.bias && l0.In >= 1 && l0.In <= 16, SHM_EINVAL, "lstm_fwd_pair_sp: layer 0 in mode 2 needs x, W_ih, bias and In <= 16 (got %d)", l0.In);
    else SHM_REQUIRE(l0.mode == 1 && l0.xproj, SHM_EINVAL, "lstm_fwd_pair_sp: layer 0 must be the fp32-input layer (mode 2) or the constant-input layer (mode 1)");
    SHM_REQUIRE(l1.in_hi == l0.h_hi && l1.in_lo == l0.h_lo && l1.Wx_hi && l1.Wx_lo && l1.bias, SHM_EINVAL, "lstm_fwd_pair_sp: layer 1 must read layer 0's h planes and bring W_ih hi / lo and its bias");
    SHM_REQUIRE(!(l1.drop_p > 0.f) || (l1.xd_hi && l1.xd_lo), SHM_EINVAL, "lstm_fwd_pair_sp: dropout needs the masked-input planes");
    SHM_REQUIRE((int64_t)l0.T * WPT * H * 2 < ((int64_t)1 << 31), SHM_EINVAL, "lstm_fwd_pair_sp: T too long for 32-bit buffer offsets");
    SHM_REQUIRE((l0.t0 == 0 && (l0.t1 == 0 || l0.t1 == l0.T)) && (l1.t0 == 0 && (l1.t1 == 0 || l1.t1 == l1.T)), SHM_EINVAL, "lstm_fwd_pair_sp: whole windows only");
    LstmFwdPairArgsS p = a;
    p.l0.t0 = 0; p.l0.t1 = l0.T; p.l1.t0 = 0; p.l1.t1 = l1.T;
    switch (H) {
        case 32: return launch_fwd_pair_h<32>(p, s);
        case 64: return launch_fwd_pair_h<64>(p, s);
        case 128: return launch_fwd_pair_h<128>(p, s);
        default: set_error("lstm_fwd_pair_sp: hidden_dim %d unsupported (32, 64, 128)", H); return SHM_EINVAL;
    }
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
int launch_lstm_bwd_dx_pair_sp(int H, const LstmBwdDxPairArgsS& a0, hipStream_t s) {
    LstmBwdDxPairArgsS a = a0;
    LstmBwdArgsS& t = a.top;
    if (t.t1 == 0) { t.t0 = 0; t.t1 = t.T; }
    SHM_REQUIRE(t.B > 0 && t.T > 0 && t.t0 == 0 && t.t1 == t.T, SHM_EINVAL, "lstm_bwd_dx_pair_sp: whole windows only");
    SHM_REQUIRE(t.T % 4 == 0, SHM_EINVAL, "lstm_bwd_dx_pair_sp: T must be a multiple of 4 (a window tile = whole 16-row dx tiles)");
    SHM_REQUIRE(t.rec4 && t.rec2 && t.dpre_hi && t.dpre_lo && t.WhT_hi && t.WhT_lo, SHM_ESTATE, "lstm_bwd_dx_pair_sp: no saved forward state / dpre planes / weights");
    SHM_REQUIRE(!t.dh_seq, SHM_EINVAL, "lstm_bwd_dx_pair_sp: the producer is the TOP layer of a stack (no dx input of its own)");
    SHM_REQUIRE(!t.dy || (t.Wout && t.Dy > 0 && t.Dy <= 64), SHM_EINVAL, "lstm_bwd_dx_pair_sp: bad output-layer gradient");
    SHM_REQUIRE(a.dx.dpre_hi == t.dpre_hi && a.dx.dpre_lo == t.dpre_lo && a.dx.WxT_hi && a.dx.WxT_lo && a.dx.dx, SHM_EINVAL, "lstm_bwd_dx_pair_sp: the dx role must read the top layer's dpre planes");
    SHM_REQUIRE(a.sync && a.sync_other && a.sync != a.sync_other && a.err, SHM_EINVAL, "lstm_bwd_dx_pair_sp: sync blocks or status word missing");
    SHM_REQUIRE((int64_t)t.T * WPT * 4 * H * 2 < ((int64_t)1 << 31), SHM_EINVAL, "lstm_bwd_dx_pair_sp: T too long for 32-bit buffer offsets");
    switch (H) {
        case 32: return launch_bwd_dx_pair_h<32>(a, s);
        case 64: return launch_bwd_dx_pair_h<64>(a, s);
        case 128: return launch_bwd_dx_pair_h<128>(a, s);
        default: set_error("lstm_bwd_dx_pair_sp: hidden_dim %d unsupported (32, 64, 128)", H); return SHM_EINVAL;
    }
}
int launch_xproj_sp(int H, const XprojArgsS& a, hipStream_t s) {
    SHM_REQUIRE(a.rows > 0 && a.in_hi && a.in_lo && a.W_hi && a.W_lo && a.bias && a.out, SHM_EINVAL, "xproj_sp: operand missing");
    SHM_REQUIRE(!(a.drop_p > 0.f) || (a.xd_hi && a.xd_lo), SHM_EINVAL, "xproj_sp: dropout needs the masked-input planes");
    SHM_REQUIRE(!a.pair_bf16 || (a.drop_p == 0.f && a.cpt == 0), SHM_EINVAL, "xproj_sp: the bf16-pair form serves whole, undropped inputs only");
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
