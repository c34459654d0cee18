// lstm_bwd_bf16.hip — persistent LSTM layer backward (BPTT), bf16 MFMA with fp32 accumulate; gfx950.
// Throughput twin of lstm_bwd.hip (reference: autograd through nn.LSTM, 4DOF/Scripts/03_train_vae.py:269).
//
// The recurrent kernel keeps ONLY the truly sequential product on the CU:
//     dh_rec^T[H x 16] = W_hh^T[H x 4H] dpre_t^T          (W_hh^T fragments in registers, 64 VGPRs at H = 128)
// and writes dpre_t (bf16) to HBM.  Everything that is parallel over time is deferred to full-chip GEMMs over the stored
// sequence: dx = dpre W_ih (dx_gemm_bf16_kernel below) and the weight gradients (gemm_tn_bf16.hip).
// Measured motivation (round 1, ablation): with dx in the loop the kernel sat at 235 VGPRs, could prefetch its saved state
// only ONE step ahead, and spent ~55 % of its time waiting for those loads (vmcnt retires in order, behind older stores).
// With dx out, the freed registers hold a 3-step-deep rotating prefetch (loop unrolled x3 so every register index is
// static and the compiler emits counted vmcnt waits).
//
// HBM layouts (bf16 path, chosen so that every per-step access of a workgroup is one contiguous, fully coalesced block):
//   dpre_seq, h_seq : tile-blocked row-major, row r' = (tile*T + t)*WPT + j  (j = window within the WPT = 4 window tile);
//                     dpre_t leaves through the LDS operand image as 16-byte stores (a second ablation showed the
//                     previous per-lane 8-byte stores into 16 different rows cost 45 % of the kernel);
//   dx_seq          : MFMA fragment order [r'/16][wave][lane][4] — written by dx_gemm;
//   saved records   : one 12-byte record per cell and step (six bf16), fragment order (written by the forward).
// Quad tiles / one cell per lane: see lstm_fwd_bf16.hip.  After the third ablation (per-CU HBM load bandwidth, ~25 GB/s, was
// the limiter with 16-window tiles on 32 CUs) a workgroup owns 4 windows: 128 CUs, 20 B of loads per lane and step.
//
// Layer pipelining (lstm_bwd_pair_bf16_kernel, 2-layer stacks): ONE launch of 2 x tiles workgroups.  The top layer's
// workgroup of tile i writes dpre^1_t write-through (sc1) and publishes its progress DLY steps late (counted vmcnt, see
// lstm_fwd_bf16.hip); the layer-0 workgroup of tile i follows a few steps behind and forms the gradient on its own h_t
//     dx_t = dpre^1_t W_ih^1   (inter-layer dropout mask re-applied)
// itself, four steps at a time (16 MFMA rows = 4 steps x 4 windows, W_ih^1T fragments in registers) from sc1 loads of the
// top layer's dpre rows: the dx sequence is never written to HBM, and the two recurrences overlap on 2 x 128 CUs.  Roles
// come from a ticket, so a consumer only ever waits for a workgroup that is already running.
#include "kernels.hpp"

namespace shm {

namespace {

constexpr int WPT = SHM_WPT;     // windows per workgroup (see lstm_fwd_bf16.hip: quad tiles, one cell per lane)
constexpr int DLY = 4;           // pair launch: a step's dpre rows are published DLY steps after they were stored
constexpr int PAIR_GROUP = 128;  // pair launch: tiles per ticket group
constexpr unsigned SPIN_MAX = 1u << 21;

typedef __attribute__((address_space(1))) unsigned gu32;
#define SHM_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
template <int V> struct IntC { static constexpr int value = V; };
__device__ __forceinline__ float redeal1(const f32x4 a) {
    int v = __float_as_int(a[0]);
    v = __builtin_amdgcn_update_dpp(v, __float_as_int(a[1]), 0x114, 0xF, 0x2, false);   // row_shr:4  -> lanes 4..7
    v = __builtin_amdgcn_update_dpp(v, __float_as_int(a[2]), 0x118, 0xF, 0x4, false);   // row_shr:8  -> lanes 8..11
    v = __builtin_amdgcn_update_dpp(v, __float_as_int(a[3]), 0x11C, 0xF, 0x8, false);   // row_shr:12 -> lanes 12..15
    return __int_as_float(v);
}

struct Prefetch {            // operands of one time step of this lane's cell, fetched PD = 3 steps ahead
    u32x3 rec;               // {i,f | g,o | c_{t-1}, tanh(c_t)} (bf16x2 each) written by the forward
    f32x4 up;                // DH == 1: up[0] = dx of this cell;  DH == 2: dy[b,t,0..3] (raw, multiplied by W_out at use)
};

// DH: where the gradient on h_t comes from besides the recurrence — 0: nothing (encoder top layer: only h_T, folded into
// the initial dh_rec), 1: dx of the layer above (fragment order of dx_gemm), 2: output layer with exactly 4 channels,
// dy[B,T,4] x W_out (top decoder layer at the north-star shape; raw dy is prefetched, the product is formed at use),
// 3: output layer with any other channel count (dy.W_out evaluated inside the fetch: synchronous fallback).
// The time loop is branch-free: loads are clamped, never predicated, and T % 3 steps are peeled, so that the compiler can
// count the loads in flight across the back-edge instead of draining them (s_waitcnt vmcnt(0)) at every loop header.
// PRODUCE: dpre_seq is consumed by the layer-0 workgroup of the same tile inside this launch (flag_out).
template <int H, int DH, bool PRODUCE>
__device__ __forceinline__ void lstm_bwd_body(const LstmBwdArgsB& a, const int tile, gu32* flag_out) {
    constexpr int NW = H / 16, NTHR = 64 * NW;
    constexpr int KS = H / 8;                      // k-steps of 32 over K = 4H
    constexpr int RS = lds_row_stride(H);          // quarter q <-> gate q, KQ = H
    constexpr int PBUF = 4 * 16 * RS;
    extern __shared__ __attribute__((aligned(16))) bf16_t smem_b[];
    bf16_t* dp_lds = smem_b;                                               // [2][PBUF]
    float* scr = reinterpret_cast<float*>(smem_b + 2 * PBUF);              // [NW][64] accumulator re-deal scratch
    float* wo_lds = scr + NW * 64;                                         // [Dy][H]

    const int T = a.T, B = a.B, Dy = a.Dy;
    const int tid = threadIdx.x, lane = tid & 63, ht = tid >> 6, q = lane >> 4, j = lane & 15;
    const int cj = j & 3, cr = j >> 2, cu = ht * 16 + q * 4 + cr;          // my cell: window cj of the tile, unit cu
    const int cb = tile * WPT + cj;
    const bool cvalid = cb < B;
    const int bc = cvalid ? cb : B - 1;            // clamped window for loads; its contribution is zeroed through vmask
    const float vmask = cvalid ? 1.f : 0.f;
    // vector-memory instructions a dpre-storing wave issues per step (record, upstream gradient, dpre), a lower bound
    constexpr int OPS = 2 + ((DH == 1 || DH == 2) ? 1 : 0);
    const __amdgpu_buffer_rsrc_t out_rsrc = make_rsrc(PRODUCE ? a.dpre_seq + (size_t)tile * T * WPT * 4 * H : nullptr,
                                                      PRODUCE ? (uint32_t)T * WPT * 4 * H * 2 : 0);

    for (int e = tid; e < 2 * PBUF; e += NTHR) smem_b[e] = (bf16_t)0.f;   // image rows j >= WPT stay zero
    if (DH >= 2)
        for (int e = tid; e < Dy * H; e += NTHR) wo_lds[e] = a.Wout[e];

    bf16x8 whT[KS];
    {
        const bf16x8* p = reinterpret_cast<const bf16x8*>(a.WhT) + (size_t)ht * KS * 64 + lane;
#pragma unroll
        for (int s = 0; s < KS; ++s) whT[s] = p[s * 64];
    }

    const u32x3* sv = reinterpret_cast<const u32x3*>(a.sv_g) + ((size_t)tile * T * NW + ht) * 64 + lane;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    auto fetch = [&](Prefetch& P, int t) {
        const int tc = t < 0 ? 0 : t;                                  // clamp: steps < 0 are fetched but never used
        P.rec = sv[(size_t)tc * NW * 64];
        const size_t row = (size_t)bc * T + tc;                        // natural row of dy [B,T,Dy]
        if constexpr (DH == 1) {
            // dx_gemm writes 16-row fragments: row r' = (tile*T + t)*WPT + cj sits in fragment r'/16 at lane (q, r'%16), element cr
            const size_t rp = ((size_t)tile * T + tc) * WPT + cj;
            P.up = zero4;
            P.up[0] = a.dh_seq[(((rp >> 4) * NW + ht) * 64 + q * 16 + (rp & 15)) * 4 + cr];
        } else if constexpr (DH == 2) P.up = *reinterpret_cast<const f32x4*>(a.dy + row * 4);       // Dy == 4 exactly
        else if constexpr (DH == 3) {
            float d = 0.f;
            for (int dd = 0; dd < Dy; ++dd) d = fmaf(a.dy[row * Dy + dd], wo_lds[dd * H + cu], d);
            P.up = zero4; P.up[0] = d;
        } else P.up = zero4;
    };

    float dc = 0.f, dh_rec = 0.f;
    float tsum[4] = {0.f, 0.f, 0.f, 0.f};

    // one time step; P holds this step's operands and is refilled with step t-3 once they are consumed
    auto step = [&](Prefetch& P, int t) {
        // Pin the consumption point of this slot: without it the compiler hoists the unpacking of the LATER slots into this
        // step, i.e. touches registers whose loads are still in flight and turns the counted waits into a drain.
        asm volatile("" : "+v"(P.rec.v[0]), "+v"(P.rec.v[1]), "+v"(P.rec.v[2]), "+v"(P.up));
        bf16_t* dpl = dp_lds + (t & 1) * PBUF;
        float dha;
        if constexpr (DH == 2) {
            dha = 0.f;
#pragma unroll
            for (int dd = 0; dd < 4; ++dd) dha = fmaf(P.up[dd], wo_lds[dd * H + cu], dha);
        } else dha = P.up[0];
        // bf16 -> f32 is a 16-bit shift; done on the integer words (a bf16x2 bit_cast of two different words was folded
        // into one by the compiler: both pairs came out as (i, f))
        const float s_i = __uint_as_float(P.rec[0] << 16), s_f = __uint_as_float(P.rec[0] & 0xffff0000u);
        const float s_g = __uint_as_float(P.rec[1] << 16), s_o = __uint_as_float(P.rec[1] & 0xffff0000u);
        const float c_prev = __uint_as_float(P.rec[2] << 16), tc = __uint_as_float(P.rec[2] & 0xffff0000u);
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
            dpl[(g * 16 + cj) * RS + cu] = (bf16_t)dpre[g];
            tsum[g] += dpre[g];
        }
        fetch(P, t - 4);                                   // refill the slot: 4 steps of latency cover
        __syncthreads();
        // every storing wave has, in step t+1, waited until its dpre store of step t+DLY had retired
        if constexpr (PRODUCE) {
            if (tid == 0 && t + DLY <= T - 1) __hip_atomic_store(flag_out, (unsigned)(T - (t + DLY)), SHM_RLX_AGENT);
        }
        // dpre_t -> HBM out of the now complete image: the WPT x 4H block of (tile, t) is contiguous in dpre_seq
        {
            constexpr int NCH = WPT * 4 * H / 8;           // 16-byte chunks (<= NTHR)
            if (tid < NCH) {
                const int jj = tid / (4 * H / 8), c8 = (tid - jj * (4 * H / 8)) * 8, g = c8 / H, u = c8 - g * H;
                if constexpr (PRODUCE) {
                    const u32x4 v = *reinterpret_cast<const u32x4*>(dpl + (g * 16 + jj) * RS + u);
                    __builtin_amdgcn_raw_buffer_store_b128(v, out_rsrc, (uint32_t)(t * WPT * 4 * H + tid * 8) * 2, 0, 16);
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DLY - 1) * OPS) : "memory");   // retires the store of step t+DLY-1
                } else {
                    *reinterpret_cast<bf16x8*>(a.dpre_seq + ((size_t)tile * T + t) * WPT * 4 * H + (size_t)tid * 8) =
                        *reinterpret_cast<const bf16x8*>(dpl + (g * 16 + jj) * RS + u);
                }
            }
        }
        const bf16_t* brow = dpl + (q * 16 + j) * RS;
        f32x4 acc0 = zero4, acc1 = zero4;
#pragma unroll
        for (int s = 0; s < KS; s += 2) {
            const bf16x8 bv0 = *reinterpret_cast<const bf16x8*>(brow + 8 * s);
            const bf16x8 bv1 = *reinterpret_cast<const bf16x8*>(brow + 8 * s + 8);
            acc0 = mfma_bf16_16x16x32(whT[s], bv0, acc0);
            acc1 = mfma_bf16_16x16x32(whT[s + 1], bv1, acc1);
        }
        // accumulator lanes (q, j<4, r) -> cell lanes (q, jj): register jj>>2 of lane jj&3, by DPP (see lstm_fwd_bf16.hip)
        dh_rec = redeal1(acc0 + acc1);
    };

    __syncthreads();                                       // images zeroed, wo_lds visible
    Prefetch P0, P1, P2, P3;
    if (a.dh_last) dh_rec = vmask * a.dh_last[(size_t)bc * H + cu];    // gradient on h_{T-1}
    int t = T - 1;
    for (int k = T & 3; k > 0; --k, --t) { fetch(P0, t); step(P0, t); }     // peel so that the main loop runs whole quadruples
    fetch(P0, t); fetch(P1, t - 1); fetch(P2, t - 2); fetch(P3, t - 3);
    // Drain once here: the waits inside the loop are static, so anything still in flight from the prologue (weights,
    // peeled steps, these first fetches) would otherwise force a vmcnt(0) into every iteration.
    __builtin_amdgcn_s_waitcnt(0x0F70);                    // vmcnt(0) only
    for (; t >= 3; t -= 4) {
        step(P0, t);
        step(P1, t - 1);
        step(P2, t - 2);
        step(P3, t - 3);
    }
    if constexpr (PRODUCE) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_store(flag_out, (unsigned)T, SHM_RLX_AGENT);
    }
    if (cvalid && a.dpre_tsum) {
#pragma unroll
        for (int g = 0; g < 4; ++g) a.dpre_tsum[(size_t)cb * 4 * H + g * H + cu] = tsum[g];
    }
}

template <int H, int DH>
__global__ void __launch_bounds__(4 * H) lstm_bwd_bf16_kernel(LstmBwdArgsB a) {
    lstm_bwd_body<H, DH, false>(a, blockIdx.x, nullptr);
}

// Layer 0 of a pair launch: BPTT of the layer below `up`, with dx_t = dpre^up_t W_ih^up formed here (see the file header).
// Chunk c = steps 4c..4c+3 of the tile = 16 consecutive rows of up.dpre_seq.  While the recurrence walks chunk c (t descending),
// the rows of chunk c-1 go from registers into the LDS image (sub-step 3) and through the MFMA into dxbuf (sub-step 2), and
// the rows of chunk c-2 are requested: every global access is 4 steps or more ahead of its use.
template <int H, bool DROP>
__device__ __forceinline__ void lstm_bwd_fused_dx_body(const LstmBwdArgsB& a, const LstmBwdArgsB& up, const int tile, gu32* flag_in, gu32* err) {
    static_assert(WPT == 4, "a 16-row MFMA tile = 4 steps x 4 windows");
    constexpr int NW = H / 16, NTHR = 64 * NW, KS = H / 8, RS = lds_row_stride(H), PBUF = 4 * 16 * RS;
    extern __shared__ __attribute__((aligned(16))) bf16_t smem_b[];
    bf16_t* dp_lds = smem_b;                                               // [2][PBUF] this layer's dpre_t
    bf16_t* dx_img = smem_b + 2 * PBUF;                                    // [PBUF] 16 rows of the upper layer's dpre
    float* scr = reinterpret_cast<float*>(smem_b + 3 * PBUF);              // [NW][64] accumulator re-deal scratch
    float* dxbuf = scr + NW * 64;                                          // [2][16 rows][H] dx of two chunks

    const int T = a.T, B = a.B;
    const int tid = threadIdx.x, lane = tid & 63, ht = tid >> 6, q = lane >> 4, j = lane & 15;
    const int cj = j & 3, cr = j >> 2, cu = ht * 16 + q * 4 + cr;
    const int cb = tile * WPT + cj;
    const bool cvalid = cb < B;
    const float vmask = cvalid ? 1.f : 0.f;

    for (int e = tid; e < 2 * PBUF; e += NTHR) smem_b[e] = (bf16_t)0.f;   // image rows j >= WPT stay zero

    bf16x8 whT[KS], wxT[KS];
    {
        const bf16x8* p = reinterpret_cast<const bf16x8*>(a.WhT) + (size_t)ht * KS * 64 + lane;
        const bf16x8* px = reinterpret_cast<const bf16x8*>(up.WxT) + (size_t)ht * KS * 64 + lane;
#pragma unroll
        for (int s = 0; s < KS; ++s) { whT[s] = p[s * 64]; wxT[s] = px[s * 64]; }
    }
    const uint32_t thr = drop_threshold(up.drop_p);
    const float dscale = DROP ? 1.0f / (1.0f - up.drop_p) : 1.0f;

    const u32x3* sv = reinterpret_cast<const u32x3*>(a.sv_g) + ((size_t)tile * T * NW + ht) * 64 + lane;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    auto fetch = [&](u32x3& R, int t) { R = sv[(size_t)(t < 0 ? 0 : t) * NW * 64]; };

    // ---- the upper layer's dpre rows: sc1 loads behind the producer's progress word
    const __amdgpu_buffer_rsrc_t up_rsrc = make_rsrc(up.dpre_seq + (size_t)tile * T * WPT * 4 * H, (uint32_t)T * WPT * 4 * H * 2);
    int avail = 0;                                         // wave-uniform: the producer's steps >= T - avail are in memory
    const uint64_t flag_in_s = uniform_addr((const void*)flag_in);
    auto wait_for = [&](int need) {
        if (avail >= need) return;
        unsigned v, spins = 0;
#pragma clang loop unroll(disable)
        for (;;) {
            v = poll_uniform(flag_in_s);                      // scalar load: no vmcnt drain (common.hpp)
            if ((int)v >= need) break;
            if (++spins > SPIN_MAX) {                      // give up: finish with garbage, flagged in the err word
                if (lane == 0) __hip_atomic_store(err, 0x80000001u + (unsigned)tile, SHM_RLX_AGENT);
                v = 0x7fffffff;
                break;
            }
            __builtin_amdgcn_s_sleep(8);
        }
        avail = (int)v;
    };
    constexpr int CPT = (16 * 4 * H / 8) / NTHR;           // = 2 16-byte pieces per thread and chunk
    bf16x8 ck[CPT];
    auto chunk_load = [&](int c) {                         // rows of steps >= T read as zero (buffer range check)
        const int cc = c < 0 ? 0 : c;
        wait_for(T - 4 * cc);
#pragma unroll
        for (int i = 0; i < CPT; ++i)
            ck[i] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(up_rsrc, (uint32_t)(cc * 16 * 4 * H + (tid + i * NTHR) * 8) * 2, 0, 16));
    };
    auto chunk_commit = [&]() {
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const int e = tid + i * NTHR, jj = e / (4 * H / 8), c8 = (e - jj * (4 * H / 8)) * 8, g = c8 / H, u = c8 - g * H;
            asm volatile("" : "+v"(ck[i]));
            *reinterpret_cast<bf16x8*>(dx_img + (g * 16 + jj) * RS + u) = ck[i];
        }
    };
    auto chunk_mma = [&](int c) {                          // same contraction order and mask indexing as dx_gemm_bf16_kernel
        const bf16_t* brow = dx_img + (q * 16 + j) * RS;
        f32x4 acc0 = zero4, acc1 = zero4;
#pragma unroll
        for (int s = 0; s < KS; s += 2) {
            const bf16x8 bv0 = *reinterpret_cast<const bf16x8*>(brow + 8 * s);
            const bf16x8 bv1 = *reinterpret_cast<const bf16x8*>(brow + 8 * s + 8);
            acc0 = mfma_bf16_16x16x32(wxT[s], bv0, acc0);
            acc1 = mfma_bf16_16x16x32(wxT[s + 1], bv1, acc1);
        }
        f32x4 d = acc0 + acc1;
        if (DROP) {
            const uint32_t base = (uint32_t)((((size_t)tile * T + 4 * c) * WPT + j) * H + ht * 16 + q * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) d[e] = drop_hash(up.drop_seed, up.drop_stack, base + e) >= thr ? d[e] * dscale : 0.f;
        }
        *reinterpret_cast<f32x4*>(dxbuf + (((c & 1) * 16) + j) * H + ht * 16 + q * 4) = d;
    };

    float dc = 0.f, dh_rec = 0.f;
    float tsum[4] = {0.f, 0.f, 0.f, 0.f};

    auto step = [&](u32x3& R, int t, auto subc) {
        constexpr int SUB = decltype(subc)::value;         // t & 3 in the main loop, -1 in the peeled steps
        asm volatile("" : "+v"(R.v[0]), "+v"(R.v[1]), "+v"(R.v[2]));
        bf16_t* dpl = dp_lds + (t & 1) * PBUF;
        const int c = t >> 2;
        const float dha = dxbuf[(((c & 1) * 16) + (t & 3) * 4 + cj) * H + cu];
        const float s_i = __uint_as_float(R[0] << 16), s_f = __uint_as_float(R[0] & 0xffff0000u);
        const float s_g = __uint_as_float(R[1] << 16), s_o = __uint_as_float(R[1] & 0xffff0000u);
        const float c_prev = __uint_as_float(R[2] << 16), tc = __uint_as_float(R[2] & 0xffff0000u);
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
            dpl[(g * 16 + cj) * RS + cu] = (bf16_t)dpre[g];
            tsum[g] += dpre[g];
        }
        fetch(R, t - 8);                                   // refill the slot: 8 steps (~7 us) of latency cover
        if constexpr (SUB == 3) { chunk_commit(); chunk_load(c - 2); }
        __syncthreads();
        {
            constexpr int NCH = WPT * 4 * H / 8;
            if (tid < NCH) {
                const int jj = tid / (4 * H / 8), c8 = (tid - jj * (4 * H / 8)) * 8, g = c8 / H, u = c8 - g * H;
                *reinterpret_cast<bf16x8*>(a.dpre_seq + ((size_t)tile * T + t) * WPT * 4 * H + (size_t)tid * 8) =
                    *reinterpret_cast<const bf16x8*>(dpl + (g * 16 + jj) * RS + u);
            }
        }
        const bf16_t* brow = dpl + (q * 16 + j) * RS;
        f32x4 acc0 = zero4, acc1 = zero4;
#pragma unroll
        for (int s = 0; s < KS; s += 2) {
            const bf16x8 bv0 = *reinterpret_cast<const bf16x8*>(brow + 8 * s);
            const bf16x8 bv1 = *reinterpret_cast<const bf16x8*>(brow + 8 * s + 8);
            acc0 = mfma_bf16_16x16x32(whT[s], bv0, acc0);
            acc1 = mfma_bf16_16x16x32(whT[s + 1], bv1, acc1);
        }
        if constexpr (SUB == 2) chunk_mma(c - 1);
        dh_rec = redeal1(acc0 + acc1);
    };

    __syncthreads();                                       // images zeroed
    auto sync_chunk = [&](int c) { chunk_load(c); chunk_commit(); __syncthreads(); chunk_mma(c); __syncthreads(); };
    const int ctop = (T - 1) >> 2;
    int c1 = ctop;
    sync_chunk(ctop);
    if (T & 3) { c1 = ctop - 1; if (c1 >= 0) sync_chunk(c1); }      // the peeled steps use the partial top chunk
    chunk_load(c1 - 1);

    // eight record slots = eight steps of look-ahead (two hand-off chunks); an odd chunk at the end runs on the first four slots, which
    // by then hold exactly its steps
    u32x3 R0, R1, R2, R3, R4, R5, R6, R7;
    int t = T - 1;
    for (int k = T & 3; k > 0; --k, --t) { fetch(R0, t); step(R0, t, IntC<-1>{}); }
    fetch(R0, t); fetch(R1, t - 1); fetch(R2, t - 2); fetch(R3, t - 3);
    fetch(R4, t - 4); fetch(R5, t - 5); fetch(R6, t - 6); fetch(R7, t - 7);
    __builtin_amdgcn_s_waitcnt(0x0F70);                    // vmcnt(0) once (see lstm_bwd_body)
    for (; t >= 7; t -= 8) {
        step(R0, t, IntC<3>{});
        step(R1, t - 1, IntC<2>{});
        step(R2, t - 2, IntC<1>{});
        step(R3, t - 3, IntC<0>{});
        step(R4, t - 4, IntC<3>{});
        step(R5, t - 5, IntC<2>{});
        step(R6, t - 6, IntC<1>{});
        step(R7, t - 7, IntC<0>{});
    }
    if (t >= 3) {
        step(R0, t, IntC<3>{});
        step(R1, t - 1, IntC<2>{});
        step(R2, t - 2, IntC<1>{});
        step(R3, t - 3, IntC<0>{});
    }
    if (cvalid && a.dpre_tsum) {
#pragma unroll
        for (int g = 0; g < 4; ++g) a.dpre_tsum[(size_t)cb * 4 * H + g * H + cu] = tsum[g];
    }
}

// sync: [0] ticket, flag of tile i at sync[16 * (1 + i)], all zero at launch (see lstm_fwd_bf16.hip); err: sticky word outside
template <int H, int DHTOP, bool DROP>
__global__ void __launch_bounds__(4 * H) lstm_bwd_pair_bf16_kernel(LstmBwdPairArgsB p) {
    extern __shared__ __attribute__((aligned(16))) bf16_t smem_b[];
    gu32* sync = (gu32*)p.sync;
    const int tiles = gridDim.x >> 1;
    int* role_lds = reinterpret_cast<int*>(reinterpret_cast<char*>(smem_b) + p.role_lds_off);
    if (threadIdx.x == 0) *role_lds = (int)__hip_atomic_fetch_add(sync, 1u, SHM_RLX_AGENT);
    __syncthreads();
    const int ticket = *role_lds;
    __syncthreads();
    // ticket groups of PAIR_GROUP tiles: producers first, then their consumers (see lstm_fwd_bf16.hip)
    const int grp = ticket / (2 * PAIR_GROUP), r = ticket - grp * 2 * PAIR_GROUP, base = grp * PAIR_GROUP;
    const int ntg = min(PAIR_GROUP, tiles - base);
    const bool producer = r < ntg;
    const int tile = base + (producer ? r : r - ntg);
    if (threadIdx.x == 0 && producer) {                   // nobody touches the other block in this launch
        p.sync_other[16 * (1 + tile)] = 0u;
        if (ticket == 0) p.sync_other[0] = 0u;
    }
    if (producer) lstm_bwd_body<H, DHTOP, true>(p.l1, tile, sync + 16 * (1 + tile));
    else lstm_bwd_fused_dx_body<H, DROP>(p.l0, p.l1, tile, sync + 16 * (1 + tile), (gu32*)p.err);
}

// ---------------------------------------------------------------------------------------------------------------------
// dx[r][k] = sum_m dpre[r][m] W_ih[m][k]  for all r = (window, step) at once — the time-parallel half of BPTT.
// Workgroup = 8 (H/16) waves; wave w owns output columns [16w, 16w+16) with its W_ih^T fragments in registers and walks
// ROWS_PER_WG rows in 16-row tiles staged through LDS (same operand image and MFMA mapping as the recurrent kernel, without
// the recurrence).  dropout: the inter-layer mask of the layer input is re-applied to dx (nn.LSTM dropout backward).
template <int H, bool DROP>
__global__ void __launch_bounds__(4 * H) dx_gemm_bf16_kernel(const bf16_t* __restrict__ dpre, const bf16_t* __restrict__ WxT,
                                                             float* __restrict__ dx, int R, int tiles_per_wg, float drop_p,
                                                             uint64_t drop_seed, uint32_t drop_stack) {
    constexpr int NW = H / 16, NTHR = 64 * NW, KS = H / 8, RS = lds_row_stride(H), PBUF = 4 * 16 * RS;
    __shared__ __attribute__((aligned(16))) bf16_t img[2 * PBUF];
    const int tid = threadIdx.x, lane = tid & 63, ht = tid >> 6, q = lane >> 4, j = lane & 15;
    bf16x8 w[KS];
    {
        const bf16x8* p = reinterpret_cast<const bf16x8*>(WxT) + (size_t)ht * KS * 64 + lane;
#pragma unroll
        for (int s = 0; s < KS; ++s) w[s] = p[s * 64];
    }
    const uint32_t thr = drop_threshold(drop_p);
    const float dscale = DROP ? 1.0f / (1.0f - drop_p) : 1.0f;
    const int n_tiles = (R + 15) / 16;               // R = ceil(B/WPT)*WPT*T rows in tile-blocked order, 16 rows per MFMA fragment
    const int t0 = blockIdx.x * tiles_per_wg, t1 = min(n_tiles, t0 + tiles_per_wg);
    // staging: a 16-row tile is 16 x 4H bf16 = 16*4H/8 chunks of 16 B; chunk c of row jj -> image slot (gate, jj, unit)
    constexpr int CHUNKS = 16 * 4 * H / 8, CPT = CHUNKS / NTHR;          // = 2 chunks per thread (any H)
    bf16x8 st[CPT];
    auto load_tile = [&](int tl) {
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const int e = tid + i * NTHR, jj = e / (4 * H / 8), c8 = (e - jj * (4 * H / 8)) * 8;
            const int r = tl * 16 + jj;
            st[i] = r < R ? *reinterpret_cast<const bf16x8*>(dpre + (size_t)r * 4 * H + c8)
                          : bf16x8{(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const int e = tid + i * NTHR, jj = e / (4 * H / 8), c8 = (e - jj * (4 * H / 8)) * 8;
            const int g = c8 / H, u = c8 - g * H;
            *reinterpret_cast<bf16x8*>(img + buf * PBUF + (g * 16 + jj) * RS + u) = st[i];
        }
    };
    if (t0 < t1) { load_tile(t0); store_tile(0); }
    __syncthreads();
    for (int tl = t0; tl < t1; ++tl) {
        const int cur = (tl - t0) & 1;
        if (tl + 1 < t1) load_tile(tl + 1);
        const bf16_t* brow = img + cur * PBUF + (q * 16 + j) * RS;
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KS; s += 2) {
            const bf16x8 bv0 = *reinterpret_cast<const bf16x8*>(brow + 8 * s);
            const bf16x8 bv1 = *reinterpret_cast<const bf16x8*>(brow + 8 * s + 8);
            acc0 = mfma_bf16_16x16x32(w[s], bv0, acc0);
            acc1 = mfma_bf16_16x16x32(w[s + 1], bv1, acc1);
        }
        f32x4 d = acc0 + acc1;
        if (DROP) {                                  // same element index as the forward's staging of the layer input
            const uint32_t base = (uint32_t)(((size_t)tl * 16 + j) * H + ht * 16 + q * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) d[e] = drop_hash(drop_seed, drop_stack, base + e) >= thr ? d[e] * dscale : 0.f;
        }
        *reinterpret_cast<f32x4*>(dx + (((size_t)tl * NW + ht) * 64 + lane) * 4) = d;      // fragment order
        if (tl + 1 < t1) store_tile(cur ^ 1);
        __syncthreads();
    }
}

template <int H>
int launch_h(const LstmBwdArgsB& a, hipStream_t s) {
    const int tiles = ceil_div(a.B, WPT);
    const size_t lds = sizeof(bf16_t) * 2 * 4 * 16 * lds_row_stride(H) + sizeof(float) * ((H / 16) * 64 + (a.dy ? a.Dy * H : 0));
    void (*kern)(LstmBwdArgsB) = a.dy ? (a.Dy == 4 ? lstm_bwd_bf16_kernel<H, 2> : lstm_bwd_bf16_kernel<H, 3>)
                                 : a.dh_seq ? lstm_bwd_bf16_kernel<H, 1> : lstm_bwd_bf16_kernel<H, 0>;
    SHM_LAUNCH(kern, dim3(tiles), dim3(4 * H), lds, s, a);
    SHM_LAUNCH_CHECK();
    return SHM_OK;
}

template <int H>
int launch_dx(const LstmBwdArgsB& a, hipStream_t s) {
    const int R = ceil_div(a.B, WPT) * WPT * a.T, n_tiles = ceil_div(R, 16);
    const int tiles_per_wg = n_tiles >= 2048 ? 8 : 1;            // 1024 workgroups at the north-star shape
    const int grid = ceil_div(n_tiles, tiles_per_wg);
    if (a.drop_p > 0.f)
        SHM_LAUNCH((dx_gemm_bf16_kernel<H, true>), dim3(grid), dim3(4 * H), 0, s, a.dpre_seq, a.WxT, a.dx_seq, R, tiles_per_wg,
                           a.drop_p, a.drop_seed, a.drop_stack);
    else
        SHM_LAUNCH((dx_gemm_bf16_kernel<H, false>), dim3(grid), dim3(4 * H), 0, s, a.dpre_seq, a.WxT, a.dx_seq, R, tiles_per_wg,
                           0.f, a.drop_seed, a.drop_stack);
    SHM_LAUNCH_CHECK();
    return SHM_OK;
}

template <int H>
int launch_pair(const LstmBwdPairArgsB& p0, hipStream_t s) {
    LstmBwdPairArgsB p = p0;
    const LstmBwdArgsB &t = p.l1, &b = p.l0;
    SHM_REQUIRE(t.sv_g && b.sv_g && t.dpre_seq && b.dpre_seq && t.dpre_seq != b.dpre_seq, SHM_ESTATE, "lstm_bwd_pair: saved state / two dpre buffers needed");
    SHM_REQUIRE(t.WhT && b.WhT && t.WxT && t.In == H, SHM_EINVAL, "lstm_bwd_pair: packed weights missing");
    SHM_REQUIRE(t.B == b.B && t.T == b.T && p.sync && p.sync_other && p.sync != p.sync_other && p.err, SHM_EINVAL, "lstm_bwd_pair: shape mismatch or no sync block");
    SHM_REQUIRE(!t.dy || (t.Wout && t.Dy > 0 && t.Dy <= 64), SHM_EINVAL, "lstm_bwd_pair: bad output-layer gradient");
    SHM_REQUIRE((int64_t)t.T * WPT * 4 * H * 2 < ((int64_t)1 << 31), SHM_EINVAL, "lstm_bwd_pair: T too long for 32-bit buffer offsets");
    const int tiles = ceil_div(t.B, WPT);
    const size_t img = sizeof(bf16_t) * 4 * 16 * lds_row_stride(H);
    const size_t lds_top = 2 * img + sizeof(float) * ((H / 16) * 64 + (t.dy ? t.Dy * H : 0));
    const size_t lds_bot = 3 * img + sizeof(float) * ((H / 16) * 64 + 2 * 16 * H);
    const size_t lds = ((lds_top > lds_bot ? lds_top : lds_bot) + 15) & ~(size_t)15;
    p.role_lds_off = (int)lds;
    const bool drop = t.drop_p > 0.f;
    void (*kern)(LstmBwdPairArgsB);
    if (t.dy) kern = t.Dy == 4 ? (drop ? lstm_bwd_pair_bf16_kernel<H, 2, true> : lstm_bwd_pair_bf16_kernel<H, 2, false>)
                               : (drop ? lstm_bwd_pair_bf16_kernel<H, 3, true> : lstm_bwd_pair_bf16_kernel<H, 3, false>);
    else kern = drop ? lstm_bwd_pair_bf16_kernel<H, 0, true> : lstm_bwd_pair_bf16_kernel<H, 0, false>;
    SHM_LAUNCH(kern, dim3(2 * tiles), dim3(4 * H), lds + 16, s, p);
    SHM_LAUNCH_CHECK();
    return SHM_OK;
}

}  // namespace

int launch_lstm_bwd_pair_bf16(int H, const LstmBwdPairArgsB& p, hipStream_t s) {
    SHM_REQUIRE(p.l1.B > 0 && p.l1.T > 0, SHM_EINVAL, "lstm_bwd_pair: empty shape");
    switch (H) {
        case 32: return launch_pair<32>(p, s);
        case 64: return launch_pair<64>(p, s);
        case 128: return launch_pair<128>(p, s);
        default: set_error("lstm_bwd_pair: hidden_dim %d unsupported (32, 64, 128)", H); return SHM_EINVAL;
    }
}

int launch_lstm_bwd_bf16(int H, const LstmBwdArgsB& a, hipStream_t s) {
    SHM_REQUIRE(a.B > 0 && a.T > 0, SHM_EINVAL, "lstm_bwd_bf16: empty shape");
    SHM_REQUIRE(a.sv_g && a.dpre_seq, SHM_ESTATE, "lstm_bwd_bf16: no saved forward state / dpre buffer");
    SHM_REQUIRE(!a.dy || (a.Wout && a.Dy > 0 && a.Dy <= 64), SHM_EINVAL, "lstm_bwd_bf16: bad output-layer gradient");
    switch (H) {
        case 32: return launch_h<32>(a, s);
        case 64: return launch_h<64>(a, s);
        case 128: return launch_h<128>(a, s);
        default: set_error("lstm_bwd_bf16: hidden_dim %d unsupported (32, 64, 128)", H); return SHM_EINVAL;
    }
}

// dx_seq[B*T, H] = dpre_seq[B*T, 4H] . W_ih (packed W_ih^T), inter-layer dropout mask re-applied; call after the recurrent kernel
int launch_dx_gemm_bf16(int H, const LstmBwdArgsB& a, hipStream_t s) {
    SHM_REQUIRE(a.WxT && a.dx_seq && a.dpre_seq && a.In == H, SHM_EINVAL, "dx_gemm_bf16: needs In == H and its operands");
    switch (H) {
        case 32: return launch_dx<32>(a, s);
        case 64: return launch_dx<64>(a, s);
        case 128: return launch_dx<128>(a, s);
        default: set_error("dx_gemm_bf16: hidden_dim %d unsupported (32, 64, 128)", H); return SHM_EINVAL;
    }
}

}  // namespace shm
