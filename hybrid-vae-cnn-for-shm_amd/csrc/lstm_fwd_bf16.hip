// lstm_fwd_bf16.hip — persistent LSTM layer forward, bf16 MFMA (v_mfma_f32_16x16x32_bf16), fp32 accumulate and
// fp32 cell state; gfx950.  Throughput path behind TemporalVAE.encode/decode (reference
// 4DOF/Scripts/Models/temporal_vae.py:53,69: the nn.LSTM cell loop).
//
// Decomposition ("quad tiles"): one workgroup = WPT = 4 windows x all T steps, so B = 512 windows spread over 128 CUs
// (16-window tiles would use 32).  The recurrence is latency-bound and, per CU, bound by how many HBM bytes a single CU can
// keep in flight (~25 GB/s), so the design minimises per-CU work and bytes rather than MFMA padding:
//   * wave w owns hidden units [16w,16w+16) x gates i,f,g,o; W_hh and W_ih fragments live in REGISTERS for the whole
//     sequence (64 VGPRs per matrix per wave at H = 128); the MFMA computes gates^T[64 x 16] = W . [h|x]^T with the 4 windows
//     in columns 0..3 of the B operand (columns 4..15 are zero rows of the LDS image);
//   * the accumulator (lane (q,j) = units 4q..4q+3 of window j, only j < 4 meaningful) is re-dealt with DPP row shifts so
//     that EVERY lane owns exactly one cell: lane (q,jj) = unit 4q + (jj>>2) of window jj&3.  The cell update (10
//     transcendental instructions) then costs a quarter of the VALU issue of the 16-window mapping, and c_t is one register;
//   * saved state for BPTT is ONE 12-byte record per lane per step {i,f | g,o | c_{t-1}, tanh(c_t)} (six bf16) in
//     fragment order (1 KiB contiguous per wave), stored at the start of the NEXT step, behind the MFMA issue;
//   * h_t returns to the LDS operand image as bf16; h_seq is tile-blocked row-major (row r' = (tile*T + t)*4 + j) and is
//     copied out of the completed image one step later as 16-byte coalesced stores;
//   * the layer input of a bf16 layer is staged ONE ELEMENT PER THREAD (4H inputs per step = 4H threads; dropout mask by the
//     counter hash of that element) — staging 16-byte pieces in wave 0 made that wave the straggler of every barrier;
//   * (r3) the input projection of a layer with a time-varying input is formed FOUR STEPS PER MFMA TILE: the 16 columns of the B
//     operand are 4 steps x 4 windows (row 4s + w of a group's x image), so a group of four steps costs 4 x XK input MFMAs instead
//     of 16 x XK (measured: every MFMA per step costs ~2 us of a 256-step launch, hidden or not).  The recurrent product then has to
//     land in columns 4s..4s+3 at step s = t & 3: the h image of step t keeps its windows in rows 4s..4s+3 (one image per s, the
//     other rows stay zero) and the re-deal takes its source lanes from that column group.  Per output element the MFMA sequence is
//     what it was (bias, W_ih k-steps, W_hh k-steps): bit-identical results.  The group's MFMAs are issued in its last step,
//     hand-interleaved with the cell update's vector instructions (sched_barrier-pinned); x_{t+5} is committed and x_{t+9}
//     requested at step t (four rotating register slots, loop unrolled by four, counted s_waitcnt vmcnt);
//   * (r3) everything a step does for ANOTHER step (record of t-1, masked input of t+5) is stored through range-checked buffer
//     descriptors, so the time loop has no branch around it and the scheduler places that vector work BETWEEN the recurrent MFMAs
//     (a wave issues in order: behind the block it waits for the matrix pipe); h_{t-1} is copied out behind the MFMA issue, not
//     in front of it (the copying wave used to start its matrix work an LDS round trip late, at every step).
// Ablation of one layer-1 step (round 1): recurrent MFMAs 21 %, input MFMAs 10 %, accumulator re-deal through LDS 16 %
// (-> DPP), input staging + dropout hashing in wave 0 9 % (-> per thread), cell update + barrier + LDS operand round trip 45 %.
//
// Layer pipelining (lstm_fwd_pair_bf16_kernel): a 2-layer stack runs as ONE launch of 2 x tiles workgroups.  Layer 1 of
// tile i consumes h^0_t of tile i a few steps behind the layer-0 workgroup that produces it, so both layers' recurrences
// overlap on 2 x 128 CUs instead of running back to back on 128.  The hand-off is the sc1 form of the CDNA4 guide
// (cdna_hip_programming.md §6 Guideline 16, R1): every h_seq byte is stored write-through (buffer_store ... sc1) by wave 0,
// which retires the stores of step t-DLY with a COUNTED s_waitcnt (vmcnt completes in order, so waiting until at most
// (DLY-1) x ops-per-step are outstanding retires everything DLY steps old) and then publishes "steps < n are in memory"
// with one sc1 flag store; the consumer polls that word with sc1 loads and reads h_seq ONLY with sc1 buffer loads.
// Roles come from a ticket (atomicAdd), not from blockIdx: a consumer only ever waits for a workgroup that drew a smaller
// ticket, i.e. one that is already running, so the launch cannot deadlock whatever the dispatch order or residency.
#include "kernels.hpp"

namespace shm {

namespace {

constexpr int WPT = SHM_WPT;     // windows per workgroup
constexpr int DLY = 4;           // pair launch: a step's h_seq rows are published DLY steps after they were stored
constexpr int PAIR_GROUP = 128;  // pair launch: tiles per ticket group (two workgroups per tile -> 256 CUs)
constexpr unsigned SPIN_MAX = 1u << 21;   // bounded consumer spin (a safety net: ~1 s); on expiry the err word is set

template <int V> struct IntC { static constexpr int value = V; };
typedef __attribute__((address_space(1))) unsigned gu32;
#define SHM_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}

// XK: k-steps (of 32) of the input operand.  XF32: dynamic input is fp32 [B,T,In] with In <= 16 (encoder layer 0).
// CONSTIN: the input is one fp32 vector per window (decoder layer 0).  SAVE: keep the BPTT records.
// CONSUME: the input sequence is being produced by another workgroup of this launch (flag_in).  PRODUCE: h_seq is
// consumed by another workgroup of this launch (flag_out).
template <int H, int XK, bool XF32, bool DROP, bool CONSTIN, bool SAVE, bool CONSUME, bool PRODUCE>
__device__ __forceinline__ void lstm_fwd_body(const LstmFwdArgsB& a, const int tile, gu32* flag_in, gu32* flag_out, gu32* err) {
    static_assert(!CONSUME || (!XF32 && !CONSTIN), "a consumer reads the bf16 h_seq of the layer below");
    constexpr int NW = H / 16, NTHR = 64 * NW;
    constexpr int KSH = H / 32;                 // k-steps of the recurrent operand
    constexpr int KQH = H / 4, RSH = lds_row_stride(KQH);
    constexpr int KQX = XK * 8, RSX = lds_row_stride(KQX);
    constexpr int HBUF = 4 * 16 * RSH, XBUF = 4 * 16 * RSX;       // bf16 elements per operand image
    // GRP (every layer with a time-varying input): the input projection is formed FOUR STEPS PER MFMA TILE — the 16 columns of the B operand
    // are 4 steps x 4 windows (row 4s + w of the x image of a group), so a group costs 4*XK input MFMAs instead of 16*XK.  The
    // recurrent product must then land in columns 4s..4s+3 at step s = t & 3: the h image of step t keeps its 4 windows in rows
    // 4s..4s+3 (one image per s, so the other rows of an image are never written and stay zero), and the re-deal takes its
    // source lanes from that column group.  Per output element the MFMA sequence is unchanged (bias, W_ih k-steps, W_hh k-steps).
    constexpr bool GRP = !CONSTIN;
    constexpr int NHB = GRP ? 4 : 2;
    extern __shared__ __attribute__((aligned(16))) bf16_t smem_b[];
    bf16_t* h_lds = smem_b;                     // [NHB][HBUF]
    bf16_t* x_lds = smem_b + NHB * HBUF;        // [2][XBUF]

    const int In = a.In, T = a.T, B = a.B;
    const int tid = threadIdx.x, lane = tid & 63, ht = tid >> 6, q = lane >> 4, j = lane & 15;
    // cell owned by this lane after the re-deal
    const int cj = j & 3, cr = j >> 2, cu = ht * 16 + q * 4 + cr;      // window within the tile, unit
    const int cb = tile * WPT + cj;
    const bool cvalid = cb < B;

    for (int e = tid; e < NHB * HBUF + 2 * XBUF; e += NTHR) smem_b[e] = (bf16_t)0.f;

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
    for (int g = 0; g < 4; ++g) bias4[g] = *reinterpret_cast<const f32x4*>(a.bias + g * H + ht * 16 + q * 4);

    const uint32_t thr = drop_threshold(a.drop_p);
    const float dscale = DROP ? 1.0f / (1.0f - a.drop_p) : 1.0f;

    // operand-image slot of my cell (window cj, unit cu)
    const int my_h_off = ((cu / KQH) * 16 + cj) * RSH + (cu % KQH);
    const int my_x_off = ((cu / KQX) * 16 + cj) * RSX + (cu % KQX);      // CONSTIN staging (In == H)

    // ---- block copies between HBM and the images: the WPT x H block of a step is contiguous; thread c < WPT*H/8 moves the
    // 16-byte chunk (row jj, cols c8..c8+7)
    constexpr int NCH = WPT * H / 8;
    const int xc = tid < NCH ? tid : 0, xjj = xc / (H / 8), xc8 = (xc - xjj * (H / 8)) * 8;
    const int h_img_off = ((xc8 / KQH) * 16 + xjj) * RSH + (xc8 % KQH);

    struct XSlot { unsigned e; float s0; };                // e: the bf16 element in the low half (plain load) or the dword holding it (consumer)
    // bf16 input (layers >= 1): the WPT x H block of a step has exactly NTHR elements, thread <-> element E = tid = jj*H + unit.
    // Every thread moves (and, with dropout, masks) ITS element; staging 16-byte pieces in wave 0 put the masking of all 4H
    // inputs (8 hashes + selects per lane) on one wave's path and every other wave waited for it at the barrier.
    const int ejj = tid / H, eu = tid - ejj * H;
    const int e_img_off = ((eu / KQX) * 16 + ejj) * RSX + (eu % KQX);
    // XF32 staging: element e < WPT*In of the tile -> (window jj, channel kk); In <= 16 so one element per thread suffices
    const int xe = tid < WPT * In ? tid : 0, xj0 = xe / In, xk0 = xe - xj0 * In;
    const int xb0 = tile * WPT + xj0 < B ? tile * WPT + xj0 : B - 1;
    // pair launch: the tile's block of the producer's h_seq, read with sc1 (L1-bypassing) buffer loads only
    const uint32_t seq_bytes = (uint32_t)T * WPT * H * 2;
    const __amdgpu_buffer_rsrc_t in_rsrc = make_rsrc(CONSUME ? reinterpret_cast<const bf16_t*>(a.in) + (size_t)tile * T * WPT * H : nullptr, CONSUME ? seq_bytes : 0);
    const __amdgpu_buffer_rsrc_t out_rsrc = make_rsrc(PRODUCE ? a.h_seq + (size_t)tile * T * WPT * H : nullptr, PRODUCE ? seq_bytes : 0);
    // Stores that a step issues for ANOTHER step (the record of step t - 1, the masked input of step t + 5) go through descriptors of the
    // tile's block: for a step outside [0, T) the hardware drops them (range check), so the time loop needs no branch around them — a
    // branch ends the basic block, and the scheduler can then not place this vector work between the recurrent MFMAs.
    const __amdgpu_buffer_rsrc_t xd_rsrc = make_rsrc((DROP && !XF32 && !CONSTIN) ? a.xdrop + (size_t)tile * T * WPT * H : nullptr,
                                                     (DROP && !XF32 && !CONSTIN) ? seq_bytes : 0);
    const __amdgpu_buffer_rsrc_t rec_rsrc = make_rsrc(SAVE ? reinterpret_cast<const char*>(a.sv_g) + (size_t)tile * T * NW * 64 * 12 : nullptr,
                                                      SAVE ? (uint32_t)T * NW * 64 * 12 : 0);
    int avail = 0;                                        // wave-uniform: producer steps < avail are in memory
    const uint64_t flag_in_s = uniform_addr((const void*)flag_in);
    auto wait_for = [&](int need) {
        if (avail >= need) return;
        unsigned v, spins = 0;
#pragma clang loop unroll(disable)
        for (;;) {
            v = poll_uniform(flag_in_s);                      // scalar load: no vmcnt drain (common.hpp)
            if ((int)v >= need) break;
            if (++spins > SPIN_MAX) {                      // give up: finish with garbage, flagged in the err word
                if (lane == 0) __hip_atomic_store(err, 1u + (unsigned)tile, SHM_RLX_AGENT);
                v = 0x7fffffff;
                break;
            }
            __builtin_amdgcn_s_sleep(8);
        }
        avail = (int)v;
    };
    auto x_fetch = [&](XSlot& X, int t) {
        if constexpr (CONSTIN) return;
        const int tc = t < T ? t : T - 1;                  // clamp: steps >= T are fetched but never committed
        if constexpr (XF32) {
            X.s0 = reinterpret_cast<const float*>(a.in)[((size_t)xb0 * T + tc) * In + xk0];
        } else if constexpr (CONSUME) {
            wait_for(tc + 1);
            // 4-byte sc1 loads (the width the guide's hand-off table is measured for); two threads share a dword
            X.e = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(in_rsrc, (uint32_t)(tc * WPT * H + (tid & ~1)) * 2, 0, 16);
        } else {
            X.e = reinterpret_cast<const unsigned short*>(a.in)[((size_t)tile * T + tc) * WPT * H + tid];
        }
    };
    auto x_commit = [&](XSlot& X, int t, int buf) {      // t: the step the slot holds (GRP: the image and row follow from t)
        bf16_t* xl = GRP ? x_lds + ((t >> 2) & 1) * XBUF + (t & 3) * 4 * RSX : x_lds + buf * XBUF;
        if constexpr (XF32) {
            asm volatile("" : "+v"(X.s0));                 // pin the use (see lstm_bwd_bf16.hip)
            if (tid < WPT * In) {
                xl[((xk0 / KQX) * 16 + xj0) * RSX + (xk0 % KQX)] = (bf16_t)X.s0;
                if (a.x_tb) a.x_tb[(((size_t)tile * T + t) * WPT + xj0) * 16 + xk0] = (bf16_t)X.s0;   // operand of the dW_ih GEMM
            }
        } else {
            asm volatile("" : "+v"(X.e));
            float f = CONSUME ? __uint_as_float((tid & 1) ? (X.e & 0xffff0000u) : (X.e << 16)) : __uint_as_float(X.e << 16);
            if (DROP) {     // element index = (tile-blocked row) * H + unit: identical in the backward's dx product and on the host
                const uint32_t idx = (uint32_t)(((size_t)tile * T + t) * WPT * H + tid);
                f = drop_hash(a.drop_seed, a.drop_stack, idx) >= thr ? f * dscale : 0.f;
            }
            const bf16_t v = (bf16_t)f;
            // keep the masked input for the dW_ih GEMM (same tile-blocked layout as the source)
            if (DROP) __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, v), xd_rsrc, (uint32_t)(t * WPT * H + tid) * 2, 0, 0);
            xl[e_img_off] = v;
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
    // accumulator lanes (q, j<4, r=0..3) -> cell lanes (q, jj): one float per gate.  Lane jj of a 16-lane row takes register
    // jj>>2 of lane jj&3: three DPP moves per gate (row_shr:4r written only to lanes 4r..4r+3 through the bank mask) instead of
    // an LDS round trip — measured 0.17 us of every ~1 us step when it went through LDS (ablation, round 1).
    // S: the column group (lanes 4S..4S+3 of each row) that holds the step's windows (0 unless GRP): lane jj takes register r = jj>>2
    // of lane 4S + (jj&3), i.e. row_shr:4(r-S) for r > S, row_shl:4(S-r) for r < S, its own register for r == S.
    auto redeal = [&](auto sc, const f32x4 (&acc)[4], float (&gate)[4]) {
        constexpr int S = decltype(sc)::value;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            int v = __float_as_int(acc[g][S]);
            if constexpr (S != 0) v = __builtin_amdgcn_update_dpp(v, __float_as_int(acc[g][0]), 0x100 + 4 * (S > 0 ? S : 1), 0xF, 0x1, false);
            if constexpr (S < 1) v = __builtin_amdgcn_update_dpp(v, __float_as_int(acc[g][1]), 0x110 + 4 * (S < 1 ? 1 - S : 1), 0xF, 0x2, false);
            if constexpr (S > 1) v = __builtin_amdgcn_update_dpp(v, __float_as_int(acc[g][1]), 0x100 + 4 * (S > 1 ? S - 1 : 1), 0xF, 0x2, false);
            if constexpr (S < 2) v = __builtin_amdgcn_update_dpp(v, __float_as_int(acc[g][2]), 0x110 + 4 * (S < 2 ? 2 - S : 1), 0xF, 0x4, false);
            if constexpr (S > 2) v = __builtin_amdgcn_update_dpp(v, __float_as_int(acc[g][2]), 0x100 + 4 * (S > 2 ? S - 2 : 1), 0xF, 0x4, false);
            if constexpr (S != 3) v = __builtin_amdgcn_update_dpp(v, __float_as_int(acc[g][3]), 0x110 + 4 * (S < 3 ? 3 - S : 1), 0xF, 0x8, false);
            gate[g] = __int_as_float(v);
        }
    };

    __syncthreads();
    // xproj: bias + W_ih x of the step about to run.  Constant input: computed once.  Otherwise it is formed ONE STEP AHEAD,
    // interleaved with the cell update of the running step: the 4*XK input MFMAs do not depend on h_{t-1}, and issued between
    // the transcendental VALU work they run on the matrix pipe while the vector pipe computes sigmoid/tanh (both waves of a
    // SIMD reach this phase together, so without the interleave the matrix pipe idles through every cell update).
    f32x4 xproj[4];
    if constexpr (CONSTIN) {
        // decoder layer 0: the same h0 at every step (temporal_vae.py:68) -> project once, keep in registers
        const float* h0 = reinterpret_cast<const float*>(a.in);
        const float v = cvalid ? h0[(size_t)cb * H + cu] : 0.f;
        x_lds[my_x_off] = (bf16_t)v;
        if (a.h0_bf) a.h0_bf[(size_t)cb * H + cu] = (bf16_t)v;     // whole tiles (padding windows: zeros)
        __syncthreads();
#pragma unroll
        for (int g = 0; g < 4; ++g) xproj[g] = bias4[g];
        mma_x(xproj, x_lds);
    } else if constexpr (GRP) {
        XSlot P0, P1, P2, P3;                              // group 0 (steps 0..3) and x_4, which step 0 no longer has time to commit
        x_fetch(P0, 0); x_fetch(P1, 1); x_fetch(P2, 2); x_fetch(P3, 3);
        x_commit(P0, 0, 0);
        if (T > 1) x_commit(P1, 1, 0);
        if (T > 2) x_commit(P2, 2, 0);
        if (T > 3) x_commit(P3, 3, 0);
        x_fetch(P0, 4);
        if (T > 4) x_commit(P0, 4, 0);
        __syncthreads();
#pragma unroll
        for (int g = 0; g < 4; ++g) xproj[g] = bias4[g];
        mma_x(xproj, x_lds);                               // group 0's input projection
    } else {
        XSlot X;
        x_fetch(X, 0);
        x_commit(X, 0, 0);
        x_fetch(X, 1);
        if (T > 1) x_commit(X, 1, 1);
        __syncthreads();
#pragma unroll
        for (int g = 0; g < 4; ++g) xproj[g] = bias4[g];
        mma_x(xproj, x_lds);                               // step 0's input projection
    }

    float c = 0.f, h = 0.f;
    u32x3* sv = SAVE ? reinterpret_cast<u32x3*>(a.sv_g) + ((size_t)tile * T * NW + ht) * 64 + lane : nullptr;

    // vector-memory instructions wave 0 issues per step besides the flag store (h_seq, record, x prefetch, masked x)
    constexpr int OPS = 1 + (SAVE ? 1 : 0) + (CONSTIN ? 0 : 1) + (DROP ? 1 : 0);
    auto h_flush = [&](int t_done, int buf) {            // GRP: buf = (t_done + 1) & 3 and the windows sit in rows 4 * buf ..
        if (tid < NCH) {
            const bf16_t* src = h_lds + buf * (GRP ? HBUF + 4 * RSH : HBUF) + h_img_off;
            if constexpr (PRODUCE) {
                const u32x4 v = *reinterpret_cast<const u32x4*>(src);
                __builtin_amdgcn_raw_buffer_store_b128(v, out_rsrc, (uint32_t)(t_done * WPT * H + xc * 8) * 2, 0, 16);
                // in-order vmcnt: with at most (DLY-1)*OPS younger operations outstanding, the h_seq store of step
                // t_done - DLY + 1 ... has retired; publish one step more conservatively than that
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DLY - 1) * OPS) : "memory");
                if (tid == 0 && t_done >= DLY) __hip_atomic_store(flag_out, (unsigned)(t_done - DLY + 1), SHM_RLX_AGENT);
            } else {
                *reinterpret_cast<bf16x8*>(a.h_seq + ((size_t)tile * T + t_done) * WPT * H + (size_t)xc * 8) = *reinterpret_cast<const bf16x8*>(src);
            }
        }
    };
    u32x3 rec_prev = {{0u, 0u, 0u}};
    // X: slot holding x_{t+2}; committed at the end of the step (into the buffer of x_t, whose projection step t-1 consumed)
    // and refilled with x_{t+5}
    auto step = [&](auto sc, XSlot& X, int t) {         // sc: t & 3 (GRP) or 0
        constexpr int S = decltype(sc)::value;
        const int cur = GRP ? S : t & 1;                   // GRP: image S holds h_{t-1} in rows 4S.., image (S+1)&3 receives h_t
        constexpr int NXT = (S + 1) & 3;
        f32x4 acc[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = xproj[g];
        // The recurrent MFMAs (they need only h_{t-1}) are issued with the step's h-independent vector work between them: the
        // BPTT record of step t-1 (kept in registers across the barrier) and the staging / masking of x_{t+2}.  A wave issues
        // in order, so vector work placed behind the MFMA block would wait for the matrix pipe, and placed in front of the
        // barrier (where it used to be) it delayed every wave's arrival.
        __builtin_amdgcn_sched_barrier(0);
        mma_h(acc, h_lds + cur * HBUF);
        if constexpr (SAVE) {                              // record of step t - 1 (t == 0: out of range, dropped)
            typedef unsigned u32v3 __attribute__((ext_vector_type(3)));
            const u32v3 r3 = {rec_prev[0], rec_prev[1], rec_prev[2]};
            __builtin_amdgcn_raw_buffer_store_b96(r3, rec_rsrc, (uint32_t)(((t - 1) * NW + ht) * 64 + lane) * 12, 0, 0);
        }
        if constexpr (GRP) {                               // x_{4G+7} must be in the image before step 4G+3 forms group G+1's projection
            if constexpr (XF32) { if (t + 5 < T) x_commit(X, t + 5, 0); }
            else x_commit(X, t + 5, 0);                    // steps >= T: rows nobody reads, stores out of range (dropped)
            x_fetch(X, t + 9);
        } else if constexpr (!CONSTIN) {
            if (t + 2 < T) x_commit(X, t + 2, cur);
            x_fetch(X, t + 5);
        }
#pragma unroll
        for (int i = 0; i < 4 * KSH; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // one MFMA ...
            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);     // ... two vector instructions
        }
        __builtin_amdgcn_sched_barrier(0);
        // h_{t-1} sits complete in h_lds[cur] until the end of the next step: copied out BEHIND the MFMA issue (in front of it, the copying
        // wave started its matrix work an LDS round trip late at every step and the other seven waited for it at the barrier)
        if (t > 0) h_flush(t - 1, cur);

        float gate[4];
        redeal(sc, acc, gate);
        // Cell update, hand-interleaved with next step's input MFMAs (x_{t+1}, buffer cur^1, in LDS since the last barrier):
        // two MFMAs between each group of vector instructions, the order pinned with sched_barrier — left to itself the
        // compiler issues the MFMAs as one block in front of the transcendentals.
        // GRP: the whole projection of the next group is formed in the group's last step (its x rows were committed up to step t - 1),
        // the other three steps carry no input MFMAs
        constexpr bool XM = !CONSTIN && (!GRP || S == 3);
        bf16x8 bx[XK > 0 ? XK : 1];
        if constexpr (XM) {
            const bf16_t* row = x_lds + (GRP ? ((t >> 2) + 1) & 1 : cur ^ 1) * XBUF + (q * 16 + j) * RSX;
#pragma unroll
            for (int sx = 0; sx < XK; ++sx) bx[sx] = *reinterpret_cast<const bf16x8*>(row + 8 * sx);
#pragma unroll
            for (int g = 0; g < 4; ++g) xproj[g] = bias4[g];
        }
        auto xm = [&](auto ic) {                           // input MFMAs number 2i, 2i+1 (k-step major, gates 0..3 inner)
            constexpr int I = decltype(ic)::value;
            if constexpr (XM) {
#pragma unroll
                for (int k = 2 * I; k < 2 * I + 2; ++k)
                    if (k < 4 * XK) xproj[k & 3] = mfma_bf16_16x16x32(wx[k & 3][k >> 2], bx[k >> 2], xproj[k & 3]);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        constexpr float L2E = 1.4426950408889634f;
        __builtin_amdgcn_sched_barrier(0);
        xm(IntC<0>{});
        const float e_i = __builtin_amdgcn_exp2f(-L2E * gate[0]), e_f = __builtin_amdgcn_exp2f(-L2E * gate[1]);
        __builtin_amdgcn_sched_barrier(0);
        xm(IntC<1>{});
        const float e_g = __builtin_amdgcn_exp2f(2.0f * L2E * gate[2]), e_o = __builtin_amdgcn_exp2f(-L2E * gate[3]);
        __builtin_amdgcn_sched_barrier(0);
        xm(IntC<2>{});
        const float gi = __builtin_amdgcn_rcpf(1.0f + e_i), gf = __builtin_amdgcn_rcpf(1.0f + e_f);
        __builtin_amdgcn_sched_barrier(0);
        xm(IntC<3>{});
        const float r_g = __builtin_amdgcn_rcpf(1.0f + e_g), go = __builtin_amdgcn_rcpf(1.0f + e_o);
        __builtin_amdgcn_sched_barrier(0);
        xm(IntC<4>{});
        const float gg = 1.0f - 2.0f * r_g;
        const float c_prev = c;
        c = gf * c + gi * gg;
        const float e_c = __builtin_amdgcn_exp2f(2.0f * L2E * c);
        __builtin_amdgcn_sched_barrier(0);
        xm(IntC<5>{});
        const float r_c = __builtin_amdgcn_rcpf(1.0f + e_c);
        __builtin_amdgcn_sched_barrier(0);
        xm(IntC<6>{});
        const float tc = 1.0f - 2.0f * r_c;
        h = go * tc;
        __builtin_amdgcn_sched_barrier(0);
        xm(IntC<7>{});
        if constexpr (GRP) h_lds[NXT * (HBUF + 4 * RSH) + my_h_off] = (bf16_t)h;
        else h_lds[(cur ^ 1) * HBUF + my_h_off] = (bf16_t)h;
        if constexpr (SAVE) {                              // stored at the start of the next step (or after the loop)
            f32x2 a0 = {gi, gf}, a1 = {gg, go};
            rec_prev[0] = __builtin_bit_cast(unsigned, __builtin_convertvector(a0, bf16x2));
            rec_prev[1] = __builtin_bit_cast(unsigned, __builtin_convertvector(a1, bf16x2));
            f32x2 a2 = {c_prev, tc};
            rec_prev[2] = __builtin_bit_cast(unsigned, __builtin_convertvector(a2, bf16x2));
        }
        __syncthreads();
    };
    int t = 0;
    if constexpr (GRP) {
        XSlot X0, X1, X2, X3;                              // slot t & 3 holds x_{t+5} at the top of step t
        x_fetch(X0, 5); x_fetch(X1, 6); x_fetch(X2, 7); x_fetch(X3, 8);
        __builtin_amdgcn_s_waitcnt(0x0F70);                // vmcnt(0) once: only steady-state loads cross the back-edge
        for (; t + 3 < T; t += 4) {
            step(IntC<0>{}, X0, t);
            step(IntC<1>{}, X1, t + 1);
            step(IntC<2>{}, X2, t + 2);
            step(IntC<3>{}, X3, t + 3);
        }
        if (t < T) { step(IntC<0>{}, X0, t); ++t; }        // T % 4 steps of a last, partial group (its projection exists already)
        if (t < T) { step(IntC<1>{}, X1, t); ++t; }
        if (t < T) { step(IntC<2>{}, X2, t); ++t; }
    } else {
        XSlot X0, X1, X2;
        for (int k = T % 3; k > 0; --k, ++t) { x_fetch(X0, t + 2); step(IntC<0>{}, X0, t); }   // peel: the main loop runs whole triples
        x_fetch(X0, t + 2); x_fetch(X1, t + 3); x_fetch(X2, t + 4);
        __builtin_amdgcn_s_waitcnt(0x0F70);                // vmcnt(0) once: only steady-state loads cross the back-edge
        for (; t + 2 < T; t += 3) {
            step(IntC<0>{}, X0, t);
            step(IntC<0>{}, X1, t + 1);
            step(IntC<0>{}, X2, t + 2);
        }
    }
    if constexpr (SAVE) sv[(size_t)(T - 1) * NW * 64] = rec_prev;
    h_flush(T - 1, GRP ? T & 3 : T & 1);                   // the last image (its barrier has passed)
    if constexpr (PRODUCE) {
        if (tid < NCH) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (tid == 0) __hip_atomic_store(flag_out, (unsigned)T, SHM_RLX_AGENT);
        }
    }
    if (a.h_last && cvalid) a.h_last[(size_t)cb * H + cu] = h;
}

template <int H, int XK, bool XF32, bool DROP, bool CONSTIN, bool SAVE>
__global__ void __launch_bounds__(4 * H) lstm_fwd_bf16_kernel(LstmFwdArgsB a) {
    lstm_fwd_body<H, XK, XF32, DROP, CONSTIN, SAVE, false, false>(a, blockIdx.x, nullptr, nullptr, nullptr);
}

// Two layers, one launch.  sync: [0] ticket, flag of tile i at sync[16 * (1 + i)] (a 64-byte line each), all zero at launch:
// the previous pair launch zeroed it (the plan alternates two blocks; shm_vae_pack_weights zeroes the first); err: a sticky
// word outside those blocks.
template <int H, bool XF32_0, bool DROP, bool SAVE>
__global__ void __launch_bounds__(4 * H) lstm_fwd_pair_bf16_kernel(LstmFwdPairArgsB p) {
    extern __shared__ __attribute__((aligned(16))) bf16_t smem_b[];
    gu32* sync = (gu32*)p.sync;
    const int tiles = gridDim.x >> 1;
    // the ticket word lives at the end of the dynamic LDS region (a static __shared__ would misalign the images)
    int* role_lds = reinterpret_cast<int*>(reinterpret_cast<char*>(smem_b) + p.role_lds_off);
    if (threadIdx.x == 0) *role_lds = (int)__hip_atomic_fetch_add(sync, 1u, SHM_RLX_AGENT);
    __syncthreads();
    const int ticket = *role_lds;
    __syncthreads();
    // tickets are dealt in groups of PAIR_GROUP tiles: first the group's producers, then its consumers, so that with more tiles
    // than half the CUs a consumer still runs beside its producer (a consumer's producer always holds a smaller ticket)
    const int grp = ticket / (2 * PAIR_GROUP), r = ticket - grp * 2 * PAIR_GROUP, base = grp * PAIR_GROUP;
    const int ntg = min(PAIR_GROUP, tiles - base);
    const bool producer = r < ntg;
    const int tile = base + (producer ? r : r - ntg);
    if (threadIdx.x == 0 && producer) {                   // nobody touches the other block in this launch
        p.sync_other[16 * (1 + tile)] = 0u;
        if (ticket == 0) p.sync_other[0] = 0u;
    }
    if (producer) {
        if constexpr (XF32_0) lstm_fwd_body<H, 1, true, false, false, SAVE, false, true>(p.l0, tile, nullptr, sync + 16 * (1 + tile), (gu32*)p.err);
        else lstm_fwd_body<H, H / 32, false, false, true, SAVE, false, true>(p.l0, tile, nullptr, sync + 16 * (1 + tile), (gu32*)p.err);
    } else {
        lstm_fwd_body<H, H / 32, false, DROP, false, SAVE, true, false>(p.l1, tile, sync + 16 * (1 + tile), nullptr, (gu32*)p.err);
    }
}

template <int H, int XK, bool XF32, bool CONSTIN>
int launch_x(const LstmFwdArgsB& a, hipStream_t s) {
    const int tiles = ceil_div(a.B, WPT);
    const size_t lds = sizeof(bf16_t) * 4 * 16 * (4 * lds_row_stride(H / 4) + 2 * lds_row_stride(XK * 8)) + sizeof(float) * (H / 16) * 256;   // four h images (GRP), two x images
    const bool drop = a.drop_p > 0.f && !XF32 && !CONSTIN, save = a.sv_g != nullptr;
    SHM_REQUIRE((int64_t)a.T * (H / 16) * 64 * 12 < ((int64_t)1 << 31), SHM_EINVAL, "lstm_fwd_bf16: T too long for 32-bit buffer offsets");
    void (*kern)(LstmFwdArgsB);
    if constexpr (XF32 || CONSTIN)
        kern = save ? lstm_fwd_bf16_kernel<H, XK, XF32, false, CONSTIN, true> : lstm_fwd_bf16_kernel<H, XK, XF32, false, CONSTIN, false>;
    else
        kern = drop ? (save ? lstm_fwd_bf16_kernel<H, XK, false, true, false, true> : lstm_fwd_bf16_kernel<H, XK, false, true, false, false>)
                    : (save ? lstm_fwd_bf16_kernel<H, XK, false, false, false, true> : lstm_fwd_bf16_kernel<H, XK, false, false, false, false>);
    SHM_LAUNCH(kern, dim3(tiles), dim3(4 * H), lds, s, a);
    SHM_LAUNCH_CHECK();
    return SHM_OK;
}

template <int H>
int launch_h(const LstmFwdArgsB& a, hipStream_t s) {
    SHM_REQUIRE(a.h_seq != nullptr, SHM_EINVAL, "lstm_fwd_bf16: h_seq is NULL");
    if (a.in_f32 && !a.in_const) {
        SHM_REQUIRE(a.In <= 16, SHM_EINVAL, "lstm_fwd_bf16: fp32 input width %d too wide (<= 16)", a.In);
        return launch_x<H, 1, true, false>(a, s);
    }
    SHM_REQUIRE(a.In == H, SHM_EINVAL, "lstm_fwd_bf16: layer input width %d must equal hidden width %d", a.In, H);
    SHM_REQUIRE(!(a.drop_p > 0.f && !a.in_const) || a.xdrop != nullptr, SHM_EINVAL, "lstm_fwd_bf16: dropout needs the xdrop buffer");
    if (a.in_const) return launch_x<H, H / 32, false, true>(a, s);
    return launch_x<H, H / 32, false, false>(a, s);
}

}  // namespace

namespace {
template <int H>
int launch_pair_h(const LstmFwdPairArgsB& p0, hipStream_t s) {
    LstmFwdPairArgsB p = p0;
    const LstmFwdArgsB &a = p.l0, &b = p.l1;
    SHM_REQUIRE(a.h_seq && b.h_seq && p.sync && p.sync_other && p.sync != p.sync_other && p.err, SHM_EINVAL, "lstm_fwd_pair: h_seq or sync is NULL");
    SHM_REQUIRE(b.in == (const void*)a.h_seq && !b.in_f32 && !b.in_const && b.In == H, SHM_EINVAL, "lstm_fwd_pair: layer 1 must read layer 0's h_seq");
    SHM_REQUIRE(a.B == b.B && a.T == b.T, SHM_EINVAL, "lstm_fwd_pair: shape mismatch");
    SHM_REQUIRE((a.sv_g != nullptr) == (b.sv_g != nullptr), SHM_EINVAL, "lstm_fwd_pair: both layers save or neither");
    SHM_REQUIRE((int64_t)a.T * WPT * H * 2 < ((int64_t)1 << 31) && (int64_t)a.T * (H / 16) * 64 * 12 < ((int64_t)1 << 31), SHM_EINVAL,
                "lstm_fwd_pair: T too long for 32-bit buffer offsets");
    const bool xf32 = a.in_f32 && !a.in_const;
    if (xf32) SHM_REQUIRE(a.In <= 16, SHM_EINVAL, "lstm_fwd_pair: fp32 input width %d too wide (<= 16)", a.In);
    else SHM_REQUIRE(a.in_const && a.In == H, SHM_EINVAL, "lstm_fwd_pair: layer 0 must be the fp32 input or the constant-input layer");
    const bool drop = b.drop_p > 0.f, save = a.sv_g != nullptr;
    SHM_REQUIRE(!drop || b.xdrop, SHM_EINVAL, "lstm_fwd_pair: dropout needs the xdrop buffer");
    const int tiles = ceil_div(a.B, WPT);
    const size_t lds = sizeof(bf16_t) * 4 * 16 * (4 * lds_row_stride(H / 4) + 2 * lds_row_stride(H / 4)) + sizeof(float) * (H / 16) * 256;
    p.role_lds_off = (int)lds;
    void (*kern)(LstmFwdPairArgsB);
    if (xf32) kern = drop ? (save ? lstm_fwd_pair_bf16_kernel<H, true, true, true> : lstm_fwd_pair_bf16_kernel<H, true, true, false>)
                          : (save ? lstm_fwd_pair_bf16_kernel<H, true, false, true> : lstm_fwd_pair_bf16_kernel<H, true, false, false>);
    else kern = drop ? (save ? lstm_fwd_pair_bf16_kernel<H, false, true, true> : lstm_fwd_pair_bf16_kernel<H, false, true, false>)
                     : (save ? lstm_fwd_pair_bf16_kernel<H, false, false, true> : lstm_fwd_pair_bf16_kernel<H, false, false, false>);
    SHM_LAUNCH(kern, dim3(2 * tiles), dim3(4 * H), lds + 16, s, p);
    SHM_LAUNCH_CHECK();
    return SHM_OK;
}
}  // namespace

size_t lstm_fwd_pair_sync_bytes(int B) { return (size_t)64 * (1 + ceil_div(B, WPT)); }

int launch_lstm_fwd_pair_bf16(int H, const LstmFwdPairArgsB& p, hipStream_t s) {
    SHM_REQUIRE(p.l0.B > 0 && p.l0.T > 0 && p.l0.In > 0, SHM_EINVAL, "lstm_fwd_pair: empty shape");
    switch (H) {
        case 32: return launch_pair_h<32>(p, s);
        case 64: return launch_pair_h<64>(p, s);
        case 128: return launch_pair_h<128>(p, s);
        default: set_error("lstm_fwd_pair: hidden_dim %d unsupported (32, 64, 128)", H); return SHM_EINVAL;
    }
}

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
