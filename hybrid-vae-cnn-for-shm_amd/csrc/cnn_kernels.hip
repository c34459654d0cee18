// cnn_kernels.hip — the CNN fault classifier (reference 4DOF/Scripts/Models/cnn_model.py:14-51) as gfx950 kernels, fp32.
//
//   conv3x3 (pad 1, im2col-free: the whole [C,T,D] window sits in LDS, one thread per output pixel, all output
//   channels in registers) -> BatchNorm2d (batch statistics in train mode, running statistics in eval) -> ReLU ->
//   MaxPool2d(2) with the first-maximum rule, twice; flatten (C,H,W); Linear+ReLU+Dropout; Linear; CrossEntropy.
// Backward mirrors it: pool/ReLU/BatchNorm backward in two passes (per-channel sums, then apply), conv data gradient
// (conv2 only) and weight gradients with per-window partials reduced in a fixed order (bitwise reproducible).
// The windows are tiny (T*D <= 1200 pixels) and the step is launch/latency-bound, so the kernels favour simple,
// fully-coalesced NCHW accesses over MFMA; one workgroup per window throughout.
#include <type_traits>

#include "cnn_kernels.hpp"

namespace shm {

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
// sum over the 256 threads of a block; red: >= 4 floats of LDS; all threads get the result
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// Staging loops (global -> LDS): U loads in flight per thread.  Written as "dst[i] = src[i]" per trip, every global read waits out its
// whole latency before the next one is issued — a 32-trip staging loop was most of a 30 us kernel.
template <int NTHR, int U = 8, typename LoadF, typename StoreF>
__device__ __forceinline__ void stage_elems(int n, int tid, LoadF load, StoreF store) {
    for (int e0 = tid; e0 < n; e0 += NTHR * U) {
        float v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const int e = e0 + u * NTHR; v[u] = load(e < n ? e : 0); }
#pragma unroll
        for (int u = 0; u < U; ++u) { const int e = e0 + u * NTHR; if (e < n) store(e, v[u]); }
    }
}

// ------------------------------------------------------------------------------------------------ conv forward
// grid = B, block = 256.  LDS: x[CIN][H*W] | w[(ci*9+k)][COUT] | red[4 waves][COUT][2]
template <int CIN, int COUT, bool STATS>
__global__ void __launch_bounds__(256) conv3x3_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ y,
                                                          float* __restrict__ part, int H, int W) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int HW = H * W, b = blockIdx.x, tid = threadIdx.x;
    float* xs = sm;
    float* ws = sm + ((CIN * HW + 3) & ~3);          // keep the weight rows 16-byte aligned for the float4 reads
    float* red = ws + CIN * 9 * COUT;
    stage_elems<256>(CIN * HW, tid, [&](int e) { return x[(size_t)b * CIN * HW + e]; }, [&](int e, float v) { xs[e] = v; });
    stage_elems<256>(COUT * CIN * 9, tid, [&](int e) { return w[e]; }, [&](int e, float v) {   // w[co][ci][k] -> ws[(ci*9+k)*COUT + co]
        const int co = e / (CIN * 9), r = e - co * CIN * 9;
        ws[r * COUT + co] = v;
    });
    __syncthreads();
    float s1[COUT], s2[COUT];
    if (STATS) {
#pragma unroll
        for (int co = 0; co < COUT; ++co) s1[co] = s2[co] = 0.f;
    }
    for (int p = tid; p < HW; p += 256) {
        const int h = p / W, ww = p - h * W;
        float acc[COUT];
#pragma unroll
        for (int co = 0; co < COUT; ++co) acc[co] = bias[co];
        for (int ci = 0; ci < CIN; ++ci) {
            float v[9];
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const int hh = h + kh - 1, w2 = ww + kw - 1;
                    v[kh * 3 + kw] = (hh >= 0 && hh < H && w2 >= 0 && w2 < W) ? xs[ci * HW + hh * W + w2] : 0.f;
                }
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const float* wr = ws + (ci * 9 + k) * COUT;
#pragma unroll
                for (int co = 0; co < COUT; co += 4) {
                    const f32x4 w4 = *reinterpret_cast<const f32x4*>(wr + co);
                    acc[co] = fmaf(v[k], w4[0], acc[co]); acc[co + 1] = fmaf(v[k], w4[1], acc[co + 1]);
                    acc[co + 2] = fmaf(v[k], w4[2], acc[co + 2]); acc[co + 3] = fmaf(v[k], w4[3], acc[co + 3]);
                }
            }
        }
#pragma unroll
        for (int co = 0; co < COUT; ++co) {
            y[((size_t)b * COUT + co) * HW + p] = acc[co];
            if (STATS) { s1[co] += acc[co]; s2[co] = fmaf(acc[co], acc[co], s2[co]); }
        }
    }
    if (STATS) {
        // wave sums by shuffles, ONE barrier, then a fixed-order sum over the four waves (a block-wide sum per channel and moment
        // cost 2 x COUT pairs of barriers: most of this kernel's time)
        const int wv = tid >> 6, lane = tid & 63;
#pragma unroll
        for (int co = 0; co < COUT; ++co) {
            const float a = wave_sum(s1[co]), q = wave_sum(s2[co]);
            if (lane == 0) { red[(wv * COUT + co) * 2] = a; red[(wv * COUT + co) * 2 + 1] = q; }
        }
        __syncthreads();
        for (int e = tid; e < 2 * COUT; e += 256)
            part[(size_t)b * COUT * 2 + e] = (red[e] + red[2 * COUT + e]) + (red[4 * COUT + e] + red[6 * COUT + e]);
    }
}

// The pool / ReLU / BatchNorm backward of one element, for the kernels that form dy on the fly instead of reading it:
// dy = gamma rstd (da - k1 - xhat k2), xhat = (y - mean) rstd, da = the pooled gradient if this position was the cell's first maximum.
__device__ __forceinline__ float bn_dy(const BnGrad& g, int c, float y, float da) {
    const float rs = g.rstd[c];
    return g.gamma[c] * rs * (da - g.k1[c] - (y - g.mean[c]) * rs * g.k2[c]);
}

// The same convolution on the matrix pipe, exact fp32 (v_mfma_f32_16x16x4_f32): y[co][p] = b[co] + sum_k W[co][k] col[k][p] with k = ci * 9 + tap
// in the weights' own order (no re-layout) and col = the im2col view of x (never materialised): M = 16 output channels, N = 16
// consecutive positions, K = CIN * 9 padded to a multiple of 4.  The A fragments of a wave (all of W it needs: KS values per M-tile)
// live in registers for the whole window; a B value is one 4-byte LDS read out of the halo-padded x at (position offset + k offset).
// grid = B, block = 512 (8 waves, position tiles dealt round-robin).  STATS: per-channel sum / sum of squares of this window,
// lane-local over the wave's tiles, combined by shuffles, one barrier, then a fixed-order sum over the waves.
// DGRAD: the same kernel as the DATA gradient of a convolution — dx[ci][p] = sum_{co, tap} dy[co][p - off(tap)] W[co][ci][tap] is a
// 3x3 convolution of dy with the weights transposed and flipped (A[ci][co][tap'] = W[co][ci][8 - tap']; CIN / COUT below are then the
// conv's output / input channel counts), no bias, no statistics; its input x = dy is what the K-split weight-gradient kernel leaves
// (or the separate BatchNorm-backward launch writes).  Replaces the vector kernel conv3x3_bwd_data_kernel (40 us at B = 512, LDS-bound on the
// weight broadcasts).
template <int CIN, int COUT, bool STATS, bool DGRAD = false>
__global__ void __launch_bounds__(512, 4) conv3x3_fwd_mfma_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                               const float* __restrict__ bias, float* __restrict__ y,
                                                               float* __restrict__ part, int H, int W) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    // MTW output-channel tiles per wave: all of them while the weights fit comfortably in registers (conv1: 1 x 5 values), ONE when
    // they would not (conv2: 36 values per tile; two tiles per wave needed 187 VGPRs = one workgroup per CU) — then the waves are
    // split into MT groups and each group walks the position tiles for its own output-channel tile.
    constexpr int NK = CIN * 9, KS = (NK + 3) / 4, MT = COUT / 16, NWV = 8, MTW = KS * MT <= 40 ? MT : 1, MG = MT / MTW;
    static_assert(NWV % MG == 0, "wave groups");
    const int HW = H * W, WP = W + 2, XP = (H + 2) * WP, ntiles = (HW + 15) / 16;
    const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, q = lane >> 4, j = lane & 15;
    const int mg = wave % MG, m0 = mg * MTW;                  // this wave's output-channel tiles: m0 .. m0 + MTW - 1
    float* xs = sm;                                           // [CIN][H + 2][W + 2], zero halo
    float* red = sm + ((CIN * XP + 3) & ~3);                  // STATS: [NWV][COUT][2]
    int* pos = reinterpret_cast<int*>(red + NWV * COUT * 2);  // [HW] halo'd offset (h + 1) * WP + (w + 1) of position p
    // Staging: the halo is only ever zero, so it is stored, not loaded — zero-fill, then scatter the CIN * HW interior values with their
    // halo'd offsets from a table built once per workgroup (the two divisions per cell used to sit inside the staging loop, and every
    // halo cell cost a masked load).  SU loads in flight per thread: a 16-channel 256-position window is one batch of 8.
    for (int e = tid; e < CIN * XP; e += 512) xs[e] = 0.f;
    for (int p = tid; p < HW; p += 512) { const int h = p / W; pos[p] = (h + 1) * WP + (p - h * W) + 1; }
    __syncthreads();
    {
        constexpr int SU = CIN >= 16 ? 16 : 8;
        int ci = tid / HW, p = tid - ci * HW;                  // element e = tid + 512 k  ->  (ci, p), advanced without dividing
        for (int e0 = tid; e0 < CIN * HW; e0 += 512 * SU) {
            float v[SU];
            int dst[SU];
#pragma unroll
            for (int u = 0; u < SU; ++u) {
                const int cu = ci < CIN ? ci : CIN - 1;
                v[u] = x[((size_t)b * CIN + cu) * HW + p];
                dst[u] = cu * XP + pos[p];
                p += 512;
                while (p >= HW) { p -= HW; ++ci; }
            }
#pragma unroll
            for (int u = 0; u < SU; ++u)
                if (e0 + u * 512 < CIN * HW) xs[dst[u]] = v[u];
        }
    }
    // K order.  CIN % 4 == 0 (conv2): k = tap * CIN + ci, so the k-step s = k / 4 of lane quarter q is channel 4 (s % (CIN/4)) + q at tap
    // s / (CIN/4): the B address is (this lane's position + q * XP) + a wave-uniform offset per k-step — no per-lane offset table
    // (36 registers at CIN = 16), which is what lets the B reads of a whole tile be issued as a block.  Otherwise (conv1, K = 18): the
    // weights' own order k = ci * 9 + tap with a per-lane table of KS offsets.
    // A: lane (q, j) holds W[(m0 + m) * 16 + j][k(s, q)].
    constexpr bool TAPMAJOR = CIN % 4 == 0;
    constexpr int CQ = TAPMAJOR ? CIN / 4 : 1;
    // A lives in registers for the whole window (every position tile reuses it) unless K is long (DGRAD of conv2: 72 k-steps) — then a
    // wave walks its position tiles in ONE trip and fetches each A value as it is used (ASTREAM; L1 / L2 hits, 147 -> <= 128 VGPRs,
    // i.e. two workgroups per CU instead of one).
    constexpr bool ASTREAM = KS > 40;
    auto a_val = [&](int m, int s2, int& koff_out) -> float {
        int k, ok;
        if constexpr (TAPMAJOR) {
            const int ci = 4 * (s2 % CQ) + q, tap = s2 / CQ;
            k = ci * 9 + tap; ok = 1; koff_out = 0;
        } else {
            k = 4 * s2 + q; ok = k < NK;
            const int kk = ok ? k : 0, ci = kk / 9, tap = kk - ci * 9;
            koff_out = ci * XP + (tap / 3 - 1) * WP + (tap % 3 - 1);
            k = kk;
        }
        const int o = (m0 + m) * 16 + j;
        if constexpr (DGRAD) { const int ci = k / 9, tap = k - ci * 9; return ok ? w[((size_t)ci * COUT + o) * 9 + (8 - tap)] : 0.f; }
        else return ok ? w[(size_t)o * NK + k] : 0.f;          // zero weight: the padded k contributes nothing
    };
    float a[MTW][ASTREAM ? 1 : KS];
    int koff[TAPMAJOR ? 1 : KS];
    int qoff = 0;
    if constexpr (!ASTREAM) {
#pragma unroll
        for (int s2 = 0; s2 < KS; ++s2)
#pragma unroll
            for (int m = 0; m < MTW; ++m) { int ko; a[m][s2] = a_val(m, s2, ko); if constexpr (!TAPMAJOR) koff[s2] = ko; }
    }
    if constexpr (TAPMAJOR) qoff = q * XP;
    auto boff = [&](int s2) -> int {                          // B offset of k-step s2 relative to the lane's (position [+ q * XP]) address
        if constexpr (TAPMAJOR) { const int tap = s2 / CQ; return 4 * (s2 % CQ) * XP + (tap / 3 - 1) * WP + (tap % 3 - 1); }
        else return koff[s2];
    };
    float bia[MTW][4], s1[MTW][4], s2v[MTW][4];
#pragma unroll
    for (int m = 0; m < MTW; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) { bia[m][r] = DGRAD ? 0.f : bias[(m0 + m) * 16 + 4 * q + r]; s1[m][r] = 0.f; s2v[m][r] = 0.f; }
    __syncthreads();
    // Two position tiles per trip (two independent accumulator chains per output-channel tile), their B values read from LDS in blocks
    // of KB k-steps BEFORE the block's MFMAs: written as read-one / multiply-one the compiler waited out the LDS latency in front of
    // every MFMA (matrix pipe 19 % busy, 40 us for conv2 at B = 512).
    constexpr int KB = KS < 12 ? KS : (KS > 40 ? 8 : 12), NTI = 2;      // (KS > 40: the A values of a block are fetched with it)
    const int stride_t = NWV / MG;
    for (int nt0 = wave / MG; nt0 < ntiles; nt0 += NTI * stride_t) {
        const float* xb[NTI];
        int pp[NTI];
        f32x4 acc[NTI][MTW];
#pragma unroll
        for (int t = 0; t < NTI; ++t) {
            const int nt = nt0 + t * stride_t;
            pp[t] = nt < ntiles ? nt * 16 + j : HW;              // HW = "no position": computed on a valid cell, never stored
            const int pc = pp[t] < HW ? pp[t] : HW - 1, h = pc / W, ww = pc - h * W;
            xb[t] = xs + (h + 1) * WP + (ww + 1) + qoff;
#pragma unroll
            for (int m = 0; m < MTW; ++m) acc[t][m] = f32x4{bia[m][0], bia[m][1], bia[m][2], bia[m][3]};
        }
#pragma unroll
        for (int s0 = 0; s0 < KS; s0 += KB) {
            float bv[NTI][KB], av[MTW][KB];
            if constexpr (ASTREAM) {
                __builtin_amdgcn_sched_barrier(0);      // keep the blocks apart: hoisted to the top, the 72 A loads spill
#pragma unroll
                for (int u = 0; u < KB; ++u)
#pragma unroll
                    for (int m = 0; m < MTW; ++m) { int ko; if (s0 + u < KS) av[m][u] = a_val(m, s0 + u, ko); }
            }
#pragma unroll
            for (int u = 0; u < KB; ++u)
#pragma unroll
                for (int t = 0; t < NTI; ++t)
                    if (s0 + u < KS) bv[t][u] = xb[t][boff(s0 + u)];
#pragma unroll
            for (int u = 0; u < KB; ++u)
#pragma unroll
                for (int t = 0; t < NTI; ++t)
#pragma unroll
                    for (int m = 0; m < MTW; ++m)
                        if (s0 + u < KS) acc[t][m] = mfma_f32_16x16x4(ASTREAM ? av[m][u] : a[m][s0 + u], bv[t][u], acc[t][m]);
        }
#pragma unroll
        for (int t = 0; t < NTI; ++t) {
            const int p = pp[t];
            if (p < HW) {
#pragma unroll
                for (int m = 0; m < MTW; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float v = acc[t][m][r];
                        y[((size_t)b * COUT + (m0 + m) * 16 + 4 * q + r) * HW + p] = v;
                        if (STATS) { s1[m][r] += v; s2v[m][r] = fmaf(v, v, s2v[m][r]); }
                    }
            }
        }
    }
    if (STATS) {
        // every wave leaves a row [COUT][2] (zeros for the channel tiles it does not own); fixed-order sum over the waves
        for (int e = lane; e < 2 * COUT; e += 64) red[wave * COUT * 2 + e] = 0.f;
#pragma unroll
        for (int m = 0; m < MTW; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float u = s1[m][r], v = s2v[m][r];
#pragma unroll
                for (int off = 1; off < 16; off <<= 1) { u += __shfl_xor(u, off, 64); v += __shfl_xor(v, off, 64); }
                if (j == 0) { red[(wave * COUT + (m0 + m) * 16 + 4 * q + r) * 2] = u; red[(wave * COUT + (m0 + m) * 16 + 4 * q + r) * 2 + 1] = v; }
            }
        __syncthreads();
        for (int e = tid; e < 2 * COUT; e += 512) {
            float v = 0.f;
#pragma unroll
            for (int wv = 0; wv < NWV; ++wv) v += red[wv * COUT * 2 + e];
            part[(size_t)b * COUT * 2 + e] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------ BatchNorm statistics
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
// (sum, sum of squares) of channel c over the B per-window partial pairs, as one lane's share: windows lane, lane + 64, ... added in
// that order in fp64 (the caller finishes with a butterfly).  The pairs are fetched as 8-byte loads, eight in flight: written as one
// dependent load pair per trip, this loop's latency was most of the 5 us the finalize launches take.
__device__ __forceinline__ void partial_pair_sums(const float* __restrict__ part, int C, int c, int B, int lane, double& s, double& q) {
    s = 0.0; q = 0.0;
    for (int b0 = lane; b0 < B; b0 += 64 * 8) {
        f32x2 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int b = b0 + u * 64;
            v[u] = *reinterpret_cast<const f32x2*>(part + ((size_t)(b < B ? b : lane) * C + c) * 2);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (b0 + u * 64 < B) { s += (double)v[u][0]; q += (double)v[u][1]; }
    }
}
// grid = C, one wave per channel: lanes stride over the B per-window partials (fp64), then a butterfly — fixed order.
// train: batch mean / biased var, running statistics updated with momentum 0.1 and the unbiased variance; eval: running statistics.
// (One thread per channel walking all B partials serially took 126 us per launch at B = 512.)
__global__ void __launch_bounds__(64) bn_finalize_kernel(int C, int B, long long n_per_channel, int train, const float* __restrict__ part,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         float* __restrict__ run_mean, float* __restrict__ run_var,
                                                         float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                         float* __restrict__ scale, float* __restrict__ shift) {
    const int c = blockIdx.x;
    float mean, var;
    const float rm0 = run_mean[c], rv0 = run_var[c], gm = gamma[c], bt = beta[c];      // issued beside the partial loads, not behind the reduction
    if (train) {
        double s, q;
        partial_pair_sums(part, C, c, B, threadIdx.x, s, q);
        s = wave_sum_f64(s); q = wave_sum_f64(q);
        if (threadIdx.x != 0) return;
        const double m = s / (double)n_per_channel;
        double v = q / (double)n_per_channel - m * m;
        if (v < 0.0) v = 0.0;
        mean = (float)m; var = (float)v;
        const double unb = n_per_channel > 1 ? v * (double)n_per_channel / (double)(n_per_channel - 1) : v;
        run_mean[c] = 0.9f * rm0 + 0.1f * mean;
        run_var[c] = 0.9f * rv0 + 0.1f * (float)unb;
    } else {
        if (threadIdx.x != 0) return;
        mean = rm0; var = rv0;
    }
    const float rstd = 1.0f / sqrtf(var + 1e-5f);
    mean_out[c] = mean; rstd_out[c] = rstd;
    scale[c] = gm * rstd;
    shift[c] = bt - mean * gm * rstd;
}

// ------------------------------------------------------------------------------------------------ BN + ReLU + MaxPool2
// grid = B.  idx: 0..3 = position (kh*2+kw) of the first maximum of the ReLU'd 2x2 cell, 4 = cell is dead (max <= 0)
// xh (training): the normalised conv output (y - mean) rstd AT the argmax of every live cell — all the BatchNorm-backward statistics
// pass needs of y, so that pass reads a pooled-size tensor instead of gathering through the full-size one.
__global__ void __launch_bounds__(256) bn_relu_pool_kernel(int C, int H, int W, const float* __restrict__ y,
                                                           const float* __restrict__ scale, const float* __restrict__ shift,
                                                           float* __restrict__ out, unsigned char* __restrict__ idx,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd, float* __restrict__ xh) {
    const int Hp = H / 2, Wp = W / 2, b = blockIdx.x;
    const int n = C * Hp * Wp;
    for (int e = threadIdx.x; e < n; e += 256) {
        const int c = e / (Hp * Wp), r = e - c * Hp * Wp, hp = r / Wp, wp = r - hp * Wp;
        const float* yp = y + ((size_t)b * C + c) * H * W + (size_t)(2 * hp) * W + 2 * wp;
        const float sc = scale[c], sh = shift[c];
        const float y0 = yp[0], y1 = yp[1], y2 = yp[W], y3 = yp[W + 1];
        float best = fmaxf(fmaf(y0, sc, sh), 0.f), yb = y0;
        int bi = 0;
        const float v1 = fmaxf(fmaf(y1, sc, sh), 0.f), v2 = fmaxf(fmaf(y2, sc, sh), 0.f), v3 = fmaxf(fmaf(y3, sc, sh), 0.f);
        if (v1 > best) { best = v1; bi = 1; yb = y1; }
        if (v2 > best) { best = v2; bi = 2; yb = y2; }
        if (v3 > best) { best = v3; bi = 3; yb = y3; }
        out[(size_t)b * n + e] = best;
        if (idx) idx[(size_t)b * n + e] = best > 0.f ? (unsigned char)bi : (unsigned char)4;
        if (xh) xh[(size_t)b * n + e] = (yb - mean[c]) * rstd[c];
    }
}

// ------------------------------------------------------------------------------------------------ dense layers
// out[b][o] = sum_f in[b][f] W[o][f] + bias[o]; optional ReLU (+ pre-activation copy) and dropout.  Tile = 8 rows x 32 outputs
// (one output per thread): fc1 at B = 512 is 256 workgroups — the 32 x 32 tiling gave 64 and 112 us per launch.  The F loop is
// staged 64 wide with the next chunk's global loads issued before the current chunk is consumed.
__global__ void __launch_bounds__(256) linear_fwd_kernel(int B, int F, int O, const float* __restrict__ in, const float* __restrict__ Wt,
                                                         const float* __restrict__ bias, float* __restrict__ pre, float* __restrict__ out,
                                                         int relu, float drop_p, unsigned long long drop_seed) {
    constexpr int FC = 64;
    __shared__ float sI[8][FC + 1];
    __shared__ float sW[32][FC + 1];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int b0 = blockIdx.y * 8, o0 = blockIdx.x * 32;
    float acc = 0.f;
    float ri[2], rw[8];
    auto load = [&](int f0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int e = threadIdx.x + i * 256, r = e >> 6, k = e & 63;
            ri[i] = (b0 + r < B && f0 + k < F) ? in[(size_t)(b0 + r) * F + f0 + k] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int e = threadIdx.x + i * 256, r = e >> 6, k = e & 63;
            rw[i] = (o0 + r < O && f0 + k < F) ? Wt[(size_t)(o0 + r) * F + f0 + k] : 0.f;
        }
    };
    load(0);
    for (int f0 = 0; f0 < F; f0 += FC) {
#pragma unroll
        for (int i = 0; i < 2; ++i) { const int e = threadIdx.x + i * 256; sI[e >> 6][e & 63] = ri[i]; }
#pragma unroll
        for (int i = 0; i < 8; ++i) { const int e = threadIdx.x + i * 256; sW[e >> 6][e & 63] = rw[i]; }
        __syncthreads();
        if (f0 + FC < F) load(f0 + FC);
#pragma unroll 16
        for (int k = 0; k < FC; ++k) acc = fmaf(sI[ty][k], sW[tx][k], acc);
        __syncthreads();
    }
    const uint32_t thr = drop_threshold(drop_p);
    const float dscale = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
    {
        const int b = b0 + ty, o = o0 + tx;
        if (b < B && o < O) {
            float v = acc + bias[o];
            if (pre) pre[(size_t)b * O + o] = v;
            if (relu) v = fmaxf(v, 0.f);
            if (drop_p > 0.f) v = drop_hash(drop_seed, 2u, (uint32_t)(b * O + o)) >= thr ? v * dscale : 0.f;
            out[(size_t)b * O + o] = v;
        }
    }
}

// dx[b][f] = sum_o dy[b][o] W[o][f]; optional ReLU mask (pre > 0) and dropout mask on the result (fc2 -> fc1 hand-off)
__global__ void __launch_bounds__(256) linear_bwd_dx_kernel(int B, int F, int O, const float* __restrict__ dy, const float* __restrict__ Wt,
                                                            const float* __restrict__ pre, float* __restrict__ dx, float drop_p,
                                                            unsigned long long drop_seed) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)B * F) return;
    const int b = (int)(i / F), f = (int)(i - (size_t)b * F);
    float acc = 0.f;
    for (int o = 0; o < O; ++o) acc = fmaf(dy[(size_t)b * O + o], Wt[(size_t)o * F + f], acc);
    if (drop_p > 0.f) acc = drop_hash(drop_seed, 2u, (uint32_t)i) >= drop_threshold(drop_p) ? acc / (1.0f - drop_p) : 0.f;
    if (pre) acc = pre[i] > 0.f ? acc : 0.f;
    dx[i] = acc;
}

// The two dense products above on the matrix pipe, exact fp32 (v_mfma_f32_16x16x4_f32), for layers at least 16 wide.
// Forward: one workgroup = one 16 x 16 output tile, its four waves split K = F (both operands are K-contiguous rows, so a lane's four
// k-steps are one 16-byte global read each for the input row and the weight row — no LDS staging), accumulators added in wave order,
// then the same epilogue as linear_fwd_kernel.  Requires F % 16 == 0.
__global__ void __launch_bounds__(256) linear_fwd_mfma_kernel(int B, int F, int O, const float* __restrict__ in, const float* __restrict__ Wt,
                                                              const float* __restrict__ bias, float* __restrict__ pre, float* __restrict__ out,
                                                              int relu, float drop_p, unsigned long long drop_seed) {
    __shared__ __attribute__((aligned(16))) float red[4][256];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, q = lane >> 4, j = lane & 15;
    const int o0 = blockIdx.x * 16, b0 = blockIdx.y * 16;
    const int kblocks = F / 16, per = (kblocks + 3) / 4, k0 = wave * per, k1 = min(kblocks, k0 + per);
    const float* arow = in + (size_t)min(b0 + j, B - 1) * F + 4 * q;           // rows past the batch / the layer shadow the last valid one
    const float* wrow = Wt + (size_t)min(o0 + j, O - 1) * F + 4 * q;           // and are dropped in the epilogue
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    // four k-blocks of operands in flight (a rotating register ring; loads past the range re-read the last block and are not used)
    constexpr int PF = 4;
    f32x4 ar[PF], wr[PF];
    const int klast = max(k1 - 1, k0);
#pragma unroll
    for (int u = 0; u < PF; ++u) {
        const int kk = min(k0 + u, klast);
        ar[u] = *reinterpret_cast<const f32x4*>(arow + kk * 16); wr[u] = *reinterpret_cast<const f32x4*>(wrow + kk * 16);
    }
    for (int kb = k0; kb < k1; kb += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            if (kb + u < k1) {
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = mfma_f32_16x16x4(ar[u][e], wr[u][e], acc);
            }
            const int kk = min(kb + u + PF, klast);
            ar[u] = *reinterpret_cast<const f32x4*>(arow + kk * 16); wr[u] = *reinterpret_cast<const f32x4*>(wrow + kk * 16);
        }
    }
    *reinterpret_cast<f32x4*>(&red[wave][lane * 4]) = acc;    // accumulator (q, j), r  <->  row 4q + r (batch), column j (output)
    __syncthreads();
    {
        const int l = tid >> 2, r = tid & 3, b = b0 + 4 * (l >> 4) + r, o = o0 + (l & 15);
        if (b < B && o < O) {
            float v = ((red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid])) + bias[o];
            if (pre) pre[(size_t)b * O + o] = v;
            if (relu) v = fmaxf(v, 0.f);
            if (drop_p > 0.f) v = drop_hash(drop_seed, 2u, (uint32_t)(b * O + o)) >= drop_threshold(drop_p) ? v * (1.0f / (1.0f - drop_p)) : 0.f;
            out[(size_t)b * O + o] = v;
        }
    }
}
// Backward data: dx[b][f] = sum_o dy[b][o] W[o][f].  A wave = one 16 (batch) x 16 (f) tile, K = O: dy rows are K-contiguous (16-byte reads),
// the weight operand is four 4-byte reads per k-block (rows o, 16 consecutive f per quarter-wave).  Requires O % 16 == 0.
__global__ void __launch_bounds__(256) linear_bwd_dx_mfma_kernel(int B, int F, int O, const float* __restrict__ dy, const float* __restrict__ Wt,
                                                                 const float* __restrict__ pre, float* __restrict__ dx, float drop_p,
                                                                 unsigned long long drop_seed) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, q = lane >> 4, j = lane & 15;
    const int f0 = (blockIdx.x * 4 + wave) * 16, b0 = blockIdx.y * 16;
    if (f0 >= F) return;
    const float* arow = dy + (size_t)min(b0 + j, B - 1) * O + 4 * q;
    const float* wcol = Wt + (size_t)(4 * q) * F + min(f0 + j, F - 1);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    constexpr int PF = 4;                                     // k-blocks of operands in flight (see linear_fwd_mfma_kernel)
    f32x4 ar[PF], wr[PF];
    const int nkb = O / 16;
    auto fetch = [&](int u, int kb) {
        const int kk = min(kb, nkb - 1);
        ar[u] = *reinterpret_cast<const f32x4*>(arow + kk * 16);
        const float* wp = wcol + (size_t)kk * 16 * F;
        wr[u][0] = wp[0]; wr[u][1] = wp[F]; wr[u][2] = wp[2 * (size_t)F]; wr[u][3] = wp[3 * (size_t)F];
    };
#pragma unroll
    for (int u = 0; u < PF; ++u) fetch(u, u);
    for (int kb = 0; kb < nkb; kb += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            if (kb + u < nkb) {
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = mfma_f32_16x16x4(ar[u][e], wr[u][e], acc);
            }
            fetch(u, kb + u + PF);
        }
    }
    const int f = f0 + j;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int b = b0 + 4 * q + r;
        if (b < B && f < F) {
            const size_t i = (size_t)b * F + f;
            float v = acc[r];
            if (drop_p > 0.f) v = drop_hash(drop_seed, 2u, (uint32_t)i) >= drop_threshold(drop_p) ? v / (1.0f - drop_p) : 0.f;
            if (pre) v = pre[i] > 0.f ? v : 0.f;
            dx[i] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------ dense head, backward
// Output layer, everything in one launch (was: split-K GEMM + column sum + data gradient = three launches and 30 us for ~0.3 MFLOP):
//   dz1[b][f] = relu'(z1) * dropmask * sum_o dl[b][o] W2[o][f]                 (the hand-off into fc1's backward)
//   part[g][o][f] = sum_{b in group g} dl[b][o] a1[b][f],  part[g][NC*F1 + o] = sum_{b in g} dl[b][o]
// grid = ceil(B / 16), block = 2 * F1 threads (F1 = hidden width, <= 256): thread (half, f) walks 8 of the group's 16 windows; the two
// halves are added in a fixed order.  The per-group partials go through the backward pass' one reduce launch.
template <int NCMAX>
__global__ void __launch_bounds__(512) dense_out_bwd_kernel(int B, int F1, int NC, const float* __restrict__ dl, const float* __restrict__ W2,
                                                            const float* __restrict__ z1, const float* __restrict__ a1, float* __restrict__ dz1,
                                                            float* __restrict__ part, float drop_p, unsigned long long drop_seed) {
    extern __shared__ __attribute__((aligned(16))) float sm[];      // [F1][NCMAX] partials of the upper half | [16][NCMAX] dl of the group
    float* dls = sm + (size_t)F1 * NCMAX;
    const int g = blockIdx.x, tid = threadIdx.x, f = tid % F1, half = tid / F1, b0 = g * 16;
    for (int e = tid; e < 16 * NCMAX; e += blockDim.x) {
        const int r = e / NCMAX, o = e - r * NCMAX;
        dls[e] = (b0 + r < B && o < NC) ? dl[(size_t)(b0 + r) * NC + o] : 0.f;
    }
    float w2[NCMAX], acc[NCMAX];
#pragma unroll
    for (int o = 0; o < NCMAX; ++o) { w2[o] = o < NC ? W2[(size_t)o * F1 + f] : 0.f; acc[o] = 0.f; }
    __syncthreads();
    const uint32_t thr = drop_threshold(drop_p);
    const float dscale = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int r = half * 8 + i, b = b0 + r;
        if (b < B) {
            const size_t e = (size_t)b * F1 + f;
            const float av = a1[e], zv = z1[e];
            float d = 0.f;
#pragma unroll
            for (int o = 0; o < NCMAX; ++o) { d = fmaf(dls[r * NCMAX + o], w2[o], d); acc[o] = fmaf(dls[r * NCMAX + o], av, acc[o]); }
            if (drop_p > 0.f) d = drop_hash(drop_seed, 2u, (uint32_t)e) >= thr ? d * dscale : 0.f;
            dz1[e] = zv > 0.f ? d : 0.f;
        }
    }
    if (half == 1) {
#pragma unroll
        for (int o = 0; o < NCMAX; ++o) sm[f * NCMAX + o] = acc[o];
    }
    __syncthreads();
    float* out = part + (size_t)g * ((size_t)NC * F1 + NC);
    if (half == 0) {
#pragma unroll
        for (int o = 0; o < NCMAX; ++o)
            if (o < NC) out[(size_t)o * F1 + f] = acc[o] + sm[f * NCMAX + o];
    }
    if (tid < NC) {
        float v = 0.f;
        for (int r = 0; r < 16; ++r) v += dls[r * NCMAX + tid];
        out[(size_t)NC * F1 + tid] = v;
    }
}

// Hidden layer's weight gradient as ONE exact-fp32 MFMA product without split-K partials: dW[o][f] = sum_b dz[b][o] in[b][f], M = O (all
// MT = O / 16 tiles in every wave), N = 16 columns f per workgroup, K = B dealt over the 8 waves in blocks of four windows; both
// operands are read straight from global memory (lane (q, j): window 4 s + q, 16 consecutive o / f — 64-byte segments; dz is L2
// resident, `in` is read exactly once by the whole grid), four k-blocks in flight per wave.  The wave accumulators are added in wave
// order through LDS.  Workgroup 0 also leaves the bias gradient (column sums of dz) out of its A operands.  (Was: a 256-way split-K
// GEMM + its share of the reduce launch + a column-sum launch: 26 us.)   Requires O % 16 == 0, F % 16 == 0.
template <int MT>
__global__ void __launch_bounds__(512) dense_dw_mfma_kernel(int B, int F, const float* __restrict__ dz, const float* __restrict__ in,
                                                            float* __restrict__ dW, float* __restrict__ db) {
    extern __shared__ __attribute__((aligned(16))) float sm[];      // [8 waves][MT][256] accumulators | [8][MT * 16] row sums
    constexpr int O = MT * 16, NWV = 8, PF = 4;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, q = lane >> 4, j = lane & 15, f0 = blockIdx.x * 16;
    const int nks = (B + 3) / 4;                                      // k-blocks of four windows; wave w takes w, w + 8, ...
    f32x4 acc[MT];
    float bsum[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) { acc[m] = f32x4{0.f, 0.f, 0.f, 0.f}; bsum[m] = 0.f; }
    float av[PF][MT], bv[PF];
    auto fetch = [&](int u, int ks) {
        const int b = 4 * ks + q;
        const bool ok = ks < nks && b < B;
        const int bc = ok ? b : 0;
#pragma unroll
        for (int m = 0; m < MT; ++m) { const float v = dz[(size_t)bc * O + m * 16 + j]; av[u][m] = ok ? v : 0.f; }
        const float v = in[(size_t)bc * F + f0 + j];
        bv[u] = ok ? v : 0.f;
    };
#pragma unroll
    for (int u = 0; u < PF; ++u) fetch(u, wave + u * NWV);
    for (int ks0 = wave; ks0 < nks; ks0 += PF * NWV) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int ks = ks0 + u * NWV;
            if (ks < nks) {
#pragma unroll
                for (int m = 0; m < MT; ++m) { acc[m] = mfma_f32_16x16x4(av[u][m], bv[u], acc[m]); bsum[m] += av[u][m]; }
            }
            fetch(u, ks + PF * NWV);
        }
    }
    float* redb = sm + (size_t)NWV * MT * 256;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        *reinterpret_cast<f32x4*>(sm + ((size_t)(wave * MT + m) * 64 + lane) * 4) = acc[m];
        float v = bsum[m];
        v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
        if (q == 0) redb[wave * O + m * 16 + j] = v;
    }
    __syncthreads();
    for (int e = tid; e < MT * 256; e += 512) {
        const int m = e >> 8, l = (e & 255) >> 2, r = e & 3;         // accumulator (q, j), r  <->  row 4 q + r (o), column j (f)
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < NWV; ++w) v += sm[(size_t)(w * MT + m) * 256 + (e & 255)];
        dW[(size_t)(m * 16 + 4 * (l >> 4) + r) * F + f0 + (l & 15)] = v;
    }
    if (blockIdx.x == 0 && db) {
        for (int e = tid; e < O; e += 512) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < NWV; ++w) v += redb[w * O + e];
            db[e] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------ cross-entropy
// one block: loss = mean_b -log softmax(logits)[y_b]; dlogits = (softmax - onehot) / B    (nn.CrossEntropyLoss())
__global__ void __launch_bounds__(256) ce_loss_kernel(int B, int NC, const float* __restrict__ logits, const long long* __restrict__ labels,
                                                      float* __restrict__ dlogits, float* __restrict__ loss) {
    __shared__ float red[4];
    float ls = 0.f;
    for (int b = threadIdx.x; b < B; b += 256) {
        const float* l = logits + (size_t)b * NC;
        float mx = l[0];
        for (int c = 1; c < NC; ++c) mx = fmaxf(mx, l[c]);
        float se = 0.f;
        for (int c = 0; c < NC; ++c) se += expf(l[c] - mx);
        const float lse = mx + logf(se);
        const int yb = (int)labels[b];
        ls += lse - l[yb];
        if (dlogits)
            for (int c = 0; c < NC; ++c) dlogits[(size_t)b * NC + c] = (expf(l[c] - lse) - (c == yb ? 1.f : 0.f)) / (float)B;
    }
    const float tot = block_sum(ls, red);
    if (threadIdx.x == 0) *loss = tot / (float)B;
}

// class = argmax, p1 = softmax[:, 1]   (06_test_full_pipeline.py:367-371)
__global__ void __launch_bounds__(256) softmax_argmax_kernel(int B, int NC, const float* __restrict__ logits, long long* __restrict__ cls,
                                                             float* __restrict__ p1) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    const float* l = logits + (size_t)b * NC;
    float mx = l[0]; int am = 0;
    for (int c = 1; c < NC; ++c) if (l[c] > mx) { mx = l[c]; am = c; }
    float se = 0.f;
    for (int c = 0; c < NC; ++c) se += expf(l[c] - mx);
    if (cls) cls[b] = am;
    if (p1) p1[b] = NC > 1 ? expf(l[1] - mx) / se : 1.f;
}

// xin = stack([z, (z - zhat)^2], dim=1)   (05_train_cnn.py:137-138)
__global__ void __launch_bounds__(256) build_input_kernel(int B, int TD, const float* __restrict__ z, const float* __restrict__ zhat,
                                                          float* __restrict__ xin) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)B * TD) return;
    const size_t b = i / TD, r = i - b * TD;
    const float a = z[i], d = a - zhat[i];
    xin[(b * 2) * TD + r] = a;
    xin[(b * 2 + 1) * TD + r] = d * d;
}

// ------------------------------------------------------------------------------------------------ pool/ReLU/BN backward
// pass A (grid = B): per-channel sums S1 = sum da, S2 = sum da*xhat over this window, da = dp routed to the argmax cell.
// A wave per channel (channels w, w + 4, ...): wave-level sums only — the block-wide form paid two barriers per channel and sum.
__global__ void __launch_bounds__(256) bn_bwd_stats_kernel(int C, int H, int W, const float* __restrict__ dp, const unsigned char* __restrict__ idx,
                                                           const float* __restrict__ y, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           float* __restrict__ part, const float* __restrict__ xh) {
    const int Hp = H / 2, Wp = W / 2, b = blockIdx.x, npc = Hp * Wp, wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int c = wv; c < C; c += 4) {
        float s1 = 0.f, s2 = 0.f;
        const float m = mean[c], rs = rstd[c];
        for (int r = lane; r < npc; r += 64) {
            const size_t e = ((size_t)b * C + c) * npc + r;
            const int k = idx[e];
            if (k < 4) {
                float xv;
                if (xh) xv = xh[e];                                   // saved by the forward's pooling pass
                else {
                    const int hp = r / Wp, wp = r - hp * Wp;
                    xv = (y[((size_t)b * C + c) * H * W + (size_t)(2 * hp + (k >> 1)) * W + 2 * wp + (k & 1)] - m) * rs;
                }
                const float g = dp[e];
                s1 += g; s2 = fmaf(g, xv, s2);
            }
        }
        s1 = wave_sum(s1); s2 = wave_sum(s2);
        if (lane == 0) { part[((size_t)b * C + c) * 2] = s1; part[((size_t)b * C + c) * 2 + 1] = s2; }
    }
}
__global__ void __launch_bounds__(64) bn_bwd_finalize_kernel(int C, int B, long long n_per_channel, const float* __restrict__ part,
                                                             float* __restrict__ k1, float* __restrict__ k2, float* __restrict__ dgamma,
                                                             float* __restrict__ dbeta) {
    const int c = blockIdx.x;                                  // one wave per channel (see bn_finalize_kernel)
    double s, q;
    partial_pair_sums(part, C, c, B, threadIdx.x, s, q);
    s = wave_sum_f64(s); q = wave_sum_f64(q);
    if (threadIdx.x != 0) return;
    dbeta[c] = (float)s; dgamma[c] = (float)q;
    k1[c] = (float)(s / (double)n_per_channel); k2[c] = (float)(q / (double)n_per_channel);
}
// pass B (grid = B): dy = gamma*rstd * (da - k1 - xhat*k2) at EVERY conv-output position
__global__ void __launch_bounds__(256) bn_bwd_apply_kernel(int C, int H, int W, const float* __restrict__ dp, const unsigned char* __restrict__ idx,
                                                           const float* __restrict__ y, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const float* __restrict__ gamma, const float* __restrict__ k1, const float* __restrict__ k2,
                                                           float* __restrict__ dy) {
    const int Hp = H / 2, Wp = W / 2, b = blockIdx.x, HW = H * W;
    for (int c = 0; c < C; ++c) {                              // per-channel constants once per channel, one division per element
        const float m = mean[c], rs = rstd[c], sc = gamma[c] * rstd[c], c1 = k1[c], c2 = k2[c];
        const size_t cb = ((size_t)b * C + c) * HW, pb = ((size_t)b * C + c) * Hp * Wp;
        for (int r0 = threadIdx.x; r0 < HW; r0 += 4 * 256) {     // four elements per trip: their reads are issued together
            float yv[4], dv[4];
            int kk[4], sel[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int r = min(r0 + u * 256, HW - 1), h = r / W, w = r - h * W, hp = h >> 1, wp = w >> 1;
                const bool in = hp < Hp && wp < Wp;
                const size_t pe = pb + (in ? (size_t)hp * Wp + wp : 0);
                yv[u] = y[cb + r];
                kk[u] = in ? (int)idx[pe] : -1;
                dv[u] = dp[pe];
                sel[u] = (h & 1) * 2 + (w & 1);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int r = r0 + u * 256;
                if (r < HW) {
                    const float da = kk[u] == sel[u] ? dv[u] : 0.f;
                    const float xh = (yv[u] - m) * rs;
                    dy[cb + r] = sc * (da - c1 - xh * c2);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ conv backward
// data gradient: dx[ci][h][w] = sum_co sum_k dy[co][h+1-kh][w+1-kw] W[co][ci][kh][kw]      grid = B
// FUSE: dy is not read but formed while staging, from the pooled gradient and the saved conv output (bn_dy above) — the separate
// bn_bwd_apply launch and its dy tensor (written once, read by both conv gradients) are gone.
template <int CIN, int COUT, bool FUSE>
__global__ void __launch_bounds__(256) conv3x3_bwd_data_kernel(const float* __restrict__ dy, BnGrad bn, const float* __restrict__ w,
                                                               float* __restrict__ dx, int H, int W) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int HW = H * W, b = blockIdx.x, tid = threadIdx.x;
    float* ds = sm;                          // [COUT][HW]
    float* ws = sm + COUT * HW;              // [(co*9+k)][CIN]
    if constexpr (FUSE) {
        const int Hp = H / 2, Wp = W / 2, npc = Hp * Wp;
        for (int e0 = tid; e0 < COUT * HW; e0 += 256 * 8) {
            float yv[8], dv[8];
            int kk[8], sel[8], cc[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = min(e0 + u * 256, COUT * HW - 1), co = e / HW, p = e - co * HW, h = p / W, ww = p - h * W, hp = h >> 1, wp = ww >> 1;
                const bool in = hp < Hp && wp < Wp;
                const size_t pe = ((size_t)b * COUT + co) * npc + (in ? hp * Wp + wp : 0);
                yv[u] = bn.y[(size_t)b * COUT * HW + e];
                kk[u] = in ? (int)bn.idx[pe] : -1;
                dv[u] = bn.dp[pe];
                sel[u] = (h & 1) * 2 + (ww & 1);
                cc[u] = co;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = e0 + u * 256;
                if (e < COUT * HW) ds[e] = bn_dy(bn, cc[u], yv[u], kk[u] == sel[u] ? dv[u] : 0.f);
            }
        }
    } else {
        stage_elems<256>(COUT * HW, tid, [&](int e) { return dy[(size_t)b * COUT * HW + e]; }, [&](int e, float v) { ds[e] = v; });
    }
    stage_elems<256>(COUT * CIN * 9, tid, [&](int e) { return w[e]; }, [&](int e, float v) {   // w[co][ci][k] -> ws[(co*9+k)*CIN + ci]
        const int co = e / (CIN * 9), r = e - co * CIN * 9, ci = r / 9, k = r - ci * 9;
        ws[(co * 9 + k) * CIN + ci] = v;
    });
    __syncthreads();
    for (int p = tid; p < HW; p += 256) {
        const int h = p / W, ww = p - h * W;
        float acc[CIN];
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci) acc[ci] = 0.f;
        for (int co = 0; co < COUT; ++co) {
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const int hh = h + 1 - kh, w2 = ww + 1 - kw;
                    const float v = (hh >= 0 && hh < H && w2 >= 0 && w2 < W) ? ds[co * HW + hh * W + w2] : 0.f;
                    const float* wr = ws + (co * 9 + kh * 3 + kw) * CIN;
#pragma unroll
                    for (int ci = 0; ci < CIN; ci += 4) {
                        const f32x4 w4 = *reinterpret_cast<const f32x4*>(wr + ci);
                        acc[ci] = fmaf(v, w4[0], acc[ci]); acc[ci + 1] = fmaf(v, w4[1], acc[ci + 1]);
                        acc[ci + 2] = fmaf(v, w4[2], acc[ci + 2]); acc[ci + 3] = fmaf(v, w4[3], acc[ci + 3]);
                    }
                }
        }
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci) dx[((size_t)b * CIN + ci) * HW + p] = acc[ci];
    }
}

// weight gradient partial of one window: part[b][co][ci][k] = sum_p dy[co][p] x[ci][p + off(k)], part[b][W_n + co] = sum_p dy[co][p]
template <int CIN, int COUT>
__global__ void __launch_bounds__(256) conv3x3_bwd_weight_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ part,
                                                                 int H, int W) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int HW = H * W, b = blockIdx.x, tid = threadIdx.x;
    constexpr int NW_ = COUT * CIN * 9;
    const int XS = HW + 1;           // odd row stride: the lanes of a wave walk different ci rows -> different banks
    float* ds = sm;                  // [COUT][HW]
    float* xs = sm + COUT * HW;      // [CIN][HW + 1]
    for (int e = tid; e < COUT * HW; e += 256) ds[e] = dy[(size_t)b * COUT * HW + e];
    for (int e = tid; e < CIN * HW; e += 256) { const int ci = e / HW; xs[ci * XS + (e - ci * HW)] = x[(size_t)b * CIN * HW + e]; }
    __syncthreads();
    float* out = part + (size_t)b * (NW_ + COUT);
    // one (co, ci) pair per thread and pass: the 9 taps share every dy value and a 3-row window of x, so a tap costs about one
    // LDS read instead of two (the one-tap-per-thread form spent 312 us per launch on conv2 at B = 512).  With fewer pairs than
    // threads (conv1: 32) the rows are dealt over SL slices per pair and the slice sums combined in a fixed order.
    constexpr int PAIRS = COUT * CIN, SL = PAIRS >= 256 ? 1 : 256 / PAIRS;
    float* red = xs + CIN * XS;      // [SL][PAIRS][9] (SL > 1 only)
    for (int e0 = 0; e0 < PAIRS; e0 += 256 / SL) {
        const int e = e0 + tid % (256 / SL), slice = tid / (256 / SL);
        float acc[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) acc[k] = 0.f;
        if (e < PAIRS) {
            const int co = e / CIN, ci = e - co * CIN;
            const float* dr = ds + co * HW;
            const float* xr = xs + ci * XS;
            for (int h = slice; h < H; h += SL) {
                const bool up = h > 0, dn = h + 1 < H;
                for (int w2 = 0; w2 < W; ++w2) {
                    const float d = dr[h * W + w2];
                    const bool lf = w2 > 0, rt = w2 + 1 < W;
                    const float* xc = xr + h * W + w2;
                    // tap k = (dh+1)*3 + (dw+1) pairs dy[h][w] with x[h+dh][w+dw]
                    if (up) { if (lf) acc[0] = fmaf(d, xc[-W - 1], acc[0]); acc[1] = fmaf(d, xc[-W], acc[1]); if (rt) acc[2] = fmaf(d, xc[-W + 1], acc[2]); }
                    if (lf) acc[3] = fmaf(d, xc[-1], acc[3]);
                    acc[4] = fmaf(d, xc[0], acc[4]);
                    if (rt) acc[5] = fmaf(d, xc[1], acc[5]);
                    if (dn) { if (lf) acc[6] = fmaf(d, xc[W - 1], acc[6]); acc[7] = fmaf(d, xc[W], acc[7]); if (rt) acc[8] = fmaf(d, xc[W + 1], acc[8]); }
                }
            }
            if (SL == 1) {
#pragma unroll
                for (int k = 0; k < 9; ++k) out[e * 9 + k] = acc[k];
            } else {
#pragma unroll
                for (int k = 0; k < 9; ++k) red[(slice * PAIRS + e) * 9 + k] = acc[k];
            }
        }
    }
    if (SL > 1) {
        __syncthreads();
        for (int i = tid; i < PAIRS * 9; i += 256) {
            float v = 0.f;
            for (int sl = 0; sl < SL; ++sl) v += red[sl * PAIRS * 9 + i];
            out[i] = v;
        }
    }
    for (int e = tid; e < COUT; e += 256) {               // bias gradient of this window
        const float* dr = ds + e * HW;
        float acc = 0.f;
        for (int p = 0; p < HW; ++p) acc += dr[p];
        out[NW_ + e] = acc;
    }
}

// The same per-window partial on the matrix pipe, exact fp32 (v_mfma_f32_16x16x4_f32): part[co][n] = sum_p dy[co][p] * col[p][n] with
// n = ci * 9 + tap and col = the im2col view of x (never materialised): M = 16 output channels, N = 16 (ci, tap) columns, K = positions.
// dy rows are position-contiguous, so a lane's four k-steps are ONE 16-byte global read (a rotating register ring keeps four
// k-blocks in flight; no LDS copy of dy); x sits in LDS with a zero halo and a per-position offset table, so a column value is one
// 4-byte LDS read at table[p] + (ci, tap offset).  With >= 8 output tiles (conv2: 2 x 9) the waves take different tiles; otherwise
// (conv1: 1 x 2) they split the positions and their accumulators are added in wave order.  The bias gradient (row sums of dy) falls
// out of the A fragments: every lane adds its four values per k-block, the four lane quarters are combined by shuffles.
// Requires H * W % 4 == 0 (16-byte reads of dy rows).
// FUSE: the A operand dy is formed on the fly (bn_dy): a lane reads 16 bytes of the saved conv output y, looks its four positions' pooled
// cells up in an LDS table (pref: cell << 2 | position within the 2x2 cell) and the cells' (gradient, argmax) records in an LDS copy of
// the window's pooled gradient — the bn_bwd_apply launch and the dy tensor are gone.
template <int CIN, int COUT, bool FUSE>
__global__ void __launch_bounds__(512) conv3x3_bwd_weight_mfma_kernel(const float* __restrict__ dy, BnGrad bn, const float* __restrict__ x,
                                                                      float* __restrict__ part, int H, int W) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int NCOL = CIN * 9, MT = COUT / 16, NT = (NCOL + 15) / 16, NWV = 8, NW_ = COUT * NCOL, PF = 4;
    constexpr bool TILE_PAR = MT * NT >= NWV;
    const int HW = H * W, KP = (HW + 15) & ~15, NKB = KP / 16, WP = W + 2, XP = (H + 2) * WP, XPP = (CIN * XP + 3) & ~3;
    const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, q = lane >> 4, j = lane & 15;
    const int Hp = H / 2, Wp = W / 2, npc = Hp * Wp, NPC1 = npc + 1;
    float* xs = sm;                                           // [CIN][H + 2][W + 2], zero halo
    int* posoff = reinterpret_cast<int*>(xs + XPP);           // [KP]: (h + 1) * WP + (w + 1) of position p
    int* pref = posoff + KP;                                  // FUSE: [KP] pooled cell << 2 | (h & 1) * 2 + (w & 1); cell npc = "no cell"
    f32x2* pc = reinterpret_cast<f32x2*>(pref + (FUSE ? KP : 0));   // FUSE: [COUT][npc + 1] (pooled gradient, argmax as int bits)
    float* red = reinterpret_cast<float*>(pc + (FUSE ? (size_t)COUT * NPC1 : 0));   // position-split only: [NWV][MT * NT][256] accumulators | [NWV][COUT] row sums
    if constexpr (FUSE) {
        for (int p = tid; p < KP; p += 512) {
            const int h = p / W, w2 = p - h * W, hp = h >> 1, wp = w2 >> 1;
            pref[p] = ((p < HW && hp < Hp && wp < Wp) ? hp * Wp + wp : npc) << 2 | ((h & 1) * 2 + (w2 & 1));
        }
        for (int e0 = tid; e0 < COUT * NPC1; e0 += 512 * 4) {
            float dv[4]; int kv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = min(e0 + u * 512, COUT * NPC1 - 1), co = e / NPC1, cell = e - co * NPC1;
                const size_t pe = ((size_t)b * COUT + co) * npc + (cell < npc ? cell : 0);
                dv[u] = bn.dp[pe]; kv[u] = cell < npc ? (int)bn.idx[pe] : 4;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (e0 + u * 512 < COUT * NPC1) pc[e0 + u * 512] = f32x2{dv[u], __int_as_float(kv[u])};
        }
    }
    // eight loads in flight per thread: issued one per trip, every global read would wait out its full latency before the next
    for (int e0 = tid; e0 < CIN * XP; e0 += 512 * 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = e0 + u * 512, ec = e < CIN * XP ? e : 0;
            const int ci = ec / XP, r = ec - ci * XP, hh = r / WP - 1, ww = r - (hh + 1) * WP - 1;
            v[u] = (hh >= 0 && hh < H && ww >= 0 && ww < W) ? x[((size_t)b * CIN + ci) * HW + hh * W + ww] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (e0 + u * 512 < CIN * XP) xs[e0 + u * 512] = v[u];
    }
    for (int p = tid; p < KP; p += 512) posoff[p] = p < HW ? (p / W + 1) * WP + (p % W + 1) : WP + 1;   // padding: any valid cell (dy is 0 there)
    __syncthreads();
    float* out = part + (size_t)b * (NW_ + COUT);
    const float* dyb = dy + (size_t)b * COUT * HW;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    auto column = [&](int nt, int& boff, float& keep) {       // this lane's column n = nt * 16 + j of the im2col view
        const int n = nt * 16 + j, ok = n < NCOL, nn = ok ? n : 0, ci = nn / 9, tap = nn - ci * 9;
        boff = ci * XP + (tap / 3 - 1) * WP + (tap % 3 - 1);
        keep = ok ? 1.f : 0.f;
    };
    // this lane's dy rows are channels m * 16 + j: their BatchNorm-backward constants (FUSE)
    float cm[MT], crs[MT], csc[MT], ck1[MT], ck2[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        if constexpr (FUSE) {
            const int c = m * 16 + j;
            cm[m] = bn.mean[c]; crs[m] = bn.rstd[c]; csc[m] = bn.gamma[c] * bn.rstd[c]; ck1[m] = bn.k1[c]; ck2[m] = bn.k2[c];
        } else { cm[m] = crs[m] = csc[m] = ck1[m] = ck2[m] = 0.f; }
    }
    // positions kb*16 + 4q .. +3 of this lane's dy row of M-tile m (zero past the window); arow: the row in dy, or in y (FUSE).
    // Two halves so that the ring really prefetches: a_issue starts the 16-byte global read (and, FUSE, resolves the four pooled
    // gradients from LDS); a_value, called when the k-block is consumed, is the first use of the global data (FUSE: the BatchNorm
    // backward formula).  With the formula inside the issue half every k-block waited out its own global read: 46 vs 20 us (conv1).
    struct ARaw { f32x4 v; f32x4 da; };
    auto a_issue = [&](const float* arow, int kb, int m) -> ARaw {
        const int kk = kb < NKB ? kb : NKB - 1, p = kk * 16 + 4 * q;
        ARaw r;
        r.v = *reinterpret_cast<const f32x4*>(arow + (p < HW ? p : 0));
        r.da = zero4;
        if constexpr (FUSE) {
            const int4 pr = *reinterpret_cast<const int4*>(pref + kk * 16 + 4 * q);
            const f32x2* pcr = pc + (size_t)(m * 16 + j) * NPC1;
            const int prs[4] = {pr.x, pr.y, pr.z, pr.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const f32x2 rec = pcr[prs[e] >> 2];
                r.da[e] = __float_as_int(rec[1]) == (prs[e] & 3) ? rec[0] : 0.f;
            }
        }
        return r;
    };
    auto a_value = [&](const ARaw& r, int kb, int m) -> f32x4 {
        f32x4 v = r.v;
        if constexpr (FUSE) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = csc[m] * (r.da[e] - ck1[m] - (v[e] - cm[m]) * crs[m] * ck2[m]);
        }
        return kb * 16 + 4 * q < HW ? v : zero4;
    };
    const float* arows = FUSE ? bn.y + (size_t)b * COUT * HW : dyb;
    if constexpr (TILE_PAR) {
        constexpr int G = NWV / MT, NTW = (NT + G - 1) / G;
        const int mt = wave % MT, g = wave / MT, nt0 = (g * NT) / G, ntn = ((g + 1) * NT) / G - nt0;
        int boff[NTW]; float keep[NTW]; f32x4 acc[NTW];
#pragma unroll
        for (int i = 0; i < NTW; ++i) { column(nt0 + (i < ntn ? i : 0), boff[i], keep[i]); acc[i] = zero4; }
        const float* arow = arows + (size_t)(mt * 16 + j) * HW;
        ARaw ar[PF];
#pragma unroll
        for (int u = 0; u < PF; ++u) ar[u] = a_issue(arow, u, mt);
        float bsum = 0.f;
        for (int kb0 = 0; kb0 < NKB; kb0 += PF) {
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const int kb = kb0 + u;
                if (kb < NKB) {
                    const f32x4 a = a_value(ar[u], kb, mt);
                    bsum += (a[0] + a[1]) + (a[2] + a[3]);
                    const int* pp = posoff + kb * 16 + 4 * q;
                    const int p0 = pp[0], p1 = pp[1], p2 = pp[2], p3 = pp[3];
#pragma unroll
                    for (int i = 0; i < NTW; ++i)
                        if (i < ntn) {
                            const float* xb = xs + boff[i];
                            acc[i] = mfma_f32_16x16x4(a[0], xb[p0] * keep[i], acc[i]);
                            acc[i] = mfma_f32_16x16x4(a[1], xb[p1] * keep[i], acc[i]);
                            acc[i] = mfma_f32_16x16x4(a[2], xb[p2] * keep[i], acc[i]);
                            acc[i] = mfma_f32_16x16x4(a[3], xb[p3] * keep[i], acc[i]);
                        }
                }
                ar[u] = a_issue(arow, kb + PF, mt);
            }
        }
#pragma unroll
        for (int i = 0; i < NTW; ++i) {
            const int n = (nt0 + i) * 16 + j;
            if (i < ntn && n < NCOL) {
#pragma unroll
                for (int r = 0; r < 4; ++r) out[(size_t)(mt * 16 + 4 * q + r) * NCOL + n] = acc[i][r];
            }
        }
        bsum += __shfl_xor(bsum, 16, 64); bsum += __shfl_xor(bsum, 32, 64);
        if (g == 0 && q == 0) out[NW_ + mt * 16 + j] = bsum;            // bias gradient of this window
    } else {
        constexpr int TILES = MT * NT;
        int boff[NT]; float keep[NT]; f32x4 acc[TILES];
        float bsum[MT];
#pragma unroll
        for (int i = 0; i < NT; ++i) column(i, boff[i], keep[i]);
#pragma unroll
        for (int i = 0; i < TILES; ++i) acc[i] = zero4;
#pragma unroll
        for (int m = 0; m < MT; ++m) bsum[m] = 0.f;
        // wave w takes k-blocks w, w + 8, ...; the ring holds the next PF of them
        ARaw ar[PF][MT];
#pragma unroll
        for (int u = 0; u < PF; ++u)
#pragma unroll
            for (int m = 0; m < MT; ++m) ar[u][m] = a_issue(arows + (size_t)(m * 16 + j) * HW, wave + u * NWV, m);
        for (int kb0 = wave; kb0 < NKB; kb0 += PF * NWV) {
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const int kb = kb0 + u * NWV;
                if (kb < NKB) {
                    const int* pp = posoff + kb * 16 + 4 * q;
                    const int p0 = pp[0], p1 = pp[1], p2 = pp[2], p3 = pp[3];
                    float bv[NT][4];
#pragma unroll
                    for (int i = 0; i < NT; ++i) {
                        const float* xb = xs + boff[i];
                        bv[i][0] = xb[p0] * keep[i]; bv[i][1] = xb[p1] * keep[i]; bv[i][2] = xb[p2] * keep[i]; bv[i][3] = xb[p3] * keep[i];
                    }
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
                        const f32x4 a = a_value(ar[u][m], kb, m);
                        bsum[m] += (a[0] + a[1]) + (a[2] + a[3]);
#pragma unroll
                        for (int i = 0; i < NT; ++i)
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[m * NT + i] = mfma_f32_16x16x4(a[e], bv[i][e], acc[m * NT + i]);
                    }
                }
#pragma unroll
                for (int m = 0; m < MT; ++m) ar[u][m] = a_issue(arows + (size_t)(m * 16 + j) * HW, kb + PF * NWV, m);
            }
        }
        float* redb = red + (size_t)NWV * TILES * 256;          // [NWV][COUT]
#pragma unroll
        for (int i = 0; i < TILES; ++i) *reinterpret_cast<f32x4*>(red + ((size_t)(wave * TILES + i) * 64 + lane) * 4) = acc[i];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            float v = bsum[m];
            v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
            if (q == 0) redb[wave * COUT + m * 16 + j] = v;
        }
        __syncthreads();
        for (int e = tid; e < TILES * 256; e += 512) {
            const int tile = e >> 8, l = (e & 255) >> 2, r = e & 3, m = tile / NT, nt = tile - m * NT;
            const int n = nt * 16 + (l & 15);
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < NWV; ++w) v += red[(size_t)(w * TILES + tile) * 256 + (e & 255)];
            if (n < NCOL) out[(size_t)(m * 16 + 4 * (l >> 4) + r) * NCOL + n] = v;
        }
        for (int e = tid; e < COUT; e += 512) {                  // bias gradient of this window
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < NWV; ++w) v += redb[w * COUT + e];
            out[NW_ + e] = v;
        }
    }
}

// Weight gradient of a layer with many output tiles (conv2: 2 x 9), BatchNorm backward fused, K-SPLIT form.  The tile-parallel form above
// gives every wave a few column tiles and ALL positions, so the four waves that share an output-channel tile each rebuild the same A
// operand (the fused dy: ~50 VALU per k-block) — the kernel was VALU-issue-bound (3400 VALU per wave for 144 MFMAs; 44 us at B = 512).
// (HW % 4 == 0, so a lane's four positions are all inside the window or all outside.)
// Here wave (mt, ks) owns output-channel tile mt, ALL NT column tiles (NT accumulators) and every KSL-th k-block: each dy value is formed
// exactly once per workgroup, every B value read feeds one MFMA as before, and the KSL partial sums meet in a fixed-order tree through
// LDS (the staged operands' space is reused once every wave is done with them).  Requires CIN * 9 % 16 == 0, H * W % 4 == 0.
template <int CIN, int COUT>
__global__ void __launch_bounds__(512, 4) conv3x3_bwd_weight_ksplit_kernel(BnGrad bn, const float* __restrict__ x, float* __restrict__ part,
                                                                           float* __restrict__ dy_out, int H, int W) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int NCOL = CIN * 9, MT = COUT / 16, NT = NCOL / 16, NWV = 8, KSL = NWV / MT, NW_ = COUT * NCOL;
    static_assert(NCOL % 16 == 0 && KSL == 4, "conv2-shaped layers only");
    const int HW = H * W, KP = (HW + 15) & ~15, NKB = KP / 16, WP = W + 2, XP = (H + 2) * WP, XPP = (CIN * XP + 3) & ~3;
    const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, q = lane >> 4, j = lane & 15;
    const int Hp = H / 2, Wp = W / 2, npc = Hp * Wp, NPC1 = npc + 1;
    const int mt = wave % MT, ks = wave / MT;
    float* xs = sm;                                           // [CIN][H + 2][W + 2], zero halo
    int* posoff = reinterpret_cast<int*>(xs + XPP);           // [KP]: (h + 1) * WP + (w + 1) of position p
    int* pref = posoff + KP;                                  // [KP] pooled cell << 2 | (h & 1) * 2 + (w & 1); cell npc = "no cell"
    f32x2* pc = reinterpret_cast<f32x2*>(pref + KP);          // [COUT][npc + 1] (pooled gradient, argmax as int bits)
    // [KSL][COUT] row sums (bias gradient): behind the operands AND behind the exchange buffer that later overlays them (small windows)
    const size_t opnd = (size_t)XPP + 2 * (size_t)KP + 2 * (size_t)COUT * NPC1, xchg = (size_t)2 * MT * NT * 256;
    float* redb = sm + (opnd > xchg ? opnd : xchg);
    for (int e = tid; e < CIN * XP; e += 512) xs[e] = 0.f;
    for (int p = tid; p < KP; p += 512) {
        const int h = p / W, w2 = p - h * W, hp = h >> 1, wp = w2 >> 1;
        posoff[p] = p < HW ? (h + 1) * WP + (w2 + 1) : WP + 1;            // padding: any valid cell (dy is 0 there)
        pref[p] = ((p < HW && hp < Hp && wp < Wp) ? hp * Wp + wp : npc) << 2 | ((h & 1) * 2 + (w2 & 1));
    }
    __syncthreads();
    {
        // interior of the halo'd window: coalesced reads, eight in flight, (channel, position) advanced without dividing
        int ci = tid / HW, p = tid - ci * HW;
        for (int e0 = tid; e0 < CIN * HW; e0 += 512 * 8) {
            float v[8]; int dst[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int cu = ci < CIN ? ci : CIN - 1;
                v[u] = x[((size_t)b * CIN + cu) * HW + p];
                dst[u] = cu * XP + posoff[p];
                p += 512;
                while (p >= HW) { p -= HW; ++ci; }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (e0 + u * 512 < CIN * HW) xs[dst[u]] = v[u];
        }
        int co = tid / NPC1, cell = tid - co * NPC1;
        for (int e0 = tid; e0 < COUT * NPC1; e0 += 512 * 4) {
            float dv[4]; int kv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int cu = co < COUT ? co : COUT - 1;
                const size_t pe = ((size_t)b * COUT + cu) * npc + (cell < npc ? cell : 0);
                dv[u] = bn.dp[pe]; kv[u] = cell < npc ? (int)bn.idx[pe] : 4;
                cell += 512;
                while (cell >= NPC1) { cell -= NPC1; ++co; }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (e0 + u * 512 < COUT * NPC1) pc[e0 + u * 512] = f32x2{dv[u], __int_as_float(kv[u])};
        }
    }
    // this lane's A row is channel mt * 16 + j; its B columns are n = nt * 16 + j -> (ci, tap)
    const int c = mt * 16 + j;
    const float cm = bn.mean[c], crs = bn.rstd[c], csc = bn.gamma[c] * crs, ck1 = bn.k1[c], ck2 = bn.k2[c];
    int boff[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) { const int n = nt * 16 + j, ci = n / 9, tap = n - ci * 9; boff[nt] = ci * XP + (tap / 3 - 1) * WP + (tap % 3 - 1); }
    const float* arow = bn.y + ((size_t)b * COUT + c) * HW;
    const f32x2* pcr = pc + (size_t)c * NPC1;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = zero4;
    float bsum = 0.f;
    __syncthreads();
    // the wave's k-blocks: ks, ks + KSL, ...; the global halves of up to PF of them in flight (see a_issue / a_value above)
    constexpr int PF = 4;
    f32x4 yv[PF];
    auto y_load = [&](int kb) -> f32x4 { const int kk = kb < NKB ? kb : NKB - 1, p = kk * 16 + 4 * q; return *reinterpret_cast<const f32x4*>(arow + (p < HW ? p : 0)); };
#pragma unroll
    for (int u = 0; u < PF; ++u) yv[u] = y_load(ks + u * KSL);
    for (int kb0 = ks; kb0 < NKB; kb0 += PF * KSL) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int kb = kb0 + u * KSL;
            if (kb < NKB) {
                const int4 pr = *reinterpret_cast<const int4*>(pref + kb * 16 + 4 * q);
                const int4 po = *reinterpret_cast<const int4*>(posoff + kb * 16 + 4 * q);
                const int prs[4] = {pr.x, pr.y, pr.z, pr.w}, pos[4] = {po.x, po.y, po.z, po.w};
                f32x4 a;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const f32x2 rec = pcr[prs[e] >> 2];
                    const float da = __float_as_int(rec[1]) == (prs[e] & 3) ? rec[0] : 0.f;
                    a[e] = kb * 16 + 4 * q + e < HW ? csc * (da - ck1 - (yv[u][e] - cm) * crs * ck2) : 0.f;
                }
                bsum += (a[0] + a[1]) + (a[2] + a[3]);
                // every dy value exists exactly once in this grid: leave it for the data-gradient kernel (16 bytes per lane, coalesced
                // per row), which then stages plain values instead of redoing the pooled-gradient gathers and the formula
                if (dy_out && kb * 16 + 4 * q < HW) *reinterpret_cast<f32x4*>(dy_out + ((size_t)b * COUT + c) * HW + kb * 16 + 4 * q) = a;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float bv[NT];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) bv[nt] = xs[boff[nt] + pos[e]];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[nt] = mfma_f32_16x16x4(a[e], bv[nt], acc[nt]);
                }
            }
            yv[u] = y_load(kb + PF * KSL);
        }
    }
    bsum += __shfl_xor(bsum, 16, 64); bsum += __shfl_xor(bsum, 32, 64);
    __syncthreads();                                           // every wave is done with xs / pc: their space becomes the exchange buffer
    f32x4* red = reinterpret_cast<f32x4*>(sm);                 // [2][MT][NT][64] accumulators
    if (q == 0) redb[ks * COUT + c] = bsum;
    if (ks >= 2) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) red[((size_t)((ks - 2) * MT + mt) * NT + nt) * 64 + lane] = acc[nt];
    }
    __syncthreads();
    if (ks < 2) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] += red[((size_t)(ks * MT + mt) * NT + nt) * 64 + lane];      // slice 0 += 2, slice 1 += 3
    }
    __syncthreads();
    if (ks == 1) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) red[((size_t)mt * NT + nt) * 64 + lane] = acc[nt];
    }
    __syncthreads();
    float* out = part + (size_t)b * (NW_ + COUT);
    if (ks == 0) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const f32x4 v = acc[nt] + red[((size_t)mt * NT + nt) * 64 + lane];                                  // (0 + 2) + (1 + 3)
#pragma unroll
            for (int r = 0; r < 4; ++r) out[(size_t)(mt * 16 + 4 * q + r) * NCOL + nt * 16 + j] = v[r];
        }
    }
    for (int e = tid; e < COUT; e += 512) out[NW_ + e] = (redb[e] + redb[2 * COUT + e]) + (redb[COUT + e] + redb[3 * COUT + e]);
}

template <typename K>
int opt_in_lds(K kern, size_t lds) {
    SHM_REQUIRE(lds <= 160 * 1024, SHM_EINVAL, "cnn: window needs %zu bytes of LDS (> 160 KiB): T*D too large", lds);
    if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    return SHM_OK;
}

}  // namespace

// =================================================================================================== launchers
// LDS bytes of conv3x3_fwd_mfma_kernel: halo'd window | statistics rows | position -> halo'd offset table
static size_t conv_mfma_lds(int cin, int cout, int H, int W) {
    const size_t XP = (size_t)(H + 2) * (W + 2);
    return sizeof(float) * (((cin * XP + 3) & ~(size_t)3) + 8 * (size_t)cout * 2 + (((size_t)H * W + 3) & ~(size_t)3));
}
template <int CIN, int COUT>
int conv_fwd_mfma(int B, int H, int W, const float* x, const float* w, const float* bias, float* y, float* part, hipStream_t s) {
    const size_t lds = conv_mfma_lds(CIN, COUT, H, W);
    int rc;
    if (part) { if ((rc = opt_in_lds(conv3x3_fwd_mfma_kernel<CIN, COUT, true>, lds))) return rc;
                SHM_LAUNCH((conv3x3_fwd_mfma_kernel<CIN, COUT, true>), dim3(B), dim3(512), lds, s, x, w, bias, y, part, H, W); }
    else { if ((rc = opt_in_lds(conv3x3_fwd_mfma_kernel<CIN, COUT, false>, lds))) return rc;
           SHM_LAUNCH((conv3x3_fwd_mfma_kernel<CIN, COUT, false>), dim3(B), dim3(512), lds, s, x, w, bias, y, part, H, W); }
    SHM_LAUNCH_CHECK();
    return SHM_OK;
}
int cnn_conv_fwd(int layer, int B, int H, int W, const float* x, const float* w, const float* bias, float* y, float* part, hipStream_t s) {
    const int HW = H * W;
    int rc;
    // Both convolutions run on the matrix pipe.  conv1 (2 -> 16 channels, memory-bound): 16 vs 29 us at the north-star shape.  conv2:
    // 23 vs 40 us for the vector kernel below at B = 512 since round 2 (B reads issued in blocks, two position tiles in flight;
    // before that the two tied).  SHM_CNN_VALU_CONV: measurement knob.
    if (!getenv("SHM_CNN_VALU_CONV")) {
        if (layer == 1) return conv_fwd_mfma<2, 16>(B, H, W, x, w, bias, y, part, s);
        return conv_fwd_mfma<16, 32>(B, H, W, x, w, bias, y, part, s);
    }
    if (layer == 1) {
        const size_t lds = sizeof(float) * (((2 * HW + 3) & ~3) + 2 * 9 * 16 + 8 * 16);
        if (part) { if ((rc = opt_in_lds(conv3x3_fwd_kernel<2, 16, true>, lds))) return rc;
                    SHM_LAUNCH((conv3x3_fwd_kernel<2, 16, true>), dim3(B), dim3(256), lds, s, x, w, bias, y, part, H, W); }
        else { if ((rc = opt_in_lds(conv3x3_fwd_kernel<2, 16, false>, lds))) return rc;
               SHM_LAUNCH((conv3x3_fwd_kernel<2, 16, false>), dim3(B), dim3(256), lds, s, x, w, bias, y, part, H, W); }
    } else {
        const size_t lds = sizeof(float) * (((16 * HW + 3) & ~3) + 16 * 9 * 32 + 8 * 32);
        if (part) { if ((rc = opt_in_lds(conv3x3_fwd_kernel<16, 32, true>, lds))) return rc;
                    SHM_LAUNCH((conv3x3_fwd_kernel<16, 32, true>), dim3(B), dim3(256), lds, s, x, w, bias, y, part, H, W); }
        else { if ((rc = opt_in_lds(conv3x3_fwd_kernel<16, 32, false>, lds))) return rc;
               SHM_LAUNCH((conv3x3_fwd_kernel<16, 32, false>), dim3(B), dim3(256), lds, s, x, w, bias, y, part, H, W); }
    }
    SHM_LAUNCH_CHECK();
    return SHM_OK;
}
int cnn_bn_finalize(int C, int B, long long n, int train, const float* part, const float* gamma, const float* beta, float* run_mean,
                    float* run_var, float* mean, float* rstd, float* scale, float* shift, hipStream_t s) {
    SHM_LAUNCH(bn_finalize_kernel, dim3(C), dim3(64), 0, s, C, B, n, train, part, gamma, beta, run_mean, run_var, mean, rstd, scale, shift);
    SHM_LAUNCH_CHECK();
    return SHM_OK;
}
int cnn_bn_relu_pool(int B, int C, int H, int W, const float* y, const float* scale, const float* shift, float* out, unsigned char* idx, hipStream_t s,
                     const float* mean, const float* rstd, float* xh) {
    SHM_LAUNCH(bn_relu_pool_kernel, dim3(B), dim3(256), 0, s, C, H, W, y, scale, shift, out, idx, mean, rstd, xh);
    SHM_LAUNCH_CHECK();
    return SHM_OK;
}
int cnn_linear_fwd(int B, int F, int O, const float* in, const float* Wt, const float* bias, float* pre, float* out, int relu, float drop_p,
                   uint64_t seed, hipStream_t s) {
    const bool aligned = ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(Wt)) & 15) == 0;
    if (O >= 16 && F % 16 == 0 && aligned && !getenv("SHM_CNN_VALU_DENSE"))
        SHM_LAUNCH(linear_fwd_mfma_kernel, dim3(ceil_div(O, 16), ceil_div(B, 16)), dim3(256), 0, s, B, F, O, in, Wt, bias, pre, out, relu, drop_p,
                   (unsigned long long)seed);
    else
        SHM_LAUNCH(linear_fwd_kernel, dim3(ceil_div(O, 32), ceil_div(B, 8)), dim3(256), 0, s, B, F, O, in, Wt, bias, pre, out, relu, drop_p,
                   (unsigned long long)seed);
    SHM_LAUNCH_CHECK();
    return SHM_OK;
}
int cnn_linear_bwd_dx(int B, int F, int O, const float* dy, const float* Wt, const float* pre, float* dx, float drop_p, uint64_t seed, hipStream_t s) {
    const size_t n = (size_t)B * F;
    if (O >= 16 && O % 16 == 0 && (reinterpret_cast<uintptr_t>(dy) & 15) == 0 && !getenv("SHM_CNN_VALU_DENSE"))
        SHM_LAUNCH(linear_bwd_dx_mfma_kernel, dim3(ceil_div(F, 64), ceil_div(B, 16)), dim3(256), 0, s, B, F, O, dy, Wt, pre, dx, drop_p, (unsigned long long)seed);
    else
        SHM_LAUNCH(linear_bwd_dx_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, B, F, O, dy, Wt, pre, dx, drop_p, (unsigned long long)seed);
    SHM_LAUNCH_CHECK();
    return SHM_OK;
}
size_t cnn_dense_out_bwd_partial_elems(int B, int F1, int NC) { return (size_t)ceil_div(B, 16) * ((size_t)NC * F1 + NC); }
int cnn_dense_out_bwd(int B, int F1, int NC, const float* dl, const float* W2, const float* z1, const float* a1, float* dz1, float* part,
                      float* dW2, float* db2, float drop_p, uint64_t seed, hipStream_t s, ReduceJobs& jobs) {
    SHM_REQUIRE(F1 >= 16 && F1 <= 256 && NC >= 1 && NC <= 16, SHM_EINVAL, "cnn dense_out_bwd: F1 %d / NC %d unsupported", F1, NC);
    const int G = ceil_div(B, 16);
    if (NC <= 4) {
        const size_t lds = sizeof(float) * ((size_t)F1 * 4 + 16 * 4);
        SHM_LAUNCH((dense_out_bwd_kernel<4>), dim3(G), dim3(2 * F1), lds, s, B, F1, NC, dl, W2, z1, a1, dz1, part, drop_p, (unsigned long long)seed);
    } else {
        const size_t lds = sizeof(float) * ((size_t)F1 * 16 + 16 * 16);
        SHM_LAUNCH((dense_out_bwd_kernel<16>), dim3(G), dim3(2 * F1), lds, s, B, F1, NC, dl, W2, z1, a1, dz1, part, drop_p, (unsigned long long)seed);
    }
    SHM_LAUNCH_CHECK();
    const size_t nw = (size_t)NC * F1, n = nw + NC;
    int rc;
    if ((rc = reduce_jobs_add(jobs, part, G, nw, n, dW2, nullptr))) return rc;
    return reduce_jobs_add(jobs, part + nw, G, (size_t)NC, n, db2, nullptr);
}
bool cnn_dense_dw_direct_ok(int F, int O) { return O == 128 && F % 16 == 0 && !getenv("SHM_CNN_SPLITK_DENSE"); }
int cnn_dense_dw(int B, int F, int O, const float* dz, const float* in, float* dW, float* db, hipStream_t s) {
    SHM_REQUIRE(cnn_dense_dw_direct_ok(F, O), SHM_EINVAL, "cnn dense_dw: O %d / F %d unsupported", O, F);
    constexpr int MT = 8;
    const size_t lds = sizeof(float) * ((size_t)8 * MT * 256 + 8 * MT * 16);
    int rc = opt_in_lds(dense_dw_mfma_kernel<MT>, lds);
    if (rc) return rc;
    SHM_LAUNCH((dense_dw_mfma_kernel<MT>), dim3(F / 16), dim3(512), lds, s, B, F, dz, in, dW, db);
    SHM_LAUNCH_CHECK();
    return SHM_OK;
}
int cnn_ce_loss(int B, int NC, const float* logits, const long long* labels, float* dlogits, float* loss, hipStream_t s) {
    SHM_LAUNCH(ce_loss_kernel, dim3(1), dim3(256), 0, s, B, NC, logits, labels, dlogits, loss);
    SHM_LAUNCH_CHECK();
    return SHM_OK;
}
int cnn_softmax_argmax(int B, int NC, const float* logits, long long* cls, float* p1, hipStream_t s) {
    SHM_LAUNCH(softmax_argmax_kernel, dim3(ceil_div(B, 256)), dim3(256), 0, s, B, NC, logits, cls, p1);
    SHM_LAUNCH_CHECK();
    return SHM_OK;
}
int cnn_build_input(int B, int TD, const float* z, const float* zhat, float* xin, hipStream_t s) {
    const size_t n = (size_t)B * TD;
    SHM_LAUNCH(build_input_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, B, TD, z, zhat, xin);
    SHM_LAUNCH_CHECK();
    return SHM_OK;
}
int cnn_bn_bwd_stats(int B, int C, int H, int W, const float* dp, const unsigned char* idx, const float* y, const float* mean, const float* rstd,
                     float* part, float* k1, float* k2, float* dgamma, float* dbeta, hipStream_t s, const float* xh) {
    SHM_LAUNCH(bn_bwd_stats_kernel, dim3(B), dim3(256), 0, s, C, H, W, dp, idx, y, mean, rstd, part, xh);
    SHM_LAUNCH_CHECK();
    SHM_LAUNCH(bn_bwd_finalize_kernel, dim3(C), dim3(64), 0, s, C, B, (long long)B * H * W, part, k1, k2, dgamma, dbeta);
    SHM_LAUNCH_CHECK();
    return SHM_OK;
}
int cnn_bn_bwd_apply(int B, int C, int H, int W, const BnGrad& g, float* dy, hipStream_t s) {
    SHM_LAUNCH(bn_bwd_apply_kernel, dim3(B), dim3(256), 0, s, C, H, W, g.dp, g.idx, g.y, g.mean, g.rstd, g.gamma, g.k1, g.k2, dy);
    SHM_LAUNCH_CHECK();
    return SHM_OK;
}
int cnn_conv_bwd_data2(int B, int H, int W, const float* dy, const BnGrad* fuse, const float* w, float* dx, hipStream_t s) {
    const size_t lds = sizeof(float) * ((size_t)32 * H * W + 32 * 9 * 16);
    int rc;
    // matrix-pipe form (the forward kernel on transposed / flipped weights) when dy exists in memory and the halo'd window fits
    const size_t lds_m = conv_mfma_lds(32, 16, H, W);
    if (!fuse && lds_m <= 150 * 1024 && !getenv("SHM_CNN_VALU_DGRAD")) {
        if ((rc = opt_in_lds(conv3x3_fwd_mfma_kernel<32, 16, false, true>, lds_m))) return rc;
        SHM_LAUNCH((conv3x3_fwd_mfma_kernel<32, 16, false, true>), dim3(B), dim3(512), lds_m, s, dy, w, nullptr, dx, nullptr, H, W);
        SHM_LAUNCH_CHECK();
        return SHM_OK;
    }
    if (fuse) {
        if ((rc = opt_in_lds(conv3x3_bwd_data_kernel<16, 32, true>, lds))) return rc;
        SHM_LAUNCH((conv3x3_bwd_data_kernel<16, 32, true>), dim3(B), dim3(256), lds, s, dy, *fuse, w, dx, H, W);
    } else {
        if ((rc = opt_in_lds(conv3x3_bwd_data_kernel<16, 32, false>, lds))) return rc;
        SHM_LAUNCH((conv3x3_bwd_data_kernel<16, 32, false>), dim3(B), dim3(256), lds, s, dy, BnGrad{}, w, dx, H, W);
    }
    SHM_LAUNCH_CHECK();
    return SHM_OK;
}
// LDS bytes of the matrix-pipe weight-gradient kernel (fused: + the pooled-cell table and the window's pooled gradient records)
static size_t wgrad_mfma_lds(int layer, int H, int W, bool fused) {
    const int HW = H * W, KP = (HW + 15) & ~15, XP = (H + 2) * (W + 2), npc1 = (H / 2) * (W / 2) + 1;
    const int cin = layer == 1 ? 2 : 16, cout = layer == 1 ? 16 : 32, tiles = (cout / 16) * ((cin * 9 + 15) / 16);
    return sizeof(float) * ((size_t)((cin * XP + 3) & ~3) + KP + (fused ? (size_t)KP + 2 * (size_t)cout * npc1 : 0) +
                            (tiles >= 8 ? 0 : (size_t)8 * tiles * 256 + 8 * cout));
}
bool cnn_conv_bwd_weight_fusable(int layer, int H, int W, const float* y) {
    return wgrad_mfma_lds(layer, H, W, true) <= 150 * 1024 && (H * W) % 4 == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0 &&
           !getenv("SHM_CNN_VALU_WGRAD") && !getenv("SHM_CNN_UNFUSED_BN");
}
bool cnn_conv_bwd_weight_leaves_dy(int layer, int H, int W, const float* y) {
    return layer == 2 && cnn_conv_bwd_weight_fusable(layer, H, W, y) && !getenv("SHM_CNN_TILEPAR_WGRAD");
}
int cnn_conv_bwd_weight(int layer, int B, int H, int W, const float* dy, const BnGrad* fuse, const float* x, float* part, hipStream_t s, float* dy_out) {
    int rc;
    // matrix-pipe version unless its staged window does not fit the LDS (or SHM_CNN_VALU_WGRAD is set: measurement knob)
    const int HW = H * W;
    const size_t lds_m = wgrad_mfma_lds(layer, H, W, fuse != nullptr);
    if (fuse) {
        SHM_REQUIRE(cnn_conv_bwd_weight_fusable(layer, H, W, fuse->y), SHM_EINVAL, "cnn conv_bwd_weight: fused form not available for %dx%d", H, W);
        if (layer == 1) {
            if ((rc = opt_in_lds(conv3x3_bwd_weight_mfma_kernel<2, 16, true>, lds_m))) return rc;
            SHM_LAUNCH((conv3x3_bwd_weight_mfma_kernel<2, 16, true>), dim3(B), dim3(512), lds_m, s, dy, *fuse, x, part, H, W);
        } else if (!getenv("SHM_CNN_TILEPAR_WGRAD")) {
            // K-split form: same staged operands + [4][32] row sums; the exchange buffer (2 x 18 KB) reuses the operands' space
            const size_t xchg = (size_t)2 * 2 * 9 * 1024, lds_k = (lds_m > xchg ? lds_m : xchg) + sizeof(float) * 4 * 32;
            if ((rc = opt_in_lds(conv3x3_bwd_weight_ksplit_kernel<16, 32>, lds_k))) return rc;
            SHM_LAUNCH((conv3x3_bwd_weight_ksplit_kernel<16, 32>), dim3(B), dim3(512), lds_k, s, *fuse, x, part, dy_out, H, W);
        } else {
            if ((rc = opt_in_lds(conv3x3_bwd_weight_mfma_kernel<16, 32, true>, lds_m))) return rc;
            SHM_LAUNCH((conv3x3_bwd_weight_mfma_kernel<16, 32, true>), dim3(B), dim3(512), lds_m, s, dy, *fuse, x, part, H, W);
        }
        SHM_LAUNCH_CHECK();
        return SHM_OK;
    }
    if (lds_m <= 150 * 1024 && HW % 4 == 0 && (reinterpret_cast<uintptr_t>(dy) & 15) == 0 && !getenv("SHM_CNN_VALU_WGRAD")) {
        if (layer == 1) {
            if ((rc = opt_in_lds(conv3x3_bwd_weight_mfma_kernel<2, 16, false>, lds_m))) return rc;
            SHM_LAUNCH((conv3x3_bwd_weight_mfma_kernel<2, 16, false>), dim3(B), dim3(512), lds_m, s, dy, BnGrad{}, x, part, H, W);
        } else {
            if ((rc = opt_in_lds(conv3x3_bwd_weight_mfma_kernel<16, 32, false>, lds_m))) return rc;
            SHM_LAUNCH((conv3x3_bwd_weight_mfma_kernel<16, 32, false>), dim3(B), dim3(512), lds_m, s, dy, BnGrad{}, x, part, H, W);
        }
        SHM_LAUNCH_CHECK();
        return SHM_OK;
    }
    if (layer == 1) {
        const size_t lds = sizeof(float) * ((size_t)(16 + 2) * H * W + 2 + 8 * 32 * 9);
        if ((rc = opt_in_lds(conv3x3_bwd_weight_kernel<2, 16>, lds))) return rc;
        SHM_LAUNCH((conv3x3_bwd_weight_kernel<2, 16>), dim3(B), dim3(256), lds, s, dy, x, part, H, W);
    } else {
        const size_t lds = sizeof(float) * ((size_t)(32 + 16) * H * W + 16);
        if ((rc = opt_in_lds(conv3x3_bwd_weight_kernel<16, 32>, lds))) return rc;
        SHM_LAUNCH((conv3x3_bwd_weight_kernel<16, 32>), dim3(B), dim3(256), lds, s, dy, x, part, H, W);
    }
    SHM_LAUNCH_CHECK();
    return SHM_OK;
}

}  // namespace shm
