// cnn_openlab.hip — the openLAB fault classifier, inference (eval) forward, fp32; gfx950.
// Reference: 20250506_openLAB_tests/Codes/Models/cnn_model.py:14-62 —
//   block(cin, cout, kt, 3, pad (kt/2, 1)) = Conv2d -> GroupNorm(8, eps 1e-5) -> SiLU;  features = block(C,32,7) MaxPool(2,1)
//   block(32,64,5) MaxPool(2,1) block(64,128,5) MaxPool(2,1) block(128,256,3) AdaptiveAvgPool(1,1);
//   classifier = Linear(256,128) SiLU Dropout Linear(128, num_classes)            (Dropout is the identity in eval mode)
// Used by the hybrid VAE-threshold -> CNN inference of the openLAB windows (BASELINE config 5).
//
// One workgroup per window and block.  GroupNorm needs the statistics of a whole (group, window) slab before anything can be
// normalised, so a block is three passes of the same workgroup over a per-window scratch slab that stays in L2/MALL:
//   1. convolution: blocks whose channel counts are multiples of 16 (2-4: 99 % of the work) run as an implicit GEMM on the matrix
//      pipe in exact fp32 (ol_conv_mfma below); block 1 (one input channel) is direct vector code, a thread owning 8 output
//      channels x 4 rows of one column with the weights re-laid [ci][dt][df][co];
//   2. per-group mean / rstd (one wave per group, fp64 accumulation, fixed order);
//   3. normalise + affine + SiLU + MaxPool(2,1) into the next block's input — or, in the last block, the global average
//      and the two dense layers (256 -> 128 -> classes), all in the same workgroup.
// fp32 throughout (the parity bar is 1e-4 on the logits; measured ~1e-6).
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "kernels.hpp"

namespace shm {

namespace {

constexpr int OL_GROUPS = 8, OL_CC = 8;     // GroupNorm groups; output channels per thread in the convolution

__device__ __forceinline__ float silu(float v) { return v / (1.0f + expf(-v)); }
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// w [COUT][CIN][KT][3] -> wt [CIN][KT][3][COUT]
__global__ void __launch_bounds__(256) ol_weight_relayout_kernel(const float* __restrict__ w, float* __restrict__ wt, int COUT, int CK) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < COUT * CK) { const int co = i / CK, r = i - co * CK; wt[(size_t)r * COUT + co] = w[i]; }
}

// w [COUT][CIN][TAPS] -> wt2 [TAPS][COUT][CIN] (input channel fastest): the A operand of the MFMA convolution
__global__ void __launch_bounds__(256) ol_weight_relayout_mfma_kernel(const float* __restrict__ w, float* __restrict__ wt2, int COUT, int CIN, int TAPS) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < COUT * CIN * TAPS) {
        const int tap = i / (COUT * CIN), r = i - tap * COUT * CIN, co = r / CIN, ci = r - co * CIN;
        wt2[i] = w[((size_t)co * CIN + ci) * TAPS + tap];
    }
}

constexpr int OL_THREADS = 512, OL_WAVES = OL_THREADS / 64;
constexpr int OL_NTW = 7;                   // 16-position tiles a wave accumulates per unit of work (28 accumulator registers)
constexpr int OL_CPAD = 4;                  // floats of padding per staged position: consecutive positions land on different banks
constexpr size_t OL_LDS_BUDGET = 72 * 1024; // staged input rows of one chunk: two workgroups per CU (one stages / normalises while the other multiplies)

// Convolution of one window on the matrix pipe, exact fp32 (v_mfma_f32_16x16x4_f32): Y[co][pos] = b[co] + sum_{tap, ci} W[tap][co][ci] X[ci][pos + tap].
// M = 16 output channels, N = 16 consecutive positions (row-major h, w), K = input channels of one tap, 16 per loop trip: lane quarter q
// owns channels 4q..4q+3 of the trip, so ONE 16-byte LDS read (B operand: the staged input, channel-fastest, zero rows above / below) and ONE 16-byte
// global read (A operand: wt2) feed four MFMAs.  A wave keeps the accumulators of up to OL_NTW position tiles and reuses every weight
// fragment for all of them; weight fragments are prefetched one trip ahead.  Rows are processed in chunks of HC so that the staged
// input fits the LDS budget at any sequence length.
__device__ __forceinline__ void ol_conv_mfma(const float* __restrict__ xin, const float* __restrict__ wt2, const float* __restrict__ bias,
                                             float* __restrict__ yb, float* xs, int CIN, int COUT, int KT, int H, int W, int HC) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, q = lane >> 4, j = lane & 15;
    const int CINP = CIN + OL_CPAD, PT = KT / 2, HW = H * W, CB = CIN / 16, trips = KT * 3 * CB;
    for (int hc0 = 0; hc0 < H; hc0 += HC) {
        const int hc = min(HC, H - hc0), rows = hc + KT - 1;
        __syncthreads();                                              // the previous chunk's reads are done
        // staged rows hc0 - PT .. hc0 + hc + PT - 1 (zero outside the window), all W columns, channel-fastest; no column halo: a tap
        // that leaves the row is masked at the read
        // global reads walk (row, column) of one channel plane; eight loads in flight per thread (issued one per trip, every read
        // would wait out its full latency before the next)
        for (int e0 = tid; e0 < CIN * rows * W; e0 += OL_THREADS * 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = e0 + u * OL_THREADS, ec = e < CIN * rows * W ? e : 0;
                const int ci = ec / (rows * W), rem = ec - ci * rows * W, hh = hc0 - PT + rem / W;
                v[u] = (hh >= 0 && hh < H) ? xin[(size_t)ci * HW + (size_t)(hc0 - PT) * W + rem] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = e0 + u * OL_THREADS;
                if (e < CIN * rows * W) { const int ci = e / (rows * W), rem = e - ci * rows * W; xs[rem * CINP + ci] = v[u]; }
            }
        }
        __syncthreads();
        const int npos = hc * W, ntiles = (npos + 15) / 16, ngroups = (ntiles + OL_NTW - 1) / OL_NTW, mtiles = COUT / 16;
        const int gsz = (ntiles + ngroups - 1) / ngroups;             // tiles per group, evened out (8 tiles = 4 + 4, not 7 + 1)
        for (int u = wave; u < mtiles * ngroups; u += OL_WAVES) {
            const int mt = u / ngroups, g = u - mt * ngroups, nt0 = g * gsz, ntn = min(gsz, ntiles - nt0), co0 = mt * 16;
            int pbase[OL_NTW];
            unsigned ok_l = 0, ok_r = 0;                              // bit n: this lane's position of tile n has a left / right neighbour column
            f32x4 acc[OL_NTW];
#pragma unroll
            for (int n = 0; n < OL_NTW; ++n) {
                const int p = min((nt0 + n) * 16 + j, npos - 1), w0 = p % W;
                pbase[n] = p * CINP + 4 * q;                          // tap (dt, df) adds (dt * W + df - 1) * CINP, trip cb adds 16 * cb
                ok_l |= (w0 > 0 ? 1u : 0u) << n;
                ok_r |= (w0 < W - 1 ? 1u : 0u) << n;
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[n][r] = bias[co0 + 4 * q + r];
            }
            const float* wlane = wt2 + (size_t)(co0 + j) * CIN + 4 * q;  // + tap * COUT * CIN + 16 * cb
            f32x4 a_nxt = *reinterpret_cast<const f32x4*>(wlane);
            int tap = 0, cb = 0;
            for (int it = 0; it < trips; ++it) {
                const f32x4 a = a_nxt;
                const int dt = tap / 3, df = tap - dt * 3;
                const int off = (dt * W + df - 1) * CINP + 16 * cb;
                const unsigned okm = df == 1 ? ~0u : (df == 0 ? ok_l : ok_r);
                if (++cb == CB) { cb = 0; ++tap; }
                if (it + 1 < trips) a_nxt = *reinterpret_cast<const f32x4*>(wlane + (size_t)tap * COUT * CIN + 16 * cb);
                f32x4 bv[OL_NTW];
#pragma unroll
                for (int n = 0; n < OL_NTW; ++n)
                    if (n < ntn) {
                        const bool ok = (okm >> n) & 1u;              // masked lanes read their own (valid) position and drop the value
                        const f32x4 v = *reinterpret_cast<const f32x4*>(xs + pbase[n] + (ok ? off : 0));
                        bv[n] = ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
                    }
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int n = 0; n < OL_NTW; ++n)
                        if (n < ntn) acc[n] = mfma_f32_16x16x4(a[e], bv[n][e], acc[n]);
            }
#pragma unroll
            for (int n = 0; n < OL_NTW; ++n) {
                const int p = (nt0 + n) * 16 + j;
                if (n < ntn && p < npos) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) yb[(size_t)(co0 + 4 * q + r) * HW + hc0 * W + p] = acc[n][r];
                }
            }
        }
    }
}

struct OlHead { const float* w1; const float* b1; const float* w2; const float* b2; float* logits; int NC; };

// in [B][CIN][H][W], y scratch [B][COUT][H][W], out [B][COUT][H/2][W] (POOL) or logits (LAST)
template <bool LAST>
__global__ void __launch_bounds__(OL_THREADS, 4) ol_block_kernel(const float* __restrict__ in, const float* __restrict__ wt, const float* __restrict__ bias,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ y,
                                                       float* __restrict__ out, int CIN, int COUT, int KT, int H, int W, int HC, OlHead hd) {
    extern __shared__ __attribute__((aligned(16))) float xs[];       // MFMA blocks: the staged input chunk
    __shared__ float g_mean[OL_GROUPS], g_rstd[OL_GROUPS];
    __shared__ float feat[256], hid[128], fpart[OL_THREADS / 256][256];
    const int b = blockIdx.x, tid = threadIdx.x, HW = H * W, PT = KT / 2;
    const float* xin = in + (size_t)b * CIN * HW;
    float* yb = y + (size_t)b * COUT * HW;

    // HC > 0: this block's convolution runs on the matrix pipe (wt is then the [tap][co][ci] layout)
    if (HC > 0) ol_conv_mfma(xin, wt, bias, yb, xs, CIN, COUT, KT, H, W, HC);
    else

    // ---- pass 1: convolution + bias.  A thread owns 8 consecutive output channels x 4 consecutive rows of one column: per
    // (input channel, column tap) it loads the 4 + KT - 1 input rows it needs once and reuses every weight vector for 4 rows
    // (~9 FMAs per load; the one-position-per-thread form was load-bound at 8 TFLOP/s).
    {
        constexpr int RB = 4, KTMAX = 7;
        const int chunks = COUT / OL_CC, hgroups = (H + RB - 1) / RB;
        for (int item = tid; item < chunks * W * hgroups; item += OL_THREADS) {
            const int ch = item / (W * hgroups), r0 = item - ch * W * hgroups, hg = r0 / W, w0 = r0 - hg * W;
            const int co0 = ch * OL_CC, h0 = hg * RB;
            float acc[RB][OL_CC];
#pragma unroll
            for (int hp = 0; hp < RB; ++hp)
#pragma unroll
                for (int k = 0; k < OL_CC; ++k) acc[hp][k] = bias[co0 + k];
            for (int ci = 0; ci < CIN; ++ci) {
                const float* xc = xin + (size_t)ci * HW;
#pragma unroll
                for (int df = 0; df < 3; ++df) {
                    const int ww = w0 + df - 1;
                    if (ww < 0 || ww >= W) continue;
                    float xr[RB + KTMAX - 1];
#pragma unroll
                    for (int i = 0; i < RB + KTMAX - 1; ++i) {
                        const int hh = h0 - PT + i;
                        xr[i] = (i < RB + KT - 1 && hh >= 0 && hh < H) ? xc[hh * W + ww] : 0.f;
                    }
                    const float* wr = wt + ((size_t)ci * KT * 3 + df) * COUT + co0;
#pragma unroll
                    for (int dt = 0; dt < KTMAX; ++dt) {
                        if (dt < KT) {
                            const f32x4 wa = *reinterpret_cast<const f32x4*>(wr + (size_t)dt * 3 * COUT);
                            const f32x4 wb = *reinterpret_cast<const f32x4*>(wr + (size_t)dt * 3 * COUT + 4);
#pragma unroll
                            for (int hp = 0; hp < RB; ++hp) {
                                const float xv = xr[hp + dt];
                                acc[hp][0] = fmaf(xv, wa[0], acc[hp][0]); acc[hp][1] = fmaf(xv, wa[1], acc[hp][1]);
                                acc[hp][2] = fmaf(xv, wa[2], acc[hp][2]); acc[hp][3] = fmaf(xv, wa[3], acc[hp][3]);
                                acc[hp][4] = fmaf(xv, wb[0], acc[hp][4]); acc[hp][5] = fmaf(xv, wb[1], acc[hp][5]);
                                acc[hp][6] = fmaf(xv, wb[2], acc[hp][6]); acc[hp][7] = fmaf(xv, wb[3], acc[hp][7]);
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int hp = 0; hp < RB; ++hp)
                if (h0 + hp < H) {
#pragma unroll
                    for (int k = 0; k < OL_CC; ++k) yb[(size_t)(co0 + k) * HW + (h0 + hp) * W + w0] = acc[hp][k];
                }
        }
    }
    __syncthreads();

    // ---- pass 2: GroupNorm statistics (biased variance), wave w -> group w
    {
        const int wv = tid >> 6, lane = tid & 63, cg = COUT / OL_GROUPS, n = cg * HW;
        static_assert(OL_WAVES == OL_GROUPS, "one GroupNorm group per wave");
        for (int g = wv; g < wv + 1; ++g) {
            const float* yg = yb + (size_t)g * n;
            double s = 0.0, q = 0.0;
            int i = lane;
            for (; i + 7 * 64 < n; i += 8 * 64) {            // eight loads in flight; the sums keep their lane-sequential order
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = yg[i + u * 64];
#pragma unroll
                for (int u = 0; u < 8; ++u) { const double d = (double)v[u]; s += d; q += d * d; }
            }
            for (; i < n; i += 64) { const double v = (double)yg[i]; s += v; q += v * v; }
            s = wave_sum_f64(s); q = wave_sum_f64(q);
            if (lane == 0) {
                const double m = s / n;
                double var = q / n - m * m;
                if (var < 0.0) var = 0.0;
                g_mean[g] = (float)m; g_rstd[g] = (float)(1.0 / sqrt(var + 1e-5));
            }
        }
    }
    __syncthreads();

    // ---- pass 3: normalise + affine + SiLU, then MaxPool(2,1) or the head
    const int cg = COUT / OL_GROUPS;
    if constexpr (!LAST) {
        const int Hp = H / 2;
        float* ob = out + (size_t)b * COUT * Hp * W;
        const int nout = COUT * Hp * W;
        for (int e0 = tid; e0 < nout; e0 += OL_THREADS * 4) {        // the eight reads of four outputs are issued together
            float a0[4], a1[4], sc[4], sh[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = min(e0 + u * OL_THREADS, nout - 1);
                const int co = e / (Hp * W), r = e - co * Hp * W, hp = r / W, w0 = r - hp * W, g = co / cg;
                sc[u] = gamma[co] * g_rstd[g]; sh[u] = beta[co] - g_mean[g] * sc[u];
                a0[u] = yb[(size_t)co * HW + (2 * hp) * W + w0];
                a1[u] = yb[(size_t)co * HW + (2 * hp + 1) * W + w0];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = e0 + u * OL_THREADS;
                if (e < nout) ob[e] = fmaxf(silu(fmaf(a0[u], sc[u], sh[u])), silu(fmaf(a1[u], sc[u], sh[u])));
            }
        }
    } else {
        // AdaptiveAvgPool(1,1): OL_THREADS / 256 threads per channel (COUT == 256), each over a contiguous share of the positions; fixed order
        {
            constexpr int PARTS = OL_THREADS / 256;
            const int co = tid % 256, part = tid / 256, g = co / cg;
            const int per = (HW + PARTS - 1) / PARTS, p0 = part * per, p1 = min(HW, p0 + per);
            const float sc = gamma[co] * g_rstd[g], sh = beta[co] - g_mean[g] * sc;
            float s = 0.f;
            for (int p = p0; p < p1; ++p) s += silu(fmaf(yb[(size_t)co * HW + p], sc, sh));
            fpart[part][co] = s;
            __syncthreads();
            if (tid < 256) {
                float t = 0.f;
#pragma unroll
                for (int pp = 0; pp < PARTS; ++pp) t += fpart[pp][tid];
                feat[tid] = t / (float)HW;
            }
        }
        __syncthreads();
        // Linear(256, 128) + SiLU: a wave per 16 outputs, lanes along the 256 inputs (coalesced weight rows), wave reduction
        {
            const int wv = tid >> 6, lane = tid & 63;
            for (int o = wv * (128 / OL_WAVES); o < (wv + 1) * (128 / OL_WAVES); ++o) {
                float a = 0.f;
#pragma unroll
                for (int k = 0; k < 4; ++k) a = fmaf(feat[lane + 64 * k], hd.w1[(size_t)o * 256 + lane + 64 * k], a);
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off, 64);
                if (lane == 0) hid[o] = silu(a + hd.b1[o]);
            }
        }
        __syncthreads();
        if (tid < hd.NC) {
            float a = hd.b2[tid];
            for (int j = 0; j < 128; ++j) a = fmaf(hid[j], hd.w2[(size_t)tid * 128 + j], a);
            hd.logits[(size_t)b * hd.NC + tid] = a;
        }
    }
}

struct OlBlock { int cin, cout, kt; };
void ol_blocks(int in_ch, OlBlock (&bl)[4]) { bl[0] = {in_ch, 32, 7}; bl[1] = {32, 64, 5}; bl[2] = {64, 128, 5}; bl[3] = {128, 256, 3}; }

bool ol_config_ok(const shm_olcnn_config* c) {
    if (!c) { set_error("olcnn: config is NULL"); return false; }
    if (c->in_channels < 1 || c->in_channels > 64) { set_error("olcnn: in_channels %d unsupported (1..64)", c->in_channels); return false; }
    if (c->num_classes < 1 || c->num_classes > 64) { set_error("olcnn: num_classes %d unsupported (1..64)", c->num_classes); return false; }
    if (c->seq_len < 8 || c->num_features < 1 || c->num_features > 64) {
        set_error("olcnn: seq_len %d (>= 8) / num_features %d (1..64) unsupported", c->seq_len, c->num_features); return false; }
    return true;
}

int ol_layout(const shm_olcnn_config* c, std::vector<shm_tensor_desc>& out) {
    OlBlock bl[4];
    ol_blocks(c->in_channels, bl);
    int64_t off = 0;
    auto add = [&](const char* name, int ndim, int s0, int s1, int s2, int s3) {
        shm_tensor_desc d;
        memset(&d, 0, sizeof(d));
        snprintf(d.name, sizeof(d.name), "%s", name);
        d.offset = off; d.ndim = ndim; d.shape[0] = s0; d.shape[1] = s1; d.shape[2] = s2; d.shape[3] = s3;
        int64_t n = s0;
        if (ndim > 1) n *= s1;
        if (ndim > 2) n *= s2;
        if (ndim > 3) n *= s3;
        off = (off + n + 3) & ~(int64_t)3;
        out.push_back(d);
    };
    char nm[48];
    for (int i = 0; i < 4; ++i) {
        snprintf(nm, sizeof(nm), "features.%d.0.weight", 2 * i); add(nm, 4, bl[i].cout, bl[i].cin, bl[i].kt, 3);
        snprintf(nm, sizeof(nm), "features.%d.0.bias", 2 * i); add(nm, 1, bl[i].cout, 0, 0, 0);
        snprintf(nm, sizeof(nm), "features.%d.1.weight", 2 * i); add(nm, 1, bl[i].cout, 0, 0, 0);
        snprintf(nm, sizeof(nm), "features.%d.1.bias", 2 * i); add(nm, 1, bl[i].cout, 0, 0, 0);
    }
    add("classifier.1.weight", 2, 128, 256, 0, 0); add("classifier.1.bias", 1, 128, 0, 0, 0);
    add("classifier.4.weight", 2, c->num_classes, 128, 0, 0); add("classifier.4.bias", 1, c->num_classes, 0, 0, 0);
    return (int)off;
}

// workspace: re-laid conv weights | y scratch (largest block output) | two ping-pong pooled activations
struct OlWs { size_t wt[4], y, a0, a1, total; };
OlWs ol_ws(const shm_olcnn_config* c, int B) {
    OlBlock bl[4];
    ol_blocks(c->in_channels, bl);
    OlWs w;
    size_t off = 0;
    auto take = [&](size_t floats) { size_t o = off; off += (floats * sizeof(float) + 255) & ~(size_t)255; return o; };
    for (int i = 0; i < 4; ++i) w.wt[i] = take((size_t)bl[i].cout * bl[i].cin * bl[i].kt * 3);
    const size_t HW = (size_t)c->seq_len * c->num_features;
    w.y = take((size_t)B * 32 * HW);              // every block's pre-pool output is <= 32 * T * W floats per window (channels double, rows halve)
    w.a0 = take((size_t)B * 32 * (HW / 2 + c->num_features));
    w.a1 = take((size_t)B * 32 * (HW / 2 + c->num_features));
    w.total = off;
    return w;
}

}  // namespace

}  // namespace shm

using namespace shm;

extern "C" int64_t shm_olcnn_param_count(const shm_olcnn_config* cfg) {
    if (!ol_config_ok(cfg)) return SHM_EINVAL;
    std::vector<shm_tensor_desc> v;
    return ol_layout(cfg, v);
}

extern "C" int shm_olcnn_param_layout(const shm_olcnn_config* cfg, shm_tensor_desc* out, int max) {
    if (!ol_config_ok(cfg)) return SHM_EINVAL;
    std::vector<shm_tensor_desc> v;
    ol_layout(cfg, v);
    for (int i = 0; i < (int)v.size() && i < max && out; ++i) out[i] = v[i];
    return (int)v.size();
}

extern "C" size_t shm_olcnn_workspace_bytes(const shm_olcnn_config* cfg, int32_t B) {
    if (!ol_config_ok(cfg) || B < 1) return 0;
    return ol_ws(cfg, B).total;
}

extern "C" int shm_olcnn_forward(const shm_olcnn_config* cfg, const float* params, const float* x, float* logits, int32_t B, void* ws, void* stream) {
    if (!ol_config_ok(cfg)) return SHM_EINVAL;
    SHM_REQUIRE(params && x && logits && ws && B >= 1, SHM_EINVAL, "olcnn_forward: NULL argument or empty batch");
    SHM_REQUIRE((reinterpret_cast<uintptr_t>(ws) & 255) == 0 && (reinterpret_cast<uintptr_t>(params) & 15) == 0, SHM_EWORKSPACE,
                "olcnn_forward: workspace must be 256-byte and params 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    std::vector<shm_tensor_desc> lay;
    ol_layout(cfg, lay);
    OlBlock bl[4];
    ol_blocks(cfg->in_channels, bl);
    const OlWs w = ol_ws(cfg, B);
    char* wsb = reinterpret_cast<char*>(ws);
    auto P = [&](int i) { return params + lay[i].offset; };
    const int W = cfg->num_features;
    int hc[4];                                          // rows per staged chunk of an MFMA block; 0: vector-ALU block (input channels not a multiple of 16)
    size_t lds[4];
    for (int i = 0; i < 4; ++i) {
        const int ck = bl[i].cin * bl[i].kt * 3, n = bl[i].cout * ck;
        const size_t row_bytes = (size_t)W * (bl[i].cin + OL_CPAD) * sizeof(float);
        const long fit = (long)(OL_LDS_BUDGET / row_bytes) - (bl[i].kt - 1);
        const bool mfma = !getenv("SHM_OLCNN_VALU") && bl[i].cin % 16 == 0 && bl[i].cout % 16 == 0 && fit >= 1;
        hc[i] = mfma ? (int)(fit < cfg->seq_len ? fit : cfg->seq_len) : 0;
        lds[i] = mfma ? (size_t)(hc[i] + bl[i].kt - 1) * row_bytes : 0;
        if (mfma) SHM_LAUNCH(ol_weight_relayout_mfma_kernel, dim3((n + 255) / 256), dim3(256), 0, s, P(4 * i), reinterpret_cast<float*>(wsb + w.wt[i]), bl[i].cout, bl[i].cin, bl[i].kt * 3);
        else      SHM_LAUNCH(ol_weight_relayout_kernel, dim3((n + 255) / 256), dim3(256), 0, s, P(4 * i), reinterpret_cast<float*>(wsb + w.wt[i]), bl[i].cout, ck);
        SHM_LAUNCH_CHECK();
    }
    static bool attr_set = false;
    if (!attr_set) {
        SHM_HIP_CHECK(hipFuncSetAttribute((const void*)ol_block_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)OL_LDS_BUDGET));
        SHM_HIP_CHECK(hipFuncSetAttribute((const void*)ol_block_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)OL_LDS_BUDGET));
        attr_set = true;
    }
    OlHead hd;
    hd.w1 = P(16); hd.b1 = P(17); hd.w2 = P(18); hd.b2 = P(19); hd.logits = logits; hd.NC = cfg->num_classes;
    const float* in = x;
    float* y = reinterpret_cast<float*>(wsb + w.y);
    float* pp[2] = {reinterpret_cast<float*>(wsb + w.a0), reinterpret_cast<float*>(wsb + w.a1)};
    int H = cfg->seq_len;
    for (int i = 0; i < 4; ++i) {
        if (hc[i] > 0) {                                 // rows per chunk, evened out over the chunks this block's H needs
            const int nchunks = (H + hc[i] - 1) / hc[i];
            hc[i] = (H + nchunks - 1) / nchunks;
            lds[i] = (size_t)(hc[i] + bl[i].kt - 1) * W * (bl[i].cin + OL_CPAD) * sizeof(float);
        }
        const float* wt = reinterpret_cast<const float*>(wsb + w.wt[i]);
        if (i < 3) {
            SHM_LAUNCH(ol_block_kernel<false>, dim3(B), dim3(OL_THREADS), lds[i], s, in, wt, P(4 * i + 1), P(4 * i + 2), P(4 * i + 3), y, pp[i & 1],
                               bl[i].cin, bl[i].cout, bl[i].kt, H, W, hc[i], hd);
            in = pp[i & 1];
            H /= 2;
        } else {
            SHM_REQUIRE(H >= 1, SHM_EINVAL, "olcnn_forward: seq_len too short");
            SHM_LAUNCH(ol_block_kernel<true>, dim3(B), dim3(OL_THREADS), lds[i], s, in, wt, P(4 * i + 1), P(4 * i + 2), P(4 * i + 3), y, (float*)nullptr,
                               bl[i].cin, bl[i].cout, bl[i].kt, H, W, hc[i], hd);
        }
        SHM_LAUNCH_CHECK();
    }
    return SHM_OK;
}
