// small_kernels.hip — the latency-trivial pieces around the LSTM recurrences (gfx950, wave64):
//   head_fwd / head_bwd : LayerNorm -> fc_mu, fc_logvar -> reparameterise -> tanh(fc_latent_to_hidden)
//                         (reference 4DOF/Scripts/Models/temporal_vae.py:55-57, 60-63, 67) and its backward
//   out_fwd             : output_layer (temporal_vae.py:70) + per-window score (04_vae_thresholding.py:122)
//   vae_loss            : recon/KL/total loss and their gradients (03_train_vae.py:264-266)
//   colsum, pack_*      : bias gradients; MFMA fragment-order weight packing (layouts in kernels.hpp)
// One wave per window row; reductions are wave shuffles in a fixed order (bitwise reproducible).
#include "kernels.hpp"

namespace shm {

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

constexpr int HMAX = 128;   // hidden_dim limit of the head kernels
constexpr int ZMAX = 32;    // latent_dim limit

// ------------------------------------------------------------------------------------------------ head forward
// Noise: eps (injected, parity tests and the drop-in module's torch draw) or, with eps == nullptr and rng.on, the counter RNG of
// common.hpp keyed by (seed, offset); whichever is used is stored in eps_store for the backward.  Neither: z = mu.
// z_in != nullptr: decode-only (temporal_vae.py:65-67) — the encoder half is skipped and h0 = tanh(fc_latent_to_hidden(z_in)).
__global__ void __launch_bounds__(256) head_fwd_kernel(int B, int H, int Z, int layer_norm, const float* __restrict__ h_last,
                                                       const float* __restrict__ eps, HeadParams p, HeadBufs hb,
                                                       float* mu_out, float* lv_out, HeadNoise rng, const float* __restrict__ z_in,
                                                       float* __restrict__ eps_store) {
    __shared__ float ys[4][HMAX];
    __shared__ float zs[4][ZMAX];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row_raw = blockIdx.x * 4 + w;
    const bool valid = row_raw < B;
    const int row = valid ? row_raw : B - 1;
    if (z_in) {
        if (lane < Z) {
            const float z = z_in[(size_t)row * Z + lane];
            zs[w][lane] = z;
            if (valid) hb.z[(size_t)row * Z + lane] = z;
        }
        __syncthreads();
        for (int k = lane; k < H; k += 64) {
            float acc = p.l2h_b[k];
            const float* wr = p.l2h_w + (size_t)k * Z;
            for (int z = 0; z < Z; ++z) acc = fmaf(zs[w][z], wr[z], acc);
            if (valid) hb.h0[(size_t)row * H + k] = tanhf(acc);
        }
        return;
    }
    const float* h = h_last + (size_t)row * H;

    float hv[HMAX / 64], s = 0.f;
#pragma unroll
    for (int i = 0; i < HMAX / 64; ++i) { const int k = lane + 64 * i; hv[i] = k < H ? h[k] : 0.f; s += hv[i]; }
    if (layer_norm) {
        const float mean = wave_sum(s) / (float)H;
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < HMAX / 64; ++i) { const int k = lane + 64 * i; const float d = k < H ? hv[i] - mean : 0.f; v += d * d; }
        const float var = wave_sum(v) / (float)H;           // biased variance (nn.LayerNorm)
        const float rstd = 1.0f / sqrtf(var + 1e-5f);
#pragma unroll
        for (int i = 0; i < HMAX / 64; ++i) {
            const int k = lane + 64 * i;
            if (k < H) {
                const float xh = (hv[i] - mean) * rstd;
                const float y = xh * p.ln_w[k] + p.ln_b[k];
                ys[w][k] = y;
                if (valid) { hb.xhat[(size_t)row * H + k] = xh; hb.y[(size_t)row * H + k] = y; }
            }
        }
        if (valid && lane == 0) hb.rstd[row] = rstd;
    } else {
#pragma unroll
        for (int i = 0; i < HMAX / 64; ++i) {
            const int k = lane + 64 * i;
            if (k < H) { ys[w][k] = hv[i]; if (valid) hb.y[(size_t)row * H + k] = hv[i]; }
        }
    }
    __syncthreads();
    if (lane < Z) {
        float m = p.mu_b[lane], l = p.lv_b[lane];
        const float* wm = p.mu_w + (size_t)lane * H;
        const float* wl = p.lv_w + (size_t)lane * H;
        for (int k = 0; k < H; ++k) { const float y = ys[w][k]; m = fmaf(y, wm[k], m); l = fmaf(y, wl[k], l); }
        const float sd = expf(0.5f * l);
        const float e = eps ? eps[(size_t)row * Z + lane] : rng.on ? eps_normal(rng.seed, rng.offset, (uint32_t)(row * Z + lane)) : 0.f;
        const float z = m + e * sd;
        zs[w][lane] = z;
        if (valid) {
            if (eps_store) eps_store[(size_t)row * Z + lane] = e;
            hb.mu[(size_t)row * Z + lane] = m; hb.lv[(size_t)row * Z + lane] = l; hb.z[(size_t)row * Z + lane] = z;
            if (mu_out) mu_out[(size_t)row * Z + lane] = m;
            if (lv_out) lv_out[(size_t)row * Z + lane] = l;
        }
    }
    __syncthreads();
    // fc_latent_to_hidden + tanh: a lane's weight row is Z contiguous floats — 16-byte loads, both of the lane's rows in flight together
    const bool z4 = Z % 4 == 0 && Z <= ZMAX && (reinterpret_cast<uintptr_t>(p.l2h_w) & 15) == 0;
    if (z4) {
        f32x4 wv[HMAX / 64][ZMAX / 4];
        float bb[HMAX / 64];
#pragma unroll
        for (int i = 0; i < HMAX / 64; ++i) {
            const int k = lane + 64 * i, kc = k < H ? k : 0;
            bb[i] = p.l2h_b[kc];
#pragma unroll
            for (int z = 0; z < ZMAX / 4; ++z)
                if (4 * z < Z) wv[i][z] = *reinterpret_cast<const f32x4*>(p.l2h_w + (size_t)kc * Z + 4 * z);
        }
#pragma unroll
        for (int i = 0; i < HMAX / 64; ++i) {
            const int k = lane + 64 * i;
            float acc = bb[i];
#pragma unroll
            for (int z = 0; z < ZMAX / 4; ++z)
                if (4 * z < Z) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc = fmaf(zs[w][4 * z + e], wv[i][z][e], acc);
                }
            if (valid && k < H) hb.h0[(size_t)row * H + k] = tanhf(acc);
        }
        return;
    }
    for (int k = lane; k < H; k += 64) {
        float acc = p.l2h_b[k];
        const float* wr = p.l2h_w + (size_t)k * Z;
        for (int z = 0; z < Z; ++z) acc = fmaf(zs[w][z], wr[z], acc);
        if (valid) hb.h0[(size_t)row * H + k] = tanhf(acc);
    }
}

// ------------------------------------------------------------------------------------------------ head backward
// Block = 4 windows, HB_WAVES = 16 waves.  dh0 = dpre_tsum . W_ih(dec0) is the only real work (4H x H weights, L2-resident): the
// 4H contraction is split over the 16 waves (32 weight rows each at H = 128; with 4 waves the 128-iteration load/FMA chain made the
// kernel 37 us), every weight element is loaded once per block and feeds all 4 windows; waves 0..3 then take one window each.
constexpr int HB_WAVES = 16;
__global__ void __launch_bounds__(64 * HB_WAVES) head_bwd_kernel(int B, int H, int Z, int layer_norm, const float* __restrict__ dpre_tsum,
                                                       const float* __restrict__ W_ih_dec0, const float* __restrict__ eps,
                                                       const float* __restrict__ d_mu, const float* __restrict__ d_lv,
                                                       HeadParams p, HeadBufs hb) {
    __shared__ float ds[4][4 * HMAX];        // dpre_tsum rows
    __shared__ float part[HB_WAVES][4][HMAX];   // [m-slice][window][k]
    __shared__ float dl[4][HMAX];            // d(pre-tanh) of fc_latent_to_hidden
    __shared__ float dmz[4][2 * ZMAX];       // dmu | dlv
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int w = wv & 3;                     // window of the per-window phases (waves 0..3 only)
    const bool rowwave = wv < 4;
    const int row_raw = blockIdx.x * 4 + w;
    const bool valid = rowwave && row_raw < B;
    const int row = row_raw < B ? row_raw : B - 1;
    const int G = 4 * H;

    if (rowwave)
        for (int m = lane; m < G; m += 64) ds[w][m] = dpre_tsum[(size_t)row * G + m];
    __syncthreads();
    if (H == 128 && (reinterpret_cast<uintptr_t>(W_ih_dec0) & 15) == 0) {
        // H = 128: a weight row is 32 lanes x 16 bytes, so a wave reads TWO rows per load instruction (lane half hp takes the odd / even
        // row of a pair) and its 32 rows are 16 loads per lane, all in flight at once (the 4-byte form needed 64 loads in four
        // dependent batches); the two halves are combined by one shuffle, wave slices as before.
        const int hp = lane >> 5, k4 = (lane & 31) * 4, mq = G / HB_WAVES;       // mq = 32 rows per wave
        f32x4 wrow[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) wrow[i] = *reinterpret_cast<const f32x4*>(W_ih_dec0 + (size_t)(wv * mq + 2 * i + hp) * H + k4);
        f32x4 acc4[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) acc4[r] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int m = wv * mq + 2 * i + hp;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float d = ds[r][m];
#pragma unroll
                for (int e = 0; e < 4; ++e) acc4[r][e] = fmaf(d, wrow[i][e], acc4[r][e]);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float o = __shfl_xor(acc4[r][e], 32, 64);               // even-row half + odd-row half
                if (hp == 0) part[wv][r][k4 + e] = acc4[r][e] + o;
            }
    } else {
        float acc[4][HMAX / 64];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < HMAX / 64; ++i) acc[r][i] = 0.f;
        const int mq = G / HB_WAVES;
#pragma unroll 8
        for (int m = wv * mq; m < (wv + 1) * mq; ++m) {
            float wv[HMAX / 64];
#pragma unroll
            for (int i = 0; i < HMAX / 64; ++i) { const int k = lane + 64 * i; wv[i] = k < H ? W_ih_dec0[(size_t)m * H + k] : 0.f; }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float d = ds[r][m];
#pragma unroll
                for (int i = 0; i < HMAX / 64; ++i) acc[r][i] = fmaf(d, wv[i], acc[r][i]);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < HMAX / 64; ++i) { const int k = lane + 64 * i; if (k < H) part[wv][r][k] = acc[r][i]; }
    }
    __syncthreads();
    if (!rowwave) return;                     // the per-window phases below synchronise waves 0..3 only through wave-local data
    for (int k = lane; k < H; k += 64) {
        float acc = 0.f;
#pragma unroll
        for (int sl = 0; sl < HB_WAVES; sl += 4)
            acc += (part[sl][w][k] + part[sl + 1][w][k]) + (part[sl + 2][w][k] + part[sl + 3][w][k]);
        const float h0 = hb.h0[(size_t)row * H + k];
        const float dp = acc * (1.0f - h0 * h0);
        dl[w][k] = dp;
        if (valid) { hb.dh0[(size_t)row * H + k] = acc; hb.dpre_l2h[(size_t)row * H + k] = dp; }
    }
    __builtin_amdgcn_wave_barrier();          // dl[w] / dmz[w] are private to wave w: LDS operations of one wave execute in order
    // dz = dl . W_l2h: lane (zq, kq) takes quarter kq of the H rows for latent zq when 4 Z <= 64 (H / 4 scalar loads per lane, issued
    // as one unrolled block; the one-lane-per-latent form was H dependent-latency loads), quarters combined by shuffles in a fixed order
    float dz_q = 0.f;
    const bool quart = 4 * Z <= 64 && H % 4 == 0 && H <= HMAX;
    if (quart) {
        const int zq = lane % Z, kq = lane / Z, hq = H / 4;
        if (kq < 4) {
            float wq[HMAX / 4];
#pragma unroll
            for (int i = 0; i < HMAX / 4; ++i)
                if (i < hq) wq[i] = p.l2h_w[(size_t)(kq * hq + i) * Z + zq];
#pragma unroll
            for (int i = 0; i < HMAX / 4; ++i)
                if (i < hq) dz_q = fmaf(dl[w][kq * hq + i], wq[i], dz_q);
        }
        const float d1 = __shfl(dz_q, lane + Z, 64), d2 = __shfl(dz_q, lane + 2 * Z, 64), d3 = __shfl(dz_q, lane + 3 * Z, 64);
        dz_q = (dz_q + d1) + (d2 + d3);
    }
    if (lane < Z) {
        float dz = 0.f;
        if (quart) dz = dz_q;
        else for (int k = 0; k < H; ++k) dz = fmaf(dl[w][k], p.l2h_w[(size_t)k * Z + lane], dz);
        const size_t i = (size_t)row * Z + lane;
        const float sd = expf(0.5f * hb.lv[i]);
        const float dm = dz + d_mu[i];
        const float dv = dz * eps[i] * 0.5f * sd + d_lv[i];
        dmz[w][lane] = dm; dmz[w][ZMAX + lane] = dv;
        if (valid) { hb.dmu[i] = dm; hb.dlv[i] = dv; }
    }
    __builtin_amdgcn_wave_barrier();          // dl[w] / dmz[w] are private to wave w: LDS operations of one wave execute in order
    float dxh[HMAX / 64], xh[HMAX / 64], s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < HMAX / 64; ++i) {
        const int k = lane + 64 * i;
        dxh[i] = 0.f; xh[i] = 0.f;
        if (k < H) {
            float dy = 0.f;
            for (int z0 = 0; z0 < Z; z0 += 8) {                      // 16 weight loads in flight per trip, same order of additions
                float wm8[8], wl8[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int z = z0 + u < Z ? z0 + u : 0;
                    wm8[u] = p.mu_w[(size_t)z * H + k]; wl8[u] = p.lv_w[(size_t)z * H + k];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (z0 + u < Z) dy += dmz[w][z0 + u] * wm8[u] + dmz[w][ZMAX + z0 + u] * wl8[u];
            }
            if (layer_norm) {
                xh[i] = hb.xhat[(size_t)row * H + k];
                dxh[i] = dy * p.ln_w[k];
                if (valid) { hb.dy[(size_t)row * H + k] = dy; hb.dyx[(size_t)row * H + k] = dy * xh[i]; }
                s1 += dxh[i]; s2 += dxh[i] * xh[i];
            } else if (valid) {
                hb.dh_last[(size_t)row * H + k] = dy;
            }
        }
    }
    if (layer_norm) {
        const float m1 = wave_sum(s1) / (float)H, m2 = wave_sum(s2) / (float)H;
        const float rstd = hb.rstd[row];
#pragma unroll
        for (int i = 0; i < HMAX / 64; ++i) {
            const int k = lane + 64 * i;
            if (k < H && valid) hb.dh_last[(size_t)row * H + k] = rstd * (dxh[i] - m1 - xh[i] * m2);
        }
    }
}

// All eight head parameter gradients in ONE launch: every output element is a fixed-order sum over the B windows of
// a product P[b][i] * Q[b][j] (weights) or of P[b][i] alone (biases, LayerNorm).  Block = 64 outputs x 4 window groups.
struct HeadGradPtrs { float *l2h_w, *l2h_b, *mu_w, *mu_b, *lv_w, *lv_b, *ln_w, *ln_b; };
constexpr int HG_GROUPS = 16;       // window groups per block: B / 16 products per thread (with 4 groups the 128-trip load chain was 11.7 us)
__global__ void __launch_bounds__(64 * HG_GROUPS) head_grads_kernel(int B, int H, int Z, int layer_norm, HeadBufs hb, HeadGradPtrs g) {
    __shared__ float red[HG_GROUPS][64];
    const int e = blockIdx.x * 64 + (threadIdx.x & 63), grp = threadIdx.x >> 6;
    const int HZ = H * Z;
    const int n_total = 3 * HZ + H + 2 * Z + (layer_norm ? 2 * H : 0);
    const float *P = nullptr, *Q = nullptr;
    float* out = nullptr;
    int i = 0, j = 0, ldp = 0, ldq = 0;
    if (e < HZ) { i = e / Z; j = e % Z; P = hb.dpre_l2h; ldp = H; Q = hb.z; ldq = Z; out = g.l2h_w + e; }
    else if (e < 2 * HZ) { const int f = e - HZ; i = f / H; j = f % H; P = hb.dmu; ldp = Z; Q = hb.y; ldq = H; out = g.mu_w + f; }
    else if (e < 3 * HZ) { const int f = e - 2 * HZ; i = f / H; j = f % H; P = hb.dlv; ldp = Z; Q = hb.y; ldq = H; out = g.lv_w + f; }
    else if (e < n_total) {
        int f = e - 3 * HZ;
        if (f < H) { i = f; P = hb.dpre_l2h; ldp = H; out = g.l2h_b + f; }
        else if ((f -= H) < Z) { i = f; P = hb.dmu; ldp = Z; out = g.mu_b + f; }
        else if ((f -= Z) < Z) { i = f; P = hb.dlv; ldp = Z; out = g.lv_b + f; }
        else if ((f -= Z) < H) { i = f; P = hb.dyx; ldp = H; out = g.ln_w + f; }
        else { f -= H; i = f; P = hb.dy; ldp = H; out = g.ln_b + f; }
    }
    float s = 0.f;
    if (P) {
        if (Q) {
#pragma unroll 8
            for (int b = grp; b < B; b += HG_GROUPS) s = fmaf(P[(size_t)b * ldp + i], Q[(size_t)b * ldq + j], s);
        } else {
#pragma unroll 8
            for (int b = grp; b < B; b += HG_GROUPS) s += P[(size_t)b * ldp + i];
        }
    }
    red[grp][threadIdx.x & 63] = s;
    __syncthreads();
    if (grp == 0 && out) {
        float t = 0.f;
#pragma unroll
        for (int q4 = 0; q4 < HG_GROUPS; q4 += 4)
            t += (red[q4][threadIdx.x] + red[q4 + 1][threadIdx.x]) + (red[q4 + 2][threadIdx.x] + red[q4 + 3][threadIdx.x]);
        *out = t;
    }
}

// ------------------------------------------------------------------------------------------------ output layer + score
constexpr int DMAX = 16;
template <typename TH, bool TILED>
__global__ void __launch_bounds__(256) out_fwd_kernel(int T, int D, int H, const TH* __restrict__ h_seq,
                                                      const float* __restrict__ Wout, const float* __restrict__ bout,
                                                      const float* __restrict__ x, float* __restrict__ recon,
                                                      float* __restrict__ score) {
    __shared__ float red[4];
    const int b = blockIdx.x, w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float wreg[DMAX][HMAX / 64];
#pragma unroll
    for (int d = 0; d < DMAX; ++d)
#pragma unroll
        for (int i = 0; i < HMAX / 64; ++i) {
            const int k = lane + 64 * i;
            wreg[d][i] = (d < D && k < H) ? Wout[(size_t)d * H + k] : 0.f;
        }
    const float bias = lane < D ? bout[lane] : 0.f;
    float sq = 0.f;
    for (int t = w; t < T; t += 4) {
        const TH* h = TILED ? h_seq + (((size_t)(b / SHM_WPT) * T + t) * SHM_WPT + (b % SHM_WPT)) * H : h_seq + ((size_t)b * T + t) * H;
        float hv[HMAX / 64];
#pragma unroll
        for (int i = 0; i < HMAX / 64; ++i) { const int k = lane + 64 * i; hv[i] = k < H ? (float)h[k] : 0.f; }
        float mine = 0.f;
#pragma unroll
        for (int d = 0; d < DMAX; ++d) {
            if (d < D) {
                float a = 0.f;
#pragma unroll
                for (int i = 0; i < HMAX / 64; ++i) a = fmaf(hv[i], wreg[d][i], a);
                a = wave_sum(a);
                if (lane == d) mine = a;
            }
        }
        if (lane < D) {
            const size_t o = ((size_t)b * T + t) * D + lane;
            const float r = mine + bias;
            if (recon) recon[o] = r;
            const float e = x ? x[o] - r : 0.f;           // x == nullptr: decode-only, no score
            sq = fmaf(e, e, sq);
        }
    }
    sq = wave_sum(sq);
    if (lane == 0) red[w] = sq;
    __syncthreads();
    if (threadIdx.x == 0 && score) score[b] = ((red[0] + red[1]) + (red[2] + red[3])) / (float)(T * D);
}

// bf16 path: recon^T = W_out . h^T on MFMA.  One workgroup = one SHM_WPT-window tile: its T*WPT rows of h_seq are one
// contiguous block; a wave takes 16 rows (= 4 steps x 4 windows) per MFMA chain, A = h rows straight from HBM (16-byte
// fragments), B = W_out^T fragments in registers.  Accumulator row 4q+r <-> (step t0+q, window r), column j = channel, so
// each lane adds the squared error of ONE window (r) and the per-window score needs no atomics.
// 8 waves per tile, 4 groups (64 rows) requested per wave and pass: a tile at T = 256 is two memory round trips
// (4 waves walking 16 groups each with dependent loads took 57 us; the kernel is pure latency).  d_recon (optional) is the
// gradient of the mean-squared error, 2 (recon - x) / (B T D), written here so that the loss kernel has no pass over [B,T,D].
constexpr int OUT_WAVES = 8;
// SPLIT (split-precision path): h_seq / h_lo are the hi / lo planes and W_out is split the same way: three products per fragment.
template <int H, bool SPLIT>
__global__ void __launch_bounds__(64 * OUT_WAVES) out_fwd_mfma_kernel(int B, int T, int D, const bf16_t* __restrict__ h_seq,
                                                           const bf16_t* __restrict__ h_lo,
                                                           const float* __restrict__ Wout, const float* __restrict__ bout,
                                                           const float* __restrict__ x, float* __restrict__ recon,
                                                           float* __restrict__ score, float* __restrict__ d_recon,
                                                           float* score_part, unsigned* tile_cnt) {
    static_assert(SHM_WPT == 4, "row <-> (step, window) mapping below assumes 4-window tiles");
    constexpr int KS = H / 32;
    __shared__ float red[OUT_WAVES][4][64];
    const float dsc = 2.0f / ((float)B * (float)T * (float)D);
    const int tile = blockIdx.x, w = threadIdx.x >> 6, lane = threadIdx.x & 63, q = lane >> 4, j = lane & 15;
    bf16x8 wf[KS], wfl[SPLIT ? KS : 1];  // B operand: lane (q,j) holds W_out[j][8q + 32s .. +7] (zero for j >= D)
#pragma unroll
    for (int s = 0; s < KS; ++s) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float w = j < D ? Wout[(size_t)j * H + 8 * q + 32 * s + e] : 0.f;
            wf[s][e] = (bf16_t)w;
            if constexpr (SPLIT) wfl[s][e] = (bf16_t)(w - (float)wf[s][e]);
        }
    }
    const float bias = j < D ? bout[j] : 0.f;
    const int n_groups = (T * 4 + 15) / 16;                           // 16-row groups of this tile (4 steps each)
    // gridDim.y workgroups share a tile (B = 512 is only 128 tiles on 256 CUs): part y takes a contiguous range of the groups
    const int part = blockIdx.y, nsplit = gridDim.y;
    const int per_part = (n_groups + nsplit - 1) / nsplit;
    const int g_lo = part * per_part, g_hi = min(n_groups, g_lo + per_part);
    const bf16_t* hb = h_seq + (size_t)tile * T * 4 * H;
    const bf16_t* hbl = SPLIT ? h_lo + (size_t)tile * T * 4 * H : nullptr;
    float sq[4] = {0.f, 0.f, 0.f, 0.f};
    // GB groups per wave and pass: all their fragment loads (GB x KS x 16 B per lane) are in flight before the first MFMA
    constexpr int GB = SPLIT ? 2 : 4;
    const bf16x8 zero8 = {(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
    for (int g0 = g_lo + w * GB; g0 < g_hi; g0 += OUT_WAVES * GB) {
        bf16x8 af[GB][KS], al[SPLIT ? GB : 1][KS];
        float xv[GB][4];                                              // the targets of this lane's outputs, fetched beside the A fragments
#pragma unroll
        for (int gi = 0; gi < GB; ++gi) {
            const int tg = (g0 + gi) * 4 + q;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int b = tile * 4 + r;
                const bool ok = x && j < D && tg < T && g0 + gi < g_hi && b < B;
                const float v = x ? x[ok ? ((size_t)b * T + tg) * D + j : 0] : 0.f;
                xv[gi][r] = ok ? v : 0.f;
            }
        }
#pragma unroll
        for (int gi = 0; gi < GB; ++gi) {
            const int row = (g0 + gi) * 16 + j;                       // A operand row supplied by this lane
            const bool rv = row < T * 4 && g0 + gi < g_hi;
            const bf16_t* src = hb + (size_t)(rv ? row : 0) * H + 8 * q;
#pragma unroll
            for (int s = 0; s < KS; ++s) { const bf16x8 v = *reinterpret_cast<const bf16x8*>(src + 32 * s); af[gi][s] = rv ? v : zero8; }
            if constexpr (SPLIT) {
                const bf16_t* srl = hbl + (size_t)(rv ? row : 0) * H + 8 * q;
#pragma unroll
                for (int s = 0; s < KS; ++s) { const bf16x8 v = *reinterpret_cast<const bf16x8*>(srl + 32 * s); al[gi][s] = rv ? v : zero8; }
            }
        }
#pragma unroll
        for (int gi = 0; gi < GB; ++gi) {
            const int grp = g0 + gi;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                acc = mfma_bf16_16x16x32(af[gi][s], wf[s], acc);
                if constexpr (SPLIT) {
                    acc = mfma_bf16_16x16x32(al[gi][s], wf[s], acc);
                    acc = mfma_bf16_16x16x32(af[gi][s], wfl[s], acc);
                }
            }
            const int t = grp * 4 + q;                                // accumulator rows 4q + r: step t, window r
            if (j < D && t < T && grp < g_hi) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int b = tile * 4 + r;
                    if (b < B) {
                        const size_t o = ((size_t)b * T + t) * D + j;
                        const float rc = acc[r] + bias;
                        if (recon) recon[o] = rc;
                        const float e = x ? xv[gi][r] - rc : 0.f;  // x == nullptr: decode-only, no score
                        if (d_recon) d_recon[o] = -e * dsc;
                        sq[r] = fmaf(e, e, sq[r]);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) red[w][r][lane] = sq[r];
    __syncthreads();
    float tot = 0.f;
    if (threadIdx.x < 4) {                                            // fixed-order sum: deterministic
        const int r = threadIdx.x;
        for (int ww = 0; ww < OUT_WAVES; ++ww)
            for (int l = 0; l < 64; ++l) tot += red[ww][r][l];
    }
    if (nsplit == 1) {
        if (threadIdx.x < 4 && score && tile * 4 + (int)threadIdx.x < B) score[tile * 4 + threadIdx.x] = tot / (float)(T * D);
        return;
    }
    // several parts per tile: each leaves its partial sums, the LAST one to arrive adds them in part order (so the result does not
    // depend on which one that is) and re-arms the counter for the next launch.  Partials cross workgroups, possibly XCDs: they are
    // stored and loaded as agent-scope atomics (write-through / cache-bypassing), retired with vmcnt(0) before the ticket is taken —
    // no fence: a release fence here would write back the whole L2 (the recon / d_recon lines this kernel just produced).
    __shared__ unsigned ticket;
    const size_t tiles = gridDim.x;
    if (threadIdx.x < 4) __hip_atomic_store(score_part + ((size_t)part * tiles + tile) * 4 + threadIdx.x, tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) ticket = __hip_atomic_fetch_add(tile_cnt + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (ticket != (unsigned)(nsplit - 1)) return;
    if (threadIdx.x < 4) {
        float sum = 0.f;
        for (int pp = 0; pp < nsplit; ++pp)
            sum += __hip_atomic_load(score_part + ((size_t)pp * tiles + tile) * 4 + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (score && tile * 4 + (int)threadIdx.x < B) score[tile * 4 + threadIdx.x] = sum / (float)(T * D);
    }
    if (threadIdx.x == 0) __hip_atomic_store(tile_cnt + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ------------------------------------------------------------------------------------------------ loss + its gradients
__global__ void __launch_bounds__(256) vae_loss_kernel(int B, int T, int D, int Z, const float* __restrict__ x,
                                                       const float* __restrict__ recon, const float* __restrict__ mu,
                                                       const float* __restrict__ lv, const float* __restrict__ score,
                                                       float kl_w, float* __restrict__ d_recon, float* __restrict__ d_mu,
                                                       float* __restrict__ d_lv, float* __restrict__ losses) {
    const size_t n = (size_t)B * T * D;
    const float sc = 2.0f / (float)n;
    if (d_recon)                                          // bf16 path: already written by out_fwd
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
            d_recon[i] = (recon[i] - x[i]) * sc;
    if (blockIdx.x == 0) {
        __shared__ float red[2][4];
        const int nz = B * Z;
        const float kz = kl_w / (float)nz;
        float ks = 0.f, rs = 0.f;
        // one block walks all B*Z latents: 16-byte loads, four elements per thread and pass (the scalar loop was one dependent
        // L2 round trip per element and took 19 us); per-thread summation order is fixed, so the result is reproducible
        const int nz4 = ((reinterpret_cast<uintptr_t>(mu) | reinterpret_cast<uintptr_t>(lv) | reinterpret_cast<uintptr_t>(d_mu) |
                          reinterpret_cast<uintptr_t>(d_lv)) & 15) == 0 ? (nz & ~3) : 0;
        // eight passes' loads in flight at once (B * Z = 8192 is eight passes: issued one pass per trip, each waited out its own L2
        // round trip, and this one-block kernel took 10.7 us); same per-thread order of additions
        for (int i0 = 4 * threadIdx.x; i0 < nz4; i0 += 8 * 4 * 256) {
            f32x4 m8[8], l8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * 4 * 256, ic = i < nz4 ? i : 4 * (int)threadIdx.x;
                m8[u] = *reinterpret_cast<const f32x4*>(mu + ic); l8[u] = *reinterpret_cast<const f32x4*>(lv + ic);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * 4 * 256;
                if (i < nz4) {
                    f32x4 gm, gl;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float e = expf(l8[u][k]);
                        ks += 1.0f + l8[u][k] - m8[u][k] * m8[u][k] - e;
                        gm[k] = kz * m8[u][k]; gl[k] = -0.5f * kz * (1.0f - e);
                    }
                    if (d_mu) { *reinterpret_cast<f32x4*>(d_mu + i) = gm; *reinterpret_cast<f32x4*>(d_lv + i) = gl; }
                }
            }
        }
        for (int i = nz4 + threadIdx.x; i < nz; i += 256) {
            const float m = mu[i], l = lv[i], e = expf(l);
            ks += 1.0f + l - m * m - e;
            if (d_mu) { d_mu[i] = kz * m; d_lv[i] = -0.5f * kz * (1.0f - e); }
        }
        {
            float sv[4];                                  // B <= 1024 windows in one round of loads, more in a loop
            for (int i0 = threadIdx.x; i0 < B; i0 += 4 * 256) {
#pragma unroll
                for (int u = 0; u < 4; ++u) { const int i = i0 + u * 256; sv[u] = score[i < B ? i : 0]; }
#pragma unroll
                for (int u = 0; u < 4; ++u) if (i0 + u * 256 < B) rs += sv[u];
            }
        }
        ks = wave_sum(ks); rs = wave_sum(rs);
        if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = ks; red[1][threadIdx.x >> 6] = rs; }
        __syncthreads();
        if (threadIdx.x == 0) {
            const float kl = -0.5f * ((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])) / (float)nz;
            const float rc = ((red[1][0] + red[1][1]) + (red[1][2] + red[1][3])) / (float)B;
            losses[0] = rc + kl_w * kl; losses[1] = rc; losses[2] = kl;
        }
    }
}

// ------------------------------------------------------------------------------------------------ column sums
// rows per block: enough chunks to fill the chip (>= ~256 blocks) but at least 32 rows each; fixed summation order
__host__ __device__ inline int cs_rows(int R) { const int r = (R + 255) / 256; return r < 32 ? 32 : r; }
__global__ void __launch_bounds__(256) colsum_kernel(const float* __restrict__ A, int lda, int R, int N,
                                                     float* __restrict__ out) {
    // block: 64 columns x 4 row-groups over one chunk of rows; writes out[chunk*N + c]
    __shared__ float red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
    const int CS_ROWS = cs_rows(R);
    const int r0 = blockIdx.y * CS_ROWS, r1 = min(R, r0 + CS_ROWS);
    float s = 0.f;
    if (c < N)
        for (int r = r0 + g; r < r1; r += 4) s += A[(size_t)r * lda + c];
    red[g][threadIdx.x & 63] = s;
    __syncthreads();
    if (g == 0 && c < N)
        out[(size_t)blockIdx.y * N + c] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
// block = 32 outputs x 8 partial-groups: group g sums partials g, g+8, ... (4 independent chains), then a fixed-order
// combine over the groups — deterministic, and latency-tolerant even for few outputs and many partials.
__global__ void __launch_bounds__(256) reduce_partials_kernel(const float* __restrict__ partial, int nparts, size_t n_out, size_t n,
                                                              float* __restrict__ out0, float* __restrict__ out1, float* __restrict__ outB) {
    __shared__ float red[8][32];
    if (blockIdx.y == 1) { partial += n_out; out0 = outB; out1 = nullptr; }     // second product of a shared-A GEMM
    const int o = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const size_t i = (size_t)blockIdx.x * 32 + o;          // n = element stride between partials, n_out = outputs
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f, s5 = 0.f, s6 = 0.f, s7 = 0.f;
    if (i < n_out) {
        int p = grp;
        for (; p + 56 < nparts; p += 64) {
            s0 += partial[(size_t)p * n + i];
            s1 += partial[(size_t)(p + 8) * n + i];
            s2 += partial[(size_t)(p + 16) * n + i];
            s3 += partial[(size_t)(p + 24) * n + i];
            s4 += partial[(size_t)(p + 32) * n + i];
            s5 += partial[(size_t)(p + 40) * n + i];
            s6 += partial[(size_t)(p + 48) * n + i];
            s7 += partial[(size_t)(p + 56) * n + i];
        }
        for (; p < nparts; p += 8) s0 += partial[(size_t)p * n + i];
    }
    red[grp][o] = ((s0 + s1) + (s2 + s3)) + ((s4 + s5) + (s6 + s7));
    __syncthreads();
    if (grp == 0 && i < n_out) {
        const float v = ((red[0][o] + red[1][o]) + (red[2][o] + red[3][o])) + ((red[4][o] + red[5][o]) + (red[6][o] + red[7][o]));
        out0[i] = v;
        if (out1) out1[i] = v;
    }
}

// the same reduction with 16 bytes per lane (128 outputs per block; a partial row is read in 512-byte runs): same chains, same order per
// output, so the result is bit-identical to reduce_partials_kernel.  Requires n_out, n multiples of 4 and 16-byte aligned pointers.
__global__ void __launch_bounds__(256) reduce_partials_v4_kernel(const float* __restrict__ partial, int nparts, size_t n_out, size_t n,
                                                                 float* __restrict__ out0, float* __restrict__ out1, float* __restrict__ outB) {
    __shared__ f32x4 red4[8][32];
    if (blockIdx.y == 1) { partial += n_out; out0 = outB; out1 = nullptr; }     // second product of a shared-A GEMM
    const int o = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const size_t i4 = ((size_t)blockIdx.x * 32 + o) * 4;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 s0 = z4, s1 = z4, s2 = z4, s3 = z4, s4 = z4, s5 = z4, s6 = z4, s7 = z4;
    if (i4 < n_out) {
        auto ld = [&](int p) { return *reinterpret_cast<const f32x4*>(partial + (size_t)p * n + i4); };
        int p = grp;
        for (; p + 56 < nparts; p += 64) {
            s0 += ld(p); s1 += ld(p + 8); s2 += ld(p + 16); s3 += ld(p + 24);
            s4 += ld(p + 32); s5 += ld(p + 40); s6 += ld(p + 48); s7 += ld(p + 56);
        }
        for (; p < nparts; p += 8) s0 += ld(p);
    }
    red4[grp][o] = ((s0 + s1) + (s2 + s3)) + ((s4 + s5) + (s6 + s7));
    __syncthreads();
    if (grp == 0 && i4 < n_out) {
        const f32x4 v = ((red4[0][o] + red4[1][o]) + (red4[2][o] + red4[3][o])) + ((red4[4][o] + red4[5][o]) + (red4[6][o] + red4[7][o]));
        *reinterpret_cast<f32x4*>(out0 + i4) = v;
        if (out1) *reinterpret_cast<f32x4*>(out1 + i4) = v;
    }
}

// every deferred reduction of a backward stage in one launch: block -> (job, 32 outputs) through the prefix table
__global__ void __launch_bounds__(256) reduce_batch_kernel(ReduceJobs jobs) {
    __shared__ float red[8][32];
    int j = 0;
    while (j + 1 < jobs.n && (int)blockIdx.x >= jobs.first_block[j + 1]) ++j;
    const ReduceJob jb = jobs.job[j];
    const int o = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const float* __restrict__ partial = jb.partial;
    const size_t n = jb.stride;
    if (jb.vec) {
        // 16 bytes per lane: a block covers 128 outputs, a partial row is read in 512-byte runs (the 4-byte form streamed the 67 MB of a
        // split-K GEMM's partials at 3.9 TB/s); same summation order per output as below
        __shared__ f32x4 red4[8][32];
        const size_t i4 = ((size_t)(blockIdx.x - jobs.first_block[j]) * 32 + o) * 4;
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
        f32x4 s0 = z4, s1 = z4, s2 = z4, s3 = z4, s4 = z4, s5 = z4, s6 = z4, s7 = z4;
        if (i4 < jb.n_out) {
            auto ld = [&](int p) { return *reinterpret_cast<const f32x4*>(partial + (size_t)p * n + i4); };
            int p = grp;
            for (; p + 56 < jb.nparts; p += 64) {
                s0 += ld(p); s1 += ld(p + 8); s2 += ld(p + 16); s3 += ld(p + 24);
                s4 += ld(p + 32); s5 += ld(p + 40); s6 += ld(p + 48); s7 += ld(p + 56);
            }
            for (; p < jb.nparts; p += 8) s0 += ld(p);
        }
        red4[grp][o] = ((s0 + s1) + (s2 + s3)) + ((s4 + s5) + (s6 + s7));
        __syncthreads();
        if (grp == 0 && i4 < jb.n_out) {
            const f32x4 v = ((red4[0][o] + red4[1][o]) + (red4[2][o] + red4[3][o])) + ((red4[4][o] + red4[5][o]) + (red4[6][o] + red4[7][o]));
            *reinterpret_cast<f32x4*>(jb.out0 + i4) = v;
            if (jb.out1) *reinterpret_cast<f32x4*>(jb.out1 + i4) = v;
        }
        return;
    }
    const size_t i = (size_t)(blockIdx.x - jobs.first_block[j]) * 32 + o;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f, s5 = 0.f, s6 = 0.f, s7 = 0.f;
    if (i < jb.n_out) {
        int p = grp;
        for (; p + 56 < jb.nparts; p += 64) {
            s0 += partial[(size_t)p * n + i];
            s1 += partial[(size_t)(p + 8) * n + i];
            s2 += partial[(size_t)(p + 16) * n + i];
            s3 += partial[(size_t)(p + 24) * n + i];
            s4 += partial[(size_t)(p + 32) * n + i];
            s5 += partial[(size_t)(p + 40) * n + i];
            s6 += partial[(size_t)(p + 48) * n + i];
            s7 += partial[(size_t)(p + 56) * n + i];
        }
        for (; p < jb.nparts; p += 8) s0 += partial[(size_t)p * n + i];
    }
    red[grp][o] = ((s0 + s1) + (s2 + s3)) + ((s4 + s5) + (s6 + s7));
    __syncthreads();
    if (grp == 0 && i < jb.n_out) {
        const float v = ((red[0][o] + red[1][o]) + (red[2][o] + red[3][o])) + ((red[4][o] + red[5][o]) + (red[6][o] + red[7][o]));
        size_t dst = i;
        if (jb.tr_M > 0) { const size_t nn = i / jb.tr_M, mm = i - nn * jb.tr_M; dst = mm * (jb.n_out / jb.tr_M) + nn; }
        jb.out0[dst] = v;
        if (jb.out1) jb.out1[dst] = v;
    }
}

// ------------------------------------------------------------------------------------------------ on-device windowing
// out[n][t][d] = nan_to_num((series[starts[n] + t][d] - mean[d]) / std[d]) — make_windows + normalize of the reference's
// data prep (4DOF/Scripts/03_train_vae.py:98-117) without the T-fold host copy: the raw series stays on the device and every
// window row is gathered from it (HBM-bound; the series is re-read T times out of L2/MALL, the output is written once).
__global__ void __launch_bounds__(256) window_gather_kernel(const float* __restrict__ series, long long n_rows, int D,
                                                            const int* __restrict__ starts, int N, int T,
                                                            const float* __restrict__ mean, const float* __restrict__ stdv,
                                                            float clip_z, float* __restrict__ out) {
    const size_t per = (size_t)T * D, total = (size_t)N * per;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t n = i / per, r = i - n * per;
        const int d = (int)(r % D);
        long long row = (long long)starts[n] + (long long)(r / D);
        row = row < 0 ? 0 : (row >= n_rows ? n_rows - 1 : row);          // the host wrapper validates; never read out of bounds
        float v = series[(size_t)row * D + d];
        if (mean) v = (v - mean[d]) / stdv[d];
        // openLAB prep (Codes/04_train_vae.py:113-117): np.clip BEFORE nan_to_num, so +-inf becomes +-clip_z and only NaN becomes 0
        if (clip_z > 0.f && v == v) v = fminf(fmaxf(v, -clip_z), clip_z);
        out[i] = (v != v || v == INFINITY || v == -INFINITY) ? 0.f : v;  // nan_to_num(nan=0, posinf=0, neginf=0)
    }
}

// ------------------------------------------------------------------------------------------------ narrow TN product
// out[n][m] = sum_r Bn[r][n] * A[r][m]  with n < N <= 16 (channel count): dW_out = d_recon^T h_top, dW_ih(enc0) = dpre^T x.
// A is streamed once (HBM-bound, 2 columns per thread); Bn is a broadcast load.  Block partials -> reduce_partials.
constexpr int NARROW_BLOCKS = 1024;
template <typename T, int CPT> struct VecLoad;
template <> struct VecLoad<float, 2> { static __device__ __forceinline__ void ld(const float* p, float (&v)[2]) { const f32x2 t = *reinterpret_cast<const f32x2*>(p); v[0] = t[0]; v[1] = t[1]; } };
template <> struct VecLoad<float, 8> { static __device__ __forceinline__ void ld(const float* p, float (&v)[8]) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i] = a[i]; v[4 + i] = b[i]; } } };
template <> struct VecLoad<bf16_t, 2> { static __device__ __forceinline__ void ld(const bf16_t* p, float (&v)[2]) { const bf16x2 t = *reinterpret_cast<const bf16x2*>(p); v[0] = (float)t[0]; v[1] = (float)t[1]; } };
template <> struct VecLoad<bf16_t, 8> { static __device__ __forceinline__ void ld(const bf16_t* p, float (&v)[8]) {
    const bf16x8 t = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)t[i]; } };

// CPT columns per thread: 8 (one 16/32-byte vector load per row) for <= 4 channels, 2 for up to 16 channels.
template <typename TA, int NMAX, int CPT>
__global__ void __launch_bounds__(256) narrow_tn_kernel(const TA* __restrict__ A, const TA* __restrict__ A_lo, int lda, int M, const float* __restrict__ Bn, int N,
                                                        int R, float* __restrict__ partial, int with_bsum, int tile_T, int tile_B) {
    extern __shared__ float nred[];                     // [groups][NMAX][M] (+ [groups][NMAX] column sums of Bn)
    const int tpr = M / CPT;                            // threads per row
    const int groups = 256 / tpr;                       // rows in flight per block
    const int grp = threadIdx.x / tpr, m = (threadIdx.x - grp * tpr) * CPT;
    const int rows_per_block = (R + gridDim.x - 1) / gridDim.x;
    const int r0 = blockIdx.x * rows_per_block, r1 = min(R, r0 + rows_per_block);
    float acc[NMAX][CPT], bs[NMAX];
#pragma unroll
    for (int n = 0; n < NMAX; ++n) {
        bs[n] = 0.f;
#pragma unroll
        for (int c = 0; c < CPT; ++c) acc[n][c] = 0.f;
    }
    const int stride = N * M + (with_bsum ? N : 0);      // floats per block partial
    if (grp < groups) {
        // branch-free body (padding windows read row 0 and are multiplied by zero) so that the unrolled loads of several
        // rows are all in flight together: the kernel is a pure stream over A
#pragma unroll 8
        for (int r = r0 + grp; r < r1; r += groups) {
            size_t rb = (size_t)r;                        // row of Bn that pairs with row r of A
            float keep = 1.f;
            if (tile_T > 0) {
                const int blk = r / SHM_WPT, tl = blk / tile_T, tt = blk - tl * tile_T, bb = tl * SHM_WPT + (r - blk * SHM_WPT);
                const bool ok = bb < tile_B;              // else: padding window
                rb = ok ? (size_t)bb * tile_T + tt : 0;
                keep = ok ? 1.f : 0.f;
            }
            float av[CPT];
            VecLoad<TA, CPT>::ld(A + (size_t)r * lda + m, av);
            if (A_lo) {                                   // split-precision planes: the value is hi + lo (exact in fp32)
                float al[CPT];
                VecLoad<TA, CPT>::ld(A_lo + (size_t)r * lda + m, al);
#pragma unroll
                for (int c = 0; c < CPT; ++c) av[c] += al[c];
            }
#pragma unroll
            for (int n = 0; n < NMAX; ++n) {
                if (n < N) {
                    const float bv = Bn[rb * N + n] * keep;
                    bs[n] += bv;
#pragma unroll
                    for (int c = 0; c < CPT; ++c) acc[n][c] = fmaf(bv, av[c], acc[n][c]);
                }
            }
        }
#pragma unroll
        for (int n = 0; n < NMAX; ++n)
#pragma unroll
            for (int c = 0; c < CPT; ++c) nred[(grp * NMAX + n) * M + m + c] = acc[n][c];
        if (m == 0) {
#pragma unroll
            for (int n = 0; n < NMAX; ++n) nred[groups * NMAX * M + grp * NMAX + n] = bs[n];
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < N * M; e += 256) {
        const int n = e / M, mm = e - n * M;
        float sum = 0.f;
        for (int g2 = 0; g2 < groups; ++g2) sum += nred[(g2 * NMAX + n) * M + mm];
        partial[(size_t)blockIdx.x * stride + e] = sum;
    }
    if (with_bsum && threadIdx.x < N) {
        float sum = 0.f;
        for (int g2 = 0; g2 < groups; ++g2) sum += nred[groups * NMAX * M + g2 * NMAX + threadIdx.x];
        partial[(size_t)blockIdx.x * stride + N * M + threadIdx.x] = sum;
    }
}
// out2[m][n] = in[n][m]
__global__ void __launch_bounds__(256) transpose_small_kernel(const float* __restrict__ in, int N, int M, float* __restrict__ out) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e < N * M) { const int n = e / M, m = e - n * M; out[(size_t)m * N + n] = in[e]; }
}

// ------------------------------------------------------------------------------------------------ weight packing
__global__ void __launch_bounds__(256) pack_fwd_f32_kernel(const float* __restrict__ W, int H, int K, float* __restrict__ dst) {
    const int Kp = (K + 15) & ~15, KQ = Kp >> 2, S4 = KQ >> 2;
    const size_t n = (size_t)4 * H * Kp;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int e = i & 3, lane = (i >> 2) & 63;
        size_t r = i >> 8;
        const int g = r & 3; r >>= 2;
        const int s4 = r % S4, ht = r / S4;
        const int row = g * H + ht * 16 + (lane & 15), k = (lane >> 4) * KQ + 4 * s4 + e;
        dst[i] = k < K ? W[(size_t)row * K + k] : 0.f;
    }
}
__global__ void __launch_bounds__(256) pack_bwd_f32_kernel(const float* __restrict__ W, int H, int M, float* __restrict__ dst) {
    const int S4 = H >> 2, Mp = (M + 15) & ~15;
    const size_t n = (size_t)Mp * 4 * H;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int e = i & 3, lane = (i >> 2) & 63;
        const size_t r = i >> 8;
        const int s4 = r % S4, mt = r / S4;
        const int k = (lane >> 4) * H + 4 * s4 + e, m = mt * 16 + (lane & 15);
        dst[i] = m < M ? W[(size_t)k * M + m] : 0.f;
    }
}
__global__ void __launch_bounds__(256) pack_fwd_bf16_kernel(const float* __restrict__ W, int H, int K, bf16_t* __restrict__ dst) {
    const int Kp = (K + 31) & ~31, KQ = Kp >> 2, KS = KQ >> 3;
    const size_t n = (size_t)4 * H * Kp;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int e = i & 7, lane = (i >> 3) & 63;
        size_t r = i >> 9;
        const int g = r & 3; r >>= 2;
        const int s = r % KS, ht = r / KS;
        const int row = g * H + ht * 16 + (lane & 15), k = (lane >> 4) * KQ + 8 * s + e;
        dst[i] = (bf16_t)(k < K ? W[(size_t)row * K + k] : 0.f);
    }
}
__global__ void __launch_bounds__(256) pack_bwd_bf16_kernel(const float* __restrict__ W, int H, int M, bf16_t* __restrict__ dst) {
    const int KS = H >> 3, Mp = (M + 15) & ~15;
    const size_t n = (size_t)Mp * 4 * H;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int e = i & 7, lane = (i >> 3) & 63;
        const size_t r = i >> 9;
        const int s = r % KS, mt = r / KS;
        const int k = (lane >> 4) * H + 8 * s + e, m = mt * 16 + (lane & 15);
        dst[i] = (bf16_t)(m < M ? W[(size_t)k * M + m] : 0.f);
    }
}
// every per-step re-layout of the parameters in ONE launch: blockIdx.y selects the job
__global__ void __launch_bounds__(256) pack_all_kernel(PackJobs jobs) {
    const PackJob jb = jobs.job[blockIdx.y];
    const size_t stride = (size_t)gridDim.x * 256, i0 = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int H = jb.H;
    // kinds 5 / 6: the LO part of the split-precision pair, bf16(w - bf16(w)), in the layout of kinds 0 / 1
    auto part = [&](float w, bool lo) { const bf16_t hi = (bf16_t)w; return lo ? (bf16_t)(w - (float)hi) : hi; };
    if (jb.kind == 0 || jb.kind == 5) {            // forward operand, bf16 (layout in kernels.hpp)
        const int K = jb.K, Kp = (K + 31) & ~31, KQ = Kp >> 2, KS = KQ >> 3;
        bf16_t* dst = reinterpret_cast<bf16_t*>(jb.dst);
        for (size_t i = i0; i < (size_t)4 * H * Kp; i += stride) {
            const int e = i & 7, lane = (i >> 3) & 63;
            size_t r = i >> 9;
            const int g = r & 3; r >>= 2;
            const int s = r % KS, ht = r / KS;
            const int row = g * H + ht * 16 + (lane & 15), k = (lane >> 4) * KQ + 8 * s + e;
            dst[i] = part(k < K ? jb.src[(size_t)row * K + k] : 0.f, jb.kind == 5);
        }
    } else if (jb.kind == 1 || jb.kind == 6) {     // backward operand (W^T), bf16
        const int M = jb.K, KS = H >> 3, Mp = (M + 15) & ~15;
        bf16_t* dst = reinterpret_cast<bf16_t*>(jb.dst);
        for (size_t i = i0; i < (size_t)Mp * 4 * H; i += stride) {
            const int e = i & 7, lane = (i >> 3) & 63;
            const size_t r = i >> 9;
            const int s = r % KS, mt = r / KS;
            const int k = (lane >> 4) * H + 8 * s + e, m = mt * 16 + (lane & 15);
            dst[i] = part(m < M ? jb.src[(size_t)k * M + m] : 0.f, jb.kind == 6);
        }
    } else if (jb.kind == 7 || jb.kind == 8) {     // forward operand as fp16 hi (7) / fp8 residual (8): W ~ fp16(W) + e4m3((W - fp16(W)) 2^15) 2^-15
        const int K = jb.K, Kp = (K + 31) & ~31, KQ = Kp >> 2, KS = KQ >> 3;
        for (size_t i = i0; i < (size_t)4 * H * Kp; i += stride) {
            const int e = i & 7, lane = (i >> 3) & 63;
            size_t r = i >> 9;
            const int g = r & 3; r >>= 2;
            const int s = r % KS, ht = r / KS;
            const int row = g * H + ht * 16 + (lane & 15), k = (lane >> 4) * KQ + 8 * s + e;
            const float w = k < K ? jb.src[(size_t)row * K + k] : 0.f;
            const _Float16 hi = (_Float16)w;
            if (jb.kind == 7) reinterpret_cast<_Float16*>(jb.dst)[i] = hi;
            else {
                const float d = fminf(fmaxf((w - (float)hi) * SHM_W8_SCALE, -SHM_F8_MAX), SHM_F8_MAX);
                reinterpret_cast<unsigned char*>(jb.dst)[i] = (unsigned char)(__builtin_amdgcn_cvt_pk_fp8_f32(d, 0.f, 0, false) & 0xff);
            }
        }
    } else if (jb.kind == 2) {     // bias sum b_ih + b_hh
        float* dst = reinterpret_cast<float*>(jb.dst);
        for (size_t i = i0; i < (size_t)4 * H; i += stride) dst[i] = jb.src[i] + jb.src2[i];
    } else if (jb.kind == 3) {     // plain copy of K floats (the flat parameter copy)
        float* dst = reinterpret_cast<float*>(jb.dst);
        for (size_t i = i0; i < (size_t)jb.K; i += stride) dst[i] = jb.src[i];
    } else {                       // kind 4: zero K dwords (sticky error word, zero line, sync block)
        unsigned* dst = reinterpret_cast<unsigned*>(jb.dst);
        for (size_t i = i0; i < (size_t)jb.K; i += stride) dst[i] = 0u;
    }
}
__global__ void __launch_bounds__(256) bias_sum_kernel(const float* __restrict__ a, const float* __restrict__ b, int n, float* __restrict__ dst) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = a[i] + b[i];
}

}  // namespace

int launch_head_fwd(int B, int H, int Z, int layer_norm, const float* h_last, const float* eps, const HeadParams& p,
                    const HeadBufs& hb, float* mu_out, float* lv_out, hipStream_t s, const HeadNoise& rng, const float* z_in, float* eps_store) {
    SHM_REQUIRE(H <= HMAX && Z <= ZMAX, SHM_EINVAL, "head: H=%d (<=%d) Z=%d (<=%d) unsupported", H, HMAX, Z, ZMAX);
    SHM_LAUNCH(head_fwd_kernel, dim3(ceil_div(B, 4)), dim3(256), 0, s, B, H, Z, layer_norm, h_last, eps, p, hb, mu_out, lv_out, rng, z_in, eps_store);
    SHM_LAUNCH_CHECK();
    return SHM_OK;
}
int launch_head_bwd(int B, int H, int Z, int layer_norm, const float* dpre_tsum, const float* W_ih_dec0, const float* eps,
                    const float* d_mu, const float* d_lv, const HeadParams& p, const HeadBufs& hb, hipStream_t s) {
    SHM_REQUIRE(H <= HMAX && Z <= ZMAX, SHM_EINVAL, "head: H=%d Z=%d unsupported", H, Z);
    SHM_REQUIRE((4 * H) % HB_WAVES == 0, SHM_EINVAL, "head: 4H = %d must be a multiple of %d", 4 * H, HB_WAVES);
    SHM_LAUNCH(head_bwd_kernel, dim3(ceil_div(B, 4)), dim3(64 * HB_WAVES), 0, s, B, H, Z, layer_norm, dpre_tsum, W_ih_dec0, eps, d_mu, d_lv, p, hb);
    SHM_LAUNCH_CHECK();
    return SHM_OK;
}
int launch_head_grads(int B, int H, int Z, int layer_norm, const HeadBufs& hb, float* l2h_w, float* l2h_b, float* mu_w, float* mu_b,
                      float* lv_w, float* lv_b, float* ln_w, float* ln_b, hipStream_t s) {
    HeadGradPtrs g{l2h_w, l2h_b, mu_w, mu_b, lv_w, lv_b, ln_w, ln_b};
    const int n_total = 3 * H * Z + H + 2 * Z + (layer_norm ? 2 * H : 0);
    SHM_LAUNCH(head_grads_kernel, dim3(ceil_div(n_total, 64)), dim3(64 * HG_GROUPS), 0, s, B, H, Z, layer_norm, hb, g);
    SHM_LAUNCH_CHECK();
    return SHM_OK;
}
int out_fwd_parts(int B, int T) {
    const int tiles = ceil_div(B, SHM_WPT), n_groups = (T * 4 + 15) / 16;
    int n = 256 / tiles;                                              // fill the 256 CUs
    n = n < 1 ? 1 : (n > 4 ? 4 : n);
    while (n > 1 && n_groups / n < 8) n >>= 1;                        // but keep at least a wave-pass worth of groups per part
    return n == 3 ? 2 : n;
}
int launch_out_fwd(int B, int T, int D, int H, const void* h_seq, int h_bf16, const float* Wout, const float* bout, const float* x,
                   float* recon, float* score, hipStream_t s, float* d_recon, float* score_part, unsigned* tile_cnt, const void* h_lo) {
    SHM_REQUIRE(!d_recon || h_bf16, SHM_EINVAL, "out: d_recon is written by the bf16 kernel only");
    SHM_REQUIRE(D <= DMAX && H <= HMAX, SHM_EINVAL, "out: D=%d (<=%d) H=%d unsupported", D, DMAX, H);
    if (h_bf16) {
        const int tiles = ceil_div(B, SHM_WPT);
        const int parts = (score_part && tile_cnt) ? out_fwd_parts(B, T) : 1;
        const dim3 grid(tiles, parts);
        const bf16_t* hs = (const bf16_t*)h_seq;
        const bf16_t* hl = (const bf16_t*)h_lo;
#define SHM_OUT(HH, SP) SHM_LAUNCH((out_fwd_mfma_kernel<HH, SP>), grid, dim3(64 * OUT_WAVES), 0, s, B, T, D, hs, hl, Wout, bout, x, recon, score, d_recon, score_part, tile_cnt)
        if (hl) { if (H == 128) SHM_OUT(128, true); else if (H == 64) SHM_OUT(64, true); else SHM_OUT(32, true); }
        else { if (H == 128) SHM_OUT(128, false); else if (H == 64) SHM_OUT(64, false); else SHM_OUT(32, false); }
#undef SHM_OUT
    }
    else
        SHM_LAUNCH((out_fwd_kernel<float, false>), dim3(B), dim3(256), 0, s, T, D, H, (const float*)h_seq, Wout, bout, x, recon, score);
    SHM_LAUNCH_CHECK();
    return SHM_OK;
}
int launch_vae_loss(int B, int T, int D, int Z, const float* x, const float* recon, const float* mu, const float* lv,
                    float kl_w, float* d_recon, float* d_mu, float* d_lv, float* score, float* losses, hipStream_t s) {
    const size_t n = (size_t)B * T * D;
    const int blocks = d_recon ? (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048) : 1;
    SHM_LAUNCH(vae_loss_kernel, dim3(blocks), dim3(256), 0, s, B, T, D, Z, x, recon, mu, lv, score, kl_w, d_recon, d_mu, d_lv, losses);
    SHM_LAUNCH_CHECK();
    return SHM_OK;
}
size_t colsum_scratch_elems(int R, int N) { return (size_t)ceil_div(R, cs_rows(R)) * N; }
// 16-byte form when the shapes and pointers allow it and the reduction is big enough to be a stream (bit-identical results)
static bool reduce_v4_ok(const float* partial, size_t n_out, size_t stride, const float* a, const float* b, const float* c) {
    const uintptr_t al = reinterpret_cast<uintptr_t>(partial) | reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c);
    return n_out % 4 == 0 && stride % 4 == 0 && (al & 15) == 0 && n_out >= 4096 && !getenv("SHM_REDUCE_SCALAR");
}
int launch_reduce_partials(const float* partial, int nparts, size_t n, float* out0, float* out1, hipStream_t s) {
    if (reduce_v4_ok(partial, n, n, out0, out1, nullptr))
        SHM_LAUNCH(reduce_partials_v4_kernel, dim3((unsigned)((n + 127) / 128)), dim3(256), 0, s, partial, nparts, n, n, out0, out1, (float*)nullptr);
    else
        SHM_LAUNCH(reduce_partials_kernel, dim3((unsigned)((n + 31) / 32)), dim3(256), 0, s, partial, nparts, n, n, out0, out1, (float*)nullptr);
    SHM_LAUNCH_CHECK();
    return SHM_OK;
}
int launch_reduce_partials_strided(const float* partial, int nparts, size_t n, size_t stride, float* out0, hipStream_t s) {
    if (reduce_v4_ok(partial, n, stride, out0, nullptr, nullptr))
        SHM_LAUNCH(reduce_partials_v4_kernel, dim3((unsigned)((n + 127) / 128)), dim3(256), 0, s, partial, nparts, n, stride, out0, (float*)nullptr, (float*)nullptr);
    else
        SHM_LAUNCH(reduce_partials_kernel, dim3((unsigned)((n + 31) / 32)), dim3(256), 0, s, partial, nparts, n, stride, out0, (float*)nullptr, (float*)nullptr);
    SHM_LAUNCH_CHECK();
    return SHM_OK;
}
int launch_window_gather(const float* series, long long n_rows, int D, const int* starts, int N, int T, const float* mean, const float* stdv,
                         float clip_z, float* out, hipStream_t s) {
    if (N == 0) return SHM_OK;
    const size_t total = (size_t)N * T * D;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    SHM_LAUNCH(window_gather_kernel, dim3(blocks), dim3(256), 0, s, series, n_rows, D, starts, N, T, mean, stdv, clip_z, out);
    SHM_LAUNCH_CHECK();
    return SHM_OK;
}
int reduce_jobs_add(ReduceJobs& j, const float* partial, int nparts, size_t n_out, size_t stride, float* out0, float* out1, int tr_M) {
    SHM_REQUIRE(j.n < MAX_REDUCE_JOBS, SHM_EINVAL, "reduce batch: more than %d jobs", MAX_REDUCE_JOBS);
    ReduceJob& r = j.job[j.n++];
    r.partial = partial; r.nparts = nparts; r.n_out = (unsigned)n_out; r.stride = stride; r.out0 = out0; r.out1 = out1; r.tr_M = tr_M;
    return SHM_OK;
}
int launch_reduce_batch(ReduceJobs& jobs, hipStream_t s) {
    if (jobs.n == 0) return SHM_OK;
    int blocks = 0;
    for (int i = 0; i < jobs.n; ++i) {
        ReduceJob& r = jobs.job[i];
        const uintptr_t al = reinterpret_cast<uintptr_t>(r.partial) | reinterpret_cast<uintptr_t>(r.out0) | reinterpret_cast<uintptr_t>(r.out1);
        r.vec = (r.n_out % 4 == 0 && r.stride % 4 == 0 && (al & 15) == 0 && r.tr_M == 0 && r.n_out >= 4096 && !getenv("SHM_REDUCE_SCALAR")) ? 1 : 0;
        jobs.first_block[i] = blocks;
        blocks += (int)((r.n_out + (r.vec ? 127 : 31)) / (r.vec ? 128 : 32));
    }
    jobs.first_block[jobs.n] = blocks;
    SHM_LAUNCH(reduce_batch_kernel, dim3(blocks), dim3(256), 0, s, jobs);
    SHM_LAUNCH_CHECK();
    jobs.n = 0;
    return SHM_OK;
}
// partial[split][2][n]: both products of a shared-A GEMM in one launch
int launch_reduce_partials_pair(const float* partial, int nparts, size_t n, float* outA, float* outB, hipStream_t s) {
    if (reduce_v4_ok(partial, n, 2 * n, outA, outB, nullptr))
        SHM_LAUNCH(reduce_partials_v4_kernel, dim3((unsigned)((n + 127) / 128), 2), dim3(256), 0, s, partial, nparts, n, 2 * n, outA, (float*)nullptr, outB);
    else
    SHM_LAUNCH(reduce_partials_kernel, dim3((unsigned)((n + 31) / 32), 2), dim3(256), 0, s, partial, nparts, n, 2 * n, outA, (float*)nullptr, outB);
    SHM_LAUNCH_CHECK();
    return SHM_OK;
}

int launch_colsum(const float* A, int lda, int R, int N, float* out0, float* out1, float* scratch, hipStream_t s, ReduceJobs* defer) {
    const int chunks = ceil_div(R, cs_rows(R));
    SHM_LAUNCH(colsum_kernel, dim3(ceil_div(N, 64), chunks), dim3(256), 0, s, A, lda, R, N, scratch);
    SHM_LAUNCH_CHECK();
    if (defer) return reduce_jobs_add(*defer, scratch, chunks, (size_t)N, (size_t)N, out0, out1);   // same fixed-order sum, in the batch launch
    return launch_reduce_partials(scratch, chunks, (size_t)N, out0, out1, s);
}
int launch_pack_fwd_f32(const float* W, int H, int K, float* dst, hipStream_t s) {
    const size_t n = fwd_pack_elems(H, K);
    SHM_LAUNCH(pack_fwd_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, W, H, K, dst);
    SHM_LAUNCH_CHECK();
    return SHM_OK;
}
int launch_pack_bwd_f32(const float* W, int H, int M, float* dst, hipStream_t s) {
    const size_t n = bwd_pack_elems(H, M);
    SHM_LAUNCH(pack_bwd_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, W, H, M, dst);
    SHM_LAUNCH_CHECK();
    return SHM_OK;
}
size_t narrow_tn_scratch_elems(int M, int N) { return (size_t)(NARROW_BLOCKS + 1) * (M * N + N); }
int launch_narrow_tn(const void* A, int a_bf16, int lda, int M, const float* Bn, int N, int R, float* out, int trans_out,
                     float* bsum_out, float* scratch, hipStream_t s, int tile_T, int tile_B, ReduceJobs* defer, const void* A_lo) {
    const int cpt = N <= 4 ? 8 : 2, nmax = N <= 4 ? 4 : 16;
    SHM_REQUIRE(N >= 1 && N <= 16 && M >= cpt && M <= 512 && M % cpt == 0 && 256 % (M / cpt) == 0 && lda % cpt == 0, SHM_EINVAL,
                "narrow_tn: unsupported shape M=%d N=%d lda=%d", M, N, lda);
    SHM_REQUIRE(!(bsum_out && trans_out), SHM_EINVAL, "narrow_tn: column sums only with the natural output layout");
    const int groups = 256 / (M / cpt);
    const size_t lds = sizeof(float) * (groups * nmax * M + groups * nmax);
    const int blocks = R < NARROW_BLOCKS * groups * 2 ? ceil_div(R, groups * 2) : NARROW_BLOCKS;
    const int wb = bsum_out ? 1 : 0;
    const size_t stride = (size_t)M * N + (wb ? N : 0);
#define SHM_NARROW(TA, NM, CP) SHM_LAUNCH((narrow_tn_kernel<TA, NM, CP>), dim3(blocks), dim3(256), lds, s, (const TA*)A, (const TA*)A_lo, lda, M, Bn, N, R, scratch, wb, tile_T, tile_B)
    if (a_bf16) { if (nmax == 4) SHM_NARROW(bf16_t, 4, 8); else SHM_NARROW(bf16_t, 16, 2); }
    else { if (nmax == 4) SHM_NARROW(float, 4, 8); else SHM_NARROW(float, 16, 2); }
#undef SHM_NARROW
    SHM_LAUNCH_CHECK();
    if (defer) {       // straight into the outputs, later, together with the other reductions of the stage
        int rc = reduce_jobs_add(*defer, scratch, blocks, (size_t)M * N, stride, out, nullptr, trans_out ? M : 0);
        if (!rc && wb) rc = reduce_jobs_add(*defer, scratch + (size_t)M * N, blocks, (size_t)N, stride, bsum_out, nullptr);
        return rc;
    }
    float* red = scratch + (size_t)blocks * stride;          // [N*M (+N)] reduced
    int rc = launch_reduce_partials(scratch, blocks, stride, red, nullptr, s);
    if (rc) return rc;
    if (trans_out) {
        SHM_LAUNCH(transpose_small_kernel, dim3(ceil_div(N * M, 256)), dim3(256), 0, s, red, N, M, out);
        SHM_LAUNCH_CHECK();
    } else {
        SHM_HIP_CHECK(hipMemcpyAsync(out, red, sizeof(float) * M * N, hipMemcpyDeviceToDevice, s));
        if (wb) SHM_HIP_CHECK(hipMemcpyAsync(bsum_out, red + (size_t)M * N, sizeof(float) * N, hipMemcpyDeviceToDevice, s));
    }
    return SHM_OK;
}
int launch_pack_fwd_bf16(const float* W, int H, int K, bf16_t* dst, hipStream_t s) {
    const size_t n = fwd_pack_elems_bf16(H, K);
    SHM_LAUNCH(pack_fwd_bf16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, W, H, K, dst);
    SHM_LAUNCH_CHECK();
    return SHM_OK;
}
int launch_pack_bwd_bf16(const float* W, int H, int M, bf16_t* dst, hipStream_t s) {
    const size_t n = bwd_pack_elems_bf16(H, M);
    SHM_LAUNCH(pack_bwd_bf16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, W, H, M, dst);
    SHM_LAUNCH_CHECK();
    return SHM_OK;
}
int launch_pack_all(const PackJobs& jobs, hipStream_t s) {
    if (jobs.n == 0) return SHM_OK;
    SHM_LAUNCH(pack_all_kernel, dim3(64, jobs.n), dim3(256), 0, s, jobs);
    SHM_LAUNCH_CHECK();
    return SHM_OK;
}
int launch_bias_sum(const float* b_ih, const float* b_hh, int n, float* dst, hipStream_t s) {
    SHM_LAUNCH(bias_sum_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, s, b_ih, b_hh, n, dst);
    SHM_LAUNCH_CHECK();
    return SHM_OK;
}

}  // namespace shm
