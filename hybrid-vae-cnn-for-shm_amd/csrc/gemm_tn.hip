// gemm_tn.hip — weight-gradient GEMM  C[M,N] = sum_r A[r,M]^T B[r,N]  on fp32 MFMA (v_mfma_f32_16x16x4_f32), gfx950.
//
// This is the deferred half of BPTT (reference: autograd's accumulation into weight_ih/weight_hh.grad during
// loss.backward(), 4DOF/Scripts/03_train_vae.py:269): A = stored pre-activation gradients dpre[B*T, 4H],
// B = the layer's inputs (x_t, h_{t-1}) [B*T, In|H].  The contraction runs over r = (window, step), i.e. 131 072
// rows at the north-star shape, against a tiny 512x128 output, so the grid is split along r (split-K) and the
// partial tiles are summed in a fixed order by reduce_partials (deterministic, no float atomics).
// Options used by the LSTM: shift_T (B row r comes from step t-1, zero at t = 0), dropout mask on B
// (inter-layer dropout of the layer input), transposed output.
#include "kernels.hpp"

namespace shm {

namespace {

constexpr int KC = 32;        // contraction rows staged per iteration

// 4 waves arranged WAVES_M x WAVES_N; each wave computes WMT x WNT tiles of 16x16.
template <int WMT, int WNT, int WAVES_M, int WAVES_N, typename TA, typename TB>
__global__ void __launch_bounds__(256) gemm_tn_f32_kernel(GemmTnArgs a, int tiles_n, int rows_per_split, int nsplit) {
    const TA* Ap = reinterpret_cast<const TA*>(a.A);
    const TB* Bp = reinterpret_cast<const TB*>(a.Bm);
    constexpr int MT = 16 * WMT * WAVES_M, NT = 16 * WNT * WAVES_N;
    constexpr int LDA = MT + 16, LDB = NT + 16 + (((NT + 16) % 32) == 0 ? 16 : 0);   // row stride % 32 == 16: b32 reads conflict-free
    __shared__ float sA[KC][LDA];
    __shared__ float sB[KC][LDB];

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, q = lane >> 4, i16 = lane & 15;
    const int wm = w / WAVES_N, wn = w % WAVES_N;
    const int tile = blockIdx.x, tm = tile / tiles_n, tn = tile % tiles_n;
    const int m0 = tm * MT, n0 = tn * NT;
    const int split = blockIdx.y;
    const int r_begin = split * rows_per_split;
    const int r_end = min(a.R, r_begin + rows_per_split);

    const uint32_t thr = drop_threshold(a.drop_p);
    const bool drop = a.drop_p > 0.f;
    const float dscale = drop ? 1.0f / (1.0f - a.drop_p) : 1.0f;

    f32x4 acc[WMT][WNT];
#pragma unroll
    for (int i = 0; i < WMT; ++i)
#pragma unroll
        for (int jn = 0; jn < WNT; ++jn) acc[i][jn] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int r0 = r_begin; r0 < r_end; r0 += KC) {
        // ---- stage A[r0..r0+KC) x [m0..m0+MT) and B likewise (element-guarded; rows are contiguous in memory)
        for (int e = tid; e < KC * MT; e += 256) {
            const int kk = e / MT, mm = e - kk * MT, r = r0 + kk, m = m0 + mm;
            sA[kk][mm] = (r < r_end && m < a.M) ? (float)Ap[(size_t)r * a.lda + m] : 0.f;
        }
        for (int e = tid; e < KC * NT; e += 256) {
            const int kk = e / NT, nn = e - kk * NT, r = r0 + kk, n = n0 + nn;
            float v = 0.f;
            if (r < r_end && n < a.N) {
                if (a.shift_T > 0) {
                    if (r % a.shift_T != 0) v = (float)Bp[(size_t)(r - 1) * a.ldb + n];
                } else {
                    v = (float)Bp[(size_t)r * a.ldb + n];
                    if (drop) v = drop_hash(a.drop_seed, a.drop_stack, (uint32_t)((size_t)r * a.N + n)) >= thr ? v * dscale : 0.f;
                }
            }
            sB[kk][nn] = v;
        }
        __syncthreads();
#pragma unroll
        for (int k4 = 0; k4 < KC / 4; ++k4) {
            float av[WMT], bv[WNT];
#pragma unroll
            for (int i = 0; i < WMT; ++i) av[i] = sA[k4 * 4 + q][(wm * WMT + i) * 16 + i16];
#pragma unroll
            for (int jn = 0; jn < WNT; ++jn) bv[jn] = sB[k4 * 4 + q][(wn * WNT + jn) * 16 + i16];
#pragma unroll
            for (int i = 0; i < WMT; ++i)
#pragma unroll
                for (int jn = 0; jn < WNT; ++jn) acc[i][jn] = mfma_f32_16x16x4(av[i], bv[jn], acc[i][jn]);
        }
        __syncthreads();
    }

    // ---- epilogue: D[row = 4q + r][col = i16]
#pragma unroll
    for (int i = 0; i < WMT; ++i)
#pragma unroll
        for (int jn = 0; jn < WNT; ++jn)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ml = (wm * WMT + i) * 16 + q * 4 + r, nl = (wn * WNT + jn) * 16 + i16;
                const int m = m0 + ml, n = n0 + nl;
                if (nsplit == 1) {
                    if (m < a.M && n < a.N) {
                        if (a.trans_out) a.C[(size_t)n * a.ldc + m] = acc[i][jn][r];
                        else a.C[(size_t)m * a.ldc + n] = acc[i][jn][r];
                    }
                } else {
                    // partial[split][m][n] over the padded-free logical matrix (only valid elements stored)
                    if (m < a.M && n < a.N) {
                        const size_t o = a.trans_out ? (size_t)n * a.M + m : (size_t)m * a.N + n;
                        a.partial[(size_t)split * a.M * a.N + o] = acc[i][jn][r];
                    }
                }
            }
}

struct Geo { int tiles_m, tiles_n, nsplit, rows_per_split; bool narrow; };

Geo geometry(int R, int M, int N) {
    Geo g;
    g.narrow = N <= 16;
    const int MT = 64, NT = g.narrow ? 16 : 64;
    g.tiles_m = ceil_div(M, MT);
    g.tiles_n = ceil_div(N, NT);
    const int tiles = g.tiles_m * g.tiles_n;
    int nsplit = 1024 / tiles;                        // ~4 workgroups per CU
    const int max_split = ceil_div(R, 4 * KC);        // at least 128 rows per split
    if (nsplit > max_split) nsplit = max_split;
    if (nsplit < 1) nsplit = 1;
    g.rows_per_split = round_up(ceil_div(R, nsplit), KC);
    g.nsplit = ceil_div(R, g.rows_per_split);
    return g;
}

}  // namespace

size_t gemm_tn_partial_elems(int R, int M, int N) {
    const Geo g = geometry(R, M, N);
    return g.nsplit > 1 ? (size_t)g.nsplit * M * N : 0;
}

int launch_gemm_tn_f32(const GemmTnArgs& a, hipStream_t s, ReduceJobs* defer) {
    SHM_REQUIRE(a.R > 0 && a.M > 0 && a.N > 0, SHM_EINVAL, "gemm_tn: empty shape R=%d M=%d N=%d", a.R, a.M, a.N);
    const Geo g = geometry(a.R, a.M, a.N);
    SHM_REQUIRE(g.nsplit == 1 || (a.partial && a.partial_elems >= (size_t)g.nsplit * a.M * a.N), SHM_EWORKSPACE,
                "gemm_tn: split-K scratch too small");
    SHM_REQUIRE(a.ldc == (a.trans_out ? a.M : a.N) || g.nsplit == 1, SHM_EINVAL, "gemm_tn: split-K needs a dense output");
    dim3 grid(g.tiles_m * g.tiles_n, g.nsplit);
#define SHM_GEMM_LAUNCH(TA, TB)                                                                                              \
    do {                                                                                                                     \
        if (g.narrow)                                                                                                        \
            SHM_LAUNCH((gemm_tn_f32_kernel<1, 1, 4, 1, TA, TB>), grid, dim3(256), 0, s, a, g.tiles_n, g.rows_per_split, g.nsplit); \
        else                                                                                                                 \
            SHM_LAUNCH((gemm_tn_f32_kernel<2, 2, 2, 2, TA, TB>), grid, dim3(256), 0, s, a, g.tiles_n, g.rows_per_split, g.nsplit); \
    } while (0)
    if (a.a_bf16 && a.b_bf16) SHM_GEMM_LAUNCH(bf16_t, bf16_t);
    else if (a.a_bf16) SHM_GEMM_LAUNCH(bf16_t, float);
    else if (a.b_bf16) SHM_GEMM_LAUNCH(float, bf16_t);
    else SHM_GEMM_LAUNCH(float, float);
#undef SHM_GEMM_LAUNCH
    SHM_LAUNCH_CHECK();
    if (g.nsplit > 1) {
        if (defer) return reduce_jobs_add(*defer, a.partial, g.nsplit, (size_t)a.M * a.N, (size_t)a.M * a.N, a.C, nullptr);
        return launch_reduce_partials(a.partial, g.nsplit, (size_t)a.M * a.N, a.C, nullptr, s);
    }
    return SHM_OK;
}

}  // namespace shm
