// gemm_tn_bf16.hip — weight-gradient GEMM  C[M,N] = sum_r A[r,M]^T B[r,N]  on bf16 MFMA (v_mfma_f32_16x16x32_bf16),
// fp32 accumulate; gfx950.  bf16 twin of gemm_tn.hip for the wide products (N = H) of the deferred BPTT half
// (reference: autograd accumulation into weight_ih/weight_hh.grad, 4DOF/Scripts/03_train_vae.py:269).
//
// Both operands arrive K-major (row = contraction index r) — the layout the LSTM kernels write coalesced — while an
// MFMA fragment wants 8 consecutive k per lane.  The tiles are staged in LDS exactly as they sit in HBM
// ([32 rows][128 cols] bf16, 16-byte row pad) and every fragment is formed by two ds_read_b64_tr_b16 transpose
// reads (4 rows x 16 columns each): no shuffles, no second layout in HBM.
// Workgroup = 4 waves (2x2), tile 128x128, wave 64x64 = 4x4 MFMA tiles; split-K over r with an ordered reduction.
#include "kernels.hpp"

namespace shm {

namespace {

constexpr int KC = 32, MT = 128, NT = 128, LD = 136;    // LD*2 B = 272: 2-way at worst on the transpose reads

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ bf16x8 tr_frag(const bf16_t* p_lo, const bf16_t* p_hi) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p_lo));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p_hi));
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

template <typename T>
__device__ __forceinline__ bf16x8 load8(const T* p);
template <>
__device__ __forceinline__ bf16x8 load8<bf16_t>(const bf16_t* p) { return *reinterpret_cast<const bf16x8*>(p); }
template <>
__device__ __forceinline__ bf16x8 load8<float>(const float* p) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
    const bf16x4 x = cvt_bf16x4(a), y = cvt_bf16x4(b);
    return __builtin_shufflevector(x, y, 0, 1, 2, 3, 4, 5, 6, 7);
}

template <typename TB>
__global__ void __launch_bounds__(256) gemm_tn_bf16_kernel(GemmTnArgsB a, int tiles_n, int rows_per_split, int nsplit) {
    __shared__ __attribute__((aligned(16))) bf16_t sA[KC * LD];
    __shared__ __attribute__((aligned(16))) bf16_t sB[KC * LD];
    const bf16_t* A = reinterpret_cast<const bf16_t*>(a.A);
    const TB* Bm = reinterpret_cast<const TB*>(a.Bm);

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int g = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3, i16 = lane & 15;
    const int wm = w >> 1, wn = w & 1;
    const int tile = blockIdx.x, tm = tile / tiles_n, tn = tile % tiles_n;
    const int m0 = tm * MT, n0 = tn * NT;
    const int split = blockIdx.y;
    const int r_begin = split * rows_per_split, r_end = min(a.R, r_begin + rows_per_split);

    const uint32_t thr = drop_threshold(a.drop_p);
    const bool drop = a.drop_p > 0.f;
    const float dscale = drop ? 1.0f / (1.0f - a.drop_p) : 1.0f;
    const bf16x8 zero8 = {(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jn = 0; jn < 4; ++jn) acc[i][jn] = f32x4{0.f, 0.f, 0.f, 0.f};

    // transpose-read addresses: lane 4q'+p of a 16-lane group supplies row q', columns 4p..4p+3 of the 4x16 block
    const int tr_row = 8 * g + qq, tr_col = 4 * pp;

    for (int r0 = r_begin; r0 < r_end; r0 += KC) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int e = tid + it * 256, row = e >> 4, c8 = (e & 15) * 8, r = r0 + row;
            bf16x8 va = zero8, vb = zero8;
            if (r < r_end) {
                if (m0 + c8 < a.M) va = *reinterpret_cast<const bf16x8*>(A + (size_t)r * a.lda + m0 + c8);
                if (n0 + c8 < a.N) {
                    if (a.shift_T > 0) {     // h_{t-1}: rows are tile-blocked (row = (tile*T + t)*16 + j) -> 16 rows back, zero at t = 0
                        if ((r / a.shift_rows) % a.shift_T != 0) vb = load8<TB>(Bm + (size_t)(r - a.shift_rows) * a.ldb + n0 + c8);
                    } else {
                        vb = load8<TB>(Bm + (size_t)r * a.ldb + n0 + c8);
                        if (drop) {
                            const uint32_t idx = (uint32_t)((size_t)r * a.N + n0 + c8);
#pragma unroll
                            for (int k = 0; k < 8; ++k)
                                vb[k] = drop_hash(a.drop_seed, a.drop_stack, idx + k) >= thr ? (bf16_t)((float)vb[k] * dscale) : (bf16_t)0.f;
                        }
                    }
                }
            }
            *reinterpret_cast<bf16x8*>(sA + row * LD + c8) = va;
            *reinterpret_cast<bf16x8*>(sB + row * LD + c8) = vb;
        }
        __syncthreads();
        bf16x8 af[4], bfr[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bf16_t* p = sA + tr_row * LD + wm * 64 + i * 16 + tr_col;
            af[i] = tr_frag(p, p + 4 * LD);
        }
#pragma unroll
        for (int jn = 0; jn < 4; ++jn) {
            const bf16_t* p = sB + tr_row * LD + wn * 64 + jn * 16 + tr_col;
            bfr[jn] = tr_frag(p, p + 4 * LD);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int jn = 0; jn < 4; ++jn) acc[i][jn] = mfma_bf16_16x16x32(af[i], bfr[jn], acc[i][jn]);
        __syncthreads();
    }

#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jn = 0; jn < 4; ++jn)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm * 64 + i * 16 + g * 4 + r, n = n0 + wn * 64 + jn * 16 + i16;
                if (m < a.M && n < a.N) {
                    float* dst = nsplit == 1 ? a.C : a.partial + (size_t)split * a.M * a.N;
                    dst[(size_t)m * a.N + n] = acc[i][jn][r];
                }
            }
}

struct Geo { int tiles_m, tiles_n, nsplit, rows_per_split; };
Geo geometry(int R, int M, int N) {
    Geo g;
    g.tiles_m = ceil_div(M, MT); g.tiles_n = ceil_div(N, NT);
    const int tiles = g.tiles_m * g.tiles_n;
    int nsplit = 512 / tiles;
    const int max_split = ceil_div(R, 8 * KC);
    if (nsplit > max_split) nsplit = max_split;
    if (nsplit < 1) nsplit = 1;
    g.rows_per_split = round_up(ceil_div(R, nsplit), KC);
    g.nsplit = ceil_div(R, g.rows_per_split);
    return g;
}

}  // namespace

size_t gemm_tn_bf16_partial_elems(int R, int M, int N) {
    const Geo g = geometry(R, M, N);
    return g.nsplit > 1 ? (size_t)g.nsplit * M * N : 0;
}

int launch_gemm_tn_bf16(const GemmTnArgsB& a, hipStream_t s) {
    SHM_REQUIRE(a.R > 0 && a.M > 0 && a.N > 0, SHM_EINVAL, "gemm_tn_bf16: empty shape");
    SHM_REQUIRE((a.M & 7) == 0 && (a.N & 7) == 0 && (a.lda & 7) == 0 && (a.ldb & 7) == 0, SHM_EINVAL,
                "gemm_tn_bf16: M, N, lda, ldb must be multiples of 8 (got %d %d %d %d)", a.M, a.N, a.lda, a.ldb);
    const Geo g = geometry(a.R, a.M, a.N);
    SHM_REQUIRE(g.nsplit == 1 || (a.partial && a.partial_elems >= (size_t)g.nsplit * a.M * a.N), SHM_EWORKSPACE,
                "gemm_tn_bf16: split-K scratch too small");
    dim3 grid(g.tiles_m * g.tiles_n, g.nsplit);
    if (a.b_f32)
        hipLaunchKernelGGL(gemm_tn_bf16_kernel<float>, grid, dim3(256), 0, s, a, g.tiles_n, g.rows_per_split, g.nsplit);
    else
        hipLaunchKernelGGL(gemm_tn_bf16_kernel<bf16_t>, grid, dim3(256), 0, s, a, g.tiles_n, g.rows_per_split, g.nsplit);
    SHM_LAUNCH_CHECK();
    if (g.nsplit > 1) return launch_reduce_partials(a.partial, g.nsplit, (size_t)a.M * a.N, a.C, nullptr, s);
    return SHM_OK;
}

}  // namespace shm
