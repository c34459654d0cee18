// gemm_tn_bf16.hip — weight-gradient GEMM on bf16 MFMA (v_mfma_f32_16x16x32_bf16), fp32 accumulate; gfx950.
//   C_k[M,N] = sum_r A[r,M]^T B_k[r,N],  k = 0..NB-1   (one launch serves every product that shares A)
// The deferred half of BPTT (reference: autograd accumulation into weight_ih/weight_hh.grad, 4DOF/Scripts/03_train_vae.py:269):
// A = dpre_seq[R,4H], B_0 = the layer input sequence (inter-layer dropout mask re-applied), B_1 = the layer's own h
// sequence taken one step back (h_{t-1}).  R = 131 072 rows at the north-star shape against a 512 x 128 output, so the
// kernel is HBM-bound on A: both products of a layer run in ONE launch and A is read once.
//
// Both operands arrive K-major (row = contraction index) — the layout the LSTM kernels write coalesced — while an MFMA
// fragment wants 8 consecutive k per lane: the tiles are staged in LDS exactly as they sit in HBM ([64 rows][128 cols]
// bf16, 16-byte row pad) and every fragment is formed by two ds_read_b64_tr_b16 transpose reads (4 rows x 16 columns).
// Workgroup = 4 waves (2x2), tile 128 x 128, wave 64 x 64 = 4 x 4 MFMA tiles, 64 rows (two k-steps) per stage; the next
// stage's global loads are issued into registers before the current stage is consumed.  Split-K over r with partial tiles
// summed in a fixed order (reduce_partials): deterministic.  Blocks of one split share A and B, so the 1-D grid is
// dealt so that they land on the same XCD (blockIdx % 8 — a speed hint only, nothing depends on it).
#include "kernels.hpp"

namespace shm {

namespace {

constexpr int KC = 64, MT = 128, NT = 128, LD = 136;    // LD*2 B = 272: 2-way at worst on the transpose reads

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ bf16x8 tr_frag(const bf16_t* p_lo, const bf16_t* p_hi) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p_lo));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p_hi));
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

__global__ void __launch_bounds__(256) gemm_tn_bf16_kernel(GemmTnArgsB a, int tiles_m, int tiles_n, int rows_per_split, int nsplit) {
    __shared__ __attribute__((aligned(16))) bf16_t sA[KC * LD];
    __shared__ __attribute__((aligned(16))) bf16_t sB[KC * LD];
    const bf16_t* A = reinterpret_cast<const bf16_t*>(a.A);

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int g = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3, i16 = lane & 15;
    const int wm = w >> 1, wn = w & 1;
    // 1-D grid -> (split, tile): all tiles of a split on one XCD (blockIdx % 8)
    const int tiles = tiles_m * tiles_n * a.NB;
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    const int tile = local % tiles, split = (local / tiles) * 8 + xcd;
    if (split >= nsplit) return;
    const int kb = tile / (tiles_m * tiles_n), tmn = tile - kb * tiles_m * tiles_n, tm = tmn / tiles_n, tn = tmn - tm * tiles_n;
    const int m0 = tm * MT, n0 = tn * NT;
    const bf16_t* Bm = reinterpret_cast<const bf16_t*>(kb == 0 ? a.Bm : a.Bm2);
    const bf16_t* const A_hi = A;
    const bf16_t* const A_lo = reinterpret_cast<const bf16_t*>(a.A_lo);
    const bf16_t* const B_hi = Bm;
    const bf16_t* const B_lo = reinterpret_cast<const bf16_t*>(kb == 0 ? a.Bm_lo : a.Bm2_lo);
    const int nseg = a.nseg > 1 ? 3 : 1;                 // split-precision operands: (A_hi,B_hi), (A_hi,B_lo), (A_lo,B_hi) over the same rows
    const int shift_T = kb == 0 ? a.shift_T : a.shift_T2;
    const float drop_p = kb == 0 ? a.drop_p : 0.f;
    const int r_begin = split * rows_per_split, r_end = min(a.R, r_begin + rows_per_split);

    const uint32_t thr = drop_threshold(drop_p);
    const bool drop = drop_p > 0.f;
    const float dscale = drop ? 1.0f / (1.0f - drop_p) : 1.0f;
    const bf16x8 zero8 = {(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jn = 0; jn < 4; ++jn) acc[i][jn] = f32x4{0.f, 0.f, 0.f, 0.f};

    // staging: 64 rows x 128 cols = 1024 chunks of 16 B per operand, 4 per thread; chunk e -> (row e/16, cols 8*(e%16)..)
    bf16x8 ra[4], rb[4];
    const int nst = (r_end - r_begin + KC - 1) / KC, total = nst * nseg;
    auto load_stage = [&](int k) {
        const int seg = k / nst, r0 = r_begin + (k - seg * nst) * KC;
        A = seg == 2 ? A_lo : A_hi;
        Bm = seg == 1 ? B_lo : B_hi;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int e = tid + it * 256, row = e >> 4, c8 = (e & 15) * 8, r = r0 + row;
            ra[it] = zero8; rb[it] = zero8;
            if (r < r_end) {
                if (m0 + c8 < a.M) ra[it] = *reinterpret_cast<const bf16x8*>(A + (size_t)r * a.lda + m0 + c8);
                if (n0 + c8 < a.N) {
                    if (shift_T > 0) {     // h_{t-1}: rows are tile-blocked -> shift_rows back, zero at t = 0
                        if ((r / a.shift_rows) % shift_T != 0) rb[it] = *reinterpret_cast<const bf16x8*>(Bm + (size_t)(r - a.shift_rows) * a.ldb + n0 + c8);
                    } else {
                        rb[it] = *reinterpret_cast<const bf16x8*>(Bm + (size_t)r * a.ldb + n0 + c8);
                        if (drop) {
                            const uint32_t idx = (uint32_t)((size_t)r * a.N + n0 + c8);
#pragma unroll
                            for (int k = 0; k < 8; ++k)
                                rb[it][k] = drop_hash(a.drop_seed, a.drop_stack, idx + k) >= thr ? (bf16_t)((float)rb[it][k] * dscale) : (bf16_t)0.f;
                        }
                    }
                }
            }
        }
    };
    auto store_stage = [&]() {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int e = tid + it * 256, row = e >> 4, c8 = (e & 15) * 8;
            *reinterpret_cast<bf16x8*>(sA + row * LD + c8) = ra[it];
            *reinterpret_cast<bf16x8*>(sB + row * LD + c8) = rb[it];
        }
    };

    // transpose-read addresses: lane 4q'+p of a 16-lane group supplies row q', columns 4p..4p+3 of the 4x16 block
    const int tr_row = 8 * g + qq, tr_col = 4 * pp;

    load_stage(0);
    for (int k = 0; k < total; ++k) {
        store_stage();
        __syncthreads();
        if (k + 1 < total) load_stage(k + 1);                // next stage in flight while this one is consumed
#pragma unroll
        for (int ks = 0; ks < KC / 32; ++ks) {
            bf16x8 af[4], bfr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bf16_t* p = sA + (ks * 32 + tr_row) * LD + wm * 64 + i * 16 + tr_col;
                af[i] = tr_frag(p, p + 4 * LD);
            }
#pragma unroll
            for (int jn = 0; jn < 4; ++jn) {
                const bf16_t* p = sB + (ks * 32 + tr_row) * LD + wn * 64 + jn * 16 + tr_col;
                bfr[jn] = tr_frag(p, p + 4 * LD);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int jn = 0; jn < 4; ++jn) acc[i][jn] = mfma_bf16_16x16x32(af[i], bfr[jn], acc[i][jn]);
        }
        __syncthreads();
    }

    float* C = kb == 0 ? a.C : a.C2;
    float* dst = nsplit == 1 ? C : a.partial + ((size_t)split * a.NB + kb) * a.M * a.N;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jn = 0; jn < 4; ++jn)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm * 64 + i * 16 + g * 4 + r, n = n0 + wn * 64 + jn * 16 + i16;
                if (m < a.M && n < a.N) dst[(size_t)m * a.N + n] = acc[i][jn][r];
            }
}

// ---------------------------------------------------------------------------------------------------------------------
// v2 — the H = 128 shape (M % 256 == 0, N == 128): the kernel is bound by how many UNIQUE HBM bytes a CU keeps in flight
// (v1 reached 2.2 TB/s: one 32 KB stage per workgroup in flight, 63 % of it re-reads of tiles other workgroups also load).
//   * tile = 256 (M) x all N columns of BOTH products: A is read once per launch, B twice;
//   * operands go HBM -> LDS by LDS-DMA (global_load_lds_dwordx4, no staging registers) into a ring of 4 stage buffers of
//     32 rows, three stages (3 x 32 KB at NB = 2) in flight per CU, retired with counted s_waitcnt vmcnt and ONE raw
//     s_barrier per stage (the wait precedes the barrier, the reads follow it; a buffer is refilled one barrier after its
//     last read);
//   * LDS-DMA writes lane-linear (1 KB per wave instruction, no padding possible), so the image is XOR-swizzled on the
//     SOURCE side: the 32-byte block b of row r lands in block b ^ f(r), f(r) = 4*((r>>3)&1) + (r&3), which makes the 8 rows
//     a half-wave touches in one ds_read_b64_tr_b16 hit 8 different bank groups;
//   * rows past the split end and the t = 0 rows of the shifted product read a zero line instead of being branched around.
constexpr int V2_KC = 32, V2_NBUF = 4, V2_MT = 256, V2_CH = 2;

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_cvoid;
__device__ __forceinline__ int v2_swz(int r) { return (((r >> 3) & 1) << 2) | (r & 3); }

// XP: a third, 16-column product sharing A (dW_ih of the fp32-input layer: Bx = the tile-blocked bf16 copy of x)
template <int NB, bool XP>
__global__ void __launch_bounds__(512) gemm_tn_bf16_v2_kernel(GemmTnArgsB a, int tiles_m, int rows_per_split, int nsplit) {
    constexpr int NT = 128 * NB;                         // B image row: [product 0 | product 1]
    constexpr int A_BYTES = V2_KC * V2_MT * 2, B_BYTES = V2_KC * NT * 2, X_BYTES = XP ? V2_KC * 16 * 2 : 0;
    constexpr int ST_BYTES = A_BYTES + B_BYTES + X_BYTES;
    constexpr int GL_A = 2, GL_B = NB;                   // LDS-DMA instructions per wave and stage
    constexpr int NJ = 2 * NB;                           // 16-column MFMA tiles per wave along N (wave tile 128 x 32*NB)
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int g = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3, i16 = lane & 15;
    const int wm = w >> 2, wn = w & 3;
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    const int tm = local % tiles_m, split = (local / tiles_m) * 8 + xcd;
    if (split >= nsplit) return;
    const int m0 = tm * V2_MT;
    const int r_begin = split * rows_per_split, r_end = min(a.R, r_begin + rows_per_split);
    const int nst = (r_end - r_begin + V2_KC - 1) / V2_KC;
    const int nseg = a.nseg > 1 ? 3 : 1, total = nst * nseg;   // split-precision operands: plane pairs (hi,hi), (hi,lo), (lo,hi) over the same rows
    const bf16_t* zeros = reinterpret_cast<const bf16_t*>(a.zeros);

    // ---- per-lane LDS-DMA sources.  A: instruction j = 2w + i covers stage rows 2j, 2j+1 (512 B each): lane -> row l>>5, chunk l&31
    int a_row[GL_A], a_col[GL_A];
#pragma unroll
    for (int i = 0; i < GL_A; ++i) {
        const int rs = 2 * (2 * w + i) + (lane >> 5), c = lane & 31;
        a_row[i] = rs; a_col[i] = (((c >> 1) ^ v2_swz(rs)) << 1 | (c & 1)) * 8;
    }
    // B: NB == 2: instruction j = 2w + i covers rows 2j, 2j+1 of [Bm | Bm2] (512 B); NB == 1: instruction w covers rows 4w..4w+3 (256 B)
    int b_row[GL_B], b_col[GL_B], b_prod[GL_B];
#pragma unroll
    for (int i = 0; i < GL_B; ++i) {
        int rs, c;
        if (NB == 2) { rs = 2 * (2 * w + i) + (lane >> 5); c = lane & 31; }
        else { rs = 4 * w + (lane >> 4); c = lane & 15; }
        const int cs = ((c >> 1) ^ v2_swz(rs)) << 1 | (c & 1);       // source chunk of the [Bm | Bm2] row
        b_row[i] = rs; b_prod[i] = cs >> 4; b_col[i] = (cs & 15) * 8;
    }
    auto issue = [&](int k) {                            // stage k -> ring slot k % NBUF (always issued: the counts stay static)
        char* buf = smem + (k & (V2_NBUF - 1)) * ST_BYTES;
        // Split-precision operands: the three plane pairs of a CHUNK of V2_CH stages (64 rows) are walked back to back, so that the second
        // read of A_hi (pairs 0 and 1) and of B_hi (pairs 0 and 2) is served by the XCD's L2 (32 workgroups x 64 KB per XCD) instead
        // of HBM: the launch then moves A_hi + A_lo + B_hi + B_lo once, not three operand passes.  Over-issued tail stages
        // (k >= total) read the zero line.
        int seg = nseg, st = 0;
        if (k < total) {
            if (nseg == 1) { seg = 0; st = k; }
            else {
                const int full = nst / V2_CH, per = V2_CH * 3;
                if (k < full * per) { const int c = k / per, r = k - c * per; seg = r / V2_CH; st = c * V2_CH + (r - seg * V2_CH); }
                else { const int rem = nst - full * V2_CH, kk = k - full * per; seg = kk / rem; st = full * V2_CH + (kk - seg * rem); }
            }
        }
        const int r0 = seg < nseg ? r_begin + st * V2_KC : r_end;
        const bf16_t* A = reinterpret_cast<const bf16_t*>(seg == 2 ? a.A_lo : a.A);
        const bool blo = seg == 1;
#pragma unroll
        for (int i = 0; i < GL_A; ++i) {
            const int r = r0 + a_row[i];
            const bf16_t* src = r < r_end ? A + (size_t)r * a.lda + m0 + a_col[i] : zeros;
            __builtin_amdgcn_global_load_lds((gbl_cvoid*)src, (lds_void*)(buf + (2 * w + i) * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < GL_B; ++i) {
            const int r = r0 + b_row[i];
            const bool second = NB == 2 ? b_prod[i] == 1 : false;
            const bf16_t* base = reinterpret_cast<const bf16_t*>(second ? (blo ? a.Bm2_lo : a.Bm2) : (blo ? a.Bm_lo : a.Bm));
            const int sh = second ? a.shift_T2 : a.shift_T;
            bool ok = r < r_end;
            int rr = ok ? r : r_begin;
            if (sh > 0) { ok = ok && ((rr / a.shift_rows) % sh != 0); rr -= a.shift_rows; }    // h_{t-1}, zero at t = 0
            else if (!second && a.bcast_T > 0) rr = (rr / (a.shift_rows * a.bcast_T)) * a.shift_rows + rr % a.shift_rows;   // one row per window
            const bf16_t* src = ok ? base + (size_t)rr * a.ldb + b_col[i] : zeros;
            __builtin_amdgcn_global_load_lds((gbl_cvoid*)src, (lds_void*)(buf + A_BYTES + ((NB == 2 ? 2 * w : 0) + (NB == 2 ? i : w)) * 1024), 16, 0, 0);
        }
        if constexpr (XP) {
            if (w == 0) {                                 // 32 rows x 32 B: lane -> row l>>1, half l&1
                const int r = r0 + (lane >> 1);
                const bf16_t* src = r < r_end ? reinterpret_cast<const bf16_t*>(blo ? a.Bx_lo : a.Bx) + (size_t)r * 16 + (lane & 1) * 8 : zeros;
                __builtin_amdgcn_global_load_lds((gbl_cvoid*)src, (lds_void*)(buf + A_BYTES + B_BYTES), 16, 0, 0);
            }
        }
    };

    f32x4 acc[8][NJ], accx[XP ? 8 : 1];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int jn = 0; jn < NJ; ++jn) acc[i][jn] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < (XP ? 8 : 1); ++i) accx[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    // transpose reads: lane (g, qq, pp) supplies row 8g + qq (+4), columns 4pp..4pp+3 of a 16-column block.
    // The reads are inline asm: for a compiler-visible ds_read hipcc waits vmcnt(0) while any LDS-DMA is in flight (it cannot
    // tell the ring slots apart), which would serialise the ring; ordering is ours — the counted vmcnt + barrier above.
    const int tr_r = 8 * g + qq, fz = v2_swz(tr_r);      // v2_swz(tr_r + 4) == v2_swz(tr_r)
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void*)smem;
    uint32_t a_off[8], b_off[NJ];
#pragma unroll
    for (int i = 0; i < 8; ++i) a_off[i] = tr_r * (V2_MT * 2) + (((wm * 8 + i) ^ fz) << 5) + pp * 8;
#pragma unroll
    for (int jn = 0; jn < NJ; ++jn) b_off[jn] = A_BYTES + tr_r * (NT * 2) + (((wn * NJ + jn) ^ fz) << 5) + pp * 8;
    auto compute = [&](int k) {
        const uint32_t buf = lds0 + (k & (V2_NBUF - 1)) * ST_BYTES;
        s16x4 alo[8], ahi[8], blo[NJ], bhi[NJ], xlo = {0, 0, 0, 0}, xhi = {0, 0, 0, 0};
        if constexpr (XP)
            asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:128"
                         : "=&v"(xlo), "=&v"(xhi) : "v"(buf + A_BYTES + B_BYTES + tr_r * 32 + pp * 8));
#pragma unroll
        for (int i = 0; i < 8; ++i)
            asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:%3"
                         : "=&v"(alo[i]), "=&v"(ahi[i]) : "v"(buf + a_off[i]), "n"(4 * V2_MT * 2));
#pragma unroll
        for (int jn = 0; jn < NJ; ++jn)
            asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:%3"
                         : "=&v"(blo[jn]), "=&v"(bhi[jn]) : "v"(buf + b_off[jn]), "n"(4 * NT * 2));
        if constexpr (NB == 2)
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(alo[0]), "+v"(ahi[0]), "+v"(alo[1]), "+v"(ahi[1]), "+v"(alo[2]), "+v"(ahi[2]), "+v"(alo[3]), "+v"(ahi[3]),
                           "+v"(alo[4]), "+v"(ahi[4]), "+v"(alo[5]), "+v"(ahi[5]), "+v"(alo[6]), "+v"(ahi[6]), "+v"(alo[7]), "+v"(ahi[7]),
                           "+v"(blo[0]), "+v"(bhi[0]), "+v"(blo[1]), "+v"(bhi[1]), "+v"(blo[2]), "+v"(bhi[2]), "+v"(blo[3]), "+v"(bhi[3]));
        else
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(alo[0]), "+v"(ahi[0]), "+v"(alo[1]), "+v"(ahi[1]), "+v"(alo[2]), "+v"(ahi[2]), "+v"(alo[3]), "+v"(ahi[3]),
                           "+v"(alo[4]), "+v"(ahi[4]), "+v"(alo[5]), "+v"(ahi[5]), "+v"(alo[6]), "+v"(ahi[6]), "+v"(alo[7]), "+v"(ahi[7]),
                           "+v"(blo[0]), "+v"(bhi[0]), "+v"(blo[NJ - 1]), "+v"(bhi[NJ - 1]), "+v"(xlo), "+v"(xhi));
        bf16x8 bfr[NJ];
#pragma unroll
        for (int jn = 0; jn < NJ; ++jn) bfr[jn] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(blo[jn], bhi[jn], 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const bf16x8 af = __builtin_bit_cast(bf16x8, __builtin_shufflevector(alo[i], ahi[i], 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
            for (int jn = 0; jn < NJ; ++jn) acc[i][jn] = mfma_bf16_16x16x32(af, bfr[jn], acc[i][jn]);
        }
        if constexpr (XP) {
            if (wn == 0) {
                const bf16x8 xf = __builtin_bit_cast(bf16x8, __builtin_shufflevector(xlo, xhi, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    accx[i] = mfma_bf16_16x16x32(__builtin_bit_cast(bf16x8, __builtin_shufflevector(alo[i], ahi[i], 0, 1, 2, 3, 4, 5, 6, 7)), xf, accx[i]);
            }
        }
    };

    issue(0); issue(1); issue(2);
    for (int k = 0; k < total; ++k) {
        // stage k has landed; k+1, k+2 stay in flight (2 x LDS-DMA instructions per wave and stage)
        if (NB == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (XP && w == 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        __builtin_amdgcn_s_barrier();                     // everyone's part of stage k is in LDS, everyone is done with slot (k+3) % 4
        issue(k + 3);
        compute(k);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // no LDS-DMA may outlive the workgroup

#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int jn = 0; jn < NJ; ++jn) {
            const int ncol = (wn * NJ + jn) * 16 + i16, kb = ncol >> 7, n = ncol & 127;
            float* C = kb == 0 ? a.C : a.C2;
            float* dst = nsplit == 1 ? C : a.partial + ((size_t)split * NB + kb) * a.M * a.N;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm * 128 + i * 16 + g * 4 + r;
                dst[(size_t)m * a.N + n] = acc[i][jn][r];
            }
        }
    if constexpr (XP) {
        if (wn == 0 && i16 < a.Nx) {
            float* dst = nsplit == 1 ? a.Cx : a.partial + (size_t)nsplit * NB * a.M * a.N + (size_t)split * a.M * a.Nx;
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) dst[(size_t)(m0 + wm * 128 + i * 16 + g * 4 + r) * a.Nx + i16] = accx[i][r];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// v3 (round 3) — the split-precision operands of the H = 128 shape: C = A_hi^T B_hi + A_hi^T B_lo + A_lo^T B_hi.
// v2 serves them by walking its rows three times (plane pairs of a 64-row chunk back to back, the re-reads out of L2): three stages,
// three barriers, three rounds of LDS-DMA issue and of fragment reads per 32 rows — and that per-stage overhead, not HBM, is what bounds
// it (0.19 ms alone for 403 MB).  Here ONE stage holds all four planes of its 32 rows (A_hi | A_lo | B_hi | B_lo [| X_hi | X_lo]); the
// three products run from that image: per 32 rows one barrier, 8 (not 12) LDS-DMA instructions per wave, 24 (not 36) fragment pairs, the
// same 96 MFMAs.  A stage is 64 KB at NB = 2 (two ring slots, one stage in flight while one is consumed) and 50 KB at NB = 1 (three).
// Same swizzle, same zero line, same split-K as v2.  Products are summed per stage (hi*hi, hi*lo, lo*hi), v2 summed them per 64-row chunk:
// results agree to fp32 rounding, not bit for bit.
template <int NB, bool XP>
__global__ void __launch_bounds__(512) gemm_tn_sp_v3_kernel(GemmTnArgsB a, int tiles_m, int rows_per_split, int nsplit) {
    constexpr int NT = 128 * NB;
    constexpr int A_BYTES = V2_KC * V2_MT * 2, B_BYTES = V2_KC * NT * 2, X_BYTES = XP ? V2_KC * 16 * 2 : 0;
    constexpr int OFF_ALO = A_BYTES, OFF_BHI = 2 * A_BYTES, OFF_BLO = 2 * A_BYTES + B_BYTES, OFF_XHI = 2 * A_BYTES + 2 * B_BYTES, OFF_XLO = OFF_XHI + X_BYTES;
    constexpr int ST_BYTES = 2 * (A_BYTES + B_BYTES + X_BYTES);
    constexpr int NBUF = NB == 2 ? 2 : 3;
    constexpr int GL_A = 2, GL_B = NB;
    constexpr int NJ = 2 * NB;
    constexpr int PS = 2 * (GL_A + GL_B);                // LDS-DMA instructions per wave and stage (wave 0 of an XP launch: + 2)
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int g = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3, i16 = lane & 15;
    const int wm = w >> 2, wn = w & 3;
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    const int tm = local % tiles_m, split = (local / tiles_m) * 8 + xcd;
    if (split >= nsplit) return;
    const int m0 = tm * V2_MT;
    const int r_begin = split * rows_per_split, r_end = min(a.R, r_begin + rows_per_split);
    const int nst = (r_end - r_begin + V2_KC - 1) / V2_KC;
    const bf16_t* zeros = reinterpret_cast<const bf16_t*>(a.zeros);

    int a_row[GL_A], a_col[GL_A];
#pragma unroll
    for (int i = 0; i < GL_A; ++i) {
        const int rs = 2 * (2 * w + i) + (lane >> 5), c = lane & 31;
        a_row[i] = rs; a_col[i] = (((c >> 1) ^ v2_swz(rs)) << 1 | (c & 1)) * 8;
    }
    int b_row[GL_B], b_col[GL_B], b_prod[GL_B];
#pragma unroll
    for (int i = 0; i < GL_B; ++i) {
        int rs, c;
        if (NB == 2) { rs = 2 * (2 * w + i) + (lane >> 5); c = lane & 31; }
        else { rs = 4 * w + (lane >> 4); c = lane & 15; }
        const int cs = ((c >> 1) ^ v2_swz(rs)) << 1 | (c & 1);
        b_row[i] = rs; b_prod[i] = cs >> 4; b_col[i] = (cs & 15) * 8;
    }
    auto issue = [&](int k) {                            // stage k -> ring slot k % NBUF (always issued: the counts stay static)
        char* buf = smem + (k % NBUF) * ST_BYTES;
        const int r0 = k < nst ? r_begin + k * V2_KC : r_end;      // over-issued tail stages read the zero line
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
            const bf16_t* A = reinterpret_cast<const bf16_t*>(pl ? a.A_lo : a.A);
#pragma unroll
            for (int i = 0; i < GL_A; ++i) {
                const int r = r0 + a_row[i];
                const bf16_t* src = r < r_end ? A + (size_t)r * a.lda + m0 + a_col[i] : zeros;
                __builtin_amdgcn_global_load_lds((gbl_cvoid*)src, (lds_void*)(buf + pl * OFF_ALO + (2 * w + i) * 1024), 16, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < GL_B; ++i) {
                const int r = r0 + b_row[i];
                const bool second = NB == 2 ? b_prod[i] == 1 : false;
                const bf16_t* base = reinterpret_cast<const bf16_t*>(second ? (pl ? a.Bm2_lo : a.Bm2) : (pl ? a.Bm_lo : a.Bm));
                const int sh = second ? a.shift_T2 : a.shift_T;
                bool ok = r < r_end;
                int rr = ok ? r : r_begin;
                if (sh > 0) { ok = ok && ((rr / a.shift_rows) % sh != 0); rr -= a.shift_rows; }
                else if (!second && a.bcast_T > 0) rr = (rr / (a.shift_rows * a.bcast_T)) * a.shift_rows + rr % a.shift_rows;
                const bf16_t* src = ok ? base + (size_t)rr * a.ldb + b_col[i] : zeros;
                __builtin_amdgcn_global_load_lds((gbl_cvoid*)src, (lds_void*)(buf + (pl ? OFF_BLO : OFF_BHI) + ((NB == 2 ? 2 * w : 0) + (NB == 2 ? i : w)) * 1024), 16, 0, 0);
            }
            if constexpr (XP) {
                if (w == 0) {
                    const int r = r0 + (lane >> 1);
                    const bf16_t* src = r < r_end ? reinterpret_cast<const bf16_t*>(pl ? a.Bx_lo : a.Bx) + (size_t)r * 16 + (lane & 1) * 8 : zeros;
                    __builtin_amdgcn_global_load_lds((gbl_cvoid*)src, (lds_void*)(buf + (pl ? OFF_XLO : OFF_XHI)), 16, 0, 0);
                }
            }
        }
    };

    f32x4 acc[8][NJ], accx[XP ? 8 : 1];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int jn = 0; jn < NJ; ++jn) acc[i][jn] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < (XP ? 8 : 1); ++i) accx[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int tr_r = 8 * g + qq, fz = v2_swz(tr_r);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void*)smem;
    uint32_t a_off[8], b_off[NJ];
#pragma unroll
    for (int i = 0; i < 8; ++i) a_off[i] = tr_r * (V2_MT * 2) + (((wm * 8 + i) ^ fz) << 5) + pp * 8;
#pragma unroll
    for (int jn = 0; jn < NJ; ++jn) b_off[jn] = tr_r * (NT * 2) + (((wn * NJ + jn) ^ fz) << 5) + pp * 8;
    auto frag = [](s16x4 lo, s16x4 hi) { return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7)); };
    auto compute = [&](int k) {
        const uint32_t buf = lds0 + (k % NBUF) * ST_BYTES;
        s16x4 alo[8], ahi[8], blo[NJ], bhi[NJ], clo[NJ], chi[NJ], xlo = {0, 0, 0, 0}, xhi = {0, 0, 0, 0}, ylo = {0, 0, 0, 0}, yhi = {0, 0, 0, 0};
        // A_hi, B_hi, B_lo (, X_hi, X_lo) fragments; inline asm for the reason given in v2
        if constexpr (XP) {
            asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:128" : "=&v"(xlo), "=&v"(xhi) : "v"(buf + OFF_XHI + tr_r * 32 + pp * 8));
            asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:128" : "=&v"(ylo), "=&v"(yhi) : "v"(buf + OFF_XLO + tr_r * 32 + pp * 8));
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
            asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:%3" : "=&v"(alo[i]), "=&v"(ahi[i]) : "v"(buf + a_off[i]), "n"(4 * V2_MT * 2));
#pragma unroll
        for (int jn = 0; jn < NJ; ++jn) {
            asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:%3" : "=&v"(blo[jn]), "=&v"(bhi[jn]) : "v"(buf + OFF_BHI + b_off[jn]), "n"(4 * NT * 2));
            asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:%3" : "=&v"(clo[jn]), "=&v"(chi[jn]) : "v"(buf + OFF_BLO + b_off[jn]), "n"(4 * NT * 2));
        }
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(alo[0]), "+v"(ahi[0]), "+v"(alo[1]), "+v"(ahi[1]), "+v"(alo[2]), "+v"(ahi[2]), "+v"(alo[3]), "+v"(ahi[3]),
                       "+v"(alo[4]), "+v"(ahi[4]), "+v"(alo[5]), "+v"(ahi[5]), "+v"(alo[6]), "+v"(ahi[6]), "+v"(alo[7]), "+v"(ahi[7]),
                       "+v"(xlo), "+v"(xhi), "+v"(ylo), "+v"(yhi));
#pragma unroll
        for (int jn = 0; jn < NJ; ++jn) asm volatile("" : "+v"(blo[jn]), "+v"(bhi[jn]), "+v"(clo[jn]), "+v"(chi[jn]));
        bf16x8 bh[NJ], bl[NJ];
#pragma unroll
        for (int jn = 0; jn < NJ; ++jn) { bh[jn] = frag(blo[jn], bhi[jn]); bl[jn] = frag(clo[jn], chi[jn]); }
        // A_lo fragments are requested now and waited for behind the two A_hi products
        s16x4 dlo[8], dhi[8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
            asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:%3" : "=&v"(dlo[i]), "=&v"(dhi[i]) : "v"(buf + OFF_ALO + a_off[i]), "n"(4 * V2_MT * 2));
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const bf16x8 af = frag(alo[i], ahi[i]);
#pragma unroll
            for (int jn = 0; jn < NJ; ++jn) acc[i][jn] = mfma_bf16_16x16x32(af, bh[jn], acc[i][jn]);
#pragma unroll
            for (int jn = 0; jn < NJ; ++jn) acc[i][jn] = mfma_bf16_16x16x32(af, bl[jn], acc[i][jn]);
            if constexpr (XP) {
                if (wn == 0) { accx[i] = mfma_bf16_16x16x32(af, frag(xlo, xhi), accx[i]); accx[i] = mfma_bf16_16x16x32(af, frag(ylo, yhi), accx[i]); }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(dlo[0]), "+v"(dhi[0]), "+v"(dlo[1]), "+v"(dhi[1]), "+v"(dlo[2]), "+v"(dhi[2]), "+v"(dlo[3]), "+v"(dhi[3]),
                       "+v"(dlo[4]), "+v"(dhi[4]), "+v"(dlo[5]), "+v"(dhi[5]), "+v"(dlo[6]), "+v"(dhi[6]), "+v"(dlo[7]), "+v"(dhi[7]));
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const bf16x8 af = frag(dlo[i], dhi[i]);
#pragma unroll
            for (int jn = 0; jn < NJ; ++jn) acc[i][jn] = mfma_bf16_16x16x32(af, bh[jn], acc[i][jn]);
            if constexpr (XP) {
                if (wn == 0) accx[i] = mfma_bf16_16x16x32(af, frag(xlo, xhi), accx[i]);
            }
        }
    };

#pragma unroll
    for (int k = 0; k < NBUF - 1; ++k) issue(k);
    for (int k = 0; k < nst; ++k) {
        // stage k has landed; NBUF - 2 younger stages stay in flight
        if constexpr (NBUF == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (XP && w == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NBUF - 2) * (PS + 2)) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NBUF - 2) * PS) : "memory");
        __builtin_amdgcn_s_barrier();                     // everyone's part of stage k is in LDS, everyone is done with the slot stage k + NBUF - 1 goes to
        issue(k + NBUF - 1);
        compute(k);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // no LDS-DMA may outlive the workgroup

#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int jn = 0; jn < NJ; ++jn) {
            const int ncol = (wn * NJ + jn) * 16 + i16, kb = ncol >> 7, n = ncol & 127;
            float* C = kb == 0 ? a.C : a.C2;
            float* dst = nsplit == 1 ? C : a.partial + ((size_t)split * NB + kb) * a.M * a.N;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm * 128 + i * 16 + g * 4 + r;
                dst[(size_t)m * a.N + n] = acc[i][jn][r];
            }
        }
    if constexpr (XP) {
        if (wn == 0 && i16 < a.Nx) {
            float* dst = nsplit == 1 ? a.Cx : a.partial + (size_t)nsplit * NB * a.M * a.N + (size_t)split * a.M * a.Nx;
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) dst[(size_t)(m0 + wm * 128 + i * 16 + g * 4 + r) * a.Nx + i16] = accx[i][r];
        }
    }
}

struct Geo { int tiles_m, tiles_n, nsplit, rows_per_split; };
bool v2_ok(const GemmTnArgsB& a) {
    if (a.bcast_T > 0 && !(a.M % V2_MT == 0 && a.N == 128)) return false;
    return a.M % V2_MT == 0 && a.N == 128 && a.lda == a.M && a.ldb == a.N && a.zeros != nullptr && !(a.drop_p > 0.f);
}
Geo geometry_v2(int R, int M) {
    Geo g;
    g.tiles_m = M / V2_MT; g.tiles_n = 1;
    int nsplit = 256 / g.tiles_m;                         // one 512-thread workgroup per CU
    const int max_split = ceil_div(R, 8 * V2_KC);
    if (nsplit > max_split) nsplit = max_split;
    if (nsplit < 1) nsplit = 1;
    g.rows_per_split = round_up(ceil_div(R, nsplit), V2_KC);
    g.nsplit = ceil_div(R, g.rows_per_split);
    return g;
}
Geo geometry(int R, int M, int N, int NB) {
    Geo g;
    g.tiles_m = ceil_div(M, MT); g.tiles_n = ceil_div(N, NT);
    const int tiles = g.tiles_m * g.tiles_n * NB;
    int nsplit = 512 / tiles;                         // ~2 workgroups per CU, each with a stage in flight
    const int max_split = ceil_div(R, 4 * KC);
    if (nsplit > max_split) nsplit = max_split;
    if (nsplit < 1) nsplit = 1;
    g.rows_per_split = round_up(ceil_div(R, nsplit), KC);
    g.nsplit = ceil_div(R, g.rows_per_split);
    return g;
}

}  // namespace

size_t gemm_tn_bf16_partial_elems(int R, int M, int N, int NB) {
    const Geo g = geometry(R, M, N, NB);
    size_t e = g.nsplit > 1 ? (size_t)g.nsplit * NB * M * N : 0;
    if (M % V2_MT == 0 && N == 128) {
        const Geo g2 = geometry_v2(R, M);
        const size_t e2 = g2.nsplit > 1 ? (size_t)g2.nsplit * (NB * M * N + (NB == 1 ? M * 16 : 0)) : 0;   // + the optional 16-column product
        if (e2 > e) e = e2;
    }
    return e;
}

int launch_gemm_tn_bf16(const GemmTnArgsB& a, hipStream_t s, ReduceJobs* defer) {
    SHM_REQUIRE(a.R > 0 && a.M > 0 && a.N > 0 && (a.NB == 1 || a.NB == 2), SHM_EINVAL, "gemm_tn_bf16: bad shape");
    SHM_REQUIRE((a.M & 7) == 0 && (a.N & 7) == 0 && (a.lda & 7) == 0 && (a.ldb & 7) == 0, SHM_EINVAL,
                "gemm_tn_bf16: M, N, lda, ldb must be multiples of 8 (got %d %d %d %d)", a.M, a.N, a.lda, a.ldb);
    SHM_REQUIRE(a.NB == 1 || (a.Bm2 && a.C2), SHM_EINVAL, "gemm_tn_bf16: second operand missing");
    SHM_REQUIRE(a.nseg <= 1 || (a.A_lo && a.Bm_lo && (a.NB == 1 || a.Bm2_lo) && (!a.Bx || a.Bx_lo) && !(a.drop_p > 0.f)), SHM_EINVAL,
                "gemm_tn_bf16: split-precision operands need every lo plane (and no in-kernel dropout)");
    const bool v2 = v2_ok(a);
    SHM_REQUIRE(a.bcast_T == 0 || v2, SHM_EINVAL, "gemm_tn_bf16: the per-window operand needs the H = 128 kernel");
    const Geo g = v2 ? geometry_v2(a.R, a.M) : geometry(a.R, a.M, a.N, a.NB);
    const bool xp = v2 && a.NB == 1 && a.Bx != nullptr;
    SHM_REQUIRE(!a.Bx || (xp && a.Cx && a.Nx >= 1 && a.Nx <= 16), SHM_EINVAL, "gemm_tn_bf16: the 16-column product needs the H = 128 kernel with NB == 1");
    SHM_REQUIRE(g.nsplit == 1 || (a.partial && a.partial_elems >= (size_t)g.nsplit * (a.NB * a.M * a.N + (xp ? a.M * a.Nx : 0))), SHM_EWORKSPACE,
                "gemm_tn_bf16: split-K scratch too small");
    static const bool sp_v2 = [] { const char* e = getenv("SHM_GEMM_SP_V2"); return e && e[0] == '1'; }();     // A/B knob: the round-2 three-pass form
    if (v2 && a.nseg > 1 && !sp_v2) {
        const int grid = round_up(g.nsplit, 8) * g.tiles_m;
        const size_t st = 2 * ((size_t)V2_KC * V2_MT * 2 + (size_t)V2_KC * 128 * a.NB * 2 + (xp ? V2_KC * 16 * 2 : 0));
        const size_t lds = (a.NB == 2 ? 2 : 3) * st;
        void (*kern)(GemmTnArgsB, int, int, int) = a.NB == 2 ? gemm_tn_sp_v3_kernel<2, false> : (xp ? gemm_tn_sp_v3_kernel<1, true> : gemm_tn_sp_v3_kernel<1, false>);
        static const void* raised[3] = {nullptr, nullptr, nullptr};
        bool done = false;
        for (const void* k : raised) done = done || k == (const void*)kern;
        if (!done) {
            SHM_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            for (auto& k : raised) if (!k) { k = (const void*)kern; break; }
        }
        SHM_LAUNCH(kern, dim3(grid), dim3(512), lds, s, a, g.tiles_m, g.rows_per_split, g.nsplit);
    } else if (v2) {
        const int grid = round_up(g.nsplit, 8) * g.tiles_m;
        const size_t lds = (size_t)V2_NBUF * (V2_KC * V2_MT * 2 + V2_KC * 128 * a.NB * 2 + (xp ? V2_KC * 16 * 2 : 0));
        if (a.NB == 2) {
            static bool attr2 = false;
            if (!attr2) { SHM_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_tn_bf16_v2_kernel<2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); attr2 = true; }
            SHM_LAUNCH((gemm_tn_bf16_v2_kernel<2, false>), dim3(grid), dim3(512), lds, s, a, g.tiles_m, g.rows_per_split, g.nsplit);
        } else if (xp) {
            static bool attrx = false;
            if (!attrx) { SHM_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_tn_bf16_v2_kernel<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); attrx = true; }
            SHM_LAUNCH((gemm_tn_bf16_v2_kernel<1, true>), dim3(grid), dim3(512), lds, s, a, g.tiles_m, g.rows_per_split, g.nsplit);
        } else {
            static bool attr1 = false;
            if (!attr1) { SHM_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_tn_bf16_v2_kernel<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); attr1 = true; }
            SHM_LAUNCH((gemm_tn_bf16_v2_kernel<1, false>), dim3(grid), dim3(512), lds, s, a, g.tiles_m, g.rows_per_split, g.nsplit);
        }
    } else {
        const int tiles = g.tiles_m * g.tiles_n * a.NB;
        const int grid = round_up(g.nsplit, 8) * tiles;
        SHM_LAUNCH(gemm_tn_bf16_kernel, dim3(grid), dim3(256), 0, s, a, g.tiles_m, g.tiles_n, g.rows_per_split, g.nsplit);
    }
    SHM_LAUNCH_CHECK();
    if (g.nsplit > 1) {
        // partial[split][kb][M*N]: reduce each product with a stride of NB*M*N between splits
        if (xp) {
            const float* px = a.partial + (size_t)g.nsplit * a.NB * a.M * a.N;
            int rc = defer ? reduce_jobs_add(*defer, px, g.nsplit, (size_t)a.M * a.Nx, (size_t)a.M * a.Nx, a.Cx, nullptr)
                           : launch_reduce_partials(px, g.nsplit, (size_t)a.M * a.Nx, a.Cx, nullptr, s);
            if (rc) return rc;
        }
        if (defer) {
            for (int kb = 0; kb < a.NB; ++kb) {
                int rc = reduce_jobs_add(*defer, a.partial + (size_t)kb * a.M * a.N, g.nsplit, (size_t)a.M * a.N, (size_t)a.NB * a.M * a.N,
                                         kb == 0 ? a.C : a.C2, nullptr);
                if (rc) return rc;
            }
            return SHM_OK;
        }
        if (a.NB == 2) return launch_reduce_partials_pair(a.partial, g.nsplit, (size_t)a.M * a.N, a.C, a.C2, s);
        for (int kb = 0; kb < a.NB; ++kb) {
            int rc = launch_reduce_partials_strided(a.partial + (size_t)kb * a.M * a.N, g.nsplit, (size_t)a.M * a.N, (size_t)a.NB * a.M * a.N,
                                                    kb == 0 ? a.C : a.C2, s);
            if (rc) return rc;
        }
    }
    return SHM_OK;
}

}  // namespace shm
