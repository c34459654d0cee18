// kernels.hpp — internal launcher interface between api.hip and the kernel translation units.
#pragma once
#include "common.hpp"

namespace shm {

// Optional per-launch-group timing with HIP events on the launch stream (shm_profile_begin / shm_profile_end; bench.py's roofline
// objects).  A no-op unless the calling thread armed it; defined in api.hip.
struct KernelTimer {
    hipStream_t s; int idx = -1;
    KernelTimer(const char* name, hipStream_t stream);
    ~KernelTimer();
};

// ---- packed-weight geometry (fp32 MFMA 16x16x4 path) -------------------------------------------------
// Forward operand  W[4H][K]:  Kpad = round_up(K,16), KQ = Kpad/4 (k-range owned by one lane quarter), S4 = KQ/4.
//   packed[((ht*S4 + s4)*4 + g)*64 + lane][e] = W[g*H + ht*16 + (lane&15)][(lane>>4)*KQ + 4*s4 + e]   (0 beyond K)
// Backward operand W^T (out[m] = sum_k W[k][m] dpre[k], k over 4H): KQ = H, S4 = H/4, MT = ceil(M/16)
//   packed[(mt*S4 + s4)*64 + lane][e] = W[(lane>>4)*H + 4*s4 + e][mt*16 + (lane&15)]                  (0 beyond M)
inline int fwd_kpad(int K) { return round_up(K, 16); }
inline size_t fwd_pack_elems(int H, int K) { return (size_t)4 * H * fwd_kpad(K); }
inline size_t bwd_pack_elems(int H, int M) { return (size_t)round_up(M, 16) * 4 * H; }

struct LstmFwdArgs {
    const float* in;        // [B,T,In] (or [B,In] when in_const)
    const float* Wx;        // packed forward operand of W_ih
    const float* Wh;        // packed forward operand of W_hh
    const float* bias;      // [4H] = b_ih + b_hh
    float* h_seq;           // [B,T,H] or nullptr
    float* h_last;          // [B,H] or nullptr
    float* saved;           // fragment-order [tile][T][5][H/16][64][4] (i,f,g,o,c) or nullptr
    int B, T, In, in_const;
    float drop_p;           // dropout on the INPUT sequence (nn.LSTM inter-layer dropout), 0 = off
    uint64_t drop_seed;
    uint32_t drop_stack;
};
int launch_lstm_fwd_f32(int H, const LstmFwdArgs& a, hipStream_t s);

struct LstmBwdArgs {
    const float* saved;     // as written by the forward
    const float* dh_seq;    // [B,T,H] gradient arriving on h_t from the layer above, or nullptr
    const float* dy;        // [B,T,Dy] gradient on the output layer (top decoder layer), or nullptr
    const float* Wout;      // [Dy,H] output_layer.weight (with dy)
    const float* dh_last;   // [B,H] extra gradient on h_{T-1} (encoder top layer), or nullptr
    const float* WhT;       // packed backward operand of W_hh
    const float* WxT;       // packed backward operand of W_ih (need_dx) or nullptr
    float* dpre_seq;        // [B,T,4H] pre-activation gradients (for the deferred weight-gradient GEMMs)
    float* dpre_tsum;       // [B,4H]  sum over t of dpre (bias gradient / const-input layer)
    float* dx_seq;          // [B,T,In] gradient w.r.t. the layer input (need_dx), dropout mask applied
    int B, T, In, Dy, need_dx;
    float drop_p; uint64_t drop_seed; uint32_t drop_stack;   // mask of the layer INPUT (same stream as forward)
};
int launch_lstm_bwd_f32(int H, const LstmBwdArgs& a, hipStream_t s);

// ---- head (LayerNorm, fc_mu, fc_logvar, reparameterise, fc_latent_to_hidden + tanh) ---------------------
struct HeadBufs {           // all [B, *] fp32, workspace
    float *y, *xhat, *rstd, *mu, *lv, *z, *h0;
    float *dpre_l2h, *dmu, *dlv, *dyx, *dy, *dh_last, *dh0;
};
struct HeadParams {
    const float *ln_w, *ln_b, *mu_w, *mu_b, *lv_w, *lv_b, *l2h_w, *l2h_b;
};
struct HeadNoise { int on; uint64_t seed, offset; };      // counter RNG for the reparameterisation noise (common.hpp: eps_normal)
// eps: injected noise [B,Z] or nullptr (then rng, or none: z = mu); z_in: decode-only mode (h0 from a given z); eps_store: where the
// noise actually used is kept for the backward
int launch_head_fwd(int B, int H, int Z, int layer_norm, const float* h_last, const float* eps, const HeadParams& p,
                    const HeadBufs& hb, float* mu_out, float* lv_out, hipStream_t s, const HeadNoise& rng = HeadNoise{0, 0, 0},
                    const float* z_in = nullptr, float* eps_store = nullptr);
// dh0 = dpre_tsum[B,4H] . W_ih_dec0[4H,H]; then back through the head; d_mu/d_lv: upstream grads [B,Z]
int launch_head_bwd(int B, int H, int Z, int layer_norm, const float* dpre_tsum, const float* W_ih_dec0,
                    const float* eps, const float* d_mu, const float* d_lv, const HeadParams& p, const HeadBufs& hb,
                    hipStream_t s);

// all head parameter gradients (fc_latent_to_hidden, fc_mu, fc_logvar, LayerNorm) in one launch
int launch_head_grads(int B, int H, int Z, int layer_norm, const HeadBufs& hb, float* l2h_w, float* l2h_b, float* mu_w, float* mu_b,
                      float* lv_w, float* lv_b, float* ln_w, float* ln_b, hipStream_t s);

// ---- output layer + per-window score; loss --------------------------------------------------------------
// h_bf16: h_seq is bf16 in tile-blocked row order (row of (b,t) = ((b/16)*T + t)*16 + b%16); else fp32 natural [B,T,H]
int launch_out_fwd(int B, int T, int D, int H, const void* h_seq, int h_bf16, const float* Wout, const float* bout, const float* x,
                   float* recon, float* score, hipStream_t s, float* d_recon = nullptr, float* score_part = nullptr,
                   unsigned* tile_cnt = nullptr, const void* h_lo = nullptr);   // h_lo: split-precision path, h_seq / h_lo = the hi / lo planes   // bf16: score_part [4][ceil(B/4)*4] + tile_cnt [ceil(B/4)] (zeroed) let several workgroups share a tile   // d_recon (bf16 path, optional): 2 (recon - x) / (B T D)
int launch_vae_loss(int B, int T, int D, int Z, const float* x, const float* recon, const float* mu, const float* lv,
                    float kl_w, float* d_recon, float* d_mu, float* d_lv, float* score, float* losses, hipStream_t s);

// ---- C[M,N] (+)= sum_r A[r,M]^T B[r,N]  (weight gradients), fp32 MFMA, deterministic split-K ----------------
struct GemmTnArgs {
    const void* A; int lda;        // [R, M] fp32, or bf16 when a_bf16
    const void* Bm; int ldb;       // [R, N] fp32, or bf16 when b_bf16
    int a_bf16, b_bf16;
    float* C; int ldc;             // [M, N] row-major (or transposed when trans_out)
    int R, M, N;
    int shift_T;                   // >0: B row r is taken from row r-1, zero when r % shift_T == 0  (h_{t-1})
    int trans_out;                 // write C[n*ldc + m]
    float drop_p; uint64_t drop_seed; uint32_t drop_stack;   // dropout mask applied to B elements (idx = r*N + n)
    float* partial;                // split-K scratch
    size_t partial_elems;
};
// Deferred split-K / block-partial reductions: launchers that are handed a ReduceJobs append their reduction instead of
// launching it; launch_reduce_batch then reduces every appended job in ONE launch (each job keeps its own scratch until then).
struct ReduceJob {
    const float* partial; int nparts; unsigned n_out; size_t stride;   // out[i] = sum_p partial[p*stride + i], fixed order
    float* out0; float* out1;                                          // out1 optional (same values)
    int tr_M;                                                          // > 0: element i = n*tr_M + m is written to out0[m*(n_out/tr_M) + n]
    int vec;                                                           // set by launch_reduce_batch: 16-byte loads, 128 outputs per block
};
constexpr int MAX_REDUCE_JOBS = 12;
struct ReduceJobs { ReduceJob job[MAX_REDUCE_JOBS]; int first_block[MAX_REDUCE_JOBS + 1]; int n; };
int launch_window_gather(const float* series, long long n_rows, int D, const int* starts, int N, int T, const float* mean, const float* stdv,
                         float clip_z, float* out, hipStream_t s);
int reduce_jobs_add(ReduceJobs& j, const float* partial, int nparts, size_t n_out, size_t stride, float* out0, float* out1, int tr_M = 0);
int launch_reduce_batch(ReduceJobs& jobs, hipStream_t s);
int launch_gemm_tn_f32(const GemmTnArgs& a, hipStream_t s, ReduceJobs* defer = nullptr);
size_t gemm_tn_partial_elems(int R, int M, int N);
// out[n] = sum_r A[r*lda + n], written to out0 and (if non-null) out1; scratch: colsum_scratch_elems(R,N) floats
int launch_colsum(const float* A, int lda, int R, int N, float* out0, float* out1, float* scratch, hipStream_t s, ReduceJobs* defer = nullptr);
size_t colsum_scratch_elems(int R, int N);
// narrow product out[n][m] = sum_r Bn[r][n] A[r][m] (N <= 16 channels; trans_out writes out[m][n]); A fp32 or bf16
// bsum_out (optional, natural layout only): [N] column sums of Bn (the bias gradient that goes with the product)
// tile_T > 0: A's rows are tile-blocked (row r' = (tile*tile_T + t)*16 + j) while Bn is natural [B, tile_T, N]; rows of windows
// >= tile_B are padding and skipped.
int launch_narrow_tn(const void* A, int a_bf16, int lda, int M, const float* Bn, int N, int R, float* out, int trans_out,
                     float* bsum_out, float* scratch, hipStream_t s, int tile_T = 0, int tile_B = 0, ReduceJobs* defer = nullptr,
                     const void* A_lo = nullptr);        // A_lo: lo plane of a split-precision A (the product uses hi + lo)
size_t narrow_tn_scratch_elems(int M, int N);
// out[i] (and out1[i]) = sum_p partial[p*n + i], p < nparts (fixed order)
int launch_reduce_partials(const float* partial, int nparts, size_t n, float* out0, float* out1, hipStream_t s);
// same with an explicit element stride between consecutive partials
int launch_reduce_partials_strided(const float* partial, int nparts, size_t n, size_t stride, float* out0, hipStream_t s);
int launch_reduce_partials_pair(const float* partial, int nparts, size_t n, float* outA, float* outB, hipStream_t s);   // partial[split][2][n]

// ==== bf16 throughput path (fp32 accumulate / cell state / master weights) ===================================
// Windows per workgroup of the recurrent kernels ("quad tiles", lstm_fwd_bf16.hip).  Sequence buffers of this path are
// tile-blocked: row r' of (window b, step t) = ((b / SHM_WPT) * T + t) * SHM_WPT + b % SHM_WPT.
#define SHM_WPT 4
// Packed forward operand (bf16): Kpad = round_up(K,32), KQ = Kpad/4, KS = KQ/8 k-steps of 32
//   packed[((ht*KS + s)*4 + g)*64 + lane][e] = W[g*H + ht*16 + (lane&15)][(lane>>4)*KQ + 8*s + e], e = 0..7
// Packed backward operand (bf16, W^T): KS = H/8 k-steps over K = 4H (lane quarter q <-> gate q)
//   packed[(mt*KS + s)*64 + lane][e] = W[(lane>>4)*H + 8*s + e][mt*16 + (lane&15)]
inline size_t fwd_pack_elems_bf16(int H, int K) { return (size_t)4 * H * round_up(K, 32); }
inline size_t bwd_pack_elems_bf16(int H, int M) { return (size_t)round_up(M, 16) * 4 * H; }

struct LstmFwdArgsB {
    const void* in;         // fp32 [B,T,In] (encoder layer 0), fp32 [B,H] (in_const) or bf16 [B,T,H]
    const bf16_t* Wx; const bf16_t* Wh;
    const float* bias;
    bf16_t* h_seq;          // tile-blocked [ceil(B/WPT)*T*WPT, H] bf16
    bf16_t* xdrop;          // same layout: the dropout-masked input sequence (written when drop_p > 0), operand of the dW_ih GEMM
    float* h_last;          // [B,H] fp32 or nullptr
    bf16_t* sv_g;           // BPTT records [tile][T][H/16][64] x 16 B {i,f|g,o bf16, c_{t-1}, tanh(c_t) fp32} or nullptr
    bf16_t* h0_bf;          // constant-input layer only, optional: bf16 copy of the input [ceil(B/WPT)*WPT, H] (operand of its dW_ih product)
    bf16_t* x_tb;           // fp32-input layer only, optional: tile-blocked bf16 copy of x, [ceil(B/WPT)*T*WPT, 16] (cols >= In not written);
                            // the dW_ih product of this layer reads it as a third B operand of the dW_hh GEMM
    int B, T, In, in_const, in_f32;
    float drop_p; uint64_t drop_seed; uint32_t drop_stack;
};
int launch_lstm_fwd_bf16(int H, const LstmFwdArgsB& a, hipStream_t s);
// both layers of a 2-layer stack in one launch, layer 1 running a few steps behind layer 0 on its own workgroups
// sync: all-zero at launch; sync_other: the block the NEXT pair launch will use, zeroed by this one (the plan alternates two blocks,
// which replaces a memset in front of every launch); err: sticky
struct LstmFwdPairArgsB { LstmFwdArgsB l0, l1; unsigned* sync; unsigned* sync_other; unsigned* err; int role_lds_off; };
size_t lstm_fwd_pair_sync_bytes(int B);
int launch_lstm_fwd_pair_bf16(int H, const LstmFwdPairArgsB& p, hipStream_t s);

struct LstmBwdArgsB {
    const bf16_t* sv_g; const float* sv_c;
    const float* dh_seq; const float* dy; const float* Wout; const float* dh_last;
    const bf16_t* WhT; const bf16_t* WxT;
    bf16_t* dpre_seq;       // [B,T,4H] bf16
    float* dpre_tsum;       // [B,4H] fp32
    float* dx_seq;          // [B,T,In] fp32
    int B, T, In, Dy, need_dx;
    float drop_p; uint64_t drop_seed; uint32_t drop_stack;
};
int launch_lstm_bwd_bf16(int H, const LstmBwdArgsB& a, hipStream_t s);      // recurrent part: dpre_seq, dpre_tsum
int launch_dx_gemm_bf16(int H, const LstmBwdArgsB& a, hipStream_t s);       // time-parallel part: dx_seq = dpre_seq . W_ih
// both layers of a 2-layer stack in one launch: l1 = top layer (its dpre_seq is handed to l0 inside the launch), l0 forms
// dx = dpre^1 W_ih^1 itself; the two dpre_seq / dpre_tsum buffers must be distinct
struct LstmBwdPairArgsB { LstmBwdArgsB l1, l0; unsigned* sync; unsigned* sync_other; unsigned* err; int role_lds_off; };
int launch_lstm_bwd_pair_bf16(int H, const LstmBwdPairArgsB& p, hipStream_t s);

struct GemmTnArgsB {
    const void* A; int lda;        // [R, M] bf16 (dpre_seq)
    const void* Bm; int ldb;       // [R, N] bf16, product 0 (dropout mask optional)
    const void* Bm2;               // [R, N] bf16, product 1 (NB == 2)
    float* C; float* C2;           // [M, N] dense outputs of product 0 / 1
    int NB;                        // 1 or 2 products sharing A
    int R, M, N;
    int shift_T, shift_T2;         // per product: > 0 -> B row r is taken from row r - shift_rows, zero when (r / shift_rows) % shift_T == 0
    int shift_rows;
    int bcast_T;                   // > 0: product 0's B is one row per WINDOW ([ceil(B/WPT)*WPT, N]); tile-blocked row r reads row (r / (shift_rows*bcast_T)) * shift_rows + r % shift_rows
    float drop_p; uint64_t drop_seed; uint32_t drop_stack;     // dropout on product 0's B (element index r*N + n)
    float* partial; size_t partial_elems;
    const void* Bx; float* Cx; int Nx;   // optional third product sharing A (H = 128 kernel, NB == 1): Bx [R, 16] bf16, Cx [M, Nx] = A^T Bx[:, :Nx]
    const void* zeros;             // >= 16 zero bytes (the H = 128 kernel reads them for rows that must contribute nothing); may be NULL
    // split-precision operands (nseg = 3): A / Bm / Bm2 / Bx are the HI planes, these the LO planes; the kernel walks its rows three
    // times — (A, B), (A, B_lo), (A_lo, B) — into one accumulator, i.e. hi*hi + hi*lo + lo*hi in fp32
    int nseg;
    const void* A_lo; const void* Bm_lo; const void* Bm2_lo; const void* Bx_lo;
};
int launch_gemm_tn_bf16(const GemmTnArgsB& a, hipStream_t s, ReduceJobs* defer = nullptr);
size_t gemm_tn_bf16_partial_elems(int R, int M, int N, int NB);

// ==== split-precision path (SHM_PREC_SPLIT): every MFMA operand is a bf16 PAIR hi + lo (hi = bf16(x), lo = bf16(x - hi): 16
// significant bits), products hi*hi + hi*lo + lo*hi (+ lo*lo in the recurrences) accumulate in fp32; cell state, records,
// gradients and sums stay fp32.  Holds the reference's fp32 results to ~1e-6 at bf16 MFMA rates (lstm_sp.hip).
// Sequence buffers are two bf16 PLANES (hi, lo) in the tile-blocked row order of the bf16 path.
struct LstmFwdArgsS {
    const float* xproj;     // mode 0: W_ih x_t + b for every step, cell order [tile][t][H/16][64] x f32x4 {i,f,g,o};
                            // mode 1: the same once per window [tile][H/16][64] (constant input: decoder layer 0)
    const float* x;         // mode 2: fp32 [B,T,In], In <= 16 (encoder layer 0: projected in the kernel)
    const bf16_t* Wh_hi; const bf16_t* Wh_lo;     // forward fragment order (kernels.hpp: bf16 packing)
    const bf16_t* Wx_hi; const bf16_t* Wx_lo;     // mode 2
    const float* bias;                            // mode 2: b_ih + b_hh (modes 0/1: folded into xproj)
    bf16_t* h_hi; bf16_t* h_lo;                   // [ceil(B/WPT)*T*WPT, H] each
    float* h_last;                                // [B,H] fp32 or nullptr
    float* rec4; float* rec2;                     // BPTT records [tile][t][H/16][64] x {i,f,g,o} / {c_{t-1}, tanh c_t}, or nullptr
    bf16_t* x_hi; bf16_t* x_lo;                   // mode 2, optional: tile-blocked [rows,16] copies of x (dW_ih product)
    int B, T, In, mode;
    // time chunk [t0, t1) of the window (t1 == 0: all of it).  A chunk with t0 > 0 continues from c_state (cell order [tile][H/16][64],
    // written by the chunk before it) and from the h planes of step t0 - 1; a chunk with t1 < T leaves c_state for the next one.
    int t0, t1;
    float* c_state;
    // fused layer 1 of a pair launch (lstm_fwd_sp_fused): the input is the hidden sequence of the layer below (its h planes), projected
    // inside the recurrent workgroup with Wx_hi (kept in LDS) / Wx_lo (streamed from L2) and bias; dropout on that input as in XprojArgsS
    const bf16_t* in_hi; const bf16_t* in_lo;
    bf16_t* xd_hi; bf16_t* xd_lo;
    float drop_p; uint64_t drop_seed; uint32_t drop_stack;
};
int launch_lstm_fwd_sp(int H, const LstmFwdArgsS& a, hipStream_t s);
// both layers of a 2-layer stack in one launch: l0 in mode 1 / 2 hands its h planes to l1 inside the launch; l1 forms its own input
// projection (no xproj launch, no projection stream).  sync / sync_other / err as in LstmFwdPairArgsB.  Bit-identical to the
// one-launch-per-layer form.
struct LstmFwdPairArgsS { LstmFwdArgsS l0, l1; unsigned* sync; unsigned* sync_other; unsigned* err; int role_lds_off; };
int launch_lstm_fwd_pair_sp(int H, const LstmFwdPairArgsS& p, hipStream_t s);

struct LstmBwdArgsS {
    const float* rec4; const float* rec2;
    const float* dh_seq;    // fragment order (dx of the layer above, written by launch_dx_sp) or nullptr
    const float* dy; const float* Wout; const float* dh_last;
    const bf16_t* WhT_hi; const bf16_t* WhT_lo;   // backward fragment order
    bf16_t* dpre_hi; bf16_t* dpre_lo;             // [rows, 4H] planes
    float* dpre_tsum;       // [B,4H]
    int B, T, Dy;
    // time chunk: steps t1-1 .. t0 (t1 == 0: the whole window).  t1 < T: (dc, dh_rec) are taken from state ([tile][H/16][64] x f32x2, left by
    // the chunk above) and sum_t dpre is ADDED to dpre_tsum; t0 > 0: state is written for the chunk below.
    int t0, t1;
    float* state;
};
int launch_lstm_bwd_sp(int H, const LstmBwdArgsS& a, hipStream_t s);

// time-parallel halves: out = W_ih in + b (cell order, see LstmFwdArgsS::xproj) and dx = dpre W_ih (fragment order)
struct XprojArgsS {
    const bf16_t* in_hi; const bf16_t* in_lo;     // [rows, H] planes (rows = tile-blocked (tile, t, window); t_per_tile steps per tile)
    const bf16_t* W_hi; const bf16_t* W_lo;       // forward fragment order of W_ih [4H, H]
    const float* bias;                            // [4H]
    float* out;                                   // cell order
    bf16_t* xd_hi; bf16_t* xd_lo;                 // dropout: the masked + rescaled input planes (operand of the dW_ih GEMM)
    int rows;
    float drop_p; uint64_t drop_seed; uint32_t drop_stack;
    // time chunk of every window tile (cpt > 0): the launch covers n_chunk_tiles = tiles * cpt 16-row tiles; chunk-local tile k is the
    // global tile (k / cpt) * seq_tiles + tile_off + k % cpt (seq_tiles = T / 4 tiles per window tile, tile_off = t0 / 4, cpt = (t1 - t0) / 4)
    int cpt, seq_tiles, tile_off, n_chunk_tiles;
    int pair_bf16;                                // 1: W_hi / W_lo are a bf16 pair (pack kinds 0 / 5), 16-bit arithmetic (constant-input decoder layer 0);
                                                  // 0: W_hi fp16 + W_lo e4m3 residual (kinds 7 / 8), the arithmetic the fused layer-1 kernel shares
};
int launch_xproj_sp(int H, const XprojArgsS& a, hipStream_t s);
struct DxArgsS {
    const bf16_t* dpre_hi; const bf16_t* dpre_lo; // [rows, 4H]
    const bf16_t* WxT_hi; const bf16_t* WxT_lo;   // backward fragment order of W_ih
    float* dx;                                    // fragment order [rows/16][H/16][64] x f32x4
    int rows;
    float drop_p; uint64_t drop_seed; uint32_t drop_stack;
    int cpt, seq_tiles, tile_off, n_chunk_tiles;  // time chunk, as in XprojArgsS
};
int launch_dx_sp(int H, const DxArgsS& a, hipStream_t s);
// (r3) the top layer's backward recurrence and dx = dpre W_ih of the layer below in ONE launch: the dx workgroup of a tile follows the
// recurrence of that tile inside the launch (sc1 hand-off of the dpre planes).  top: the top layer's arguments (whole window, no dh_seq);
// dx: the arguments launch_dx_sp would get (rows / chunk fields unused).  T % 4 == 0.  Bit-identical to the two launches.
struct LstmBwdDxPairArgsS { LstmBwdArgsS top; DxArgsS dx; unsigned* sync; unsigned* sync_other; unsigned* err; int role_lds_off; };
int launch_lstm_bwd_dx_pair_sp(int H, const LstmBwdDxPairArgsS& p, hipStream_t s);
// fp32 [n_valid] -> bf16 planes hi, lo of n >= n_valid elements (the tail is zero-filled: padding windows of the last tile)
int launch_split_planes(const float* src, size_t n_valid, size_t n, bf16_t* hi, bf16_t* lo, hipStream_t s);

// all bf16 re-layouts + bias sums of one parameter update in a single launch
// kind 0: fwd operand (K = in width), 1: bwd operand (K = M), 2: bias sum, 3: copy K floats, 4: zero K dwords,
// 5 / 6: the LO parts (bf16(w - bf16(w))) of kinds 0 / 1
struct PackJob { const float* src; const float* src2; void* dst; int kind, H, K; };
struct PackJobs { PackJob job[48]; int n; };
int launch_pack_all(const PackJobs& jobs, hipStream_t s);
int launch_pack_fwd_bf16(const float* W, int H, int K, bf16_t* dst, hipStream_t s);
int launch_pack_bwd_bf16(const float* W, int H, int M, bf16_t* dst, hipStream_t s);

// ---- weight packing -----------------------------------------------------------------------------------------
int launch_pack_fwd_f32(const float* W, int H, int K, float* dst, hipStream_t s);
int launch_pack_bwd_f32(const float* W, int H, int M, float* dst, hipStream_t s);
int launch_bias_sum(const float* b_ih, const float* b_hh, int n, float* dst, hipStream_t s);

}  // namespace shm
