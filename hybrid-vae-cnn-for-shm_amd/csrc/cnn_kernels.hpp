// cnn_kernels.hpp — launcher interface of cnn_kernels.hip (used by cnn_api.hip).
#pragma once
#include "kernels.hpp"

namespace shm {

// layer 1: Conv2d(2,16,3,pad 1); layer 2: Conv2d(16,32,3,pad 1)   (reference cnn_model.py:17,23).  part != nullptr: also emit
// per-window per-channel (sum, sum of squares) for the train-mode BatchNorm statistics.
int cnn_conv_fwd(int layer, int B, int H, int W, const float* x, const float* w, const float* bias, float* y, float* part, hipStream_t s);
int cnn_bn_finalize(int C, int B, long long n_per_channel, int train, const float* part, const float* gamma, const float* beta,
                    float* run_mean, float* run_var, float* mean, float* rstd, float* scale, float* shift, hipStream_t s);
// xh != nullptr (with mean / rstd): also keep the normalised conv output at every cell's argmax (for cnn_bn_bwd_stats)
int cnn_bn_relu_pool(int B, int C, int H, int W, const float* y, const float* scale, const float* shift, float* out, unsigned char* idx, hipStream_t s,
                     const float* mean = nullptr, const float* rstd = nullptr, float* xh = nullptr);
int cnn_linear_fwd(int B, int F, int O, const float* in, const float* Wt, const float* bias, float* pre, float* out, int relu, float drop_p,
                   uint64_t seed, hipStream_t s);
int cnn_linear_bwd_dx(int B, int F, int O, const float* dy, const float* Wt, const float* pre, float* dx, float drop_p, uint64_t seed, hipStream_t s);
// output layer backward in one launch: dz1 (ReLU / dropout masks of the hidden layer re-applied), per-16-window partials of dW2 | db2
// queued on `jobs` (part: cnn_dense_out_bwd_partial_elems floats)
size_t cnn_dense_out_bwd_partial_elems(int B, int F1, int NC);
int cnn_dense_out_bwd(int B, int F1, int NC, const float* dl, const float* W2, const float* z1, const float* a1, float* dz1, float* part,
                      float* dW2, float* db2, float drop_p, uint64_t seed, hipStream_t s, ReduceJobs& jobs);
// hidden layer: dW[O][F] = dz^T in and db = column sums of dz, one direct fp32 MFMA product (no split-K partials)
bool cnn_dense_dw_direct_ok(int F, int O);
int cnn_dense_dw(int B, int F, int O, const float* dz, const float* in, float* dW, float* db, hipStream_t s);
int cnn_ce_loss(int B, int NC, const float* logits, const long long* labels, float* dlogits, float* loss, hipStream_t s);
int cnn_softmax_argmax(int B, int NC, const float* logits, long long* cls, float* p1, hipStream_t s);
int cnn_build_input(int B, int TD, const float* z, const float* zhat, float* xin, hipStream_t s);
// pool / ReLU / BatchNorm backward.  Pass A (cnn_bn_bwd_stats): the per-channel sums -> k1, k2, dgamma, dbeta.  Pass B, the gradient at every
// conv-output position, is either its own launch writing dy (cnn_bn_bwd_apply) or — `fuse` below — formed inside the conv gradient
// kernels from the operands in BnGrad, so dy never exists in memory.
struct BnGrad { const float* y; const float* dp; const unsigned char* idx; const float* mean; const float* rstd; const float* gamma; const float* k1; const float* k2; };
int cnn_bn_bwd_stats(int B, int C, int H, int W, const float* dp, const unsigned char* idx, const float* y, const float* mean, const float* rstd,
                     float* part, float* k1, float* k2, float* dgamma, float* dbeta, hipStream_t s, const float* xh = nullptr);
int cnn_bn_bwd_apply(int B, int C, int H, int W, const BnGrad& g, float* dy, hipStream_t s);
int cnn_conv_bwd_data2(int B, int H, int W, const float* dy, const BnGrad* fuse, const float* w, float* dx, hipStream_t s);
bool cnn_conv_bwd_weight_fusable(int layer, int H, int W, const float* y);
// dy_out (fused form of layer 2 only, see cnn_conv_bwd_weight_leaves_dy): the kernel also writes the dy it forms, for the data gradient
bool cnn_conv_bwd_weight_leaves_dy(int layer, int H, int W, const float* y);
int cnn_conv_bwd_weight(int layer, int B, int H, int W, const float* dy, const BnGrad* fuse, const float* x, float* part, hipStream_t s,
                        float* dy_out = nullptr);

}  // namespace shm
