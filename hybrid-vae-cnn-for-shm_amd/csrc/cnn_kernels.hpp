// cnn_kernels.hpp — launcher interface of cnn_kernels.hip (used by cnn_api.hip).
#pragma once
#include "kernels.hpp"

namespace shm {

// layer 1: Conv2d(2,16,3,pad 1); layer 2: Conv2d(16,32,3,pad 1)   (reference cnn_model.py:17,23).  part != nullptr: also emit
// per-window per-channel (sum, sum of squares) for the train-mode BatchNorm statistics.
int cnn_conv_fwd(int layer, int B, int H, int W, const float* x, const float* w, const float* bias, float* y, float* part, hipStream_t s);
int cnn_bn_finalize(int C, int B, long long n_per_channel, int train, const float* part, const float* gamma, const float* beta,
                    float* run_mean, float* run_var, float* mean, float* rstd, float* scale, float* shift, hipStream_t s);
int cnn_bn_relu_pool(int B, int C, int H, int W, const float* y, const float* scale, const float* shift, float* out, unsigned char* idx, hipStream_t s);
int cnn_linear_fwd(int B, int F, int O, const float* in, const float* Wt, const float* bias, float* pre, float* out, int relu, float drop_p,
                   uint64_t seed, hipStream_t s);
int cnn_linear_bwd_dx(int B, int F, int O, const float* dy, const float* Wt, const float* pre, float* dx, float drop_p, uint64_t seed, hipStream_t s);
int cnn_ce_loss(int B, int NC, const float* logits, const long long* labels, float* dlogits, float* loss, hipStream_t s);
int cnn_softmax_argmax(int B, int NC, const float* logits, long long* cls, float* p1, hipStream_t s);
int cnn_build_input(int B, int TD, const float* z, const float* zhat, float* xin, hipStream_t s);
int cnn_bn_bwd(int B, int C, int H, int W, const float* dp, const unsigned char* idx, const float* y, const float* mean, const float* rstd,
               const float* gamma, float* part, float* k1, float* k2, float* dgamma, float* dbeta, float* dy, hipStream_t s);
int cnn_conv_bwd_data2(int B, int H, int W, const float* dy, const float* w, float* dx, hipStream_t s);
int cnn_conv_bwd_weight(int layer, int B, int H, int W, const float* dy, const float* x, float* part, hipStream_t s);

}  // namespace shm
