// cnn_api.hip — C-ABI entry points of the CNN classifier (include/shm_hip.h, second half).
// Launch sequence mirrors CNN.forward (reference 4DOF/Scripts/Models/cnn_model.py:45-51) and the train step of
// 4DOF/Scripts/05_train_cnn.py:270-276.
#include <string.h>

#include <new>
#include <vector>

#include "cnn_kernels.hpp"

using namespace shm;

struct shm_cnn_plan {
    shm_cnn_config cfg;
    int B, training;
    int H0, W0, H1, W1, H2, W2, F;
    // parameter element offsets (flat buffer)
    int64_t c1w, c1b, g1, b1, c2w, c2b, g2, b2, f1w, f1b, f2w, f2b, n_params;
    // workspace byte offsets
    size_t xh1, xh2;                 // training plans: normalised conv output at the pooled cells' argmax (BatchNorm-backward statistics)
    size_t y1, p1, i1, y2, p2, i2, z1, a1, logits, dlogits, da1, dz1, dflat, dy2, dp1, dy1;
    size_t part1, part2, mean1, rstd1, scale1, shift1, mean2, rstd2, scale2, shift2, k1, k2, wpart, wpart1, gemm_partial, gemm_partial2, colsum_scratch, colsum_scratch2;
    size_t gemm_partial_elems;
    size_t ws_bytes;
    bool fwd_saved, loss_done, xh_saved;
    uint64_t drop_seed;
    float drop_p_used;
};

namespace {

constexpr int C1 = 16, C2 = 32, FC1 = 128;

bool cfg_ok(const shm_cnn_config* c) {
    if (!c) { set_error("cnn config is NULL"); return false; }
    if (c->in_channels != 2) { set_error("cnn: in_channels %d unsupported (2: window + squared residual)", c->in_channels); return false; }
    if (c->num_classes < 1 || c->num_classes > 16) { set_error("cnn: num_classes %d unsupported (1..16)", c->num_classes); return false; }
    if (c->seq_len < 4 || c->num_features < 4) { set_error("cnn: window %dx%d too small for two 2x2 poolings", c->seq_len, c->num_features); return false; }
    if ((long long)c->seq_len * c->num_features > 2048) { set_error("cnn: window %dx%d exceeds the LDS-resident limit (T*D <= 2048)", c->seq_len, c->num_features); return false; }
    if (!(c->dropout >= 0.f && c->dropout < 1.f)) { set_error("cnn: dropout %f out of range", c->dropout); return false; }
    return true;
}

int64_t layout(const shm_cnn_config* c, std::vector<shm_tensor_desc>* out) {
    const int F = C2 * (c->seq_len / 4) * (c->num_features / 4);
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
        if (out) out->push_back(d);
    };
    add("conv1.0.weight", 4, C1, c->in_channels, 3, 3); add("conv1.0.bias", 1, C1, 0, 0, 0);
    add("conv1.1.weight", 1, C1, 0, 0, 0); add("conv1.1.bias", 1, C1, 0, 0, 0);
    add("conv2.0.weight", 4, C2, C1, 3, 3); add("conv2.0.bias", 1, C2, 0, 0, 0);
    add("conv2.1.weight", 1, C2, 0, 0, 0); add("conv2.1.bias", 1, C2, 0, 0, 0);
    add("fc1.0.weight", 2, FC1, F, 0, 0); add("fc1.0.bias", 1, FC1, 0, 0, 0);
    add("fc2.weight", 2, c->num_classes, FC1, 0, 0); add("fc2.bias", 1, c->num_classes, 0, 0, 0);
    return off;
}

template <typename T>
inline T* at(void* ws, size_t off) { return reinterpret_cast<T*>(reinterpret_cast<char*>(ws) + off); }

int check(const shm_cnn_plan* p, const void* ws) {
    SHM_REQUIRE(p != nullptr, SHM_EINVAL, "cnn plan is NULL");
    SHM_REQUIRE(ws != nullptr && (reinterpret_cast<uintptr_t>(ws) & 255) == 0, SHM_EWORKSPACE, "cnn workspace must be non-NULL and 256-byte aligned");
    return SHM_OK;
}

}  // namespace

extern "C" int64_t shm_cnn_param_count(const shm_cnn_config* cfg) { return cfg_ok(cfg) ? layout(cfg, nullptr) : SHM_EINVAL; }

extern "C" int shm_cnn_param_layout(const shm_cnn_config* cfg, shm_tensor_desc* out, int max) {
    if (!cfg_ok(cfg)) return SHM_EINVAL;
    std::vector<shm_tensor_desc> v;
    layout(cfg, &v);
    for (int i = 0; i < (int)v.size() && i < max && out; ++i) out[i] = v[i];
    return (int)v.size();
}

extern "C" int shm_cnn_plan_create(const shm_cnn_config* cfg, int32_t B, int32_t training, shm_cnn_plan** out) {
    if (!cfg_ok(cfg)) return SHM_EINVAL;
    SHM_REQUIRE(out && B >= 1, SHM_EINVAL, "cnn plan_create: bad arguments (B=%d)", B);
    shm_cnn_plan* p = new (std::nothrow) shm_cnn_plan();
    SHM_REQUIRE(p, SHM_EINVAL, "cnn plan_create: out of host memory");
    memset(static_cast<void*>(p), 0, sizeof(*p));
    p->cfg = *cfg; p->B = B; p->training = training ? 1 : 0;
    p->H0 = cfg->seq_len; p->W0 = cfg->num_features; p->H1 = p->H0 / 2; p->W1 = p->W0 / 2; p->H2 = p->H1 / 2; p->W2 = p->W1 / 2;
    p->F = C2 * p->H2 * p->W2;
    std::vector<shm_tensor_desc> v;
    p->n_params = layout(cfg, &v);
    int64_t* slots[12] = {&p->c1w, &p->c1b, &p->g1, &p->b1, &p->c2w, &p->c2b, &p->g2, &p->b2, &p->f1w, &p->f1b, &p->f2w, &p->f2b};
    for (int i = 0; i < 12; ++i) *slots[i] = v[i].offset;

    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    const size_t Bz = (size_t)B, NC = cfg->num_classes;
    const size_t n_y1 = Bz * C1 * p->H0 * p->W0, n_p1 = Bz * C1 * p->H1 * p->W1, n_y2 = Bz * C2 * p->H1 * p->W1, n_p2 = Bz * p->F;
    p->y1 = take(4 * n_y1); p->p1 = take(4 * n_p1); p->i1 = take(n_p1);
    p->y2 = take(4 * n_y2); p->p2 = take(4 * n_p2); p->i2 = take(n_p2);
    p->z1 = take(4 * Bz * FC1); p->a1 = take(4 * Bz * FC1); p->logits = take(4 * Bz * NC);
    p->part1 = take(4 * Bz * C1 * 2); p->part2 = take(4 * Bz * C2 * 2);
    p->mean1 = take(4 * C1); p->rstd1 = take(4 * C1); p->scale1 = take(4 * C1); p->shift1 = take(4 * C1);
    p->mean2 = take(4 * C2); p->rstd2 = take(4 * C2); p->scale2 = take(4 * C2); p->shift2 = take(4 * C2);
    if (training) {
        p->xh1 = take(4 * n_p1); p->xh2 = take(4 * n_p2);
        p->dlogits = take(4 * Bz * NC); p->da1 = take(4 * Bz * FC1); p->dz1 = take(4 * Bz * FC1);
        p->dflat = take(4 * n_p2); p->dy2 = take(4 * n_y2); p->dp1 = take(4 * n_p1); p->dy1 = take(4 * n_y1);
        p->k1 = take(4 * C2); p->k2 = take(4 * C2);
        p->wpart = take(4 * Bz * (C2 * C1 * 9 + C2));
        size_t gp = gemm_tn_partial_elems(B, FC1, p->F);
        const size_t gp2 = cnn_dense_out_bwd_partial_elems(B, FC1, (int)NC);
        if (gp2 > gp) gp = gp2;
        p->gemm_partial_elems = gp;
        p->gemm_partial = take(4 * (gp ? gp : 1));
        p->colsum_scratch = take(4 * colsum_scratch_elems(B, FC1));
        // second set: every reduction of the backward pass runs in ONE launch at its end, so no partial buffer may be reused before
        p->wpart1 = take(4 * Bz * (C1 * 2 * 9 + C1));
        p->gemm_partial2 = take(4 * (gp ? gp : 1));
        p->colsum_scratch2 = take(4 * colsum_scratch_elems(B, FC1));
    }
    p->ws_bytes = off;
    *out = p;
    return SHM_OK;
}
extern "C" void shm_cnn_plan_destroy(shm_cnn_plan* plan) { delete plan; }
extern "C" size_t shm_cnn_plan_workspace_bytes(const shm_cnn_plan* plan) { return plan ? plan->ws_bytes : 0; }

extern "C" int shm_cnn_forward(shm_cnn_plan* p, const float* params, float* running, const float* x, float* logits, int32_t train_mode,
                               uint64_t drop_seed, void* ws, void* stream) {
    int rc = check(p, ws);
    if (rc) return rc;
    SHM_REQUIRE(params && running && x, SHM_EINVAL, "cnn forward: params, running or x is NULL");
    hipStream_t s = (hipStream_t)stream;
    const int B = p->B, NC = p->cfg.num_classes;
    const int train = train_mode ? 1 : 0;
    p->fwd_saved = false; p->loss_done = false;
    float* y1 = at<float>(ws, p->y1); float* p1 = at<float>(ws, p->p1);
    float* y2 = at<float>(ws, p->y2); float* p2 = at<float>(ws, p->p2);
    const bool keep_xh = train && p->training && !getenv("SHM_CNN_NO_XHAT");      // the backward's statistics pass then never touches y1 / y2
    p->xh_saved = keep_xh;
    // block 1
    { KernelTimer kt("cnn.conv1_fwd", (hipStream_t)stream);
    if ((rc = cnn_conv_fwd(1, B, p->H0, p->W0, x, params + p->c1w, params + p->c1b, y1, train ? at<float>(ws, p->part1) : nullptr, s))) return rc;
    }
    { KernelTimer kt("cnn.bn1_fin", (hipStream_t)stream);
    if ((rc = cnn_bn_finalize(C1, B, (long long)B * p->H0 * p->W0, train, at<float>(ws, p->part1), params + p->g1, params + p->b1, running,
                              running + 16, at<float>(ws, p->mean1), at<float>(ws, p->rstd1), at<float>(ws, p->scale1), at<float>(ws, p->shift1), s))) return rc;
    }
    { KernelTimer kt("cnn.bn1_pool", (hipStream_t)stream);
    if ((rc = cnn_bn_relu_pool(B, C1, p->H0, p->W0, y1, at<float>(ws, p->scale1), at<float>(ws, p->shift1), p1, at<unsigned char>(ws, p->i1), s,
                               at<float>(ws, p->mean1), at<float>(ws, p->rstd1), keep_xh ? at<float>(ws, p->xh1) : nullptr))) return rc;
    }
    // block 2
    { KernelTimer kt("cnn.conv2_fwd", (hipStream_t)stream);
    if ((rc = cnn_conv_fwd(2, B, p->H1, p->W1, p1, params + p->c2w, params + p->c2b, y2, train ? at<float>(ws, p->part2) : nullptr, s))) return rc;
    }
    { KernelTimer kt("cnn.bn2_fin", (hipStream_t)stream);
    if ((rc = cnn_bn_finalize(C2, B, (long long)B * p->H1 * p->W1, train, at<float>(ws, p->part2), params + p->g2, params + p->b2, running + 32,
                              running + 64, at<float>(ws, p->mean2), at<float>(ws, p->rstd2), at<float>(ws, p->scale2), at<float>(ws, p->shift2), s))) return rc;
    }
    { KernelTimer kt("cnn.bn2_pool", (hipStream_t)stream);
    if ((rc = cnn_bn_relu_pool(B, C2, p->H1, p->W1, y2, at<float>(ws, p->scale2), at<float>(ws, p->shift2), p2, at<unsigned char>(ws, p->i2), s,
                               at<float>(ws, p->mean2), at<float>(ws, p->rstd2), keep_xh ? at<float>(ws, p->xh2) : nullptr))) return rc;
    }
    // dense head (flatten is a no-op on the NCHW pooled tensor)
    const float dp = train ? p->cfg.dropout : 0.f;
    { KernelTimer kt("cnn.fc1_fwd", (hipStream_t)stream);
    if ((rc = cnn_linear_fwd(B, p->F, FC1, p2, params + p->f1w, params + p->f1b, at<float>(ws, p->z1), at<float>(ws, p->a1), 1, dp, drop_seed, s))) return rc;
    }
    float* lg = at<float>(ws, p->logits);
    { KernelTimer kt("cnn.fc2_fwd", (hipStream_t)stream);
    if ((rc = cnn_linear_fwd(B, FC1, NC, at<float>(ws, p->a1), params + p->f2w, params + p->f2b, nullptr, lg, 0, 0.f, 0, s))) return rc;
    }
    if (logits) SHM_HIP_CHECK(hipMemcpyAsync(logits, lg, sizeof(float) * B * NC, hipMemcpyDeviceToDevice, s));
    p->drop_seed = drop_seed; p->drop_p_used = dp;
    p->fwd_saved = train && p->training;
    return SHM_OK;
}

extern "C" int shm_cnn_loss(shm_cnn_plan* p, const int64_t* labels, float* loss, void* ws, void* stream) {
    int rc = check(p, ws);
    if (rc) return rc;
    SHM_REQUIRE(p->training && p->fwd_saved, SHM_ESTATE, "cnn loss: needs a train-mode forward on a training plan first");
    SHM_REQUIRE(labels && loss, SHM_EINVAL, "cnn loss: labels or loss is NULL");
    { KernelTimer kt("cnn.ce_loss", (hipStream_t)stream);
    if ((rc = cnn_ce_loss(p->B, p->cfg.num_classes, at<float>(ws, p->logits), (const long long*)labels, at<float>(ws, p->dlogits), loss,
                          (hipStream_t)stream))) return rc;
    }
    p->loss_done = true;
    return SHM_OK;
}

extern "C" int shm_cnn_eval_loss(const float* logits, const int64_t* labels, int32_t B, int32_t num_classes, float* loss, void* stream) {
    SHM_REQUIRE(logits && labels && loss && B >= 1 && num_classes >= 1, SHM_EINVAL, "cnn eval loss: NULL argument or empty batch");
    return cnn_ce_loss(B, num_classes, logits, (const long long*)labels, nullptr, loss, (hipStream_t)stream);
}

extern "C" int shm_cnn_backward(shm_cnn_plan* p, const float* params, const float* x, const float* d_logits, float* grads, void* ws, void* stream) {
    int rc = check(p, ws);
    if (rc) return rc;
    SHM_REQUIRE(p->training && p->fwd_saved, SHM_ESTATE, "cnn backward: needs a train-mode forward on a training plan first");
    SHM_REQUIRE(params && x && grads, SHM_EINVAL, "cnn backward: params, x or grads is NULL");
    SHM_REQUIRE(d_logits || p->loss_done, SHM_ESTATE, "cnn backward: no upstream gradient and shm_cnn_loss was not called");
    hipStream_t s = (hipStream_t)stream;
    const int B = p->B, NC = p->cfg.num_classes, F = p->F;
    const float* dl = d_logits ? d_logits : at<float>(ws, p->dlogits);
    ReduceJobs rjobs;                                        // every split-K / per-window partial of this pass, reduced in one launch at the end
    rjobs.n = 0;
    auto gemm = [&](const float* A, int M, const float* Bm, int N, float* C, size_t part_off) -> int {
        GemmTnArgs g;
        memset(&g, 0, sizeof(g));
        g.A = A; g.lda = M; g.Bm = Bm; g.ldb = N; g.C = C; g.ldc = N; g.R = B; g.M = M; g.N = N;
        g.partial = at<float>(ws, part_off); g.partial_elems = p->gemm_partial_elems;
        return launch_gemm_tn_f32(g, s, &rjobs);
    };
    // fc2, one launch: dW = dl^T a1 and db = colsum(dl) as per-16-window partials, da1 = dl W2 with the dropout and ReLU masks of fc1
    // re-applied -> dz1
    { KernelTimer kt("cnn.fc2_bwd", (hipStream_t)stream);
    if ((rc = cnn_dense_out_bwd(B, FC1, NC, dl, params + p->f2w, at<float>(ws, p->z1), at<float>(ws, p->a1), at<float>(ws, p->dz1),
                                at<float>(ws, p->gemm_partial), grads + p->f2w, grads + p->f2b, p->drop_p_used, p->drop_seed, s, rjobs))) return rc;
    }
    // fc1: dW = dz1^T p2 and db = colsum(dz1) as one direct MFMA product (split-K GEMM + column sum when the layer does not tile)
    const float* dz1 = at<float>(ws, p->dz1);
    if (cnn_dense_dw_direct_ok(F, FC1)) {
        KernelTimer kt("cnn.fc1_dw", (hipStream_t)stream);
        if ((rc = cnn_dense_dw(B, F, FC1, dz1, at<float>(ws, p->p2), grads + p->f1w, grads + p->f1b, s))) return rc;
    } else {
        { KernelTimer kt("cnn.fc1_dw", (hipStream_t)stream);
        if ((rc = gemm(dz1, FC1, at<float>(ws, p->p2), F, grads + p->f1w, p->gemm_partial2))) return rc;
        }
        KernelTimer kt("cnn.fc1_db", (hipStream_t)stream);
        if ((rc = launch_colsum(dz1, FC1, B, FC1, grads + p->f1b, nullptr, at<float>(ws, p->colsum_scratch2), s, &rjobs))) return rc;
    }
    { KernelTimer kt("cnn.fc1_dx", (hipStream_t)stream);
    if ((rc = cnn_linear_bwd_dx(B, F, FC1, dz1, params + p->f1w, nullptr, at<float>(ws, p->dflat), 0.f, 0, s))) return rc;
    }
    // block 2.  The BatchNorm backward is its statistics pass only: the per-position gradient dy2 is formed inside the two conv gradient
    // kernels (never written) whenever the matrix-pipe weight gradient can run; otherwise the apply launch writes it as before.
    float* wpart = at<float>(ws, p->wpart);
    const BnGrad g2{at<float>(ws, p->y2), at<float>(ws, p->dflat), at<unsigned char>(ws, p->i2), at<float>(ws, p->mean2), at<float>(ws, p->rstd2),
                    params + p->g2, at<float>(ws, p->k1), at<float>(ws, p->k2)};
    const bool fuse2 = cnn_conv_bwd_weight_fusable(2, p->H1, p->W1, g2.y);
    // the K-split weight-gradient kernel forms every dy2 value exactly once and leaves it in the workspace: the data gradient reads it
    const bool dy2_left = fuse2 && cnn_conv_bwd_weight_leaves_dy(2, p->H1, p->W1, g2.y);
    { KernelTimer kt("cnn.bn2_bwd", (hipStream_t)stream);
    if ((rc = cnn_bn_bwd_stats(B, C2, p->H1, p->W1, g2.dp, g2.idx, g2.y, g2.mean, g2.rstd, at<float>(ws, p->part2), at<float>(ws, p->k1), at<float>(ws, p->k2),
                               grads + p->g2, grads + p->b2, s, p->xh_saved ? at<float>(ws, p->xh2) : nullptr))) return rc;
    if (!fuse2 && (rc = cnn_bn_bwd_apply(B, C2, p->H1, p->W1, g2, at<float>(ws, p->dy2), s))) return rc;
    }
    { KernelTimer kt("cnn.conv2_dw", (hipStream_t)stream);
    if ((rc = cnn_conv_bwd_weight(2, B, p->H1, p->W1, at<float>(ws, p->dy2), fuse2 ? &g2 : nullptr, at<float>(ws, p->p1), wpart, s,
                                  dy2_left ? at<float>(ws, p->dy2) : nullptr))) return rc;
    }
    {
        // partial rows are [C2*C1*9 weights | C2 biases]: the weight block and the bias block of every partial row, each with the row stride
        const size_t nw = (size_t)C2 * C1 * 9, n = nw + C2;
        if ((rc = reduce_jobs_add(rjobs, wpart, B, nw, n, grads + p->c2w, nullptr))) return rc;
        if ((rc = reduce_jobs_add(rjobs, wpart + nw, B, (size_t)C2, n, grads + p->c2b, nullptr))) return rc;
    }
    { KernelTimer kt("cnn.conv2_dx", (hipStream_t)stream);
    if ((rc = cnn_conv_bwd_data2(B, p->H1, p->W1, at<float>(ws, p->dy2), (fuse2 && !dy2_left) ? &g2 : nullptr, params + p->c2w, at<float>(ws, p->dp1), s))) return rc;
    }
    // block 1 (k1 / k2 of block 2 are consumed: the stream orders the reuse)
    const BnGrad g1{at<float>(ws, p->y1), at<float>(ws, p->dp1), at<unsigned char>(ws, p->i1), at<float>(ws, p->mean1), at<float>(ws, p->rstd1),
                    params + p->g1, at<float>(ws, p->k1), at<float>(ws, p->k2)};
    const bool fuse1 = cnn_conv_bwd_weight_fusable(1, p->H0, p->W0, g1.y);
    { KernelTimer kt("cnn.bn1_bwd", (hipStream_t)stream);
    if ((rc = cnn_bn_bwd_stats(B, C1, p->H0, p->W0, g1.dp, g1.idx, g1.y, g1.mean, g1.rstd, at<float>(ws, p->part1), at<float>(ws, p->k1), at<float>(ws, p->k2),
                               grads + p->g1, grads + p->b1, s, p->xh_saved ? at<float>(ws, p->xh1) : nullptr))) return rc;
    if (!fuse1 && (rc = cnn_bn_bwd_apply(B, C1, p->H0, p->W0, g1, at<float>(ws, p->dy1), s))) return rc;
    }
    float* wpart1 = at<float>(ws, p->wpart1);
    { KernelTimer kt("cnn.conv1_dw", (hipStream_t)stream);
    if ((rc = cnn_conv_bwd_weight(1, B, p->H0, p->W0, at<float>(ws, p->dy1), fuse1 ? &g1 : nullptr, x, wpart1, s))) return rc;
    }
    {
        const size_t nw = (size_t)C1 * 2 * 9, n = nw + C1;
        if ((rc = reduce_jobs_add(rjobs, wpart1, B, nw, n, grads + p->c1w, nullptr))) return rc;
        if ((rc = reduce_jobs_add(rjobs, wpart1 + nw, B, (size_t)C1, n, grads + p->c1b, nullptr))) return rc;
    }
    KernelTimer kt("cnn.reduce", s);
    return launch_reduce_batch(rjobs, s);
}

extern "C" int shm_cnn_build_input(int32_t B, int32_t T, int32_t D, const float* z, const float* zhat, float* xin, void* stream) {
    SHM_REQUIRE(B >= 1 && T >= 1 && D >= 1 && z && zhat && xin, SHM_EINVAL, "cnn build_input: bad arguments");
    return cnn_build_input(B, T * D, z, zhat, xin, (hipStream_t)stream);
}

extern "C" int shm_cnn_predict(int32_t B, int32_t num_classes, const float* logits, int64_t* cls, float* p1, void* stream) {
    SHM_REQUIRE(B >= 1 && num_classes >= 1 && logits, SHM_EINVAL, "cnn predict: bad arguments");
    return cnn_softmax_argmax(B, num_classes, logits, (long long*)cls, p1, (hipStream_t)stream);
}
