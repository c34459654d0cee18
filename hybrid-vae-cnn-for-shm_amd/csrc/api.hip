// api.hip — C-ABI entry points (include/shm_hip.h): parameter layout, plan/workspace, forward, loss, backward.
//
// The launch sequence mirrors TemporalVAE.forward (reference 4DOF/Scripts/Models/temporal_vae.py:72-77) and the
// train step of 4DOF/Scripts/03_train_vae.py:262-271; each stage is one persistent kernel per LSTM layer.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <functional>
#include <map>
#include <new>
#include <string>
#include <vector>

#include "kernels.hpp"

#ifndef SHM_SP_CHUNKS_DEFAULT
#define SHM_SP_CHUNKS_DEFAULT 1
#endif

namespace shm {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- optional per-kernel timing with HIP events on the launch stream (bench.py's roofline line) -------------
struct ProfEntry { char name[32]; hipEvent_t e0, e1; };
static thread_local bool g_prof_on = false;
static thread_local std::vector<ProfEntry>* g_prof = nullptr;
KernelTimer::KernelTimer(const char* name, hipStream_t stream) : s(stream) {
    if (!g_prof_on || !g_prof) return;
    ProfEntry e;
    snprintf(e.name, sizeof(e.name), "%s", name);
    if (hipEventCreate(&e.e0) != hipSuccess || hipEventCreate(&e.e1) != hipSuccess) return;
    (void)hipEventRecord(e.e0, s);
    g_prof->push_back(e);
    idx = (int)g_prof->size() - 1;
}
KernelTimer::~KernelTimer() { if (idx >= 0) (void)hipEventRecord((*g_prof)[idx].e1, s); }

struct LayerInst {
    int In = 0;
    bool in_const = false;
    int64_t w_ih = 0, w_hh = 0, b_ih = 0, b_hh = 0;      // element offsets into the flat parameter buffer
    size_t Wx = 0, Wh = 0, bsum = 0, WhT = 0, WxT = 0;   // byte offsets into the workspace (packed operands)
    size_t h_seq = 0, saved = 0, saved_c = 0;            // bf16 path: saved = BPTT records
    size_t xdrop = 0;                                    // bf16 path, layers >= 1: dropout-masked input sequence
    // split-precision path: the LO planes / LO fragment arrays that pair with the fields above, and the hoisted input projection
    size_t Wx_lo = 0, Wh_lo = 0, WhT_lo = 0, WxT_lo = 0, h_lo = 0, xdrop_lo = 0, xproj = 0;
    size_t c_state = 0, b_state = 0;                     // split path, time-chunked launches: cell state / (dc, dh_rec) carried between chunks
};

}  // namespace shm

struct shm_vae_plan {
    shm_vae_config cfg;
    int B, T, training;
    shm::LayerInst enc[2], dec[2];
    int64_t ln_w, ln_b, mu_w, mu_b, lv_w, lv_b, l2h_w, l2h_b, out_w, out_b, n_params;
    // workspace byte offsets
    size_t h_last, eps, y, xhat, rstd, mu, lv, z, h0, dpre_l2h, dmu, dlv, dyx, dy, dh_last, dh0;
    size_t recon, score, d_recon, d_mu, d_lv, dpre_seq, dx_seq, dpre_tsum, gemm_partial, colsum_scratch;
    size_t gemm_partial_b, gemm_partial_c;   // bf16 pair path: split-K scratch of the layer-0 GEMM / the fp32 dec0 dW_ih GEMM (all reduced in one batch)
    size_t dpre_seq2, dpre_tsum2;   // bf16 path, 2 layers: the top layer's dpre / sum_t dpre (both layers' live at once in a pair launch)
    size_t gemm_partial_elems;
    size_t narrow_scratch;  // block partials of narrow_tn (enc0 dW_ih)
    size_t narrow_scratch2; // block partials of narrow_tn (dW_out): both may be pending in one reduce batch
    size_t score_part;      // bf16 path: out_fwd's per-part squared-error sums [4][tiles*4]
    size_t pair_sync;       // bf16 path, 2 layers: ticket / progress words of the layer-pipelined launches (zeroed per launch)
    size_t h0_bf;           // bf16 path: bf16 copy of the decoder input h0 [qt*WPT, H], written by decoder layer 0's forward
    size_t x_tb;            // bf16 path: tile-blocked bf16 copy of x [Rq, 16], written by encoder layer 0's forward, read by its dW_ih product
    size_t misc;            // bf16 path: [64 B sticky error word of the pair launches][256 zero bytes], zeroed by pack_weights
    size_t dpre_lo, x_tb_lo, h0_lo;   // split-precision path: LO planes of dpre_seq / x_tb / h0_bf
    size_t params_copy;     // flat fp32 copy of the parameters (head / output layer read them unpacked)
    size_t ws_bytes;
    // call-order state
    bool packed, fwd_saved, loss_done, drop_active;
    bool err_armed;         // the sticky hand-off error word has been cleared once (first pack_weights on this plan / explicit reset)
    // split path: the deferred weight-gradient products of a layer run on a side stream of the plan, beside the next layer's recurrence
    // (a recurrent launch holds one workgroup on 128 of the 256 CUs).  Created on the first backward; two dpre plane sets alternate.
    hipStream_t side, side2;
    hipEvent_t ev_fork, ev_join, ev_join2, ev_rec[4], ev_chunk[16];
    bool side_ready;
    size_t dpre_set[4], dpre_set_lo[4], dpre_set_tsum[4];   // split path: dpre hi / lo planes and sum_t dpre per (stack, layer)
    int sync_next;          // which of the two pair-launch sync blocks is all-zero and goes to the next pair launch
    uint64_t drop_seed;
    float* recon_ptr;       // where the last forward left recon (caller buffer or workspace)
};

namespace shm {

static bool config_ok(const shm_vae_config* c) {
    if (!c) { set_error("config is NULL"); return false; }
    if (c->hidden_dim != 32 && c->hidden_dim != 64 && c->hidden_dim != 128) {
        set_error("hidden_dim %d unsupported (32, 64, 128)", c->hidden_dim); return false; }
    if (c->num_layers < 1 || c->num_layers > 2) { set_error("num_layers %d unsupported (1, 2)", c->num_layers); return false; }
    if (c->latent_dim < 1 || c->latent_dim > 32) { set_error("latent_dim %d unsupported (1..32)", c->latent_dim); return false; }
    if (c->input_dim < 1 || c->input_dim > 16) { set_error("input_dim %d unsupported (1..16)", c->input_dim); return false; }
    if (c->precision != SHM_PREC_F32 && c->precision != SHM_PREC_BF16 && c->precision != SHM_PREC_SPLIT) {
        set_error("precision %d unknown (0 = fp32, 1 = bf16, 2 = split bf16x2)", c->precision); return false; }
    if (!(c->dropout >= 0.f && c->dropout < 1.f)) { set_error("dropout %f out of range", c->dropout); return false; }
    return true;
}

// state_dict order of TemporalVAE (temporal_vae.py:28-49)
static int build_layout(const shm_vae_config* c, std::vector<shm_tensor_desc>& out) {
    const int D = c->input_dim, H = c->hidden_dim, Z = c->latent_dim, L = c->num_layers;
    int64_t off = 0;
    auto add = [&](const char* name, int ndim, int s0, int s1) {
        shm_tensor_desc d;
        memset(&d, 0, sizeof(d));
        snprintf(d.name, sizeof(d.name), "%s", name);
        d.offset = off; d.ndim = ndim; d.shape[0] = s0; d.shape[1] = s1;
        off += (int64_t)s0 * (ndim > 1 ? s1 : 1);
        // keep every tensor 16-byte aligned inside the flat buffer (vector loads in the kernels)
        off = (off + 3) & ~(int64_t)3;
        out.push_back(d);
    };
    char nm[48];
    auto lstm = [&](const char* pre, int in0) {
        for (int l = 0; l < L; ++l) {
            const int in = l == 0 ? in0 : H;
            snprintf(nm, sizeof(nm), "%s.weight_ih_l%d", pre, l); add(nm, 2, 4 * H, in);
            snprintf(nm, sizeof(nm), "%s.weight_hh_l%d", pre, l); add(nm, 2, 4 * H, H);
            snprintf(nm, sizeof(nm), "%s.bias_ih_l%d", pre, l); add(nm, 1, 4 * H, 0);
            snprintf(nm, sizeof(nm), "%s.bias_hh_l%d", pre, l); add(nm, 1, 4 * H, 0);
        }
    };
    lstm("encoder_lstm", D);
    if (c->layer_norm) { add("layer_norm.weight", 1, H, 0); add("layer_norm.bias", 1, H, 0); }
    add("fc_mu.weight", 2, Z, H); add("fc_mu.bias", 1, Z, 0);
    add("fc_logvar.weight", 2, Z, H); add("fc_logvar.bias", 1, Z, 0);
    add("fc_latent_to_hidden.weight", 2, H, Z); add("fc_latent_to_hidden.bias", 1, H, 0);
    lstm("decoder_lstm", H);
    add("output_layer.weight", 2, D, H); add("output_layer.bias", 1, D, 0);
    return (int)off;
}

// Layer-pipelined launches can be switched off (one launch per layer) for A/B measurements and for the bit-identity test:
// SHM_PAIR_LAUNCH=0 in the environment, or shm_set_pair_launch(0).  Results are bit-identical either way.
// misc region of the bf16 workspace: [0,64) sticky error word, [64,320) a line of zeros (GEMM operand for invalid rows),
// [320, ...) one arrival counter per 4-window tile for out_fwd's shared tiles
static size_t misc_bytes(int B) { return 320 + (((size_t)ceil_div(B, SHM_WPT) * 4 + 63) & ~(size_t)63); }
static int g_pair_launch = -1;
static bool pair_launch_enabled() {
    if (g_pair_launch < 0) { const char* e = getenv("SHM_PAIR_LAUNCH"); g_pair_launch = (e && e[0] == '0') ? 0 : 1; }
    return g_pair_launch != 0;
}

template <typename T>
static inline T* at(void* ws, size_t off) { return reinterpret_cast<T*>(reinterpret_cast<char*>(ws) + off); }

static HeadBufs head_bufs(const shm_vae_plan* p, void* ws) {
    HeadBufs hb;
    hb.y = at<float>(ws, p->y); hb.xhat = at<float>(ws, p->xhat); hb.rstd = at<float>(ws, p->rstd);
    hb.mu = at<float>(ws, p->mu); hb.lv = at<float>(ws, p->lv); hb.z = at<float>(ws, p->z); hb.h0 = at<float>(ws, p->h0);
    hb.dpre_l2h = at<float>(ws, p->dpre_l2h); hb.dmu = at<float>(ws, p->dmu); hb.dlv = at<float>(ws, p->dlv);
    hb.dyx = at<float>(ws, p->dyx); hb.dy = at<float>(ws, p->dy); hb.dh_last = at<float>(ws, p->dh_last);
    hb.dh0 = at<float>(ws, p->dh0);
    return hb;
}
static HeadParams head_params(const shm_vae_plan* p, const float* params) {
    HeadParams hp;
    hp.ln_w = p->cfg.layer_norm ? params + p->ln_w : nullptr; hp.ln_b = p->cfg.layer_norm ? params + p->ln_b : nullptr;
    hp.mu_w = params + p->mu_w; hp.mu_b = params + p->mu_b; hp.lv_w = params + p->lv_w; hp.lv_b = params + p->lv_b;
    hp.l2h_w = params + p->l2h_w; hp.l2h_b = params + p->l2h_b;
    return hp;
}

}  // namespace shm

using namespace shm;

extern "C" const char* shm_last_error(void) { return g_err; }
extern "C" int shm_version(void) { return 1; }

extern "C" int64_t shm_vae_param_count(const shm_vae_config* cfg) {
    if (!config_ok(cfg)) return SHM_EINVAL;
    std::vector<shm_tensor_desc> v;
    return build_layout(cfg, v);
}

extern "C" int shm_vae_param_layout(const shm_vae_config* cfg, shm_tensor_desc* out, int max) {
    if (!config_ok(cfg)) return SHM_EINVAL;
    std::vector<shm_tensor_desc> v;
    build_layout(cfg, v);
    for (int i = 0; i < (int)v.size() && i < max && out; ++i) out[i] = v[i];
    return (int)v.size();
}

extern "C" int shm_vae_plan_create(const shm_vae_config* cfg, int32_t B, int32_t T, int32_t training, shm_vae_plan** out) {
    if (!config_ok(cfg)) return SHM_EINVAL;
    SHM_REQUIRE(out != nullptr, SHM_EINVAL, "plan_create: out is NULL");
    SHM_REQUIRE(B >= 1 && T >= 1, SHM_EINVAL, "plan_create: empty batch or window (B=%d, T=%d)", B, T);
    SHM_REQUIRE((int64_t)B * T * 4 * cfg->hidden_dim < (int64_t)1 << 31, SHM_EINVAL, "plan_create: B*T*4H exceeds 2^31 (B=%d T=%d)", B, T);
    shm_vae_plan* p = new (std::nothrow) shm_vae_plan();
    SHM_REQUIRE(p != nullptr, SHM_EINVAL, "plan_create: out of host memory");
    memset(static_cast<void*>(p), 0, sizeof(*p));
    p->cfg = *cfg; p->B = B; p->T = T; p->training = training ? 1 : 0;
    const int D = cfg->input_dim, H = cfg->hidden_dim, Z = cfg->latent_dim, L = cfg->num_layers;

    std::vector<shm_tensor_desc> lay;
    p->n_params = build_layout(cfg, lay);
    auto find = [&](const char* name) -> int64_t {
        for (auto& d : lay) if (strcmp(d.name, name) == 0) return d.offset;
        return -1;
    };
    char nm[48];
    for (int l = 0; l < L; ++l) {
        for (int s = 0; s < 2; ++s) {
            LayerInst& li = s == 0 ? p->enc[l] : p->dec[l];
            const char* pre = s == 0 ? "encoder_lstm" : "decoder_lstm";
            li.In = (l == 0 && s == 0) ? D : H;
            li.in_const = (s == 1 && l == 0);
            snprintf(nm, sizeof(nm), "%s.weight_ih_l%d", pre, l); li.w_ih = find(nm);
            snprintf(nm, sizeof(nm), "%s.weight_hh_l%d", pre, l); li.w_hh = find(nm);
            snprintf(nm, sizeof(nm), "%s.bias_ih_l%d", pre, l); li.b_ih = find(nm);
            snprintf(nm, sizeof(nm), "%s.bias_hh_l%d", pre, l); li.b_hh = find(nm);
        }
    }
    if (cfg->layer_norm) { p->ln_w = find("layer_norm.weight"); p->ln_b = find("layer_norm.bias"); }
    p->mu_w = find("fc_mu.weight"); p->mu_b = find("fc_mu.bias");
    p->lv_w = find("fc_logvar.weight"); p->lv_b = find("fc_logvar.bias");
    p->l2h_w = find("fc_latent_to_hidden.weight"); p->l2h_b = find("fc_latent_to_hidden.bias");
    p->out_w = find("output_layer.weight"); p->out_b = find("output_layer.bias");

    size_t off = 0;
    auto take_bytes = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    auto take = [&](size_t elems) { return take_bytes(elems * sizeof(float)); };
    const bool sp = cfg->precision == SHM_PREC_SPLIT;             // split path: the bf16 path's tile-blocked layouts, every operand as hi + lo planes
    const bool bf = cfg->precision == SHM_PREC_BF16 || sp;
    const size_t tiles = (size_t)ceil_div(B, 16);                 // fp32 path: 16-window tiles
    const size_t qt = (size_t)ceil_div(B, SHM_WPT);               // bf16 path: SHM_WPT-window tiles
    const size_t Rq = (qt * SHM_WPT * T + 15) / 16 * 16;          // bf16 path: tile-blocked rows, padded to whole MFMA fragments
    const size_t BT = (size_t)B * T;
    for (int l = 0; l < L; ++l) {
        for (int s = 0; s < 2; ++s) {
            LayerInst& li = s == 0 ? p->enc[l] : p->dec[l];
            li.bsum = take(4 * H);
            if (sp) {
                li.Wx = take_bytes(2 * fwd_pack_elems_bf16(H, li.In)); li.Wx_lo = take_bytes(2 * fwd_pack_elems_bf16(H, li.In));
                li.Wh = take_bytes(2 * fwd_pack_elems_bf16(H, H)); li.Wh_lo = take_bytes(2 * fwd_pack_elems_bf16(H, H));
                li.h_seq = take_bytes(2 * Rq * H); li.h_lo = take_bytes(2 * Rq * H);
                if (l > 0) li.xproj = take(Rq * 4 * H);                   // W_ih x_t + b of every step, cell order (whole tiles)
                else if (li.in_const) li.xproj = take(qt * SHM_WPT * 4 * H);   // once per window
                if (l > 0 && cfg->dropout > 0.f) { li.xdrop = take_bytes(2 * Rq * H); li.xdrop_lo = take_bytes(2 * Rq * H); }
                li.c_state = take(qt * SHM_WPT * H);
                if (training) li.b_state = take(2 * qt * SHM_WPT * H);
                if (training) {
                    li.WhT = take_bytes(2 * bwd_pack_elems_bf16(H, H)); li.WhT_lo = take_bytes(2 * bwd_pack_elems_bf16(H, H));
                    if (l > 0) { li.WxT = take_bytes(2 * bwd_pack_elems_bf16(H, li.In)); li.WxT_lo = take_bytes(2 * bwd_pack_elems_bf16(H, li.In)); }
                    li.saved = take_bytes(16 * qt * T * 4 * H);         // {i,f,g,o} fp32 per cell and step
                    li.saved_c = take_bytes(8 * qt * T * 4 * H);        // {c_{t-1}, tanh c_t}
                }
            } else if (bf) {
                li.Wx = take_bytes(2 * fwd_pack_elems_bf16(H, li.In));
                li.Wh = take_bytes(2 * fwd_pack_elems_bf16(H, H));
                li.h_seq = take_bytes(2 * Rq * H);                        // whole tiles: unconditional stores in the kernel
                if (l > 0 && cfg->dropout > 0.f) li.xdrop = take_bytes(2 * Rq * H);   // train-mode forward writes its masked input here
                if (training) {
                    li.WhT = take_bytes(2 * bwd_pack_elems_bf16(H, H));
                    if (l > 0) li.WxT = take_bytes(2 * bwd_pack_elems_bf16(H, li.In));
                    li.saved = take_bytes(12 * qt * T * 4 * H);         // one 12-byte record per lane (4H lanes) and step
                }
            } else {
                li.Wx = take(fwd_pack_elems(H, li.In));
                li.Wh = take(fwd_pack_elems(H, H));
                li.h_seq = take(BT * H);
                if (training) {
                    li.WhT = take(bwd_pack_elems(H, H));
                    if (l > 0) li.WxT = take(bwd_pack_elems(H, li.In));
                    li.saved = take(tiles * T * 5 * 16 * H);
                }
            }
        }
    }
    if (bf && L == 2) p->pair_sync = take_bytes(2 * lstm_fwd_pair_sync_bytes(B));      // two blocks, used alternately (bf16 and split pair launches)
    if (bf) p->misc = take_bytes(misc_bytes(B));                     // sticky error word | 256 B of zeros | out_fwd's per-tile arrival counters
    if (bf) p->score_part = take((size_t)4 * ceil_div(B, SHM_WPT) * SHM_WPT);
    if (bf) p->x_tb = take_bytes(2 * Rq * 16);
    if (bf) p->h0_bf = take_bytes(2 * qt * SHM_WPT * H);
    if (sp) { p->x_tb_lo = take_bytes(2 * Rq * 16); p->h0_lo = take_bytes(2 * qt * SHM_WPT * H); }
    p->params_copy = take((size_t)p->n_params);
    p->h_last = take((size_t)B * H); p->eps = take((size_t)B * Z);
    p->y = take((size_t)B * H); p->xhat = take((size_t)B * H); p->rstd = take(B);
    p->mu = take((size_t)B * Z); p->lv = take((size_t)B * Z); p->z = take((size_t)B * Z); p->h0 = take((size_t)B * H);
    p->score = take(B);
    if (training) {
        p->dpre_l2h = take((size_t)B * H); p->dmu = take((size_t)B * Z); p->dlv = take((size_t)B * Z);
        p->dyx = take((size_t)B * H); p->dy = take((size_t)B * H); p->dh_last = take((size_t)B * H); p->dh0 = take((size_t)B * H);
        p->recon = take(BT * D); p->d_recon = take(BT * D); p->d_mu = take((size_t)B * Z); p->d_lv = take((size_t)B * Z);
        p->dpre_seq = bf ? take_bytes(2 * Rq * 4 * H) : take(BT * 4 * H);   // bf16: whole tiles (unconditional stores)
        p->dx_seq = take(bf ? Rq * H : BT * H); p->dpre_tsum = take((size_t)B * 4 * H);
        if (sp) {
            p->dpre_lo = take_bytes(2 * Rq * 4 * H);
            for (int k = 0; k < 4; ++k) {
                if ((k & 1) >= L) continue;                       // set = stack * 2 + layer
                p->dpre_set[k] = k == 0 ? p->dpre_seq : take_bytes(2 * Rq * 4 * H);
                p->dpre_set_lo[k] = k == 0 ? p->dpre_lo : take_bytes(2 * Rq * 4 * H);
                p->dpre_set_tsum[k] = k == 0 ? p->dpre_tsum : take((size_t)B * 4 * H);
            }
        }
        if (bf && !sp && L == 2) { p->dpre_seq2 = take_bytes(2 * Rq * 4 * H); p->dpre_tsum2 = take((size_t)B * 4 * H); }
        size_t gp = 0;
        auto upd = [&](int R, int M, int N) { const size_t e = gemm_tn_partial_elems(R, M, N); if (e > gp) gp = e; };
        upd((int)BT, H, D); upd((int)BT, 4 * H, H); upd((int)BT, 4 * H, D); upd(B, 4 * H, H); upd(B, H, Z); upd(B, Z, H);
        if (bf) {
            size_t e = gemm_tn_bf16_partial_elems((int)(qt * SHM_WPT * T), 4 * H, H, 2); if (e > gp) gp = e;
            e = gemm_tn_bf16_partial_elems((int)(qt * SHM_WPT * T), 4 * H, H, 1); if (e > gp) gp = e;
        }
        p->gemm_partial_elems = gp;
        p->narrow_scratch = take(narrow_tn_scratch_elems(4 * H, D));
        p->narrow_scratch2 = take(narrow_tn_scratch_elems(H, D));
        p->gemm_partial = take(gp > 0 ? gp : 1);
        if (bf && !sp && L == 2) {
            size_t eb = gemm_tn_bf16_partial_elems((int)(qt * SHM_WPT * T), 4 * H, H, 1);    // encoder layer 0: dW_hh (+ the 16-column dW_ih)
            const size_t eb2 = gemm_tn_bf16_partial_elems((int)(qt * SHM_WPT * T), 4 * H, H, 2);  // decoder layer 0: dW_ih and dW_hh
            if (eb2 > eb) eb = eb2;
            p->gemm_partial_b = take(eb > 0 ? eb : 1);
        }
        if (sp) p->gemm_partial_b = take(gp > 0 ? gp : 1);      // split path: the LAST layer's products run on the caller's stream beside the side stream's (own scratch)
        if (bf) {        // the fp32 dec0 dW_ih GEMM's own split-K scratch (pair path and split path batch it with the layer's other reductions)
            const size_t ec = gemm_tn_partial_elems(B, 4 * H, H);
            p->gemm_partial_c = take(ec > 0 ? ec : 1);
        }
        size_t cs = colsum_scratch_elems((int)BT, D);
        const size_t cs2 = colsum_scratch_elems(B, 4 * H);
        if (cs2 > cs) cs = cs2;
        p->colsum_scratch = take(cs);
    }
    p->ws_bytes = off;
    *out = p;
    return SHM_OK;
}

extern "C" void shm_vae_plan_destroy(shm_vae_plan* plan) {
    if (plan && plan->side_ready) {
        (void)hipStreamSynchronize(plan->side);
        (void)hipStreamSynchronize(plan->side2);
        (void)hipEventDestroy(plan->ev_fork); (void)hipEventDestroy(plan->ev_join); (void)hipEventDestroy(plan->ev_join2);
        for (auto& e : plan->ev_rec) (void)hipEventDestroy(e);
        for (auto& e : plan->ev_chunk) (void)hipEventDestroy(e);
        (void)hipStreamDestroy(plan->side); (void)hipStreamDestroy(plan->side2);
    }
    delete plan;
}

static int ensure_side_stream(shm_vae_plan* p) {
    if (p->side_ready) return SHM_OK;
    SHM_HIP_CHECK(hipStreamCreateWithFlags(&p->side, hipStreamNonBlocking));
    SHM_HIP_CHECK(hipStreamCreateWithFlags(&p->side2, hipStreamNonBlocking));
    SHM_HIP_CHECK(hipEventCreateWithFlags(&p->ev_fork, hipEventDisableTiming));
    SHM_HIP_CHECK(hipEventCreateWithFlags(&p->ev_join, hipEventDisableTiming));
    SHM_HIP_CHECK(hipEventCreateWithFlags(&p->ev_join2, hipEventDisableTiming));
    for (auto& e : p->ev_rec) SHM_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto& e : p->ev_chunk) SHM_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    p->side_ready = true;
    return SHM_OK;
}
// SHM_SIDE_STREAM=0: everything on the caller's stream (A/B measurements; results are bit-identical either way)
static bool side_stream_enabled() {
    static int on = -1;
    if (on < 0) { const char* e = getenv("SHM_SIDE_STREAM"); on = (e && e[0] == '0') ? 0 : 1; }
    return on != 0;
}
// Split path, 2-layer stacks: the window is cut into NC time chunks and the two layers run as a launch-level pipeline — layer 0 walks
// chunk c+1 on the caller's stream while the side stream projects chunk c and runs layer 1 on it (a recurrent launch holds 128 of the
// 256 CUs, so the two recurrences share the chip).  SHM_SP_CHUNKS=N (1 = off, the default: measured in round 2 at the north-star shape,
// NC = 2 / 4 / 8 gave 179 / 182 / 151 k windows/s against 183 k unchunked — two recurrences side by side each slow from 57 to 85-90 us
// per 64 steps because one of them already draws ~3 TB/s of record traffic).  Chunks are whole 16-row tiles: T % (4 NC) == 0.
static int sp_chunks(const shm_vae_plan* p, bool records) {
    const char* e = getenv("SHM_SP_CHUNKS");             // read per call: a test flips it inside one process
    // default: off for the training passes (measured above), 4 for a forward that saves no BPTT records (inference plans: scoring,
    // encode / decode) — without the 402 MB record stream per layer two recurrences do fit side by side: 502 -> 561 k windows/s scored
    // round 3: the forward of a 2-layer stack is ONE layer-pipelined launch (lstm_fwd_pair_sp_kernel) unless pair launches are switched
    // off; the launch-level chunk pipeline remains for that case and for an explicit SHM_SP_CHUNKS
    if (!e && pair_launch_enabled()) return 1;
    int want = e ? atoi(e) : (records ? SHM_SP_CHUNKS_DEFAULT : 4);
    if (want < 1) want = 1;
    if (want > 8) want = 8;
    if (p->cfg.precision != SHM_PREC_SPLIT || p->cfg.num_layers != 2 || !side_stream_enabled()) return 1;
    int nc = want;
    while (nc > 1 && (p->T % (4 * nc) != 0 || p->T / nc < 16)) --nc;
    return nc;
}
extern "C" size_t shm_vae_plan_workspace_bytes(const shm_vae_plan* plan) { return plan ? plan->ws_bytes : 0; }

static int check_ws(const shm_vae_plan* p, const void* ws) {
    SHM_REQUIRE(p != nullptr, SHM_EINVAL, "plan is NULL");
    SHM_REQUIRE(ws != nullptr, SHM_EWORKSPACE, "workspace is NULL");
    SHM_REQUIRE((reinterpret_cast<uintptr_t>(ws) & 255) == 0, SHM_EWORKSPACE, "workspace must be 256-byte aligned");
    return SHM_OK;
}

extern "C" int shm_vae_pack_weights(shm_vae_plan* p, const float* params, void* ws, void* stream) {
    int rc = check_ws(p, ws);
    if (rc) return rc;
    SHM_REQUIRE(params != nullptr, SHM_EINVAL, "pack_weights: params is NULL");
    SHM_REQUIRE((reinterpret_cast<uintptr_t>(params) & 15) == 0, SHM_EINVAL, "pack_weights: params must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const int H = p->cfg.hidden_dim, L = p->cfg.num_layers;
    const bool sp = p->cfg.precision == SHM_PREC_SPLIT;
    if (p->cfg.precision == SHM_PREC_BF16 || sp) {    // bf16 / split paths: every re-layout, the flat copy and the state words in one launch
        SHM_REQUIRE(p->n_params < ((int64_t)1 << 31), SHM_EINVAL, "pack_weights: parameter count exceeds 2^31");
        PackJobs jobs;
        jobs.n = 0;
        auto add = [&](const float* src, const float* src2, void* dst, int kind, int K) {
            PackJob& j = jobs.job[jobs.n++];
            j.src = src; j.src2 = src2; j.dst = dst; j.kind = kind; j.H = H; j.K = K;
        };
        add(params, nullptr, at<float>(ws, p->params_copy), 3, (int)p->n_params);
        // zero line + tile counters on every pack; the sticky error word in front of them ONLY the first time (the workspace arrives
        // uninitialised) — afterwards it survives re-packing, so a hand-off timeout stays visible until shm_vae_plan_reset_status
        if (!p->err_armed) { add(nullptr, nullptr, at<char>(ws, p->misc), 4, (int)(misc_bytes(p->B) / 4)); p->err_armed = true; }
        else add(nullptr, nullptr, at<char>(ws, p->misc + 64), 4, (int)((misc_bytes(p->B) - 64) / 4));
        if (p->pair_sync)                                                                      // the block the next pair launch uses
            add(nullptr, nullptr, at<char>(ws, p->pair_sync + (size_t)p->sync_next * lstm_fwd_pair_sync_bytes(p->B)), 4,
                (int)(lstm_fwd_pair_sync_bytes(p->B) / 4));
        for (int l = 0; l < L; ++l)
            for (int st = 0; st < 2; ++st) {
                const LayerInst& li = st == 0 ? p->enc[l] : p->dec[l];
                add(params + li.b_ih, params + li.b_hh, at<float>(ws, li.bsum), 2, 0);
                // split path, layers whose input is a hidden SEQUENCE (layers >= 1): W_ih as fp16 +
                // fp8 residual (kinds 7 / 8, lstm_sp.hip: x_convert); everything else as bf16 (pairs)
                const bool w16 = sp && l > 0;             // (the constant-input decoder layer 0 keeps the bf16 pair: its projection is its own launch)
                add(params + li.w_ih, nullptr, at<bf16_t>(ws, li.Wx), w16 ? 7 : 0, li.In);
                add(params + li.w_hh, nullptr, at<bf16_t>(ws, li.Wh), 0, H);
                if (sp) {
                    add(params + li.w_ih, nullptr, at<bf16_t>(ws, li.Wx_lo), w16 ? 8 : 5, li.In);
                    add(params + li.w_hh, nullptr, at<bf16_t>(ws, li.Wh_lo), 5, H);
                }
                if (p->training) {
                    add(params + li.w_hh, nullptr, at<bf16_t>(ws, li.WhT), 1, H);
                    if (l > 0) add(params + li.w_ih, nullptr, at<bf16_t>(ws, li.WxT), 1, li.In);
                    if (sp) {
                        add(params + li.w_hh, nullptr, at<bf16_t>(ws, li.WhT_lo), 6, H);
                        if (l > 0) add(params + li.w_ih, nullptr, at<bf16_t>(ws, li.WxT_lo), 6, li.In);
                    }
                }
            }
        if ((rc = launch_pack_all(jobs, s))) return rc;
        p->packed = true;
        return SHM_OK;
    }
    SHM_HIP_CHECK(hipMemcpyAsync(at<float>(ws, p->params_copy), params, sizeof(float) * p->n_params, hipMemcpyDeviceToDevice, s));
    for (int l = 0; l < L; ++l) {
        for (int st = 0; st < 2; ++st) {
            const LayerInst& li = st == 0 ? p->enc[l] : p->dec[l];
            if ((rc = launch_bias_sum(params + li.b_ih, params + li.b_hh, 4 * H, at<float>(ws, li.bsum), s))) return rc;
            if (p->cfg.precision == SHM_PREC_BF16) {
                if ((rc = launch_pack_fwd_bf16(params + li.w_ih, H, li.In, at<bf16_t>(ws, li.Wx), s))) return rc;
                if ((rc = launch_pack_fwd_bf16(params + li.w_hh, H, H, at<bf16_t>(ws, li.Wh), s))) return rc;
                if (p->training) {
                    if ((rc = launch_pack_bwd_bf16(params + li.w_hh, H, H, at<bf16_t>(ws, li.WhT), s))) return rc;
                    if (l > 0 && (rc = launch_pack_bwd_bf16(params + li.w_ih, H, li.In, at<bf16_t>(ws, li.WxT), s))) return rc;
                }
                continue;
            }
            if ((rc = launch_pack_fwd_f32(params + li.w_ih, H, li.In, at<float>(ws, li.Wx), s))) return rc;
            if ((rc = launch_pack_fwd_f32(params + li.w_hh, H, H, at<float>(ws, li.Wh), s))) return rc;
            if (p->training) {
                if ((rc = launch_pack_bwd_f32(params + li.w_hh, H, H, at<float>(ws, li.WhT), s))) return rc;
                if (l > 0 && (rc = launch_pack_bwd_f32(params + li.w_ih, H, li.In, at<float>(ws, li.WxT), s))) return rc;
            }
        }
    }
    p->packed = true;
    return SHM_OK;
}

// what: 0 = TemporalVAE.forward (temporal_vae.py:72-77), 1 = encode only (:51-58), 2 = decode only from z_in (:65-70)
enum { VAE_FULL = 0, VAE_ENCODE = 1, VAE_DECODE = 2 };
static int vae_forward_impl(shm_vae_plan* p, const float* x, const float* eps, const HeadNoise& rng, const float* z_in, float* recon, float* mu,
                            float* logvar, float* score, int32_t train_mode, uint64_t drop_seed, void* ws, void* stream, int what) {
    int rc = check_ws(p, ws);
    if (rc) return rc;
    SHM_REQUIRE(p->packed, SHM_ESTATE, "forward: call shm_vae_pack_weights first");
    hipStream_t s = (hipStream_t)stream;
    const shm_vae_config& c = p->cfg;
    const int B = p->B, T = p->T, D = c.input_dim, H = c.hidden_dim, Z = c.latent_dim, L = c.num_layers;
    const bool save = p->training != 0;
    const float drop = (train_mode && L > 1) ? c.dropout : 0.f;
    p->drop_active = drop > 0.f; p->drop_seed = drop_seed;
    p->fwd_saved = false; p->loss_done = false;

    auto run_stack = [&](LayerInst* st, const float* in0, uint32_t stack, float* h_last) -> int {
        auto fwd_args = [&](int l) {
            LstmFwdArgsB a;
            memset(&a, 0, sizeof(a));
            a.in = l == 0 ? (const void*)in0 : (const void*)at<bf16_t>(ws, st[l - 1].h_seq);
            a.in_f32 = l == 0 ? 1 : 0;
            a.Wx = at<bf16_t>(ws, st[l].Wx); a.Wh = at<bf16_t>(ws, st[l].Wh); a.bias = at<float>(ws, st[l].bsum);
            a.h_seq = at<bf16_t>(ws, st[l].h_seq);
            a.h_last = (l == L - 1) ? h_last : nullptr;
            a.sv_g = save ? at<bf16_t>(ws, st[l].saved) : nullptr;
            a.xdrop = (l > 0 && c.dropout > 0.f) ? at<bf16_t>(ws, st[l].xdrop) : nullptr;
            a.x_tb = (l == 0 && !st[l].in_const) ? at<bf16_t>(ws, p->x_tb) : nullptr;
            a.h0_bf = st[l].in_const ? at<bf16_t>(ws, p->h0_bf) : nullptr;
            a.B = B; a.T = T; a.In = st[l].In; a.in_const = st[l].in_const ? 1 : 0;
            a.drop_p = l > 0 ? drop : 0.f; a.drop_seed = drop_seed; a.drop_stack = stack;
            return a;
        };
        if (c.precision == SHM_PREC_BF16 && L == 2 && pair_launch_enabled()) {   // both layers in one launch, layer 1 a few steps behind
            LstmFwdPairArgsB pa;
            pa.l0 = fwd_args(0); pa.l1 = fwd_args(1); pa.role_lds_off = 0;
            pa.sync = at<unsigned>(ws, p->pair_sync + (size_t)p->sync_next * lstm_fwd_pair_sync_bytes(B));
            pa.sync_other = at<unsigned>(ws, p->pair_sync + (size_t)(p->sync_next ^ 1) * lstm_fwd_pair_sync_bytes(B));
            p->sync_next ^= 1;
            pa.err = at<unsigned>(ws, p->misc);
            KernelTimer kt(stack == 0 ? "lstm_fwd.enc" : "lstm_fwd.dec", s);
            return launch_lstm_fwd_pair_bf16(H, pa, s);
        }
        for (int l = 0; l < L; ++l) {
            char nm[32];
            snprintf(nm, sizeof(nm), "lstm_fwd.%s%d", stack == 0 ? "enc" : "dec", l);
            if (c.precision == SHM_PREC_BF16) {
                const LstmFwdArgsB a = fwd_args(l);
                KernelTimer kt(nm, s);
                int r = launch_lstm_fwd_bf16(H, a, s);
                if (r) return r;
                continue;
            }
            LstmFwdArgs a;
            memset(&a, 0, sizeof(a));
            a.in = l == 0 ? in0 : at<float>(ws, st[l - 1].h_seq);
            a.Wx = at<float>(ws, st[l].Wx); a.Wh = at<float>(ws, st[l].Wh); a.bias = at<float>(ws, st[l].bsum);
            a.h_seq = at<float>(ws, st[l].h_seq);
            a.h_last = (l == L - 1) ? h_last : nullptr;
            a.saved = save ? at<float>(ws, st[l].saved) : nullptr;
            a.B = B; a.T = T; a.In = st[l].In; a.in_const = st[l].in_const ? 1 : 0;
            a.drop_p = l > 0 ? drop : 0.f; a.drop_seed = drop_seed; a.drop_stack = stack;
            KernelTimer kt(nm, s);
            int r = launch_lstm_fwd_f32(H, a, s);
            if (r) return r;
        }
        return SHM_OK;
    };
    // split-precision path: per layer [time-parallel input projection ->] recurrence with W_hh hi / lo in registers (lstm_sp.hip)
    const bool sp = c.precision == SHM_PREC_SPLIT;
    const int NC = sp ? sp_chunks(p, save) : 1;
    if (NC > 1 && (rc = ensure_side_stream(p))) return rc;
    const bool sp_pair = sp && L == 2 && NC == 1 && pair_launch_enabled() && !getenv("SHM_SP_CHUNKS");
    auto run_stack_sp = [&](LayerInst* st, const float* in0, uint32_t stack, float* h_last) -> int {
        const int qt = ceil_div(B, SHM_WPT);
        if (sp_pair) {
            // both layers in ONE launch: layer 0 hands its h planes to layer 1 inside the launch, layer 1 forms its own input projection
            // (no xproj launch, no 268 MB projection stream); bit-identical to the one-launch-per-layer form below
            const char* sn = stack == 0 ? "enc" : "dec";
            char nm[32];
            LstmFwdPairArgsS pa;
            memset(&pa, 0, sizeof(pa));
            for (int l = 0; l < 2; ++l) {
                LstmFwdArgsS& a = l ? pa.l1 : pa.l0;
                a.Wh_hi = at<bf16_t>(ws, st[l].Wh); a.Wh_lo = at<bf16_t>(ws, st[l].Wh_lo);
                a.h_hi = at<bf16_t>(ws, st[l].h_seq); a.h_lo = at<bf16_t>(ws, st[l].h_lo);
                a.rec4 = save ? at<float>(ws, st[l].saved) : nullptr; a.rec2 = save ? at<float>(ws, st[l].saved_c) : nullptr;
                a.B = B; a.T = T; a.In = st[l].In;
            }
            if (!st[0].in_const) {
                pa.l0.mode = 2; pa.l0.x = in0; pa.l0.Wx_hi = at<bf16_t>(ws, st[0].Wx); pa.l0.Wx_lo = at<bf16_t>(ws, st[0].Wx_lo); pa.l0.bias = at<float>(ws, st[0].bsum);
                if (save) { pa.l0.x_hi = at<bf16_t>(ws, p->x_tb); pa.l0.x_lo = at<bf16_t>(ws, p->x_tb_lo); }
            } else {        // decoder layer 0: one projection per window (temporal_vae.py:68 repeats h0 over T), its own small launch
                XprojArgsS xa;
                memset(&xa, 0, sizeof(xa));
                xa.W_hi = at<bf16_t>(ws, st[0].Wx); xa.W_lo = at<bf16_t>(ws, st[0].Wx_lo); xa.bias = at<float>(ws, st[0].bsum);
                xa.out = at<float>(ws, st[0].xproj);
                int r = launch_split_planes(in0, (size_t)B * H, (size_t)qt * SHM_WPT * H, at<bf16_t>(ws, p->h0_bf), at<bf16_t>(ws, p->h0_lo), s);
                if (r) return r;
                xa.in_hi = at<bf16_t>(ws, p->h0_bf); xa.in_lo = at<bf16_t>(ws, p->h0_lo); xa.rows = qt * SHM_WPT; xa.pair_bf16 = 1;
                snprintf(nm, sizeof(nm), "xproj.%s0", sn);
                KernelTimer kt(nm, s);
                if ((r = launch_xproj_sp(H, xa, s))) return r;
                pa.l0.mode = 1; pa.l0.xproj = at<float>(ws, st[0].xproj);
            }
            pa.l1.mode = 3; pa.l1.h_last = h_last;
            pa.l1.Wx_hi = at<bf16_t>(ws, st[1].Wx); pa.l1.Wx_lo = at<bf16_t>(ws, st[1].Wx_lo); pa.l1.bias = at<float>(ws, st[1].bsum);
            pa.l1.in_hi = at<bf16_t>(ws, st[0].h_seq); pa.l1.in_lo = at<bf16_t>(ws, st[0].h_lo);
            pa.l1.drop_p = drop; pa.l1.drop_seed = drop_seed; pa.l1.drop_stack = stack;
            if (drop > 0.f) { pa.l1.xd_hi = at<bf16_t>(ws, st[1].xdrop); pa.l1.xd_lo = at<bf16_t>(ws, st[1].xdrop_lo); }
            pa.sync = at<unsigned>(ws, p->pair_sync + (size_t)p->sync_next * lstm_fwd_pair_sync_bytes(B));
            pa.sync_other = at<unsigned>(ws, p->pair_sync + (size_t)(p->sync_next ^ 1) * lstm_fwd_pair_sync_bytes(B));
            p->sync_next ^= 1;
            pa.err = at<unsigned>(ws, p->misc);
            snprintf(nm, sizeof(nm), "lstm_fwd.%s", sn);
            KernelTimer kt(nm, s);
            return launch_lstm_fwd_pair_sp(H, pa, s);
        }
        if (NC > 1) {
            // launch-level layer pipeline over NC time chunks (see sp_chunks): s carries layer 0, the side stream projection + layer 1
            const char* sn = stack == 0 ? "enc" : "dec";
            char nm[32];
            hipStream_t ss = p->side;
            const int Tc = T / NC;
            LstmFwdArgsS a0, a1;
            memset(&a0, 0, sizeof(a0)); memset(&a1, 0, sizeof(a1));
            for (int l = 0; l < 2; ++l) {
                LstmFwdArgsS& a = l ? a1 : a0;
                a.Wh_hi = at<bf16_t>(ws, st[l].Wh); a.Wh_lo = at<bf16_t>(ws, st[l].Wh_lo);
                a.h_hi = at<bf16_t>(ws, st[l].h_seq); a.h_lo = at<bf16_t>(ws, st[l].h_lo);
                a.rec4 = save ? at<float>(ws, st[l].saved) : nullptr; a.rec2 = save ? at<float>(ws, st[l].saved_c) : nullptr;
                a.B = B; a.T = T; a.In = st[l].In; a.c_state = at<float>(ws, st[l].c_state);
            }
            a1.h_last = h_last; a1.mode = 0; a1.xproj = at<float>(ws, st[1].xproj);
            if (!st[0].in_const) {
                a0.mode = 2; a0.x = in0; a0.Wx_hi = at<bf16_t>(ws, st[0].Wx); a0.Wx_lo = at<bf16_t>(ws, st[0].Wx_lo); a0.bias = at<float>(ws, st[0].bsum);
                if (save) { a0.x_hi = at<bf16_t>(ws, p->x_tb); a0.x_lo = at<bf16_t>(ws, p->x_tb_lo); }
            } else {
                XprojArgsS xa;
                memset(&xa, 0, sizeof(xa));
                xa.W_hi = at<bf16_t>(ws, st[0].Wx); xa.W_lo = at<bf16_t>(ws, st[0].Wx_lo); xa.bias = at<float>(ws, st[0].bsum);
                xa.out = at<float>(ws, st[0].xproj);
                int r = launch_split_planes(in0, (size_t)B * H, (size_t)qt * SHM_WPT * H, at<bf16_t>(ws, p->h0_bf), at<bf16_t>(ws, p->h0_lo), s);
                if (r) return r;
                xa.in_hi = at<bf16_t>(ws, p->h0_bf); xa.in_lo = at<bf16_t>(ws, p->h0_lo); xa.rows = qt * SHM_WPT; xa.pair_bf16 = 1;
                snprintf(nm, sizeof(nm), "xproj.%s0", sn);
                KernelTimer kt(nm, s);
                if ((r = launch_xproj_sp(H, xa, s))) return r;
                a0.mode = 1; a0.xproj = at<float>(ws, st[0].xproj);
            }
            XprojArgsS x1;
            memset(&x1, 0, sizeof(x1));
            x1.W_hi = at<bf16_t>(ws, st[1].Wx); x1.W_lo = at<bf16_t>(ws, st[1].Wx_lo); x1.bias = at<float>(ws, st[1].bsum);
            x1.out = at<float>(ws, st[1].xproj);
            x1.in_hi = at<bf16_t>(ws, st[0].h_seq); x1.in_lo = at<bf16_t>(ws, st[0].h_lo); x1.rows = qt * SHM_WPT * T;
            x1.drop_p = drop; x1.drop_seed = drop_seed; x1.drop_stack = stack;
            if (drop > 0.f) { x1.xd_hi = at<bf16_t>(ws, st[1].xdrop); x1.xd_lo = at<bf16_t>(ws, st[1].xdrop_lo); }
            x1.cpt = Tc / 4; x1.seq_tiles = T / 4; x1.n_chunk_tiles = qt * x1.cpt;
            for (int ch = 0; ch < NC; ++ch) {
                int r;
                a0.t0 = a1.t0 = ch * Tc; a0.t1 = a1.t1 = (ch + 1) * Tc;
                {
                    snprintf(nm, sizeof(nm), "lstm_fwd.%s0", sn);
                    KernelTimer kt(nm, s);
                    if ((r = launch_lstm_fwd_sp(H, a0, s))) return r;
                }
                hipEvent_t ev = p->ev_chunk[(int)stack * 8 + ch];
                SHM_HIP_CHECK(hipEventRecord(ev, s));
                SHM_HIP_CHECK(hipStreamWaitEvent(ss, ev, 0));
                x1.tile_off = a0.t0 / 4;
                {
                    snprintf(nm, sizeof(nm), "xproj.%s1", sn);
                    KernelTimer kt(nm, ss);
                    if ((r = launch_xproj_sp(H, x1, ss))) return r;
                }
                snprintf(nm, sizeof(nm), "lstm_fwd.%s1", sn);
                KernelTimer kt(nm, ss);
                if ((r = launch_lstm_fwd_sp(H, a1, ss))) return r;
            }
            SHM_HIP_CHECK(hipEventRecord(p->ev_join, ss));
            SHM_HIP_CHECK(hipStreamWaitEvent(s, p->ev_join, 0));
            return SHM_OK;
        }
        for (int l = 0; l < L; ++l) {
            const LayerInst& li = st[l];
            const char* sn = stack == 0 ? "enc" : "dec";
            char nm[32];
            LstmFwdArgsS a;
            memset(&a, 0, sizeof(a));
            a.Wh_hi = at<bf16_t>(ws, li.Wh); a.Wh_lo = at<bf16_t>(ws, li.Wh_lo);
            a.h_hi = at<bf16_t>(ws, li.h_seq); a.h_lo = at<bf16_t>(ws, li.h_lo);
            a.h_last = (l == L - 1) ? h_last : nullptr;
            a.rec4 = save ? at<float>(ws, li.saved) : nullptr; a.rec2 = save ? at<float>(ws, li.saved_c) : nullptr;
            a.B = B; a.T = T; a.In = li.In;
            if (l == 0 && !li.in_const) {
                a.mode = 2; a.x = in0; a.Wx_hi = at<bf16_t>(ws, li.Wx); a.Wx_lo = at<bf16_t>(ws, li.Wx_lo); a.bias = at<float>(ws, li.bsum);
                if (save) { a.x_hi = at<bf16_t>(ws, p->x_tb); a.x_lo = at<bf16_t>(ws, p->x_tb_lo); }
            } else {
                XprojArgsS xa;
                memset(&xa, 0, sizeof(xa));
                xa.W_hi = at<bf16_t>(ws, li.Wx); xa.W_lo = at<bf16_t>(ws, li.Wx_lo); xa.bias = at<float>(ws, li.bsum);
                xa.out = at<float>(ws, li.xproj);
                if (li.in_const) {        // decoder layer 0: one projection per window (temporal_vae.py:68 repeats h0 over T)
                    int r = launch_split_planes(in0, (size_t)B * H, (size_t)qt * SHM_WPT * H, at<bf16_t>(ws, p->h0_bf), at<bf16_t>(ws, p->h0_lo), s);
                    if (r) return r;
                    xa.in_hi = at<bf16_t>(ws, p->h0_bf); xa.in_lo = at<bf16_t>(ws, p->h0_lo); xa.rows = qt * SHM_WPT; xa.pair_bf16 = 1;
                    a.mode = 1;
                } else {
                    xa.in_hi = at<bf16_t>(ws, st[l - 1].h_seq); xa.in_lo = at<bf16_t>(ws, st[l - 1].h_lo); xa.rows = qt * SHM_WPT * T;
                    xa.drop_p = drop; xa.drop_seed = drop_seed; xa.drop_stack = stack;
                    if (drop > 0.f) { xa.xd_hi = at<bf16_t>(ws, li.xdrop); xa.xd_lo = at<bf16_t>(ws, li.xdrop_lo); }
                    a.mode = 0;
                }
                snprintf(nm, sizeof(nm), "xproj.%s%d", sn, l);
                KernelTimer kt(nm, s);
                int r = launch_xproj_sp(H, xa, s);
                if (r) return r;
                a.xproj = at<float>(ws, li.xproj);
            }
            snprintf(nm, sizeof(nm), "lstm_fwd.%s%d", sn, l);
            KernelTimer kt(nm, s);
            int r = launch_lstm_fwd_sp(H, a, s);
            if (r) return r;
        }
        return SHM_OK;
    };
    const float* prm = at<float>(ws, p->params_copy);
    if (what != VAE_DECODE) {
        if (sp) { if ((rc = run_stack_sp(p->enc, x, 0, at<float>(ws, p->h_last)))) return rc; }
        else if ((rc = run_stack(p->enc, x, 0, at<float>(ws, p->h_last)))) return rc;
    }
    {
        // the noise that is used (injected, or drawn here by the counter RNG) is kept in the workspace for the backward
        KernelTimer kt("head_fwd", s);
        if ((rc = launch_head_fwd(B, H, Z, c.layer_norm, at<float>(ws, p->h_last), eps, head_params(p, prm),
                                  head_bufs(p, ws), mu, logvar, s, rng, what == VAE_DECODE ? z_in : nullptr, at<float>(ws, p->eps)))) return rc;
    }
    if (what == VAE_ENCODE) return SHM_OK;
    if (sp) { if ((rc = run_stack_sp(p->dec, at<float>(ws, p->h0), 1, nullptr))) return rc; }
    else if ((rc = run_stack(p->dec, at<float>(ws, p->h0), 1, nullptr))) return rc;
    float* rec = recon ? recon : (p->training ? at<float>(ws, p->recon) : nullptr);
    {
        const bool tb = c.precision != SHM_PREC_F32;       // tile-blocked bf16 sequences (bf16 and split paths)
        KernelTimer kt("out_fwd", s);
        if ((rc = launch_out_fwd(B, T, D, H, at<char>(ws, p->dec[L - 1].h_seq), tb, prm + p->out_w, prm + p->out_b, x, rec,
                                 at<float>(ws, p->score), s,
                                 (tb && p->training) ? at<float>(ws, p->d_recon) : nullptr,
                                 tb ? at<float>(ws, p->score_part) : nullptr,
                                 tb ? at<unsigned>(ws, p->misc + 320) : nullptr,
                                 sp ? at<char>(ws, p->dec[L - 1].h_lo) : nullptr))) return rc;
    }
    if (score) SHM_HIP_CHECK(hipMemcpyAsync(score, at<float>(ws, p->score), sizeof(float) * B, hipMemcpyDeviceToDevice, s));
    p->recon_ptr = rec;
    p->fwd_saved = save && what == VAE_FULL;
    return SHM_OK;
}

extern "C" int shm_vae_forward(shm_vae_plan* p, const float* x, const float* eps, float* recon, float* mu, float* logvar,
                               float* score, int32_t train_mode, uint64_t drop_seed, void* ws, void* stream) {
    SHM_REQUIRE(x && eps, SHM_EINVAL, "forward: x or eps is NULL");
    return vae_forward_impl(p, x, eps, HeadNoise{0, 0, 0}, nullptr, recon, mu, logvar, score, train_mode, drop_seed, ws, stream, VAE_FULL);
}

extern "C" int shm_vae_forward_sample(shm_vae_plan* p, const float* x, uint64_t eps_seed, uint64_t eps_offset, float* recon, float* mu,
                                      float* logvar, float* score, int32_t train_mode, uint64_t drop_seed, void* ws, void* stream) {
    SHM_REQUIRE(x != nullptr, SHM_EINVAL, "forward_sample: x is NULL");
    return vae_forward_impl(p, x, nullptr, HeadNoise{1, eps_seed, eps_offset}, nullptr, recon, mu, logvar, score, train_mode, drop_seed, ws, stream, VAE_FULL);
}

extern "C" int shm_vae_encode(shm_vae_plan* p, const float* x, float* mu, float* logvar, int32_t train_mode, uint64_t drop_seed, void* ws,
                              void* stream) {
    SHM_REQUIRE(x && (mu || logvar), SHM_EINVAL, "encode: x or both outputs NULL");
    return vae_forward_impl(p, x, nullptr, HeadNoise{0, 0, 0}, nullptr, nullptr, mu, logvar, nullptr, train_mode, drop_seed, ws, stream, VAE_ENCODE);
}

extern "C" int shm_vae_decode(shm_vae_plan* p, const float* z, float* recon, int32_t train_mode, uint64_t drop_seed, void* ws, void* stream) {
    SHM_REQUIRE(z && recon, SHM_EINVAL, "decode: z or recon is NULL");
    return vae_forward_impl(p, nullptr, nullptr, HeadNoise{0, 0, 0}, z, recon, nullptr, nullptr, nullptr, train_mode, drop_seed, ws, stream, VAE_DECODE);
}

extern "C" void shm_eps_normal(uint64_t seed, uint64_t offset, int64_t n, float* out) {
    for (int64_t i = 0; i < n; ++i) out[i] = eps_normal(seed, offset, (uint32_t)i);
}

extern "C" int shm_set_pair_launch(int on) {
    const int prev = pair_launch_enabled() ? 1 : 0;
    g_pair_launch = on ? 1 : 0;
    return prev;
}

extern "C" int shm_vae_plan_status(const shm_vae_plan* p, const void* ws, void* stream) {
    int rc = check_ws(p, ws);
    if (rc) return rc;
    if (!p->pair_sync || !p->err_armed) return SHM_OK;
    unsigned word = 0;
    SHM_HIP_CHECK(hipMemcpyAsync(&word, reinterpret_cast<const char*>(ws) + p->misc, 4,
                                 hipMemcpyDeviceToHost, (hipStream_t)stream));
    SHM_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    SHM_REQUIRE(word == 0, SHM_ESTATE, "layer-pipelined launch: the consumer of tile %u gave up waiting for its producer (status word 0x%x); every "
                "result of this plan since the word was last reset is invalid", (word & 0x7fffffffu) - 1, word);
    return SHM_OK;
}

extern "C" int shm_vae_plan_status_word(const shm_vae_plan* p, const void* ws, uint32_t* dst, void* stream) {
    int rc = check_ws(p, ws);
    if (rc) return rc;
    SHM_REQUIRE(dst != nullptr, SHM_EINVAL, "plan_status_word: dst is NULL");
    if (!p->pair_sync || !p->err_armed) { SHM_HIP_CHECK(hipMemsetAsync(dst, 0, 4, (hipStream_t)stream)); return SHM_OK; }
    SHM_HIP_CHECK(hipMemcpyAsync(dst, reinterpret_cast<const char*>(ws) + p->misc, 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return SHM_OK;
}

extern "C" int64_t shm_vae_plan_status_offset(const shm_vae_plan* p) { return (p && p->pair_sync) ? (int64_t)p->misc : -1; }

extern "C" int shm_vae_plan_reset_status(shm_vae_plan* p, void* ws, void* stream) {
    int rc = check_ws(p, ws);
    if (rc) return rc;
    if (!p->pair_sync) return SHM_OK;
    SHM_HIP_CHECK(hipMemsetAsync(reinterpret_cast<char*>(ws) + p->misc, 0, 64, (hipStream_t)stream));
    p->err_armed = true;
    return SHM_OK;
}

extern "C" int shm_vae_loss(shm_vae_plan* p, const float* x, float kl_w, float* losses, void* ws, void* stream) {
    int rc = check_ws(p, ws);
    if (rc) return rc;
    SHM_REQUIRE(p->training && p->fwd_saved && p->recon_ptr, SHM_ESTATE, "loss: needs a forward on a training plan first");
    SHM_REQUIRE(x && losses, SHM_EINVAL, "loss: x or losses is NULL");
    const shm_vae_config& c = p->cfg;
    if ((rc = launch_vae_loss(p->B, p->T, c.input_dim, c.latent_dim, x, p->recon_ptr, at<float>(ws, p->mu), at<float>(ws, p->lv),
                              kl_w, c.precision != SHM_PREC_F32 ? nullptr : at<float>(ws, p->d_recon), at<float>(ws, p->d_mu), at<float>(ws, p->d_lv),
                              at<float>(ws, p->score), losses, (hipStream_t)stream))) return rc;
    p->loss_done = true;
    return SHM_OK;
}

extern "C" int shm_vae_eval_loss(const float* mu, const float* logvar, const float* score, int32_t B, int32_t T, int32_t D, int32_t Z,
                                 float kl_w, float* losses, void* stream) {
    SHM_REQUIRE(mu && logvar && score && losses && B >= 1 && Z >= 1, SHM_EINVAL, "eval_loss: NULL argument or empty batch");
    return launch_vae_loss(B, T, D, Z, nullptr, nullptr, mu, logvar, kl_w, nullptr, nullptr, nullptr, const_cast<float*>(score), losses,
                           (hipStream_t)stream);
}

extern "C" int shm_vae_backward(shm_vae_plan* p, const float* x, const float* d_recon, const float* d_mu, const float* d_logvar,
                                float* grads, void* ws, void* stream) {
    int rc = check_ws(p, ws);
    if (rc) return rc;
    SHM_REQUIRE(p->training && p->fwd_saved, SHM_ESTATE, "backward: needs a forward on a training plan first");
    SHM_REQUIRE(x && grads, SHM_EINVAL, "backward: x or grads is NULL");
    const bool own = !d_recon && !d_mu && !d_logvar;
    SHM_REQUIRE(own || (d_recon && d_mu && d_logvar), SHM_EINVAL, "backward: pass all three upstream gradients or none");
    SHM_REQUIRE(!own || p->loss_done, SHM_ESTATE, "backward: no upstream gradients and shm_vae_loss was not called");
    if (own) { d_recon = at<float>(ws, p->d_recon); d_mu = at<float>(ws, p->d_mu); d_logvar = at<float>(ws, p->d_lv); }
    hipStream_t s = (hipStream_t)stream;
    const shm_vae_config& c = p->cfg;
    const int B = p->B, T = p->T, D = c.input_dim, H = c.hidden_dim, Z = c.latent_dim, L = c.num_layers;
    const int BT = B * T, G = 4 * H;
    const float drop = p->drop_active ? c.dropout : 0.f;
    const float* prm = at<float>(ws, p->params_copy);
    float* dpre_seq = at<float>(ws, p->dpre_seq);          // fp32 view (fp32 path only)
    float* dx_seq = at<float>(ws, p->dx_seq);
    float* dpre_tsum = at<float>(ws, p->dpre_tsum);
    float* cs = at<float>(ws, p->colsum_scratch);
    const HeadBufs hb = head_bufs(p, ws);

    const bool sp = c.precision == SHM_PREC_SPLIT;
    const bool bf = c.precision == SHM_PREC_BF16 || sp;
    // gs: the stream of the deferred weight-gradient products.  Split path: the plan's side stream, forked from / joined to s with events
    const bool use_side = sp && side_stream_enabled();
    hipStream_t gs = s;
    if (use_side) {
        if ((rc = ensure_side_stream(p))) return rc;
        gs = p->side;
        SHM_HIP_CHECK(hipEventRecord(p->ev_fork, s));
        SHM_HIP_CHECK(hipStreamWaitEvent(gs, p->ev_fork, 0));
    }
    // SHM_SP_BWD_CHUNKS=N: launch-level chunk pipeline for the BACKWARD only (the forward keeps its pair launch): a measurement knob
    int NCb = use_side ? sp_chunks(p, true) : 1;
    if (use_side && L == 2) {
        const char* eb = getenv("SHM_SP_BWD_CHUNKS");
        int want = eb ? atoi(eb) : 1;
        if (want > 8) want = 8;
        while (want > 1 && (T % (4 * want) != 0 || T / want < 16)) --want;
        if (want > 1) NCb = want;
    }
    // wide products on bf16 operands: dW_ih = dpre^T x_in (dropout re-applied) and dW_hh = dpre^T h_{t-1} in ONE launch (A read once)
    // lo planes of the split path: {A, Bin, Bh, Bx}
    struct LoPlanes { const void* A; const void* Bin; const void* Bh; const void* Bx; };
    auto gemm_bf = [&](const void* A, int M, const void* Bin, float* Cin, float dp, uint32_t stack, const void* Bh, float* Chh, int R,
                       ReduceJobs* defer = nullptr, float* part = nullptr, const void* Bx = nullptr, float* Cx = nullptr,
                       const LoPlanes* lo = nullptr) -> int {
        GemmTnArgsB g;
        memset(&g, 0, sizeof(g));
        if (lo) { g.nseg = 3; g.A_lo = lo->A; g.Bx_lo = lo->Bx; if (Bin) { g.Bm_lo = lo->Bin; g.Bm2_lo = lo->Bh; } else g.Bm_lo = lo->Bh; }
        g.Bx = Bx; g.Cx = Cx; g.Nx = D;
        if (dp < 0.f) { g.bcast_T = T; dp = 0.f; }      // dp < 0 marks "Bin holds one row per window"
        g.A = A; g.lda = M; g.ldb = H; g.R = R; g.M = M; g.N = H; g.shift_rows = SHM_WPT;
        if (Bin) { g.NB = 2; g.Bm = Bin; g.C = Cin; g.shift_T = 0; g.Bm2 = Bh; g.C2 = Chh; g.shift_T2 = T; }
        else { g.NB = 1; g.Bm = Bh; g.C = Chh; g.shift_T = T; }
        g.drop_p = dp; g.drop_seed = p->drop_seed; g.drop_stack = stack;
        g.partial = part ? part : at<float>(ws, p->gemm_partial); g.partial_elems = p->gemm_partial_elems;
        g.zeros = at<char>(ws, p->misc + 64);
        return launch_gemm_tn_bf16(g, gs, defer);
    };
    auto gemm = [&](const void* A, int lda, int M, const void* Bm, int ldb, int N, int R, float* C, int shift_T, int trans_out,
                    float dp, uint32_t stack, int a_bf16 = 0, int b_bf16 = 0, ReduceJobs* defer = nullptr, float* part = nullptr) -> int {
        GemmTnArgs g;
        memset(&g, 0, sizeof(g));
        g.a_bf16 = a_bf16; g.b_bf16 = b_bf16;
        g.A = A; g.lda = lda; g.Bm = Bm; g.ldb = ldb; g.C = C; g.ldc = trans_out ? M : N; g.R = R; g.M = M; g.N = N;
        g.shift_T = shift_T; g.trans_out = trans_out; g.drop_p = dp; g.drop_seed = p->drop_seed; g.drop_stack = stack;
        g.partial = part ? part : at<float>(ws, p->gemm_partial); g.partial_elems = p->gemm_partial_elems;
        return launch_gemm_tn_f32(g, gs, defer);
    };

    // output layer (temporal_vae.py:70): dW_out[d][k] = sum_r d_recon[r][d] h[r][k]
    const void* h_top = at<char>(ws, p->dec[L - 1].h_seq);
    const bool pair_bwd = bf && !sp && L == 2 && pair_launch_enabled();
    ReduceJobs rjobs;                    // pair path: every reduction of a stack in ONE launch at the end of the stack
    rjobs.n = 0;
    {
        KernelTimer kt("dw_gemm.out", gs);
        if ((rc = launch_narrow_tn(h_top, bf ? 1 : 0, H, H, d_recon, D, bf ? ceil_div(B, SHM_WPT) * SHM_WPT * T : BT, grads + p->out_w, 0, grads + p->out_b,
                                   at<float>(ws, p->narrow_scratch2), gs, bf ? T : 0, bf ? B : 0, pair_bwd ? &rjobs : nullptr,
                                   sp ? at<char>(ws, p->dec[L - 1].h_lo) : nullptr))) return rc;
    }

    // split path: weight-gradient products waiting to be enqueued on the side stream (see run_stack_bwd)
    std::vector<std::function<int()>> deferred;
    // final_flush: the last layer of the backward (encoder layer 0) has no recurrence left to hide behind — its products go to the CALLER's
    // stream (own split-K scratch), so they fill the CUs the last recurrence has just vacated while the side stream is still finishing the
    // layer before (round 3: the step ended with the main stream idle for ~0.29 ms behind two serial side-stream GEMMs)
    bool final_flush = false;
    auto flush_deferred = [&]() -> int {
        for (auto& f : deferred) { const int r = f(); if (r) { deferred.clear(); return r; } }
        deferred.clear();
        return SHM_OK;
    };
    auto run_stack_bwd = [&](LayerInst* st, uint32_t stack, const float* x0, bool is_dec) -> int {
        const int Rt = ceil_div(B, SHM_WPT) * SHM_WPT * T;   // bf16 rows incl. tile padding (padding rows of dpre are exactly zero)
        auto bwd_args_bf = [&](int l, bf16_t* dps, float* tsum) {
            LstmBwdArgsB a;
            memset(&a, 0, sizeof(a));
            a.sv_g = at<bf16_t>(ws, st[l].saved);
            if (l == L - 1) {
                if (is_dec) { a.dy = d_recon; a.Wout = prm + p->out_w; a.Dy = D; }
                else a.dh_last = hb.dh_last;
            } else {
                a.dh_seq = dx_seq;
            }
            a.WhT = at<bf16_t>(ws, st[l].WhT);
            a.need_dx = l > 0 ? 1 : 0;
            a.WxT = l > 0 ? at<bf16_t>(ws, st[l].WxT) : nullptr;
            a.dpre_seq = dps; a.dpre_tsum = tsum; a.dx_seq = dx_seq;
            a.B = B; a.T = T; a.In = st[l].In;
            a.drop_p = l > 0 ? drop : 0.f; a.drop_seed = p->drop_seed; a.drop_stack = stack;
            return a;
        };
        // deferred weight / bias gradients of layer l from its stored dpre sequence
        auto layer_grads_bf = [&](int l, const void* dps, const float* tsum, ReduceJobs* defer = nullptr) -> int {
            int r;
            float* part = defer ? at<float>(ws, l > 0 ? p->gemm_partial : p->gemm_partial_b) : nullptr;
            if (l > 0) {      // with dropout the forward left the masked input in xdrop: no re-hashing in the GEMM
                const void* xin = drop > 0.f ? at<char>(ws, st[l].xdrop) : at<char>(ws, st[l - 1].h_seq);
                if ((r = gemm_bf(dps, G, xin, grads + st[l].w_ih, 0.f, stack, at<char>(ws, st[l].h_seq), grads + st[l].w_hh, Rt, defer, part))) return r;
            } else if (st[l].in_const && H == 128) {
                // constant-input layer at H = 128: dW_ih = sum_t dpre_t^T bf16(h0) is the first product of the SAME GEMM, its B
                // operand read as one row per window (the fp32 product over sum_t dpre stays as the fallback below)
                if ((r = gemm_bf(dps, G, at<char>(ws, p->h0_bf), grads + st[l].w_ih, -1.f, stack, at<char>(ws, st[l].h_seq), grads + st[l].w_hh, Rt,
                                 defer, part))) return r;
                goto bias;
            } else {
                // fp32-input layer at H = 128: dW_ih = dpre^T bf16(x) rides along as a 16-column third product (A is read once)
                const bool xfuse = !st[l].in_const && H == 128;
                if ((r = gemm_bf(dps, G, nullptr, nullptr, 0.f, 0, at<char>(ws, st[l].h_seq), grads + st[l].w_hh, Rt, defer, part,
                                 xfuse ? at<char>(ws, p->x_tb) : nullptr, xfuse ? grads + st[l].w_ih : nullptr))) return r;
                if (xfuse) goto bias;
            }
            if (l > 0) {
            } else if (st[l].in_const) {
                if ((r = gemm(tsum, G, G, hb.h0, H, H, B, grads + st[l].w_ih, 0, 0, 0.f, 0, 0, 0, defer, defer ? at<float>(ws, p->gemm_partial_c) : nullptr))) return r;
            } else {
                if ((r = launch_narrow_tn(dps, 1, G, G, x0, D, Rt, grads + st[l].w_ih, 1, nullptr, at<float>(ws, p->narrow_scratch), s, T, B, defer))) return r;
            }
        bias:
            // bias grads = column sums of sum_t dpre over the B windows: the same fixed-order reduction, batched or not
            if (defer) return reduce_jobs_add(*defer, tsum, B, (size_t)G, (size_t)G, grads + st[l].b_ih, grads + st[l].b_hh);
            return launch_reduce_partials(tsum, B, (size_t)G, grads + st[l].b_ih, grads + st[l].b_hh, s);
        };
        // weight / bias gradients of split-path layer l from its dpre planes, on gs once `ev` has fired
        auto defer_grads_sp = [&](int l, hipEvent_t ev) {
            const char* sn = stack == 0 ? "enc" : "dec";
            const int set = (int)stack * 2 + l;
            bf16_t* dph = at<bf16_t>(ws, p->dpre_set[set]); bf16_t* dpl = at<bf16_t>(ws, p->dpre_set_lo[set]);
            float* tsum = at<float>(ws, p->dpre_set_tsum[set]);
            const LayerInst* stp = st;
            deferred.push_back([=, &gemm_bf, &gemm, &gs, &final_flush]() -> int {
                int r2;
                const hipStream_t side_stream = gs;
                struct Restore { hipStream_t& g; hipStream_t v; ~Restore() { g = v; } } restore{gs, side_stream};
                const bool on_main = final_flush && gs != s;
                if (on_main) gs = s;                                   // gemm_bf / gemm launch on gs
                else if (ev) SHM_HIP_CHECK(hipStreamWaitEvent(gs, ev, 0));
                char gnm[32];
                snprintf(gnm, sizeof(gnm), "dw_gemm.%s%d", sn, l);
                KernelTimer kt(gnm, gs);
                LoPlanes lo;
                lo.A = dpl; lo.Bh = at<char>(ws, stp[l].h_lo); lo.Bin = nullptr; lo.Bx = nullptr;
                // every split-K / per-window reduction of this layer in ONE launch behind its products (was three or four small launches)
                ReduceJobs rj;
                rj.n = 0;
                float* part = at<float>(ws, on_main ? p->gemm_partial_b : p->gemm_partial);      // one layer at a time per stream: reduced below before the next product starts
                if (l > 0) {
                    const bool dr = drop > 0.f;
                    const void* xin = dr ? at<char>(ws, stp[l].xdrop) : at<char>(ws, stp[l - 1].h_seq);
                    lo.Bin = dr ? at<char>(ws, stp[l].xdrop_lo) : at<char>(ws, stp[l - 1].h_lo);
                    if ((r2 = gemm_bf(dph, G, xin, grads + stp[l].w_ih, 0.f, stack, at<char>(ws, stp[l].h_seq), grads + stp[l].w_hh, Rt, &rj, part,
                                      nullptr, nullptr, &lo))) return r2;
                } else if (stp[l].in_const) {
                    if ((r2 = gemm_bf(dph, G, nullptr, nullptr, 0.f, 0, at<char>(ws, stp[l].h_seq), grads + stp[l].w_hh, Rt, &rj, part,
                                      nullptr, nullptr, &lo))) return r2;
                    // dW_ih = (sum_t dpre_t)^T h0: B rows, exact fp32 (temporal_vae.py:68 repeats h0, so the T products collapse)
                    if ((r2 = gemm(tsum, G, G, hb.h0, H, H, B, grads + stp[l].w_ih, 0, 0, 0.f, 0, 0, 0, &rj, at<float>(ws, p->gemm_partial_c)))) return r2;
                } else {
                    const bool xfuse = H == 128;      // dW_ih = dpre^T x as a 16-column third product of the same launch
                    lo.Bx = at<char>(ws, p->x_tb_lo);
                    if ((r2 = gemm_bf(dph, G, nullptr, nullptr, 0.f, 0, at<char>(ws, stp[l].h_seq), grads + stp[l].w_hh, Rt, &rj, part,
                                      xfuse ? at<char>(ws, p->x_tb) : nullptr, xfuse ? grads + stp[l].w_ih : nullptr, &lo))) return r2;
                    if (!xfuse && (r2 = launch_narrow_tn(dph, 1, G, G, x0, D, Rt, grads + stp[l].w_ih, 1, nullptr, at<float>(ws, p->narrow_scratch), gs, T, B,
                                                         &rj, dpl))) return r2;
                }
                if ((r2 = reduce_jobs_add(rj, tsum, B, (size_t)G, (size_t)G, grads + stp[l].b_ih, grads + stp[l].b_hh))) return r2;
                return launch_reduce_batch(rj, gs);
            });
        };
        if (sp && NCb > 1) {
            // launch-level layer pipeline over NCb time chunks, walking down in time: s carries layer 1, the plan's second side stream
            // the dx product of a chunk and layer 0 on it; the weight-gradient products follow on gs once a layer's last chunk is done
            const char* sn = stack == 0 ? "enc" : "dec";
            hipStream_t ss = p->side2;
            const int qt = ceil_div(B, SHM_WPT), Tc = T / NCb;
            char nm[32];
            LstmBwdArgsS b0, b1;
            memset(&b0, 0, sizeof(b0)); memset(&b1, 0, sizeof(b1));
            for (int l = 0; l < 2; ++l) {
                LstmBwdArgsS& a = l ? b1 : b0;
                const int set = (int)stack * 2 + l;
                a.rec4 = at<float>(ws, st[l].saved); a.rec2 = at<float>(ws, st[l].saved_c);
                a.WhT_hi = at<bf16_t>(ws, st[l].WhT); a.WhT_lo = at<bf16_t>(ws, st[l].WhT_lo);
                a.dpre_hi = at<bf16_t>(ws, p->dpre_set[set]); a.dpre_lo = at<bf16_t>(ws, p->dpre_set_lo[set]);
                a.dpre_tsum = at<float>(ws, p->dpre_set_tsum[set]); a.B = B; a.T = T;
                a.state = at<float>(ws, st[l].b_state);
            }
            if (is_dec) { b1.dy = d_recon; b1.Wout = prm + p->out_w; b1.Dy = D; }
            else b1.dh_last = hb.dh_last;
            b0.dh_seq = dx_seq;
            DxArgsS da;
            memset(&da, 0, sizeof(da));
            da.dpre_hi = b1.dpre_hi; da.dpre_lo = b1.dpre_lo; da.WxT_hi = at<bf16_t>(ws, st[1].WxT); da.WxT_lo = at<bf16_t>(ws, st[1].WxT_lo);
            da.dx = dx_seq; da.rows = Rt; da.drop_p = drop; da.drop_seed = p->drop_seed; da.drop_stack = stack;
            da.cpt = Tc / 4; da.seq_tiles = T / 4; da.n_chunk_tiles = qt * da.cpt;
            // the side stream starts behind whatever s has done so far (head_bwd wrote dh_last; the previous stack's layer 0 read dx_seq)
            SHM_HIP_CHECK(hipEventRecord(p->ev_fork, s));
            SHM_HIP_CHECK(hipStreamWaitEvent(ss, p->ev_fork, 0));
            for (int ch = NCb - 1; ch >= 0; --ch) {
                int r;
                b0.t0 = b1.t0 = ch * Tc; b0.t1 = b1.t1 = (ch + 1) * Tc;
                {
                    snprintf(nm, sizeof(nm), "lstm_bwd.%s1", sn);
                    KernelTimer kt(nm, s);
                    if ((r = launch_lstm_bwd_sp(H, b1, s))) return r;
                }
                if (ch == NCb - 1 && (r = flush_deferred())) return r;      // the previous stack's products, behind this stack's first recurrence
                hipEvent_t ev = p->ev_chunk[(int)stack * 8 + ch];
                SHM_HIP_CHECK(hipEventRecord(ev, s));
                SHM_HIP_CHECK(hipStreamWaitEvent(ss, ev, 0));
                da.tile_off = b0.t0 / 4;
                {
                    snprintf(nm, sizeof(nm), "dx_gemm.%s1", sn);
                    KernelTimer kt(nm, ss);
                    if ((r = launch_dx_sp(H, da, ss))) return r;
                }
                snprintf(nm, sizeof(nm), "lstm_bwd.%s0", sn);
                KernelTimer kt(nm, ss);
                if ((r = launch_lstm_bwd_sp(H, b0, ss))) return r;
            }
            // ev_chunk[stack*8 + 0] fired after layer 1's last chunk; layer 0's last chunk ends the side stream's work
            defer_grads_sp(1, p->ev_chunk[(int)stack * 8]);
            hipEvent_t ev0 = p->ev_rec[(int)stack * 2];
            SHM_HIP_CHECK(hipEventRecord(ev0, ss));
            SHM_HIP_CHECK(hipStreamWaitEvent(s, ev0, 0));       // head_bwd / the caller read layer 0's sum_t dpre and nothing may overtake dx_seq
            defer_grads_sp(0, ev0);
            return SHM_OK;
        }
        if (sp) {
            const char* sn = stack == 0 ? "enc" : "dec";
            for (int l = L - 1; l >= 0; --l) {
                char nm[32];
                // one dpre plane set per (stack, layer): the side stream may still be reading a layer's dpre when later recurrences run
                const int set = (int)stack * 2 + l;
                bf16_t* dph = at<bf16_t>(ws, p->dpre_set[set]); bf16_t* dpl = at<bf16_t>(ws, p->dpre_set_lo[set]);
                float* tsum = at<float>(ws, p->dpre_set_tsum[set]);
                LstmBwdArgsS a;
                memset(&a, 0, sizeof(a));
                a.rec4 = at<float>(ws, st[l].saved); a.rec2 = at<float>(ws, st[l].saved_c);
                if (l == L - 1) {
                    if (is_dec) { a.dy = d_recon; a.Wout = prm + p->out_w; a.Dy = D; }
                    else a.dh_last = hb.dh_last;
                } else a.dh_seq = dx_seq;
                a.WhT_hi = at<bf16_t>(ws, st[l].WhT); a.WhT_lo = at<bf16_t>(ws, st[l].WhT_lo);
                a.dpre_hi = dph; a.dpre_lo = dpl; a.dpre_tsum = tsum; a.B = B; a.T = T;
                int r;
                DxArgsS da;
                memset(&da, 0, sizeof(da));
                if (l > 0) {
                    da.dpre_hi = dph; da.dpre_lo = dpl; da.WxT_hi = at<bf16_t>(ws, st[l].WxT); da.WxT_lo = at<bf16_t>(ws, st[l].WxT_lo);
                    da.dx = dx_seq; da.rows = Rt; da.drop_p = drop; da.drop_seed = p->drop_seed; da.drop_stack = stack;
                }
                // (r3) the top layer's recurrence and dx of the layer below in ONE launch (the dx workgroup of a tile follows the recurrence of
                // that tile): same-box A/B 2.12 -> 2.04 ms per step.  SHM_SP_BWD_DX_PAIR=0 or pair launches off: two launches, same bits.
                static const bool dx_pair = [] { const char* e = getenv("SHM_SP_BWD_DX_PAIR"); return !(e && e[0] == '0'); }();
                const bool fuse_dx = dx_pair && l > 0 && l == L - 1 && T % 4 == 0 && p->pair_sync && pair_launch_enabled();
                if (fuse_dx) {
                    LstmBwdDxPairArgsS pa;
                    memset(&pa, 0, sizeof(pa));
                    pa.top = a; pa.dx = da;
                    pa.sync = at<unsigned>(ws, p->pair_sync + (size_t)p->sync_next * lstm_fwd_pair_sync_bytes(B));
                    pa.sync_other = at<unsigned>(ws, p->pair_sync + (size_t)(p->sync_next ^ 1) * lstm_fwd_pair_sync_bytes(B));
                    p->sync_next ^= 1;
                    pa.err = at<unsigned>(ws, p->misc);
                    snprintf(nm, sizeof(nm), "lstm_bwd.%s%d", sn, l);
                    KernelTimer kt(nm, s);
                    r = launch_lstm_bwd_dx_pair_sp(H, pa, s);
                } else {
                    snprintf(nm, sizeof(nm), "lstm_bwd.%s%d", sn, l);
                    KernelTimer kt(nm, s);
                    r = launch_lstm_bwd_sp(H, a, s);
                }
                if (r) return r;
                // the recurrence is enqueued: NOW hand the previous layers' weight-gradient products to the side stream (a recurrent
                // launch holds 128 CUs; enqueued first it gets them, and the GEMMs fill what is left instead of the other way round)
                if ((r = flush_deferred())) return r;
                if (l > 0 && !fuse_dx) {  // gradient on the layer below's h_t, all steps at once
                    snprintf(nm, sizeof(nm), "dx_gemm.%s%d", sn, l);
                    KernelTimer kt(nm, s);
                    if ((r = launch_dx_sp(H, da, s))) return r;
                }
                hipEvent_t ev = nullptr;
                if (use_side) { ev = p->ev_rec[set]; SHM_HIP_CHECK(hipEventRecord(ev, s)); }
                defer_grads_sp(l, ev);
                if (!use_side && (r = flush_deferred())) return r;      // single stream: run it right away, as before
            }
            return SHM_OK;
        }
        if (bf && L == 2 && pair_launch_enabled()) {
            // both layers in one launch (layer 0 forms dx itself, a few steps behind layer 1); layer 0 keeps the primary
            // dpre / tsum buffers because the head reads decoder layer 0's sum_t dpre
            bf16_t* dps1 = at<bf16_t>(ws, p->dpre_seq2); float* tsum1 = at<float>(ws, p->dpre_tsum2);
            LstmBwdPairArgsB pa;
            pa.l1 = bwd_args_bf(1, dps1, tsum1); pa.l0 = bwd_args_bf(0, at<bf16_t>(ws, p->dpre_seq), dpre_tsum);
            pa.sync = at<unsigned>(ws, p->pair_sync + (size_t)p->sync_next * lstm_fwd_pair_sync_bytes(B));
            pa.sync_other = at<unsigned>(ws, p->pair_sync + (size_t)(p->sync_next ^ 1) * lstm_fwd_pair_sync_bytes(B));
            p->sync_next ^= 1;
            pa.err = at<unsigned>(ws, p->misc); pa.role_lds_off = 0;
            int r;
            {
                KernelTimer kt(stack == 0 ? "lstm_bwd.enc" : "lstm_bwd.dec", s);
                r = launch_lstm_bwd_pair_bf16(H, pa, s);
            }
            if (r) return r;
            {
                KernelTimer kt(stack == 0 ? "dw_gemm.enc1" : "dw_gemm.dec1", s);
                if ((r = layer_grads_bf(1, dps1, tsum1, &rjobs))) return r;
            }
            KernelTimer kt(stack == 0 ? "dw_gemm.enc0" : "dw_gemm.dec0", s);
            if ((r = layer_grads_bf(0, at<char>(ws, p->dpre_seq), dpre_tsum, &rjobs))) return r;
            return launch_reduce_batch(rjobs, s);
        }
        for (int l = L - 1; l >= 0; --l) {
            char nm[32], gnm[32];
            snprintf(nm, sizeof(nm), "lstm_bwd.%s%d", stack == 0 ? "enc" : "dec", l);
            snprintf(gnm, sizeof(gnm), "dw_gemm.%s%d", stack == 0 ? "enc" : "dec", l);
            if (bf) {
                const LstmBwdArgsB a = bwd_args_bf(l, at<bf16_t>(ws, p->dpre_seq), dpre_tsum);
                int r;
                {
                    KernelTimer kt(nm, s);
                    r = launch_lstm_bwd_bf16(H, a, s);
                }
                if (r) return r;
                if (l > 0) {
                    char dnm[32];
                    snprintf(dnm, sizeof(dnm), "dx_gemm.%s%d", stack == 0 ? "enc" : "dec", l);
                    KernelTimer kt(dnm, s);
                    if ((r = launch_dx_gemm_bf16(H, a, s))) return r;
                }
                KernelTimer kt(gnm, s);
                if ((r = layer_grads_bf(l, at<char>(ws, p->dpre_seq), dpre_tsum))) return r;
                continue;
            }
            LstmBwdArgs a;
            memset(&a, 0, sizeof(a));
            a.saved = at<float>(ws, st[l].saved);
            if (l == L - 1) {
                if (is_dec) { a.dy = d_recon; a.Wout = prm + p->out_w; a.Dy = D; }
                else a.dh_last = hb.dh_last;
            } else {
                a.dh_seq = dx_seq;
            }
            a.WhT = at<float>(ws, st[l].WhT);
            a.need_dx = l > 0 ? 1 : 0;
            a.WxT = l > 0 ? at<float>(ws, st[l].WxT) : nullptr;
            a.dpre_seq = dpre_seq; a.dpre_tsum = dpre_tsum; a.dx_seq = dx_seq;
            a.B = B; a.T = T; a.In = st[l].In;
            a.drop_p = l > 0 ? drop : 0.f; a.drop_seed = p->drop_seed; a.drop_stack = stack;
            int r;
            {
                KernelTimer kt(nm, s);
                r = launch_lstm_bwd_f32(H, a, s);
            }
            if (r) return r;
            KernelTimer kt(gnm, s);
            // deferred weight gradients of this layer
            if ((r = gemm(dpre_seq, G, G, at<float>(ws, st[l].h_seq), H, H, BT, grads + st[l].w_hh, T, 0, 0.f, 0))) return r;
            if (l > 0) {
                if ((r = gemm(dpre_seq, G, G, at<float>(ws, st[l - 1].h_seq), H, H, BT, grads + st[l].w_ih, 0, 0, drop, stack))) return r;
            } else if (st[l].in_const) {
                if ((r = gemm(dpre_tsum, G, G, hb.h0, H, H, B, grads + st[l].w_ih, 0, 0, 0.f, 0))) return r;
            } else {
                if ((r = launch_narrow_tn(dpre_seq, 0, G, G, x0, D, BT, grads + st[l].w_ih, 1, nullptr, at<float>(ws, p->narrow_scratch), s))) return r;
            }
            if ((r = launch_colsum(dpre_tsum, G, B, G, grads + st[l].b_ih, grads + st[l].b_hh, cs, s))) return r;
        }
        return SHM_OK;
    };

    if ((rc = run_stack_bwd(p->dec, 1, nullptr, true))) return rc;

    // head: dpre_tsum still holds sum_t dpre of decoder layer 0
    // (split path: decoder layer 0 wrote plane set 2)
    KernelTimer kt_head("head_bwd+grads", s);
    if ((rc = launch_head_bwd(B, H, Z, c.layer_norm, sp ? at<float>(ws, p->dpre_set_tsum[2]) : dpre_tsum, prm + p->dec[0].w_ih, at<float>(ws, p->eps), d_mu, d_logvar,
                              head_params(p, prm), hb, s))) return rc;
    if ((rc = launch_head_grads(B, H, Z, c.layer_norm, hb, grads + p->l2h_w, grads + p->l2h_b, grads + p->mu_w, grads + p->mu_b,
                                grads + p->lv_w, grads + p->lv_b, c.layer_norm ? grads + p->ln_w : nullptr,
                                c.layer_norm ? grads + p->ln_b : nullptr, s))) return rc;

    kt_head.~KernelTimer(); kt_head.idx = -1;
    if ((rc = run_stack_bwd(p->enc, 0, x, false))) return rc;
    final_flush = use_side && !getenv("SHM_SP_TAIL_ON_SIDE");
    if ((rc = flush_deferred())) return rc;
    if (use_side) {                                          // join: everything after this call on s sees the complete gradient bucket
        SHM_HIP_CHECK(hipEventRecord(p->ev_join, gs));
        SHM_HIP_CHECK(hipStreamWaitEvent(s, p->ev_join, 0));
    }
    return SHM_OK;
}

extern "C" int shm_window_gather(const float* series, int64_t n_rows, int32_t D, const int32_t* starts, int32_t N, int32_t T,
                                 const float* mean, const float* stdv, float clip_z, float* out, void* stream) {
    SHM_REQUIRE(series && ((out && starts) || N == 0), SHM_EINVAL, "window_gather: NULL series, starts or out");
    SHM_REQUIRE(n_rows >= 1 && D >= 1 && N >= 0 && T >= 1 && T <= n_rows, SHM_EINVAL, "window_gather: bad shape (rows=%lld D=%d N=%d T=%d)",
                (long long)n_rows, D, N, T);
    SHM_REQUIRE((mean == nullptr) == (stdv == nullptr), SHM_EINVAL, "window_gather: pass both mean and std or neither");
    return launch_window_gather(series, n_rows, D, starts, N, T, mean, stdv, clip_z, out, (hipStream_t)stream);
}

extern "C" int shm_profile_begin(void) {
    if (!g_prof) g_prof = new std::vector<ProfEntry>();
    for (auto& e : *g_prof) { (void)hipEventDestroy(e.e0); (void)hipEventDestroy(e.e1); }
    g_prof->clear();
    g_prof_on = true;
    return SHM_OK;
}

extern "C" int shm_profile_end(char* json, size_t cap) {
    g_prof_on = false;
    SHM_REQUIRE(g_prof && json && cap > 2, SHM_EINVAL, "profile_end: no profile or no buffer");
    SHM_HIP_CHECK(hipDeviceSynchronize());
    std::map<std::string, std::pair<int, double>> agg;
    for (auto& e : *g_prof) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, e.e0, e.e1) == hipSuccess) { auto& a = agg[e.name]; a.first += 1; a.second += ms; }
        (void)hipEventDestroy(e.e0); (void)hipEventDestroy(e.e1);
    }
    g_prof->clear();
    std::string out = "{";
    bool first = true;
    for (auto& kv : agg) {
        char buf[160];
        snprintf(buf, sizeof(buf), "%s\"%s\": {\"calls\": %d, \"ms\": %.6f}", first ? "" : ", ", kv.first.c_str(), kv.second.first, kv.second.second);
        out += buf; first = false;
    }
    out += "}";
    SHM_REQUIRE(out.size() + 1 <= cap, SHM_EINVAL, "profile_end: buffer too small (%zu needed)", out.size() + 1);
    memcpy(json, out.c_str(), out.size() + 1);
    return SHM_OK;
}

extern "C" void shm_dropout_keep_mask(uint64_t drop_seed, int32_t stack, int32_t B, int32_t T, int32_t H, float p, uint8_t* keep) {
    const uint32_t thr = drop_threshold(p);
    const size_t n = (size_t)B * T * H;
    for (size_t i = 0; i < n; ++i) keep[i] = drop_hash(drop_seed, (uint32_t)stack, (uint32_t)i) >= thr ? 1 : 0;
}

extern "C" void shm_dropout_keep_mask_tiled(uint64_t drop_seed, int32_t stack, int32_t B, int32_t T, int32_t H, float p, uint8_t* keep) {
    const uint32_t thr = drop_threshold(p);
    for (int b = 0; b < B; ++b)
        for (int t = 0; t < T; ++t) {
            const size_t row = ((size_t)(b / SHM_WPT) * T + t) * SHM_WPT + (b % SHM_WPT);   // tile-blocked row of (b, t)
            for (int k = 0; k < H; ++k)
                keep[((size_t)b * T + t) * H + k] = drop_hash(drop_seed, (uint32_t)stack, (uint32_t)(row * H + k)) >= thr ? 1 : 0;
        }
}
