/*
 * shm_hip.h — C ABI of the MI355X (gfx950) LSTM-VAE / CNN hot path.
 *
 * The reference (Ogunleyemma1/Hybrid-VAE-CNN-for-SHM) has no FFI layer: its boundary for this path is
 * the Python module API of Scripts/Models.  Each entry point below replaces the PyTorch
 * arithmetic behind one of those reference call sites; the Python mirror in
 * hybrid-vae-cnn-for-shm_amd/Scripts/Models binds them through ctypes (INTEGRATION.md shows the stub).
 *
 * Conventions
 *   - plain C types only: device pointers, sizes, a hipStream_t passed as void*;
 *   - every buffer is owned by the caller (torch tensors in the Python mirror); the library allocates
 *     no device memory — a plan is a host-side description of offsets into a caller-supplied workspace;
 *   - calls are asynchronous on the given stream; one host thread per plan;
 *   - return value 0 = ok, otherwise a negative SHM_E* code or a positive hipError_t;
 *     shm_last_error() gives the text for the calling thread.  No C++ exception crosses the ABI.
 *   - parameters travel as ONE flat fp32 buffer in the reference's state_dict order
 *     (4DOF/Scripts/Models/temporal_vae.py:28-49; shm_vae_param_layout() enumerates it), gradients
 *     and Adam moments as flat buffers of the same layout.
 */
#ifndef SHM_HIP_H
#define SHM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SHM_OK            0
#define SHM_EINVAL       -1   /* bad argument / unsupported shape */
#define SHM_EWORKSPACE   -2   /* workspace too small */
#define SHM_ESTATE       -3   /* call order violated (e.g. backward without a training forward) */

/* SHM_PREC_SPLIT: every MFMA operand is a pair of bf16 values hi + lo (16 significant bits), products hi*hi + hi*lo + lo*hi in fp32:
 * holds the reference's fp32 results to ~1e-6 (the 1e-4 parity bar) at bf16 matrix rates — the fast parity path.  Since round 3 the
 * forward input projections W_ih x of layers whose input is a hidden SEQUENCE (layers >= 1) use W_ih ~ fp16(W) + e4m3((W - fp16 W) 2^15) 2^-15 (15
 * significant bits, measured at the same 5e-6 level) with x as an fp16 pair + e4m3: that W_ih fits a CU beside W_hh, which lets both
 * layers of a stack run as one layer-pipelined launch. */
enum { SHM_PREC_F32 = 0, SHM_PREC_BF16 = 1, SHM_PREC_SPLIT = 2 };

/* Constructor arguments of TemporalVAE (4DOF/Scripts/Models/temporal_vae.py:14-21; the 1_DOF variant
 * 1_DOF/Scripts/Models/temporal_vae.py:11 has no LayerNorm -> layer_norm = 0). */
typedef struct shm_vae_config {
    int32_t input_dim;     /* D */
    int32_t latent_dim;    /* Z  (<= 64) */
    int32_t hidden_dim;    /* H  in {32, 64, 128} */
    int32_t num_layers;    /* L  in {1, 2} */
    int32_t layer_norm;    /* 1 = nn.LayerNorm on h_n[-1] (eps 1e-5, biased variance) */
    int32_t precision;     /* SHM_PREC_F32: exact-fp32 MFMA; SHM_PREC_BF16: bf16 MFMA, fp32 accumulate/state; SHM_PREC_SPLIT: bf16 hi+lo pairs */
    float   dropout;       /* nn.LSTM inter-layer dropout p (train mode, L > 1 only) */
} shm_vae_config;

typedef struct shm_tensor_desc {
    char    name[48];      /* state_dict key, e.g. "encoder_lstm.weight_ih_l0" */
    int64_t offset;        /* element offset into the flat fp32 buffer */
    int32_t ndim;
    int32_t shape[4];
} shm_tensor_desc;

typedef struct shm_vae_plan shm_vae_plan;   /* opaque, host memory only */

const char* shm_last_error(void);
int  shm_version(void);

/* ---- parameter layout (replaces nn.Module.state_dict() ordering, temporal_vae.py:28-49) ---- */
int64_t shm_vae_param_count(const shm_vae_config* cfg);
/* writes up to max descriptors, returns the number of parameter tensors (26 for L=2 with LayerNorm) */
int  shm_vae_param_layout(const shm_vae_config* cfg, shm_tensor_desc* out, int max);

/* ---- plan: fixes (config, batch B, window length T) and lays out the workspace ---- */
int  shm_vae_plan_create(const shm_vae_config* cfg, int32_t B, int32_t T, int32_t training, shm_vae_plan** out);
void shm_vae_plan_destroy(shm_vae_plan* plan);
size_t shm_vae_plan_workspace_bytes(const shm_vae_plan* plan);

/* Re-pack the flat fp32 parameters into MFMA fragment order inside the workspace.  Must run after every
 * parameter change and before forward; it also keeps a flat copy for the head / output layer, so forward and
 * backward read parameters only from the workspace. */
int  shm_vae_pack_weights(shm_vae_plan* plan, const float* params, void* workspace, void* stream);

/* TemporalVAE.forward (temporal_vae.py:72-77) + per-window score ((x-recon)^2).mean(dim=(1,2)),
 * 04_vae_thresholding.py:122.  eps[B,Z] is the reparameterisation noise (temporal_vae.py:62) supplied by
 * the caller.  recon/mu/logvar/score may each be NULL.  train_mode != 0 applies inter-layer dropout with
 * the counter RNG keyed by drop_seed and, if the plan is a training plan, saves the state backward needs. */
int  shm_vae_forward(shm_vae_plan* plan, const float* x, const float* eps,
                     float* recon, float* mu, float* logvar, float* score,
                     int32_t train_mode, uint64_t drop_seed, void* workspace, void* stream);

/* The same forward with the reparameterisation noise drawn ON THE DEVICE inside the head kernel (no eps buffer, no separate RNG launch):
 * eps[b][z] = N(0,1) of the counter RNG (Philox4x32-10 + Box-Muller) keyed by (eps_seed, eps_offset, element b*Z + z) — the production
 * form of torch.randn_like at temporal_vae.py:62 (same distribution, its own stream; a trainer passes offset = step).  The noise used is
 * kept in the workspace for shm_vae_backward.  shm_eps_normal is the host replica of that stream (tests rebuild the draw with it). */
int  shm_vae_forward_sample(shm_vae_plan* plan, const float* x, uint64_t eps_seed, uint64_t eps_offset,
                            float* recon, float* mu, float* logvar, float* score,
                            int32_t train_mode, uint64_t drop_seed, void* workspace, void* stream);
void shm_eps_normal(uint64_t eps_seed, uint64_t eps_offset, int64_t n, float* out);

/* TemporalVAE.encode (temporal_vae.py:51-58): x[B,T,D] -> mu, logvar [B,Z] (either may be NULL); the decoder does not run. */
int  shm_vae_encode(shm_vae_plan* plan, const float* x, float* mu, float* logvar,
                    int32_t train_mode, uint64_t drop_seed, void* workspace, void* stream);
/* TemporalVAE.decode (temporal_vae.py:65-70): z[B,Z] -> recon[B,T,D] with T = the plan's window length (the reference's seq_len
 * argument); the encoder does not run.  Neither call leaves state for shm_vae_backward. */
int  shm_vae_decode(shm_vae_plan* plan, const float* z, float* recon,
                    int32_t train_mode, uint64_t drop_seed, void* workspace, void* stream);

/* On-device windowing + normalisation: out[n][t][d] = nan_to_num((series[starts[n]+t][d] - mean[d]) / std[d]),
 * i.e. make_windows + normalize of 4DOF/Scripts/03_train_vae.py:98-117 (dup. 04:73-110, 05:76-115, 06:91-126) without the T-fold
 * host copy: series [n_rows, D] fp32 and starts [N] int32 are device buffers; every starts[n] must satisfy
 * 0 <= starts[n] <= n_rows - T (the caller validates; the kernel clamps rather than read out of bounds).  mean/std [D] may both be NULL.
 * clip_z > 0 adds the openLAB variant's np.clip(z, -clip_z, clip_z) between the z-score and the NaN replacement
 * (20250506_openLAB_tests/Codes/04_train_vae.py:113-117: +-inf then becomes +-clip_z, only NaN becomes 0); clip_z <= 0: no clip. */
int  shm_window_gather(const float* series, int64_t n_rows, int32_t D, const int32_t* starts, int32_t N, int32_t T,
                       const float* mean, const float* std, float clip_z, float* out, void* stream);

/* Health of the layer-pipelined launches (bf16 and split paths, 2 layers: both LSTM layers of a stack run in ONE launch, the upper
 * layer consuming the lower layer's h_t a few steps behind through in-launch flags; split path also: dx of the layer below formed
 * behind the top layer's backward recurrence inside its launch).  Synchronises the stream.  SHM_OK, or
 * SHM_ESTATE if a consumer workgroup ever gave up waiting (the word is sticky: every result since its last reset is then invalid).
 * No reference counterpart: nn.LSTM (temporal_vae.py:32,44) hides its layer loop. */
int  shm_vae_plan_status(const shm_vae_plan* plan, const void* workspace, void* stream);
/* The same word without a host synchronisation: copies the plan's sticky status word (0 = healthy) into dst (device, 4 bytes) on the
 * stream, so a trainer can carry it in the statistics buffer it reads back anyway.  The word is cleared by the FIRST shm_vae_pack_weights
 * on a plan (the workspace arrives uninitialised) and by shm_vae_plan_reset_status only: re-packing the weights after an optimiser step
 * does not clear it.  shm_vae_plan_status_offset: byte offset of the word inside the workspace (-1: this plan has no such word). */
int  shm_vae_plan_status_word(const shm_vae_plan* plan, const void* workspace, uint32_t* dst, void* stream);
int64_t shm_vae_plan_status_offset(const shm_vae_plan* plan);
int  shm_vae_plan_reset_status(shm_vae_plan* plan, void* workspace, void* stream);
/* Measurement / test knob: 0 = one launch per LSTM layer, 1 = layer-pipelined launches (default; also SHM_PAIR_LAUNCH=0 in
 * the environment).  Returns the previous setting.  Both settings give bit-identical results. */
int  shm_set_pair_launch(int on);

/* Loss of 03_train_vae.py:264-266: recon = mean((recon-x)^2), kl = -0.5*mean(1+lv-mu^2-exp(lv)),
 * loss = recon + kl_w*kl.  Writes losses[0..2] = {loss, recon, kl} (device, fp32) and leaves
 * dLoss/d(recon,mu,logvar) in the workspace for shm_vae_backward(..., NULL, NULL, NULL, ...). */
int  shm_vae_loss(shm_vae_plan* plan, const float* x, float kl_w, float* losses, void* workspace, void* stream);

/* Validation-pass loss (03_train_vae.py:284-299, no state, no gradients): from the outputs of an inference forward — mu, logvar [B,Z]
 * and the per-window score [B] — writes losses[0..2] = {recon + kl_w*kl, recon = mean(score), kl}.  T, D only document the shape. */
int  shm_vae_eval_loss(const float* mu, const float* logvar, const float* score, int32_t B, int32_t T, int32_t D, int32_t Z,
                       float kl_w, float* losses, void* stream);

/* loss.backward() (03_train_vae.py:269): BPTT through decoder, head and encoder.  d_recon[B,T,D],
 * d_mu[B,Z], d_logvar[B,Z] are the upstream gradients (all NULL: use those left by shm_vae_loss).
 * grads: flat fp32 buffer, overwritten (zero_grad(set_to_none=True) semantics). */
int  shm_vae_backward(shm_vae_plan* plan, const float* x,
                      const float* d_recon, const float* d_mu, const float* d_logvar,
                      float* grads, void* workspace, void* stream);

/* clip_grad_norm_(max_norm) + torch.optim.Adam step (03_train_vae.py:222,270-271) over flat buffers.
 * grad_scale multiplies the gradients first (1/world_size after a sum all-reduce).  max_norm <= 0: no clipping.
 * weight decay is L2-coupled (g += wd*p).  step is 1-based.  scratch: >= shm_adam_scratch_bytes(n) device bytes.
 * out_norm (device, 1 float, may be NULL) receives the pre-clip total norm; with max_norm <= 0 AND out_norm NULL the norm is not
 * computed at all (one launch less: the CNN step of 05_train_cnn.py:270-276 neither clips nor logs it). */
size_t shm_adam_scratch_bytes(int64_t n);
int  shm_adam_step(int64_t n, float* params, float* grads, float* exp_avg, float* exp_avg_sq,
                   int32_t step, float lr, float beta1, float beta2, float eps, float weight_decay,
                   float max_norm, float grad_scale, void* scratch, float* out_norm, void* stream);
/* The same step; its last launch also copies one 32-bit status word: *status_dst = status_src ? *status_src : 0 (both device
 * pointers, status_dst may be NULL).  A trainer passes workspace + shm_vae_plan_status_offset(plan) and a slot of the statistics
 * buffer it reads back anyway: the sticky hand-off word then reaches the host every step without a copy launch of its own. */
int  shm_adam_step_status(int64_t n, float* params, float* grads, float* exp_avg, float* exp_avg_sq,
                          int32_t step, float lr, float beta1, float beta2, float eps, float weight_decay,
                          float max_norm, float grad_scale, void* scratch, float* out_norm,
                          const uint32_t* status_src, uint32_t* status_dst, void* stream);

/* ======================================================================================================
 * CNN fault classifier (reference 4DOF/Scripts/Models/cnn_model.py:14-51): Conv2d(C,16,3,p1)+BatchNorm2d+ReLU+MaxPool2d(2),
 * Conv2d(16,32,3,p1)+BatchNorm2d+ReLU+MaxPool2d(2), Flatten, Linear(F,128)+ReLU+Dropout, Linear(128,classes);
 * F = 32*(T/4)*(D/4) (cnn_model.py:30 hard-codes 32*25*3 for T=100, D=12).  fp32 throughout.
 * Parameters: flat fp32 buffer in state_dict order of the 12 learnable tensors (shm_cnn_param_layout);
 * BatchNorm buffers travel separately: running[0:16]=conv1 mean, [16:32]=conv1 var, [32:64]=conv2 mean, [64:96]=conv2 var. */
typedef struct shm_cnn_config {
    int32_t in_channels;   /* 2: window + squared VAE residual (05_train_cnn.py:137-138) */
    int32_t num_classes;   /* 2 */
    int32_t seq_len;       /* T  (cnn_model.py:4 SEQ_LEN) */
    int32_t num_features;  /* D  (cnn_model.py:5 NUM_FEATURES) */
    float   dropout;       /* fc1 Dropout p (train mode) */
} shm_cnn_config;
typedef struct shm_cnn_plan shm_cnn_plan;

int64_t shm_cnn_param_count(const shm_cnn_config* cfg);
int  shm_cnn_param_layout(const shm_cnn_config* cfg, shm_tensor_desc* out, int max);
int  shm_cnn_plan_create(const shm_cnn_config* cfg, int32_t B, int32_t training, shm_cnn_plan** out);
void shm_cnn_plan_destroy(shm_cnn_plan* plan);
size_t shm_cnn_plan_workspace_bytes(const shm_cnn_plan* plan);

/* CNN.forward (cnn_model.py:45-51).  x[B,C,T,D] NCHW -> logits[B,classes].  train_mode != 0: BatchNorm uses batch
 * statistics and updates `running` (momentum 0.1, unbiased variance), Dropout is active (counter RNG, drop_seed) and, on a
 * training plan, the state backward needs is kept.  train_mode == 0: running statistics, no dropout (model.eval()). */
int  shm_cnn_forward(shm_cnn_plan* plan, const float* params, float* running, const float* x, float* logits,
                     int32_t train_mode, uint64_t drop_seed, void* workspace, void* stream);
/* nn.CrossEntropyLoss() (05_train_cnn.py:257,274): loss (device, 1 float) and dLoss/dlogits left in the workspace. */
int  shm_cnn_loss(shm_cnn_plan* plan, const int64_t* labels, float* loss, void* workspace, void* stream);
/* Validation CrossEntropy (05_train_cnn.py:284-292): mean over the batch of -log softmax(logits)[label]; stateless, no gradient. */
int  shm_cnn_eval_loss(const float* logits, const int64_t* labels, int32_t B, int32_t num_classes, float* loss, void* stream);
/* loss.backward() (05_train_cnn.py:275).  d_logits NULL: use the one left by shm_cnn_loss.  grads: flat fp32, overwritten. */
int  shm_cnn_backward(shm_cnn_plan* plan, const float* params, const float* x, const float* d_logits, float* grads,
                      void* workspace, void* stream);
/* xin = stack([z, (z - zhat)^2], dim=1)  (05_train_cnn.py:137-138, 06_test_full_pipeline.py:364-365); z, zhat [B,T,D]. */
int  shm_cnn_build_input(int32_t B, int32_t T, int32_t D, const float* z, const float* zhat, float* xin, void* stream);
/* argmax(logits, 1) and softmax(logits, 1)[:, 1]  (06_test_full_pipeline.py:367-371); either output may be NULL. */
int  shm_cnn_predict(int32_t B, int32_t num_classes, const float* logits, int64_t* cls, float* p1, void* stream);

/* Per-kernel timing with HIP events recorded on the launch stream (used by bench.py for the roofline line).
 * shm_profile_begin() arms it for the calling thread; shm_profile_end() synchronises the device and writes a JSON
 * object {"lstm_fwd.enc0": {"calls": n, "ms": total}, ...} into json[cap]. */
int  shm_profile_begin(void);
int  shm_profile_end(char* json, size_t cap);

/* Host-side replica of the dropout counter RNG (so tests can rebuild the masks the kernels used).
 * keep[i] = 1 if element (b,t,k) of the layer-0 output of stack `stack` (0 = encoder, 1 = decoder) is kept. */
void shm_dropout_keep_mask(uint64_t drop_seed, int32_t stack, int32_t B, int32_t T, int32_t H, float p, uint8_t* keep);
/* Same for SHM_PREC_BF16 plans, whose sequences live in 16-window tile-blocked row order (the RNG is indexed by the
 * element's position in that layout); keep[] is still written in natural [B,T,H] order. */
void shm_dropout_keep_mask_tiled(uint64_t drop_seed, int32_t stack, int32_t B, int32_t T, int32_t H, float p, uint8_t* keep);

/* ---- openLAB fault classifier, inference forward (reference 20250506_openLAB_tests/Codes/Models/cnn_model.py:14-62:
 * 4 x [Conv2d(kt x 3) -> GroupNorm(8) -> SiLU] with MaxPool(2,1) between and AdaptiveAvgPool at the end, Linear(256,128)-SiLU-
 * Dropout-Linear(128,classes)); eval mode only (Dropout = identity), used by the hybrid VAE-threshold -> CNN inference of
 * 20250506_openLAB_tests/Codes/10_test_hybrid_pipeline.py:266-302.  Stateless: params is the flat fp32 state_dict in
 * shm_olcnn_param_layout order (features.{0,2,4,6}.{0,1}.{weight,bias}, classifier.{1,4}.{weight,bias}), x is [B, in_channels,
 * seq_len, num_features] fp32, logits [B, num_classes]; workspace: shm_olcnn_workspace_bytes, 256-byte aligned, caller-owned. */
typedef struct shm_olcnn_config {
    int32_t in_channels;   /* 1 in the reference */
    int32_t num_classes;   /* 2 */
    int32_t seq_len;       /* T: 200 in the reference, any length >= 8 (the conv stack is size-agnostic) */
    int32_t num_features;  /* W: 4 */
} shm_olcnn_config;
int64_t shm_olcnn_param_count(const shm_olcnn_config* cfg);
int     shm_olcnn_param_layout(const shm_olcnn_config* cfg, shm_tensor_desc* out, int max);
size_t  shm_olcnn_workspace_bytes(const shm_olcnn_config* cfg, int32_t B);
int     shm_olcnn_forward(const shm_olcnn_config* cfg, const float* params, const float* x, float* logits, int32_t B,
                          void* workspace, void* stream);

/* ---- synthetic data generators on the device --------------------------------------------------------------
 * All buffers are device memory owned by the caller; calls are asynchronous on `stream`.
 *
 * shm_sim_smooth_force: the smoothing half of init_force (4DOF/Scripts/utils/simulation_4dof.py:31-42): base [R][steps][nd] fp64 (the
 * caller's Gaussian draw x rms) -> force [R][steps][nd] fp32 = centred rolling mean over `window` rows with min_periods = 1
 * (rows t - window/2 .. t + (window-1)/2 that exist); the reference uses window = max(1, int(0.5 / dt)).
 *
 * shm_sim_newmark: run_simulation (simulation_4dof.py:80-137) for R independent runs, zero initial displacement / velocity.
 * sys [R][nd + 2 nd^2] fp64 per run = { masses | damping matrix C row-major | inverse of (a0 M + a1 C + K) row-major } with
 * a0 = 1/(beta dt^2), a1 = gamma/(beta dt) — the host side forms them (compute_matrices, :45-77, is an nd x nd eigenproblem per run).
 * force fp32 [R][steps][nd], or [steps][nd] shared by every run if shared_force != 0 (the structural-fault cases,
 * 02_generate_fault_datasets.py:152-170).  out [R][steps][3 nd] = x_1..x_nd, v_1..v_nd, a_1..a_nd per step (the reference's
 * DataFrame columns), fp64, or fp32 if out_f32 != 0; states are clipped to +-1e5 as in :131-133.  nd 1..6.
 *
 * shm_sim_sensor_fault: generate_sensor_faults (02_generate_fault_datasets.py:128-149,190-213): dst = src [R][steps][3 nd] (fp32 if
 * is_f32 != 0, else fp64; dst may be src for an in-place update, otherwise the buffers must not overlap) with the columns x_dof, v_dof,
 * a_dof (dof 1-based) of every run corrupted:
 *   kind 0 noise  + mag * z[r][j][t]                      (z standard normal, [R][3][nz = steps])
 *   kind 1 spikes + mag + mag/4 * z[r][j][s] at row idx[r][j][s]   ([R][3][nz], positions distinct per column)
 *   kind 2 drift  + linspace(0, mag, steps)[t]
 *   kind 3 bias   + mag
 * with mag = rel_mag * sample standard deviation (ddof 1) of the uncorrupted column (1.0 if that is 0). */
int shm_sim_smooth_force(const double* base, int32_t R, int32_t steps, int32_t nd, int32_t window, float* force, void* stream);
int shm_sim_newmark(const double* sys, const float* force, int32_t shared_force, int32_t R, int32_t steps, int32_t nd,
                    double dt, double beta, double gamma, void* out, int32_t out_f32, void* stream);
int shm_sim_sensor_fault(const void* src, void* dst, int32_t is_f32, int32_t R, int32_t steps, int32_t nd, int32_t kind, int32_t dof,
                         double rel_mag, const double* z, const int32_t* idx, int32_t nz, void* stream);

/* ---- measured ceilings of the box (measurement only; SURVEY §8d asks for the measured peak beside the nominal one) ------------------
 * shm_peak_mfma: every wave of 256 * blocks_per_cu four-wave workgroups issues iters x 8 independent MFMAs (bf16 != 0:
 * v_mfma_f32_16x16x32_bf16, else v_mfma_f32_16x16x4_f32) with register operands; *tflops = the rate of the second of two passes.
 * shm_peak_hbm: one pass over `bytes` with 16-byte accesses — mode 0 read, 1 write, 2 copy (read + write counted); *gbps likewise.
 * scratch: one device float.  Both synchronise the stream. */
int shm_peak_mfma(int32_t bf16, int32_t blocks_per_cu, int32_t iters, float* scratch, double* tflops, void* stream);
int shm_peak_hbm(const void* src, void* dst, size_t bytes, int32_t mode, float* scratch, double* gbps, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SHM_HIP_H */
