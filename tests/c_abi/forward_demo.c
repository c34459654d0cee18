/* forward_demo.c — drives the C ABI of libshm_hip.so from plain C (no Python, no torch): the boundary a maintainer of the
 * reference would bind.  Reads a flat fp32 parameter buffer, a window batch and the reparameterisation noise from a file written
 * by tests/test_gpu_c_abi.py, runs shm_vae_pack_weights + shm_vae_forward on hipMalloc'd buffers, and writes recon | mu | logvar |
 * score back.  Build: gcc -D__HIP_PLATFORM_AMD__ forward_demo.c -I<repo>/include -I/opt/rocm/include -L<csrc> -lshm_hip -L/opt/rocm/lib -lamdhip64
 *
 * input file : int32 header {D, Z, H, L, layer_norm, precision, B, T, n_params} then float32 params[n_params], x[B*T*D], eps[B*Z]
 * output file: float32 recon[B*T*D], mu[B*Z], logvar[B*Z], score[B]
 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "shm_hip.h"

#define CHECK_HIP(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(r_)); return 2; } } while (0)
#define CHECK_SHM(e) do { int r_ = (e); if (r_ != 0) { fprintf(stderr, "%s failed (%d): %s\n", #e, r_, shm_last_error()); return 3; } } while (0)

int main(int argc, char** argv) {
    if (argc != 3) { fprintf(stderr, "usage: %s <in.bin> <out.bin>\n", argv[0]); return 1; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 1; }
    int32_t hd[9];
    if (fread(hd, sizeof(int32_t), 9, f) != 9) return 1;
    const int D = hd[0], Z = hd[1], H = hd[2], L = hd[3], B = hd[6], T = hd[7];
    const int64_t n_params = hd[8];
    shm_vae_config cfg = {D, Z, H, L, hd[4], hd[5], 0.3f};
    if (shm_vae_param_count(&cfg) != n_params) { fprintf(stderr, "parameter count mismatch: %lld vs %lld\n", (long long)shm_vae_param_count(&cfg), (long long)n_params); return 1; }
    const size_t nx = (size_t)B * T * D, nz = (size_t)B * Z;
    float* host = (float*)malloc(sizeof(float) * (n_params + nx + nz));
    if (fread(host, sizeof(float), n_params + nx + nz, f) != (size_t)(n_params + nx + nz)) return 1;
    fclose(f);

    shm_vae_plan* plan = NULL;
    CHECK_SHM(shm_vae_plan_create(&cfg, B, T, /*training=*/0, &plan));
    float *params, *x, *eps, *recon, *mu, *lv, *score;
    void* ws;
    CHECK_HIP(hipMalloc((void**)&params, sizeof(float) * n_params));
    CHECK_HIP(hipMalloc((void**)&x, sizeof(float) * nx));
    CHECK_HIP(hipMalloc((void**)&eps, sizeof(float) * nz));
    CHECK_HIP(hipMalloc((void**)&recon, sizeof(float) * nx));
    CHECK_HIP(hipMalloc((void**)&mu, sizeof(float) * nz));
    CHECK_HIP(hipMalloc((void**)&lv, sizeof(float) * nz));
    CHECK_HIP(hipMalloc((void**)&score, sizeof(float) * B));
    CHECK_HIP(hipMalloc(&ws, shm_vae_plan_workspace_bytes(plan)));
    CHECK_HIP(hipMemcpy(params, host, sizeof(float) * n_params, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(x, host + n_params, sizeof(float) * nx, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(eps, host + n_params + nx, sizeof(float) * nz, hipMemcpyHostToDevice));

    hipStream_t s;
    CHECK_HIP(hipStreamCreate(&s));
    CHECK_SHM(shm_vae_pack_weights(plan, params, ws, s));
    CHECK_SHM(shm_vae_forward(plan, x, eps, recon, mu, lv, score, /*train_mode=*/0, /*drop_seed=*/0, ws, s));
    CHECK_SHM(shm_vae_plan_status(plan, ws, s));          /* synchronises the stream */

    float* out = (float*)malloc(sizeof(float) * (nx + 2 * nz + B));
    CHECK_HIP(hipMemcpy(out, recon, sizeof(float) * nx, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(out + nx, mu, sizeof(float) * nz, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(out + nx + nz, lv, sizeof(float) * nz, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(out + nx + 2 * nz, score, sizeof(float) * B, hipMemcpyDeviceToHost));
    f = fopen(argv[2], "wb");
    if (!f) { perror(argv[2]); return 1; }
    fwrite(out, sizeof(float), nx + 2 * nz + B, f);
    fclose(f);
    shm_vae_plan_destroy(plan);
    printf("forward_demo ok: B=%d T=%d D=%d H=%d L=%d precision=%d score[0]=%g\n", B, T, D, H, L, hd[5], out[nx + 2 * nz]);
    return 0;
}
