// simulate.hip — synthetic 4-DOF structural-response generator on the device; gfx950.
// Reference: 4DOF/Scripts/utils/simulation_4dof.py — init_force :25-42 (Gaussian draw smoothed by a centred rolling mean),
//   run_simulation :80-137 (Newmark-beta, average acceleration by default, states clipped to +-1e5);
//   4DOF/Scripts/02_generate_fault_datasets.py — inject_noise / spikes / drift / bias :128-149 applied to one DOF's x, v, a columns
//   with a magnitude relative to the column's sample standard deviation :190-213.
//
// The recurrence is sequential in time and only nd x nd wide, so the parallel axis is the RUN: one thread integrates one run in fp64
// with its system matrices in registers; a workgroup is one wave of 64 runs.  Per step a thread produces 3 nd values of ITS run —
// written directly that would be a 64-way scatter — so 8 steps (4 for fp64 tables) are staged in LDS as [run][8 x 3 nd] and the wave then copies every
// run's 8-step block (768 B in fp32 at nd = 4) out with consecutive lanes on consecutive 16-byte pieces.  HBM-bound on the response table
// for large run counts (48 KB per run in fp32); latency-bound (1000 dependent steps) for a handful of runs.
// Compiled with -ffp-contract=off like the rest of the library, so products and sums round the way the numpy restatement does.
#include <math.h>

#include "kernels.hpp"

namespace shm {

namespace {

constexpr int SIM_RUNS = 64;        // runs per workgroup (one wave)
constexpr int sim_chunk(size_t elem) { return elem == 4 ? 8 : 4; }      // steps staged in LDS between copy-outs (<= 37 KB: >= 4 waves per CU)
constexpr double SIM_CLIP = 1e5;

__device__ __forceinline__ double clip5(double v) { return fmin(fmax(v, -SIM_CLIP), SIM_CLIP); }   // np.clip: NaN stays NaN
__device__ __forceinline__ void put(float* p, double v) { *p = (float)v; }
__device__ __forceinline__ void put(double* p, double v) { *p = v; }

// force[r][t][d] = fp32( mean over the existing rows of base[r][t - w/2 .. t + (w-1)/2][d] ), summed first row to last in fp64.
// One workgroup per (run, 128-step segment): the segment's rows plus the window halo go through LDS once (each base value is
// needed by `window` outputs), then every thread sums its window out of LDS — consecutive threads on consecutive (t, d).
constexpr int SMOOTH_SEG = 128;
__global__ void __launch_bounds__(256) force_smooth_kernel(const double* __restrict__ base, int steps, int nd, int window,
                                                           float* __restrict__ force) {
    extern __shared__ double tile[];                 // [(SMOOTH_SEG + window - 1)][nd]
    const int back = window / 2, fwd = (window - 1) / 2;
    const int t0 = blockIdx.y * SMOOTH_SEG, r = blockIdx.x;
    const int nseg = min(SMOOTH_SEG, steps - t0);
    const int row0 = t0 - back;                      // table row of tile row 0 (may be negative)
    const int nrows = nseg + back + fwd;
    const double* b = base + (size_t)r * steps * nd;
    for (int e = threadIdx.x; e < nrows * nd; e += 256) {
        const int row = row0 + e / nd;
        tile[e] = (row >= 0 && row < steps) ? b[(size_t)row * nd + (e % nd)] : 0.0;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < nseg * nd; e += 256) {
        const int t = t0 + e / nd, d = e % nd;
        const int lo = t - back < 0 ? 0 : t - back;
        const int hi = t + fwd >= steps ? steps - 1 : t + fwd;
        double s = 0.0;
        for (int u = lo; u <= hi; ++u) s += tile[(u - row0) * nd + d];
        force[((size_t)r * steps + t) * nd + d] = (float)(s / (double)(hi - lo + 1));
    }
}

struct NewmarkCoef { double c0, c1, c2, c3, c4, c5, dt, gamma; };

// sys[r] = { mass[ND] | C[ND][ND] | inverse of (c0 M + c1 C + K) [ND][ND] }, fp64.  force: [.. ][steps][ND] fp32, run r at
// force + r * force_stride (stride 0: every run shares one excitation, as the structural-fault cases do).
template <int ND, typename TO>
__global__ void __launch_bounds__(SIM_RUNS) newmark_kernel(const double* __restrict__ sys, const float* __restrict__ force,
                                                           long long force_stride, int R, int steps, NewmarkCoef k,
                                                           TO* __restrict__ out) {
    constexpr int NV = 3 * ND;                       // values per step
    constexpr int SIM_CH = sim_chunk(sizeof(TO));    // steps staged between copy-outs
    constexpr int PER = SIM_CH * NV;                 // values of one run in a full chunk: contiguous in the table
    constexpr int ROW = PER + 1;                     // +1: lanes step through LDS with an odd stride (conflict-free staging writes)
    constexpr int VW = (NV * sizeof(TO)) % 16 == 0 ? 16 / sizeof(TO) : 1;      // values per 16-byte store when every step row is 16-B aligned
    __shared__ TO stage[SIM_RUNS * ROW];             // staged in the output type: fp32 tables are rounded once, here
    const int lane = threadIdx.x;
    const int r = blockIdx.x * SIM_RUNS + lane;
    const int rr = r < R ? r : R - 1;                // idle lanes of the last workgroup shadow the last run and store nothing

    double m[ND], C[ND][ND], Ki[ND][ND];
    {
        const double* s = sys + (size_t)rr * (ND + 2 * ND * ND);
#pragma unroll
        for (int i = 0; i < ND; ++i) m[i] = s[i];
#pragma unroll
        for (int i = 0; i < ND; ++i)
#pragma unroll
            for (int j = 0; j < ND; ++j) { C[i][j] = s[ND + i * ND + j]; Ki[i][j] = s[ND + ND * ND + i * ND + j]; }
    }
    const float* f = force + (size_t)rr * force_stride;
    double x[ND], v[ND], a[ND];
#pragma unroll
    for (int i = 0; i < ND; ++i) {
        x[i] = 0.0; v[i] = 0.0;
        // a_0 = M^-1 (F_0 - C v_0 - K x_0) with x_0 = v_0 = 0, as the full product the reference forms: the zero off-diagonals of
        // M^-1 still spread a NaN / inf of F_0 to every row (0 x NaN), and nan_to_num then maps NaN -> 0, +-inf -> +-DBL_MAX
        double a0 = 0.0;
#pragma unroll
        for (int j = 0; j < ND; ++j) a0 += (j == i ? 1.0 / m[i] : 0.0) * (double)f[j];
        a[i] = a0 != a0 ? 0.0 : fmin(fmax(a0, -1.7976931348623157e308), 1.7976931348623157e308);
    }
    TO* mine = stage + lane * ROW;
    const int wg_runs = min(SIM_RUNS, R - blockIdx.x * SIM_RUNS);
    TO* out_wg = out + (size_t)blockIdx.x * SIM_RUNS * steps * NV;

    auto advance = [&](int t) {
        double p[ND], q[ND], rhs[ND], xn[ND];
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            p[i] = k.c0 * x[i] + k.c2 * v[i] + k.c3 * a[i];
            q[i] = k.c1 * x[i] + k.c4 * v[i] + k.c5 * a[i];
        }
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            double cq = 0.0;
#pragma unroll
            for (int j = 0; j < ND; ++j) cq += C[i][j] * q[j];
            rhs[i] = (double)f[(size_t)t * ND + i] + m[i] * p[i] + cq;
        }
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            double s = 0.0;
#pragma unroll
            for (int j = 0; j < ND; ++j) s += Ki[i][j] * rhs[j];
            xn[i] = s;
        }
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const double an = k.c0 * (xn[i] - x[i]) - k.c2 * v[i] - k.c3 * a[i];
            const double vn = v[i] + k.dt * ((1.0 - k.gamma) * a[i] + k.gamma * an);
            x[i] = clip5(xn[i]); v[i] = clip5(vn); a[i] = clip5(an);       // the clipped state is what the next step sees
        }
    };
    auto stash = [&](int u) {
#pragma unroll
        for (int i = 0; i < ND; ++i) { mine[u * NV + i] = (TO)x[i]; mine[u * NV + ND + i] = (TO)v[i]; mine[u * NV + 2 * ND + i] = (TO)a[i]; }
    };

    int t0 = 0;
    for (; t0 + SIM_CH <= steps; t0 += SIM_CH) {     // full chunks: compile-time extents, 16-byte stores
#pragma unroll
        for (int u = 0; u < SIM_CH; ++u) {
            if (t0 + u > 0) advance(t0 + u);
            stash(u);
        }
        __syncthreads();                             // one wave per workgroup: orders the LDS writes before the cross-lane reads
        // copy-out: run q's PER values are contiguous in the table at out[q][t0][0]; consecutive lanes take consecutive 16-byte pieces
        constexpr int NVEC = PER / VW;
        for (int i = lane; i < wg_runs * NVEC; i += SIM_RUNS) {
            const int q = i / NVEC, e = (i - q * NVEC) * VW;
            typedef TO vec_t __attribute__((ext_vector_type(VW)));
            vec_t val;
            if constexpr (VW == 1) val = stage[q * ROW + e];
            else {
#pragma unroll
                for (int w = 0; w < VW; ++w) val[w] = stage[q * ROW + e + w];
            }
            *reinterpret_cast<vec_t*>(out_wg + ((size_t)q * steps + t0) * NV + e) = val;
        }
        __syncthreads();
    }
    if (t0 < steps) {                                // ragged tail
        const int nst = steps - t0;
        for (int u = 0; u < nst; ++u) {
            if (t0 + u > 0) advance(t0 + u);
            stash(u);
        }
        __syncthreads();
        const int per = nst * NV;
        for (int i = lane; i < wg_runs * per; i += SIM_RUNS) {
            const int q = i / per, e = i - q * per;
            out_wg[((size_t)q * steps + t0) * NV + e] = stage[q * ROW + e];
        }
    }
}

// One workgroup per run: sample standard deviation of the three columns x_dof, v_dof, a_dof (two passes, fp64), then ONE pass that
// copies the run's table src -> dst (src may equal dst) with the fault added to those columns.  kind: 0 noise, 1 spikes, 2 drift, 3 bias.
template <typename T>
__global__ void __launch_bounds__(256) sensor_fault_kernel(const T* src, T* dst, int steps, int nd, int kind,
                                                           int dof, double rel_mag, const double* __restrict__ z,
                                                           const int* __restrict__ idx, int nz) {
    __shared__ double red[3][256];
    const int r = blockIdx.x, tid = threadIdx.x, ncols = 3 * nd, c0 = dof - 1;
    const T* in = src + (size_t)r * steps * ncols;
    T* out = dst + (size_t)r * steps * ncols;
    const bool inplace = (const T*)dst == src;
    double acc[3], mean[3], mag[3];
    auto block_sum3 = [&]() {                        // acc[] -> block totals in acc[], same fixed order for every thread
#pragma unroll
        for (int j = 0; j < 3; ++j) red[j][tid] = acc[j];
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (tid < s) {
#pragma unroll
                for (int j = 0; j < 3; ++j) red[j][tid] += red[j][tid + s];
            }
            __syncthreads();
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[j] = red[j][0];
        __syncthreads();
    };
#pragma unroll
    for (int j = 0; j < 3; ++j) acc[j] = 0.0;
    for (int t = tid; t < steps; t += 256) {
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[j] += (double)in[(size_t)t * ncols + c0 + j * nd];
    }
    block_sum3();
#pragma unroll
    for (int j = 0; j < 3; ++j) { mean[j] = acc[j] / (double)steps; acc[j] = 0.0; }
    for (int t = tid; t < steps; t += 256) {
#pragma unroll
        for (int j = 0; j < 3; ++j) { const double dv = (double)in[(size_t)t * ncols + c0 + j * nd] - mean[j]; acc[j] += dv * dv; }
    }
    block_sum3();
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const double sd = steps > 1 ? sqrt(acc[j] / (double)(steps - 1)) : NAN;    // pandas .std(): ddof = 1
        mag[j] = (sd > 0.0 ? sd : 1.0) * rel_mag;                                  // NaN > 0 is false, as in the reference
    }
    const double* zr = z ? z + (size_t)r * 3 * nz : nullptr;
    const int total = steps * ncols;
    for (int i = tid; i < total; i += 256) {
        const int t = i / ncols, c = i - t * ncols;
        const int jj = c - c0;
        const T val = in[i];
        if (jj >= 0 && jj % nd == 0 && kind != 1) {
            const int j = jj / nd;
            double add;
            if (kind == 0) add = mag[j] * zr[(size_t)j * nz + t];
            else if (kind == 2) add = (t == steps - 1 && steps > 1) ? mag[j] : (double)t * (steps > 1 ? mag[j] / (double)(steps - 1) : 0.0);   // np.linspace(0, mag, steps)
            else add = mag[j];
            put(out + i, (double)val + add);
        } else if (!inplace) {
            out[i] = val;
        }
    }
    if (kind == 1) {                                 // sparse: after the copy, read-modify-write the spike rows (distinct per column)
        __syncthreads();
        const int* ir = idx + (size_t)r * 3 * nz;
        for (int s = tid; s < 3 * nz; s += 256) {
            const int j = s / nz, t = ir[s];
            if (t >= 0 && t < steps) {
                T* p = out + (size_t)t * ncols + c0 + j * nd;
                put(p, (double)*p + (mag[j] + (mag[j] / 4.0) * zr[s]));
            }
        }
    }
}

template <int ND>
int launch_newmark(const double* sys, const float* force, long long force_stride, int R, int steps, const NewmarkCoef& k, void* out,
                   int out_f32, hipStream_t s) {
    const dim3 grid((R + SIM_RUNS - 1) / SIM_RUNS), block(SIM_RUNS);
    if (out_f32) SHM_LAUNCH((newmark_kernel<ND, float>), grid, block, 0, s, sys, force, force_stride, R, steps, k, (float*)out);
    else         SHM_LAUNCH((newmark_kernel<ND, double>), grid, block, 0, s, sys, force, force_stride, R, steps, k, (double*)out);
    SHM_LAUNCH_CHECK();
    return SHM_OK;
}

}  // namespace

}  // namespace shm

using namespace shm;

extern "C" int shm_sim_smooth_force(const double* base, int32_t R, int32_t steps, int32_t nd, int32_t window, float* force, void* stream) {
    SHM_REQUIRE(R >= 0 && steps >= 1 && nd >= 1 && window >= 1, SHM_EINVAL, "sim_smooth_force: bad shape (R=%d steps=%d nd=%d window=%d)",
                R, steps, nd, window);
    const size_t lds = (size_t)(SMOOTH_SEG + window - 1) * nd * sizeof(double);
    SHM_REQUIRE(lds <= 64 * 1024, SHM_EINVAL, "sim_smooth_force: window %d x %d channels exceeds the LDS tile (64 KB)", window, nd);
    SHM_REQUIRE((steps + SMOOTH_SEG - 1) / SMOOTH_SEG <= 65535, SHM_EINVAL, "sim_smooth_force: steps %d too large", steps);
    if (R == 0) return SHM_OK;
    SHM_REQUIRE(base && force, SHM_EINVAL, "sim_smooth_force: NULL base or force");
    SHM_LAUNCH(force_smooth_kernel, dim3(R, (steps + SMOOTH_SEG - 1) / SMOOTH_SEG), dim3(256), lds, (hipStream_t)stream,
                       base, steps, nd, window, force);
    SHM_LAUNCH_CHECK();
    return SHM_OK;
}

extern "C" int shm_sim_newmark(const double* sys, const float* force, int32_t shared_force, int32_t R, int32_t steps, int32_t nd,
                               double dt, double beta, double gamma, void* out, int32_t out_f32, void* stream) {
    SHM_REQUIRE(R >= 0 && steps >= 1, SHM_EINVAL, "sim_newmark: bad shape (R=%d steps=%d)", R, steps);
    SHM_REQUIRE(nd >= 1 && nd <= 6, SHM_EINVAL, "sim_newmark: num_dofs %d unsupported (1..6)", nd);
    SHM_REQUIRE(dt > 0.0 && beta > 0.0, SHM_EINVAL, "sim_newmark: dt %g and beta %g must be positive", dt, beta);
    if (R == 0) return SHM_OK;
    SHM_REQUIRE(sys && force && out, SHM_EINVAL, "sim_newmark: NULL sys, force or out");
    NewmarkCoef k;
    k.c0 = 1.0 / (beta * (dt * dt)); k.c1 = gamma / (beta * dt); k.c2 = 1.0 / (beta * dt);
    k.c3 = (1.0 / (2 * beta)) - 1.0; k.c4 = (gamma / beta) - 1.0; k.c5 = (dt / 2.0) * ((gamma / beta) - 2.0);
    k.dt = dt; k.gamma = gamma;
    const long long fs = shared_force ? 0 : (long long)steps * nd;
    hipStream_t s = (hipStream_t)stream;
    switch (nd) {
        case 1: return launch_newmark<1>(sys, force, fs, R, steps, k, out, out_f32, s);
        case 2: return launch_newmark<2>(sys, force, fs, R, steps, k, out, out_f32, s);
        case 3: return launch_newmark<3>(sys, force, fs, R, steps, k, out, out_f32, s);
        case 4: return launch_newmark<4>(sys, force, fs, R, steps, k, out, out_f32, s);
        case 5: return launch_newmark<5>(sys, force, fs, R, steps, k, out, out_f32, s);
        default: return launch_newmark<6>(sys, force, fs, R, steps, k, out, out_f32, s);
    }
}

extern "C" int shm_sim_sensor_fault(const void* src, void* dst, int32_t is_f32, int32_t R, int32_t steps, int32_t nd, int32_t kind,
                                    int32_t dof, double rel_mag, const double* z, const int32_t* idx, int32_t nz, void* stream) {
    SHM_REQUIRE(R >= 0 && steps >= 1 && nd >= 1, SHM_EINVAL, "sim_sensor_fault: bad shape (R=%d steps=%d nd=%d)", R, steps, nd);
    SHM_REQUIRE((long long)steps * 3 * nd < (1ll << 31), SHM_EINVAL, "sim_sensor_fault: a run's table exceeds 2^31 values");
    SHM_REQUIRE(kind >= 0 && kind <= 3, SHM_EINVAL, "sim_sensor_fault: kind %d unknown (0 noise, 1 spikes, 2 drift, 3 bias)", kind);
    SHM_REQUIRE(dof >= 1 && dof <= nd, SHM_EINVAL, "sim_sensor_fault: dof %d outside 1..%d", dof, nd);
    if (R == 0) return SHM_OK;
    SHM_REQUIRE(src && dst, SHM_EINVAL, "sim_sensor_fault: NULL src or dst");
    if (kind == 0) SHM_REQUIRE(z && nz == steps, SHM_EINVAL, "sim_sensor_fault: noise needs z [R][3][steps] (nz=%d, steps=%d)", nz, steps);
    if (kind == 1) SHM_REQUIRE(nz >= 0 && (nz == 0 || (z && idx)), SHM_EINVAL, "sim_sensor_fault: spikes need z and idx [R][3][nz]");
    if (kind != 1 && kind != 0) nz = 0;
    const dim3 grid(R), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (is_f32) SHM_LAUNCH((sensor_fault_kernel<float>), grid, block, 0, s, (const float*)src, (float*)dst, steps, nd, kind, dof, rel_mag, z, idx, nz);
    else        SHM_LAUNCH((sensor_fault_kernel<double>), grid, block, 0, s, (const double*)src, (double*)dst, steps, nd, kind, dof, rel_mag, z, idx, nz);
    SHM_LAUNCH_CHECK();
    return SHM_OK;
}
