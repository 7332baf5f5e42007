// lbbnn_flow_chain: planar / radial / Householder / Sylvester transforms on a 1-D z, chained in one
// single-workgroup launch (see include/lbbnn.h).  Latency-bound: each step is one or two block reductions
// followed by an elementwise update; thread t owns elements t, t+1024, ... of z, which lives in z_out.
#include <cmath>
#include "lbbnn_device.h"
#include "lbbnn_internal.h"

namespace {

using namespace lbbnn;
constexpr int NT = 1024, NWV = NT / 64;
constexpr int MS = LBBNN_MAX_SYLVESTER_M;

// sums v[0..n) over the block (n <= MS + MS*MS handled in groups), result to every thread via LDS `res`
template <int NV>
__device__ __forceinline__ void bsum_n(double (&v)[NV], double* scratch, double* res) {
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = wave_sum(v[k]);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0)
#pragma unroll
        for (int k = 0; k < NV; ++k) scratch[k * NWV + w] = v[k];
    __syncthreads();
    if (threadIdx.x < NV) {
        double s = 0;
        for (int i = 0; i < NWV; ++i) s += scratch[threadIdx.x * NWV + i];
        res[threadIdx.x] = s;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = res[k];
}

__global__ __launch_bounds__(NT) void flow_chain_kernel(const lbbnn_flow_chain_t ch, const float* z_in, const float* q0_mean,
                                                       const float* q0_log_var, const float* eps, const uint64_t* rng,
                                                       uint32_t stream, int I, float* z, float* logdet_out, float* log_q0,
                                                       float* z_last, int ld_in, int ld_out) {
    // rows mode (lbbnn_flow_chain_rows): workgroup r carries row r of z_in through the chain
    if (z_in) z_in += (size_t)blockIdx.x * ld_in;
    z += (size_t)blockIdx.x * ld_out;
    if (logdet_out) logdet_out += blockIdx.x;
    if (z_last) z_last += blockIdx.x;
    __shared__ double scratch[8 * NWV];
    __shared__ double res[8];
    __shared__ double sBA[MS * MS], slin[MS];
    __shared__ float sh[MS];
    const int tid = threadIdx.x;
    double logdet = 0.0;
    if (z_in) {
        if (z_in != z) for (int i = tid; i < I; i += NT) z[i] = z_in[i];
    } else {
        uint64_t seed = 0, offs = 0;
        if (!eps) { seed = rng[0]; offs = rng[1]; }
        double lq[1] = {0.0};
        for (int i = tid; i < I; i += NT) {
            float e;
            if (eps) e = eps[i];
            else { float n[4]; philox_normal4(seed, offs, stream, (uint64_t)(i >> 2), 0u, n); e = n[i & 3]; }
            const float lv = q0_log_var[i], qm = q0_mean[i], ev = expf(lv);
            const float z0 = qm + sqrtf(ev) * e;
            z[i] = z0;
            const float d = z0 - qm;
            lq[0] += (double)(-0.5f * 1.1447298858494002f - 0.5f * lv - 0.5f * ((d * d) / ev));
        }
        if (log_q0) {
            bsum_n<1>(lq, scratch, res);
            if (tid == 0) log_q0[0] = (float)lq[0];
        }
    }
    for (int s = 0; s < ch.n; ++s) {
        const lbbnn_flow_step_t st = ch.step[s];
        if (st.type == LBBNN_FLOW_PLANAR) {
            double v[2] = {0.0, 0.0};
            for (int i = tid; i < I; i += NT) { v[0] += (double)(st.p1[i] * z[i]); v[1] += (double)(st.p0[i] * st.p1[i]); }
            bsum_n<2>(v, scratch, res);
            const float th = tanhf((float)v[0] + st.p2[0]);
            for (int i = tid; i < I; i += NT) z[i] += st.p0[i] * th;
            logdet += log(fabs(1.0 + (double)(1.f - th * th) * v[1]));
        } else if (st.type == LBBNN_FLOW_HOUSEHOLDER) {
            double v[2] = {0.0, 0.0};
            for (int i = tid; i < I; i += NT) { v[0] += (double)(st.p0[i] * z[i]); v[1] += (double)(st.p0[i] * st.p0[i]); }
            bsum_n<2>(v, scratch, res);
            const float c = 2.f * (float)v[0] / (float)v[1];
            for (int i = tid; i < I; i += NT) z[i] -= c * st.p0[i];
        } else if (st.type == LBBNN_FLOW_RADIAL) {
            double v[1] = {0.0};
            for (int i = tid; i < I; i += NT) { const float d = z[i] - st.p0[i]; v[0] += (double)(d * d); }
            bsum_n<1>(v, scratch, res);
            const float la = st.p1[0], beta = st.p2[0];
            const float alpha = la > 20.f ? la : log1pf(expf(la));      // nn.Softplus(beta=1, threshold=20)
            const float r = sqrtf((float)v[0]);
            const float H1 = beta / (alpha + r), H2 = -beta * r / ((alpha + r) * (alpha + r));
            for (int i = tid; i < I; i += NT) z[i] = z[i] + H1 + H2;
            logdet += (double)((float)(I - 1) * logf(1.f + H1) + logf(1.f + H1 + H2));
        } else {                                                        // Sylvester
            const int M = st.M;
            const float *A = st.p0, *Bm = st.p1;
            // lin = B z + b  and  BA = B A  (M + M*M block sums, 4 at a time)
            for (int base = 0; base < M + M * M; base += 4) {
                double v[4] = {0.0, 0.0, 0.0, 0.0};
                for (int k = 0; k < 4; ++k) {
                    const int q = base + k;
                    if (q >= M + M * M) break;
                    double acc = 0.0;
                    if (q < M) { for (int i = tid; i < I; i += NT) acc += (double)(Bm[(size_t)q * I + i] * z[i]); }
                    else {
                        const int m = (q - M) / M, n = (q - M) % M;
                        for (int i = tid; i < I; i += NT) acc += (double)(Bm[(size_t)m * I + i] * A[(size_t)i * M + n]);
                    }
                    v[k] = acc;
                }
                bsum_n<4>(v, scratch, res);
                if (tid == 0)
                    for (int k = 0; k < 4; ++k) {
                        const int q = base + k;
                        if (q < M) slin[q] = v[k] + (double)st.p2[q];
                        else if (q < M + M * M) sBA[q - M] = v[k];
                    }
            }
            __syncthreads();
            if (tid < M) sh[tid] = tanhf((float)slin[tid]);
            __syncthreads();
            for (int i = tid; i < I; i += NT) {
                float acc = 0.f;
                for (int m = 0; m < M; ++m) acc += A[(size_t)i * M + m] * sh[m];
                z[i] += acc;
            }
            // det(I + diag(1 - h^2) BA) by LU with partial pivoting (every thread redundantly: M <= 8)
            double Mx[MS][MS];
            for (int m = 0; m < M; ++m)
                for (int n = 0; n < M; ++n)
                    Mx[m][n] = (m == n ? 1.0 : 0.0) + (double)(1.f - sh[m] * sh[m]) * sBA[m * M + n];
            double det = 1.0;
            for (int c = 0; c < M; ++c) {
                int piv = c;
                for (int r2 = c + 1; r2 < M; ++r2) if (fabs(Mx[r2][c]) > fabs(Mx[piv][c])) piv = r2;
                if (piv != c) { for (int n = 0; n < M; ++n) { const double t = Mx[c][n]; Mx[c][n] = Mx[piv][n]; Mx[piv][n] = t; } det = -det; }
                det *= Mx[c][c];
                if (Mx[c][c] == 0.0) break;
                for (int r2 = c + 1; r2 < M; ++r2) {
                    const double f = Mx[r2][c] / Mx[c][c];
                    for (int n = c; n < M; ++n) Mx[r2][n] -= f * Mx[c][n];
                }
            }
            logdet += log(det);
            __syncthreads();
        }
    }
    if (tid == 0 && logdet_out) logdet_out[0] = (float)logdet;
    if (z_last && tid == ((I - 1) % NT)) z_last[0] = z[I - 1];
}

}  // namespace

extern "C" int lbbnn_flow_chain(const lbbnn_flow_chain_t* chain, const float* z_in, const float* q0_mean,
                                const float* q0_log_var, const float* eps, const uint64_t* rng, uint32_t rng_stream, int I,
                                float* z_out, float* logdet, float* log_q0, float* z_last, void* stream) {
    if (!chain || !z_out) return LBBNN_E_NULL;
    if (I <= 0 || I > LBBNN_MAX_FLOW_DIM || chain->n < 0 || chain->n > LBBNN_MAX_FLOW_T) return LBBNN_E_SHAPE;
    if (!z_in && (!q0_mean || !q0_log_var || (!eps && !rng))) return LBBNN_E_NULL;
    for (int s = 0; s < chain->n; ++s) {
        const lbbnn_flow_step_t& st = chain->step[s];
        if (st.type < LBBNN_FLOW_PLANAR || st.type > LBBNN_FLOW_SYLVESTER || !st.p0) return LBBNN_E_SHAPE;
        if (st.type != LBBNN_FLOW_HOUSEHOLDER && (!st.p1 || !st.p2)) return LBBNN_E_NULL;
        if (st.type == LBBNN_FLOW_SYLVESTER && (st.M < 1 || st.M > LBBNN_MAX_SYLVESTER_M)) return LBBNN_E_SHAPE;
    }
    hipLaunchKernelGGL(flow_chain_kernel, dim3(1), dim3(NT), 0, static_cast<hipStream_t>(stream), *chain, z_in, q0_mean,
                       q0_log_var, eps, rng, rng_stream, I, z_out, logdet, log_q0, z_last, 0, 0);
    return (int)hipGetLastError();
}

extern "C" int lbbnn_flow_chain_rows(const lbbnn_flow_chain_t* chain, const float* z_in, int ldz, int R, int I,
                                     float* z_out, int ldo, float* logdet_rows, void* stream) {
    if (!chain || !z_in || !z_out) return LBBNN_E_NULL;
    if (I <= 0 || I > LBBNN_MAX_FLOW_DIM || R < 0 || ldz < I || ldo < I || chain->n < 0 || chain->n > LBBNN_MAX_FLOW_T)
        return LBBNN_E_SHAPE;
    for (int s = 0; s < chain->n; ++s) {
        const lbbnn_flow_step_t& st = chain->step[s];
        if (st.type < LBBNN_FLOW_PLANAR || st.type > LBBNN_FLOW_SYLVESTER || !st.p0) return LBBNN_E_SHAPE;
        if (st.type != LBBNN_FLOW_HOUSEHOLDER && (!st.p1 || !st.p2)) return LBBNN_E_NULL;
        if (st.type == LBBNN_FLOW_SYLVESTER && (st.M < 1 || st.M > LBBNN_MAX_SYLVESTER_M)) return LBBNN_E_SHAPE;
    }
    if (R == 0) return 0;
    hipLaunchKernelGGL(flow_chain_kernel, dim3(R), dim3(NT), 0, static_cast<hipStream_t>(stream), *chain, z_in, nullptr,
                       nullptr, nullptr, nullptr, 0u, I, z_out, logdet_rows, nullptr, nullptr, ldz, ldo);
    return (int)hipGetLastError();
}
