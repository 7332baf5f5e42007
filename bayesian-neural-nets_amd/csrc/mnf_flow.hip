// K3 / K5 and small utilities of the MNF layer (gfx950).  All kernels are batched over layers
// (blockIdx.y / blockIdx.x selects the layer) so a network forward needs one launch per kind.
//
// K3  mnf_flow_planar : z sampling + planar normalizing flows + log_q0.
//     Work is O(T*I) (a few KB): launch/latency bound, so everything a layer needs from the flows
//     is done in ONE launch of two independent workgroups per layer (forward multiplier z_k | KL
//     branch z2 + r_flow), each a single 256-thread workgroup.  Fast path: every parameter element a
//     thread needs is loaded into registers before the first reduction (one HBM/L2 latency for the
//     whole kernel); generic path: z resident in LDS.  Dot products use fixed-order wave butterflies
//     => deterministic.
// K5  kl_finalize     : O(O+I) tail of the KL: kl_bias, tanh/mean of the auxiliary activations,
//     log_rb, and the final scalar.
#include "lbbnn_device.h"
#include "lbbnn_internal.h"

namespace {

using namespace lbbnn;

typedef lbbnn_planar_flow_t PlanarSet;
struct FlowBatch { FlowArgs l[LBBNN_MAX_LAYERS]; };
struct FinalizeBatch { FinalizeArgs l[LBBNN_MAX_LAYERS]; };

// Apply the T planar transforms of `ps` to the LDS-resident z (flows2.py:86-95); returns sum of log-dets.
__device__ __forceinline__ float planar_apply(const PlanarSet& ps, float* z, int I, double* scratch) {
    float logdet = 0.f;
    for (int t = 0; t < ps.T; ++t) {
        const float* __restrict__ u = ps.u[t];
        const float* __restrict__ w = ps.w[t];
        double s_wz = 0.0, s_uw = 0.0;
        for (int i = threadIdx.x; i < I; i += 256) {
            const float wi = w[i];
            s_wz += (double)(wi * z[i]);
            s_uw += (double)(u[i] * wi);
        }
        s_wz = block_sum<double, 4>(s_wz, scratch);
        s_uw = block_sum<double, 4>(s_uw, scratch);
        const float inner = (float)s_wz + ps.b[t][0];                 // dot(w,z) + bias          :87
        const float th = tanhf(inner);
        for (int i = threadIdx.x; i < I; i += 256) z[i] += u[i] * th; // z + u*tanh(inner)        :88
        // dot(u, (1-tanh^2)*w) = (1-tanh^2)*dot(u,w)                                             :89,:95
        logdet += logf(fabsf(1.f + (1.f - th * th) * (float)s_uw));
        __syncthreads();
    }
    return logdet;
}

__global__ __launch_bounds__(256) void mnf_flow_planar_kernel(const FlowBatch bt) {
    const FlowArgs& a = bt.l[blockIdx.y];
    if (blockIdx.x == 1 && !a.want_kl) return;
    extern __shared__ __attribute__((aligned(16))) float z[];
    __shared__ double scratch[4];
    const bool klblk = blockIdx.x == 1;
    const float* eps = klblk ? a.eps_kl : a.eps_fwd;
    uint64_t seed = 0, offs = 0;
    if (!eps) { seed = a.rng[0]; offs = a.rng[1]; }
    const uint32_t stream = (klblk ? LBBNN_STREAM_EPS_Z2 : LBBNN_STREAM_EPS_Z) * 64u + a.layer;

    // z0 = q0_mean + exp(q0_log_var)^.5 * eps                       LBBNN-GP-MF-MNF.py:183-185
    double lq0 = 0.0;
    for (int i = threadIdx.x; i < a.I; i += 256) {
        float e;
        if (eps) e = eps[i];
        else { float n[4]; philox_normal4(seed, offs, stream, (uint64_t)(i >> 2), 0u, n); e = n[i & 3]; }
        const float lv = a.q0_log_var[i], qm = a.q0_mean[i];
        const float ev = expf(lv);
        const float z0 = qm + sqrtf(ev) * e;
        z[i] = z0;
        if (klblk) {                                                  // log_q0, -0.5*log(pi)   :213-214
            const float d = z0 - qm;
            lq0 += (double)(-0.5f * 1.1447298858494002f - 0.5f * lv - 0.5f * ((d * d) / ev));
        }
    }
    __syncthreads();
    const float ldq = planar_apply(a.zf, z, a.I, scratch);            // z_flow                 :186
    float* zo = klblk ? a.z_kl : a.z_fwd;
    for (int i = threadIdx.x; i < a.I; i += 256) zo[i] = z[i];
    if (!klblk) {
        if (threadIdx.x == 0 && a.scal) a.scal[4] = ldq;              // logdet returned by sample_z(B) :187
        return;
    }

    lq0 = block_sum<double, 4>(lq0, scratch);
    const float ldr = planar_apply(a.rf, z, a.I, scratch);            // r_flow(z2)             :222
    if (threadIdx.x == 0) {
        a.scal[0] = ldq;
        a.scal[1] = (float)lq0;
        a.scal[2] = ldr;
        a.scal[3] = z[a.I - 1];                                       // z_b[-1]: last ELEMENT  :224
    }
}


// ---- fast path: I <= 256*KMAX and T <= 4.  Everything one thread needs (its KMAX elements of z, of
// q0_mean / q0_log_var / eps and of every transform's u, w) is loaded into registers BEFORE the first
// reduction, so the whole kernel pays one HBM/L2 latency instead of one per transform; z never leaves
// registers and the only LDS traffic is the two-value block reductions.
constexpr int FT = 4;     // max transforms per flow on the fast path

template <typename T2>
__device__ __forceinline__ void block_sum2(double& a, double& b, T2* scratch) {
    a = wave_sum(a); b = wave_sum(b);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) { scratch[w] = a; scratch[4 + w] = b; }
    __syncthreads();
    a = (scratch[0] + scratch[1]) + (scratch[2] + scratch[3]);
    b = (scratch[4] + scratch[5]) + (scratch[6] + scratch[7]);
}

template <int KMAX>
__device__ __forceinline__ float planar_apply_reg(const float (&u)[FT][KMAX], const float (&w)[FT][KMAX],
                                                  const float (&bias)[FT], int T, float (&z)[KMAX], double* scratch) {
    float logdet = 0.f;
#pragma unroll
    for (int t = 0; t < FT; ++t) {
        if (t < T) {
            double s_wz = 0.0, s_uw = 0.0;
#pragma unroll
            for (int k = 0; k < KMAX; ++k) { s_wz += (double)(w[t][k] * z[k]); s_uw += (double)(u[t][k] * w[t][k]); }
            block_sum2(s_wz, s_uw, scratch);
            const float th = tanhf((float)s_wz + bias[t]);
#pragma unroll
            for (int k = 0; k < KMAX; ++k) z[k] += u[t][k] * th;
            logdet += logf(fabsf(1.f + (1.f - th * th) * (float)s_uw));
        }
    }
    return logdet;
}

template <int KMAX>
__global__ __launch_bounds__(256) void mnf_flow_planar_fast_kernel(const FlowBatch bt) {
    const FlowArgs& a = bt.l[blockIdx.y];
    if (blockIdx.x == 1 && !a.want_kl) return;
    __shared__ double scratch[8];
    const bool klblk = blockIdx.x == 1;
    const float* eps = klblk ? a.eps_kl : a.eps_fwd;
    const int tid = threadIdx.x;
    float zu[FT][KMAX], zw[FT][KMAX], ru[FT][KMAX], rw[FT][KMAX], zb[FT], rb[FT];
    float qm[KMAX], lv[KMAX], e[KMAX];
    // ---- issue every load first (indices past I read as 0 and contribute nothing)
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        const int i = tid + 256 * k;
        const bool in = i < a.I;
        qm[k] = in ? a.q0_mean[i] : 0.f;
        lv[k] = in ? a.q0_log_var[i] : 0.f;
        e[k] = (in && eps) ? eps[i] : 0.f;
#pragma unroll
        for (int t = 0; t < FT; ++t) {
            zu[t][k] = (in && t < a.zf.T) ? a.zf.u[t][i] : 0.f;
            zw[t][k] = (in && t < a.zf.T) ? a.zf.w[t][i] : 0.f;
            ru[t][k] = (in && klblk && t < a.rf.T) ? a.rf.u[t][i] : 0.f;
            rw[t][k] = (in && klblk && t < a.rf.T) ? a.rf.w[t][i] : 0.f;
        }
    }
#pragma unroll
    for (int t = 0; t < FT; ++t) {
        zb[t] = t < a.zf.T ? a.zf.b[t][0] : 0.f;
        rb[t] = (klblk && t < a.rf.T) ? a.rf.b[t][0] : 0.f;
    }
    if (!eps) {
        const uint64_t seed = a.rng[0], offs = a.rng[1];
        const uint32_t stream = (klblk ? LBBNN_STREAM_EPS_Z2 : LBBNN_STREAM_EPS_Z) * 64u + a.layer;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const int i = tid + 256 * k;
            float n[4];
            philox_normal4(seed, offs, stream, (uint64_t)(i >> 2), 0u, n);
            e[k] = n[i & 3];
        }
    }
    float z[KMAX];
    double lq0 = 0.0;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        const int i = tid + 256 * k;
        const float ev = expf(lv[k]);
        const float z0 = qm[k] + sqrtf(ev) * e[k];                    // LBBNN-GP-MF-MNF.py:183-185
        z[k] = (i < a.I) ? z0 : 0.f;
        if (klblk && i < a.I) {
            const float d = z0 - qm[k];
            lq0 += (double)(-0.5f * 1.1447298858494002f - 0.5f * lv[k] - 0.5f * ((d * d) / ev));   // :213-214
        }
    }
    const float ldq = planar_apply_reg<KMAX>(zu, zw, zb, a.zf.T, z, scratch);
    float* zo = klblk ? a.z_kl : a.z_fwd;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) { const int i = tid + 256 * k; if (i < a.I) zo[i] = z[k]; }
    if (!klblk) {
        if (tid == 0 && a.scal) a.scal[4] = ldq;
        return;
    }
    double dummy = 0.0;
    block_sum2(lq0, dummy, scratch);
    const float ldr = planar_apply_reg<KMAX>(ru, rw, rb, a.rf.T, z, scratch);
    // z_b[-1]: the thread that owns element I-1 publishes it                        :224
    const int last = a.I - 1;
    if (tid == (last & 255)) {
        float zl = 0.f;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) if (k == (last >> 8)) zl = z[k];
        a.scal[3] = zl;
    }
    if (tid == 0) { a.scal[0] = ldq; a.scal[1] = (float)lq0; a.scal[2] = ldr; }
}

// -------------------------------------------------------------------------------------------- K5
__global__ __launch_bounds__(256) void kl_finalize_kernel(const FinalizeBatch bt) {
    const FinalizeArgs& a = bt.l[blockIdx.x];
    __shared__ double scratch[4];
    const bool mnf = a.scal != nullptr;
    uint64_t seed = 0, offs = 0;
    if (mnf && !a.eps_act) { seed = a.rng[0]; offs = a.rng[1]; }
    double s_rows = 0.0, s_bias = 0.0, s_act = 0.0;
    for (int o = threadIdx.x; o < a.O; o += 256) {
        s_rows += (double)a.kl_rows[o];
        const float sb = softplus_ref(a.bias_rho[o]);
        const float d = a.bias_mu[o] - a.bias_mu_prior;
        const float sp = a.bias_sigma_prior;
        s_bias += (double)(logf(sp / sb) - 0.5f + (sb * sb + d * d) / (2.f * sp * sp));   // …LRT.py:185-186
        if (mnf) {
            float e;
            if (a.eps_act) e = a.eps_act[o];
            else { float n[4]; philox_normal4(seed, offs, LBBNN_STREAM_EPS_ACT * 64u + a.layer, (uint64_t)(o >> 2), 0u, n); e = n[o & 3]; }
            s_act += (double)tanhf(a.act_mu[o] + sqrtf(a.act_var[o]) * e);               // …MNF.py:218-219
        }
    }
    s_rows = block_sum<double, 4>(s_rows, scratch);
    s_bias = block_sum<double, 4>(s_bias, scratch);
    double kl = s_bias + s_rows;
    if (mnf) {
        s_act = block_sum<double, 4>(s_act, scratch);
        const float m = (float)(s_act / (double)a.O);       // outer(b, act).mean(-1) = b * mean(act)   :220-221
        const float zb = a.scal[3];
        double s_rb = 0.0;
        for (int i = threadIdx.x; i < a.I; i += 256) {
            const float mr = a.r0_b1[i] * m, lv = a.r0_b2[i] * m;
            const float d = zb - mr;
            s_rb += (double)(-0.5f * 1.1447298858494002f - 0.5f * lv - 0.5f * ((d * d) / expf(lv)));  // :223-224
        }
        s_rb = block_sum<double, 4>(s_rb, scratch);
        const double log_q = -(double)a.scal[0] + (double)a.scal[1];                      // :215
        const double log_r = (double)a.scal[2] + s_rb;                                    // :225
        kl += log_q - log_r;                                                              // :235
    }
    if (threadIdx.x == 0) {
        const float k = (float)kl;
        if (a.kl_layer) *a.kl_layer = k;
        if (a.kl_out) *a.kl_out = a.accum ? (*a.kl_out + k) : k;
    }
}

// -------------------------------------------------------------------------------------------- utilities
__global__ void rng_advance_kernel(uint64_t* rng, uint64_t delta) { rng[1] += delta; }

struct FinishArgs { const float* kl[LBBNN_MAX_LAYERS]; int n; float* total; uint64_t* rng; uint64_t delta; };
__global__ void forward_finish_kernel(const FinishArgs a) {
    if (a.total) {
        float s = 0.f;
        for (int i = 0; i < a.n; ++i) s += *a.kl[i];          // fixed order: l1 + l2 + l3
        *a.total = s;
    }
    if (a.rng) a.rng[1] += a.delta;
}

__global__ __launch_bounds__(256) void philox_normal_kernel(const uint64_t* rng, uint32_t stream, long long row_base,
                                                            long long rows, long long cols, float* out) {
    // one counter (4 normals) per thread.  2-D (rows > 0): out[r][c] = N(ctr0 = row_base + r, ctr1 = c/4)[c%4],
    // the GEMM epilogue's indexing; 1-D (rows == 0): out[i] = N(ctr0 = i/4, ctr1 = 0)[i%4], the flow kernels'.
    const long long gpr = (cols + 3) / 4;
    const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long nrows = rows > 0 ? rows : 1;
    if (g >= gpr * nrows) return;
    const long long r = g / gpr, cg = g % gpr;
    float v[4];
    if (rows > 0) philox_normal4(rng[0], rng[1], stream, (uint64_t)(row_base + r), (uint32_t)cg, v);
    else          philox_normal4(rng[0], rng[1], stream, (uint64_t)cg, 0u, v);
#pragma unroll
    for (int k = 0; k < 4; ++k) if (cg * 4 + k < cols) out[r * cols + cg * 4 + k] = v[k];
}

__global__ __launch_bounds__(256) void log_softmax_rows_kernel(const float* in, int ldi, float* out, int ldo, int B, int O) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    const float* p = in + (size_t)b * ldi;
    float mx = -INFINITY;
#pragma unroll 1
    for (int o = 0; o < O; ++o) mx = fmaxf(mx, p[o]);
    float s = 0.f;
#pragma unroll 1
    for (int o = 0; o < O; ++o) s += expf(p[o] - mx);
    const float lse = mx + logf(s);
    float* q = out + (size_t)b * ldo;
#pragma unroll 1
    for (int o = 0; o < O; ++o) q[o] = p[o] - lse;
}

bool fill_set(PlanarSet& ps, const float* const* u, const float* const* w, const float* const* b, int T) {
    ps.T = T;
    for (int t = 0; t < LBBNN_MAX_FLOW_T; ++t) { ps.u[t] = nullptr; ps.w[t] = nullptr; ps.b[t] = nullptr; }
    for (int t = 0; t < T; ++t) {
        if (!u || !w || !b || !u[t] || !w[t] || !b[t]) return false;
        ps.u[t] = u[t]; ps.w[t] = w[t]; ps.b[t] = b[t];
    }
    return true;
}

}  // namespace

namespace lbbnn {

int launch_flow_planar(const FlowArgs* a, int n, hipStream_t s) {
    FlowBatch bt;
    int maxI = 0; bool small_t = true, any_kl = false;
    for (int i = 0; i < n; ++i) {
        bt.l[i] = a[i];
        maxI = a[i].I > maxI ? a[i].I : maxI;
        small_t = small_t && a[i].zf.T <= FT && (!a[i].want_kl || a[i].rf.T <= FT);
        any_kl = any_kl || a[i].want_kl;
    }
    const dim3 grid(any_kl ? 2 : 1, n), block(256);
    if (small_t && maxI <= 256 * 2)      hipLaunchKernelGGL(mnf_flow_planar_fast_kernel<2>, grid, block, 0, s, bt);
    else if (small_t && maxI <= 256 * 5) hipLaunchKernelGGL(mnf_flow_planar_fast_kernel<5>, grid, block, 0, s, bt);
    else hipLaunchKernelGGL(mnf_flow_planar_kernel, grid, block, (size_t)maxI * sizeof(float), s, bt);
    return (int)hipGetLastError();
}

int launch_kl_finalize(const FinalizeArgs* a, int n, hipStream_t s) {
    FinalizeBatch bt;
    for (int i = 0; i < n; ++i) bt.l[i] = a[i];
    hipLaunchKernelGGL(kl_finalize_kernel, dim3(n), dim3(256), 0, s, bt);
    return (int)hipGetLastError();
}

}  // namespace lbbnn

using namespace lbbnn;

extern "C" int lbbnn_mnf_flow_planar(const float* q0_mean, const float* q0_log_var,
                                     const float* const* zu, const float* const* zw, const float* const* zb, int Tz,
                                     const float* const* ru, const float* const* rw, const float* const* rb, int Tr,
                                     const float* eps_fwd, const float* eps_kl,
                                     const uint64_t* rng, uint32_t layer_id,
                                     float* z_fwd, float* z_kl, float* scal,
                                     int I, int want_kl, void* stream) {
    if (!q0_mean || !q0_log_var || !z_fwd) return LBBNN_E_NULL;
    if (I <= 0 || I > LBBNN_MAX_FLOW_DIM || Tz < 0 || Tz > LBBNN_MAX_FLOW_T || Tr < 0 || Tr > LBBNN_MAX_FLOW_T) return LBBNN_E_SHAPE;
    if (want_kl && (!z_kl || !scal)) return LBBNN_E_NULL;
    if ((!eps_fwd || (want_kl && !eps_kl)) && !rng) return LBBNN_E_NOISE;
    FlowArgs a;
    a.q0_mean = q0_mean; a.q0_log_var = q0_log_var; a.eps_fwd = eps_fwd; a.eps_kl = eps_kl; a.rng = rng;
    a.z_fwd = z_fwd; a.z_kl = z_kl; a.scal = scal; a.I = I; a.want_kl = want_kl; a.layer = layer_id & 63u;
    if (!fill_set(a.zf, zu, zw, zb, Tz)) return LBBNN_E_NULL;
    if (!fill_set(a.rf, ru, rw, rb, want_kl ? Tr : 0)) return LBBNN_E_NULL;
    return launch_flow_planar(&a, 1, static_cast<hipStream_t>(stream));
}

extern "C" int lbbnn_kl_finalize(const float* kl_rows, const float* bias_mu, const float* bias_rho, int O,
                                 const float* act_mu, const float* act_var, const float* eps_act,
                                 const float* r0_b1, const float* r0_b2, int I,
                                 const float* scal, const lbbnn_priors_t* priors,
                                 const uint64_t* rng, uint32_t layer_id,
                                 float* kl_out, float* kl_layer, int kl_accum, void* stream) {
    if (!kl_rows || !bias_mu || !bias_rho || !priors || (!kl_out && !kl_layer)) return LBBNN_E_NULL;
    if (O <= 0) return LBBNN_E_SHAPE;
    if (scal) {
        if (!act_mu || !act_var || !r0_b1 || !r0_b2) return LBBNN_E_NULL;
        if (I <= 0) return LBBNN_E_SHAPE;
        if (!eps_act && !rng) return LBBNN_E_NOISE;
    }
    FinalizeArgs a;
    a.kl_rows = kl_rows; a.bias_mu = bias_mu; a.bias_rho = bias_rho; a.act_mu = act_mu; a.act_var = act_var;
    a.eps_act = eps_act; a.r0_b1 = r0_b1; a.r0_b2 = r0_b2; a.scal = scal; a.rng = rng;
    a.kl_out = kl_out; a.kl_layer = kl_layer; a.O = O; a.I = I; a.accum = kl_accum; a.layer = layer_id & 63u;
    a.bias_mu_prior = priors->bias_mu_prior; a.bias_sigma_prior = priors->bias_sigma_prior;
    return launch_kl_finalize(&a, 1, static_cast<hipStream_t>(stream));
}

extern "C" int lbbnn_forward_finish(uint64_t* rng, uint64_t advance, const float* const* kl_layers, int n,
                                    float* kl_total, void* stream) {
    if (!rng && !kl_total) return LBBNN_E_NULL;
    if (kl_total && (n <= 0 || n > LBBNN_MAX_LAYERS || !kl_layers)) return LBBNN_E_SHAPE;
    FinishArgs a;
    a.n = kl_total ? n : 0; a.total = kl_total; a.rng = rng; a.delta = advance;
    for (int i = 0; i < LBBNN_MAX_LAYERS; ++i) a.kl[i] = (kl_total && i < n) ? kl_layers[i] : nullptr;
    for (int i = 0; i < a.n; ++i) if (!a.kl[i]) return LBBNN_E_NULL;
    hipLaunchKernelGGL(forward_finish_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream), a);
    return (int)hipGetLastError();
}

extern "C" int lbbnn_rng_advance(uint64_t* rng, uint64_t delta, void* stream) {
    if (!rng) return LBBNN_E_NULL;
    hipLaunchKernelGGL(rng_advance_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream), rng, delta);
    return (int)hipGetLastError();
}

extern "C" int lbbnn_philox_normal(const uint64_t* rng, uint32_t rng_stream, int64_t row_base, int64_t rows,
                                   int64_t cols, float* out, void* stream) {
    if (!rng || !out) return LBBNN_E_NULL;
    if (cols <= 0 || rows < 0) return LBBNN_E_SHAPE;
    const long long groups = ((cols + 3) / 4) * (rows > 0 ? rows : 1);
    hipLaunchKernelGGL(philox_normal_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), rng, rng_stream, (long long)row_base, (long long)rows,
                       (long long)cols, out);
    return (int)hipGetLastError();
}

extern "C" int lbbnn_log_softmax_rows(const float* in, int ldi, float* out, int ldo, int B, int O, void* stream) {
    if (!in || !out) return LBBNN_E_NULL;
    if (B <= 0 || O <= 0 || O > 64 || ldi < O || ldo < O) return LBBNN_E_SHAPE;
    hipLaunchKernelGGL(log_softmax_rows_kernel, dim3((B + 255) / 256), dim3(256), 0,
                       static_cast<hipStream_t>(stream), in, ldi, out, ldo, B, O);
    return (int)hipGetLastError();
}

extern "C" int lbbnn_abi_version(void) { return LBBNN_ABI_VERSION; }

extern "C" const char* lbbnn_error_string(int code) {
    switch (code) {
        case LBBNN_OK: return "ok";
        case LBBNN_E_NULL: return "lbbnn: a required pointer is NULL";
        case LBBNN_E_SHAPE: return "lbbnn: bad dimension";
        case LBBNN_E_ALIGN: return "lbbnn: misaligned pointer or leading dimension";
        case LBBNN_E_FLAGS: return "lbbnn: bad flags";
        case LBBNN_E_NOISE: return "lbbnn: no noise source (explicit draw or rng state)";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "lbbnn: unknown error";
    }
}
