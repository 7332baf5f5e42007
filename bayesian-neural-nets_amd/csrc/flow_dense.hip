// K4 -- dense coupling flows (RNVP, MNF type) on the single kept z row (gfx950).  See include/lbbnn.h.
//
// All work is GEMV-shaped and small (I*H + 3*H*H + 2*H*I MACs per transform, H = 75/100), so the design
// goal is latency: the two I-long stages are spread over many workgroups, the H x H chain (17 k MACs) is
// recomputed by every output workgroup instead of costing a launch, and both paths of a layer (forward
// draw | KL branch) ride in the same launches (blockIdx.y = path).
#include "lbbnn_device.h"
#include "lbbnn_internal.h"

namespace {

using namespace lbbnn;

constexpr int HMAX = LBBNN_MAX_HIDDEN;
constexpr int CB = 256;            // outputs per workgroup in the output stage

struct InitArgs {
    const float* q0_mean; const float* q0_log_var; const float* eps[2]; const uint64_t* rng;
    float* z[2]; float* lq0_part;   // per-block partials of log_q0 (KL path)
    int I; uint32_t layer; int npaths;
};

// z0 = q0_mean + exp(q0_log_var)^.5 * eps  (LBBNN-GP-MF-MNF.py:183-185) for both paths; log_q0 partials (:213-214)
struct InitBatch { InitArgs l[LBBNN_MAX_LAYERS]; };

__global__ __launch_bounds__(256) void dense_init_kernel(const InitBatch bt) {
    __shared__ double scratch[4];
    const LBBNN_CONST_AS InitArgs& a = kernarg_as<InitBatch>()->l[blockIdx.z];      // layer = blockIdx.z, no scratch copy
    const int path = blockIdx.y;
    if (blockIdx.x * 256 >= a.I) return;                                            // grid sized for the widest layer
    const int i = blockIdx.x * 256 + threadIdx.x;
    double lq = 0.0;
    if (i < a.I) {
        float e;
        if (a.eps[path]) e = a.eps[path][i];
        else {
            float n[4];
            philox_normal4(a.rng[0], a.rng[1], (path ? LBBNN_STREAM_EPS_Z2 : LBBNN_STREAM_EPS_Z) * 64u + a.layer, (uint64_t)(i >> 2), 0u, n);
            e = n[i & 3];
        }
        const float lv = a.q0_log_var[i], qm = a.q0_mean[i];
        const float ev = expf(lv);
        const float z0 = qm + sqrtf(ev) * e;
        a.z[path][i] = z0;
        if (path == 1) {
            const float d = z0 - qm;
            lq = (double)(-0.5f * 1.1447298858494002f - 0.5f * lv - 0.5f * ((d * d) / ev));
        }
    }
    if (path == 1) {
        lq = block_sum<double, 4>(lq, scratch);
        if (threadIdx.x == 0) a.lq0_part[blockIdx.x] = (float)lq;
    }
}

struct StageArgs {
    lbbnn_dense_transform_t tr;
    float* z[2];          // current z of each path (updated in place by the output stage)
    float* h[2];          // hidden activations of stage A, HMAX floats per path
    float* zcopy;         // if non-NULL: path 1 also stores its new z here (z_kl, after the last z_flow transform)
    float* ld_part[2];    // per-workgroup log-det partials of this transform
    int I; int path_lo, npaths;   // paths handled: path_lo .. path_lo + npaths - 1
};

// Stage A: one wave per hidden unit j: h[j] = act( sum_i W_in[j,i] * (m_i z_i) + b_in[j] )
//   RNVP: LeakyReLU(0.1) (flows2.py:176-185,212)   MNF: tanh (flows2.py:235)
struct StageBatch { StageArgs l[LBBNN_MAX_LAYERS]; };

__global__ __launch_bounds__(256) void dense_stage_a_kernel(const StageBatch bt) {
    const LBBNN_CONST_AS StageArgs& a = kernarg_as<StageBatch>()->l[blockIdx.z];
    const int path = a.path_lo + blockIdx.y;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (j >= a.tr.hidden) return;
    const float* __restrict__ w = a.tr.w_in + (size_t)j * a.I;
    const float* __restrict__ z = a.z[path];
    const float* __restrict__ m = path ? a.tr.mask_kl : a.tr.mask_fwd;
    float s = 0.f;
    for (int i = lane; i < a.I; i += 64) s += w[i] * (m[i] * z[i]);
    s = wave_sum(s);
    if (lane == 0) {
        const float v = s + a.tr.b_in[j];
        a.h[path][j] = a.tr.kind == LBBNN_FLOW_RNVP ? (v >= 0.f ? v : 0.1f * v) : tanhf(v);
    }
}

// Output stage: (RNVP) finish the MLP chain h1 -> h4 in LDS, then per output i the two H-long dots,
// the gate and the coupling update, plus this workgroup's log-det partial.
__global__ __launch_bounds__(CB) void dense_stage_c_kernel(const StageBatch bt) {
    __shared__ float hs[2][HMAX];
    __shared__ double scratch[4];
    const LBBNN_CONST_AS StageArgs& a = kernarg_as<StageBatch>()->l[blockIdx.z];
    if (blockIdx.x * CB >= a.I) return;                                             // grid sized for the widest layer
    const int path = a.path_lo + blockIdx.y;
    const int H = a.tr.hidden, tid = threadIdx.x;
    if (tid < H) hs[0][tid] = a.h[path][tid];
    __syncthreads();
    int cur = 0;
    if (a.tr.kind == LBBNN_FLOW_RNVP) {
#pragma unroll 1
        for (int l = 0; l < 3; ++l) {
            if (tid < H) {
                const float* __restrict__ w = a.tr.w_mid[l] + (size_t)tid * H;
                float s = 0.f;
                for (int k = 0; k < H; ++k) s += w[k] * hs[cur][k];
                s += a.tr.b_mid[l][tid];
                hs[cur ^ 1][tid] = (l < 2) ? (s >= 0.f ? s : 0.1f * s) : s;      // last LeakyReLU dropped (:185)
            }
            __syncthreads();
            cur ^= 1;
        }
    }
    const int i = blockIdx.x * CB + tid;
    double ld = 0.0;
    if (i < a.I) {
        const float* __restrict__ wa = a.tr.w_a + (size_t)i * H;
        const float* __restrict__ wb = a.tr.w_b + (size_t)i * H;
        float sa = 0.f, sb = 0.f;
        for (int k = 0; k < H; ++k) { const float hk = hs[cur][k]; sa += wa[k] * hk; sb += wb[k] * hk; }
        sa += a.tr.b_a[i]; sb += a.tr.b_b[i];
        const float m = (path ? a.tr.mask_kl : a.tr.mask_fwd)[i];
        const float z = a.z[path][i];
        const float g = 1.f / (1.f + expf(-sb));                               // sigmoid: gate (:214) / sigma (:237)
        float zn;
        if (a.tr.kind == LBBNN_FLOW_RNVP) zn = ((1.f - m) * z) * g + (1.f - g) * sa + m * z;       // :215
        else zn = m * z + (1.f - m) * (z * g + (1.f - g) * sa);                                     // :238
        a.z[path][i] = zn;
        if (path == 1 && a.zcopy) a.zcopy[i] = zn;
        ld = (double)((1.f - m) * logf(g));                                     // :219 / :241
    }
    ld = block_sum<double, 4>(ld, scratch);
    if (tid == 0) a.ld_part[path][blockIdx.x] = (float)ld;
}

struct FinishArgs {
    const float* ldz; const float* ldr; const float* lq0; const float* zr; const float* ldf;
    float* scal; int nblk, nblk_i, Tz, Tr, I, want_kl;
};

// scal[0] = log_det_q, [1] = log_q0, [2] = log_det_r, [3] = r_flow(z2)[-1], [4] = forward-draw log-det
struct FinishBatch { FinishArgs l[LBBNN_MAX_LAYERS]; };

__global__ void dense_finish_kernel(const FinishBatch bt) {
    const LBBNN_CONST_AS FinishArgs& a = kernarg_as<FinishBatch>()->l[blockIdx.x];
    if (threadIdx.x != 0) return;
    float s = 0.f;
    for (int t = 0; t < a.Tz * a.nblk; ++t) s += a.ldf[t];
    a.scal[4] = s;
    if (!a.want_kl) return;
    s = 0.f;
    for (int t = 0; t < a.Tz * a.nblk; ++t) s += a.ldz[t];
    a.scal[0] = s;
    s = 0.f;
    for (int t = 0; t < a.nblk_i; ++t) s += a.lq0[t];
    a.scal[1] = s;
    s = 0.f;
    for (int t = 0; t < a.Tr * a.nblk; ++t) s += a.ldr[t];
    a.scal[2] = s;
    a.scal[3] = a.zr[a.I - 1];
}

bool transform_ok(const lbbnn_dense_transform_t& t, bool need_fwd, bool need_kl) {
    if (t.kind != LBBNN_FLOW_RNVP && t.kind != LBBNN_FLOW_MNF) return false;
    if (t.hidden <= 0 || t.hidden > HMAX) return false;
    if (!t.w_in || !t.b_in || !t.w_a || !t.b_a || !t.w_b || !t.b_b) return false;
    if (t.kind == LBBNN_FLOW_RNVP)
        for (int l = 0; l < 3; ++l) if (!t.w_mid[l] || !t.b_mid[l]) return false;
    if (need_fwd && !t.mask_fwd) return false;
    if (need_kl && !t.mask_kl) return false;
    return true;
}

}  // namespace

// workspace (floats): z[2][I] | h[2][HMAX] | lq0[nblk_i] | ld_fwd[T*nblk] | ld_z[T*nblk] | ld_r[T*nblk], T <= LBBNN_MAX_FLOW_T
extern "C" int64_t lbbnn_flow_dense_workspace(int I) {
    if (I <= 0) return 0;
    const int64_t nblk = (I + CB - 1) / CB, nblk_i = (I + 255) / 256;
    return 2 * (int64_t)I + 2 * HMAX + nblk_i + 3 * (int64_t)LBBNN_MAX_FLOW_T * nblk + 64;
}

// K4 of n layers in 2 + 2*(Tz+Tr) launches (blockIdx.z = layer); the layers must agree on Tz, Tr and want_kl
static int dense_flows_impl(const lbbnn_dense_layer_t* L, int n, const uint64_t* rng, void* stream) {
    if (!L) return LBBNN_E_NULL;
    if (n <= 0 || n > LBBNN_MAX_LAYERS) return LBBNN_E_SHAPE;
    const int Tz = L[0].Tz, Tr = L[0].Tr, want_kl = L[0].want_kl;
    int maxI = 0, maxH = 1;
    for (int k = 0; k < n; ++k) {
        const lbbnn_dense_layer_t& d = L[k];
        if (!d.q0_mean || !d.q0_log_var || !d.z_fwd || !d.scal || !d.work) return LBBNN_E_NULL;
        if (d.I <= 0 || d.Tz < 0 || d.Tz > LBBNN_MAX_FLOW_T || d.Tr < 0 || d.Tr > LBBNN_MAX_FLOW_T) return LBBNN_E_SHAPE;
        if (d.Tz != Tz || d.Tr != Tr || (d.want_kl != 0) != (want_kl != 0)) return LBBNN_E_SHAPE;
        if (d.want_kl && !d.z_kl) return LBBNN_E_NULL;
        if ((!d.eps_fwd || (d.want_kl && !d.eps_kl)) && !rng) return LBBNN_E_NOISE;
        if ((d.Tz && !d.zt) || (d.want_kl && d.Tr && !d.rt)) return LBBNN_E_NULL;
        for (int t = 0; t < d.Tz; ++t) { if (!transform_ok(d.zt[t], true, d.want_kl != 0)) return LBBNN_E_NULL; maxH = d.zt[t].hidden > maxH ? d.zt[t].hidden : maxH; }
        if (d.want_kl) for (int t = 0; t < d.Tr; ++t) { if (!transform_ok(d.rt[t], false, true)) return LBBNN_E_NULL; maxH = d.rt[t].hidden > maxH ? d.rt[t].hidden : maxH; }
        maxI = d.I > maxI ? d.I : maxI;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int npaths = want_kl ? 2 : 1;
    const int gblk = (maxI + CB - 1) / CB, gblk_i = (maxI + 255) / 256;
    struct Bufs { float *zbuf1, *h0, *h1, *lq0, *ldf, *ldz, *ldr; int nblk, nblk_i; } B[LBBNN_MAX_LAYERS];
    InitBatch ib{};
    for (int k = 0; k < n; ++k) {
        const lbbnn_dense_layer_t& d = L[k];
        Bufs& b = B[k];
        b.nblk = (d.I + CB - 1) / CB; b.nblk_i = (d.I + 255) / 256;
        b.zbuf1 = d.work + d.I;                  // (work[0..I) was the path-0 scratch row of an earlier version)
        b.h0 = d.work + 2 * (size_t)d.I; b.h1 = b.h0 + HMAX; b.lq0 = b.h1 + HMAX;
        b.ldf = b.lq0 + b.nblk_i;
        b.ldz = b.ldf + (size_t)LBBNN_MAX_FLOW_T * b.nblk;
        b.ldr = b.ldz + (size_t)LBBNN_MAX_FLOW_T * b.nblk;
        InitArgs& ia = ib.l[k];
        ia.q0_mean = d.q0_mean; ia.q0_log_var = d.q0_log_var; ia.eps[0] = d.eps_fwd; ia.eps[1] = d.eps_kl; ia.rng = rng;
        ia.z[0] = d.z_fwd; ia.z[1] = b.zbuf1; ia.lq0_part = b.lq0; ia.I = d.I; ia.layer = d.layer_id & 63u; ia.npaths = npaths;
    }
    hipLaunchKernelGGL(dense_init_kernel, dim3(gblk_i, npaths, n), dim3(256), 0, s, ib);

    for (int t = 0; t < Tz + (want_kl ? Tr : 0); ++t) {
        const bool zphase = t < Tz;
        StageBatch sb{};
        for (int k = 0; k < n; ++k) {
            const lbbnn_dense_layer_t& d = L[k];
            const Bufs& b = B[k];
            StageArgs& sa = sb.l[k];
            sa.tr = zphase ? d.zt[t] : d.rt[t - Tz];
            sa.z[0] = d.z_fwd; sa.z[1] = b.zbuf1; sa.h[0] = b.h0; sa.h[1] = b.h1;
            sa.zcopy = (zphase && want_kl && t == Tz - 1) ? d.z_kl : nullptr;
            sa.ld_part[0] = b.ldf + (size_t)t * b.nblk;
            sa.ld_part[1] = zphase ? b.ldz + (size_t)t * b.nblk : b.ldr + (size_t)(t - Tz) * b.nblk;
            sa.I = d.I;
            sa.path_lo = zphase ? 0 : 1;
            sa.npaths = zphase ? npaths : 1;
        }
        const int np = zphase ? npaths : 1;
        hipLaunchKernelGGL(dense_stage_a_kernel, dim3((maxH + 3) / 4, np, n), dim3(256), 0, s, sb);
        hipLaunchKernelGGL(dense_stage_c_kernel, dim3(gblk, np, n), dim3(CB), 0, s, sb);
    }
    FinishBatch fb{};
    for (int k = 0; k < n; ++k) {
        const lbbnn_dense_layer_t& d = L[k];
        const Bufs& b = B[k];
        if (want_kl && Tz == 0) (void)hipMemcpyAsync(d.z_kl, b.zbuf1, (size_t)d.I * sizeof(float), hipMemcpyDeviceToDevice, s);
        FinishArgs& fa = fb.l[k];
        fa.ldz = b.ldz; fa.ldr = b.ldr; fa.lq0 = b.lq0; fa.zr = b.zbuf1; fa.ldf = b.ldf; fa.scal = d.scal;
        fa.nblk = b.nblk; fa.nblk_i = b.nblk_i; fa.Tz = Tz; fa.Tr = want_kl ? Tr : 0; fa.I = d.I; fa.want_kl = want_kl;
    }
    hipLaunchKernelGGL(dense_finish_kernel, dim3(n), dim3(64), 0, s, fb);
    return (int)hipGetLastError();
}

extern "C" int lbbnn_layers_dense_flows(const lbbnn_dense_layer_t* layers, int n, const uint64_t* rng, void* stream) {
    return dense_flows_impl(layers, n, rng, stream);
}

extern "C" int lbbnn_mnf_flow_dense(const float* q0_mean, const float* q0_log_var,
                                    const lbbnn_dense_transform_t* zt, int Tz,
                                    const lbbnn_dense_transform_t* rt, int Tr,
                                    const float* eps_fwd, const float* eps_kl,
                                    const uint64_t* rng, uint32_t layer_id,
                                    float* z_fwd, float* z_kl, float* scal, float* work,
                                    int I, int want_kl, void* stream) {
    lbbnn_dense_layer_t d{};
    d.q0_mean = q0_mean; d.q0_log_var = q0_log_var; d.zt = zt; d.rt = rt; d.Tz = Tz; d.Tr = Tr;
    d.eps_fwd = eps_fwd; d.eps_kl = eps_kl; d.layer_id = layer_id; d.z_fwd = z_fwd; d.z_kl = z_kl; d.scal = scal; d.work = work;
    d.I = I; d.want_kl = want_kl;
    return dense_flows_impl(&d, 1, rng, stream);
}
