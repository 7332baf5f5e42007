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
__global__ __launch_bounds__(256) void dense_init_kernel(const InitArgs a) {
    __shared__ double scratch[4];
    const int path = blockIdx.y;
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
__global__ __launch_bounds__(256) void dense_stage_a_kernel(const StageArgs a) {
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
__global__ __launch_bounds__(CB) void dense_stage_c_kernel(const StageArgs a) {
    __shared__ float hs[2][HMAX];
    __shared__ double scratch[4];
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
__global__ void dense_finish_kernel(const FinishArgs a) {
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

extern "C" int lbbnn_mnf_flow_dense(const float* q0_mean, const float* q0_log_var,
                                    const lbbnn_dense_transform_t* zt, int Tz,
                                    const lbbnn_dense_transform_t* rt, int Tr,
                                    const float* eps_fwd, const float* eps_kl,
                                    const uint64_t* rng, uint32_t layer_id,
                                    float* z_fwd, float* z_kl, float* scal, float* work,
                                    int I, int want_kl, void* stream) {
    if (!q0_mean || !q0_log_var || !z_fwd || !scal || !work) return LBBNN_E_NULL;
    if (I <= 0 || Tz < 0 || Tz > LBBNN_MAX_FLOW_T || Tr < 0 || Tr > LBBNN_MAX_FLOW_T) return LBBNN_E_SHAPE;
    if (want_kl && !z_kl) return LBBNN_E_NULL;
    if ((!eps_fwd || (want_kl && !eps_kl)) && !rng) return LBBNN_E_NOISE;
    if ((Tz && !zt) || (want_kl && Tr && !rt)) return LBBNN_E_NULL;
    for (int t = 0; t < Tz; ++t) if (!transform_ok(zt[t], true, want_kl != 0)) return LBBNN_E_NULL;
    if (want_kl) for (int t = 0; t < Tr; ++t) if (!transform_ok(rt[t], false, true)) return LBBNN_E_NULL;

    hipStream_t s = static_cast<hipStream_t>(stream);
    const int nblk = (I + CB - 1) / CB, nblk_i = (I + 255) / 256;
    const int npaths = want_kl ? 2 : 1;
    float* zbuf0 = work;                 // path 0 works directly in a scratch row, copied to z_fwd at the end
    float* zbuf1 = work + I;
    float* h0 = work + 2 * (size_t)I;
    float* h1 = h0 + HMAX;
    float* lq0 = h1 + HMAX;
    float* ldf = lq0 + nblk_i;
    float* ldz = ldf + (size_t)LBBNN_MAX_FLOW_T * nblk;
    float* ldr = ldz + (size_t)LBBNN_MAX_FLOW_T * nblk;
    (void)zbuf0;

    InitArgs ia;
    ia.q0_mean = q0_mean; ia.q0_log_var = q0_log_var; ia.eps[0] = eps_fwd; ia.eps[1] = eps_kl; ia.rng = rng;
    ia.z[0] = z_fwd; ia.z[1] = zbuf1; ia.lq0_part = lq0; ia.I = I; ia.layer = layer_id & 63u; ia.npaths = npaths;
    hipLaunchKernelGGL(dense_init_kernel, dim3(nblk_i, npaths), dim3(256), 0, s, ia);

    for (int t = 0; t < Tz + (want_kl ? Tr : 0); ++t) {
        const bool zphase = t < Tz;
        StageArgs sa;
        sa.tr = zphase ? zt[t] : rt[t - Tz];
        sa.z[0] = z_fwd; sa.z[1] = zbuf1; sa.h[0] = h0; sa.h[1] = h1;
        sa.zcopy = (zphase && want_kl && t == Tz - 1) ? z_kl : nullptr;
        sa.ld_part[0] = ldf + (size_t)t * nblk;
        sa.ld_part[1] = zphase ? ldz + (size_t)t * nblk : ldr + (size_t)(t - Tz) * nblk;
        sa.I = I;
        sa.path_lo = zphase ? 0 : 1;
        sa.npaths = zphase ? npaths : 1;
        hipLaunchKernelGGL(dense_stage_a_kernel, dim3((sa.tr.hidden + 3) / 4, sa.npaths), dim3(256), 0, s, sa);
        hipLaunchKernelGGL(dense_stage_c_kernel, dim3(nblk, sa.npaths), dim3(CB), 0, s, sa);
    }
    if (want_kl && Tz == 0) (void)hipMemcpyAsync(z_kl, zbuf1, (size_t)I * sizeof(float), hipMemcpyDeviceToDevice, s);

    FinishArgs fa;
    fa.ldz = ldz; fa.ldr = ldr; fa.lq0 = lq0; fa.zr = zbuf1; fa.ldf = ldf; fa.scal = scal;
    fa.nblk = nblk; fa.nblk_i = nblk_i; fa.Tz = Tz; fa.Tr = want_kl ? Tr : 0; fa.I = I; fa.want_kl = want_kl;
    hipLaunchKernelGGL(dense_finish_kernel, dim3(1), dim3(64), 0, s, fa);
    return (int)hipGetLastError();
}
