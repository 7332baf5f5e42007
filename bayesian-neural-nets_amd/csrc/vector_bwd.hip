// Vector-sized backward of an MNF layer with planar flows (see include/lbbnn.h): two single-workgroup kernels.
// Both are latency-bound chains of block reductions over I (or O) elements; every thread owns the indices
// i = tid + k*1024 of each vector, so the per-index state in `work` is private to its thread and only the
// block sums (fixed-order, double accumulation) cross threads.
#include <cmath>
#include <cstdlib>
#include "lbbnn_device.h"
#include "lbbnn_internal.h"

namespace {

using namespace lbbnn;
constexpr int NT = 1024, NWV = NT / 64;

__device__ __forceinline__ void bsum3(double& a, double& b, double& c, double* scratch) {
    a = wave_sum(a); b = wave_sum(b); c = wave_sum(c);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) { scratch[w] = a; scratch[NWV + w] = b; scratch[2 * NWV + w] = c; }
    __syncthreads();
    double sa = 0, sb = 0, sc = 0;
#pragma unroll
    for (int i = 0; i < NWV; ++i) { sa += scratch[i]; sb += scratch[NWV + i]; sc += scratch[2 * NWV + i]; }
    a = sa; b = sb; c = sc;
}

__device__ __forceinline__ void mnf_aux_backward_body(const float* __restrict__ act_mu, const float* __restrict__ act_var,
                                                      const float* __restrict__ eps_act, const float* __restrict__ b1,
                                                      const float* __restrict__ b2, const float* zb_last, const float* g_kl,
                                                      int O, int I, float* da_mu, float* da_var, float* aux,
                                                      const uint64_t* rng, uint32_t layer) {
    __shared__ double scratch[3 * NWV];
    const int tid = threadIdx.x;
    uint64_t seed = 0, offs = 0;
    if (!eps_act) { seed = rng[0]; offs = rng[1]; }
    auto eps_at = [&](int o) -> float {
        if (eps_act) return eps_act[o];
        float n[4];
        philox_normal4(seed, offs, LBBNN_STREAM_EPS_ACT * 64u + layer, (uint64_t)(o >> 2), 0u, n);
        return n[o & 3];
    };
    double s_act = 0, z0 = 0, z1 = 0;
    for (int o = tid; o < O; o += NT) s_act += (double)tanhf(act_mu[o] + sqrtf(act_var[o]) * eps_at(o));
    bsum3(s_act, z0, z1, scratch);
    const float m = (float)(s_act / (double)O);
    const float zb = zb_last[0];
    double S = 0;
    for (int i = tid; i < I; i += NT) {
        const float e = expf(-b2[i] * m), dlt = zb - b1[i] * m;
        S += (double)(-0.5f * b2[i] + dlt * b1[i] * e + 0.5f * dlt * dlt * b2[i] * e);
    }
    bsum3(S, z0, z1, scratch);
    const float cm = g_kl[0] * (float)(-S) / (float)O;          // kl = ... - log_rb  =>  dkl/dm = -S
    for (int o = tid; o < O; o += NT) {
        const float sd = sqrtf(act_var[o]), e = eps_at(o);
        const float a = tanhf(act_mu[o] + sd * e);
        const float d = cm * (1.f - a * a);
        da_mu[o] = d;
        da_var[o] = d * e / (2.f * sd);
    }
    if (tid == 0) aux[0] = m;
}

__global__ __launch_bounds__(NT) void mnf_aux_backward_kernel(const float* __restrict__ act_mu, const float* __restrict__ act_var,
                                                              const float* __restrict__ eps_act, const float* __restrict__ b1,
                                                              const float* __restrict__ b2, const float* zb_last, const float* g_kl,
                                                              int O, int I, float* da_mu, float* da_var, float* aux,
                                                              const uint64_t* rng, uint32_t layer) {
    mnf_aux_backward_body(act_mu, act_var, eps_act, b1, b2, zb_last, g_kl, O, I, da_mu, da_var, aux, rng, layer);
}

// V1 of up to LBBNN_MAX_LAYERS layers in one launch (one workgroup per layer): every input is a by-product of the FORWARD, so
// once d loss / d kl is known all layers' V1 can run together instead of one ~7 us single-workgroup launch per layer.
struct AuxBatch { lbbnn_aux_bwd_args_t l[LBBNN_MAX_LAYERS]; };
__global__ __launch_bounds__(NT) void mnf_aux_backward_batch_kernel(const AuxBatch bt) {
    const LBBNN_CONST_AS lbbnn_aux_bwd_args_t& a = kernarg_as<AuxBatch>()->l[blockIdx.x];
    mnf_aux_backward_body(a.act_mu, a.act_var, a.eps_act, a.r0_b1, a.r0_b2, a.zb_last, a.g_kl, a.O, a.I, a.da_mu, a.da_var, a.aux,
                          a.rng, a.layer_id & 63u);
}

// Compact form of lbbnn_flow_bwd_args_t (flows of at most kBatchT transforms) so that the arguments of several layers fit
// one kernel-argument block: lbbnn_mnf_flow_planar_backward_batch runs one workgroup per layer in a single launch.
constexpr int kBatchT = 4;
struct PlanarSet4 { const float* u[kBatchT]; const float* w[kBatchT]; const float* b[kBatchT]; int T; };
struct PlanarGrad4 { float* u[kBatchT]; float* w[kBatchT]; float* b[kBatchT]; };
struct FlowBwdCompact {
    const float *q0_mean, *q0_log_var, *eps_fwd, *eps_kl, *r0_b1, *r0_b2, *aux, *dz_fwd, *dz_kl, *g_kl;
    const float *bias_mu, *bias_rho, *g_sum, *gv_sum;
    PlanarSet4 z_flow, r_flow;
    lbbnn_priors_t priors;
    float *d_q0_mean, *d_q0_log_var, *d_r0_b1, *d_r0_b2, *d_bias_mu, *d_bias_rho;
    PlanarGrad4 d_z_flow, d_r_flow;
    float* work;
    int O, I;
    const uint64_t* rng;
    uint32_t layer_id;
    int in_lds;
    int reg_form;                 // I <= 2 * NT: the register form (one reduction) instead of the LDS chain
};
struct FlowBwdBatch { FlowBwdCompact l[LBBNN_MAX_LAYERS]; };

template <typename A>
__device__ __forceinline__ void flow_planar_backward_body(const A& a, int in_lds, float* dyn) {
    __shared__ double scratch[3 * NWV];
    __shared__ float thF[LBBNN_MAX_FLOW_T], thK[LBBNN_MAX_FLOW_T], uwZ[LBBNN_MAX_FLOW_T], thR[LBBNN_MAX_FLOW_T], uwR[LBBNN_MAX_FLOW_T];
    __shared__ float s_zb;
    const int tid = threadIdx.x, I = a.I, O = a.O;
    const int Tz = a.z_flow.T, Tr = a.r_flow.T;
    const bool has_kl = a.g_kl != nullptr;
    const float G = has_kl ? a.g_kl[0] : 0.f;
    // work layout: ZF[0..Tz], ZK[0..Tz], R[1..Tr] (R[0] aliases ZK[Tz]), DK, DF, EF, EK -- in LDS when it fits (every
    // phase of the chain re-reads what the previous one wrote: an L2 round trip per phase otherwise), else in a.work
    float* const ZF = in_lds ? dyn : a.work;
    float* const ZK = ZF + (size_t)(Tz + 1) * I;
    float* const RR = ZK + (size_t)(Tz + 1) * I;
    float* const DK = RR + (size_t)Tr * I;
    float* const DF = DK + I;
    auto Rv = [&](int t) { return t == 0 ? ZK + (size_t)Tz * I : RR + (size_t)(t - 1) * I; };

    // ---- bias terms (LBBNN-GP-MF-MNF.py:197-198, 234-236)
    for (int o = tid; o < O; o += NT) {
        const float er = expf(a.bias_rho[o]);
        const float sb = log1pf(er), dsig = er / (1.f + er);
        float gm = a.g_sum[o], gs = a.gv_sum ? a.gv_sum[o] * 2.f * sb : 0.f;
        if (has_kl) {
            const float inv = 1.f / (a.priors.bias_sigma_prior * a.priors.bias_sigma_prior);
            gm += G * (a.bias_mu[o] - a.priors.bias_mu_prior) * inv;
            gs += G * (sb * inv - 1.f / sb);
        }
        a.d_bias_mu[o] = gm;
        a.d_bias_rho[o] = gs * dsig;
    }
    // ---- forward: z0 draws, z flow on both draws (draws explicit, or re-created from the forward's Philox state;
    // kept in DK / DF until the backward sweeps overwrite them, read back at the very end from EF / EK)
    uint64_t seed = 0, offs = 0;
    if (!a.eps_fwd) { seed = a.rng[0]; offs = a.rng[1]; }
    float* const EF = DF + I;
    float* const EK = EF + I;
    for (int i = tid; i < I; i += NT) {
        float ef, ek = 0.f;
        if (a.eps_fwd) { ef = a.eps_fwd[i]; if (has_kl) ek = a.eps_kl[i]; }
        else {
            float n[4];
            philox_normal4(seed, offs, LBBNN_STREAM_EPS_Z * 64u + a.layer_id, (uint64_t)(i >> 2), 0u, n); ef = n[i & 3];
            if (has_kl) { philox_normal4(seed, offs, LBBNN_STREAM_EPS_Z2 * 64u + a.layer_id, (uint64_t)(i >> 2), 0u, n); ek = n[i & 3]; }
        }
        EF[i] = ef; EK[i] = ek;
        const float sd = expf(0.5f * a.q0_log_var[i]);
        ZF[i] = a.q0_mean[i] + sd * ef;
        if (has_kl) ZK[i] = a.q0_mean[i] + sd * ek;
    }
    for (int t = 0; t < Tz; ++t) {
        const float *u = a.z_flow.u[t], *w = a.z_flow.w[t];
        const float* zf = ZF + (size_t)t * I;
        const float* zk = ZK + (size_t)t * I;
        double sf = 0, sk = 0, uw = 0;
        for (int i = tid; i < I; i += NT) {
            sf += (double)(w[i] * zf[i]);
            if (has_kl) sk += (double)(w[i] * zk[i]);
            uw += (double)(u[i] * w[i]);
        }
        bsum3(sf, sk, uw, scratch);
        const float b = a.z_flow.b[t][0];
        const float tf = tanhf((float)sf + b), tk = tanhf((float)sk + b);
        if (tid == 0) { thF[t] = tf; thK[t] = tk; uwZ[t] = (float)uw; }
        for (int i = tid; i < I; i += NT) {
            ZF[(size_t)(t + 1) * I + i] = zf[i] + u[i] * tf;
            if (has_kl) ZK[(size_t)(t + 1) * I + i] = zk[i] + u[i] * tk;
        }
    }
    if (has_kl) {
        // ---- forward: r flow on z2 = ZK[Tz]
        for (int t = 0; t < Tr; ++t) {
            const float *u = a.r_flow.u[t], *w = a.r_flow.w[t];
            const float* z = Rv(t);
            float* zn = Rv(t + 1);
            double sr = 0, uw = 0, z0 = 0;
            for (int i = tid; i < I; i += NT) { sr += (double)(w[i] * z[i]); uw += (double)(u[i] * w[i]); }
            bsum3(sr, uw, z0, scratch);
            const float th = tanhf((float)sr + a.r_flow.b[t][0]);
            if (tid == 0) { thR[t] = th; uwR[t] = (float)uw; }
            for (int i = tid; i < I; i += NT) zn[i] = z[i] + u[i] * th;
        }
        if (tid == ((I - 1) % NT)) s_zb = Rv(Tr)[I - 1];          // owner of the last element (quirk 2)
        __syncthreads();
        const float zb = s_zb, m = a.aux[0];
        // ---- log_rb: gradients of r0_b1 / r0_b2 and of zb
        double Szb = 0, z0 = 0, z1 = 0;
        for (int i = tid; i < I; i += NT) {
            const float e = expf(-a.r0_b2[i] * m), dlt = zb - a.r0_b1[i] * m;
            a.d_r0_b1[i] = -G * dlt * m * e;
            a.d_r0_b2[i] = -G * (-0.5f * m + 0.5f * dlt * dlt * m * e);
            Szb += (double)(dlt * e);
            DK[i] = 0.f;
        }
        bsum3(Szb, z0, z1, scratch);
        if (tid == ((I - 1) % NT)) DK[I - 1] = G * (float)Szb;    // -G * dlog_rb/dzb
        // ---- r flow backward (multiplier on log_det_r is -G)
        for (int t = Tr - 1; t >= 0; --t) {
            const float *u = a.r_flow.u[t], *w = a.r_flow.w[t];
            const float* z = Rv(t);
            double d1 = 0; z0 = 0; z1 = 0;
            for (int i = tid; i < I; i += NT) d1 += (double)(DK[i] * u[i]);
            bsum3(d1, z0, z1, scratch);
            const float th = thR[t], psi = 1.f - th * th, uw = uwR[t], D = 1.f + psi * uw;
            const float dth = (float)d1 + (-G) * (-2.f * th * uw) / D;
            const float din = dth * psi, c = (-G) * psi / D;
            float *du = a.d_r_flow.u[t], *dw = a.d_r_flow.w[t];
            for (int i = tid; i < I; i += NT) {
                const float dz = DK[i];
                du[i] = dz * th + c * w[i];
                dw[i] = din * z[i] + c * u[i];
                DK[i] = dz + din * w[i];
            }
            if (tid == 0) a.d_r_flow.b[t][0] = din;
        }
        for (int i = tid; i < I; i += NT) DK[i] += a.dz_kl ? a.dz_kl[i] : 0.f;
    } else {
        for (int i = tid; i < I; i += NT) {
            a.d_r0_b1[i] = 0.f; a.d_r0_b2[i] = 0.f; DK[i] = 0.f;
        }
        for (int t = 0; t < Tr; ++t) {
            for (int i = tid; i < I; i += NT) { a.d_r_flow.u[t][i] = 0.f; a.d_r_flow.w[t][i] = 0.f; }
            if (tid == 0) a.d_r_flow.b[t][0] = 0.f;
        }
    }
    for (int i = tid; i < I; i += NT) DF[i] = a.dz_fwd ? a.dz_fwd[i] : 0.f;
    __syncthreads();                                  // thF/thK/uwZ written by thread 0
    // ---- z flow backward on both draws (multiplier on log_det_q is -G; the forward draw's log-det is unused)
    for (int t = Tz - 1; t >= 0; --t) {
        const float *u = a.z_flow.u[t], *w = a.z_flow.w[t];
        const float* zf = ZF + (size_t)t * I;
        const float* zk = ZK + (size_t)t * I;
        double dk = 0, df = 0, z0 = 0;
        for (int i = tid; i < I; i += NT) { dk += (double)(DK[i] * u[i]); df += (double)(DF[i] * u[i]); }
        bsum3(dk, df, z0, scratch);
        const float uw = uwZ[t];
        const float tk = thK[t], psik = 1.f - tk * tk, Dk = 1.f + psik * uw;
        const float dthk = (float)dk + (-G) * (-2.f * tk * uw) / Dk;
        const float dink = has_kl ? dthk * psik : 0.f, ck = has_kl ? (-G) * psik / Dk : 0.f;
        const float tf = thF[t], dinf = (float)df * (1.f - tf * tf);
        float *du = a.d_z_flow.u[t], *dw = a.d_z_flow.w[t];
        for (int i = tid; i < I; i += NT) {
            const float dzk = DK[i], dzf = DF[i];
            du[i] = dzk * tk + ck * w[i] + dzf * tf;
            dw[i] = dink * (has_kl ? zk[i] : 0.f) + ck * u[i] + dinf * zf[i];
            DK[i] = dzk + dink * w[i];
            DF[i] = dzf + dinf * w[i];
        }
        if (tid == 0) a.d_z_flow.b[t][0] = dink + dinf;
    }
    // ---- q0 (LBBNN-GP-MF-MNF.py:183-185, 201-205): dlog_q0/dlog_var = -1/2 exactly, dlog_q0/dmean = 0
    for (int i = tid; i < I; i += NT) {
        const float sd = expf(0.5f * a.q0_log_var[i]);
        a.d_q0_mean[i] = DK[i] + DF[i];
        a.d_q0_log_var[i] = 0.5f * sd * (DK[i] * EK[i] + DF[i] * EF[i]) - 0.5f * G;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 3: the same chain in REGISTERS with ONE reduction (I <= 2 * NT, at most kBatchT transforms per flow).
//
// Every vector of the chain above is a combination of a few base vectors with scalar coefficients: zf_t = z0f + Sum_{s<t}
// thF_s u_s, DK = DK_0 + Sum dinr_s wr_s + dz_kl + Sum dink_s w_s, ...  So every inner product the chain takes (nine block
// reductions one after the other above, each a round trip through LDS and two barriers) is a combination of inner products of
// BASE vectors, and those do not depend on the chain:
//     A_wf[t] = w_t.z0f   A_wk[t] = w_t.z0k   X[a][b] = w_a.u_b   A_rk[t] = wr_t.z0k   C[a][b] = wr_a.u_b   R[a][b] = wr_a.ur_b
//     Dkl[t] = dz_kl.u_t   Dfw[t] = dz_fwd.u_t   S_e = Sum_i e^{-b2_i m}   S_b = Sum_i b1_i e^{-b2_i m}
// (Szb = Sum_i (zb - b1_i m) e^{-b2_i m} = zb S_e - m S_b needs no pass of its own).  One pass forms all of them -- float
// partial sums per thread and wave, the 16 waves combined in double in a fixed order -- wave 0 then walks the scalar chains
// (forward z on both draws, forward r, backward r, backward z: ~20 tanh / divisions in a row), and every thread writes the
// gradients of its <= 2 elements from the base vectors it still holds.  Three barriers instead of ~40.
constexpr int kRT = kBatchT;                      // transforms per flow in this form
constexpr int kEPT = 2;                           // elements per thread: I <= kEPT * NT
constexpr int NS_AWF = 0, NS_AWK = NS_AWF + kRT, NS_X = NS_AWK + kRT, NS_ARK = NS_X + kRT * kRT, NS_C = NS_ARK + kRT,
              NS_R = NS_C + kRT * kRT, NS_DKL = NS_R + kRT * kRT, NS_DFW = NS_DKL + kRT, NS_SE = NS_DFW + kRT, NS_SB = NS_SE + 1,
              NS_N = NS_SB + 1;                   // 70 sums
// coefficients wave 0 hands to everyone
constexpr int NC_THF = 0, NC_THK = NC_THF + kRT, NC_THR = NC_THK + kRT, NC_DINR = NC_THR + kRT, NC_CR = NC_DINR + kRT,
              NC_DINK = NC_CR + kRT, NC_CK = NC_DINK + kRT, NC_DINF = NC_CK + kRT, NC_ZB = NC_DINF + kRT, NC_SZB = NC_ZB + 1,
              NC_N = NC_SZB + 1;

template <typename A>
__device__ __forceinline__ void flow_planar_backward_reg_body(const A& a) {
    __shared__ float wsum[NS_N][NWV];
    __shared__ double tot[NS_N];
    __shared__ float coef[NC_N];
    __shared__ float lastv[1 + 2 * kRT];          // element I - 1 of z0k, u_s (z flow), ur_s (r flow)
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, I = a.I, O = a.O;
    const int Tz = a.z_flow.T, Tr = a.r_flow.T;
    const bool has_kl = a.g_kl != nullptr;
    const float G = has_kl ? a.g_kl[0] : 0.f;
    const float m = has_kl ? a.aux[0] : 0.f;

    // ---- bias terms (LBBNN-GP-MF-MNF.py:197-198, 234-236): independent of everything below
    for (int o = tid; o < O; o += NT) {
        const float er = expf(a.bias_rho[o]);
        const float sb = log1pf(er), dsig = er / (1.f + er);
        float gm = a.g_sum[o], gs = a.gv_sum ? a.gv_sum[o] * 2.f * sb : 0.f;
        if (has_kl) {
            const float inv = 1.f / (a.priors.bias_sigma_prior * a.priors.bias_sigma_prior);
            gm += G * (a.bias_mu[o] - a.priors.bias_mu_prior) * inv;
            gs += G * (sb * inv - 1.f / sb);
        }
        a.d_bias_mu[o] = gm;
        a.d_bias_rho[o] = gs * dsig;
    }

    // ---- base vectors of this thread's elements
    uint64_t seed = 0, offs = 0;
    if (!a.eps_fwd) { seed = a.rng[0]; offs = a.rng[1]; }
    float sd[kEPT], ef[kEPT], ek[kEPT], z0f[kEPT], z0k[kEPT], dzk[kEPT], dzf[kEPT], b1[kEPT], eb[kEPT];
    float u[kRT][kEPT], w[kRT][kEPT], ur[kRT][kEPT], wr[kRT][kEPT];
    float part[NS_N];
#pragma unroll
    for (int k = 0; k < NS_N; ++k) part[k] = 0.f;
#pragma unroll
    for (int e = 0; e < kEPT; ++e) {
        const int i = tid + e * NT;
        const bool in = i < I;
        sd[e] = ef[e] = ek[e] = z0f[e] = z0k[e] = dzk[e] = dzf[e] = b1[e] = eb[e] = 0.f;
#pragma unroll
        for (int t = 0; t < kRT; ++t) { u[t][e] = w[t][e] = ur[t][e] = wr[t][e] = 0.f; }
        if (in) {
            if (a.eps_fwd) { ef[e] = a.eps_fwd[i]; if (has_kl) ek[e] = a.eps_kl[i]; }
            else {
                float n[4];
                philox_normal4(seed, offs, LBBNN_STREAM_EPS_Z * 64u + a.layer_id, (uint64_t)(i >> 2), 0u, n); ef[e] = n[i & 3];
                if (has_kl) { philox_normal4(seed, offs, LBBNN_STREAM_EPS_Z2 * 64u + a.layer_id, (uint64_t)(i >> 2), 0u, n); ek[e] = n[i & 3]; }
            }
            const float qm = a.q0_mean[i];
            sd[e] = expf(0.5f * a.q0_log_var[i]);
            z0f[e] = qm + sd[e] * ef[e];
            if (has_kl) z0k[e] = qm + sd[e] * ek[e];
            dzf[e] = a.dz_fwd ? a.dz_fwd[i] : 0.f;
            if (has_kl) {
                dzk[e] = a.dz_kl ? a.dz_kl[i] : 0.f;
                b1[e] = a.r0_b1[i];
                eb[e] = expf(-a.r0_b2[i] * m);
                part[NS_SE] += eb[e];
                part[NS_SB] += b1[e] * eb[e];
            }
#pragma unroll
            for (int t = 0; t < kRT; ++t) {
                if (t < Tz) { u[t][e] = a.z_flow.u[t][i]; w[t][e] = a.z_flow.w[t][i]; }
                if (has_kl && t < Tr) { ur[t][e] = a.r_flow.u[t][i]; wr[t][e] = a.r_flow.w[t][i]; }
            }
            if (i == I - 1) {
                lastv[0] = z0k[e];
#pragma unroll
                for (int t = 0; t < kRT; ++t) { lastv[1 + t] = u[t][e]; lastv[1 + kRT + t] = ur[t][e]; }
            }
        }
#pragma unroll
        for (int t = 0; t < kRT; ++t) {
            part[NS_AWF + t] += w[t][e] * z0f[e];
            part[NS_AWK + t] += w[t][e] * z0k[e];
            part[NS_ARK + t] += wr[t][e] * z0k[e];
            part[NS_DKL + t] += dzk[e] * u[t][e];
            part[NS_DFW + t] += dzf[e] * u[t][e];
#pragma unroll
            for (int b = 0; b < kRT; ++b) {
                part[NS_X + t * kRT + b] += w[t][e] * u[b][e];
                part[NS_C + t * kRT + b] += wr[t][e] * u[b][e];
                part[NS_R + t * kRT + b] += wr[t][e] * ur[b][e];
            }
        }
    }
    // (only the sums this layer's flows use: 24 of the 70 at Tz = Tr = 2; the others are sums of zeros and stay zero)
    auto used = [&](int k) -> bool {
        if (k >= NS_SE) return has_kl;
        if (k >= NS_DFW) return k - NS_DFW < Tz;
        if (k >= NS_DKL) return has_kl && k - NS_DKL < Tz;
        if (k >= NS_R) return has_kl && (k - NS_R) / kRT < Tr && (k - NS_R) % kRT < Tr;
        if (k >= NS_C) return has_kl && (k - NS_C) / kRT < Tr && (k - NS_C) % kRT < Tz;
        if (k >= NS_ARK) return has_kl && k - NS_ARK < Tr;
        if (k >= NS_X) return (k - NS_X) / kRT < Tz && (k - NS_X) % kRT < Tz;
        if (k >= NS_AWK) return has_kl && k - NS_AWK < Tz;
        return k < Tz;
    };
#pragma unroll
    for (int k = 0; k < NS_N; ++k) {
        float v = 0.f;
        if (used(k)) v = wave_sum(part[k]);                            // uniform
        if (lane == 0) wsum[k][wv] = v;
    }
    __syncthreads();
    if (tid < NS_N) {
        double t2 = 0.0;
#pragma unroll
        for (int k = 0; k < NWV; ++k) t2 += (double)wsum[tid][k];
        tot[tid] = t2;
    }
    __syncthreads();

    // ---- the scalar chains (flows2.py:87-95 forward, its derivative backward): wave 0, every lane the same values;
    // tanh as the forward's K3v takes it (tanh_fast: one v_exp + one v_rcp)
    if (wv == 0) {
        auto S = [&](int k) -> float { return (float)tot[k]; };
        float thF[kRT], thK[kRT], thR[kRT], dinr[kRT], cr[kRT], dink[kRT], ck[kRT], dinf[kRT];
#pragma unroll
        for (int t = 0; t < kRT; ++t) thF[t] = thK[t] = thR[t] = dinr[t] = cr[t] = dink[t] = ck[t] = dinf[t] = 0.f;
#pragma unroll
        for (int t = 0; t < kRT; ++t)
            if (t < Tz) {
                double inf = tot[NS_AWF + t], ink = tot[NS_AWK + t];
#pragma unroll
                for (int s2 = 0; s2 < t; ++s2) { inf += (double)thF[s2] * tot[NS_X + t * kRT + s2]; ink += (double)thK[s2] * tot[NS_X + t * kRT + s2]; }
                const float b = a.z_flow.b[t][0];
                thF[t] = tanh_fast((float)inf + b);
                if (has_kl) thK[t] = tanh_fast((float)ink + b);
            }
        float zb = 0.f, szb = 0.f;
        if (has_kl) {
#pragma unroll
            for (int t = 0; t < kRT; ++t)
                if (t < Tr) {
                    double inr = tot[NS_ARK + t];
#pragma unroll
                    for (int s2 = 0; s2 < kRT; ++s2) if (s2 < Tz) inr += (double)thK[s2] * tot[NS_C + t * kRT + s2];
#pragma unroll
                    for (int s2 = 0; s2 < t; ++s2) inr += (double)thR[s2] * tot[NS_R + t * kRT + s2];
                    thR[t] = tanh_fast((float)inr + a.r_flow.b[t][0]);
                }
            zb = lastv[0];
#pragma unroll
            for (int t = 0; t < kRT; ++t) { if (t < Tz) zb += lastv[1 + t] * thK[t]; }
#pragma unroll
            for (int t = 0; t < kRT; ++t) { if (t < Tr) zb += lastv[1 + kRT + t] * thR[t]; }       // z_b[-1]: last ELEMENT (quirk 2)
            szb = (float)((double)zb * tot[NS_SE] - (double)m * tot[NS_SB]);
            const float dk_last = G * szb;                                                      // DK_0 = dk_last e_{I-1}
            // r flow backward (multiplier on log_det_r is -G)
#pragma unroll
            for (int t = kRT - 1; t >= 0; --t)
                if (t < Tr) {
                    double d1 = (double)dk_last * (double)lastv[1 + kRT + t];
#pragma unroll
                    for (int s2 = t + 1; s2 < kRT; ++s2) if (s2 < Tr) d1 += (double)dinr[s2] * tot[NS_R + s2 * kRT + t];
                    const float th = thR[t], psi = 1.f - th * th, uw = S(NS_R + t * kRT + t), D = 1.f + psi * uw;
                    const float dth = (float)d1 + (-G) * (-2.f * th * uw) / D;
                    dinr[t] = dth * psi;
                    cr[t] = (-G) * psi / D;
                }
            // z flow backward, KL draw
#pragma unroll
            for (int t = kRT - 1; t >= 0; --t)
                if (t < Tz) {
                    double dk = (double)dk_last * (double)lastv[1 + t] + tot[NS_DKL + t];
#pragma unroll
                    for (int s2 = 0; s2 < kRT; ++s2) if (s2 < Tr) dk += (double)dinr[s2] * tot[NS_C + s2 * kRT + t];
#pragma unroll
                    for (int s2 = t + 1; s2 < kRT; ++s2) if (s2 < Tz) dk += (double)dink[s2] * tot[NS_X + s2 * kRT + t];
                    const float uw = S(NS_X + t * kRT + t), tk = thK[t], psik = 1.f - tk * tk, Dk = 1.f + psik * uw;
                    const float dthk = (float)dk + (-G) * (-2.f * tk * uw) / Dk;
                    dink[t] = dthk * psik;
                    ck[t] = (-G) * psik / Dk;
                }
        }
        // z flow backward, forward draw (its log-det is unused)
#pragma unroll
        for (int t = kRT - 1; t >= 0; --t)
            if (t < Tz) {
                double df = tot[NS_DFW + t];
#pragma unroll
                for (int s2 = t + 1; s2 < kRT; ++s2) if (s2 < Tz) df += (double)dinf[s2] * tot[NS_X + s2 * kRT + t];
                dinf[t] = (float)df * (1.f - thF[t] * thF[t]);
            }
        if (lane == 0) {
#pragma unroll
            for (int t = 0; t < kRT; ++t) {
                coef[NC_THF + t] = thF[t]; coef[NC_THK + t] = thK[t]; coef[NC_THR + t] = thR[t]; coef[NC_DINR + t] = dinr[t];
                coef[NC_CR + t] = cr[t]; coef[NC_DINK + t] = dink[t]; coef[NC_CK + t] = ck[t]; coef[NC_DINF + t] = dinf[t];
                if (t < Tz) a.d_z_flow.b[t][0] = dink[t] + dinf[t];
                if (t < Tr) a.d_r_flow.b[t][0] = dinr[t];
            }
            coef[NC_ZB] = zb; coef[NC_SZB] = szb;
        }
    }
    __syncthreads();

    // ---- every thread: the gradients of its elements
    float thF[kRT], thK[kRT], thR[kRT], dinr[kRT], cr[kRT], dink[kRT], ck[kRT], dinf[kRT];
#pragma unroll
    for (int t = 0; t < kRT; ++t) {
        thF[t] = coef[NC_THF + t]; thK[t] = coef[NC_THK + t]; thR[t] = coef[NC_THR + t]; dinr[t] = coef[NC_DINR + t];
        cr[t] = coef[NC_CR + t]; dink[t] = coef[NC_DINK + t]; ck[t] = coef[NC_CK + t]; dinf[t] = coef[NC_DINF + t];
    }
    const float zb = coef[NC_ZB], szb = coef[NC_SZB];
#pragma unroll
    for (int e = 0; e < kEPT; ++e) {
        const int i = tid + e * NT;
        if (i >= I) continue;
        float DK = 0.f;
        if (has_kl) {
            const float dlt = zb - b1[e] * m;
            a.d_r0_b1[i] = -G * dlt * m * eb[e];
            a.d_r0_b2[i] = -G * (-0.5f * m + 0.5f * dlt * dlt * m * eb[e]);
            DK = (i == I - 1) ? G * szb : 0.f;
            // r_t[i] = z2[i] + Sum_{s<t} thR_s ur_s[i],  z2 = z0k + Sum_s thK_s u_s
            float z2 = z0k[e];
#pragma unroll
            for (int t = 0; t < kRT; ++t) z2 += u[t][e] * thK[t];                  // (u_t = 0, thK_t = 0 beyond Tz)
            float rt[kRT];
            {
                float r = z2;
#pragma unroll
                for (int t = 0; t < kRT; ++t) { rt[t] = r; r += ur[t][e] * thR[t]; }
            }
#pragma unroll
            for (int t = kRT - 1; t >= 0; --t)
                if (t < Tr) {
                    a.d_r_flow.u[t][i] = DK * thR[t] + cr[t] * wr[t][e];
                    a.d_r_flow.w[t][i] = dinr[t] * rt[t] + cr[t] * ur[t][e];
                    DK += dinr[t] * wr[t][e];
                }
            DK += dzk[e];
        } else {
            a.d_r0_b1[i] = 0.f; a.d_r0_b2[i] = 0.f;
#pragma unroll
            for (int t = 0; t < kRT; ++t) if (t < Tr) { a.d_r_flow.u[t][i] = 0.f; a.d_r_flow.w[t][i] = 0.f; }
        }
        float DF = dzf[e];
        float zft[kRT], zkt[kRT];
        {
            float f = z0f[e], k2 = z0k[e];
#pragma unroll
            for (int t = 0; t < kRT; ++t) { zft[t] = f; zkt[t] = k2; f += u[t][e] * thF[t]; k2 += u[t][e] * thK[t]; }
        }
#pragma unroll
        for (int t = kRT - 1; t >= 0; --t)
            if (t < Tz) {
                a.d_z_flow.u[t][i] = DK * thK[t] + ck[t] * w[t][e] + DF * thF[t];
                a.d_z_flow.w[t][i] = dink[t] * (has_kl ? zkt[t] : 0.f) + ck[t] * u[t][e] + dinf[t] * zft[t];
                DK += dink[t] * w[t][e];
                DF += dinf[t] * w[t][e];
            }
        // q0 (LBBNN-GP-MF-MNF.py:183-185, 201-205): dlog_q0/dlog_var = -1/2 exactly, dlog_q0/dmean = 0
        a.d_q0_mean[i] = DK + DF;
        a.d_q0_log_var[i] = 0.5f * sd[e] * (DK * ek[e] + DF * ef[e]) - 0.5f * G;
    }
    if (!has_kl && tid == 0) {
#pragma unroll
        for (int t = 0; t < kRT; ++t) if (t < Tr) a.d_r_flow.b[t][0] = 0.f;
    }
}

__global__ __launch_bounds__(NT) void mnf_flow_planar_backward_kernel(const lbbnn_flow_bwd_args_t a, int in_lds) {
    extern __shared__ __attribute__((aligned(16))) float dyn[];
    flow_planar_backward_body(a, in_lds, dyn);
}

__global__ __launch_bounds__(NT) void mnf_flow_planar_backward_batch_kernel(const FlowBwdBatch bt) {
    extern __shared__ __attribute__((aligned(16))) float dyn[];
    const LBBNN_CONST_AS FlowBwdCompact& a = kernarg_as<FlowBwdBatch>()->l[blockIdx.x];
    if (a.reg_form) flow_planar_backward_reg_body(a);                  // (uniform per workgroup)
    else flow_planar_backward_body(a, a.in_lds, dyn);
}

// The bias terms alone (an LRT layer has no other vector-sized parameters): the same arithmetic as the first block of the
// flow backward kernels above.
__global__ __launch_bounds__(256) void bias_backward_kernel(const float* bias_mu, const float* bias_rho, const float* g_sum,
                                                            const float* gv_sum, const float* g_kl, lbbnn_priors_t priors,
                                                            float* d_bias_mu, float* d_bias_rho, int O) {
    const int o = blockIdx.x * 256 + threadIdx.x;
    if (o >= O) return;
    const float er = expf(bias_rho[o]);
    const float sb = log1pf(er), dsig = er / (1.f + er);
    float gm = g_sum[o], gs = gv_sum ? gv_sum[o] * 2.f * sb : 0.f;
    if (g_kl) {
        const float G = g_kl[0], inv = 1.f / (priors.bias_sigma_prior * priors.bias_sigma_prior);
        gm += G * (bias_mu[o] - priors.bias_mu_prior) * inv;
        gs += G * (sb * inv - 1.f / sb);
    }
    d_bias_mu[o] = gm;
    d_bias_rho[o] = gs * dsig;
}

}  // namespace

extern "C" int lbbnn_bias_backward(const float* bias_mu, const float* bias_rho, const float* g_sum, const float* gv_sum,
                                   const float* g_kl, const lbbnn_priors_t* priors, float* d_bias_mu, float* d_bias_rho, int O,
                                   void* stream) {
    if (!bias_mu || !bias_rho || !g_sum || !priors || !d_bias_mu || !d_bias_rho) return LBBNN_E_NULL;
    if (O <= 0) return LBBNN_E_SHAPE;
    hipLaunchKernelGGL(bias_backward_kernel, dim3((O + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), bias_mu,
                       bias_rho, g_sum, gv_sum, g_kl, *priors, d_bias_mu, d_bias_rho, O);
    return (int)hipGetLastError();
}

extern "C" int lbbnn_mnf_aux_backward(const float* act_mu, const float* act_var, const float* eps_act, const float* r0_b1,
                                      const float* r0_b2, const float* zb_last, const float* g_kl, int O, int I,
                                      float* da_mu, float* da_var, float* aux, const uint64_t* rng, uint32_t layer_id, void* stream) {
    if (!act_mu || !act_var || !r0_b1 || !r0_b2 || !zb_last || !g_kl || !da_mu || !da_var || !aux) return LBBNN_E_NULL;
    if (!eps_act && !rng) return LBBNN_E_NOISE;
    if (O <= 0 || I <= 0) return LBBNN_E_SHAPE;
    hipLaunchKernelGGL(mnf_aux_backward_kernel, dim3(1), dim3(NT), 0, static_cast<hipStream_t>(stream), act_mu, act_var, eps_act,
                       r0_b1, r0_b2, zb_last, g_kl, O, I, da_mu, da_var, aux, rng, layer_id & 63u);
    return (int)hipGetLastError();
}

extern "C" int lbbnn_mnf_aux_backward_batch(const lbbnn_aux_bwd_args_t* args, int n, void* stream) {
    if (!args) return LBBNN_E_NULL;
    if (n <= 0 || n > LBBNN_MAX_LAYERS) return LBBNN_E_SHAPE;
    AuxBatch bt;
    for (int k = 0; k < n; ++k) {
        const lbbnn_aux_bwd_args_t& a = args[k];
        if (!a.act_mu || !a.act_var || !a.r0_b1 || !a.r0_b2 || !a.zb_last || !a.g_kl || !a.da_mu || !a.da_var || !a.aux) return LBBNN_E_NULL;
        if (!a.eps_act && !a.rng) return LBBNN_E_NOISE;
        if (a.O <= 0 || a.I <= 0) return LBBNN_E_SHAPE;
        bt.l[k] = a;
    }
    hipLaunchKernelGGL(mnf_aux_backward_batch_kernel, dim3(n), dim3(NT), 0, static_cast<hipStream_t>(stream), bt);
    return (int)hipGetLastError();
}

extern "C" int64_t lbbnn_mnf_flow_backward_workspace(int I, int Tz, int Tr) {
    if (I <= 0 || Tz < 0 || Tr < 0) return 0;
    return (int64_t)I * (2 * (Tz + 1) + Tr + 4);
}

extern "C" int lbbnn_mnf_flow_planar_backward_batch(const lbbnn_flow_bwd_args_t* args, int n, void* stream);

extern "C" int lbbnn_mnf_flow_planar_backward(const lbbnn_flow_bwd_args_t* p, void* stream) {
    if (!p) return LBBNN_E_NULL;
    const lbbnn_flow_bwd_args_t& a = *p;
    if (!a.eps_fwd && !a.rng) return LBBNN_E_NOISE;
    if (!a.q0_mean || !a.q0_log_var || !a.bias_mu || !a.bias_rho || !a.g_sum || !a.work ||
        !a.d_q0_mean || !a.d_q0_log_var || !a.d_r0_b1 || !a.d_r0_b2 || !a.d_bias_mu || !a.d_bias_rho) return LBBNN_E_NULL;
    if (a.g_kl && ((a.eps_fwd && !a.eps_kl) || !a.r0_b1 || !a.r0_b2 || !a.aux)) return LBBNN_E_NULL;
    if (a.O <= 0 || a.I <= 0) return LBBNN_E_SHAPE;
    if (a.z_flow.T < 0 || a.z_flow.T > LBBNN_MAX_FLOW_T || a.r_flow.T < 0 || a.r_flow.T > LBBNN_MAX_FLOW_T) return LBBNN_E_SHAPE;
    for (int t = 0; t < a.z_flow.T; ++t)
        if (!a.z_flow.u[t] || !a.z_flow.w[t] || !a.z_flow.b[t] || !a.d_z_flow.u[t] || !a.d_z_flow.w[t] || !a.d_z_flow.b[t]) return LBBNN_E_NULL;
    for (int t = 0; t < a.r_flow.T; ++t)
        if (!a.r_flow.u[t] || !a.r_flow.w[t] || !a.r_flow.b[t] || !a.d_r_flow.u[t] || !a.d_r_flow.w[t] || !a.d_r_flow.b[t]) return LBBNN_E_NULL;
    // the register form lives in the batch kernel: a chain it admits takes that kernel with n = 1, so that a layer's
    // gradients do not depend on whether its chain was deferred and batched (bitwise)
    if (a.I <= 2 * NT && a.z_flow.T <= kBatchT && a.r_flow.T <= kBatchT) return lbbnn_mnf_flow_planar_backward_batch(p, 1, stream);
    const size_t bytes = (size_t)lbbnn_mnf_flow_backward_workspace(a.I, a.z_flow.T, a.r_flow.T) * sizeof(float);
    const int in_lds = bytes <= 144 * 1024 ? 1 : 0;
    static size_t raised = 0;
    if (in_lds && bytes > 64 * 1024 && bytes > raised) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mnf_flow_planar_backward_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return (int)e;
        raised = bytes;
    }
    hipLaunchKernelGGL(mnf_flow_planar_backward_kernel, dim3(1), dim3(NT), in_lds ? bytes : 0, static_cast<hipStream_t>(stream),
                       a, in_lds);
    return (int)hipGetLastError();
}

// V2 of up to LBBNN_MAX_LAYERS layers as ONE launch (one workgroup per layer): the chains are latency-bound and independent,
// so n of them take the time of one.  Every flow must have at most 4 transforms (LBBNN_E_SHAPE otherwise: call the
// single-layer entry point per layer).
extern "C" int lbbnn_mnf_flow_planar_backward_batch(const lbbnn_flow_bwd_args_t* args, int n, void* stream) {
    if (!args) return LBBNN_E_NULL;
    if (n <= 0 || n > LBBNN_MAX_LAYERS) return LBBNN_E_SHAPE;
    FlowBwdBatch bt{};
    size_t dyn_bytes = 0;
    for (int k = 0; k < n; ++k) {
        const lbbnn_flow_bwd_args_t& a = args[k];
        if (!a.eps_fwd && !a.rng) return LBBNN_E_NOISE;
        if (!a.q0_mean || !a.q0_log_var || !a.bias_mu || !a.bias_rho || !a.g_sum || !a.work ||
            !a.d_q0_mean || !a.d_q0_log_var || !a.d_r0_b1 || !a.d_r0_b2 || !a.d_bias_mu || !a.d_bias_rho) return LBBNN_E_NULL;
        if (a.g_kl && ((a.eps_fwd && !a.eps_kl) || !a.r0_b1 || !a.r0_b2 || !a.aux)) return LBBNN_E_NULL;
        if (a.O <= 0 || a.I <= 0) return LBBNN_E_SHAPE;
        if (a.z_flow.T < 0 || a.z_flow.T > kBatchT || a.r_flow.T < 0 || a.r_flow.T > kBatchT) return LBBNN_E_SHAPE;
        FlowBwdCompact& c = bt.l[k];
        c.q0_mean = a.q0_mean; c.q0_log_var = a.q0_log_var; c.eps_fwd = a.eps_fwd; c.eps_kl = a.eps_kl;
        c.r0_b1 = a.r0_b1; c.r0_b2 = a.r0_b2; c.aux = a.aux; c.dz_fwd = a.dz_fwd; c.dz_kl = a.dz_kl; c.g_kl = a.g_kl;
        c.bias_mu = a.bias_mu; c.bias_rho = a.bias_rho; c.g_sum = a.g_sum; c.gv_sum = a.gv_sum;
        c.z_flow.T = a.z_flow.T; c.r_flow.T = a.r_flow.T;
        for (int t = 0; t < a.z_flow.T; ++t) {
            if (!a.z_flow.u[t] || !a.z_flow.w[t] || !a.z_flow.b[t] || !a.d_z_flow.u[t] || !a.d_z_flow.w[t] || !a.d_z_flow.b[t]) return LBBNN_E_NULL;
            c.z_flow.u[t] = a.z_flow.u[t]; c.z_flow.w[t] = a.z_flow.w[t]; c.z_flow.b[t] = a.z_flow.b[t];
            c.d_z_flow.u[t] = a.d_z_flow.u[t]; c.d_z_flow.w[t] = a.d_z_flow.w[t]; c.d_z_flow.b[t] = a.d_z_flow.b[t];
        }
        for (int t = 0; t < a.r_flow.T; ++t) {
            if (!a.r_flow.u[t] || !a.r_flow.w[t] || !a.r_flow.b[t] || !a.d_r_flow.u[t] || !a.d_r_flow.w[t] || !a.d_r_flow.b[t]) return LBBNN_E_NULL;
            c.r_flow.u[t] = a.r_flow.u[t]; c.r_flow.w[t] = a.r_flow.w[t]; c.r_flow.b[t] = a.r_flow.b[t];
            c.d_r_flow.u[t] = a.d_r_flow.u[t]; c.d_r_flow.w[t] = a.d_r_flow.w[t]; c.d_r_flow.b[t] = a.d_r_flow.b[t];
        }
        c.priors = a.priors;
        c.d_q0_mean = a.d_q0_mean; c.d_q0_log_var = a.d_q0_log_var; c.d_r0_b1 = a.d_r0_b1; c.d_r0_b2 = a.d_r0_b2;
        c.d_bias_mu = a.d_bias_mu; c.d_bias_rho = a.d_bias_rho; c.work = a.work; c.O = a.O; c.I = a.I;
        c.rng = a.rng; c.layer_id = a.layer_id;
        const size_t bytes = (size_t)lbbnn_mnf_flow_backward_workspace(a.I, a.z_flow.T, a.r_flow.T) * sizeof(float);
        c.in_lds = bytes <= 144 * 1024 ? 1 : 0;
        static const bool reg_off = getenv("LBBNN_V2_REG") && getenv("LBBNN_V2_REG")[0] == '0';      // A/B knob
        c.reg_form = (!reg_off && a.I <= 2 * NT) ? 1 : 0;
        if (!c.reg_form && c.in_lds && bytes > dyn_bytes) dyn_bytes = bytes;
    }
    static size_t raised = 0;
    if (dyn_bytes > 64 * 1024 && dyn_bytes > raised) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mnf_flow_planar_backward_batch_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn_bytes);
        if (e != hipSuccess) return (int)e;
        raised = dyn_bytes;
    }
    hipLaunchKernelGGL(mnf_flow_planar_backward_batch_kernel, dim3(n), dim3(NT), dyn_bytes, static_cast<hipStream_t>(stream), bt);
    return (int)hipGetLastError();
}
