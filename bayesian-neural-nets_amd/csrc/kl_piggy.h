// K5 (KL finalize) of every layer of a network + the network total as ONE 256-thread workgroup body, carried by an
// extra workgroup of a GEMM launch (lbbnn_lrt_gemm_finalize): the KL tail depends on parameters only, so it can run
// while the first GEMM runs instead of costing a launch of its own after the last one.
// Same arithmetic as kl_finalize_all_kernel (mnf_flow.hip), layer after layer; `sm` is the host kernel's dynamic LDS
// (at least piggy_lds_bytes() bytes): staged input vectors of one layer, then the reduction scratch.
#pragma once
#include "lbbnn_device.h"
#include "lbbnn_internal.h"

namespace lbbnn {

__host__ __device__ __forceinline__ int piggy_layer_floats(int O, int I, bool mnf) { return 6 * pad64(O) + 2 * pad64(mnf ? I : 1); }

inline size_t piggy_lds_bytes(const FinalizePiggy& p) {
    int mx = 0;
    for (int i = 0; i < p.n; ++i)
        if (p.active[i]) { const int v = piggy_layer_floats(p.l[i].O, p.l[i].I, p.l[i].scal != nullptr); mx = v > mx ? v : mx; }
    return (size_t)mx * sizeof(float) + 3 * 8 * sizeof(double) + LBBNN_MAX_LAYERS * sizeof(float) + 16;
}

// blockDim.x == NW * 64 (NW = 2 or 4 waves); every thread of the workgroup must call it
template <int NW>
__device__ __forceinline__ void kl_finalize_piggy(const LBBNN_CONST_AS FinalizePiggy& A, float* sm) {
    static_assert(NW >= 1 && NW <= 8, "scratch holds 8 partials per sum");
    constexpr int NT = NW * 64;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    float total = 0.f;
    for (int g = 0; g < A.n; ++g) {                                              // uniform
        if (!A.active[g]) continue;
        const LBBNN_CONST_AS FinalizeArgs& a = A.l[g];
        const bool mnf = a.scal != nullptr;
        const int O = a.O, I = a.I, PO = pad64(O), PI = pad64(mnf ? I : 1);
        float* klr = sm;            float* bmu = sm + PO;       float* brho = sm + 2 * PO;
        float* amu = sm + 3 * PO;   float* avar = sm + 4 * PO;  float* eact = sm + 5 * PO;
        float* b1 = sm + 6 * PO;    float* b2 = b1 + PI;
        double* scr = reinterpret_cast<double*>(b2 + PI + ((6 * PO + 2 * PI) & 1));   // 8-byte aligned ([3][4])
        dma_stage(klr, a.kl_rows, O);
        dma_stage(bmu, a.bias_mu, O);
        dma_stage(brho, a.bias_rho, O);
        if (mnf) {
            dma_stage(amu, a.act_mu, O);
            dma_stage(avar, a.act_var, O);
            if (a.eps_act) dma_stage(eact, a.eps_act, O);
            dma_stage(b1, a.r0_b1, I);
            dma_stage(b2, a.r0_b2, I);
        }
        uint64_t seed = 0, offs = 0;
        if (mnf && !a.eps_act) { seed = a.rng[0]; offs = a.rng[1]; }
        const float zb = mnf ? a.scal[3] : 0.f, ldq = mnf ? a.scal[0] : 0.f, lq0 = mnf ? a.scal[1] : 0.f, ldr = mnf ? a.scal[2] : 0.f;
        dma_wait_all();
        const float log_sp = __logf(a.bias_sigma_prior);
        const float inv_2sp2 = 1.f / (2.f * a.bias_sigma_prior * a.bias_sigma_prior);
        double s_rows = 0.0, s_bias = 0.0, s_act = 0.0;
        for (int o4 = 4 * t; o4 < O; o4 += 4 * NT) {                               // 4 consecutive outputs share one Philox call
            float n[4] = {0.f, 0.f, 0.f, 0.f};
            if (mnf && !a.eps_act) philox_normal4(seed, offs, LBBNN_STREAM_EPS_ACT * 64u + a.layer, (uint64_t)(o4 >> 2), 0u, n);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int o = o4 + k;
                if (o >= O) break;
                s_rows += (double)klr[o];
                const float sb = softplus_fast(brho[o]);
                const float d = bmu[o] - a.bias_mu_prior;
                s_bias += (double)((log_sp - __logf(sb)) - 0.5f + (sb * sb + d * d) * inv_2sp2);   // …LRT.py:185-186
                if (mnf) {
                    const float e = a.eps_act ? eact[o] : n[k];
                    s_act += (double)tanh_fast(amu[o] + sqrtf(avar[o]) * e);                    // …MNF.py:218-219
                }
            }
        }
        s_rows = wave_sum(s_rows); s_bias = wave_sum(s_bias); s_act = wave_sum(s_act);
        if (t < 24) scr[t] = 0.0;                                 // (slots of waves beyond NW stay zero)
        __syncthreads();
        if (lane == 0) { scr[w] = s_rows; scr[8 + w] = s_bias; scr[16 + w] = s_act; }
        __syncthreads();
        s_rows = ((scr[0] + scr[1]) + (scr[2] + scr[3])) + ((scr[4] + scr[5]) + (scr[6] + scr[7]));
        s_bias = ((scr[8] + scr[9]) + (scr[10] + scr[11])) + ((scr[12] + scr[13]) + (scr[14] + scr[15]));
        s_act = ((scr[16] + scr[17]) + (scr[18] + scr[19])) + ((scr[20] + scr[21]) + (scr[22] + scr[23]));
        double kl = s_bias + s_rows;
        if (mnf) {
            const float m = (float)(s_act / (double)O);          // outer(b, act).mean(-1) = b * mean(act)   :220-221
            double s_rb = 0.0;
            for (int i = t; i < I; i += NT) {
                const float mr = b1[i] * m, lv = b2[i] * m;
                const float d = zb - mr;
                s_rb += (double)(-0.5f * 1.1447298858494002f - 0.5f * lv - 0.5f * ((d * d) * __expf(-lv)));  // :223-224
            }
            s_rb = wave_sum(s_rb);
            __syncthreads();                                     // scr free again
            if (t < 8) scr[t] = 0.0;
            __syncthreads();
            if (lane == 0) scr[w] = s_rb;
            __syncthreads();
            s_rb = ((scr[0] + scr[1]) + (scr[2] + scr[3])) + ((scr[4] + scr[5]) + (scr[6] + scr[7]));
            kl += (-(double)ldq + (double)lq0) - ((double)ldr + s_rb);                          // :215,:225,:235
        }
        if (t == 0 && a.kl_layer) *a.kl_layer = (float)kl;
        total += (float)kl;                                      // fixed order: l1 + l2 + l3 (uniform across threads)
        __syncthreads();                                         // the next layer restages sm
    }
    if (t == 0 && A.total) *A.total = total;
    // the forward's RNG advance (lbbnn_lrt_gemm_finalize_adv): no kernel of this launch reads the LIVE offset
    if (t == 0 && A.rng_adv) A.rng_adv[1] += A.adv;
}

}  // namespace lbbnn
