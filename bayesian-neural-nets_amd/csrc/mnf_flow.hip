// K3 / K5 and small utilities of the MNF layer (gfx950).  All kernels are batched over layers
// (blockIdx.y / blockIdx.x selects the layer) so a network forward needs one launch per kind.
//
// K3  mnf_flow_planar : z sampling + planar normalizing flows + log_q0.
//     Work is O(T*I) (a few KB): launch/latency bound, so everything a layer needs from the flows
//     is done in ONE launch of two independent workgroups per layer (forward multiplier z_k | KL
//     branch z2 + r_flow).  Three forms: mnf_flow_planar_fast_kernel (chains of <= 4 transforms: one
//     reduction for the whole chain, 512 threads, inputs prefetched into LDS by LDS-DMA), the _lds_kernel
//     (longer chains, one reduction per transform) and the generic kernel (vectors too large for LDS).
//     Dot products are fixed-order sums (DPP + readlane inside a wave, LDS across waves) => deterministic.
// K5  kl_finalize     : O(O+I) tail of the KL: kl_bias, tanh/mean of the auxiliary activations,
//     log_rb, and the final scalar; kl_finalize_all_kernel does it for every layer of a network, adds the
//     network total and advances the RNG offset in one launch.
#include <cstdlib>
#include "lbbnn_device.h"
#include "lbbnn_internal.h"

namespace {

using namespace lbbnn;

typedef lbbnn_planar_flow_t PlanarSet;
struct FlowBatch { FlowArgs l[LBBNN_MAX_LAYERS];
                   // ensemble (fast kernel only): gridDim.z members, member m draws at offset rng[1] + m*m_adv, writes z_fwd + m*z_ms
                   unsigned long long m_adv; long long z_ms;
                   // register kernel only: fmt.x != NULL => gridDim.x = 2 + fmt_blocks, and the extra workgroups of layer row 0
                   // turn the network input into fp16 hi | lo planes (lbbnn_format_x's job) while the flow chains run
                   FormatJob fmt; int fmt_blocks; };
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


// ---- compact path (used whenever everything fits LDS): every vector the workgroup needs (q0 mean /
// log-var, the explicit draw, u and w of every transform) is prefetched into LDS by LDS-DMA with ALL
// loads issued before a single wait, and the arithmetic runs as small loops.  (A fully unrolled
// register-resident version was measured first: ~9000 straight-line instructions executed once made the
// kernel instruction-fetch bound at 13 us.)
constexpr int kFlowLdsBudget = 144 * 1024;

__device__ __forceinline__ void block_sum2(double& a, double& b, double* scratch) {
    a = wave_sum(a); b = wave_sum(b);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) { scratch[w] = a; scratch[4 + w] = b; }
    __syncthreads();
    a = (scratch[0] + scratch[1]) + (scratch[2] + scratch[3]);
    b = (scratch[4] + scratch[5]) + (scratch[6] + scratch[7]);
}

__global__ __launch_bounds__(256) void mnf_flow_planar_lds_kernel(const FlowBatch bt) {
    const LBBNN_CONST_AS FlowArgs& a = kernarg_as<FlowBatch>()->l[blockIdx.y];   // == bt.l[blockIdx.y], no scratch copy
    const bool klblk = blockIdx.x == 1;
    if (klblk && !a.want_kl) return;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    __shared__ double scratch[8];
    __shared__ float s_bias[2 * LBBNN_MAX_FLOW_T];
    const int I = a.I, P = pad64(I), tid = threadIdx.x;
    const float* eps = klblk ? a.eps_kl : a.eps_fwd;
    const int Tz = a.zf.T, Tr = klblk ? a.rf.T : 0;
    float* z = sm;                 // [P]
    float* qm = sm + P;            // [P]
    float* lv = sm + 2 * P;        // [P]
    float* ep = sm + 3 * P;        // [P]   (unused with Philox)
    float* uw = sm + 4 * P;        // (u, w) per transform: z-flow first, then r-flow
    dma_stage(qm, a.q0_mean, I);
    dma_stage(lv, a.q0_log_var, I);
    if (eps) dma_stage(ep, eps, I);
#pragma unroll 1
    for (int t = 0; t < Tz + Tr; ++t) {
        const float* u = t < Tz ? a.zf.u[t] : a.rf.u[t - Tz];
        const float* w = t < Tz ? a.zf.w[t] : a.rf.w[t - Tz];
        dma_stage(uw + (2 * t) * P, u, I);
        dma_stage(uw + (2 * t + 1) * P, w, I);
        if (tid == 0) s_bias[t] = (t < Tz ? a.zf.b[t] : a.rf.b[t - Tz])[0];     // fetched with the vectors, not on the chain
    }
    uint64_t seed = 0, offs = 0;
    if (!eps) { seed = a.rng[0]; offs = a.rng[1]; }
    const uint32_t stream = (klblk ? LBBNN_STREAM_EPS_Z2 : LBBNN_STREAM_EPS_Z) * 64u + a.layer;
    dma_wait_all();

    // z0 = q0_mean + exp(q0_log_var)^.5 * eps  (LBBNN-GP-MF-MNF.py:183-185); log_q0 (:213-214, -0.5*log(pi)).
    // A thread owns 4 consecutive elements so that one Philox call (4 normals) serves all of them.
    double lq0 = 0.0;
#pragma unroll 1
    for (int i0 = 4 * tid; i0 < I; i0 += 1024) {
        float n[4] = {0.f, 0.f, 0.f, 0.f};
        if (!eps) philox_normal4(seed, offs, stream, (uint64_t)(i0 >> 2), 0u, n);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = i0 + k;
            if (i < I) {
                const float e = eps ? ep[i] : n[k];
                const float ev = expf(lv[i]);
                const float z0 = qm[i] + sqrtf(ev) * e;
                z[i] = z0;
                if (klblk) {
                    const float d = z0 - qm[i];
                    lq0 += (double)(-0.5f * 1.1447298858494002f - 0.5f * lv[i] - 0.5f * ((d * d) / ev));
                }
            }
        }
    }
    __syncthreads();
    float ld_q = 0.f, ld_r = 0.f;
#pragma unroll 1
    for (int t = 0; t < Tz + Tr; ++t) {
        const float* u = uw + (2 * t) * P;
        const float* w = uw + (2 * t + 1) * P;
        const float bias = s_bias[t];
        double s_wz = 0.0, s_uw = 0.0;
#pragma unroll 1
        for (int i = tid; i < I; i += 256) { s_wz += (double)(w[i] * z[i]); s_uw += (double)(u[i] * w[i]); }
        block_sum2(s_wz, s_uw, scratch);
        const float th = tanhf((float)s_wz + bias);                       // flows2.py:87
        const float ld = logf(fabsf(1.f + (1.f - th * th) * (float)s_uw));  // flows2.py:89,95
        if (t < Tz) ld_q += ld; else ld_r += ld;
        if (t == Tz - 1) {
            // last z-flow transform: update and publish z (z_k or z2) in the same sweep
            float* zo = klblk ? a.z_kl : a.z_fwd;
#pragma unroll 1
            for (int i = tid; i < I; i += 256) { const float v = z[i] + u[i] * th; z[i] = v; zo[i] = v; }
        } else {
#pragma unroll 1
            for (int i = tid; i < I; i += 256) z[i] += u[i] * th;        // flows2.py:88
        }
        __syncthreads();
    }
    if (Tz == 0) {
        float* zo = klblk ? a.z_kl : a.z_fwd;
        for (int i = tid; i < I; i += 256) zo[i] = z[i];
    }
    if (!klblk) {
        if (tid == 0 && a.scal) a.scal[4] = ld_q;
        return;
    }
    double dummy = 0.0;
    block_sum2(lq0, dummy, scratch);
    if (tid == 0) {
        a.scal[0] = ld_q;
        a.scal[1] = (float)lq0;
        a.scal[2] = ld_r;
        a.scal[3] = z[I - 1];                                            // z_b[-1]: last ELEMENT (:224)
    }
}

// ---- short chains (Tz + Tr <= 4, the reference's num_transforms = 2 included): ONE reduction instead of one per
// transform.  A planar step only moves z along u:  z_t = z_0 + sum_{s<t} th_s u_s,  so
//     w_t . z_t = w_t . z_0 + sum_{s<t} th_s (w_t . u_s)
// and every dot product the whole chain needs -- w_t.z_0, w_t.u_s (s < t), u_t.w_t -- depends on the inputs only.
// They are accumulated in one sweep and reduced in ONE block reduction (<= 14 values + log_q0); the th_t / log-det
// chain is then scalar arithmetic, and z is written in one more sweep (terms added in transform order, so z is
// bit-identical to the step-by-step form; the r flow needs no sweep at all: only its log-dets and z_b[-1] are used).
constexpr int kFastT = 4;

constexpr int kFastThreads = 512, kFastWaves = kFastThreads / 64;

__global__ __launch_bounds__(kFastThreads) void mnf_flow_planar_fast_kernel(const FlowBatch bt) {
    const LBBNN_CONST_AS FlowArgs& a = kernarg_as<FlowBatch>()->l[blockIdx.y];   // == bt.l[blockIdx.y]
    const bool klblk = blockIdx.x == 1;
    if (klblk && !a.want_kl) return;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int NV = 2 * kFastT + kFastT * (kFastT - 1) / 2 + 1;      // 15 reduced values
    __shared__ double red[NV][kFastWaves];
    __shared__ float s_bias[kFastT];
    const int I = a.I, P = pad64(I), tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float* eps = klblk ? a.eps_kl : a.eps_fwd;
    const int Tz = a.zf.T, Tr = klblk ? a.rf.T : 0, NT = Tz + Tr;
    float* z = sm;  float* qm = sm + P;  float* lv = sm + 2 * P;  float* ep = sm + 3 * P;  float* uw = sm + 4 * P;
    dma_stage(qm, a.q0_mean, I);
    dma_stage(lv, a.q0_log_var, I);
    if (eps) dma_stage(ep, eps, I);
#pragma unroll 1
    for (int t = 0; t < NT; ++t) {
        dma_stage(uw + (2 * t) * P, t < Tz ? a.zf.u[t] : a.rf.u[t - Tz], I);
        dma_stage(uw + (2 * t + 1) * P, t < Tz ? a.zf.w[t] : a.rf.w[t - Tz], I);
        if (tid == 0) s_bias[t] = (t < Tz ? a.zf.b[t] : a.rf.b[t - Tz])[0];
    }
    uint64_t seed = 0, offs = 0;
    if (!eps) { seed = a.rng[0]; offs = a.rng[1] + (uint64_t)blockIdx.z * kernarg_as<FlowBatch>()->m_adv; }
    const uint32_t stream = (klblk ? LBBNN_STREAM_EPS_Z2 : LBBNN_STREAM_EPS_Z) * 64u + a.layer;
    dma_wait_all();

    // one sweep: z0 (LBBNN-GP-MF-MNF.py:183-185), log_q0 (:213-214) and every dot product of the chain
    double acc[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) acc[k] = 0.0;
#pragma unroll 1
    for (int i = tid; i < I; i += kFastThreads) {      // one element per thread: the Philox call (4 normals) is recomputed
        float e;                                       // by the 4 threads sharing it -- parallel, so free on this chain
        if (eps) e = ep[i];
        else { float n[4]; philox_normal4(seed, offs, stream, (uint64_t)(i >> 2), 0u, n); e = n[i & 3]; }
        const float ev = expf(lv[i]);
        const float z0 = qm[i] + sqrtf(ev) * e;
        z[i] = z0;
        if (klblk) {
            const float d = z0 - qm[i];
            acc[NV - 1] += (double)(-0.5f * 1.1447298858494002f - 0.5f * lv[i] - 0.5f * ((d * d) / ev));
        }
        float u[kFastT], w[kFastT];
#pragma unroll
        for (int t = 0; t < kFastT; ++t) { u[t] = t < NT ? uw[(2 * t) * P + i] : 0.f; w[t] = t < NT ? uw[(2 * t + 1) * P + i] : 0.f; }
        int q = 2 * kFastT;
#pragma unroll
        for (int t = 0; t < kFastT; ++t) {
            acc[t] += (double)(w[t] * z0);
            acc[kFastT + t] += (double)(u[t] * w[t]);
#pragma unroll
            for (int s2 = 0; s2 < t; ++s2) acc[q++] += (double)(w[t] * u[s2]);
        }
    }
    // fixed-order reduction of the NV values: wave sums, then the per-wave partials through LDS in wave order
#pragma unroll
    for (int k = 0; k < NV; ++k) acc[k] = wave_sum(acc[k]);
    if (lane == 0)
#pragma unroll
        for (int k = 0; k < NV; ++k) red[k][wv] = acc[k];
    __syncthreads();                                              // also publishes z0 and s_bias
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        double t2 = 0.0;
#pragma unroll
        for (int w2 = 0; w2 < kFastWaves; ++w2) t2 += red[k][w2];
        acc[k] = t2;
    }

    // scalar chain (flows2.py:87-95), every thread redundantly
    float th[kFastT], ld_q = 0.f, ld_r = 0.f;
    {
        int q = 2 * kFastT;
#pragma unroll
        for (int t = 0; t < kFastT; ++t) {
            double inner = acc[t];
#pragma unroll
            for (int s2 = 0; s2 < t; ++s2) inner += (double)th[s2] * acc[q++];
            th[t] = 0.f;
            if (t < NT) {
                th[t] = tanhf((float)inner + s_bias[t]);
                const float ld = logf(fabsf(1.f + (1.f - th[t] * th[t]) * (float)acc[kFastT + t]));
                if (t < Tz) ld_q += ld; else ld_r += ld;
            }
        }
    }
    // z after the z flow (the layer's z_k / z2); the r-flow steps only add to the scalar z_b[-1]
    float* zo = (klblk ? a.z_kl : a.z_fwd) + (long long)blockIdx.z * kernarg_as<FlowBatch>()->z_ms;
#pragma unroll 1
    for (int i = tid; i < I; i += kFastThreads) {
        float v = z[i];
#pragma unroll
        for (int t = 0; t < kFastT; ++t) if (t < Tz) v += uw[(2 * t) * P + i] * th[t];
        zo[i] = v;
        if (klblk && i == I - 1) {
#pragma unroll
            for (int t = 0; t < kFastT; ++t) if (t >= Tz && t < NT) v += uw[(2 * t) * P + i] * th[t];
            a.scal[3] = v;                                                   // z_b[-1]: last ELEMENT (:224)
        }
    }
    if (tid == 0) {
        if (!klblk) { if (a.scal) a.scal[4] = ld_q; }
        else { a.scal[0] = ld_q; a.scal[1] = (float)acc[NV - 1]; a.scal[2] = ld_r; }
    }
}

// ---- round 3: the same short-chain form with everything in REGISTERS (rows of I % 4 == 0 <= 2048 floats, 16-B aligned
// vectors: every layer of the BASELINE configurations).  What the LDS form above spends its 12.7 us on (flow_stamps, round
// 2: inputs landed 3.8 us, sweep done 6.4 us, end 11.1 us) is a chain of dependent latencies on ONE workgroup: LDS-DMA
// staging + wait + LDS re-reads, one Philox call PER ELEMENT (four threads recompute the same counter), 15 double-precision
// accumulators reduced by 6-step DPP chains, libm tanhf.  Here a thread owns one float4 column group: its 2 + 2 NT input
// float4s are requested back to back straight into registers (one memory latency, no LDS round trip), ONE Philox call
// gives its four draws, the 15 partial sums are floats over four elements, reduced per wave by DPP in float and combined
// across the 8 waves in double in a fixed order (deterministic), tanh through one hardware exp (tanh_fast: ~1e-7
// absolute), z written as float4.  Same draws, same z0 bits; the dot products differ from the double-accumulated form by
// ~1e-7 relative.
constexpr int kVecThreads = 512;

__global__ __launch_bounds__(kVecThreads) void mnf_flow_planar_vec_kernel(const FlowBatch bt) {
    if (blockIdx.x >= 2) {
        // The flow chains are two latency-bound workgroups per layer on a 256-CU chip; the x-format job (25.6 MB of traffic
        // for the headline batch, ~7 us as a launch of its own) runs on the CUs they leave idle, inside the same launch.
        if (blockIdx.y == 0 && blockIdx.z == 0)
            format_x_items(bt.fmt, (size_t)(blockIdx.x - 2) * kVecThreads + threadIdx.x, (size_t)bt.fmt_blocks * kVecThreads);
        return;
    }
    const LBBNN_CONST_AS FlowArgs& a = kernarg_as<FlowBatch>()->l[blockIdx.y];
    const bool klblk = blockIdx.x == 1;
    if (klblk && !a.want_kl) return;
    constexpr int NV = 2 * kFastT + kFastT * (kFastT - 1) / 2 + 1;      // 15 reduced values
    constexpr int NW = kVecThreads / 64;
    __shared__ float red[NV][NW];
    const int I = a.I, nq = I >> 2, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const bool on = tid < nq;
    const float* eps = klblk ? a.eps_kl : a.eps_fwd;
    const int Tz = a.zf.T, Tr = klblk ? a.rf.T : 0, NT = Tz + Tr;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 qm = zero4, lv = zero4, ep = zero4, u[kFastT], w[kFastT];
    float bias[kFastT];
#pragma unroll
    for (int t = 0; t < kFastT; ++t) { u[t] = zero4; w[t] = zero4; bias[t] = 0.f; }
    if (on) {
        qm = reinterpret_cast<const float4*>(a.q0_mean)[tid];
        lv = reinterpret_cast<const float4*>(a.q0_log_var)[tid];
        if (eps) ep = reinterpret_cast<const float4*>(eps)[tid];
    }
#pragma unroll
    for (int t = 0; t < kFastT; ++t) {
        if (t < NT) {                                                          // uniform
            const float* up = t < Tz ? a.zf.u[t] : a.rf.u[t - Tz];
            const float* wp = t < Tz ? a.zf.w[t] : a.rf.w[t - Tz];
            if (on) { u[t] = reinterpret_cast<const float4*>(up)[tid]; w[t] = reinterpret_cast<const float4*>(wp)[tid]; }
            bias[t] = (t < Tz ? a.zf.b[t] : a.rf.b[t - Tz])[0];
        }
    }
    float e[4] = {ep.x, ep.y, ep.z, ep.w};
    if (!eps && on) {
        const uint64_t seed = a.rng[0], offs = a.rng[1] + (uint64_t)blockIdx.z * kernarg_as<FlowBatch>()->m_adv;
        const uint32_t stream = (klblk ? LBBNN_STREAM_EPS_Z2 : LBBNN_STREAM_EPS_Z) * 64u + a.layer;
        philox_normal4(seed, offs, stream, (uint64_t)tid, 0u, e);
    }
    const float qmv[4] = {qm.x, qm.y, qm.z, qm.w}, lvv[4] = {lv.x, lv.y, lv.z, lv.w};
    float z0[4] = {0.f, 0.f, 0.f, 0.f};
    float acc[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) acc[k] = 0.f;
    if (on) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            // (hardware exp2 with the product's residual put back, v_sqrt_f32, v_rcp_f32: ~1 ulp each, a quarter of the
            // instructions of expf / sqrtf / a true division on this kernel's one critical chain)
            const float ev = k1_exp_acc(lvv[c]);
            z0[c] = qmv[c] + sqrt_hw(ev) * e[c];                                                 // ...MNF.py:183-185
            if (klblk) {
                const float d = z0[c] - qmv[c];
                acc[NV - 1] += -0.5f * 1.1447298858494002f - 0.5f * lvv[c] - 0.5f * ((d * d) * __builtin_amdgcn_rcpf(ev));   // :213-214
            }
        }
        const float4 z4 = make_float4(z0[0], z0[1], z0[2], z0[3]);
        int qx = 2 * kFastT;
#pragma unroll
        for (int t = 0; t < kFastT; ++t) {
            acc[t] = (w[t].x * z4.x + w[t].y * z4.y) + (w[t].z * z4.z + w[t].w * z4.w);
            acc[kFastT + t] = (u[t].x * w[t].x + u[t].y * w[t].y) + (u[t].z * w[t].z + u[t].w * w[t].w);
#pragma unroll
            for (int s2 = 0; s2 < t; ++s2)
                acc[qx++] = (w[t].x * u[s2].x + w[t].y * u[s2].y) + (w[t].z * u[s2].z + w[t].w * u[s2].w);
        }
    }
#pragma unroll
    for (int k = 0; k < NV; ++k) acc[k] = wave_sum(acc[k]);
    if (lane == 0)
#pragma unroll
        for (int k = 0; k < NV; ++k) red[k][wv] = acc[k];
    __syncthreads();
    // the NW partials of every sum combined ONCE (threads 0 .. NV-1, double, fixed order), then read by everyone: the first
    // version had every thread add all NV x NW values itself (120 LDS reads + 120 double additions on the critical chain)
    __shared__ double tots[NV];
    if (tid < NV) {
        double t2 = 0.0;
#pragma unroll
        for (int w2 = 0; w2 < NW; ++w2) t2 += (double)red[tid][w2];
        tots[tid] = t2;
    }
    __syncthreads();
    double tot[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) tot[k] = tots[k];
    // scalar chain (flows2.py:87-95), every thread redundantly
    float th[kFastT], ld_q = 0.f, ld_r = 0.f;
    {
        int qx = 2 * kFastT;
#pragma unroll
        for (int t = 0; t < kFastT; ++t) {
            double inner = tot[t];
#pragma unroll
            for (int s2 = 0; s2 < t; ++s2) inner += (double)th[s2] * tot[qx++];
            th[t] = 0.f;
            if (t < NT) {
                th[t] = tanh_fast((float)inner + bias[t]);
                const float ld = 0.6931471805599453f * __builtin_amdgcn_logf(fabsf(1.f + (1.f - th[t] * th[t]) * (float)tot[kFastT + t]));
                if (t < Tz) ld_q += ld; else ld_r += ld;
            }
        }
    }
    float* zo = (klblk ? a.z_kl : a.z_fwd) + (long long)blockIdx.z * kernarg_as<FlowBatch>()->z_ms;
    if (on) {
        float v[4] = {z0[0], z0[1], z0[2], z0[3]};
#pragma unroll
        for (int t = 0; t < kFastT; ++t)
            if (t < Tz) { v[0] += u[t].x * th[t]; v[1] += u[t].y * th[t]; v[2] += u[t].z * th[t]; v[3] += u[t].w * th[t]; }
        reinterpret_cast<float4*>(zo)[tid] = make_float4(v[0], v[1], v[2], v[3]);
        if (klblk && tid == nq - 1) {
            float zl = v[3];
#pragma unroll
            for (int t = 0; t < kFastT; ++t) if (t >= Tz && t < NT) zl += u[t].w * th[t];
            a.scal[3] = zl;                                                  // z_b[-1]: last ELEMENT (:224)
        }
    }
    if (tid == 0) {
        if (!klblk) { if (a.scal) a.scal[4] = ld_q; }
        else { a.scal[0] = ld_q; a.scal[1] = (float)tot[NV - 1]; a.scal[2] = ld_r; }
    }
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

// K5, compact form: the eight input vectors are prefetched into LDS by LDS-DMA (all loads issued, one
// wait), then reduced from LDS.  Used when they fit; otherwise kl_finalize_kernel above reads from global.
__global__ __launch_bounds__(256) void kl_finalize_lds_kernel(const FinalizeBatch bt) {
    const LBBNN_CONST_AS FinalizeArgs& a = kernarg_as<FinalizeBatch>()->l[blockIdx.x];
    extern __shared__ __attribute__((aligned(16))) float sm[];
    __shared__ double scratch[8];
    const bool mnf = a.scal != nullptr;
    const int O = a.O, I = a.I, PO = pad64(O), PI = pad64(mnf ? I : 1), tid = threadIdx.x;
    float* klr = sm;            float* bmu = sm + PO;       float* brho = sm + 2 * PO;
    float* amu = sm + 3 * PO;   float* avar = sm + 4 * PO;  float* eact = sm + 5 * PO;
    float* b1 = sm + 6 * PO;    float* b2 = b1 + PI;
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

    double s_rows = 0.0, s_bias = 0.0, s_act = 0.0;
#pragma unroll 1
    for (int o = tid; o < O; o += 256) {
        s_rows += (double)klr[o];
        const float sb = softplus_ref(brho[o]);
        const float d = bmu[o] - a.bias_mu_prior;
        const float sp = a.bias_sigma_prior;
        s_bias += (double)(logf(sp / sb) - 0.5f + (sb * sb + d * d) / (2.f * sp * sp));       // …LRT.py:185-186
        if (mnf) {
            float e;
            if (a.eps_act) e = eact[o];
            else { float n[4]; philox_normal4(seed, offs, LBBNN_STREAM_EPS_ACT * 64u + a.layer, (uint64_t)(o >> 2), 0u, n); e = n[o & 3]; }
            s_act += (double)tanhf(amu[o] + sqrtf(avar[o]) * e);                               // …MNF.py:218-219
        }
    }
    block_sum2(s_rows, s_bias, scratch);
    double kl = s_bias + s_rows;
    if (mnf) {
        double s_rb = 0.0, dummy = 0.0;
        block_sum2(s_act, dummy, scratch);
        const float m = (float)(s_act / (double)O);          // outer(b, act).mean(-1) = b * mean(act)   :220-221
#pragma unroll 1
        for (int i = tid; i < I; i += 256) {
            const float mr = b1[i] * m, lvr = b2[i] * m;
            const float d = zb - mr;
            s_rb += (double)(-0.5f * 1.1447298858494002f - 0.5f * lvr - 0.5f * ((d * d) / expf(lvr)));  // :223-224
        }
        block_sum2(s_rb, dummy, scratch);
        kl += (-(double)ldq + (double)lq0) - ((double)ldr + s_rb);                             // :215,:225,:235
    }
    if (tid == 0) {
        const float k = (float)kl;
        if (a.kl_layer) *a.kl_layer = k;
        if (a.kl_out) *a.kl_out = a.accum ? (*a.kl_out + k) : k;
    }
}

// K5 of every layer + the network total + the RNG advance in ONE single-workgroup launch at the end of a forward
// (lbbnn_layers_finalize): thread group g = tid / 256 finalizes layer g exactly as kl_finalize_kernel does; the groups
// meet at block-wide barriers (every group runs the same barrier sequence), thread 0 adds the layer KLs in layer
// order and bumps the Philox offset after every group has read it.
struct FinalizeAllArgs { FinalizeArgs l[LBBNN_MAX_LAYERS]; int active[LBBNN_MAX_LAYERS]; int n; uint64_t* rng; uint64_t advance; float* total; };

__device__ __forceinline__ double group_sum(double v, double (*scr)[4], int g, int lane, int w) {
    v = wave_sum(v);
    __syncthreads();                         // protect scr from the previous use
    if (lane == 0) scr[g][w] = v;
    __syncthreads();
    return (scr[g][0] + scr[g][1]) + (scr[g][2] + scr[g][3]);
}
// three sums behind one pair of barriers
__device__ __forceinline__ void group_sum3(double& a, double& b, double& c, double (*scr)[3][4], int g, int lane, int w) {
    a = wave_sum(a); b = wave_sum(b); c = wave_sum(c);
    __syncthreads();
    if (lane == 0) { scr[g][0][w] = a; scr[g][1][w] = b; scr[g][2][w] = c; }
    __syncthreads();
    a = (scr[g][0][0] + scr[g][0][1]) + (scr[g][0][2] + scr[g][0][3]);
    b = (scr[g][1][0] + scr[g][1][1]) + (scr[g][1][2] + scr[g][1][3]);
    c = (scr[g][2][0] + scr[g][2][1]) + (scr[g][2][2] + scr[g][2][3]);
}

__global__ __launch_bounds__(LBBNN_MAX_LAYERS * 256) void kl_finalize_all_kernel(const FinalizeAllArgs fa, int staged) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    __shared__ double scr[LBBNN_MAX_LAYERS][4];
    __shared__ double scr3[LBBNN_MAX_LAYERS][3][4];
    __shared__ float s_kl[LBBNN_MAX_LAYERS];
    const LBBNN_CONST_AS FinalizeAllArgs& A = *kernarg_as<FinalizeAllArgs>();
    const int g = threadIdx.x >> 8, t = threadIdx.x & 255, lane = t & 63, w = t >> 6;
    // ---- every input vector of every layer is prefetched into LDS by ALL waves (LDS-DMA, one wait): the dependent
    // global loads of the loops below were what made the first single-launch version slower than three workgroups
    const float *klr[LBBNN_MAX_LAYERS], *bmu[LBBNN_MAX_LAYERS], *brho[LBBNN_MAX_LAYERS], *amu[LBBNN_MAX_LAYERS];
    const float *avar[LBBNN_MAX_LAYERS], *eact[LBBNN_MAX_LAYERS], *b1[LBBNN_MAX_LAYERS], *b2[LBBNN_MAX_LAYERS];
    {
        float* p = sm;
#pragma unroll
        for (int li = 0; li < LBBNN_MAX_LAYERS; ++li) {
            const LBBNN_CONST_AS FinalizeArgs& a = A.l[li];
            const bool on = li < A.n && A.active[li] != 0, mnf = on && a.scal != nullptr;
            klr[li] = a.kl_rows; bmu[li] = a.bias_mu; brho[li] = a.bias_rho; amu[li] = a.act_mu; avar[li] = a.act_var;
            eact[li] = a.eps_act; b1[li] = a.r0_b1; b2[li] = a.r0_b2;
            if (!staged || !on) continue;
            const int PO = pad64(a.O), PI = pad64(a.I);
            dma_stage(p, a.kl_rows, a.O); klr[li] = p; p += PO;
            dma_stage(p, a.bias_mu, a.O); bmu[li] = p; p += PO;
            dma_stage(p, a.bias_rho, a.O); brho[li] = p; p += PO;
            if (mnf) {
                dma_stage(p, a.act_mu, a.O); amu[li] = p; p += PO;
                dma_stage(p, a.act_var, a.O); avar[li] = p; p += PO;
                if (a.eps_act) { dma_stage(p, a.eps_act, a.O); eact[li] = p; p += PO; }
                dma_stage(p, a.r0_b1, a.I); b1[li] = p; p += PI;
                dma_stage(p, a.r0_b2, a.I); b2[li] = p; p += PI;
            }
        }
    }
    const LBBNN_CONST_AS FinalizeArgs& a = A.l[g];
    const bool on = g < A.n && A.active[g] != 0;
    const bool mnf = on && a.scal != nullptr;
    uint64_t seed = 0, offs = 0;
    if (mnf && !a.eps_act) { seed = a.rng[0]; offs = a.rng[1]; }
    const float zb = mnf ? a.scal[3] : 0.f, ldq = mnf ? a.scal[0] : 0.f, lq0 = mnf ? a.scal[1] : 0.f, ldr = mnf ? a.scal[2] : 0.f;
    if (staged) dma_wait_all();
    // select this group's vectors (constant-index chain: no dynamic indexing of the pointer arrays)
    const float *Gklr = klr[0], *Gbmu = bmu[0], *Gbrho = brho[0], *Gamu = amu[0], *Gavar = avar[0], *Geact = eact[0], *Gb1 = b1[0], *Gb2 = b2[0];
#pragma unroll
    for (int li = 1; li < LBBNN_MAX_LAYERS; ++li)
        if (g == li) { Gklr = klr[li]; Gbmu = bmu[li]; Gbrho = brho[li]; Gamu = amu[li]; Gavar = avar[li]; Geact = eact[li]; Gb1 = b1[li]; Gb2 = b2[li]; }
    // this single workgroup is issue-bound (12 waves on one CU): hardware exp / log / rcp forms, see lbbnn_device.h
    const float log_sp = on ? __logf(a.bias_sigma_prior) : 0.f;
    const float inv_2sp2 = on ? 1.f / (2.f * a.bias_sigma_prior * a.bias_sigma_prior) : 0.f;
    double s_rows = 0.0, s_bias = 0.0, s_act = 0.0;
    if (on)
        for (int o4 = 4 * t; o4 < a.O; o4 += 1024) {                  // 4 consecutive outputs share one Philox call
            float n[4] = {0.f, 0.f, 0.f, 0.f};
            if (mnf && !a.eps_act) philox_normal4(seed, offs, LBBNN_STREAM_EPS_ACT * 64u + a.layer, (uint64_t)(o4 >> 2), 0u, n);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int o = o4 + k;
                if (o >= a.O) break;
                s_rows += (double)Gklr[o];
                const float sb = softplus_fast(Gbrho[o]);
                const float d = Gbmu[o] - a.bias_mu_prior;
                s_bias += (double)((log_sp - __logf(sb)) - 0.5f + (sb * sb + d * d) * inv_2sp2);  // …LRT.py:185-186
                if (mnf) {
                    const float e = a.eps_act ? Geact[o] : n[k];
                    s_act += (double)tanh_fast(Gamu[o] + sqrtf(Gavar[o]) * e);                   // …MNF.py:218-219
                }
            }
        }
    group_sum3(s_rows, s_bias, s_act, scr3, g, lane, w);
    double kl = s_bias + s_rows;
    double s_rb = 0.0;
    if (mnf) {
        const float m = (float)(s_act / (double)a.O);       // outer(b, act).mean(-1) = b * mean(act)   :220-221
        for (int i = t; i < a.I; i += 256) {
            const float mr = Gb1[i] * m, lv = Gb2[i] * m;
            const float d = zb - mr;
            s_rb += (double)(-0.5f * 1.1447298858494002f - 0.5f * lv - 0.5f * ((d * d) * __expf(-lv)));  // :223-224
        }
    }
    s_rb = group_sum(s_rb, scr, g, lane, w);
    if (mnf) kl += (-(double)ldq + (double)lq0) - ((double)ldr + s_rb);                             // :215,:225,:235
    if (t == 0) {
        s_kl[g] = on ? (float)kl : 0.f;
        if (on && a.kl_layer) *a.kl_layer = (float)kl;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (A.total) {
            float s = 0.f;
            for (int i = 0; i < A.n; ++i) s += s_kl[i];               // fixed order: l1 + l2 + l3
            *A.total = s;
        }
        if (A.rng && A.advance) A.rng[1] += A.advance;
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

int launch_flow_planar(const FlowArgs* a, int n, hipStream_t s, int members, unsigned long long m_adv, long long z_ms,
                       const FormatJob* fmt, bool* fmt_done) {
    FlowBatch bt;
    bt.m_adv = m_adv; bt.z_ms = z_ms;
    bt.fmt = FormatJob{}; bt.fmt_blocks = 0;
    if (fmt_done) *fmt_done = false;
    int maxI = 0; bool small_t = true, any_kl = false;
    for (int i = 0; i < n; ++i) {
        bt.l[i] = a[i];
        maxI = a[i].I > maxI ? a[i].I : maxI;
        any_kl = any_kl || a[i].want_kl;
    }
    const dim3 grid(any_kl ? 2 : 1, n, members > 1 ? members : 1), block(256);
    (void)small_t;
    size_t need = 0;
    for (int i = 0; i < n; ++i) {
        const size_t v = (size_t)(4 + 2 * (a[i].zf.T + (a[i].want_kl ? a[i].rf.T : 0))) * pad64(a[i].I) * sizeof(float);
        need = v > need ? v : need;
    }
    if (need <= (size_t)kFlowLdsBudget) {
        if (need > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mnf_flow_planar_lds_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)need);
            if (e != hipSuccess) return (int)e;
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(mnf_flow_planar_fast_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)need);
            if (e != hipSuccess) return (int)e;
        }
        bool fast = true;
        for (int i = 0; i < n; ++i) fast = fast && (a[i].zf.T + (a[i].want_kl ? a[i].rf.T : 0) <= kFastT);
        // the register form: float4 column groups, one per thread
        static const bool vec_allowed = [] { const char* e = getenv("LBBNN_K3_VEC"); return !(e && e[0] == '0'); }();   // A/B knob
        bool vec = fast && vec_allowed;
        auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
        for (int i = 0; i < n && vec; ++i) {
            const FlowArgs& f = a[i];
            vec = (f.I % 4 == 0) && f.I <= 4 * kVecThreads && al16(f.q0_mean) && al16(f.q0_log_var) && al16(f.z_fwd) &&
                  (!f.eps_fwd || al16(f.eps_fwd)) && (!f.want_kl || (al16(f.z_kl) && (!f.eps_kl || al16(f.eps_kl)))) &&
                  (z_ms % 4 == 0);
            for (int t = 0; t < f.zf.T && vec; ++t) vec = al16(f.zf.u[t]) && al16(f.zf.w[t]);
            for (int t = 0; t < (f.want_kl ? f.rf.T : 0) && vec; ++t) vec = al16(f.rf.u[t]) && al16(f.rf.w[t]);
        }
        if (vec && fmt && fmt->x && any_kl) {
            // (any_kl: gridDim.x == 2, so the format workgroups are blockIdx.x >= 2)
            const size_t items = (size_t)fmt->B * (fmt->ldp >> 3);
            bt.fmt = *fmt;
            bt.fmt_blocks = (int)((items + kVecThreads - 1) / kVecThreads < 1024 ? (items + kVecThreads - 1) / kVecThreads : 1024);
            dim3 g2(2 + bt.fmt_blocks, n, grid.z);
            hipLaunchKernelGGL(mnf_flow_planar_vec_kernel, g2, dim3(kVecThreads), 0, s, bt);
            if (fmt_done) *fmt_done = true;
        } else if (vec) hipLaunchKernelGGL(mnf_flow_planar_vec_kernel, grid, dim3(kVecThreads), 0, s, bt);
        else if (fast) hipLaunchKernelGGL(mnf_flow_planar_fast_kernel, grid, dim3(kFastThreads), need, s, bt);
        else if (members > 1) return LBBNN_E_SHAPE;                     // the member dimension exists in the fast form only
        else      hipLaunchKernelGGL(mnf_flow_planar_lds_kernel, grid, block, need, s, bt);
    } else {
        if (members > 1) return LBBNN_E_SHAPE;
        hipLaunchKernelGGL(mnf_flow_planar_kernel, grid, block, (size_t)maxI * sizeof(float), s, bt);
    }
    return (int)hipGetLastError();
}

int launch_kl_finalize_all(const FinalizeArgs* a, const int* active, int n, uint64_t* rng, uint64_t advance, float* kl_total,
                           hipStream_t s) {
    FinalizeAllArgs fa;
    for (int i = 0; i < LBBNN_MAX_LAYERS; ++i) { fa.l[i] = i < n ? a[i] : FinalizeArgs{}; fa.active[i] = i < n ? active[i] : 0; }
    fa.n = n; fa.rng = rng; fa.advance = advance; fa.total = kl_total;
    size_t need = 0;
    for (int i = 0; i < n; ++i)
        if (active[i]) need += (size_t)(6 * pad64(a[i].O) + 2 * pad64(a[i].scal ? a[i].I : 1)) * sizeof(float);
    const int staged = need <= (size_t)kFlowLdsBudget ? 1 : 0;
    static size_t raised = 0;                        // dynamic-LDS limit of the function: raise once per size
    if (staged && need > 64 * 1024 && need > raised) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kl_finalize_all_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)need);
        if (e != hipSuccess) return (int)e;
        raised = need;
    }
    hipLaunchKernelGGL(kl_finalize_all_kernel, dim3(1), dim3(n * 256), staged ? need : 0, s, fa, staged);
    return (int)hipGetLastError();
}

int launch_kl_finalize(const FinalizeArgs* a, int n, hipStream_t s) {
    FinalizeBatch bt;
    size_t need = 0;
    for (int i = 0; i < n; ++i) {
        bt.l[i] = a[i];
        const size_t v = (size_t)(6 * pad64(a[i].O) + 2 * pad64(a[i].scal ? a[i].I : 1)) * sizeof(float);
        need = v > need ? v : need;
    }
    if (need <= (size_t)kFlowLdsBudget) {
        if (need > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kl_finalize_lds_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)need);
            if (e != hipSuccess) return (int)e;
        }
        hipLaunchKernelGGL(kl_finalize_lds_kernel, dim3(n), dim3(256), need, s, bt);
    } else {
        hipLaunchKernelGGL(kl_finalize_kernel, dim3(n), dim3(256), 0, s, bt);
    }
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
