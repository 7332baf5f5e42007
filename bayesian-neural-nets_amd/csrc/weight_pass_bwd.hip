// K1b -- analytic backward of the fused weight pass (see include/lbbnn.h).  HBM-bound: 20 B read + 12 B
// written per weight.  A 256-thread workgroup owns 64 column groups (4 columns each, or 1 on the unaligned path) x
// RB = 8 rows: every global access is a coalesced row segment, the three column sums (dz_fwd, dz_kl, dr0_c)
// accumulate in registers, and the per-row-block partials are reduced in a fixed order by a second launch
// (deterministic, no float atomics).
#include <cmath>
#include "lbbnn_device.h"
#include "lbbnn_internal.h"
#include "reduce_partials.h"

namespace {

using namespace lbbnn;
constexpr int RB = 8;

struct Consts { float mp, inv_sp2, log_sp, log_ap, log_1map, gk; };

// gradients of one weight; returns dmu, drho, dlam and the three column-sum contributions
__device__ __forceinline__ void elem_bwd(float mu, float rho, float lam, float gWm, float gWv, float zf, float zk, float rc,
                                         float dam, float dav, const Consts& c, bool has_kl, bool has_act,
                                         float& dmu, float& drho, float& dlam, float& czf, float& czk, float& crc) {
    // the forward's own forms (lbbnn_device.h k1_*: raw v_exp / v_rcp / v_log, no denormal scaling, no cancellation)
    const float ex = k1_exp_raw(-lam);
    const float ope = 1.f + ex;
    const float alpha = __builtin_amdgcn_rcpf(ope);
    const float er = k1_exp_acc(rho);
    const float sigma = k1_sigma_of(er);
    const float dsig = er * __builtin_amdgcn_rcpf(1.f + er);  // d softplus / d rho = sigmoid(rho)
    const float a2 = alpha * alpha, s2 = sigma * sigma;
    float Gmu = gWm * alpha * zf;
    float Gsig = gWv * 2.f * sigma * a2;
    float Gal = gWm * mu * zf + gWv * 2.f * s2 * alpha;
    czf = gWm * mu * alpha;
    czk = 0.f; crc = 0.f;
    if (has_kl) {
        const float d = mu * zk - c.mp;
        const float la = -0.6931471805599453f * __builtin_amdgcn_logf(ope);       // log(alpha) = -log(1 + exp(-lambda))
        const float T = (c.log_sp - k1_log_sigma_of(rho, er, sigma)) - 0.5f + (la - c.log_ap) + (s2 + d * d) * 0.5f * c.inv_sp2;
        Gmu += c.gk * alpha * d * zk * c.inv_sp2;
        Gsig += c.gk * alpha * (sigma * c.inv_sp2 - __builtin_amdgcn_rcpf(sigma));
        Gal += c.gk * (T - ((la - lam) - c.log_1map));                             // log(1 - alpha) = log(alpha) - lambda
        czk = c.gk * alpha * d * mu * c.inv_sp2;
    }
    if (has_act) {
        const float ma = mu * alpha;
        Gmu += dam * rc * zk * alpha;
        Gal += dam * rc * zk * mu + dav * rc * rc * 2.f * s2 * alpha;
        Gsig += dav * rc * rc * 2.f * sigma * a2;
        czk += dam * rc * ma;
        crc = dam * zk * ma + dav * 2.f * rc * s2 * a2;
    }
    dmu = Gmu;
    drho = Gsig * dsig;
    dlam = Gal * alpha * (1.f - alpha);
}

// Workgroup = 64 column groups (of W columns) x 4 row lanes, RB = 8 rows: thread (cg, rl) owns rows rl and rl + 4
// of its column group (a wave reads 64 x 16 B = 1 KiB of one row), keeps the three column sums in registers, and
// the 4 row lanes are added through LDS in a fixed order.  Grid = (column blocks, row blocks): 5 x 150 workgroups
// for 1200 x 1200 -- the first version (one workgroup per 8 full rows) had 150 workgroups for 256 CUs and an
// 85 %-idle second loop trip.
template <int W>
__global__ __launch_bounds__(256) void weight_pass_bwd_kernel(const lbbnn_wpb_args_t a, int nblk, int ldw) {
    __shared__ float red[3][4][64 * W];
    const int rb = blockIdx.y, tid = threadIdx.x, cg = tid & 63, rl = tid >> 6;
    const int r0 = rb * RB, r1 = min(r0 + RB, a.O);
    Consts c;
    c.mp = a.priors.mu_prior;
    c.inv_sp2 = 1.f / (a.priors.sigma_prior * a.priors.sigma_prior);
    c.log_sp = logf(a.priors.sigma_prior); c.log_ap = logf(a.priors.alpha_prior); c.log_1map = logf(1.f - a.priors.alpha_prior);
    const bool has_kl = a.g_kl != nullptr, has_act = a.da_mu != nullptr;
    c.gk = has_kl ? a.g_kl[0] : 0.f;
    const int i0 = (blockIdx.x * 64 + cg) * W;
    const bool live = i0 < a.I;
    float zf[W], zk[W], rc[W], szf[W], szk[W], src[W];
#pragma unroll
    for (int k = 0; k < W; ++k) {
        const int i = i0 + k;
        const bool in = i < a.I;
        zf[k] = (a.z_fwd && in) ? a.z_fwd[i] : 1.f;
        zk[k] = (a.z_kl && in) ? a.z_kl[i] : 1.f;
        rc[k] = (a.r0_c && in) ? a.r0_c[i] : 0.f;
        szf[k] = szk[k] = src[k] = 0.f;
    }
    if (live && W == 4) {
        // NR = RB / 4 rows per thread, every load of all of them in flight together: the parameters, slab 0 of both weight
        // gradients, then the remaining split-K slabs four at a time (fixed order of additions).  The first version walked
        // rows and slabs one after the other -- a chain of dependent memory latencies per workgroup (the 10-row head with
        // its 16 slabs took as long as a 1200-row layer: 22 us).
        constexpr int NR = RB / 4;
        float4 t[NR][5];
        bool ok[NR];
        size_t offs[NR];
#pragma unroll
        for (int q = 0; q < NR; ++q) {
            const int r = r0 + rl + 4 * q;
            ok[q] = r < r1;
            offs[q] = (size_t)r * a.I + i0;
            const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
            t[q][0] = t[q][1] = t[q][2] = t[q][3] = t[q][4] = z4;
            if (ok[q]) {
                t[q][0] = *reinterpret_cast<const float4*>(a.mu + offs[q]);
                t[q][1] = *reinterpret_cast<const float4*>(a.rho + offs[q]);
                t[q][2] = *reinterpret_cast<const float4*>(a.lambdal + offs[q]);
                t[q][3] = *reinterpret_cast<const float4*>(a.dWm + offs[q]);
                if (a.dWv) t[q][4] = *reinterpret_cast<const float4*>(a.dWv + offs[q]);
            }
        }
        for (int sp = 1; sp < a.nsplit; sp += 4) {
            float4 u3[NR][4], u4[NR][4];
#pragma unroll
            for (int q = 0; q < NR; ++q)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    u3[q][k] = make_float4(0.f, 0.f, 0.f, 0.f); u4[q][k] = u3[q][k];
                    if (ok[q] && sp + k < a.nsplit) {
                        u3[q][k] = *reinterpret_cast<const float4*>(a.dWm + (size_t)(sp + k) * a.split_stride + offs[q]);
                        if (a.dWv) u4[q][k] = *reinterpret_cast<const float4*>(a.dWv + (size_t)(sp + k) * a.split_stride + offs[q]);
                    }
                }
#pragma unroll
            for (int q = 0; q < NR; ++q)
#pragma unroll
                for (int k = 0; k < 4; ++k) {                          // slabs in order (an absent slab adds +0: exact)
                    t[q][3].x += u3[q][k].x; t[q][3].y += u3[q][k].y; t[q][3].z += u3[q][k].z; t[q][3].w += u3[q][k].w;
                    t[q][4].x += u4[q][k].x; t[q][4].y += u4[q][k].y; t[q][4].z += u4[q][k].z; t[q][4].w += u4[q][k].w;
                }
        }
#pragma unroll
        for (int q = 0; q < NR; ++q) {
            if (!ok[q]) continue;
            const int r = r0 + rl + 4 * q;
            const float dam = has_act ? a.da_mu[r] : 0.f, dav = has_act ? a.da_var[r] : 0.f;
            const float mu[4] = {t[q][0].x, t[q][0].y, t[q][0].z, t[q][0].w}, rho[4] = {t[q][1].x, t[q][1].y, t[q][1].z, t[q][1].w};
            const float lam[4] = {t[q][2].x, t[q][2].y, t[q][2].z, t[q][2].w}, gm[4] = {t[q][3].x, t[q][3].y, t[q][3].z, t[q][3].w};
            const float gv[4] = {t[q][4].x, t[q][4].y, t[q][4].z, t[q][4].w};
            float dm[4], dr[4], dl[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float czf, czk, crc;
                elem_bwd(mu[k], rho[k], lam[k], gm[k], gv[k], zf[k], zk[k], rc[k], dam, dav, c, has_kl, has_act,
                         dm[k], dr[k], dl[k], czf, czk, crc);
                szf[k] += czf; szk[k] += czk; src[k] += crc;
            }
            *reinterpret_cast<float4*>(a.dmu + offs[q]) = make_float4(dm[0], dm[1], dm[2], dm[3]);
            *reinterpret_cast<float4*>(a.drho + offs[q]) = make_float4(dr[0], dr[1], dr[2], dr[3]);
            *reinterpret_cast<float4*>(a.dlambdal + offs[q]) = make_float4(dl[0], dl[1], dl[2], dl[3]);
        }
    } else if (live) {
        for (int r = r0 + rl; r < r1; r += 4) {
            const size_t off = (size_t)r * a.I + i0;
            const float dam = has_act ? a.da_mu[r] : 0.f, dav = has_act ? a.da_var[r] : 0.f;
            float mu = a.mu[off], rho = a.rho[off], lam = a.lambdal[off], gm = a.dWm[off], gv = a.dWv ? a.dWv[off] : 0.f;
            for (int sp = 1; sp < a.nsplit; ++sp) {
                gm += a.dWm[(size_t)sp * a.split_stride + off];
                if (a.dWv) gv += a.dWv[(size_t)sp * a.split_stride + off];
            }
            float czf, czk, crc, dm, dr, dl;
            elem_bwd(mu, rho, lam, gm, gv, zf[0], zk[0], rc[0], dam, dav, c, has_kl, has_act, dm, dr, dl, czf, czk, crc);
            szf[0] += czf; szk[0] += czk; src[0] += crc;
            a.dmu[off] = dm; a.drho[off] = dr; a.dlambdal[off] = dl;
        }
    }
    // column sums of this workgroup's RB rows: row lanes 0..3 added in order
#pragma unroll
    for (int k = 0; k < W; ++k) { red[0][rl][cg * W + k] = szf[k]; red[1][rl][cg * W + k] = szk[k]; red[2][rl][cg * W + k] = src[k]; }
    __syncthreads();
    if (rl < 3 && live) {
        float* const dst = a.work + ((size_t)rb * 3 + rl) * ldw;
#pragma unroll
        for (int k = 0; k < W; ++k) {
            const int i = i0 + k;
            if (i < a.I) dst[i] = ((red[rl][0][cg * W + k] + red[rl][1][cg * W + k]) + red[rl][2][cg * W + k]) + red[rl][3][cg * W + k];
        }
    }
}

// column sums: out[q][i] = sum over row blocks of work[b][q][i] (reduce_partials.h: the body shared with
// lbbnn_reduce_partials_batch)
__global__ __launch_bounds__(1024) void wpb_reduce_kernel(const float* __restrict__ work, int nblk, int ldw, int I,
                                                          float* dz_fwd, float* dz_kl, float* dr0_c) {
    __shared__ float part[3][16][64];
    float* const out[3] = {dz_fwd, dz_kl, dr0_c};
    reduce_partials_body(work, 3LL * ldw, ldw, nblk, I, 3, out, blockIdx.x, part);
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace

extern "C" int64_t lbbnn_weight_pass_backward_workspace(int O, int I) {
    if (O <= 0 || I <= 0) return 0;
    return (int64_t)((O + RB - 1) / RB) * 3 * (((int64_t)I + 3) & ~3LL);
}

extern "C" int lbbnn_weight_pass_backward(const lbbnn_wpb_args_t* p, void* stream) {
    if (!p) return LBBNN_E_NULL;
    const lbbnn_wpb_args_t& a = *p;
    if (!a.mu || !a.rho || !a.lambdal || !a.dWm || !a.dmu || !a.drho || !a.dlambdal || !a.work) return LBBNN_E_NULL;
    if (a.O <= 0 || a.I <= 0 || a.nsplit < 0) return LBBNN_E_SHAPE;
    if (a.nsplit > 1 && (a.split_stride < (int64_t)a.O * a.I || (a.split_stride & 3))) return LBBNN_E_ALIGN;
    if ((a.da_mu == nullptr) != (a.da_var == nullptr)) return LBBNN_E_NULL;
    if (a.da_mu && (!a.z_kl || !a.r0_c)) return LBBNN_E_NULL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int nblk = (a.O + RB - 1) / RB;
    const int ldw = (a.I + 3) & ~3;
    const bool vec = (a.I % 4 == 0) && al16(a.mu) && al16(a.rho) && al16(a.lambdal) && al16(a.dWm) &&
                     (!a.dWv || al16(a.dWv)) && al16(a.dmu) && al16(a.drho) && al16(a.dlambdal);
    if (vec) hipLaunchKernelGGL(weight_pass_bwd_kernel<4>, dim3((a.I / 4 + 63) / 64, nblk), dim3(256), 0, s, a, nblk, ldw);
    else     hipLaunchKernelGGL(weight_pass_bwd_kernel<1>, dim3((a.I + 63) / 64, nblk), dim3(256), 0, s, a, nblk, ldw);
    if (a.dz_fwd || a.dz_kl || a.dr0_c)
        hipLaunchKernelGGL(wpb_reduce_kernel, dim3((a.I + 63) / 64), dim3(1024), 0, s, a.work, nblk, ldw, a.I,
                           a.dz_fwd, a.dz_kl, a.dr0_c);
    return (int)hipGetLastError();
}
