// K1b -- analytic backward of the fused weight pass (see include/lbbnn.h).  HBM-bound: 20 B read + 12 B
// written per weight.  A 256-thread workgroup owns 64 column groups (4 columns each, or 1 on the unaligned path) x
// RB = 16 rows: every global access is a coalesced row segment, the three column sums (dz_fwd, dz_kl, dr0_c)
// accumulate in registers, and the per-row-block partials are reduced in a fixed order by a second launch
// (deterministic, no float atomics).
#include <cmath>
#include "lbbnn_device.h"
#include "lbbnn_internal.h"

namespace {

using namespace lbbnn;
constexpr int RB = 16;

struct Consts { float mp, inv_sp2, log_sp, log_ap, log_1map, gk; };

// gradients of one weight; returns dmu, drho, dlam and the three column-sum contributions
__device__ __forceinline__ void elem_bwd(float mu, float rho, float lam, float gWm, float gWv, float zf, float zk, float rc,
                                         float dam, float dav, const Consts& c, bool has_kl, bool has_act,
                                         float& dmu, float& drho, float& dlam, float& czf, float& czk, float& crc) {
    const float ex = __expf(-lam);
    const float alpha = __frcp_rn(1.f + ex);
    const float er = __expf(rho);
    const float sigma = er < 0.04f ? er * (1.f + er * (-0.5f + er * (0.33333334f + er * (-0.25f + er * 0.2f)))) : log1pf(er);
    const float dsig = er * __frcp_rn(1.f + er);              // d softplus / d rho = sigmoid(rho)
    const float a2 = alpha * alpha, s2 = sigma * sigma;
    float Gmu = gWm * alpha * zf;
    float Gsig = gWv * 2.f * sigma * a2;
    float Gal = gWm * mu * zf + gWv * 2.f * s2 * alpha;
    czf = gWm * mu * alpha;
    czk = 0.f; crc = 0.f;
    if (has_kl) {
        const float d = mu * zk - c.mp;
        const float one_m = 1.f - alpha;
        const float T = (c.log_sp - __logf(sigma)) - 0.5f + (__logf(alpha) - c.log_ap) + (s2 + d * d) * 0.5f * c.inv_sp2;
        Gmu += c.gk * alpha * d * zk * c.inv_sp2;
        Gsig += c.gk * alpha * (sigma * c.inv_sp2 - __frcp_rn(sigma));
        Gal += c.gk * (T - (__logf(one_m) - c.log_1map));
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

// Workgroup = 64 column groups (of W columns) x 4 row lanes, RB = 16 rows: thread (cg, rl) walks rows rl, rl+4, ...
// of its column group (a wave reads 64 x 16 B = 1 KiB of one row), keeps the three column sums in registers, and
// the 4 row lanes are added through LDS in a fixed order.  Grid = (column blocks, row blocks): 5 x 75 workgroups
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
    if (live)
        for (int r = r0 + rl; r < r1; r += 4) {
            const size_t off = (size_t)r * a.I + i0;
            const float dam = has_act ? a.da_mu[r] : 0.f, dav = has_act ? a.da_var[r] : 0.f;
            float mu[W], rho[W], lam[W], gm[W], gv[W];
            if (W == 4) {
                const float4 t0 = *reinterpret_cast<const float4*>(a.mu + off), t1 = *reinterpret_cast<const float4*>(a.rho + off);
                const float4 t2 = *reinterpret_cast<const float4*>(a.lambdal + off);
                float4 t3 = *reinterpret_cast<const float4*>(a.dWm + off);
                float4 t4 = a.dWv ? *reinterpret_cast<const float4*>(a.dWv + off) : make_float4(0.f, 0.f, 0.f, 0.f);
                for (int sp = 1; sp < a.nsplit; ++sp) {               // split-K slabs, fixed order
                    const float4 u3 = *reinterpret_cast<const float4*>(a.dWm + (size_t)sp * a.split_stride + off);
                    t3.x += u3.x; t3.y += u3.y; t3.z += u3.z; t3.w += u3.w;
                    if (a.dWv) {
                        const float4 u4 = *reinterpret_cast<const float4*>(a.dWv + (size_t)sp * a.split_stride + off);
                        t4.x += u4.x; t4.y += u4.y; t4.z += u4.z; t4.w += u4.w;
                    }
                }
                mu[0] = t0.x; mu[1] = t0.y; mu[2] = t0.z; mu[3] = t0.w;  rho[0] = t1.x; rho[1] = t1.y; rho[2] = t1.z; rho[3] = t1.w;
                lam[0] = t2.x; lam[1] = t2.y; lam[2] = t2.z; lam[3] = t2.w;  gm[0] = t3.x; gm[1] = t3.y; gm[2] = t3.z; gm[3] = t3.w;
                gv[0] = t4.x; gv[1] = t4.y; gv[2] = t4.z; gv[3] = t4.w;
            } else {
                mu[0] = a.mu[off]; rho[0] = a.rho[off]; lam[0] = a.lambdal[off]; gm[0] = a.dWm[off]; gv[0] = a.dWv ? a.dWv[off] : 0.f;
                for (int sp = 1; sp < a.nsplit; ++sp) {
                    gm[0] += a.dWm[(size_t)sp * a.split_stride + off];
                    if (a.dWv) gv[0] += a.dWv[(size_t)sp * a.split_stride + off];
                }
            }
            float dm[W], dr[W], dl[W];
#pragma unroll
            for (int k = 0; k < W; ++k) {
                float czf, czk, crc;
                elem_bwd(mu[k], rho[k], lam[k], gm[k], gv[k], zf[k], zk[k], rc[k], dam, dav, c, has_kl, has_act,
                         dm[k], dr[k], dl[k], czf, czk, crc);
                szf[k] += czf; szk[k] += czk; src[k] += crc;
            }
            if (W == 4) {
                *reinterpret_cast<float4*>(a.dmu + off) = make_float4(dm[0], dm[1], dm[2], dm[3]);
                *reinterpret_cast<float4*>(a.drho + off) = make_float4(dr[0], dr[1], dr[2], dr[3]);
                *reinterpret_cast<float4*>(a.dlambdal + off) = make_float4(dl[0], dl[1], dl[2], dl[3]);
            } else {
                a.dmu[off] = dm[0]; a.drho[off] = dr[0]; a.dlambdal[off] = dl[0];
            }
        }
    // column sums of this workgroup's 16 rows: row lanes 0..3 added in order
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

// column sums: out[q][i] = sum over row blocks of work[b][q][i].  A workgroup owns 64 columns; its 16 waves
// each sum every 16th row block (coalesced 256-B rows), then one wave adds the 16 partials in a fixed order.
__global__ __launch_bounds__(1024) void wpb_reduce_kernel(const float* __restrict__ work, int nblk, int ldw, int I,
                                                          float* dz_fwd, float* dz_kl, float* dr0_c) {
    __shared__ float part[3][16][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + lane;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    if (i < I)
        for (int b = w; b < nblk; b += 16) {
            const float* p = work + (size_t)b * 3 * ldw;
            s0 += p[i]; s1 += p[ldw + i]; s2 += p[2 * ldw + i];
        }
    part[0][w][lane] = s0; part[1][w][lane] = s1; part[2][w][lane] = s2;
    __syncthreads();
    if (w < 3 && i < I) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += part[w][k][lane];
        float* out = w == 0 ? dz_fwd : (w == 1 ? dz_kl : dr0_c);
        if (out) out[i] = s;
    }
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
