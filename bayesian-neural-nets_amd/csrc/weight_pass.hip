// K1 -- fused pass over the (O,I) variational parameters of one Bayesian layer (gfx950).
//
// One HBM read of mu/rho/lambdal (12 B per weight) produces everything the rest of the layer
// needs from them: the two GEMM operands (e_w with the MNF multiplier z folded in, var_w), the
// per-row KL sums, and the auxiliary-posterior row reductions act_mu / act_var.  The reference
// spends ~25 separate full-matrix aten passes on the same values.
//
// Roofline: HBM.  Algorithmic bytes per weight: 12 read + 8 written (fp32 operands).
//
// Two kernels:
//   weight_rows_kernel (rows whose length is a multiple of 4, 16-B aligned, ld <= 2048: every layer of the BASELINE
//     configurations)  -- round 2.  ONE WAVE PER ROW, four rows per 256-thread workgroup: a lane owns the float4 column
//     groups lane, lane + 64, ... so a wave reads 1 KiB contiguous per instruction, has all of its row's loads in flight
//     at once (15 x 16 B per lane at I = 1200) and reduces its row sums with DPP moves alone -- no barrier, no LDS
//     round trip per row.  603 workgroups cover the headline net, all resident at once (<= 3 per CU).
//     The per-column vectors z_fwd, z_kl, r0_c are staged ONCE per workgroup in LDS (round 1 re-read them from L2 for
//     every row).  For planar MNF layers the workgroup computes them ITSELF (in_flow): z0 = q0_mean + q0_std eps for both
//     draws, the <= 19 dot products of the planar chains reduced once, the tanh / log-det chain as scalars -- the K3
//     kernel of round 1 (one latency-bound workgroup pair per layer, 13 us + a launch boundary AHEAD of K1 on the
//     critical path) is gone from the fused forward; the first workgroup of a layer also writes what K3 wrote
//     (z_fwd, z_kl, scal[0..4]) for K5 and for the backward pass.
//     Split (bf16 hi | lo) operands leave as full 16-B units: lane pairs swap halves by DPP, so a wave-store covers
//     1 KiB contiguous (round 1: four 8-B stores per lane into every other 16-B unit; WRITE_SIZE 2.5x the bytes).
//   weight_pass_kernel: the generic form (any I, any alignment), one 256-thread workgroup per row.
// Row sums have a fixed order in both => kl_rows / act_* are bitwise reproducible.
#include <cmath>
#include <cstdlib>
#include "lbbnn_device.h"
#include "lbbnn_internal.h"

namespace {

using namespace lbbnn;

struct WeightPassBatch { WeightPassArgs l[LBBNN_MAX_LAYERS]; int row_end[LBBNN_MAX_LAYERS]; int n;
                         uint64_t* rng; uint64_t* rng_snap; uint64_t advance; };   // see launch_weight_pass

struct Elem { float ew, vw, kl, amu, avar; };

// round-to-nearest-even fp32 -> bf16 bits (finite inputs)
__device__ __forceinline__ uint32_t bf16_rne(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ float bf16_to_f32(uint32_t b) { return __uint_as_float(b << 16); }
// split w = hi + lo (both bf16): hi = rne(w), lo = rne(w - hi); packs 4 elements into two uint2.  The roundings are
// the hardware's v_cvt_pk_bf16_f32 (RNE, two values per instruction): ~12 VALU per float4 instead of ~50 for the
// integer form above, bit-identical for finite inputs.
typedef __bf16 k1_bf16x2 __attribute__((ext_vector_type(2)));
typedef float k1_floatx2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t cvt_pk(float a, float b) {
    const k1_floatx2 v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, k1_bf16x2));
}
__device__ __forceinline__ void split4(const float4 w, uint2& hi, uint2& lo) {
    const uint32_t h0 = cvt_pk(w.x, w.y), h1 = cvt_pk(w.z, w.w);
    hi = make_uint2(h0, h1);
    lo = make_uint2(cvt_pk(w.x - __uint_as_float(h0 << 16), w.y - __uint_as_float(h0 & 0xFFFF0000u)),
                    cvt_pk(w.z - __uint_as_float(h1 << 16), w.w - __uint_as_float(h1 & 0xFFFF0000u)));
}

// fp16 forms for the row-scaled format (LBBNN_F_F16S): w * scale = hi + lo, both IEEE fp16 (v_cvt_pk_f16_f32, RNE)
typedef _Float16 k1_f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t cvt_pk_h(float a, float b) {
    const k1_floatx2 v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, k1_f16x2));
}
__device__ __forceinline__ void split4_h(const float4 w, uint2& hi, uint2& lo) {
    const uint32_t h0 = cvt_pk_h(w.x, w.y), h1 = cvt_pk_h(w.z, w.w);
    const k1_f16x2 f0 = __builtin_bit_cast(k1_f16x2, h0), f1 = __builtin_bit_cast(k1_f16x2, h1);
    hi = make_uint2(h0, h1);
    lo = make_uint2(cvt_pk_h(w.x - (float)f0[0], w.y - (float)f0[1]), cvt_pk_h(w.z - (float)f1[0], w.w - (float)f1[1]));
}
// ... of w * scale (scale a power of two: exact), the products and residuals on packed pairs
__device__ __forceinline__ void split4_hs(const float4 w, float scale, uint2& hi, uint2& lo) {
    const k1_floatx2 a = k1_floatx2{w.x, w.y} * scale, b = k1_floatx2{w.z, w.w} * scale;
    const k1_f16x2 f0 = __builtin_convertvector(a, k1_f16x2), f1 = __builtin_convertvector(b, k1_f16x2);
    const k1_floatx2 ra = a - __builtin_convertvector(f0, k1_floatx2), rb = b - __builtin_convertvector(f1, k1_floatx2);
    hi = make_uint2(__builtin_bit_cast(uint32_t, f0), __builtin_bit_cast(uint32_t, f1));
    lo = make_uint2(__builtin_bit_cast(uint32_t, __builtin_convertvector(ra, k1_f16x2)),
                    __builtin_bit_cast(uint32_t, __builtin_convertvector(rb, k1_f16x2)));
}
// max over the 64 lanes (every lane gets it); v >= 0
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_get<0xB1>(v));
    v = fmaxf(v, dpp_get<0x4E>(v));
    v = fmaxf(v, dpp_get<0x141>(v));
    v = fmaxf(v, dpp_get<0x140>(v));
    return fmaxf(fmaxf(lane_get(v, 0), lane_get(v, 16)), fmaxf(lane_get(v, 32), lane_get(v, 48)));
}
// Exact power-of-two scale that puts a row maximum m into [2^13, 2^14) (fp16 overflows at 65504 = 2^16 - 32: headroom for
// the RNE of hi), and its inverse -- the factor the GEMM epilogue applies to the accumulators.  m == 0 (or denormal): 1.
__device__ __forceinline__ void pow2_scale(float m, float& scale, float& inv) {
    int eb = (int)((__float_as_uint(m) >> 23) & 0xFFu);           // biased exponent: m in [2^(eb-127), 2^(eb-126))
    if (eb == 0) { scale = 1.f; inv = 1.f; return; }
    eb = eb < 20 ? 20 : (eb > 230 ? 230 : eb);
    scale = __uint_as_float((uint32_t)(267 - eb) << 23);          // 2^(13 - (eb - 127))
    inv = __uint_as_float((uint32_t)(eb - 13) << 23);
}

// All per-weight arithmetic (LBBNN-GP-MF-LRT.py:167-171,189-192; LBBNN-GP-MF-MNF.py:195-196,211-212,230-233).
//
// The pass must stay HBM-bound (20 B per weight), so the ~7 transcendentals per weight use the
// hardware forms (v_exp_f32 / v_log_f32 / v_rcp_f32, ~1 ulp) instead of the ~30-instruction libm
// sequences:  alpha = rcp(1 + exp(-lambda));  sigma = log1p(exp(rho)) by a 5-term series when
// exp(rho) < 0.04 (the reference's init has exp(rho) in [0.0067, 0.018]; truncation error < 1e-9 relative)
// and libm log1pf otherwise;  log(a/b) = log a - log b with the prior logs precomputed on the host.
// Measured against the fp64 oracle: operands and row sums stay within 2e-6 relative (tests).
// Ablations (tools/lab, all layers of the headline net): full 21.1 us; operand stores removed 19.1; arithmetic
// removed 14.6 -- the floor is the per-workgroup load -> reduce -> store latency of 2410 short workgroups.  A version
// walking ~4 rows per workgroup with the next row's loads in flight was measured at 29.5 us (fewer workgroups in
// flight per CU cost more than the prefetch saves) and dropped.
__device__ __forceinline__ Elem weight_elem(float mu, float rho, float lam, float zf, float zk, float rc,
                                            bool want_kl, bool want_act, const WeightPassArgs& a) {
    Elem e;
    const float alpha = k1_alpha(lam);
    const float sigma = k1_sigma(rho);
    const float ea = mu * alpha;
    e.ew = ea * zf;
    e.vw = (sigma * sigma) * (alpha * alpha);
    e.kl = 0.f; e.amu = 0.f; e.avar = 0.f;
    if (want_kl) {
        const float d = mu * zk - a.mu_prior;
        const float one_m = 1.f - alpha;
        e.kl = alpha * ((a.log_sp - __logf(sigma)) - 0.5f + (__logf(alpha) - a.log_ap)
                        + (sigma * sigma + d * d) * a.inv_2sp2)
             + one_m * (__logf(one_m) - a.log_1map);
    }
    if (want_act) {
        e.amu = rc * ((zk * mu) * alpha);
        e.avar = (rc * rc) * e.vw;
    }
    return e;
}

// grid.x = total rows of all layers in the batch; a block finds its layer by the row prefix ends.
__global__ __launch_bounds__(256) void weight_pass_kernel(const WeightPassBatch bt) {
    __shared__ float red[3][4];
    // RNG bookkeeping of a fused forward (lbbnn_layers_operands_snap): this launch follows the last kernel that reads the
    // live {seed, offset}; it copies the pair to rng_snap -- what every later kernel of the forward reads -- and advances
    // the live offset, so no launch of its own is needed for that at the end of the forward.
    if (bt.rng_snap && blockIdx.x == 0 && threadIdx.x == 0) {
        const uint64_t sd = bt.rng[0], of = bt.rng[1];
        bt.rng_snap[0] = sd; bt.rng_snap[1] = of;
        bt.rng[1] = of + bt.advance;
    }
    int li = 0;
#pragma unroll
    for (int t = 0; t < LBBNN_MAX_LAYERS - 1; ++t) if (t + 1 < bt.n && (int)blockIdx.x >= bt.row_end[t]) li = t + 1;
    WeightPassArgs a;
    LBBNN_SELECT_LAYER(a, bt.l, li);        // constant-index select: no scratch copy of the argument array
    const int o = (int)blockIdx.x - (li ? bt.row_end[li - 1] : 0);
    const bool VEC = a.vec != 0;
    const int tid = threadIdx.x;
    const size_t rowoff = (size_t)o * a.I;
    const bool want_kl = a.kl_rows != nullptr;
    const bool want_act = a.act_mu != nullptr;
    float kl = 0.f, amu = 0.f, avar = 0.f;

    if (VEC) {
        const int n4 = a.ld >> 2, i4 = a.I >> 2;
        // prefetch the second column group of this thread (rows of 257..512 float4) with the first
        float4 pmu = make_float4(0.f, 0.f, 0.f, 0.f), prho = pmu, plam = pmu;
        const bool pre = (tid + 256) < i4;
        if (pre) {
            pmu = reinterpret_cast<const float4*>(a.mu + rowoff)[tid + 256];
            prho = reinterpret_cast<const float4*>(a.rho + rowoff)[tid + 256];
            plam = reinterpret_cast<const float4*>(a.lambdal + rowoff)[tid + 256];
        }
        for (int j = tid; j < n4; j += 256) {
            float4 ew = make_float4(0.f, 0.f, 0.f, 0.f), vw = ew;
            if (j < i4) {
                const bool usep = pre && j == tid + 256;
                const float4 mu = usep ? pmu : reinterpret_cast<const float4*>(a.mu + rowoff)[j];
                const float4 rho = usep ? prho : reinterpret_cast<const float4*>(a.rho + rowoff)[j];
                const float4 lam = usep ? plam : reinterpret_cast<const float4*>(a.lambdal + rowoff)[j];
                const float4 one = make_float4(1.f, 1.f, 1.f, 1.f);
                const float4 zf = a.z_fwd ? reinterpret_cast<const float4*>(a.z_fwd)[j] : one;
                const float4 zk = a.z_kl ? reinterpret_cast<const float4*>(a.z_kl)[j] : one;
                const float4 rc = a.r0_c ? reinterpret_cast<const float4*>(a.r0_c)[j] : one;
                const Elem e0 = weight_elem(mu.x, rho.x, lam.x, zf.x, zk.x, rc.x, want_kl, want_act, a);
                const Elem e1 = weight_elem(mu.y, rho.y, lam.y, zf.y, zk.y, rc.y, want_kl, want_act, a);
                const Elem e2 = weight_elem(mu.z, rho.z, lam.z, zf.z, zk.z, rc.z, want_kl, want_act, a);
                const Elem e3 = weight_elem(mu.w, rho.w, lam.w, zf.w, zk.w, rc.w, want_kl, want_act, a);
                ew = make_float4(e0.ew, e1.ew, e2.ew, e3.ew);
                vw = make_float4(e0.vw, e1.vw, e2.vw, e3.vw);
                kl += (e0.kl + e1.kl) + (e2.kl + e3.kl);
                amu += (e0.amu + e1.amu) + (e2.amu + e3.amu);
                avar += (e0.avar + e1.avar) + (e2.avar + e3.avar);
            }
            if (!a.split) {
                if (a.e_w) reinterpret_cast<float4*>(a.e_w + (size_t)o * a.ld)[j] = ew;
                if (a.var_w) reinterpret_cast<float4*>(a.var_w + (size_t)o * a.ld)[j] = vw;
            } else {
                // split-precision operands (layout: lbbnn_device.h): this thread's 4 consecutive k are half of one 16-B
                // hi unit, the matching lo values sit one unit further; zero tail kept
                const size_t at = split_hi_index((size_t)o, 4 * j, a.ld);
                uint2 hi, lo;
                split4(ew, hi, lo);
                if (a.e_w) {
                    uint16_t* const e = reinterpret_cast<uint16_t*>(a.e_w);
                    *reinterpret_cast<uint2*>(e + at) = hi;
                    *reinterpret_cast<uint2*>(e + at + kSplitLoOffset) = lo;
                }
                if (a.var_w) {
                    uint16_t* const v = reinterpret_cast<uint16_t*>(a.var_w);
                    split4(vw, hi, lo);
                    *reinterpret_cast<uint2*>(v + at) = hi;
                    *reinterpret_cast<uint2*>(v + at + kSplitLoOffset) = lo;
                }
            }
        }
    } else {
        for (int i = tid; i < a.ld; i += 256) {
            float ew = 0.f, vw = 0.f;
            if (i < a.I) {
                const Elem e = weight_elem(a.mu[rowoff + i], a.rho[rowoff + i], a.lambdal[rowoff + i],
                                           a.z_fwd ? a.z_fwd[i] : 1.f, a.z_kl ? a.z_kl[i] : 1.f,
                                           a.r0_c ? a.r0_c[i] : 1.f, want_kl, want_act, a);
                ew = e.ew; vw = e.vw; kl += e.kl; amu += e.amu; avar += e.avar;
            }
            if (a.e_w) a.e_w[(size_t)o * a.ld + i] = ew;
            if (a.var_w) a.var_w[(size_t)o * a.ld + i] = vw;
        }
    }

    if (want_kl || want_act) {
        kl = wave_sum(kl); amu = wave_sum(amu); avar = wave_sum(avar);
        const int lane = tid & 63, w = tid >> 6;
        if (lane == 0) { red[0][w] = kl; red[1][w] = amu; red[2][w] = avar; }
        __syncthreads();
        if (tid == 0) {
            if (want_kl) a.kl_rows[o] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
            if (want_act) {
                a.act_mu[o] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
                a.act_var[o] = (red[2][0] + red[2][1]) + (red[2][2] + red[2][3]);
            }
        }
    }
    if (tid == 0 && a.bias_var && a.bias_rho) {
        const float sb = softplus_ref(a.bias_rho[o]);
        a.bias_var[o] = sb * sb;                      // bias.sigma**2, LBBNN-GP-MF-LRT.py:173
    }
}


// ------------------------------------------------------------------------------------------------ round-2 row kernel
struct ElemConst { float mu_prior, log_sp, log_ap, log_1map, inv_2sp2; };

// Round 3: the same arithmetic on PAIRS of weights.  The row kernel is a single wave of workgroups (2410 rows on 3072 wave
// slots) whose VALU work is not hidden behind anyone else's loads, so its instruction count is time: ~70 VALU per weight in
// the scalar form above.  Here the non-transcendental part runs on float2 (v_pk_mul / v_pk_fma / v_pk_add_f32: two weights
// per instruction), the transcendentals are the raw hardware forms (v_exp_f32 = 2^x, v_log_f32 = log2: the library forms
// spend 4-5 instructions each on denormal scaling these arguments cannot need), log(1 - alpha) = log(alpha) - lambda costs
// no logarithm (and no cancellation), and log(sigma) = rho + log1p-series while exp(rho) < 0.04 (the softplus series'
// companion: log(softplus(rho)) - rho = -y/2 + 5y^2/24 - y^3/8 + 251y^4/2880, |next term| < 7e-9 there).
typedef float k1_f2 __attribute__((ext_vector_type(2)));
struct Elem2 { k1_f2 ew, vw, kl, amu, avar; };

// (k1_exp_raw / k1_exp_acc / k1_alpha / k1_sigma_of of lbbnn_device.h on pairs: the same IEEE operations in the same order,
// so a packed lane and the scalar form give the same bits -- lbbnn_weight_operands_t and the generic kernel rely on it)
__device__ __forceinline__ k1_f2 exp_raw2(k1_f2 x) {
    const k1_f2 t = x * 1.4426950408889634f;
    return k1_f2{__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
}
__device__ __forceinline__ k1_f2 exp_acc2(k1_f2 x) {
    const k1_f2 l2e = {1.4426950408889634f, 1.4426950408889634f};
    const k1_f2 th = x * l2e;
    k1_f2 tl = __builtin_elementwise_fma(x, l2e, -th);
    tl = __builtin_elementwise_fma(x, k1_f2{1.9259629911e-8f, 1.9259629911e-8f}, tl);      // log2(e) - fl32(log2(e))
    const k1_f2 e = {__builtin_amdgcn_exp2f(th.x), __builtin_amdgcn_exp2f(th.y)};
    return __builtin_elementwise_fma(e, tl * 0.6931471805599453f, e);
}
__device__ __forceinline__ k1_f2 ln_raw2(k1_f2 x) {
    return k1_f2{__builtin_amdgcn_logf(x.x), __builtin_amdgcn_logf(x.y)} * 0.6931471805599453f;
}
__device__ __forceinline__ k1_f2 fma2(k1_f2 a, k1_f2 b, k1_f2 c) { return __builtin_elementwise_fma(a, b, c); }

__device__ __forceinline__ Elem2 weight_elem_p2(k1_f2 mu, k1_f2 rho, k1_f2 lam, k1_f2 zf, k1_f2 zk, k1_f2 rc,
                                                bool want_kl, bool want_act, const ElemConst& c) {
    Elem2 e;
    const k1_f2 en = exp_raw2(-lam);                                    // exp(-lambda): 0 / inf at |lambda| > 87 -> alpha 1 / 0
    const k1_f2 ope = en + 1.f;
    const k1_f2 alpha = {__builtin_amdgcn_rcpf(ope.x), __builtin_amdgcn_rcpf(ope.y)};
    const k1_f2 y = exp_acc2(rho);
    k1_f2 sigma, lsig;
    if (y.x < 0.04f && y.y < 0.04f) {
        const k1_f2 p = fma2(y, fma2(y, fma2(y, fma2(y, k1_f2{0.2f, 0.2f}, k1_f2{-0.25f, -0.25f}), k1_f2{0.33333334f, 0.33333334f}),
                                     k1_f2{-0.5f, -0.5f}), k1_f2{1.f, 1.f});
        sigma = y * p;
        if (want_kl)
            lsig = fma2(y, fma2(y, fma2(y, fma2(y, k1_f2{0.08715278f, 0.08715278f}, k1_f2{-0.125f, -0.125f}),
                                        k1_f2{0.20833333f, 0.20833333f}), k1_f2{-0.5f, -0.5f}), rho);
    } else {
        sigma = k1_f2{k1_sigma_of(y.x), k1_sigma_of(y.y)};
        lsig = k1_f2{__logf(sigma.x), __logf(sigma.y)};
    }
    const k1_f2 s2 = sigma * sigma;
    e.ew = (mu * alpha) * zf;
    e.vw = s2 * (alpha * alpha);
    e.kl = k1_f2{0.f, 0.f}; e.amu = e.kl; e.avar = e.kl;
    if (want_kl) {
        const k1_f2 d = fma2(mu, zk, k1_f2{-c.mu_prior, -c.mu_prior});
        const k1_f2 la = -ln_raw2(ope);                                 // log(alpha) = -log(1 + exp(-lambda))
        const k1_f2 l1m = la - lam;                                     // log(1 - alpha)
        const k1_f2 one_m = 1.f - alpha;
        const k1_f2 t = ((c.log_sp - lsig) - 0.5f) + (la - c.log_ap) + fma2(d, d, s2) * c.inv_2sp2;
        e.kl = alpha * t + one_m * (l1m - c.log_1map);
    }
    if (want_act) {
        e.amu = rc * ((zk * mu) * alpha);
        e.avar = (rc * rc) * e.vw;
    }
    return e;
}

constexpr int kInT = 4;                                  // in-kernel planar flows: Tz + Tr <= 4 (the K3 fast form's limit)
constexpr int kInNV = 3 * kInT + kInT * (kInT - 1) / 2 + kInT + 1;      // A_k | UW | X | A_f | LQ0 = 23 sums (19 used at T = 2 + 2)
constexpr int kRowG = 8;                                 // float4 groups per lane: ld <= 64 * 4 * 8 = 2048
constexpr int kRowB = 5;                                 // ... of which this many are loaded together
#ifndef LBBNN_K1_ROWS_PER_WG
#define LBBNN_K1_ROWS_PER_WG 1
#endif
// Rows (= waves) per workgroup of the row kernel.  Measured on the headline forward, round 3, alternating processes on one box
// (profiles/r03_ab_k1_rows.txt).  With the layer's per-column vectors (z_fwd, z_kl, r0_c: 14.4 KB at I = 1200) staged in LDS
// once per workgroup: 4 rows 0.1251 ms, 2 rows 0.1249, 1 row 0.1259, 8 rows 0.1285 -- the launch is a single wave of
// workgroups (2410 rows on 3072 wave slots) and a SIMD holds 2 or 3 of them whatever the grouping; fewer rows per workgroup
// balance the CUs better (603 four-row workgroups: 3 on 91 CUs, 2 on the rest) but stage the vectors once per row, which
// costs what the balance gains.  ONE row per workgroup with the vectors taken straight into registers, a float4 group at a
// time beside the row's parameters (no LDS, no barrier; kDirectZ below): 0.1220 against 0.1226 for the 4-row form -- kept.
// (A first run of this experiment staged only the first 2 x 64 x rows float4s of the vectors -- the loop was written for 4
// rows -- and "won" 3 us at 1 row with wrong operands: the staging loop now covers any workgroup size.)  In-kernel flows
// (INFLOW) are computed by four waves together.
constexpr int kRowsWGFlow = 4;
constexpr int kRowsWG = LBBNN_K1_ROWS_PER_WG;
template <bool INFLOW> struct RowsCfg { static constexpr int RW = INFLOW ? kRowsWGFlow : kRowsWG, NT = 64 * RW; };

struct WeightRowsBatch {
    WeightPassArgs l[LBBNN_MAX_LAYERS];
    InFlow f[LBBNN_MAX_LAYERS];
    int wg_end[LBBNN_MAX_LAYERS];
    int n;
    uint64_t* rng; uint64_t* rng_snap; uint64_t advance;
    int members;                 // ensemble (lbbnn_ensemble_operands): gridDim.y members; member m reads z_fwd + m*ld and writes
                                 // e_w + m*O*ld; var_w / bias_var are written by member 0 only; no KL outputs
};

__device__ __forceinline__ float4 ld4(const float* p, int j) { return reinterpret_cast<const float4*>(p)[j]; }
__device__ __forceinline__ float dot4(const float4 a, const float4 b) { return (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w); }
__device__ __forceinline__ float4 fma4(const float4 u, float s, const float4 z) {
    return make_float4(z.x + u.x * s, z.y + u.y * s, z.z + u.z * s, z.w + u.w * s);
}
__device__ __forceinline__ uint32_t dpp_xor1(uint32_t v) { return (uint32_t)dpp_mov<0xB1>((int)v); }   // lane ^ 1

// INFLOW: some layer of the launch computes its planar flows inside its workgroups (f.on); the fused forwards since K3v have
// none, and their instantiation carries neither that code nor its register pressure.
template <bool F16S, bool INFLOW>
__global__ __launch_bounds__(RowsCfg<INFLOW>::NT, 3) void weight_rows_kernel(const WeightRowsBatch bt_) {
    constexpr int kRowsWG = RowsCfg<INFLOW>::RW, kRowNT = RowsCfg<INFLOW>::NT;
    const LBBNN_CONST_AS WeightRowsBatch* bt = kernarg_as<WeightRowsBatch>();
    extern __shared__ __attribute__((aligned(16))) float sm[];          // zf[P] | zk[P] | rc[P]
    __shared__ double red[INFLOW ? kInNV : 1][kRowsWG];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (bt->rng_snap && blockIdx.x == 0 && tid == 0) {                   // see weight_pass_kernel
        const uint64_t sd = bt->rng[0], of = bt->rng[1];
        bt->rng_snap[0] = sd; bt->rng_snap[1] = of;
        if (bt->advance) bt->rng[1] = of + bt->advance;                  // (never together with in_flow: the host checks)
    }
    int li = 0;
#pragma unroll
    for (int t = 0; t < LBBNN_MAX_LAYERS - 1; ++t) if (t + 1 < bt->n && (int)blockIdx.x >= bt->wg_end[t]) li = t + 1;
    const LBBNN_CONST_AS WeightPassArgs& a = bt->l[li];
    const LBBNN_CONST_AS InFlow& f = bt->f[li];
    const int wg0 = li ? bt->wg_end[li - 1] : 0;
    const bool first_wg = (int)blockIdx.x == wg0;
    const int I = a.I, P = a.ld, nq = P >> 2, iq = I >> 2;
    const int mem = blockIdx.y;                                          // 0 unless an ensemble launch
    float* zf_s = sm; float* zk_s = sm + P; float* rc_s = sm + 2 * P;
    const bool want_kl = a.kl_rows != nullptr, want_act = a.act_mu != nullptr;
    const float4 one4 = make_float4(1.f, 1.f, 1.f, 1.f), zero4 = make_float4(0.f, 0.f, 0.f, 0.f);

    const int o = kRowsWG * ((int)blockIdx.x - wg0) + wv;
    const bool has_row = o < a.O;
    const int G = (nq + 63) >> 6;                                        // wave-uniform, <= kRowG
    const size_t rowoff = (size_t)o * I;
    // (Measured and dropped, round 2: requesting the row AHEAD of the prologue -- held in registers: 98 spilled VGPRs at the
    // 168-register budget three workgroups per CU need; touched line by line with unused asm loads: 35.7 us against 33.6.)

    // Round 3: with the flows computed by their own launch (f.on == 0: the fused forwards since K3v) the row's first batch of
    // parameter loads is requested HERE, ahead of the staging of the per-column vectors -- 15 x 16 B per lane in flight while
    // the z vectors make their round trip to LDS, instead of after the barrier behind it.  (With the in-kernel flows the same
    // hoist spilled: see above; that path keeps the loads below.)
    float4 mu[kRowB], rho[kRowB], lam[kRowB];
    // one-row workgroups: no sharing to stage for -- every lane takes its own float4s of the per-column vectors straight
    // into registers, with the row's parameters (no LDS round trip, no barrier)
    constexpr bool kDirectZ = !INFLOW && kRowsWG == 1;
    float4 zfr[1], zkr[1], rcr[1];
    if (!INFLOW && has_row) {
#pragma unroll
        for (int g = 0; g < kRowB; ++g) {
            const int j = lane + 64 * g;
            if (g < G && j < iq) { mu[g] = ld4(a.mu + rowoff, j); rho[g] = ld4(a.rho + rowoff, j); lam[g] = ld4(a.lambdal + rowoff, j); }
        }
    }
    // (all five groups of the three vectors up front, beside the 15 parameter registers and the operands kept for the row
    // maximum: 115 spilled VGPRs at the 168 three waves per SIMD leave -- so one group ahead of its use, two register sets)
    auto load_z = [&](int gg, float4& zf, float4& zk, float4& rc) {
        const int j = lane + 64 * gg;
        zf = one4; zk = one4; rc = one4;
        if (gg < G && j < iq) {
            if (a.z_fwd) zf = ld4(a.z_fwd + (size_t)mem * P, j);
            if (a.z_kl) zk = ld4(a.z_kl, j);
            if (a.r0_c) rc = ld4(a.r0_c, j);
        }
    };
    if constexpr (kDirectZ) { if (has_row) load_z(0, zfr[0], zkr[0], rcr[0]); }

    // ---------------------------------------------------------------- per-column vectors of this layer -> LDS
    if (INFLOW && f.on) {
        const int Tz = f.Tz, NT = f.Tz + f.Tr;
        const bool klb = f.want_kl != 0;
        uint64_t seed = 0, offs = 0;
        if (!f.eps_fwd || (klb && !f.eps_kl)) { seed = f.rng[0]; offs = f.rng[1]; }
        double acc[kInNV];
#pragma unroll
        for (int k = 0; k < kInNV; ++k) acc[k] = 0.0;
        // (the sweep parks z0 of both draws in LDS and the update re-reads u: keeping them in registers across the
        // reduction spilled at the 168-VGPR budget that three workgroups per CU need)
#pragma unroll 1
        for (int h = 0; h < 2; ++h) {
            const int j = tid + kRowNT * h;
            if (j < iq) {
                const float4 qm = ld4(f.q0_mean, j), lv = ld4(f.q0_log_var, j);
                float ef[4], ek[4] = {0.f, 0.f, 0.f, 0.f};
                if (f.eps_fwd) { const float4 e = ld4(f.eps_fwd, j); ef[0] = e.x; ef[1] = e.y; ef[2] = e.z; ef[3] = e.w; }
                else philox_normal4(seed, offs, LBBNN_STREAM_EPS_Z * 64u + f.layer, (uint64_t)j, 0u, ef);
                if (klb) {
                    if (f.eps_kl) { const float4 e = ld4(f.eps_kl, j); ek[0] = e.x; ek[1] = e.y; ek[2] = e.z; ek[3] = e.w; }
                    else philox_normal4(seed, offs, LBBNN_STREAM_EPS_Z2 * 64u + f.layer, (uint64_t)j, 0u, ek);
                }
                const float qmv[4] = {qm.x, qm.y, qm.z, qm.w}, lvv[4] = {lv.x, lv.y, lv.z, lv.w};
                float a0[4], a1[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float ev = expf(lvv[c]), sd = sqrtf(ev);
                    a0[c] = qmv[c] + sd * ef[c];                                    // …MNF.py:183-185
                    a1[c] = qmv[c] + sd * ek[c];
                    if (klb) {
                        const float d = a1[c] - qmv[c];
                        acc[kInNV - 1] += (double)(-0.5f * 1.1447298858494002f - 0.5f * lvv[c] - 0.5f * ((d * d) / ev));   // :213-214
                    }
                }
                const float4 z0f = make_float4(a0[0], a0[1], a0[2], a0[3]), z0k = make_float4(a1[0], a1[1], a1[2], a1[3]);
                reinterpret_cast<float4*>(zf_s)[j] = z0f;
                reinterpret_cast<float4*>(zk_s)[j] = z0k;
                float4 u4[kInT], w4[kInT];
#pragma unroll
                for (int t = 0; t < kInT; ++t) {
                    u4[t] = zero4; w4[t] = zero4;                 // (if, not ?: -- a select between float4 OBJECTS goes through scratch)
                    if (t < NT) { u4[t] = ld4(f.u[t], j); w4[t] = ld4(f.w[t], j); }
                }
                int qx = 2 * kInT;
#pragma unroll
                for (int t = 0; t < kInT; ++t) {
                    acc[t] += (double)dot4(w4[t], z0k);                             // A_k[t]
                    acc[kInT + t] += (double)dot4(u4[t], w4[t]);                    // UW[t]
#pragma unroll
                    for (int s2 = 0; s2 < t; ++s2) acc[qx++] += (double)dot4(w4[t], u4[s2]);      // X[t][s]
                    acc[2 * kInT + kInT * (kInT - 1) / 2 + t] += (double)dot4(w4[t], z0f);        // A_f[t]
                }
            }
        }
#pragma unroll
        for (int k = 0; k < kInNV; ++k) acc[k] = wave_sum(acc[k]);
        if (lane == 0)
#pragma unroll
            for (int k = 0; k < kInNV; ++k) red[k][wv] = acc[k];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kInNV; ++k) {
            double t2 = 0.0;
#pragma unroll
            for (int w2 = 0; w2 < kRowsWG; ++w2) t2 += red[k][w2];
            acc[k] = t2;
        }
        // the scalar chains (flows2.py:87-95), every thread redundantly: forward draw (z flow only), KL draw (z then r flow)
        float thf[kInT], thk[kInT], ldf = 0.f, ldq = 0.f, ldr = 0.f;
        {
            int qx = 2 * kInT;
#pragma unroll
            for (int t = 0; t < kInT; ++t) {
                double inf = acc[2 * kInT + kInT * (kInT - 1) / 2 + t], ink = acc[t];
#pragma unroll
                for (int s2 = 0; s2 < t; ++s2) { const double x = acc[qx++]; inf += (double)thf[s2] * x; ink += (double)thk[s2] * x; }
                thf[t] = 0.f; thk[t] = 0.f;
                if (t < NT) {
                    const float bias = f.b[t][0], uw = (float)acc[kInT + t];
                    if (t < Tz) {
                        thf[t] = tanhf((float)inf + bias);
                        ldf += logf(fabsf(1.f + (1.f - thf[t] * thf[t]) * uw));
                    }
                    if (klb) {
                        thk[t] = tanhf((float)ink + bias);
                        const float ld = logf(fabsf(1.f + (1.f - thk[t] * thk[t]) * uw));
                        if (t < Tz) ldq += ld; else ldr += ld;
                    }
                }
            }
        }
#pragma unroll 1
        for (int h = 0; h < 2; ++h) {
            const int j = tid + kRowNT * h;
            if (j < nq) {
                float4 zf = zero4, zk = zero4;
                float ulast[kInT] = {0.f, 0.f, 0.f, 0.f};
                if (j < iq) {
                    zf = reinterpret_cast<const float4*>(zf_s)[j];                    // this thread's own z0 (no barrier needed)
                    zk = reinterpret_cast<const float4*>(zk_s)[j];
#pragma unroll
                    for (int t = 0; t < kInT; ++t)
                        if (t < NT) {
                            const float4 u = ld4(f.u[t], j);
                            ulast[t] = u.w;
                            if (t < Tz) { zf = fma4(u, thf[t], zf); zk = fma4(u, thk[t], zk); }
                        }
                }
                float4 rc = one4;
                if (a.r0_c && j < iq) rc = ld4(a.r0_c, j);
                if (!klb) zk = one4;
                reinterpret_cast<float4*>(zf_s)[j] = zf;
                reinterpret_cast<float4*>(zk_s)[j] = zk;
                reinterpret_cast<float4*>(rc_s)[j] = rc;
                if (first_wg && j < iq) {                                             // what K3 wrote, for K5 / backward
                    reinterpret_cast<float4*>(f.z_fwd)[j] = zf;
                    if (klb) {
                        reinterpret_cast<float4*>(f.z_kl)[j] = zk;
                        if (j == iq - 1) {                                            // z_b[-1]: last ELEMENT (:224)
                            float v = zk.w;
#pragma unroll
                            for (int t = 0; t < kInT; ++t) if (t >= Tz && t < NT) v += ulast[t] * thk[t];
                            f.scal[3] = v;
                        }
                    }
                }
            }
        }
        if (first_wg && tid == 0 && f.scal) {
            f.scal[4] = ldf;                                                          // logdet of sample_z(B) (:187)
            if (klb) { f.scal[0] = ldq; f.scal[1] = (float)acc[kInNV - 1]; f.scal[2] = ldr; }
        }
    } else if constexpr (!kDirectZ) {
#pragma unroll
        for (int h = 0; h < (64 * kRowG + kRowNT - 1) / kRowNT; ++h) {
            const int j = tid + kRowNT * h;
            if (j < nq) {
                const bool in = j < iq;
                float4 zf = one4, zk = one4, rc = one4;
                if (a.z_fwd && in) zf = ld4(a.z_fwd + (size_t)mem * P, j);
                if (a.z_kl && in) zk = ld4(a.z_kl, j);
                if (a.r0_c && in) rc = ld4(a.r0_c, j);
                reinterpret_cast<float4*>(zf_s)[j] = zf;
                reinterpret_cast<float4*>(zk_s)[j] = zk;
                reinterpret_cast<float4*>(rc_s)[j] = rc;
            }
        }
    }
    if constexpr (!kDirectZ) __syncthreads();

    // ---------------------------------------------------------------- one row per wave
    if (!has_row) return;
    const ElemConst ec = {a.mu_prior, a.log_sp, a.log_ap, a.log_1map, a.inv_2sp2};
    k1_f2 kl2 = {0.f, 0.f}, amu2 = kl2, avar2 = kl2;          // per-lane partial row sums, two lanes of a packed register
    const bool split = a.split == 1;
    float* const ewp = a.e_w ? a.e_w + (size_t)mem * a.O * P : nullptr;
    float* const vwp = mem == 0 ? a.var_w : nullptr;
    // kRowB float4 groups per lane at a time, all of their loads in flight together: rows up to 64 * 4 * 5 = 1280 floats
    // (every layer of the BASELINE configs but the 3072 / 4096-wide VD ones, which take the generic kernel) are one batch
    // F16S instantiation: launches in which SOME layer takes row-scaled fp16 operands (split == 2).  The host guarantees
    // G <= kRowB for EVERY layer of such a launch (one batch), and every row keeps its operands in registers until the row
    // maxima are known -- the other layers of the launch (the fp32 10-class head) store theirs from the same place, so the
    // loop below has one shape per instantiation
    const bool f16s = F16S && a.split >= 2;
    const bool vhi_only = a.split == 3;               // LBBNN_F_VAR1 operands: var_w = plain fp16 rows (the hi part alone)
    float4 ew_keep[F16S ? kRowB : 1], vw_keep[F16S ? kRowB : 1];
    for (int g0 = 0; g0 < G; g0 += kRowB) {
        if (INFLOW || g0 > 0) {                                  // (the first batch of a flow-less launch is in flight already)
#pragma unroll
            for (int g = 0; g < kRowB; ++g) {
                const int j = lane + 64 * (g0 + g);
                if (g0 + g < G && j < iq) { mu[g] = ld4(a.mu + rowoff, j); rho[g] = ld4(a.rho + rowoff, j); lam[g] = ld4(a.lambdal + rowoff, j); }
            }
        }
#pragma unroll
        for (int g = 0; g < kRowB; ++g) {
            if (g0 + g >= G) break;
            const int j = lane + 64 * (g0 + g);
            float4 ew = zero4, vw = zero4;
            if constexpr (kDirectZ) { if (g0 + g > 0) load_z(g0 + g, zfr[0], zkr[0], rcr[0]); }
            if (j < iq) {
                float4 zf, zk, rc;
                if constexpr (kDirectZ) { zf = zfr[0]; zk = zkr[0]; rc = rcr[0]; }
                else {
                    zf = reinterpret_cast<const float4*>(zf_s)[j]; zk = reinterpret_cast<const float4*>(zk_s)[j];
                    rc = reinterpret_cast<const float4*>(rc_s)[j];
                }
                const Elem2 ea = weight_elem_p2(k1_f2{mu[g].x, mu[g].y}, k1_f2{rho[g].x, rho[g].y}, k1_f2{lam[g].x, lam[g].y},
                                                k1_f2{zf.x, zf.y}, k1_f2{zk.x, zk.y}, k1_f2{rc.x, rc.y}, want_kl, want_act, ec);
                const Elem2 eb = weight_elem_p2(k1_f2{mu[g].z, mu[g].w}, k1_f2{rho[g].z, rho[g].w}, k1_f2{lam[g].z, lam[g].w},
                                                k1_f2{zf.z, zf.w}, k1_f2{zk.z, zk.w}, k1_f2{rc.z, rc.w}, want_kl, want_act, ec);
                ew = make_float4(ea.ew.x, ea.ew.y, eb.ew.x, eb.ew.y);
                vw = make_float4(ea.vw.x, ea.vw.y, eb.vw.x, eb.vw.y);
                kl2 += ea.kl + eb.kl;
                amu2 += ea.amu + eb.amu;
                avar2 += ea.avar + eb.avar;
            }
            if constexpr (F16S) {
                ew_keep[g] = ew; vw_keep[g] = vw;           // stored below, once the row maxima are known
            } else if (!split) {
                if (j < nq) {
                    if (ewp) reinterpret_cast<float4*>(ewp + (size_t)o * P)[j] = ew;
                    if (vwp) reinterpret_cast<float4*>(vwp + (size_t)o * P)[j] = vw;
                }
            } else {
                // 16-B units: the even lane of a pair holds k = 8m..8m+3, the odd lane k = 8m+4..8m+7; the even lane stores
                // the hi unit (its hi half | the partner's), the odd lane the lo unit next to it -- a wave covers 1 KiB contiguous
                const bool odd = lane & 1;
                const size_t at = split_hi_index((size_t)o, 4 * (j & ~1), P) + (odd ? kSplitLoOffset : 0);
                uint2 hi, lo;
                split4(ew, hi, lo);
                {
                    const uint32_t r0 = dpp_xor1(odd ? hi.x : lo.x), r1 = dpp_xor1(odd ? hi.y : lo.y);
                    const uint4 unit = make_uint4(odd ? r0 : hi.x, odd ? r1 : hi.y, odd ? lo.x : r0, odd ? lo.y : r1);
                    if (ewp && j < nq) *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(ewp) + at) = unit;
                }
                if (vwp) {
                    split4(vw, hi, lo);
                    const uint32_t r0 = dpp_xor1(odd ? hi.x : lo.x), r1 = dpp_xor1(odd ? hi.y : lo.y);
                    const uint4 unit = make_uint4(odd ? r0 : hi.x, odd ? r1 : hi.y, odd ? lo.x : r0, odd ? lo.y : r1);
                    if (j < nq) *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(vwp) + at) = unit;
                }
            }
        }
    }
    if constexpr (F16S) {
        // LBBNN_F_F16S: one exact power-of-two scale per row and operand, then hi | lo fp16 units exactly as the bf16 form
        // lays them out (even lane of a pair: the hi unit, odd lane: the lo unit).  Layers of the launch in another format
        // (split 0 / 1) store the operands they kept, unscaled, in their own format.
        float se = 1.f, ie = 1.f, sv = 1.f, iv = 1.f;
        if (f16s) {
            float me = 0.f, mv = 0.f;
#pragma unroll
            for (int g = 0; g < kRowB; ++g) {
                if (g >= G) break;
                me = fmaxf(fmaxf(fmaxf(fmaxf(me, fabsf(ew_keep[g].x)), fabsf(ew_keep[g].y)), fabsf(ew_keep[g].z)), fabsf(ew_keep[g].w));
                mv = fmaxf(fmaxf(fmaxf(fmaxf(mv, vw_keep[g].x), vw_keep[g].y), vw_keep[g].z), vw_keep[g].w);   // chains: v_max3_f32
            }
            me = wave_max(me); mv = wave_max(mv);
            pow2_scale(me, se, ie);
            pow2_scale(mv, sv, iv);
        }
        const bool odd = lane & 1;
#pragma unroll
        for (int g = 0; g < kRowB; ++g) {
            if (g >= G) break;
            const int j = lane + 64 * g;
            if (a.split == 0) {
                if (j < nq) {
                    if (ewp) reinterpret_cast<float4*>(ewp + (size_t)o * P)[j] = ew_keep[g];
                    if (vwp) reinterpret_cast<float4*>(vwp + (size_t)o * P)[j] = vw_keep[g];
                }
                continue;
            }
            const size_t at = split_hi_index((size_t)o, 4 * (j & ~1), P) + (odd ? kSplitLoOffset : 0);
            uint2 hi, lo;
            {
                const float4 w = ew_keep[g];
                if (f16s) split4_hs(w, se, hi, lo);
                else split4(w, hi, lo);
                const uint32_t r0 = dpp_xor1(odd ? hi.x : lo.x), r1 = dpp_xor1(odd ? hi.y : lo.y);
                const uint4 unit = make_uint4(odd ? r0 : hi.x, odd ? r1 : hi.y, odd ? lo.x : r0, odd ? lo.y : r1);
                if (ewp && j < nq) *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(ewp) + at) = unit;
            }
            if (vwp && vhi_only) {
                // one RNE fp16 value per weight, row-major with a row stride of P halves: 8 B per lane, a wave-store covers
                // 512 B contiguous (32 k x 2 B = the 64-B row of one K step in the GEMM's LDS image)
                const float4 w = vw_keep[g];
                if (j < nq) *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(vwp) + (size_t)o * P + 4 * j) =
                    make_uint2(cvt_pk_h(w.x * sv, w.y * sv), cvt_pk_h(w.z * sv, w.w * sv));
            } else if (vwp) {
                const float4 w = vw_keep[g];
                if (f16s) split4_hs(w, sv, hi, lo);
                else split4(w, hi, lo);
                const uint32_t r0 = dpp_xor1(odd ? hi.x : lo.x), r1 = dpp_xor1(odd ? hi.y : lo.y);
                const uint4 unit = make_uint4(odd ? r0 : hi.x, odd ? r1 : hi.y, odd ? lo.x : r0, odd ? lo.y : r1);
                if (j < nq) *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(vwp) + at) = unit;
            }
        }
        if (f16s && lane == 0) {
            if (a.e_scale) a.e_scale[o] = ie;
            if (a.v_scale && vwp) a.v_scale[o] = iv;
        }
    }
    if (want_kl || want_act) {
        const float kl = wave_sum(kl2.x + kl2.y), amu = wave_sum(amu2.x + amu2.y), avar = wave_sum(avar2.x + avar2.y);
        if (lane == 0) {
            if (want_kl) a.kl_rows[o] = kl;
            if (want_act) { a.act_mu[o] = amu; a.act_var[o] = avar; }
        }
    }
    if (lane == 0 && mem == 0 && a.bias_var && a.bias_rho) {
        const float sb = softplus_ref(a.bias_rho[o]);
        a.bias_var[o] = sb * sb;                      // bias.sigma**2, LBBNN-GP-MF-LRT.py:173
    }
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace

namespace lbbnn {

int make_weight_pass_args(WeightPassArgs& a, const float* mu, const float* rho, const float* lambdal,
                          const float* z_fwd, const float* z_kl, const float* r0_c, const float* bias_rho,
                          const lbbnn_priors_t* priors, void* e_w, void* var_w, int ld,
                          float* kl_rows, float* act_mu, float* act_var, float* bias_var, int O, int I, int split,
                          float* e_scale, float* v_scale) {
    if (!mu || !rho || !lambdal || !priors) return LBBNN_E_NULL;
    if (split >= 2 && ((e_w && !e_scale) || (var_w && !v_scale))) return LBBNN_E_NULL;
    if (split < 0 || split > 3) return LBBNN_E_FLAGS;
    if (O <= 0 || I <= 0) return LBBNN_E_SHAPE;
    if ((e_w || var_w) && (ld < I || (ld & 31))) return LBBNN_E_ALIGN;
    if ((act_mu == nullptr) != (act_var == nullptr)) return LBBNN_E_NULL;
    if (act_mu && (!z_kl || !r0_c)) return LBBNN_E_NULL;
    if (bias_var && !bias_rho) return LBBNN_E_NULL;
    if ((e_w && !aligned16(e_w)) || (var_w && !aligned16(var_w))) return LBBNN_E_ALIGN;
    a.mu = mu; a.rho = rho; a.lambdal = lambdal; a.z_fwd = z_fwd; a.z_kl = z_kl; a.r0_c = r0_c;
    a.bias_rho = bias_rho;
    a.e_w = static_cast<float*>(e_w); a.var_w = static_cast<float*>(var_w);
    a.kl_rows = kl_rows; a.act_mu = act_mu; a.act_var = act_var; a.bias_var = bias_var;
    a.O = O; a.I = I; a.ld = (e_w || var_w) ? ld : lbbnn_operand_ld(I);
    a.mu_prior = priors->mu_prior; a.sigma_prior = priors->sigma_prior; a.alpha_prior = priors->alpha_prior;
    a.log_sp = logf(priors->sigma_prior); a.log_ap = logf(priors->alpha_prior); a.log_1map = logf(1.f - priors->alpha_prior);
    a.inv_2sp2 = 1.f / (2.f * priors->sigma_prior * priors->sigma_prior);
    a.split = split;
    a.e_scale = e_scale; a.v_scale = v_scale;
    a.vec = ((I % 4 == 0) && aligned16(mu) && aligned16(rho) && aligned16(lambdal) &&
             (!z_fwd || aligned16(z_fwd)) && (!z_kl || aligned16(z_kl)) && (!r0_c || aligned16(r0_c))) ? 1 : 0;
    if (split && !a.vec) return LBBNN_E_ALIGN;          // split operands need the vector path (I % 4 == 0, aligned)
    // the row-scaled fp16 format needs the whole row in one batch of the row kernel (the scale is the row maximum's)
    if (split >= 2 && a.ld > 64 * 4 * 5) return LBBNN_E_SHAPE;
    return 0;
}

bool in_flow_eligible(const FlowArgs& f, const WeightPassArgs& w) {
    const int nt = f.zf.T + (f.want_kl ? f.rf.T : 0);
    return w.vec && w.ld <= 64 * 4 * kRowG && nt <= kInT && aligned16(f.q0_mean) && aligned16(f.q0_log_var) &&
           (!f.eps_fwd || aligned16(f.eps_fwd)) && (!f.eps_kl || aligned16(f.eps_kl)) && aligned16(f.z_fwd) &&
           (!f.want_kl || aligned16(f.z_kl)) && [&] {
               for (int t = 0; t < f.zf.T; ++t) if (!aligned16(f.zf.u[t]) || !aligned16(f.zf.w[t])) return false;
               for (int t = 0; t < (f.want_kl ? f.rf.T : 0); ++t) if (!aligned16(f.rf.u[t]) || !aligned16(f.rf.w[t])) return false;
               return true;
           }();
}

void make_in_flow(InFlow& o, const FlowArgs& f) {
    o = InFlow{};
    o.on = 1;
    o.q0_mean = f.q0_mean; o.q0_log_var = f.q0_log_var; o.eps_fwd = f.eps_fwd; o.eps_kl = f.want_kl ? f.eps_kl : nullptr;
    o.rng = f.rng; o.z_fwd = f.z_fwd; o.z_kl = f.z_kl; o.scal = f.scal;
    o.Tz = f.zf.T; o.Tr = f.want_kl ? f.rf.T : 0; o.want_kl = f.want_kl; o.layer = f.layer;
    for (int t = 0; t < o.Tz; ++t) { o.u[t] = f.zf.u[t]; o.w[t] = f.zf.w[t]; o.b[t] = f.zf.b[t]; }
    for (int t = 0; t < o.Tr; ++t) { o.u[o.Tz + t] = f.rf.u[t]; o.w[o.Tz + t] = f.rf.w[t]; o.b[o.Tz + t] = f.rf.b[t]; }
}

int launch_weight_pass(const WeightPassArgs* a, int n, hipStream_t s, uint64_t* rng, uint64_t* rng_snap, uint64_t advance,
                       const InFlow* flows, int members) {
    bool rows_ok = true;
    for (int i = 0; i < n; ++i) rows_ok = rows_ok && a[i].vec && a[i].ld <= 64 * 4 * kRowG;
    // LBBNN_K1_ROWS=0 (environment, read once): A/B switch for measurements -- the one-row-per-workgroup kernel of round 1
    static const bool rows_allowed = [] { const char* e = getenv("LBBNN_K1_ROWS"); return !(e && e[0] == '0'); }();
    rows_ok = rows_ok && rows_allowed;
    bool any_f16 = false;
    for (int i = 0; i < n; ++i) any_f16 = any_f16 || a[i].split >= 2;
    if (any_f16) {
        for (int i = 0; i < n; ++i) if (!a[i].vec || a[i].ld > 64 * 4 * kRowB) return LBBNN_E_SHAPE;   // every layer: one batch
        if (members > 1) return LBBNN_E_FLAGS;               // (the ensemble's member dimension keeps the bf16 format)
        rows_ok = true;
    }
    if (rows_ok) {
        WeightRowsBatch bt{};
        bt.rng = rng; bt.rng_snap = (rng && rng_snap) ? rng_snap : nullptr; bt.advance = advance;
        int wgs = 0, maxld = 0;
        bool flow_on = false;
        for (int i = 0; i < n && flows; ++i) flow_on = flow_on || flows[i].on;
        const int rw = flow_on ? RowsCfg<true>::RW : RowsCfg<false>::RW;      // rows per workgroup of the instantiation launched
        for (int i = 0; i < n; ++i) {
            bt.l[i] = a[i];
            if (flows) bt.f[i] = flows[i];
            if (bt.f[i].on && advance) return LBBNN_E_FLAGS;     // the in-kernel flows read the live offset in every workgroup
            wgs += (a[i].O + rw - 1) / rw; bt.wg_end[i] = wgs;
            maxld = a[i].ld > maxld ? a[i].ld : maxld;
        }
        for (int i = n; i < LBBNN_MAX_LAYERS; ++i) bt.wg_end[i] = wgs;
        bt.n = n;
        bt.members = members;
        if (members > 1) for (int i = 0; i < n; ++i) if (bt.f[i].on || a[i].kl_rows || a[i].act_mu) return LBBNN_E_FLAGS;
        bool any_flow = false;
        for (int i = 0; i < n; ++i) any_flow = any_flow || bt.f[i].on;
        const dim3 grid(wgs, any_f16 ? 1 : (members > 1 ? members : 1));
        // (one-row workgroups hold the per-column vectors in registers: no LDS, nothing for the dispatcher to count)
        const size_t lds = (!flow_on && RowsCfg<false>::RW == 1) ? 0 : (size_t)3 * maxld * sizeof(float);
        if (any_f16 && any_flow)  hipLaunchKernelGGL((weight_rows_kernel<true, true>), grid, dim3(RowsCfg<true>::NT), lds, s, bt);
        else if (any_f16)         hipLaunchKernelGGL((weight_rows_kernel<true, false>), grid, dim3(RowsCfg<false>::NT), lds, s, bt);
        else if (any_flow)        hipLaunchKernelGGL((weight_rows_kernel<false, true>), grid, dim3(RowsCfg<true>::NT), lds, s, bt);
        else                      hipLaunchKernelGGL((weight_rows_kernel<false, false>), grid, dim3(RowsCfg<false>::NT), lds, s, bt);
        return (int)hipGetLastError();
    }
    if (members > 1) return LBBNN_E_ALIGN;                  // the member dimension exists in the row kernel only
    if (flows) for (int i = 0; i < n; ++i) if (flows[i].on) return LBBNN_E_ALIGN;    // (callers test in_flow_eligible first)
    WeightPassBatch bt;
    bt.rng = rng; bt.rng_snap = (rng && rng_snap) ? rng_snap : nullptr; bt.advance = advance;
    int rows = 0;
    for (int i = 0; i < n; ++i) { bt.l[i] = a[i]; rows += a[i].O; bt.row_end[i] = rows; }
    for (int i = n; i < LBBNN_MAX_LAYERS; ++i) bt.row_end[i] = rows;
    bt.n = n;
    hipLaunchKernelGGL(weight_pass_kernel, dim3(rows), dim3(256), 0, s, bt);
    return (int)hipGetLastError();
}

}  // namespace lbbnn

extern "C" int lbbnn_operand_ld(int I) { return I <= 0 ? 0 : ((I + 31) / 32) * 32; }

extern "C" int lbbnn_weight_pass_f16(const float* mu, const float* rho, const float* lambdal,
                                     const float* z_fwd, const float* z_kl, const float* r0_c,
                                     const float* bias_rho, const lbbnn_priors_t* priors,
                                     void* e_w, void* var_w, int ld, float* e_scale, float* v_scale,
                                     float* kl_rows, float* act_mu, float* act_var, float* bias_var,
                                     int O, int I, int flags, void* stream) {
    if (flags & ~LBBNN_F_VAR1) return LBBNN_E_FLAGS;
    lbbnn::WeightPassArgs a;
    const int rc = lbbnn::make_weight_pass_args(a, mu, rho, lambdal, z_fwd, z_kl, r0_c, bias_rho, priors, e_w, var_w, ld,
                                                kl_rows, act_mu, act_var, bias_var, O, I, (flags & LBBNN_F_VAR1) ? 3 : 2,
                                                e_scale, v_scale);
    if (rc) return rc;
    return lbbnn::launch_weight_pass(&a, 1, static_cast<hipStream_t>(stream));
}

extern "C" int lbbnn_weight_pass(const float* mu, const float* rho, const float* lambdal,
                                 const float* z_fwd, const float* z_kl, const float* r0_c,
                                 const float* bias_rho, const lbbnn_priors_t* priors,
                                 void* e_w, void* var_w, int ld,
                                 float* kl_rows, float* act_mu, float* act_var, float* bias_var,
                                 int O, int I, int flags, void* stream) {
    if (flags & ~LBBNN_F_SPLIT16) return LBBNN_E_FLAGS;
    lbbnn::WeightPassArgs a;
    const int rc = lbbnn::make_weight_pass_args(a, mu, rho, lambdal, z_fwd, z_kl, r0_c, bias_rho, priors, e_w, var_w, ld,
                                                kl_rows, act_mu, act_var, bias_var, O, I, (flags & LBBNN_F_SPLIT16) ? 1 : 0);
    if (rc) return rc;
    return lbbnn::launch_weight_pass(&a, 1, static_cast<hipStream_t>(stream));
}
