// K6 (baseline LBBNN gate x Gaussian weight sampling + Monte-Carlo log-probabilities) and
// K7 (variational-dropout operand pass) -- see include/lbbnn.h.  Both are one-pass HBM-bound kernels
// that feed the same dual-moment GEMM.
#include <cmath>
#include "lbbnn_device.h"
#include "lbbnn_internal.h"

namespace {

using namespace lbbnn;

__device__ __forceinline__ uint32_t bf16_rne_bits(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
// store one operand value at [o][i] either as fp32 or in the split bf16 hi|lo layout
// split: 0 = fp32, 1 = bf16 hi|lo (LBBNN_F_SPLIT16), 2 = fp16 in the hi unit, zero lo (LBBNN_F_HALF16: the single-product
// fp16 form of the variational-dropout operands)
__device__ __forceinline__ void store_operand(void* base, int split, size_t O, int ld, int o, int i, float v) {
    if (!split) { static_cast<float*>(base)[(size_t)o * ld + i] = v; return; }
    uint16_t* const w = static_cast<uint16_t*>(base);           // split hi|lo layout (lbbnn_device.h)
    const size_t at = split_hi_index((size_t)o, i, ld);
    if (split == 2) {
        const _Float16 h16 = (_Float16)v;                       // RNE; subnormals kept (the f16 MFMA honours them)
        w[at] = __builtin_bit_cast(uint16_t, h16);
        w[at + kSplitLoOffset] = 0;
        (void)O;
        return;
    }
    const uint32_t h = bf16_rne_bits(v);
    w[at] = (uint16_t)h;
    w[at + kSplitLoOffset] = (uint16_t)bf16_rne_bits(v - __uint_as_float(h << 16));
    (void)O;
}

// ------------------------------------------------------------------------------------------------ K6
// One workgroup per output row; rows[0..3][o] = row sums of
//   0: GaussGamma weight integrand (without the scalar C*gamma part folded: kept exact as the reference sums it)
//   1: BetaBinomial integrand      2: Gaussian.full_log_prob integrand     3: Bernoulli.log_prob integrand
__global__ __launch_bounds__(256) void gate_sample_kernel(const lbbnn_gate_args_t a, const uint64_t* rng) {
    __shared__ double red[4][4];
    const int o = blockIdx.x, tid = threadIdx.x;
    const size_t ro = (size_t)o * a.I;
    const int split = (a.flags & LBBNN_F_SPLIT16) ? 1 : 0;
    uint64_t seed = 0, offs = 0;
    if (a.mode == LBBNN_MODE_SAMPLE && !a.eps_w) { seed = rng[0]; offs = rng[1]; }
    float C = 0.f, cbb = 0.f, pa = 0.f, pb = 0.f, tau = 0.f;
    if (a.want_lp) {
        const float wa = a.weight_a[0], wb = a.weight_b[0];
        tau = a.tau_w[0];
        // a*log(b) + (a-0.5)*tau - b*tau - lgamma(a) - 0.5*log(2*pi)      LBBNN-GP-MF.py:144-145
        C = wa * logf(wb) + (wa - 0.5f) * tau - wb * tau - lgammaf(wa) - 0.5f * 1.8378770664093453f;
        pa = a.pa[0]; pb = a.pb[0];
        // lgamma(1) + lgamma(pa+pb) - lgamma(1+pa+pb) - lgamma(pa) - lgamma(pb)   (the g-independent terms of :167-173)
        cbb = lgammaf(pa + pb) - lgammaf(1.f + pa + pb) - lgammaf(pa) - lgammaf(pb);
    }
    double s_gg = 0.0, s_bb = 0.0, s_fq = 0.0, s_be = 0.0;
    for (int i = tid; i < a.ld; i += 256) {
        float w = 0.f;
        if (i < a.I) {
            const float mu = a.mu[ro + i];
            const float g = a.cgamma ? a.cgamma[ro + i] : 1.f;
            float sigma = 0.f;
            if (a.mode == LBBNN_MODE_SAMPLE) {
                float e;
                if (a.eps_w) e = a.eps_w[ro + i];
                else { float n[4]; philox_normal4(seed, offs, LBBNN_STREAM_EPS_W * 64u + a.layer_id, (uint64_t)o, (uint32_t)(i >> 2), n); e = n[i & 3]; }
                sigma = softplus_ref(a.rho[ro + i]);
                w = g * (mu + sigma * e);                                          // :232-233
            } else if (a.mode == LBBNN_MODE_MEDIMEAN) {
                w = g * mu;                                                        // :237
            } else {
                w = a.alpha_attr[ro + i] * mu;                                     // :241
            }
            if (a.want_lp) {
                if (a.mode != LBBNN_MODE_SAMPLE) sigma = softplus_ref(a.rho[ro + i]);
                const float g_wp = (a.exact & 1) ? rintf(g) : g;
                s_gg += (double)(g_wp * C - tau * (w * w) + (1.f - g_wp) + 1e-8f);                  // :144-150
                const float g_bb = (a.exact & 4) ? rintf(g) : g;
                s_bb += (double)(cbb + lgammaf(1.f + pb - g_bb) - lgammaf(2.f - g_bb));              // :167-173
                const float d = w - mu;
                const float lp = -0.9189385332046727f - logf(sigma) - (d * d) / (2.f * sigma * sigma);   // :94-97
                s_fq += (double)logf(g * expf(lp) + (1.f - g) + 1e-8f);                              // :99-101
                const float al = a.gamma_alpha[ro + i];
                const float g_be = (a.exact & 8) ? rintf(g) : g;
                s_be += (double)(g_be * logf(al + 1e-8f) + (1.f - g_be) * logf(1.f - al + 1e-8f));   // :125-127
            }
        }
        if (a.w_out) store_operand(a.w_out, split, a.O, a.ld, o, i, w);
    }
    if (a.want_lp) {
        s_gg = wave_sum(s_gg); s_bb = wave_sum(s_bb); s_fq = wave_sum(s_fq); s_be = wave_sum(s_be);
        const int lane = tid & 63, wv = tid >> 6;
        if (lane == 0) { red[0][wv] = s_gg; red[1][wv] = s_bb; red[2][wv] = s_fq; red[3][wv] = s_be; }
        __syncthreads();
        if (tid < 4) a.rows[(size_t)tid * a.O + o] = (float)((red[tid][0] + red[tid][1]) + (red[tid][2] + red[tid][3]));
    }
}

// bias sample + bias log-probabilities + final scalars (single workgroup)
__global__ __launch_bounds__(256) void gate_finalize_kernel(const lbbnn_gate_args_t a, const uint64_t* rng) {
    __shared__ double scratch[4];
    const int tid = threadIdx.x;
    uint64_t seed = 0, offs = 0;
    if (a.mode == LBBNN_MODE_SAMPLE && !a.eps_b) { seed = rng[0]; offs = rng[1]; }
    double r0 = 0, r1 = 0, r2 = 0, r3 = 0, gb = 0, qb = 0;
    for (int o = tid; o < a.O; o += 256) {
        const float sb = softplus_ref(a.bias_rho[o]);
        float b = a.bias_mu[o];
        if (a.mode == LBBNN_MODE_SAMPLE) {
            float e;
            if (a.eps_b) e = a.eps_b[o];
            else { float n[4]; philox_normal4(seed, offs, LBBNN_STREAM_EPS_B * 64u + a.layer_id, (uint64_t)(o >> 2), 0u, n); e = n[o & 3]; }
            b = a.bias_mu[o] + sb * e;                                                                // :234
        }
        a.bias_out[o] = b;
        if (a.want_lp) {
            r0 += (double)a.rows[o]; r1 += (double)a.rows[(size_t)a.O + o];
            r2 += (double)a.rows[2 * (size_t)a.O + o]; r3 += (double)a.rows[3 * (size_t)a.O + o];
            const float ba = a.bias_a[o], bb = a.bias_b[o], tb = a.tau_b[o];
            const float Cb = ba * logf(bb) + (ba - 0.5f) * tb - bb * tb - lgammaf(ba) - 0.5f * 1.8378770664093453f;
            gb += (double)(Cb - tb * (b * b) + 0.f + 1e-8f);                                          // GaussGamma(bias, 1)
            const float d = b - a.bias_mu[o];
            qb += (double)(-0.9189385332046727f - logf(sb) - (d * d) / (2.f * sb * sb));              // Gaussian.log_prob :89-92
        }
    }
    if (!a.want_lp) return;
    r0 = block_sum<double, 4>(r0, scratch); r1 = block_sum<double, 4>(r1, scratch);
    r2 = block_sum<double, 4>(r2, scratch); r3 = block_sum<double, 4>(r3, scratch);
    gb = block_sum<double, 4>(gb, scratch); qb = block_sum<double, 4>(qb, scratch);
    if (tid == 0) {
        *a.log_prior = (float)(r0 + gb + r1);          // :247-249
        *a.log_q = (float)(r2 + r3 + qb);              // :250-251
    }
}

// ------------------------------------------------------------------------------------------------ K6b
// Backward of the sampled baseline layer (LBBNN-GP-MF.py:228-255 under loss.backward(), :331-337): one pass over (O,I)
// turns the upstream gradients -- dW = G^T x of F.linear (:255), and the scalars g_lp = dL/dlog_prior, g_lq = dL/dlog_q --
// into d weight_mu, d weight_rho, d cgamma, d gamma.alpha and the row sums the scalar parameters need; a single-workgroup
// tail adds those up (weight_a, weight_b, tau_w, pa, pb) and does the (O)-sized bias chain.  Also rewrites the sampled
// weight W as a dense fp32 (O,I) matrix for the dX product.  eps_w / eps_b are re-created from the forward's Philox state.
// psi = digamma by upward recurrence to x >= 6 and the asymptotic series (|error| < 1e-6 for x >= 0.5).
__device__ __forceinline__ float digammaf_pos(float x) {
    float r = 0.f;
#pragma unroll 1
    while (x < 6.f) { r -= 1.f / x; x += 1.f; }
    const float i = 1.f / x, i2 = i * i;
    return r + logf(x) - 0.5f * i - i2 * (0.083333333f - i2 * (0.0083333333f - i2 * 0.003968254f));
}

struct GateScal { float C, tau, pa, pb, glp, glq; };

__global__ __launch_bounds__(256) void gate_backward_kernel(const lbbnn_gate_bwd_args_t a, const uint64_t* rng) {
    __shared__ double red[3][4];
    const int o = blockIdx.x, tid = threadIdx.x;
    const size_t ro = (size_t)o * a.I;
    uint64_t seed = 0, offs = 0;
    if (!a.eps_w) { seed = rng[0]; offs = rng[1]; }
    const float wa = a.weight_a[0], wb = a.weight_b[0], tau = a.tau_w[0], pb = a.pb[0];
    const float C = wa * logf(wb) + (wa - 0.5f) * tau - wb * tau - lgammaf(wa) - 0.5f * 1.8378770664093453f;
    const float glp = a.g_lp ? a.g_lp[0] : 0.f, glq = a.g_lq ? a.g_lq[0] : 0.f;
    double s_c = 0.0, s_w2 = 0.0, s_psi = 0.0;
    for (int i = tid; i < a.I; i += 256) {
        const float mu = a.mu[ro + i], g = a.cgamma[ro + i];
        float e;
        if (a.eps_w) e = a.eps_w[ro + i];
        else { float n[4]; philox_normal4(seed, offs, LBBNN_STREAM_EPS_W * 64u + a.layer_id, (uint64_t)o, (uint32_t)(i >> 2), n); e = n[i & 3]; }
        const float rho = a.rho[ro + i];
        const float sigma = softplus_ref(rho);
        const float ws = mu + sigma * e, w = g * ws;
        if (a.w_out) a.w_out[ro + i] = w;
        const float d = w - mu, is2 = 1.f / (sigma * sigma);
        const float lp = -0.9189385332046727f - logf(sigma) - (d * d) * 0.5f * is2;
        const float E = expf(lp), D = g * E + (1.f - g) + 1e-8f;
        const float q = glq * (g * E / D);                         // g_lq * d full_log_prob / d lp
        const float A = (a.dW ? a.dW[ro + i] : 0.f) + glp * (-2.f * tau * w) + q * (-d * is2);       // dL/dW
        const float g_wp = (a.exact & 1) ? rintf(g) : g, g_bb = (a.exact & 4) ? rintf(g) : g, g_be = (a.exact & 8) ? rintf(g) : g;
        const float al = a.gamma_alpha[ro + i];
        float dg = A * ws + glq * ((E - 1.f) / D);
        if (!(a.exact & 1)) dg += glp * (C - 1.f);
        const float psi1 = digammaf_pos(1.f + pb - g_bb);
        if (!(a.exact & 4)) dg += glp * (-psi1 + digammaf_pos(2.f - g_bb));
        if (!(a.exact & 8)) dg += glq * (logf(al + 1e-8f) - logf(1.f - al + 1e-8f));
        const float dws = A * g;
        a.d_mu[ro + i] = dws + q * (d * is2);
        const float dsig = dws * e + q * (-1.f / sigma + (d * d) * is2 / sigma);
        a.d_rho[ro + i] = dsig / (1.f + expf(-rho));
        a.d_cgamma[ro + i] = dg;
        a.d_alpha[ro + i] = glq * (g_be / (al + 1e-8f) - (1.f - g_be) / (1.f - al + 1e-8f));
        s_c += (double)g_wp; s_w2 += (double)(w * w); s_psi += (double)psi1;
    }
    s_c = wave_sum(s_c); s_w2 = wave_sum(s_w2); s_psi = wave_sum(s_psi);
    const int lane = tid & 63, wv = tid >> 6;
    if (lane == 0) { red[0][wv] = s_c; red[1][wv] = s_w2; red[2][wv] = s_psi; }
    __syncthreads();
    if (tid < 3) a.rows[(size_t)tid * a.O + o] = (float)((red[tid][0] + red[tid][1]) + (red[tid][2] + red[tid][3]));
}

__global__ __launch_bounds__(256) void gate_backward_tail_kernel(const lbbnn_gate_bwd_args_t a, const uint64_t* rng) {
    __shared__ double scratch[4];
    const int tid = threadIdx.x;
    uint64_t seed = 0, offs = 0;
    if (!a.eps_b) { seed = rng[0]; offs = rng[1]; }
    const float glp = a.g_lp ? a.g_lp[0] : 0.f, glq = a.g_lq ? a.g_lq[0] : 0.f;
    double r0 = 0, r1 = 0, r2 = 0;
    for (int o = tid; o < a.O; o += 256) {
        r0 += (double)a.rows[o]; r1 += (double)a.rows[(size_t)a.O + o]; r2 += (double)a.rows[2 * (size_t)a.O + o];
        const float rho = a.bias_rho[o], sb = softplus_ref(rho), bm = a.bias_mu[o];
        float e;
        if (a.eps_b) e = a.eps_b[o];
        else { float n[4]; philox_normal4(seed, offs, LBBNN_STREAM_EPS_B * 64u + a.layer_id, (uint64_t)(o >> 2), 0u, n); e = n[o & 3]; }
        const float b = bm + sb * e, d = b - bm, is2 = 1.f / (sb * sb);
        const float ba = a.bias_a[o], bb = a.bias_b[o], tb = a.tau_b[o];
        const float db = (a.g_sum ? a.g_sum[o] : 0.f) + glp * (-2.f * tb * b) + glq * (-d * is2);      // dL/db
        a.d_bias_mu[o] = db + glq * (d * is2);
        a.d_bias_rho[o] = (db * e + glq * (-1.f / sb + (d * d) * is2 / sb)) / (1.f + expf(-rho));
        a.d_bias_a[o] = glp * (logf(bb) + tb - digammaf_pos(ba));
        a.d_bias_b[o] = glp * (ba / bb - tb);
        a.d_tau_b[o] = glp * ((ba - 0.5f) - bb - b * b);
    }
    r0 = block_sum<double, 4>(r0, scratch); r1 = block_sum<double, 4>(r1, scratch); r2 = block_sum<double, 4>(r2, scratch);
    if (tid == 0) {
        const float wa = a.weight_a[0], wb = a.weight_b[0], tau = a.tau_w[0], pa = a.pa[0], pb = a.pb[0];
        const double N = (double)a.O * (double)a.I;
        a.d_scalars[0] = glp * (float)(r0 * (double)(logf(wb) + tau - digammaf_pos(wa)));             // d weight_a
        a.d_scalars[1] = glp * (float)(r0 * (double)(wa / wb - tau));                                 // d weight_b
        a.d_scalars[2] = glp * (float)(r0 * (double)((wa - 0.5f) - wb) - r1);                         // d tau_w
        const float common = digammaf_pos(pa + pb) - digammaf_pos(1.f + pa + pb);
        a.d_scalars[3] = glp * (float)(N * (double)(common - digammaf_pos(pa)));                      // d pa
        a.d_scalars[4] = glp * (float)(r2 + N * (double)(common - digammaf_pos(pb)));                 // d pb
    }
}

// ------------------------------------------------------------------------------------------------ K7
// 32x32 tiles of theta (I,O) through LDS: coalesced reads along O, coalesced writes along I.
__global__ __launch_bounds__(256) void vd_operands_kernel(const float* __restrict__ theta, void* e_w, void* var_w,
                                                          int ld, int I, int O, int split) {
    __shared__ float tile[32][33];
    const int i0 = blockIdx.x * 32, o0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;     // 32 x 8
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = i0 + ty + 8 * r, o = o0 + tx;
        tile[ty + 8 * r][tx] = (i < I && o < O) ? theta[(size_t)i * O + o] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int o = o0 + ty + 8 * r, i = i0 + tx;
        if (o < O && i < ld) {
            const float t = tile[tx][ty + 8 * r];                // zero for i >= I: keeps the operand tail zero-filled
            store_operand(e_w, split, O, ld, o, i, t);
            store_operand(var_w, split, O, ld, o, i, t * t);
        }
    }
}

// Backward operands of a Bayesian layer in ONE pass over its parameters: (e_w)^T = (mu * alpha * z)^T and
// (var_w)^T = (sigma^2 alpha^2)^T as GEMM operands [I][ld(O)] (the dX products contract over O) -- instead of the weight
// pass (fp32 operands) followed by two transposes.  Same tiling as K7; same elementwise forms as the weight pass.
__global__ __launch_bounds__(256) void weight_operands_t_kernel(const float* __restrict__ mu, const float* __restrict__ rho,
                                                                const float* __restrict__ lam, const float* __restrict__ z,
                                                                void* e_t, void* v_t, int ld, int O, int I, int split) {
    __shared__ float te[32][33], tv[32][33];
    const int o0 = blockIdx.x * 32, i0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;     // 32 x 8
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int o = o0 + ty + 8 * r, i = i0 + tx;
        float e = 0.f, v = 0.f;
        if (o < O && i < I) {
            const size_t k = (size_t)o * I + i;
            const float alpha = k1_alpha(lam[k]);
            e = (mu[k] * alpha) * (z ? z[i] : 1.f);                                  // same association as the weight pass
            if (v_t) { const float sg = k1_sigma(rho[k]); v = (sg * sg) * (alpha * alpha); }
        }
        te[ty + 8 * r][tx] = e; tv[ty + 8 * r][tx] = v;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = i0 + ty + 8 * r, o = o0 + tx;
        if (i < I && o < ld) {                                   // zero for o >= O: keeps the operand tail zero-filled
            store_operand(e_t, split, I, ld, i, o, te[tx][ty + 8 * r]);
            if (v_t) store_operand(v_t, split, I, ld, i, o, tv[tx][ty + 8 * r]);
        }
    }
}

// generic (R,C) -> operand [C][ld] transpose, optional square (same tiling as K7)
__global__ __launch_bounds__(256) void transpose_operand_kernel(const float* __restrict__ src, int R, int C, int lds_src,
                                                                void* dst, int ld, int square, int split) {
    __shared__ float tile[32][33];
    const int r0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = r0 + ty + 8 * k, c = c0 + tx;
        float v = (r < R && c < C) ? src[(size_t)r * lds_src + c] : 0.f;
        tile[ty + 8 * k][tx] = square ? v * v : v;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = c0 + ty + 8 * k, r = r0 + tx;
        if (c < C && r < ld) store_operand(dst, split, C, ld, c, r, tile[tx][ty + 8 * k]);
    }
}

}  // namespace

// (R,C) -> operand [R][ld] without a transpose: row r of dst = row r of src (squared if asked), zero tail, fp32 or split.
// The input-gradient operands of a variational-dropout layer: theta (n,m) IS W^T for dX = G . theta^T.
__global__ __launch_bounds__(256) void format_operand_kernel(const float* __restrict__ src, int R, int C, int lds_src,
                                                             void* dst, int ld, int square, int split) {
    const int r = blockIdx.y;
    for (int c = blockIdx.x * 256 + threadIdx.x; c < ld; c += gridDim.x * 256) {
        float v = c < C ? src[(size_t)r * lds_src + c] : 0.f;
        if (square) v *= v;
        store_operand(dst, split, R, ld, r, c, v);
    }
}

extern "C" int lbbnn_format_operand(const float* src, int R, int C, int lds_src, void* dst, int ld, int square, int flags,
                                    void* stream) {
    if (!src || !dst) return LBBNN_E_NULL;
    if (R <= 0 || C <= 0 || lds_src < C) return LBBNN_E_SHAPE;
    if (ld < C || (ld & 31)) return LBBNN_E_ALIGN;
    if (flags & ~LBBNN_F_SPLIT16) return LBBNN_E_FLAGS;
    hipLaunchKernelGGL(format_operand_kernel, dim3((ld + 1023) / 1024, R), dim3(256), 0, static_cast<hipStream_t>(stream), src, R,
                       C, lds_src, dst, ld, square ? 1 : 0, (flags & LBBNN_F_SPLIT16) ? 1 : 0);
    return (int)hipGetLastError();
}

extern "C" int lbbnn_transpose_operand(const float* src, int R, int C, int lds_src, void* dst, int ld,
                                       int square, int flags, void* stream) {
    if (!src || !dst) return LBBNN_E_NULL;
    if (R <= 0 || C <= 0 || lds_src < C) return LBBNN_E_SHAPE;
    if (ld < R || (ld & 31)) return LBBNN_E_ALIGN;
    if (flags & ~LBBNN_F_SPLIT16) return LBBNN_E_FLAGS;
    hipLaunchKernelGGL(transpose_operand_kernel, dim3((ld + 31) / 32, (C + 31) / 32), dim3(256), 0,
                       static_cast<hipStream_t>(stream), src, R, C, lds_src, dst, ld, square ? 1 : 0,
                       (flags & LBBNN_F_SPLIT16) ? 1 : 0);
    return (int)hipGetLastError();
}

extern "C" int lbbnn_weight_operands_t(const float* mu, const float* rho, const float* lambdal, const float* z,
                                       void* e_t, void* v_t, int ld, int O, int I, int flags, void* stream) {
    if (!mu || !lambdal || !e_t || (v_t && !rho)) return LBBNN_E_NULL;
    if (O <= 0 || I <= 0) return LBBNN_E_SHAPE;
    if (ld < O || (ld & 31)) return LBBNN_E_ALIGN;
    if (flags & ~LBBNN_F_SPLIT16) return LBBNN_E_FLAGS;
    hipLaunchKernelGGL(weight_operands_t_kernel, dim3((ld + 31) / 32, (I + 31) / 32), dim3(256), 0,
                       static_cast<hipStream_t>(stream), mu, rho, lambdal, z, e_t, v_t, ld, O, I, (flags & LBBNN_F_SPLIT16) ? 1 : 0);
    return (int)hipGetLastError();
}

extern "C" int lbbnn_gate_sample(const lbbnn_gate_args_t* p, const uint64_t* rng, void* stream) {
    if (!p) return LBBNN_E_NULL;
    const lbbnn_gate_args_t& a = *p;
    if (!a.mu || !a.bias_mu || !a.bias_out) return LBBNN_E_NULL;
    if (a.O <= 0 || a.I <= 0) return LBBNN_E_SHAPE;
    if (a.mode < 0 || a.mode > 2) return LBBNN_E_FLAGS;
    if (a.flags & ~LBBNN_F_SPLIT16) return LBBNN_E_FLAGS;
    if (a.w_out && (a.ld < a.I || (a.ld & 31))) return LBBNN_E_ALIGN;
    if (a.mode == LBBNN_MODE_SAMPLE && (!a.rho || !a.bias_rho)) return LBBNN_E_NULL;
    if (a.mode != LBBNN_MODE_MEAN && !a.cgamma) return LBBNN_E_NULL;
    if (a.mode == LBBNN_MODE_MEAN && !a.alpha_attr) return LBBNN_E_NULL;
    if (a.mode == LBBNN_MODE_SAMPLE && (!a.eps_w || !a.eps_b) && !rng) return LBBNN_E_NOISE;
    if (a.want_lp && (!a.rho || !a.bias_rho || !a.cgamma || !a.gamma_alpha || !a.weight_a || !a.weight_b || !a.tau_w ||
                      !a.pa || !a.pb || !a.bias_a || !a.bias_b || !a.tau_b || !a.rows || !a.log_prior || !a.log_q))
        return LBBNN_E_NULL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(gate_sample_kernel, dim3(a.O), dim3(256), 0, s, a, rng);
    hipLaunchKernelGGL(gate_finalize_kernel, dim3(1), dim3(256), 0, s, a, rng);
    return (int)hipGetLastError();
}

extern "C" int lbbnn_gate_backward(const lbbnn_gate_bwd_args_t* p, const uint64_t* rng, void* stream) {
    if (!p) return LBBNN_E_NULL;
    const lbbnn_gate_bwd_args_t& a = *p;
    if (!a.mu || !a.rho || !a.gamma_alpha || !a.cgamma || !a.bias_mu || !a.bias_rho || !a.bias_a || !a.bias_b || !a.tau_b ||
        !a.weight_a || !a.weight_b || !a.tau_w || !a.pa || !a.pb) return LBBNN_E_NULL;
    if (!a.d_mu || !a.d_rho || !a.d_cgamma || !a.d_alpha || !a.d_bias_mu || !a.d_bias_rho || !a.d_bias_a || !a.d_bias_b ||
        !a.d_tau_b || !a.d_scalars || !a.rows) return LBBNN_E_NULL;
    if ((!a.eps_w || !a.eps_b) && !rng) return LBBNN_E_NOISE;
    if (a.O <= 0 || a.I <= 0) return LBBNN_E_SHAPE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(gate_backward_kernel, dim3(a.O), dim3(256), 0, s, a, rng);
    hipLaunchKernelGGL(gate_backward_tail_kernel, dim3(1), dim3(256), 0, s, a, rng);
    return (int)hipGetLastError();
}

extern "C" int lbbnn_vd_operands(const float* theta, void* e_w, void* var_w, int ld, int I, int O, int flags, void* stream) {
    if (!theta || !e_w || !var_w) return LBBNN_E_NULL;
    if (I <= 0 || O <= 0) return LBBNN_E_SHAPE;
    if (ld < I || (ld & 31)) return LBBNN_E_ALIGN;
    if (flags & ~(LBBNN_F_SPLIT16 | LBBNN_F_HALF16)) return LBBNN_E_FLAGS;
    if ((flags & LBBNN_F_HALF16) && !(flags & LBBNN_F_SPLIT16)) return LBBNN_E_FLAGS;
    hipLaunchKernelGGL(vd_operands_kernel, dim3((ld + 31) / 32, (O + 31) / 32), dim3(256), 0,
                       static_cast<hipStream_t>(stream), theta, e_w, var_w, ld, I, O,
                       (flags & LBBNN_F_HALF16) ? 2 : ((flags & LBBNN_F_SPLIT16) ? 1 : 0));
    return (int)hipGetLastError();
}
