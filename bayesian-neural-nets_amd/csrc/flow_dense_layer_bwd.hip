// Vector-sized backward of an MNF layer with dense coupling flows (RNVP / MNF type) -- the dense-flow counterpart of
// lbbnn_mnf_flow_planar_backward (see include/lbbnn.h).  Spread over workgroups, three launches per transform:
//   heads  (grid over rows i of the two I x H head matrices): gate / shift recomputed from the kept head input, the
//          two outer-product gradient rows, the direct part of dz, and this workgroup's share of dy (H)
//   chain  (one workgroup): dy summed over workgroups in a fixed order, back through the H x H layers (RNVP) or the
//          tanh (MNF type), their gradients, delta at the input layer's pre-activation
//   input  (grid over columns i of the H x I input matrix): its outer-product gradient and dz_i += m_i W_in[:,i].delta
// Both draws of the z flow (forward draw F, KL draw K) share parameters, so each launch handles both and writes the
// SUM of their gradients once.  Nothing is re-run forward: the forward kept every transform's input and the hidden
// activations (lbbnn_dense_layer_t::save).
#include <cmath>
#include "lbbnn_device.h"
#include "lbbnn_internal.h"

namespace {

using namespace lbbnn;
constexpr int HMAX = LBBNN_MAX_HIDDEN;
constexpr int RW = 16;            // head rows per workgroup (4 waves x 4 rows)
constexpr int CW = 64;            // input-matrix columns per workgroup
constexpr int NT1 = 1024, NWV1 = NT1 / 64;

struct PathB {
    const float* z_in;    // (I) input of this transform on this path (kept by the forward)
    const float* mask;    // (I)
    const float* hs;      // 4 x HMAX: h1 | h2 | h3 | head input
    float* dz;            // (I) running gradient, updated in place
    const float* dz_add;  // (I) or NULL: added to dz on read by the heads kernel (the K1b part of d/dz2)
    int use_ld;           // the path's log-det enters the KL with factor -1 (multiplier -g_kl); 0: unused (forward draw)
};

struct StepArgs {
    lbbnn_dense_transform_t tr;
    lbbnn_dense_grad_t gr;
    PathB p[2];
    const float* g_kl;
    float* dypart;        // [2][nwg][HMAX]
    float* delta0;        // [2][HMAX]
    int npaths, I, nwg;
};

// every launch handles up to LBBNN_MAX_LAYERS layers (blockIdx.z = layer; grids are sized for the widest one)
struct StepBatch { StepArgs l[LBBNN_MAX_LAYERS]; };

__global__ __launch_bounds__(256) void dense_bwd_heads_kernel(const StepBatch ka) {
    __shared__ float dpart[2][4][HMAX];
    const LBBNN_CONST_AS StepArgs& A = kernarg_as<StepBatch>()->l[blockIdx.z];
    if ((int)blockIdx.x >= A.nwg) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, H = A.tr.hidden, I = A.I;
    const bool c0 = lane < H, c1 = lane + 64 < H;
    const float G = A.g_kl ? A.g_kl[0] : 0.f;
    float y0[2], y1[2], dy0[2] = {0.f, 0.f}, dy1[2] = {0.f, 0.f};
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        y0[p] = (p < A.npaths && c0) ? A.p[p].hs[3 * HMAX + lane] : 0.f;
        y1[p] = (p < A.npaths && c1) ? A.p[p].hs[3 * HMAX + lane + 64] : 0.f;
    }
    const bool rnvp = A.tr.kind == LBBNN_FLOW_RNVP;
    // all global reads of this wave's rows are requested before anything is computed: one memory latency, not one per row
    constexpr int RPW = RW / 4;
    const int last = I - 1;
    float ra0[RPW], ra1[RPW], rb0[RPW], rb1[RPW], bav[RPW], bbv[RPW], mv[2][RPW], zv[2][RPW], dv[2][RPW];
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
        const int i = min(blockIdx.x * RW + wv + 4 * j, last);                  // clamped: rows past the end are not stored
        const float* ra = A.tr.w_a + (size_t)i * H;
        const float* rb = A.tr.w_b + (size_t)i * H;
        ra0[j] = c0 ? ra[lane] : 0.f; ra1[j] = c1 ? ra[lane + 64] : 0.f;
        rb0[j] = c0 ? rb[lane] : 0.f; rb1[j] = c1 ? rb[lane + 64] : 0.f;
        bav[j] = A.tr.b_a[i]; bbv[j] = A.tr.b_b[i];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const bool on = p < A.npaths;
            mv[p][j] = on ? A.p[p].mask[i] : 0.f;
            zv[p][j] = on ? A.p[p].z_in[i] : 0.f;
            dv[p][j] = on ? A.p[p].dz[i] + (A.p[p].dz_add ? A.p[p].dz_add[i] : 0.f) : 0.f;
        }
    }
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
        const int i = blockIdx.x * RW + wv + 4 * j;
        if (i > last) break;                                                   // wave-uniform
        const float a0 = ra0[j], a1 = ra1[j], b0 = rb0[j], b1 = rb1[j];
        float ga0 = 0.f, ga1 = 0.f, gb0 = 0.f, gb1 = 0.f, sda = 0.f, sdb = 0.f;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            if (p >= A.npaths) break;
            const float sa = wave_sum(a0 * y0[p] + a1 * y1[p]) + bav[j];
            const float sb = wave_sum(b0 * y0[p] + b1 * y1[p]) + bbv[j];
            const float g = 1.f / (1.f + expf(-sb)), mi = mv[p][j], zi = zv[p][j], dout = dv[p][j];
            const float dld = A.p[p].use_ld ? -G : 0.f;
            float dg, da, dzi;
            if (rnvp) {                                                        // flows2.py:215,219
                dg = dout * ((1.f - mi) * zi - sa) + (dld != 0.f ? dld * (1.f - mi) / g : 0.f);
                da = dout * (1.f - g);
                dzi = dout * ((1.f - mi) * g + mi);
            } else {                                                           // flows2.py:238,241
                dg = dout * (1.f - mi) * (zi - sa) + (dld != 0.f ? dld * (1.f - mi) / g : 0.f);
                da = dout * (1.f - mi) * (1.f - g);
                dzi = dout * (mi + (1.f - mi) * g);
            }
            const float db = dg * g * (1.f - g);
            ga0 += da * y0[p]; ga1 += da * y1[p]; gb0 += db * y0[p]; gb1 += db * y1[p];
            dy0[p] += a0 * da + b0 * db; dy1[p] += a1 * da + b1 * db;
            sda += da; sdb += db;
            if (lane == 0) A.p[p].dz[i] = dzi;
        }
        float* gra = A.gr.w_a + (size_t)i * H;
        float* grb = A.gr.w_b + (size_t)i * H;
        if (c0) { gra[lane] = ga0; grb[lane] = gb0; }
        if (c1) { gra[lane + 64] = ga1; grb[lane + 64] = gb1; }
        if (lane == 0) { A.gr.b_a[i] = sda; A.gr.b_b[i] = sdb; }
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) { dpart[p][wv][lane] = dy0[p]; dpart[p][wv][lane + 64] = dy1[p]; }
    __syncthreads();
    {
        const int p = tid >> 7, k = tid & 127;                                 // 256 threads = 2 paths x HMAX
        if (p < A.npaths)
            A.dypart[((size_t)p * A.nwg + blockIdx.x) * HMAX + k] = dpart[p][0][k] + dpart[p][1][k] + dpart[p][2][k] + dpart[p][3][k];
    }
}

constexpr int NTC = 1024;          // chain kernel: 8 slices x HMAX units

__global__ __launch_bounds__(NTC) void dense_bwd_chain_kernel(const StepBatch ka) {
    __shared__ float d[2][2][HMAX];        // [buffer][path][unit]
    __shared__ float hh[2][HMAX];          // input activations of the layer being differentiated
    __shared__ float part[8][2][HMAX];     // [slice][path][unit]
    const LBBNN_CONST_AS StepArgs& A = kernarg_as<StepBatch>()->l[blockIdx.z];
    const int tid = threadIdx.x, H = A.tr.hidden, np = A.npaths;
    const int k = tid & 127, q = tid >> 7;                                     // unit, slice 0..7
    const bool rnvp = A.tr.kind == LBBNN_FLOW_RNVP;
    // every global read of the chain is requested here, before the first dependent step (one memory latency in total):
    // this slice's rows r = q, q+8, ... of column k of the three middle matrices, and the kept activations
    constexpr int RS = HMAX / 8;
    float wv_[3][RS], hv[4] = {0.f, 0.f, 0.f, 0.f};
    if (rnvp) {
#pragma unroll
        for (int l = 0; l < 3; ++l)
#pragma unroll
            for (int j = 0; j < RS; ++j) { const int r = q + 8 * j; wv_[l][j] = (k < H && r < H) ? A.tr.w_mid[l][r * H + k] : 0.f; }
    }
    if (q < np && k < H) {
#pragma unroll
        for (int l = 0; l < 4; ++l) hv[l] = A.p[q].hs[l * HMAX + k];
    }
    {
        // dy = sum over the heads kernel's workgroups: (path, quarter) per slice, 4 loads in flight per trip, fixed order
        const int p = q & 1, s4 = q >> 1;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        if (p < np && k < H) {
            const float* src = A.dypart + (size_t)p * A.nwg * HMAX + k;
            int w = s4;
            for (; w + 12 < A.nwg; w += 16) {
                const float v0 = src[(size_t)w * HMAX], v1 = src[(size_t)(w + 4) * HMAX], v2 = src[(size_t)(w + 8) * HMAX], v3 = src[(size_t)(w + 12) * HMAX];
                a0 += v0; a1 += v1; a2 += v2; a3 += v3;
            }
            for (; w < A.nwg; w += 4) a0 += src[(size_t)w * HMAX];
        }
        part[q][0][k] = (a0 + a1) + (a2 + a3);                                 // [q = s4*2 + p]
    }
    __syncthreads();
    if (q < 2) d[0][q][k] = (part[q][0][k] + part[2 + q][0][k]) + (part[4 + q][0][k] + part[6 + q][0][k]);
    int cur = 0;
    if (rnvp) {
#pragma unroll
        for (int l = 2; l >= 0; --l) {
            // layer l: out = W_mid[l] h_l + b_mid[l] (h_l = h1, h2, h3: post-activation, same sign as the pre-activation);
            // d[cur] = dL / d(out pre-activation)
            if (q < 2) hh[q][k] = hv[l];
            __syncthreads();
            for (int e = tid; e < H * H; e += NTC) {
                const int r = e / H, c = e - r * H;
                A.gr.w_mid[l][e] = d[cur][0][r] * hh[0][c] + d[cur][1][r] * hh[1][c];
            }
            if (tid < H) A.gr.b_mid[l][tid] = d[cur][0][tid] + d[cur][1][tid];
            {
                // transposed product, coalesced over the column k: slice q sums rows r = q, q+8, ... for both paths
                float s0 = 0.f, s1 = 0.f;
#pragma unroll
                for (int j = 0; j < RS; ++j) {
                    const int r = q + 8 * j;                                   // r < HMAX always; rows >= H hold w = 0 and d = 0
                    s0 += wv_[l][j] * d[cur][0][r]; s1 += wv_[l][j] * d[cur][1][r];
                }
                part[q][0][k] = s0; part[q][1][k] = s1;
            }
            __syncthreads();
            if (q < 2) {
                float s = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) s += part[j][q][k];
                d[cur ^ 1][q][k] = (k < H) ? s * (hh[q][k] >= 0.f ? 1.f : 0.1f) : 0.f;
            }
            __syncthreads();
            cur ^= 1;
        }
    } else {
        __syncthreads();
        if (q < 2) {
            const float th = hv[3];                                                // tanh(f(m z)) (flows2.py:235)
            d[1][q][k] = d[0][q][k] * (1.f - th * th);
        }
        __syncthreads();
        cur = 1;
    }
    if (q < 2) A.delta0[q * HMAX + k] = d[cur][q][k];
    if (tid < H) A.gr.b_in[tid] = d[cur][0][tid] + d[cur][1][tid];
}

__global__ __launch_bounds__(256) void dense_bwd_input_kernel(const StepBatch ka) {
    __shared__ float dl[2][HMAX];
    __shared__ float accs[2][4][CW];
    const LBBNN_CONST_AS StepArgs& A = kernarg_as<StepBatch>()->l[blockIdx.z];
    if ((int)blockIdx.x * CW >= A.I) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, H = A.tr.hidden, I = A.I;
    dl[tid >> 7][tid & 127] = A.delta0[tid];
    __syncthreads();
    const int i = blockIdx.x * CW + lane;
    const bool valid = i < I;
    float mz[2], acc[2] = {0.f, 0.f};
#pragma unroll
    for (int p = 0; p < 2; ++p) mz[p] = (p < A.npaths && valid) ? A.p[p].mask[i] * A.p[p].z_in[i] : 0.f;
    if (valid) {
#pragma unroll 4
        for (int h = wv; h < H; h += 4) {
            const float w = A.tr.w_in[(size_t)h * I + i];
            const float d0 = dl[0][h], d1 = dl[1][h];
            acc[0] += w * d0; acc[1] += w * d1;
            A.gr.w_in[(size_t)h * I + i] = d0 * mz[0] + d1 * mz[1];
        }
    }
    accs[0][wv][lane] = acc[0]; accs[1][wv][lane] = acc[1];
    __syncthreads();
    if (wv < A.npaths && valid) {                                              // wave p finishes path p
        const float s = accs[wv][0][lane] + accs[wv][1][lane] + accs[wv][2][lane] + accs[wv][3][lane];
        A.p[wv].dz[i] += A.p[wv].mask[i] * s;
    }
}

struct HeadArgs {
    const float *bias_mu, *bias_rho, *g_sum, *gv_sum, *g_kl, *r0_b1, *r0_b2, *aux, *zr_last, *dz_fwd;
    float *d_bias_mu, *d_bias_rho, *d_r0_b1, *d_r0_b2, *DK, *DF;
    lbbnn_priors_t priors;
    int O, I;
};

__device__ __forceinline__ double bsum1(double a, double* scratch) {
    a = wave_sum(a);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) scratch[w] = a;
    __syncthreads();
    double s = 0;
#pragma unroll
    for (int i = 0; i < NWV1; ++i) s += scratch[i];
    return s;
}

// bias terms (LBBNN-GP-MF-MNF.py:197-198, 234-236), log_rb gradients (:224-233), and the two running gradients' start:
// DK = -g_kl * dlog_rb/dz_b (only the last element is non-zero), DF = dz_fwd
struct HeadBatch { HeadArgs l[LBBNN_MAX_LAYERS]; };

__global__ __launch_bounds__(NT1) void dense_bwd_head_kernel(const HeadBatch hb) {
    __shared__ double scratch[NWV1];
    const LBBNN_CONST_AS HeadArgs& a = kernarg_as<HeadBatch>()->l[blockIdx.x];
    const int tid = threadIdx.x, I = a.I, O = a.O;
    const bool has_kl = a.g_kl != nullptr;
    const float G = has_kl ? a.g_kl[0] : 0.f;
    for (int o = tid; o < O; o += NT1) {
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
    for (int i = tid; i < I; i += NT1) a.DF[i] = a.dz_fwd ? a.dz_fwd[i] : 0.f;
    if (!has_kl) {
        for (int i = tid; i < I; i += NT1) { a.d_r0_b1[i] = 0.f; a.d_r0_b2[i] = 0.f; a.DK[i] = 0.f; }
        return;
    }
    const float zb = a.zr_last[0], m = a.aux[0];
    double Szb = 0;
    for (int i = tid; i < I; i += NT1) {
        const float e = expf(-a.r0_b2[i] * m), dlt = zb - a.r0_b1[i] * m;
        a.d_r0_b1[i] = -G * dlt * m * e;
        a.d_r0_b2[i] = -G * (-0.5f * m + 0.5f * dlt * dlt * m * e);
        Szb += (double)(dlt * e);
        if (i != I - 1) a.DK[i] = 0.f;
    }
    Szb = bsum1(Szb, scratch);
    if (tid == 0) a.DK[I - 1] = G * (float)Szb;
}

struct TailArgs {
    const float *q0_log_var, *eps_fwd, *eps_kl, *DK, *DF, *dk_add, *g_kl;
    float *d_q0_mean, *d_q0_log_var;
    const uint64_t* rng;
    uint32_t layer;
    int I;
};

// q0 (LBBNN-GP-MF-MNF.py:183-185, 201-205): dlog_q0/dlog_var = -1/2 exactly, dlog_q0/dmean = 0
struct TailBatch { TailArgs l[LBBNN_MAX_LAYERS]; };

__global__ __launch_bounds__(256) void dense_bwd_tail_kernel(const TailBatch tb) {
    const LBBNN_CONST_AS TailArgs& a = kernarg_as<TailBatch>()->l[blockIdx.z];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.I) return;
    const bool has_kl = a.g_kl != nullptr;
    const float G = has_kl ? a.g_kl[0] : 0.f;
    float ef, ek = 0.f;
    if (a.eps_fwd) { ef = a.eps_fwd[i]; if (has_kl) ek = a.eps_kl[i]; }
    else {
        float n[4];
        philox_normal4(a.rng[0], a.rng[1], LBBNN_STREAM_EPS_Z * 64u + a.layer, (uint64_t)(i >> 2), 0u, n); ef = n[i & 3];
        if (has_kl) { philox_normal4(a.rng[0], a.rng[1], LBBNN_STREAM_EPS_Z2 * 64u + a.layer, (uint64_t)(i >> 2), 0u, n); ek = n[i & 3]; }
    }
    const float sd = expf(0.5f * a.q0_log_var[i]);
    const float dk = a.DK[i] + (a.dk_add ? a.dk_add[i] : 0.f), df = a.DF[i];
    a.d_q0_mean[i] = dk + df;
    a.d_q0_log_var[i] = 0.5f * sd * (dk * ek + df * ef) - 0.5f * G;
}

int check_pair(const lbbnn_dense_transform_t& t, const lbbnn_dense_grad_t& g, bool need_fwd, bool need_kl) {
    if (t.kind != LBBNN_FLOW_RNVP && t.kind != LBBNN_FLOW_MNF) return LBBNN_E_FLAGS;
    if (t.hidden <= 0 || t.hidden > HMAX) return LBBNN_E_SHAPE;
    if (!t.w_in || !t.b_in || !t.w_a || !t.b_a || !t.w_b || !t.b_b) return LBBNN_E_NULL;
    if (!g.w_in || !g.b_in || !g.w_a || !g.b_a || !g.w_b || !g.b_b) return LBBNN_E_NULL;
    if (t.kind == LBBNN_FLOW_RNVP)
        for (int l = 0; l < 3; ++l) if (!t.w_mid[l] || !t.b_mid[l] || !g.w_mid[l] || !g.b_mid[l]) return LBBNN_E_NULL;
    if ((need_fwd && !t.mask_fwd) || (need_kl && !t.mask_kl)) return LBBNN_E_NULL;
    return 0;
}

void zero_grads(const lbbnn_dense_transform_t& t, const lbbnn_dense_grad_t& g, int I, hipStream_t s) {
    const size_t H = (size_t)t.hidden, f = sizeof(float);
    (void)hipMemsetAsync(g.w_in, 0, H * I * f, s); (void)hipMemsetAsync(g.b_in, 0, H * f, s);
    (void)hipMemsetAsync(g.w_a, 0, H * I * f, s);  (void)hipMemsetAsync(g.b_a, 0, (size_t)I * f, s);
    (void)hipMemsetAsync(g.w_b, 0, H * I * f, s);  (void)hipMemsetAsync(g.b_b, 0, (size_t)I * f, s);
    if (t.kind == LBBNN_FLOW_RNVP)
        for (int l = 0; l < 3; ++l) { (void)hipMemsetAsync(g.w_mid[l], 0, H * H * f, s); (void)hipMemsetAsync(g.b_mid[l], 0, H * f, s); }
}

}  // namespace

// work (floats): DK[I] | DF[I] | dypart[2][nwg][HMAX] | delta0[2][HMAX]
extern "C" int64_t lbbnn_mnf_flow_dense_backward_workspace(int I) {
    if (I <= 0) return 0;
    const int64_t nwg = (I + RW - 1) / RW;
    return 2 * (int64_t)I + 2 * nwg * HMAX + 2 * HMAX;
}

// n layers in the same launches (blockIdx.z = layer); they must agree on Tz, Tr and on having a KL branch
static int dense_backward_impl(const lbbnn_dense_bwd_args_t* L, int n, void* stream) {
    if (!L) return LBBNN_E_NULL;
    if (n <= 0 || n > LBBNN_MAX_LAYERS) return LBBNN_E_SHAPE;
    const int Tz = L[0].Tz, Tr = L[0].Tr;
    const bool has_kl = L[0].g_kl != nullptr;
    int maxI = 0;
    for (int k = 0; k < n; ++k) {
        const lbbnn_dense_bwd_args_t& a = L[k];
        if (!a.eps_fwd && !a.rng) return LBBNN_E_NOISE;
        if (!a.q0_mean || !a.q0_log_var || !a.bias_mu || !a.bias_rho || !a.g_sum || !a.work || !a.save ||
            !a.d_q0_mean || !a.d_q0_log_var || !a.d_r0_b1 || !a.d_r0_b2 || !a.d_bias_mu || !a.d_bias_rho) return LBBNN_E_NULL;
        if ((a.g_kl != nullptr) != has_kl || a.Tz != Tz || a.Tr != Tr) return LBBNN_E_SHAPE;
        if (has_kl && ((a.eps_fwd && !a.eps_kl) || !a.r0_b1 || !a.r0_b2 || !a.aux)) return LBBNN_E_NULL;
        if (a.O <= 0 || a.I <= 0) return LBBNN_E_SHAPE;
        if (Tz < 0 || Tz > LBBNN_MAX_FLOW_T || Tr < 0 || Tr > LBBNN_MAX_FLOW_T) return LBBNN_E_SHAPE;
        if ((Tz && (!a.zt || !a.d_zt)) || (Tr && (!a.rt || !a.d_rt))) return LBBNN_E_NULL;
        for (int t = 0; t < Tz; ++t) if (const int rc = check_pair(a.zt[t], a.d_zt[t], true, has_kl)) return rc;
        for (int t = 0; t < Tr; ++t) if (const int rc = check_pair(a.rt[t], a.d_rt[t], false, has_kl)) return rc;
        maxI = a.I > maxI ? a.I : maxI;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    struct Bufs { float *DK, *DF, *dypart, *delta0; const float *ZF, *ZK, *ZR, *HS; int nwg; } B[LBBNN_MAX_LAYERS];
    for (int k = 0; k < n; ++k) {
        const lbbnn_dense_bwd_args_t& a = L[k];
        Bufs& b = B[k];
        b.nwg = (a.I + RW - 1) / RW;
        b.DK = a.work; b.DF = b.DK + a.I; b.dypart = b.DF + a.I; b.delta0 = b.dypart + (size_t)2 * b.nwg * HMAX;
        // the forward's kept intermediates (lbbnn_flow_dense_save_size)
        b.ZF = a.save; b.ZK = b.ZF + (size_t)(Tz + 1) * a.I; b.ZR = b.ZK + (size_t)(Tz + 1) * a.I; b.HS = b.ZR + (size_t)Tr * a.I;
    }
    auto hs_at = [&](const Bufs& b, int t, int path) { return b.HS + ((size_t)t * 2 + path) * 4 * HMAX; };
    auto r_in = [&](const Bufs& b, int I, int t) { return t == 0 ? b.ZK + (size_t)Tz * I : b.ZR + (size_t)(t - 1) * I; };
    const int gwg = (maxI + RW - 1) / RW;

    HeadBatch hb{};
    for (int k = 0; k < n; ++k) {
        const lbbnn_dense_bwd_args_t& a = L[k];
        HeadArgs& h = hb.l[k];
        h.bias_mu = a.bias_mu; h.bias_rho = a.bias_rho; h.g_sum = a.g_sum; h.gv_sum = a.gv_sum; h.g_kl = a.g_kl;
        h.r0_b1 = a.r0_b1; h.r0_b2 = a.r0_b2; h.aux = a.aux; h.dz_fwd = a.dz_fwd;
        h.zr_last = r_in(B[k], a.I, Tr) + (a.I - 1);
        h.d_bias_mu = a.d_bias_mu; h.d_bias_rho = a.d_bias_rho; h.d_r0_b1 = a.d_r0_b1; h.d_r0_b2 = a.d_r0_b2;
        h.DK = B[k].DK; h.DF = B[k].DF; h.priors = a.priors; h.O = a.O; h.I = a.I;
    }
    hipLaunchKernelGGL(dense_bwd_head_kernel, dim3(n), dim3(NT1), 0, s, hb);

    auto step = [&](const StepBatch& sb) {
        hipLaunchKernelGGL(dense_bwd_heads_kernel, dim3(gwg, 1, n), dim3(256), 0, s, sb);
        hipLaunchKernelGGL(dense_bwd_chain_kernel, dim3(1, 1, n), dim3(NTC), 0, s, sb);
        hipLaunchKernelGGL(dense_bwd_input_kernel, dim3((maxI + CW - 1) / CW, 1, n), dim3(256), 0, s, sb);
    };
    if (has_kl) {
        for (int t = Tr - 1; t >= 0; --t) {                                    // r flow on z2; log_det_r enters the KL as -log_det_r
            StepBatch sb{};
            for (int k = 0; k < n; ++k) {
                const lbbnn_dense_bwd_args_t& a = L[k];
                StepArgs& sa = sb.l[k];
                sa.tr = a.rt[t]; sa.gr = a.d_rt[t];
                sa.p[0] = PathB{r_in(B[k], a.I, t), a.rt[t].mask_kl, hs_at(B[k], Tz + t, 1), B[k].DK, nullptr, 1};
                sa.g_kl = a.g_kl; sa.dypart = B[k].dypart; sa.delta0 = B[k].delta0; sa.npaths = 1; sa.I = a.I; sa.nwg = B[k].nwg;
            }
            step(sb);
        }
    } else {
        for (int k = 0; k < n; ++k) for (int t = 0; t < Tr; ++t) zero_grads(L[k].rt[t], L[k].d_rt[t], L[k].I, s);
    }
    for (int t = Tz - 1; t >= 0; --t) {                                        // z flow, both draws; log_q = -log_det_q + log_q0
        StepBatch sb{};
        for (int k = 0; k < n; ++k) {
            const lbbnn_dense_bwd_args_t& a = L[k];
            StepArgs& sa = sb.l[k];
            sa.tr = a.zt[t]; sa.gr = a.d_zt[t];
            sa.p[0] = PathB{B[k].ZF + (size_t)t * a.I, a.zt[t].mask_fwd, hs_at(B[k], t, 0), B[k].DF, nullptr, 0};
            sa.p[1] = PathB{B[k].ZK + (size_t)t * a.I, a.zt[t].mask_kl, hs_at(B[k], t, 1), B[k].DK, (t == Tz - 1) ? a.dz_kl : nullptr, 1};
            sa.g_kl = a.g_kl; sa.dypart = B[k].dypart; sa.delta0 = B[k].delta0; sa.npaths = has_kl ? 2 : 1; sa.I = a.I; sa.nwg = B[k].nwg;
        }
        step(sb);
    }
    TailBatch tb{};
    for (int k = 0; k < n; ++k) {
        const lbbnn_dense_bwd_args_t& a = L[k];
        TailArgs& ta = tb.l[k];
        ta.q0_log_var = a.q0_log_var; ta.eps_fwd = a.eps_fwd; ta.eps_kl = a.eps_kl; ta.DK = B[k].DK; ta.DF = B[k].DF;
        ta.dk_add = (Tz == 0 && has_kl) ? a.dz_kl : nullptr; ta.g_kl = a.g_kl;
        ta.d_q0_mean = a.d_q0_mean; ta.d_q0_log_var = a.d_q0_log_var; ta.rng = a.rng; ta.layer = a.layer_id & 63u; ta.I = a.I;
    }
    hipLaunchKernelGGL(dense_bwd_tail_kernel, dim3((maxI + 255) / 256, 1, n), dim3(256), 0, s, tb);
    return (int)hipGetLastError();
}

extern "C" int lbbnn_mnf_flow_dense_backward(const lbbnn_dense_bwd_args_t* args, void* stream) {
    return dense_backward_impl(args, 1, stream);
}

extern "C" int lbbnn_mnf_flow_dense_backward_batch(const lbbnn_dense_bwd_args_t* args, int n, void* stream) {
    return dense_backward_impl(args, n, stream);
}
