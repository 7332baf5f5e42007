// lbbnn_output_grad: the (B,O) elementwise head of the backward pass, fused (see include/lbbnn.h).  HBM-bound:
// reads g_out, out, std (12 B/elt; eps is re-created from Philox), writes G_m, G_v and their transposes (16 B/elt).
// A 256-thread workgroup owns a 64(b) x 64(o) tile: float4 row-major reads/writes, the transposes through two
// padded LDS tiles, and the tile's column sums as per-tile partials that a second launch adds in a fixed order.
#include "lbbnn_device.h"
#include "lbbnn_internal.h"
#include "reduce_partials.h"

namespace {

using namespace lbbnn;
constexpr int TS = 64;

__global__ __launch_bounds__(256) void output_grad_kernel(const lbbnn_outgrad_args_t a, int nbt) {
    __shared__ float tm[TS][TS + 1], tv[TS][TS + 1];
    const int b0 = blockIdx.y * TS, o0 = blockIdx.x * TS, tid = threadIdx.x;
    const bool stoch = a.std != nullptr;
    uint64_t seed = 0, offs = 0;
    if (stoch && !a.eps) { seed = a.rng[0]; offs = a.rng[1]; }
    const bool vec = ((a.O & 3) == 0) && ((a.ldg & 3) == 0) && ((a.ldo & 3) == 0) &&
                     ((reinterpret_cast<uintptr_t>(a.g_out) & 15u) == 0) && (!a.out || (reinterpret_cast<uintptr_t>(a.out) & 15u) == 0) &&
                     (!stoch || (reinterpret_cast<uintptr_t>(a.std) & 15u) == 0) && (!a.eps || (reinterpret_cast<uintptr_t>(a.eps) & 15u) == 0) &&
                     (!a.gm || (reinterpret_cast<uintptr_t>(a.gm) & 15u) == 0) && (!stoch || !a.gv || (reinterpret_cast<uintptr_t>(a.gv) & 15u) == 0);
    // ---- row-major pass: thread -> (row r = idx / 16, 4 consecutive o)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int idx = tid + 256 * k, r = idx >> 4, c = (idx & 15) * 4;
        const int b = b0 + r, o = o0 + c;
        float gm[4] = {0.f, 0.f, 0.f, 0.f}, gv[4] = {0.f, 0.f, 0.f, 0.f};
        if (b < a.B && o < a.O) {
            float g[4], y[4] = {1.f, 1.f, 1.f, 1.f}, sd[4] = {1.f, 1.f, 1.f, 1.f}, e[4] = {0.f, 0.f, 0.f, 0.f};
            const size_t ig = (size_t)b * a.ldg + o, io = (size_t)b * a.ldo + o;
            if (vec) {
                const float4 t = *reinterpret_cast<const float4*>(a.g_out + ig);
                g[0] = t.x; g[1] = t.y; g[2] = t.z; g[3] = t.w;
                if (a.relu) { const float4 u = *reinterpret_cast<const float4*>(a.out + io); y[0] = u.x; y[1] = u.y; y[2] = u.z; y[3] = u.w; }
                if (stoch) { const float4 u = *reinterpret_cast<const float4*>(a.std + io); sd[0] = u.x; sd[1] = u.y; sd[2] = u.z; sd[3] = u.w; }
                if (stoch && a.eps) { const float4 u = *reinterpret_cast<const float4*>(a.eps + (size_t)b * a.O + o); e[0] = u.x; e[1] = u.y; e[2] = u.z; e[3] = u.w; }
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const bool in = o + q < a.O;
                    g[q] = in ? a.g_out[ig + q] : 0.f;
                    if (a.relu) y[q] = in ? a.out[io + q] : 0.f;
                    if (stoch) sd[q] = in ? a.std[io + q] : 1.f;
                    if (stoch && a.eps) e[q] = in ? a.eps[(size_t)b * a.O + o + q] : 0.f;
                }
            }
            if (stoch && !a.eps) philox_normal4(seed, offs, a.rng_stream, (uint64_t)(a.row_offset + b), (uint32_t)(o >> 2), e);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                gm[q] = (a.relu && !(y[q] > 0.f)) ? 0.f : g[q];
                if (stoch) gv[q] = gm[q] * e[q] / (2.f * sd[q]);
                if (o + q >= a.O) { gm[q] = 0.f; gv[q] = 0.f; }
                else if (stoch && a.gv_scale) gv[q] *= a.gv_scale[o + q];
            }
            // (gm / gv NULL: a first layer, whose input needs no gradient, uses the transposes and the sums only -- a quarter of
            // this pass's traffic is not written)
            const size_t id = (size_t)b * a.O + o;
            if (a.gm) {
                if (vec) {
                    *reinterpret_cast<float4*>(a.gm + id) = make_float4(gm[0], gm[1], gm[2], gm[3]);
                    if (stoch && a.gv) *reinterpret_cast<float4*>(a.gv + id) = make_float4(gv[0], gv[1], gv[2], gv[3]);
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) if (o + q < a.O) { a.gm[id + q] = gm[q]; if (stoch && a.gv) a.gv[id + q] = gv[q]; }
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) { tm[r][c + q] = gm[q]; tv[r][c + q] = gv[q]; }
    }
    __syncthreads();
    // ---- transposed pass: thread -> (column oc = idx / 64, row br = idx % 64): 256-B contiguous rows of G^T
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int idx = tid + 256 * k, oc = idx >> 6, br = idx & 63;
        const int o = o0 + oc, b = b0 + br;
        if (o < a.O && b < a.B) {
            a.gmT[(size_t)o * a.B + b] = tm[br][oc];
            if (stoch) a.gvT[(size_t)o * a.B + b] = tv[br][oc];
        }
    }
    // ---- column sums of the tile (rows past B hold zeros)
    if (tid < 2 * TS) {
        const int oc = tid & 63, which = tid >> 6;
        if (which == 0 || stoch) {
            float s = 0.f;
#pragma unroll 8
            for (int r = 0; r < TS; ++r) s += which ? tv[r][oc] : tm[r][oc];
            const int o = o0 + oc;
            if (o < a.O) a.work[((size_t)which * nbt + blockIdx.y) * a.O + o] = s;
        }
    }
}

// g_sum[o] = sum over row tiles of the partials (reduce_partials.h: the body shared with lbbnn_reduce_partials_batch)
__global__ __launch_bounds__(1024) void output_grad_sum_kernel(const float* __restrict__ work, int nbt, int O, float* g_sum, float* gv_sum) {
    __shared__ float part[3][16][64];
    float* const out[3] = {g_sum, gv_sum, nullptr};
    reduce_partials_body(work, O, (long long)nbt * O, nbt, O, gv_sum ? 2 : 1, out, blockIdx.x, part);
}

// The second level of lbbnn_output_grad's column sums AND what an LRT layer does with them (lbbnn_bias_backward) in one
// launch: same partial order as reduce_partials_body (wave w adds every 16th block, then 16 partials in a fixed order), so the
// sums -- and the bias gradients -- are bitwise those of the two stand-alone launches.  Wave 0 ends with Sum_b G_m of its 64
// columns (-> d_bias_mu), wave 1 with Sum_b G_v (-> d_bias_rho; zero for a posterior-mean forward).
__global__ __launch_bounds__(1024) void bias_backward_partials_kernel(const float* __restrict__ work, int nbt, int O, int has_gv,
                                                                      const float* bias_mu, const float* bias_rho,
                                                                      const float* g_kl, lbbnn_priors_t priors,
                                                                      float* d_bias_mu, float* d_bias_rho) {
    __shared__ float part[2][16][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + lane;
    float s0 = 0.f, s1 = 0.f;
    if (i < O)
        for (int b = w; b < nbt; b += 16) {
            const float* p = work + (size_t)b * O + i;
            s0 += p[0];
            if (has_gv) s1 += p[(size_t)nbt * O];
        }
    part[0][w][lane] = s0; part[1][w][lane] = s1;
    __syncthreads();
    if (w < 2 && i < O) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += part[w][k][lane];
        const float G = g_kl ? g_kl[0] : 0.f, inv = 1.f / (priors.bias_sigma_prior * priors.bias_sigma_prior);
        if (w == 0) {
            float gm = s;
            if (g_kl) gm += G * (bias_mu[i] - priors.bias_mu_prior) * inv;
            d_bias_mu[i] = gm;
        } else {
            const float er = expf(bias_rho[i]);
            const float sb = log1pf(er), dsig = er / (1.f + er);
            float gs = has_gv ? s * 2.f * sb : 0.f;
            if (g_kl) gs += G * (sb * inv - 1.f / sb);
            d_bias_rho[i] = gs * dsig;
        }
    }
}

struct ReduceBatch { lbbnn_reduce_job_t j[LBBNN_MAX_REDUCE_JOBS]; };
__global__ __launch_bounds__(1024) void reduce_partials_batch_kernel(const ReduceBatch bt) {
    __shared__ float part[3][16][64];
    const LBBNN_CONST_AS lbbnn_reduce_job_t& j = kernarg_as<ReduceBatch>()->j[blockIdx.y];
    if ((int)blockIdx.x * 64 >= j.ncols) return;                      // (uniform per workgroup)
    float* const out[3] = {j.out[0], j.out[1], j.out[2]};
    reduce_partials_body(j.work, j.block_stride, j.q_stride, j.nblk, j.ncols, j.nq, out, blockIdx.x, part);
}

// dX = G_m.W_m + 2 x (.) (G_v.W_v):  gx += 2 * x * gxv   (one pass instead of three elementwise launches)
__global__ __launch_bounds__(256) void dx_combine_kernel(float* __restrict__ gx, const float* __restrict__ gxv,
                                                         const float* __restrict__ x, int ldx, int B, int I) {
    const int b = blockIdx.y;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < I; i += gridDim.x * 256)
        gx[(size_t)b * I + i] += 2.f * x[(size_t)b * ldx + i] * gxv[(size_t)b * I + i];
}

// dX of a <= 16-class head (LBBNN-GP-MF-MNF.py:197-198 differentiated; K = classes):
//   dX[b][i] = sum_c G_m[b][c] W_m^T[i][c] + 2 x[b][i] sum_c G_v[b][c] W_v^T[i][c]
// The GEMM kernels are built for K in the hundreds: K = 10 cost two 15.4 us launches of the generic fp32 kernel (a full tile
// machinery around ten multiply-adds per output).  Here a workgroup owns 16 rows x 256 columns: thread = one column i, its
// 2 x C operand values in registers, the 16 rows' gradients broadcast from LDS; x read and dX written coalesced, once.
constexpr int HD_ROWS = 16, HD_COLS = 256, HD_MAXC = 16;

__global__ __launch_bounds__(HD_COLS) void head_dx_kernel(const float* __restrict__ gm, const float* __restrict__ gv, int ldg,
                                                          const float* __restrict__ wmT, const float* __restrict__ wvT, int ldw,
                                                          const float* __restrict__ x, int ldx, float* __restrict__ out, int ldo,
                                                          int B, int C, int I) {
    __shared__ float sgm[HD_ROWS][HD_MAXC], sgv[HD_ROWS][HD_MAXC];
    const int b0 = blockIdx.y * HD_ROWS, i = blockIdx.x * HD_COLS + threadIdx.x;
    for (int t = threadIdx.x; t < HD_ROWS * HD_MAXC; t += HD_COLS) {
        const int r = t / HD_MAXC, c = t % HD_MAXC;
        const bool in = b0 + r < B && c < C;
        sgm[r][c] = in ? gm[(size_t)(b0 + r) * ldg + c] : 0.f;
        sgv[r][c] = (in && gv) ? gv[(size_t)(b0 + r) * ldg + c] : 0.f;
    }
    float wm[HD_MAXC], wv[HD_MAXC];
#pragma unroll
    for (int c = 0; c < HD_MAXC; ++c) {
        const bool in = i < I && c < C;
        wm[c] = in ? wmT[(size_t)i * ldw + c] : 0.f;
        wv[c] = (in && wvT) ? wvT[(size_t)i * ldw + c] : 0.f;
    }
    __syncthreads();
    if (i >= I) return;
#pragma unroll 4
    for (int r = 0; r < HD_ROWS; ++r) {
        const int b = b0 + r;
        if (b >= B) break;
        float am = 0.f, av = 0.f;
#pragma unroll
        for (int c = 0; c < HD_MAXC; ++c) { am += sgm[r][c] * wm[c]; av += sgv[r][c] * wv[c]; }
        out[(size_t)b * ldo + i] = gv ? am + 2.f * x[(size_t)b * ldx + i] * av : am;
    }
}

// ... for the class counts this is built for (the reference's head has 10 classes: LBBNN-GP-MF-MNF.py:247): the gradient row
// of a batch row is the same for every thread, so it is read with SCALAR loads (uniform address, read-only) and enters the
// multiply-adds as scalar operands -- no LDS stage, no broadcast reads (the generic kernel above spends 32 ds_reads per
// output on them and pads the class loop to 16), C x 2 multiply-adds per output.
template <int C>
__global__ __launch_bounds__(HD_COLS) void head_dx_c_kernel(const float* __restrict__ gm, const float* __restrict__ gv, int ldg,
                                                            const float* __restrict__ wmT, const float* __restrict__ wvT, int ldw,
                                                            const float* __restrict__ x, int ldx, float* __restrict__ out, int ldo,
                                                            int B, int I) {
    const int b0 = blockIdx.y * HD_ROWS, i = blockIdx.x * HD_COLS + threadIdx.x;
    if (i >= I) return;
    float wm[C], wv[C];
#pragma unroll
    for (int c = 0; c < C; ++c) { wm[c] = wmT[(size_t)i * ldw + c]; wv[c] = wvT[(size_t)i * ldw + c]; }
    const int nr = min(HD_ROWS, B - b0);
#pragma unroll 4
    for (int r = 0; r < nr; ++r) {
        const float* __restrict__ pm = gm + (size_t)(b0 + r) * ldg;
        const float* __restrict__ pv = gv + (size_t)(b0 + r) * ldg;
        const float xv = x[(size_t)(b0 + r) * ldx + i];
        float am = 0.f, av = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) { am += pm[c] * wm[c]; av += pv[c] * wv[c]; }
        out[(size_t)(b0 + r) * ldo + i] = am + 2.f * xv * av;
    }
}

// Weight gradients of a <= 16-class head as split-K slabs, without a transposed copy of x:
//   dWm[s][c][i] = sum_{b in range s} gm[b][c] x[b][i],   dWv[s][c][i] = sum_{b in range s} gv[b][c] x[b][i]^2
// (d/dW of x.e_w^T and x^2.var_w^T, LBBNN-GP-MF-LRT.py:172-173).  The generic route builds x^T | (x^2)^T as GEMM operands
// (one pass over x: 13 us for the 4096 x 1200 activation) and runs two K = 4096 launches of the small-tile GEMM whose 16-row
// tiles are 37 % padding at 10 classes (2 x 9.6 us).  Here a thread owns one column i and a quarter of a row range: x is read
// once, row-major, coalesced; the gradient rows are wave-uniform (scalar loads, scalar operands); 2 C accumulators per
// thread; the eight row lanes (waves) of a workgroup are added through LDS in a fixed order.  S slabs for
// lbbnn_weight_pass_backward to add (fixed order there too): deterministic.  Eight rows' loads in flight per thread and 2.4
// waves per SIMD: the first version (four row lanes, four rows in flight, 1.2 waves per SIMD) was a chain of exposed load
// latencies -- 27.9 us for 19.7 MB.
// (Measured and dropped: the head's dX formed in the same pass -- both kernels want the batch row's gradients as scalars and
// x[b][i] once.  34.2 us fused against 14.7 (lbbnn_head_dx) + 14.8 here: 40 accumulator / operand registers per thread and a
// 256-B store per wave and row inside the loop cost more than the second read of the 19.7 MB activation.)
// (Measured and dropped too: the same slabs on v_mfma_f32_16x16x4_f32 -- classes as M, four batch rows as K, a lane's float4
// of x as the B operands of four column tiles, eight waves per workgroup adding their tiles pairwise through LDS: correct to
// 2e-6, 17.3 us against 14.5 here.)
constexpr int HW_COLS = 64, HW_LANES = 8;

template <int C>
__global__ __launch_bounds__(HW_COLS * HW_LANES) void head_dw_kernel(const float* __restrict__ gm, const float* __restrict__ gv,
                                                                     int ldg, const float* __restrict__ x, int ldx,
                                                                     float* __restrict__ dWm, float* __restrict__ dWv,
                                                                     int B, int I, int rows_per_slab) {
    __shared__ float red[HW_LANES][2 * C][HW_COLS];
    const int cg = threadIdx.x & (HW_COLS - 1);
    const int rl = __builtin_amdgcn_readfirstlane(threadIdx.x / HW_COLS);              // the wave's index: uniform, and the compiler must know (scalar loads of the gradient rows)
    const int i = blockIdx.x * HW_COLS + cg, s = blockIdx.y;
    const int r0 = s * rows_per_slab, r1 = min(r0 + rows_per_slab, B);
    float am[C], av[C];
#pragma unroll
    for (int c = 0; c < C; ++c) { am[c] = 0.f; av[c] = 0.f; }
    if (i < I) {
#pragma unroll 4
        for (int b = r0 + rl; b < r1; b += HW_LANES) {
            const float* __restrict__ pm = gm + (size_t)b * ldg;
            const float xv = x[(size_t)b * ldx + i], x2 = xv * xv;
#pragma unroll
            for (int c = 0; c < C; ++c) am[c] += pm[c] * xv;
            if (gv) {
                const float* __restrict__ pv = gv + (size_t)b * ldg;
#pragma unroll
                for (int c = 0; c < C; ++c) av[c] += pv[c] * x2;
            }
        }
    }
#pragma unroll
    for (int c = 0; c < C; ++c) { red[rl][c][cg] = am[c]; red[rl][C + c][cg] = av[c]; }
    __syncthreads();
    // thread (cg, rl) finishes the values q = rl, rl + 8, ... of column cg: row lanes 0..7 in order
    if (i < I)
        for (int q = rl; q < 2 * C; q += HW_LANES) {
            float v = red[0][q][cg];
#pragma unroll
            for (int l = 1; l < HW_LANES; ++l) v += red[l][q][cg];
            if (q < C) dWm[((size_t)s * C + q) * I + i] = v;
            else if (dWv) dWv[((size_t)s * C + (q - C)) * I + i] = v;
        }
}

}  // namespace

extern "C" int lbbnn_head_dw(const float* gm, const float* gv, int ldg, const float* x, int ldx, float* dWm, float* dWv,
                             int B, int C, int I, int nslabs, void* stream) {
    if (!gm || !x || !dWm) return LBBNN_E_NULL;
    if ((gv == nullptr) != (dWv == nullptr)) return LBBNN_E_NULL;
    if (B <= 0 || C <= 0 || C > HD_MAXC || I <= 0 || ldg < C || ldx < I || nslabs <= 0 || nslabs > B) return LBBNN_E_SHAPE;
    const int rows = (B + nslabs - 1) / nslabs;
    const dim3 grid((I + HW_COLS - 1) / HW_COLS, nslabs), block(HW_COLS * HW_LANES);
    hipStream_t s = static_cast<hipStream_t>(stream);
#define LBBNN_HEAD_DW(CC) case CC: hipLaunchKernelGGL(head_dw_kernel<CC>, grid, block, 0, s, gm, gv, ldg, x, ldx, dWm, dWv, B, I, rows); break;
    switch (C) {
        LBBNN_HEAD_DW(1) LBBNN_HEAD_DW(2) LBBNN_HEAD_DW(3) LBBNN_HEAD_DW(4) LBBNN_HEAD_DW(5) LBBNN_HEAD_DW(6) LBBNN_HEAD_DW(7) LBBNN_HEAD_DW(8)
        LBBNN_HEAD_DW(9) LBBNN_HEAD_DW(10) LBBNN_HEAD_DW(11) LBBNN_HEAD_DW(12) LBBNN_HEAD_DW(13) LBBNN_HEAD_DW(14) LBBNN_HEAD_DW(15) LBBNN_HEAD_DW(16)
    }
#undef LBBNN_HEAD_DW
    return (int)hipGetLastError();
}

extern "C" int lbbnn_head_dx(const float* gm, const float* gv, int ldg, const float* wmT, const float* wvT, int ldw,
                             const float* x, int ldx, float* out, int ldo, int B, int C, int I, void* stream) {
    if (!gm || !wmT || !out) return LBBNN_E_NULL;
    if ((gv == nullptr) != (wvT == nullptr) || (gv && !x)) return LBBNN_E_NULL;
    if (B <= 0 || C <= 0 || C > HD_MAXC || I <= 0 || ldg < C || ldw < C || ldo < I || (x && ldx < I)) return LBBNN_E_SHAPE;
    const dim3 grid((I + HD_COLS - 1) / HD_COLS, (B + HD_ROWS - 1) / HD_ROWS);
    if (C == 10 && gv)
        hipLaunchKernelGGL(head_dx_c_kernel<10>, grid, dim3(HD_COLS), 0, static_cast<hipStream_t>(stream), gm, gv, ldg, wmT, wvT,
                           ldw, x, ldx, out, ldo, B, I);
    else
        hipLaunchKernelGGL(head_dx_kernel, grid, dim3(HD_COLS), 0, static_cast<hipStream_t>(stream), gm, gv, ldg, wmT, wvT, ldw,
                           x, ldx, out, ldo, B, C, I);
    return (int)hipGetLastError();
}

extern "C" int64_t lbbnn_output_grad_workspace(int B, int O) {
    if (B <= 0 || O <= 0) return 0;
    return 2LL * ((B + TS - 1) / TS) * O;
}

extern "C" int lbbnn_output_grad(const lbbnn_outgrad_args_t* p, void* stream) {
    if (!p) return LBBNN_E_NULL;
    const lbbnn_outgrad_args_t& a = *p;
    if (!a.g_out || !a.gmT || !a.work) return LBBNN_E_NULL;
    if (!a.g_sum && a.gv_sum) return LBBNN_E_NULL;
    if (a.relu && !a.out) return LBBNN_E_NULL;
    if (a.std && (!a.gvT || (a.g_sum && !a.gv_sum) || ((a.gm == nullptr) != (a.gv == nullptr)))) return LBBNN_E_NULL;
    if (a.std && !a.eps && !a.rng) return LBBNN_E_NOISE;
    if (a.B <= 0 || a.O <= 0 || a.ldg < a.O || ((a.relu || a.std) && a.ldo < a.O)) return LBBNN_E_SHAPE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int nbt = (a.B + TS - 1) / TS;
    hipLaunchKernelGGL(output_grad_kernel, dim3((a.O + TS - 1) / TS, nbt), dim3(256), 0, s, a, nbt);
    if (a.g_sum)                                            // (else the partials wait in `work` for lbbnn_reduce_partials_batch)
        hipLaunchKernelGGL(output_grad_sum_kernel, dim3((a.O + 63) / 64), dim3(1024), 0, s, a.work, nbt, a.O, a.g_sum,
                           a.std ? a.gv_sum : nullptr);
    return (int)hipGetLastError();
}

extern "C" int lbbnn_bias_backward_partials(const float* work, int B, int O, int has_gv, const float* bias_mu,
                                            const float* bias_rho, const float* g_kl, const lbbnn_priors_t* priors,
                                            float* d_bias_mu, float* d_bias_rho, void* stream) {
    if (!work || !bias_mu || !bias_rho || !priors || !d_bias_mu || !d_bias_rho) return LBBNN_E_NULL;
    if (B <= 0 || O <= 0) return LBBNN_E_SHAPE;
    const int nbt = (B + TS - 1) / TS;
    hipLaunchKernelGGL(bias_backward_partials_kernel, dim3((O + 63) / 64), dim3(1024), 0, static_cast<hipStream_t>(stream), work,
                       nbt, O, has_gv ? 1 : 0, bias_mu, bias_rho, g_kl, *priors, d_bias_mu, d_bias_rho);
    return (int)hipGetLastError();
}

extern "C" int lbbnn_reduce_partials_batch(const lbbnn_reduce_job_t* jobs, int n, void* stream) {
    if (n == 0) return LBBNN_OK;
    if (!jobs) return LBBNN_E_NULL;
    if (n < 0 || n > LBBNN_MAX_REDUCE_JOBS) return LBBNN_E_SHAPE;
    ReduceBatch bt;
    int maxcols = 0;
    for (int k = 0; k < n; ++k) {
        const lbbnn_reduce_job_t& j = jobs[k];
        if (!j.work) return LBBNN_E_NULL;
        if (j.nblk <= 0 || j.ncols <= 0 || j.nq < 1 || j.nq > 3 || j.block_stride < j.ncols || j.q_stride < j.ncols) return LBBNN_E_SHAPE;
        bt.j[k] = j;
        maxcols = j.ncols > maxcols ? j.ncols : maxcols;
    }
    hipLaunchKernelGGL(reduce_partials_batch_kernel, dim3((maxcols + 63) / 64, n), dim3(1024), 0, static_cast<hipStream_t>(stream), bt);
    return (int)hipGetLastError();
}

extern "C" int lbbnn_dx_combine(float* gx, const float* gxv, const float* x, int ldx, int B, int I, void* stream) {
    if (!gx || !gxv || !x) return LBBNN_E_NULL;
    if (B <= 0 || I <= 0 || ldx < I) return LBBNN_E_SHAPE;
    hipLaunchKernelGGL(dx_combine_kernel, dim3((I + 1023) / 1024, B), dim3(256), 0, static_cast<hipStream_t>(stream),
                       gx, gxv, x, ldx, B, I);
    return (int)hipGetLastError();
}
