// Dense coupling flows (RNVP / MNF type) applied to one vector, forward and analytic backward, as single-workgroup
// kernels (see include/lbbnn.h).  GEMV-shaped and small (I*H + 3*H*H + 2*H*I MACs per transform, H = 75 / 100):
//   input layer  a0[h] = W_in[h,:] . (m z) + b            one wave per h (coalesced rows), wave sum
//   middle H x H layers (RNVP)                              one thread per output
//   heads        shift_i / scale_i = T[i,:] . y + b         one thread per i (each reads its own rows of T, S)
// and mirrored for the gradients (outer products written by the owner of the row; dy by one wave per h).
#include <cmath>
#include "lbbnn_device.h"
#include "lbbnn_internal.h"

namespace {

using namespace lbbnn;
constexpr int NT = 1024, NWV = NT / 64, HMAX = LBBNN_MAX_HIDDEN;

struct DenseApplyArgs {
    lbbnn_dense_transform_t tr[LBBNN_MAX_DENSE_T];
    lbbnn_dense_grad_t gr[LBBNN_MAX_DENSE_T];
    const float *z_in, *d_zout, *d_logdet;
    float *z_out, *logdet, *dz_in, *work;
    int T, which, I, backward;
};

struct Hidden { float a[4][HMAX]; float yv[HMAX]; float d[2][HMAX]; };      // pre-activations, head input, deltas

__device__ __forceinline__ float lrelu(float x) { return x >= 0.f ? x : 0.1f * x; }
__device__ __forceinline__ float lrelu_d(float x) { return x >= 0.f ? 1.f : 0.1f; }

// hidden part of the forward of one transform: fills hd.a[..] and hd.yv (the vector the heads read)
__device__ __forceinline__ void hidden_forward(const LBBNN_CONST_AS lbbnn_dense_transform_t& tr, const float* m, const float* z, int I,
                                               Hidden& hd, float* wl) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, H = tr.hidden;
    for (int h = wv; h < H; h += NWV) {
        const float* w = tr.w_in + (size_t)h * I;
        float acc = 0.f;
        for (int i = lane; i < I; i += 64) acc += w[i] * (m[i] * z[i]);
        acc = wave_sum(acc);
        if (lane == 0) hd.a[0][h] = acc + tr.b_in[h];
    }
    __syncthreads();
    if (tr.kind == LBBNN_FLOW_RNVP) {
        for (int l = 0; l < 3; ++l) {
            // the H x H matrix goes through LDS (coalesced copy): a thread walking its own global row paid one L2
            // round trip per element -- 3 x 75 dependent loads made this tiny chain the longest part of the kernel
            for (int e = tid; e < H * H; e += NT) wl[e] = tr.w_mid[l][e];
            __syncthreads();
            if (tid < H) {
                float acc = tr.b_mid[l][tid];
                for (int k = 0; k < H; ++k) acc += wl[tid * H + k] * lrelu(hd.a[l][k]);
                hd.a[l + 1][tid] = acc;
            }
            __syncthreads();
        }
        if (tid < H) hd.yv[tid] = hd.a[3][tid];                       // last layer: activation dropped (flows2.py:176-185)
    } else {
        if (tid < H) hd.yv[tid] = tanhf(hd.a[0][tid]);                // flows2.py:235
    }
    __syncthreads();
}

__global__ __launch_bounds__(NT) void dense_apply_kernel(const DenseApplyArgs ka) {
    extern __shared__ __attribute__((aligned(16))) float wl[];      // HMAX x HMAX staging of one middle matrix
    __shared__ Hidden hd;
    __shared__ double sred[NWV];
    __shared__ float dpart[NWV][HMAX];
    const LBBNN_CONST_AS DenseApplyArgs& A = *kernarg_as<DenseApplyArgs>();
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, I = A.I, T = A.T;
    // work: Z[0..T] (input of every transform, Z[T] = output), DA, DB (head deltas), DZ (running dz)
    float* const Z = A.backward ? A.work : nullptr;
    float* const DA = A.backward ? A.work + (size_t)(T + 1) * I : nullptr;
    float* const DB = A.backward ? DA + I : nullptr;
    float* const DZ = A.backward ? DB + I : nullptr;
    // ---------------------------------------------------------------- forward
    const float* zcur = A.z_in;
    double ldsum = 0.0;
    if (A.backward) for (int i = tid; i < I; i += NT) Z[i] = A.z_in[i];
    for (int t = 0; t < T; ++t) {
        const LBBNN_CONST_AS lbbnn_dense_transform_t& tr = A.tr[t];
        const float* m = A.which ? tr.mask_kl : tr.mask_fwd;
        const int H = tr.hidden;
        float* znext = A.backward ? Z + (size_t)(t + 1) * I : A.z_out;
        if (A.backward) zcur = Z + (size_t)t * I;
        __syncthreads();                                              // zcur fully written by the previous transform
        hidden_forward(tr, m, zcur, I, hd, wl);
        // heads: one WAVE per row i (lanes over the H columns: coalesced 4*H-byte rows of T and S), two wave sums
        double ld = 0.0;
        const float y0 = lane < H ? hd.yv[lane] : 0.f, y1 = lane + 64 < H ? hd.yv[lane + 64] : 0.f;
        for (int i = wv; i < I; i += NWV) {
            const float* ra = tr.w_a + (size_t)i * H;
            const float* rb = tr.w_b + (size_t)i * H;
            float sa = (lane < H ? ra[lane] * y0 : 0.f) + (lane + 64 < H ? ra[lane + 64] * y1 : 0.f);
            float sb = (lane < H ? rb[lane] * y0 : 0.f) + (lane + 64 < H ? rb[lane + 64] * y1 : 0.f);
            sa = wave_sum(sa) + tr.b_a[i];
            sb = wave_sum(sb) + tr.b_b[i];
            const float g = 1.f / (1.f + expf(-sb)), mi = m[i], zi = zcur[i];
            float out;
            if (tr.kind == LBBNN_FLOW_RNVP) out = ((1.f - mi) * zi) * g + (1.f - g) * sa + mi * zi;      // flows2.py:215
            else                            out = mi * zi + (1.f - mi) * (zi * g + (1.f - g) * sa);      // flows2.py:238
            if (lane == 0) {
                ld += (double)((1.f - mi) * logf(g));
                // forward-only in place (z_in == z_out) is safe: hidden_forward has finished reading z, and element i
                // is read and written by this wave only
                znext[i] = out;
            }
        }
        ld = wave_sum(ld);
        __syncthreads();
        if (lane == 0) sred[wv] = ld;
        __syncthreads();
        for (int w2 = 0; w2 < NWV; ++w2) ldsum += sred[w2];
        if (!A.backward) zcur = A.z_out;
    }
    if (T == 0 && !A.backward) for (int i = tid; i < I; i += NT) A.z_out[i] = A.z_in[i];
    if (!A.backward) { if (tid == 0 && A.logdet) A.logdet[0] = (float)ldsum; return; }
    // ---------------------------------------------------------------- backward
    const float dld = A.d_logdet ? A.d_logdet[0] : 0.f;
    for (int i = tid; i < I; i += NT) DZ[i] = A.d_zout[i];
    for (int t = T - 1; t >= 0; --t) {
        const LBBNN_CONST_AS lbbnn_dense_transform_t& tr = A.tr[t];
        const LBBNN_CONST_AS lbbnn_dense_grad_t& gr = A.gr[t];
        const float* m = A.which ? tr.mask_kl : tr.mask_fwd;
        const float* z = Z + (size_t)t * I;
        const int H = tr.hidden;
        __syncthreads();
        hidden_forward(tr, m, z, I, hd, wl);
        // heads, one wave per row i: recompute shift / scale, the row's deltas (wave-uniform scalars), the two
        // outer-product rows dT[i,:], dS[i,:], and this wave's share of dy[k] = sum_i T[i,k] da_i + S[i,k] db_i with
        // the T / S values it has just read (lanes = columns, so nothing is re-read and nothing is strided)
        {
            const float y0 = lane < H ? hd.yv[lane] : 0.f, y1 = lane + 64 < H ? hd.yv[lane + 64] : 0.f;
            float dy0 = 0.f, dy1 = 0.f;
            for (int i = wv; i < I; i += NWV) {
                const float* ra = tr.w_a + (size_t)i * H;
                const float* rb = tr.w_b + (size_t)i * H;
                const float a0 = lane < H ? ra[lane] : 0.f, a1 = lane + 64 < H ? ra[lane + 64] : 0.f;
                const float b0 = lane < H ? rb[lane] : 0.f, b1 = lane + 64 < H ? rb[lane + 64] : 0.f;
                const float sa = wave_sum(a0 * y0 + a1 * y1) + tr.b_a[i];
                const float sb = wave_sum(b0 * y0 + b1 * y1) + tr.b_b[i];
                const float g = 1.f / (1.f + expf(-sb)), mi = m[i], zi = z[i], dout = DZ[i];
                float dg, da, dzi;
                if (tr.kind == LBBNN_FLOW_RNVP) {
                    dg = dout * ((1.f - mi) * zi - sa) + dld * (1.f - mi) / g;
                    da = dout * (1.f - g);
                    dzi = dout * ((1.f - mi) * g + mi);
                } else {
                    dg = dout * (1.f - mi) * (zi - sa) + dld * (1.f - mi) / g;
                    da = dout * (1.f - mi) * (1.f - g);
                    dzi = dout * (mi + (1.f - mi) * g);
                }
                const float db = dg * g * (1.f - g);
                float* ga = gr.w_a + (size_t)i * H;
                float* gb = gr.w_b + (size_t)i * H;
                if (lane < H) { ga[lane] = da * y0; gb[lane] = db * y0; }
                if (lane + 64 < H) { ga[lane + 64] = da * y1; gb[lane + 64] = db * y1; }
                dy0 += a0 * da + b0 * db;
                dy1 += a1 * da + b1 * db;
                if (lane == 0) { DZ[i] = dzi; gr.b_a[i] = da; gr.b_b[i] = db; }
            }
            __syncthreads();                                          // hd.d free; every wave's dy share complete
            dpart[wv][lane] = dy0; dpart[wv][lane + 64] = dy1;
            __syncthreads();
            if (tid < H) {
                float acc = 0.f;
                for (int w2 = 0; w2 < NWV; ++w2) acc += dpart[w2][tid];   // fixed order
                hd.d[0][tid] = acc;
            }
            __syncthreads();
        }
        int cur = 0;                                                  // hd.d[cur] = delta wrt the current layer's output
        if (tr.kind == LBBNN_FLOW_RNVP) {
            for (int l = 2; l >= 0; --l) {
                // layer l: a[l+1] = W_mid[l] lrelu(a[l]) + b_mid[l];  delta = d L / d a[l+1]
                for (int e = tid; e < H * H; e += NT) {
                    gr.w_mid[l][e] = hd.d[cur][e / H] * lrelu(hd.a[l][e % H]);
                    wl[e] = tr.w_mid[l][e];
                }
                __syncthreads();
                if (tid < H) {
                    gr.b_mid[l][tid] = hd.d[cur][tid];
                    float acc = 0.f;
                    for (int h = 0; h < H; ++h) acc += wl[h * H + tid] * hd.d[cur][h];
                    hd.d[cur ^ 1][tid] = acc * lrelu_d(hd.a[l][tid]);
                }
                __syncthreads();
                cur ^= 1;
            }
        } else {
            if (tid < H) { const float th = hd.yv[tid]; hd.d[1][tid] = hd.d[0][tid] * (1.f - th * th); }
            __syncthreads();
            cur = 1;
        }
        // input layer: dW_in[h,i] = delta[h] * (m z)_i, db_in = delta, dz_i += m_i * sum_h W_in[h,i] delta[h]
        for (int h = wv; h < H; h += NWV) {
            const float dh = hd.d[cur][h];
            float* gw = gr.w_in + (size_t)h * I;
            for (int i = lane; i < I; i += 64) gw[i] = dh * (m[i] * z[i]);
            if (lane == 0) gr.b_in[h] = dh;
        }
        for (int i = tid; i < I; i += NT) {
            float acc = 0.f;
            int h = 0;
            for (; h + 5 <= H; h += 5) {                               // 5 independent (coalesced) loads in flight per trip
                const float w0 = tr.w_in[(size_t)h * I + i], w1 = tr.w_in[(size_t)(h + 1) * I + i], w2 = tr.w_in[(size_t)(h + 2) * I + i];
                const float w3 = tr.w_in[(size_t)(h + 3) * I + i], w4 = tr.w_in[(size_t)(h + 4) * I + i];
                acc += w0 * hd.d[cur][h] + w1 * hd.d[cur][h + 1] + w2 * hd.d[cur][h + 2] + w3 * hd.d[cur][h + 3] + w4 * hd.d[cur][h + 4];
            }
            for (; h < H; ++h) acc += tr.w_in[(size_t)h * I + i] * hd.d[cur][h];
            DZ[i] += m[i] * acc;
        }
    }
    __syncthreads();
    for (int i = tid; i < I; i += NT) A.dz_in[i] = DZ[i];
}

constexpr size_t kDynLds = (size_t)HMAX * HMAX * sizeof(float);      // 64 KB on top of ~14 KB static: limit raised once

int raise_lds() {
    static bool done = false;
    if (!done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(dense_apply_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)kDynLds);
        if (e != hipSuccess) return (int)e;
        done = true;
    }
    return 0;
}

int check_transforms(const lbbnn_dense_transform_t* tr, int T, int which) {
    for (int t = 0; t < T; ++t) {
        const lbbnn_dense_transform_t& d = tr[t];
        if (d.kind != LBBNN_FLOW_RNVP && d.kind != LBBNN_FLOW_MNF) return LBBNN_E_FLAGS;
        if (d.hidden <= 0 || d.hidden > HMAX) return LBBNN_E_SHAPE;
        if (!d.w_in || !d.b_in || !d.w_a || !d.b_a || !d.w_b || !d.b_b || !(which ? d.mask_kl : d.mask_fwd)) return LBBNN_E_NULL;
        if (d.kind == LBBNN_FLOW_RNVP) for (int l = 0; l < 3; ++l) if (!d.w_mid[l] || !d.b_mid[l]) return LBBNN_E_NULL;
    }
    return 0;
}

}  // namespace

extern "C" int64_t lbbnn_flow_dense_apply_workspace(int I, int T) {
    if (I <= 0 || T < 0) return 0;
    return (int64_t)I * (T + 1 + 3);
}

extern "C" int lbbnn_flow_dense_apply(const lbbnn_dense_transform_t* tr, int T, int which_mask, const float* z_in, int I,
                                      float* z_out, float* logdet, void* stream) {
    if ((T > 0 && !tr) || !z_in || !z_out) return LBBNN_E_NULL;
    if (I <= 0 || T < 0 || T > LBBNN_MAX_DENSE_T) return LBBNN_E_SHAPE;
    if (const int rc = check_transforms(tr, T, which_mask)) return rc;
    DenseApplyArgs a{};
    for (int t = 0; t < T; ++t) a.tr[t] = tr[t];
    a.z_in = z_in; a.z_out = z_out; a.logdet = logdet; a.T = T; a.which = which_mask ? 1 : 0; a.I = I; a.backward = 0;
    if (const int rc = raise_lds()) return rc;
    hipLaunchKernelGGL(dense_apply_kernel, dim3(1), dim3(NT), kDynLds, static_cast<hipStream_t>(stream), a);
    return (int)hipGetLastError();
}

extern "C" int lbbnn_flow_dense_apply_backward(const lbbnn_dense_transform_t* tr, const lbbnn_dense_grad_t* grads, int T,
                                               int which_mask, const float* z_in, const float* d_zout, const float* d_logdet,
                                               int I, float* dz_in, float* work, void* stream) {
    if ((T > 0 && (!tr || !grads)) || !z_in || !d_zout || !dz_in || !work) return LBBNN_E_NULL;
    if (I <= 0 || T < 0 || T > LBBNN_MAX_DENSE_T) return LBBNN_E_SHAPE;
    if (const int rc = check_transforms(tr, T, which_mask)) return rc;
    DenseApplyArgs a{};
    for (int t = 0; t < T; ++t) {
        a.tr[t] = tr[t]; a.gr[t] = grads[t];
        const lbbnn_dense_grad_t& g = grads[t];
        if (!g.w_in || !g.b_in || !g.w_a || !g.b_a || !g.w_b || !g.b_b) return LBBNN_E_NULL;
        if (tr[t].kind == LBBNN_FLOW_RNVP) for (int l = 0; l < 3; ++l) if (!g.w_mid[l] || !g.b_mid[l]) return LBBNN_E_NULL;
    }
    a.z_in = z_in; a.d_zout = d_zout; a.d_logdet = d_logdet; a.dz_in = dz_in; a.work = work;
    a.T = T; a.which = which_mask ? 1 : 0; a.I = I; a.backward = 1;
    if (const int rc = raise_lds()) return rc;
    hipLaunchKernelGGL(dense_apply_kernel, dim3(1), dim3(NT), kDynLds, static_cast<hipStream_t>(stream), a);
    return (int)hipGetLastError();
}
