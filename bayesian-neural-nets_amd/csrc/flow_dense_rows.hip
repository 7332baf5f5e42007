// K4r  lbbnn_flow_dense_rows -- a dense coupling flow (RNVP / MNF type, flows2.py:188-241) applied to R ROWS at once,
// the dense affine steps on the matrix cores (v_mfma_f32_16x16x4_f32: exact fp32).
//
// Where this is on the reference's path: `sample_z(batch_size)` (LBBNN-GP-MF-MNF.py:182-187) runs `z_flow` on a (B,I)
// matrix -- every row with its own Bernoulli mask -- and keeps only the LAST row; the MNF layer kernels of this library
// (flow_dense.hip) compute that kept row alone.  This kernel is the as-written form: all R rows, used by the stand-alone
// `PropagateFlow('RNVP'|'MNF').forward(z)` of the Python boundary (flows2.py:41-46) and by the R = B mode of `sample_z`.
//
// One 256-thread workgroup owns 16 rows for the WHOLE chain (rows are independent): their z lives in LDS across all T
// transforms, so HBM sees z_in once and z_out once.  Per transform:
//   A  hidden pre-activation  P (H x 16) = W_in (H x I) . (m (.) z)^T        K = I split over the 4 waves, summed in LDS
//   B  RNVP only: three H x H layers, n-blocks over the waves
//   C  heads S_a, S_b (I x 16) = W_a / W_b (I x H) . y, n-blocks over the waves; gate / update / log-det in the epilogue,
//      z updated in place in LDS.
// MFMA orientation as in lrt_gemm.hip: "A" = 16 output features of the weight matrix, "B" = the 16 rows, so a lane's
// accumulator holds 4 consecutive features of ONE row (float4 reads of z / masks, float4 stores).  K permutation: lane
// quarter q feeds element j of its float4 (k = 16c + 4q + j) to MFMA j on both operands.
// LDS image of z: chunk-major [I/16][16 rows][16 floats], slot q of row r stored at q ^ F[(r>>2)&3], F = {0,2,3,1}:
// conflict-free under ds_read_b128's lane groups (checked exhaustively); hidden matrices [16][Hp + 8] likewise.
// Deterministic: every sum has a fixed order; no atomics.
#include "lbbnn_device.h"
#include "lbbnn_internal.h"

namespace lbbnn {

typedef float f4 __attribute__((ext_vector_type(4)));

struct DenseRowsArgs {
    lbbnn_dense_transform_t tr[LBBNN_MAX_DENSE_T];
    int T;
    const float* masks;        // [T][R][I] in {0,1}, or NULL: Bernoulli(0.5) from Philox (rng, stream)
    float* mask_out;           // NULL, or [T][R][I]: the masks used
    const uint64_t* rng;
    uint32_t stream;
    const float* z_in; int ldz;
    float* z_out; int ldo;
    float* logdet;             // (R): sum over transforms of sum_i (1-m) log gate
    int R, I;
    uint64_t row_base;         // global index of row 0 (Philox counters: data-parallel shards draw distinct masks)
};

__device__ __forceinline__ int swzr(int row) { return (0x78 >> (2 * ((row >> 2) & 3))) & 3; }   // F = {0,2,3,1}

// 4 consecutive elements W[row][k..k+3] of a row-major (nrows x K) matrix; zero outside
__device__ __forceinline__ f4 load_w4(const float* __restrict__ W, int row, int nrows, int k, int K) {
    f4 v = {0.f, 0.f, 0.f, 0.f};
    if (row >= nrows) return v;
    const float* p = W + (size_t)row * K + k;
    if (k + 3 < K && ((reinterpret_cast<uintptr_t>(p) & 15) == 0)) return *reinterpret_cast<const f4*>(p);
    if (k < K) v.x = p[0];
    if (k + 1 < K) v.y = p[1];
    if (k + 2 < K) v.z = p[2];
    if (k + 3 < K) v.w = p[3];
    return v;
}

__device__ __forceinline__ f4 load_vec4(const float* __restrict__ v, int k, int K) {
    return load_w4(v, 0, 1, k, K);
}

__device__ __forceinline__ f4 mfma4(f4 a, f4 b, f4 acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc, 0, 0, 0);
    return acc;
}

constexpr int kRows = 16;
constexpr int kMaxNB = LBBNN_MAX_HIDDEN / 16;      // hidden n-blocks (H <= 128)

// mask of elements (grow, i..i+3) of transform t
__device__ __forceinline__ f4 mask4(const float* __restrict__ masks, uint64_t seed, uint64_t offset, uint32_t stream,
                                    int t, int row, uint64_t grow, int i, int R, int I) {
    f4 m = {0.f, 0.f, 0.f, 0.f};
    if (row >= R || i >= I) return m;
    if (masks != nullptr) return load_w4(masks + ((size_t)t * R + row) * I, 0, 1, i, I);
    const Philox4 b = philox_bits4(seed, offset, stream, grow, ((uint32_t)t << 24) | (uint32_t)(i >> 2));
    m.x = (float)(b.x & 1u);
    m.y = (i + 1 < I) ? (float)(b.y & 1u) : 0.f;
    m.z = (i + 2 < I) ? (float)(b.z & 1u) : 0.f;
    m.w = (i + 3 < I) ? (float)(b.w & 1u) : 0.f;
    return m;
}

__global__ __launch_bounds__(256) void flow_dense_rows_kernel(DenseRowsArgs a_) {
    const LBBNN_CONST_AS DenseRowsArgs* a = kernarg_as<DenseRowsArgs>();
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int lr = lane & 15, q = lane >> 4;                 // row of the tile / k quarter (operands), feature quad (results)
    const int R = a->R, I = a->I, T = a->T;
    const int C = (I + 15) >> 4;                             // 16-float chunks of a z row
    const int row0 = blockIdx.x * kRows;
    const int grow_i = row0 + lr;
    const uint64_t grow = a->row_base + (uint64_t)grow_i;
    // LDS carve-up (floats)
    float* Z = lds;                                          // [C][16][16]
    float* H0 = Z + (size_t)C * 256;                         // [16][HS]
    const int HS = LBBNN_MAX_HIDDEN + 8;
    float* H1 = H0 + kRows * HS;
    float* part = H1 + kRows * HS;                           // [4 waves][kMaxNB][64][4]
    float* ldp = part + 4 * kMaxNB * 256;                    // [256]
    uint64_t seed = 0, offset = 0;
    const float* masks = a->masks;
    if (masks == nullptr) { seed = a->rng[0]; offset = a->rng[1]; }
    const uint32_t stream = a->stream;

    // ---- z rows -> LDS image (swizzled slots), zero padding for rows >= R and k >= I
    {
        const float* zin = a->z_in;
        const int ldz = a->ldz;
        for (int u = tid; u < C * 64; u += 256) {            // u = (chunk, row, slot)
            const int c = u >> 6, r = (u >> 2) & 15, s = u & 3;
            f4 v = {0.f, 0.f, 0.f, 0.f};
            if (row0 + r < R) v = load_w4(zin + (size_t)(row0 + r) * ldz, 0, 1, 16 * c + 4 * s, I);
            *reinterpret_cast<f4*>(Z + c * 256 + r * 16 + ((s ^ swzr(r)) << 2)) = v;
        }
    }
    float ld_acc = 0.f;
    __syncthreads();

    for (int t = 0; t < T; ++t) {
        const int kind = a->tr[t].kind, H = a->tr[t].hidden;
        const int NB = (H + 15) >> 4;                         // hidden n-blocks
        // ------------------------------------------------------------------ A: input layer, K = I split over the waves
        {
            const float* W = a->tr[t].w_in;
            f4 acc[kMaxNB];
#pragma unroll
            for (int nb = 0; nb < kMaxNB; ++nb) acc[nb] = f4{0.f, 0.f, 0.f, 0.f};
            for (int c = w; c < C; c += 4) {
                const f4 z = *reinterpret_cast<const f4*>(Z + c * 256 + lr * 16 + ((q ^ swzr(lr)) << 2));
                const f4 m = mask4(masks, seed, offset, stream, t, grow_i, grow, 16 * c + 4 * q, R, I);
                const f4 mz = m * z;
#pragma unroll
                for (int nb = 0; nb < kMaxNB; ++nb) {
                    if (nb < NB) {
                        const f4 wv = load_w4(W, 16 * nb + lr, H, 16 * c + 4 * q, I);
                        acc[nb] = mfma4(wv, mz, acc[nb]);
                    }
                }
            }
#pragma unroll
            for (int nb = 0; nb < kMaxNB; ++nb)
                if (nb < NB) *reinterpret_cast<f4*>(part + ((w * kMaxNB + nb) * 64 + lane) * 4) = acc[nb];
            __syncthreads();
            const float* bias = a->tr[t].b_in;
            for (int u = tid; u < NB * 64; u += 256) {
                const int nb = u >> 6, ln = u & 63;
                f4 s = *reinterpret_cast<const f4*>(part + ((0 * kMaxNB + nb) * 64 + ln) * 4);
                s += *reinterpret_cast<const f4*>(part + ((1 * kMaxNB + nb) * 64 + ln) * 4);
                s += *reinterpret_cast<const f4*>(part + ((2 * kMaxNB + nb) * 64 + ln) * 4);
                s += *reinterpret_cast<const f4*>(part + ((3 * kMaxNB + nb) * 64 + ln) * 4);
                const int n = 16 * nb + 4 * (ln >> 4);
                s += load_vec4(bias, n, H);
                if (kind == 0) {                              // RNVP: LeakyReLU(0.1)  (flows2.py:176-185)
                    s.x = s.x > 0.f ? s.x : 0.1f * s.x; s.y = s.y > 0.f ? s.y : 0.1f * s.y;
                    s.z = s.z > 0.f ? s.z : 0.1f * s.z; s.w = s.w > 0.f ? s.w : 0.1f * s.w;
                } else {                                      // MNF type: tanh  (flows2.py:235)
                    s.x = tanhf(s.x); s.y = tanhf(s.y); s.z = tanhf(s.z); s.w = tanhf(s.w);
                }
                *reinterpret_cast<f4*>(H0 + (ln & 15) * HS + n) = s;
            }
            __syncthreads();
        }
        float* Hin = H0;
        float* Hout = H1;
        // ------------------------------------------------------------------ B: RNVP middle layers (H x H)
        if (kind == 0) {
            for (int l = 0; l < 3; ++l) {
                const float* W = a->tr[t].w_mid[l];
                const float* bias = a->tr[t].b_mid[l];
                for (int nb = w; nb < NB; nb += 4) {
                    f4 acc = {0.f, 0.f, 0.f, 0.f};
                    for (int c = 0; c < NB; ++c) {
                        const f4 hv = *reinterpret_cast<const f4*>(Hin + lr * HS + 16 * c + 4 * q);
                        const f4 wv = load_w4(W, 16 * nb + lr, H, 16 * c + 4 * q, H);
                        acc = mfma4(wv, hv, acc);
                    }
                    const int n = 16 * nb + 4 * q;
                    acc += load_vec4(bias, n, H);
                    if (l < 2) {                              // the last activation of the MLP is dropped (flows2.py:184)
                        acc.x = acc.x > 0.f ? acc.x : 0.1f * acc.x; acc.y = acc.y > 0.f ? acc.y : 0.1f * acc.y;
                        acc.z = acc.z > 0.f ? acc.z : 0.1f * acc.z; acc.w = acc.w > 0.f ? acc.w : 0.1f * acc.w;
                    }
                    // features >= H stay exactly zero (zero weights rows, zero bias): they are the K padding of the next layer
                    *reinterpret_cast<f4*>(Hout + lr * HS + n) = acc;
                }
                __syncthreads();
                float* tmp = Hin; Hin = Hout; Hout = tmp;
            }
        }
        // ------------------------------------------------------------------ C: heads + gate + update, n-blocks of I over the waves
        {
            const float* Wa = a->tr[t].w_a;
            const float* Wb = a->tr[t].w_b;
            const float* ba = a->tr[t].b_a;
            const float* bb = a->tr[t].b_b;
            float* mout = a->mask_out;
            for (int nb = w; nb < C; nb += 4) {
                f4 sa = {0.f, 0.f, 0.f, 0.f}, sb = {0.f, 0.f, 0.f, 0.f};
                for (int c = 0; c < NB; ++c) {
                    const f4 hv = *reinterpret_cast<const f4*>(Hin + lr * HS + 16 * c + 4 * q);
                    const f4 wa = load_w4(Wa, 16 * nb + lr, I, 16 * c + 4 * q, H);
                    const f4 wb = load_w4(Wb, 16 * nb + lr, I, 16 * c + 4 * q, H);
                    sa = mfma4(wa, hv, sa);
                    sb = mfma4(wb, hv, sb);
                }
                const int i = 16 * nb + 4 * q;               // this lane: features i..i+3 of row lr
                sa += load_vec4(ba, i, I);
                sb += load_vec4(bb, i, I);
                float* zp = Z + nb * 256 + lr * 16 + ((q ^ swzr(lr)) << 2);
                const f4 z = *reinterpret_cast<const f4*>(zp);
                const f4 m = mask4(masks, seed, offset, stream, t, grow_i, grow, i, R, I);
                f4 x;
                float ld = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float gate = 1.0f / (1.0f + expf(-sb[j]));          // sigmoid(scale) / sigmoid(k(h))
                    const float keep = m[j] * z[j], move = (1.f - m[j]) * z[j];
                    // RNVP  (flows2.py:211-215): x = z1*gate + (1-gate)*shift + z2,  z1 = (1-m) z, z2 = m z
                    // MNF   (flows2.py:238):     x = m z + (1-m) (z sigma + (1-sigma) mu)
                    x[j] = (kind == 0) ? (move * gate + (1.f - gate) * sa[j]) + keep
                                       : keep + (1.f - m[j]) * (z[j] * gate + (1.f - gate) * sa[j]);
                    if (i + j < I) ld += (1.f - m[j]) * logf(gate);
                    else x[j] = 0.f;
                }
                if (grow_i < R) {
                    ld_acc += ld;
                    if (mout != nullptr) {
                        float* mp = mout + ((size_t)t * R + grow_i) * I + i;
#pragma unroll
                        for (int j = 0; j < 4; ++j) if (i + j < I) mp[j] = m[j];
                    }
                } else {
                    x = f4{0.f, 0.f, 0.f, 0.f};
                }
                *reinterpret_cast<f4*>(zp) = x;
            }
            __syncthreads();
        }
    }
    // ---- z_out, per-row log-det (fixed order: waves, then feature quads)
    {
        float* zo = a->z_out;
        const int ldo = a->ldo;
        for (int u = tid; u < C * 64; u += 256) {
            const int c = u >> 6, r = (u >> 2) & 15, s = u & 3;
            if (row0 + r < R) {
                const f4 v = *reinterpret_cast<const f4*>(Z + c * 256 + r * 16 + ((s ^ swzr(r)) << 2));
                float* p = zo + (size_t)(row0 + r) * ldo + 16 * c + 4 * s;
                const int k = 16 * c + 4 * s;
                if (k + 3 < I && ((reinterpret_cast<uintptr_t>(p) & 15) == 0)) *reinterpret_cast<f4*>(p) = v;
                else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (k + j < I) p[j] = v[j];
                }
            }
        }
        ldp[tid] = ld_acc;
        __syncthreads();
        if (tid < kRows && row0 + tid < R && a->logdet != nullptr) {
            double s = 0.0;
            for (int ww = 0; ww < 4; ++ww)
                for (int qq = 0; qq < 4; ++qq) s += (double)ldp[ww * 64 + qq * 16 + tid];
            a->logdet[row0 + tid] = (float)s;
        }
    }
}

// z0[r][i] = q0_mean[i] + exp(q0_log_var[i])^(1/2) * eps[r][i]  (LBBNN-GP-MF-MNF.py:183-185 for batch_size = R).
// eps explicit (R,I) or N(0,1) from Philox: counter (i/4, R-1-r) of `stream` -- the LAST row's draw is the 1-D draw the
// fused layer kernels make for the kept row (counter (i/4, 0)), so the as-written R-row mode and the R = 1 fast path see
// the same z0 there.  HBM-bound elementwise pass, 4 elements per thread.
__global__ __launch_bounds__(256) void q0_rows_kernel(const float* __restrict__ q0_mean, const float* __restrict__ q0_log_var,
                                                      const float* __restrict__ eps, const uint64_t* __restrict__ rng,
                                                      uint32_t stream, int R, int I, float* __restrict__ z0) {
    const int nq = (I + 3) >> 2;
    const size_t u = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (u >= (size_t)R * nq) return;
    const int r = (int)(u / nq), i = 4 * (int)(u % nq);
    float e[4];
    if (eps != nullptr) {
        for (int j = 0; j < 4; ++j) e[j] = (i + j < I) ? eps[(size_t)r * I + i + j] : 0.f;
    } else {
        philox_normal4(rng[0], rng[1], stream, (uint64_t)(i >> 2), (uint32_t)(R - 1 - r), e);
    }
    for (int j = 0; j < 4; ++j)
        if (i + j < I) z0[(size_t)r * I + i + j] = q0_mean[i + j] + sqrtf(expf(q0_log_var[i + j])) * e[j];
}

static size_t rows_lds_bytes(int I) {
    const int C = (I + 15) / 16;
    const int HS = LBBNN_MAX_HIDDEN + 8;
    return sizeof(float) * ((size_t)C * 256 + 2 * kRows * HS + 4 * kMaxNB * 256 + 256);
}

}  // namespace lbbnn

extern "C" int lbbnn_q0_rows(const float* q0_mean, const float* q0_log_var, const float* eps, const uint64_t* rng,
                             uint32_t rng_stream, int R, int I, float* z0, void* stream) {
    if (!q0_mean || !q0_log_var || !z0) return LBBNN_E_NULL;
    if (!eps && !rng) return LBBNN_E_NOISE;
    if (R < 0 || I < 1) return LBBNN_E_SHAPE;
    if (R == 0) return 0;
    const size_t n = (size_t)R * ((I + 3) / 4);
    hipLaunchKernelGGL(lbbnn::q0_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, q0_mean,
                       q0_log_var, eps, rng, rng_stream, R, I, z0);
    return (int)hipGetLastError();
}

extern "C" int lbbnn_flow_dense_rows_max_dim(void) {
    int I = 16;
    while (lbbnn::rows_lds_bytes(I + 16) <= 160 * 1024) I += 16;
    return I;
}

extern "C" int lbbnn_flow_dense_rows(const lbbnn_dense_transform_t* tr, int T, const float* masks, float* mask_out,
                                     const uint64_t* rng, uint32_t rng_stream, uint64_t row_base,
                                     const float* z_in, int ldz, int R, int I,
                                     float* z_out, int ldo, float* logdet_rows, void* stream) {
    using namespace lbbnn;
    if (tr == nullptr || z_in == nullptr || z_out == nullptr) return LBBNN_E_NULL;
    if (masks == nullptr && rng == nullptr) return LBBNN_E_NULL;
    if (T < 0 || T > LBBNN_MAX_DENSE_T || R < 0 || I < 1 || ldz < I || ldo < I) return LBBNN_E_SHAPE;
    if (I > lbbnn_flow_dense_rows_max_dim()) return LBBNN_E_SHAPE;
    DenseRowsArgs a{};
    for (int t = 0; t < T; ++t) {
        const lbbnn_dense_transform_t& d = tr[t];
        if (d.hidden < 1 || d.hidden > LBBNN_MAX_HIDDEN || (d.kind != 0 && d.kind != 1)) return LBBNN_E_SHAPE;
        if (!d.w_in || !d.b_in || !d.w_a || !d.b_a || !d.w_b || !d.b_b) return LBBNN_E_NULL;
        if (d.kind == 0)
            for (int l = 0; l < 3; ++l) if (!d.w_mid[l] || !d.b_mid[l]) return LBBNN_E_NULL;
        a.tr[t] = d;
    }
    if (R == 0) return 0;
    a.T = T; a.masks = masks; a.mask_out = mask_out; a.rng = rng; a.stream = rng_stream;
    a.z_in = z_in; a.ldz = ldz; a.z_out = z_out; a.ldo = ldo; a.logdet = logdet_rows; a.R = R; a.I = I;
    a.row_base = row_base;
    const size_t lds = rows_lds_bytes(I);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(flow_dense_rows_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL(flow_dense_rows_kernel, dim3((R + kRows - 1) / kRows), dim3(256), lds, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}
