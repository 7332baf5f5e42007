// K4 -- dense coupling flows (RNVP, MNF type) on the single kept z row (gfx950).  See include/lbbnn.h.
//
// All work is GEMV-shaped and small (I*H + 3*H*H + 2*H*I MACs per transform, H = 75/100), so the design
// goal is latency: the two I-long stages are spread over many workgroups, the H x H chain (17 k MACs) is
// recomputed by every output workgroup instead of costing a launch, both paths of a layer (forward
// draw | KL branch) ride in the same launches (blockIdx.y = path), and the I-long input GEMV of transform t+1 is
// folded into the output stage of transform t: each output workgroup adds its 64 rows' share of the next hidden
// pre-activations (NextA / partial_a) and every workgroup of the next launch sums the per-workgroup partials in a fixed
// order -- ONE launch per transform, no atomics.
#include "lbbnn_device.h"
#include "lbbnn_internal.h"

namespace {

using namespace lbbnn;

constexpr int HMAX = LBBNN_MAX_HIDDEN;
constexpr int CB = 64;             // outputs per workgroup in the output stage (4 waves x 16 rows)
constexpr int NTC = 256;           // threads of the output stage

// what a launch computes for the NEXT transform: P[path][wg][j] = sum over the workgroup's rows of W_in[j,i] * (m_i z_i)
struct NextA {
    const float* w_in;        // (H,I) of the next transform, NULL: nothing follows
    const float* mask[2];     // its mask per path
    float* P;                 // [2][nwg][HMAX]
    int hidden, path_lo, npaths;
};

// Two halves so that the loads can be requested at the top of a kernel, with everything else it reads, and the
// reduction run at its end: rows r = wv, wv+4, ... of W_in for this lane's column, and the lane's mask value.
constexpr int RJ = HMAX / 4;
struct PartialRegs { float w[RJ]; float m; bool on; };

__device__ __forceinline__ void partial_a_load(const LBBNN_CONST_AS NextA& nx, int path, int row0, int I, PartialRegs& r) {
    r.on = nx.w_in && path >= nx.path_lo && path < nx.path_lo + nx.npaths;            // uniform
    r.m = 0.f;
#pragma unroll
    for (int jj = 0; jj < RJ; ++jj) r.w[jj] = 0.f;
    if (!r.on) return;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, H = nx.hidden;
    const int i = row0 + lane;
    if (i >= I) return;
    r.m = nx.mask[path][i];
#pragma unroll
    for (int jj = 0; jj < RJ; ++jj) { const int j = wv + 4 * jj; if (j < H) r.w[jj] = nx.w_in[(size_t)j * I + i]; }
}

// the same without reading the mask from memory (the caller supplies r.m)
__device__ __forceinline__ void partial_a_load_nomask(const LBBNN_CONST_AS NextA& nx, int path, int row0, int I, PartialRegs& r) {
    r.on = nx.w_in && path >= nx.path_lo && path < nx.path_lo + nx.npaths;            // uniform
    r.m = 0.f;
#pragma unroll
    for (int jj = 0; jj < RJ; ++jj) r.w[jj] = 0.f;
    if (!r.on) return;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, H = nx.hidden;
    const int i = row0 + lane;
    if (i >= I) return;
#pragma unroll
    for (int jj = 0; jj < RJ; ++jj) { const int j = wv + 4 * jj; if (j < H) r.w[jj] = nx.w_in[(size_t)j * I + i]; }
}

// znew: LDS, the new z of this workgroup's CB rows (lane = row; 0 past the end of the vector)
__device__ __forceinline__ void partial_a_reduce(const LBBNN_CONST_AS NextA& nx, int path, const float* znew, int nwg, const PartialRegs& r) {
    if (!r.on) return;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, H = nx.hidden;
    const float mz = r.m * znew[lane];
    float* dst = nx.P + ((size_t)path * nwg + blockIdx.x) * HMAX;
#pragma unroll
    for (int jj = 0; jj < RJ; ++jj) {
        const int j = wv + 4 * jj;
        if (j >= H) break;                                                           // uniform
        const float sum = wave_sum(r.w[jj] * mz);
        if (lane == 0) dst[j] = sum;
    }
}

struct InitArgs {
    const float* q0_mean; const float* q0_log_var; const float* eps[2]; const uint64_t* rng;
    float* z[2]; float* lq0_part;   // per-workgroup partials of log_q0 (KL path)
    NextA nx;
    int I; uint32_t layer; int npaths;
    // draw_masks: mask vectors to fill -- [0] z flow / forward call, [1] z flow / KL call, [2] r flow; Tm[k] of them each
    float* mk[3][LBBNN_MAX_FLOW_T];
    int Tm[3];
    int draw, nx_word;              // nx_word: which of the three words holds the mask partial_a needs on path 1 (1 or 2)
};

// z0 = q0_mean + exp(q0_log_var)^.5 * eps  (LBBNN-GP-MF-MNF.py:183-185) for both paths; log_q0 partials (:213-214);
// the first transform's partial pre-activations
struct InitBatch { InitArgs l[LBBNN_MAX_LAYERS]; };

__global__ __launch_bounds__(NTC) void dense_init_kernel(const InitBatch bt) {
    __shared__ float znew[CB];
    const LBBNN_CONST_AS InitArgs& a = kernarg_as<InitBatch>()->l[blockIdx.z];      // layer = blockIdx.z, no scratch copy
    const int path = blockIdx.y;
    if (blockIdx.x * CB >= a.I) return;                                             // grid sized for the widest layer
    const int row0 = blockIdx.x * CB, tid = threadIdx.x;
    PartialRegs pr;
    if (a.draw) {
        // Bernoulli(0.5) masks from Philox: one call gives this row's bit for every transform (every wave needs its own
        // lane's bits for the partial below; wave 0 also stores them for the later launches and the backward)
        const int i = row0 + (tid & 63);
        const Philox4 b = philox_bits4(a.rng[0], a.rng[1], LBBNN_STREAM_MASK * 64u + a.layer, (uint64_t)i, 0u);
        if (tid < CB && i < a.I) {
            if (path == 0) {
                for (int t = 0; t < a.Tm[0]; ++t) a.mk[0][t][i] = (float)((b.x >> t) & 1u);
            } else {
                for (int t = 0; t < a.Tm[1]; ++t) a.mk[1][t][i] = (float)((b.y >> t) & 1u);
                for (int t = 0; t < a.Tm[2]; ++t) a.mk[2][t][i] = (float)((b.z >> t) & 1u);
            }
        }
        // the first transform's mask for partial_a: bit 0 of this path's word (not yet in memory)
        partial_a_load_nomask(a.nx, path, row0, a.I, pr);
        const uint32_t w0 = path == 0 ? b.x : (a.nx_word == 2 ? b.z : b.y);
        pr.m = (pr.on && i < a.I) ? (float)(w0 & 1u) : 0.f;
    } else {
        partial_a_load(a.nx, path, row0, a.I, pr);
    }
    if (tid < CB) {                                                                 // wave 0: one row per lane
        const int i = row0 + tid;
        float lq = 0.f, z0 = 0.f;
        if (i < a.I) {
            float e;
            if (a.eps[path]) e = a.eps[path][i];
            else {
                float n[4];
                philox_normal4(a.rng[0], a.rng[1], (path ? LBBNN_STREAM_EPS_Z2 : LBBNN_STREAM_EPS_Z) * 64u + a.layer, (uint64_t)(i >> 2), 0u, n);
                e = n[i & 3];
            }
            const float lv = a.q0_log_var[i], qm = a.q0_mean[i];
            const float ev = expf(lv);
            z0 = qm + sqrtf(ev) * e;
            a.z[path][i] = z0;
            const float d = z0 - qm;
            lq = -0.5f * 1.1447298858494002f - 0.5f * lv - 0.5f * ((d * d) / ev);
        }
        znew[tid] = z0;
        if (path == 1) {
            const double s = wave_sum((double)lq);
            if (tid == 0) a.lq0_part[blockIdx.x] = (float)s;
        }
    }
    __syncthreads();
    partial_a_reduce(a.nx, path, znew, (a.I + CB - 1) / CB, pr);
}

struct StageArgs {
    lbbnn_dense_transform_t tr;
    const float* zin[2];  // current z of each path ...
    float* zout[2];       // ... and where the output stage puts the new one (== zin: in place; the next slot when the
                          // intermediates are kept for the backward, lbbnn_dense_layer_t::save)
    float* h[2];          // hidden activations of stage A, HMAX floats per path (slot 0 of the kept 4 x HMAX block)
    int keep_chain;       // workgroup 0 of the output stage also stores h2 | h3 | head input behind h[path]
    float* zcopy[2];      // if non-NULL: the path also stores its new z here (z_fwd / z_kl after the last z_flow transform)
    float* ld_part[2];    // per-workgroup log-det partials of this transform
    int I; int path_lo, npaths;   // paths handled: path_lo .. path_lo + npaths - 1
    const float* Pin;     // [2][nwg][HMAX]: this transform's hidden pre-activation partials, written by the previous launch
    NextA nx;
};

struct StageBatch { StageArgs l[LBBNN_MAX_LAYERS]; };

// One launch per transform.  Every workgroup: h1 = act(sum of the previous launch's partials + b_in)
//   (RNVP: LeakyReLU(0.1), flows2.py:176-185,212; MNF type: tanh, flows2.py:235), (RNVP) the MLP chain h1 -> h4 in LDS,
// then for its 64 outputs the two H-long dots, the gate and the coupling update, its log-det partial, and its rows' share
// of the next transform's hidden pre-activations.
__global__ __launch_bounds__(NTC) void dense_stage_c_kernel(const StageBatch bt, int use_lds) {
    extern __shared__ __attribute__((aligned(16))) float wl[];      // the three H x H middle matrices (use_lds)
    __shared__ float hs[2][HMAX];
    __shared__ float ps[HMAX];           // second-half partial dots of the middle layers
    __shared__ double scratch[4];
    __shared__ float znew[CB];
    const LBBNN_CONST_AS StageArgs& a = kernarg_as<StageBatch>()->l[blockIdx.z];
    if (blockIdx.x * CB >= a.I) return;                                             // grid sized for the widest layer
    const int path = a.path_lo + blockIdx.y;
    const int H = a.tr.hidden, tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const bool c0 = lane < H, c1 = lane + 64 < H;
    const bool rnvp = a.tr.kind == LBBNN_FLOW_RNVP;
    // Everything this workgroup reads from global memory is requested up front, so the kernel pays ONE memory latency
    // instead of one per phase (it is a chain of tiny dependent steps: measured 21 us with the loads where they are used):
    //  - the rows of the two head matrices of this wave's 16 outputs -> registers
    //  - the three middle matrices -> LDS (coalesced copy; rows of H = 75 floats are an odd number of banks apart, so
    //    the row walk below is conflict-free)
    constexpr int RPW = CB / 4;                                                  // rows per wave
    const int row0 = blockIdx.x * CB, last = a.I - 1;
    float ra0[RPW], ra1[RPW], rb0[RPW], rb1[RPW], bav[RPW], bbv[RPW], mv[RPW], zv[RPW];
    const float* __restrict__ mask = path ? a.tr.mask_kl : a.tr.mask_fwd;
    const float* __restrict__ zin = a.zin[path];
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
        const int i = min(row0 + wv + 4 * j, last);                               // clamped: rows past the end are not stored
        const float* __restrict__ ra = a.tr.w_a + (size_t)i * H;
        const float* __restrict__ rb = a.tr.w_b + (size_t)i * H;
        ra0[j] = c0 ? ra[lane] : 0.f; ra1[j] = c1 ? ra[lane + 64] : 0.f;
        rb0[j] = c0 ? rb[lane] : 0.f; rb1[j] = c1 ? rb[lane + 64] : 0.f;
        bav[j] = a.tr.b_a[i]; bbv[j] = a.tr.b_b[i]; mv[j] = mask[i]; zv[j] = zin[i];
    }
    PartialRegs pr;
    partial_a_load(a.nx, path, row0, a.I, pr);
    if (rnvp && use_lds) {
        const int HH = H * H;
        // ALL loads of the three matrices (up to 3 x 24 per thread) are requested before the first LDS write: as a plain
        // copy loop the compiler waited for each load (or small group) before issuing the next, and even one batch per
        // matrix cost three memory round trips instead of one.  (H > 78: further rounds of the same shape.)
        constexpr int NB = 24;
        for (int base = tid; base < HH; base += NTC * NB) {
            float v[3][NB];
#pragma unroll
            for (int l = 0; l < 3; ++l) {
                const float* __restrict__ wg = a.tr.w_mid[l];
#pragma unroll
                for (int j = 0; j < NB; ++j) { const int e = base + j * NTC; v[l][j] = e < HH ? wg[e] : 0.f; }
            }
#pragma unroll
            for (int l = 0; l < 3; ++l)
#pragma unroll
                for (int j = 0; j < NB; ++j) { const int e = base + j * NTC; if (e < HH) wl[l * HH + e] = v[l][j]; }
        }
    }
    float bm[3] = {0.f, 0.f, 0.f};                                               // middle-layer biases: requested with the rest
    if (rnvp && tid < H) {
#pragma unroll
        for (int l = 0; l < 3; ++l) bm[l] = a.tr.b_mid[l][tid];
    }
    const int nwg = (a.I + CB - 1) / CB;
    if (tid < H) {
        // h1: the partials of the previous launch's workgroups in a fixed order (4 loads in flight per trip)
        const float* src = a.Pin + (size_t)path * nwg * HMAX + tid;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int w = 0;
        for (; w + 16 <= nwg; w += 16) {                                          // 16 loads in flight per trip
            float v[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) v[q] = src[(size_t)(w + q) * HMAX];
#pragma unroll
            for (int q = 0; q < 16; q += 4) { a0 += v[q]; a1 += v[q + 1]; a2 += v[q + 2]; a3 += v[q + 3]; }
        }
        {
            float v[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) v[q] = (w + q < nwg) ? src[(size_t)(w + q) * HMAX] : 0.f;
#pragma unroll
            for (int q = 0; q < 16; q += 4) { a0 += v[q]; a1 += v[q + 1]; a2 += v[q + 2]; a3 += v[q + 3]; }
        }
        const float v = ((a0 + a1) + (a2 + a3)) + a.tr.b_in[tid];
        const float h1 = rnvp ? (v >= 0.f ? v : 0.1f * v) : tanhf(v);
        hs[0][tid] = h1;
        if (a.keep_chain && blockIdx.x == 0) a.h[path][tid] = h1;
    }
    __syncthreads();
    int cur = 0;
    if (rnvp) {
#pragma unroll
        for (int l = 0; l < 3; ++l) {
            // two threads per output unit (each half of the k range), four independent partial sums each: a single
            // accumulator made this a chain of H dependent LDS round trips
            const int j = tid & 127, half = tid >> 7;
            const int kmid = (H + 1) >> 1, k0 = half ? kmid : 0, k1 = half ? H : kmid;
            float s = 0.f;
            if (j < H) {
                const float* hv = hs[cur];
                auto dot = [&](const float* w) {
                    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
                    int k = k0;
                    for (; k + 4 <= k1; k += 4) { s0 += w[k] * hv[k]; s1 += w[k + 1] * hv[k + 1]; s2 += w[k + 2] * hv[k + 2]; s3 += w[k + 3] * hv[k + 3]; }
                    for (; k < k1; ++k) s0 += w[k] * hv[k];
                    return (s0 + s1) + (s2 + s3);
                };
                s = use_lds ? dot(wl + l * H * H + j * H) : dot(a.tr.w_mid[l] + (size_t)j * H);       // LDS | global walk
                if (half) ps[j] = s;
            }
            __syncthreads();
            if (tid < H) {
                s += ps[tid];
                s += bm[l];
                s = (l < 2) ? (s >= 0.f ? s : 0.1f * s) : s;                       // last LeakyReLU dropped (:185)
                hs[cur ^ 1][tid] = s;
                if (a.keep_chain && blockIdx.x == 0) a.h[path][(l + 1) * HMAX + tid] = s;
            }
            __syncthreads();
            cur ^= 1;
        }
    } else if (a.keep_chain && blockIdx.x == 0 && tid < H) a.h[path][3 * HMAX + tid] = hs[0][tid];
    // heads: one WAVE per output row i, lanes over the H columns (coalesced 4*H-byte rows of the two head matrices)
    const float y0 = c0 ? hs[cur][lane] : 0.f, y1 = c1 ? hs[cur][lane + 64] : 0.f;
    float ldw = 0.f;
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
        const int i = row0 + wv + 4 * j;
        const float sa = wave_sum(ra0[j] * y0 + ra1[j] * y1) + bav[j];
        const float sb = wave_sum(rb0[j] * y0 + rb1[j] * y1) + bbv[j];
        const float m = mv[j], z = zv[j];
        const float g = 1.f / (1.f + expf(-sb));                               // sigmoid: gate (:214) / sigma (:237)
        float zn;
        if (rnvp) zn = ((1.f - m) * z) * g + (1.f - g) * sa + m * z;           // :215
        else zn = m * z + (1.f - m) * (z * g + (1.f - g) * sa);                 // :238
        if (lane == 0) znew[wv + 4 * j] = (i <= last) ? zn : 0.f;
        if (i <= last) {
            if (lane == 0) {
                a.zout[path][i] = zn;
                if (a.zcopy[path]) a.zcopy[path][i] = zn;
            }
            ldw += (1.f - m) * logf(g);                                         // :219 / :241 (wave-uniform)
        }
    }
    if (lane == 0) scratch[wv] = (double)ldw;
    __syncthreads();
    if (tid == 0) a.ld_part[path][blockIdx.x] = (float)((scratch[0] + scratch[1]) + (scratch[2] + scratch[3]));
    partial_a_reduce(a.nx, path, znew, nwg, pr);                                // (znew complete: the barrier above)
}

struct FinishArgs {
    const float* ldz; const float* ldr; const float* lq0; const float* zr; const float* ldf;
    float* scal; int nblk, nblk_i, Tz, Tr, I, want_kl;
};

// scal[0] = log_det_q, [1] = log_q0, [2] = log_det_r, [3] = r_flow(z2)[-1], [4] = forward-draw log-det
struct FinishBatch { FinishArgs l[LBBNN_MAX_LAYERS]; };

__global__ __launch_bounds__(64) void dense_finish_kernel(const FinishBatch bt) {
    const LBBNN_CONST_AS FinishArgs& a = kernarg_as<FinishBatch>()->l[blockIdx.x];
    const int lane = threadIdx.x;
    auto total = [&](const float* v, int n) {                                  // fixed order: strided partials, wave tree
        float s = 0.f;
        for (int t = lane; t < n; t += 64) s += v[t];
        return wave_sum(s);
    };
    const float ldf = total(a.ldf, a.Tz * a.nblk);
    if (lane == 0) a.scal[4] = ldf;
    if (!a.want_kl) return;
    const float ldz = total(a.ldz, a.Tz * a.nblk), lq0 = total(a.lq0, a.nblk_i), ldr = total(a.ldr, a.Tr * a.nblk);
    if (lane == 0) { a.scal[0] = ldz; a.scal[1] = lq0; a.scal[2] = ldr; a.scal[3] = a.zr[a.I - 1]; }
}

bool transform_ok(const lbbnn_dense_transform_t& t, bool need_fwd, bool need_kl) {
    if (t.kind != LBBNN_FLOW_RNVP && t.kind != LBBNN_FLOW_MNF) return false;
    if (t.hidden <= 0 || t.hidden > HMAX) return false;
    if (!t.w_in || !t.b_in || !t.w_a || !t.b_a || !t.w_b || !t.b_b) return false;
    if (t.kind == LBBNN_FLOW_RNVP)
        for (int l = 0; l < 3; ++l) if (!t.w_mid[l] || !t.b_mid[l]) return false;
    if (need_fwd && !t.mask_fwd) return false;
    if (need_kl && !t.mask_kl) return false;
    return true;
}

}  // namespace

// workspace (floats): z[2][I] | h[2][HMAX] | lq0[nblk] | ld_fwd[T*nblk] | ld_z[T*nblk] | ld_r[T*nblk] | P[2 (ping-pong)][2][nblk][HMAX],
// T <= LBBNN_MAX_FLOW_T
extern "C" int64_t lbbnn_flow_dense_workspace(int I) {
    if (I <= 0) return 0;
    const int64_t nblk = (I + CB - 1) / CB;
    return 2 * (int64_t)I + 2 * HMAX + nblk + 3 * (int64_t)LBBNN_MAX_FLOW_T * nblk + 4 * nblk * HMAX + 64;
}

// kept intermediates (floats): ZF[Tz+1][I] | ZK[Tz+1][I] | ZR[Tr][I] | per (transform, path): h1 | h2 | h3 | head input
extern "C" int64_t lbbnn_flow_dense_save_size(int I, int Tz, int Tr) {
    if (I <= 0 || Tz < 0 || Tr < 0) return 0;
    return (int64_t)I * (2 * (Tz + 1) + Tr) + (int64_t)(Tz + Tr) * 2 * 4 * HMAX;
}

// K4 of n layers in 2 + 2*(Tz+Tr) launches (blockIdx.z = layer); the layers must agree on Tz, Tr and want_kl
// phase 0: everything; 1: the draws + the z flow (what the weight pass needs: z_fwd, z_kl); 2: the r flow + the scalars (what only the
// KL finalize needs) -- phases 1 and 2 of one forward may run on different streams, 2 after 1 (lbbnn_layers_dense_flows_phase)
static int dense_flows_impl(const lbbnn_dense_layer_t* L, int n, const uint64_t* rng, void* stream, int phase = 0) {
    if (!L) return LBBNN_E_NULL;
    if (n <= 0 || n > LBBNN_MAX_LAYERS) return LBBNN_E_SHAPE;
    const int Tz = L[0].Tz, Tr = L[0].Tr, want_kl = L[0].want_kl;
    int maxI = 0, maxH = 1;
    for (int k = 0; k < n; ++k) {
        const lbbnn_dense_layer_t& d = L[k];
        if (!d.q0_mean || !d.q0_log_var || !d.z_fwd || !d.scal || !d.work) return LBBNN_E_NULL;
        if (d.I <= 0 || d.Tz < 0 || d.Tz > LBBNN_MAX_FLOW_T || d.Tr < 0 || d.Tr > LBBNN_MAX_FLOW_T) return LBBNN_E_SHAPE;
        if (d.Tz != Tz || d.Tr != Tr || (d.want_kl != 0) != (want_kl != 0)) return LBBNN_E_SHAPE;
        if (d.want_kl && !d.z_kl) return LBBNN_E_NULL;
        if ((!d.eps_fwd || (d.want_kl && !d.eps_kl)) && !rng) return LBBNN_E_NOISE;
        if (d.draw_masks && !rng) return LBBNN_E_NOISE;
        if ((d.Tz && !d.zt) || (d.want_kl && d.Tr && !d.rt)) return LBBNN_E_NULL;
        for (int t = 0; t < d.Tz; ++t) { if (!transform_ok(d.zt[t], true, d.want_kl != 0)) return LBBNN_E_NULL; maxH = d.zt[t].hidden > maxH ? d.zt[t].hidden : maxH; }
        if (d.want_kl) for (int t = 0; t < d.Tr; ++t) { if (!transform_ok(d.rt[t], false, true)) return LBBNN_E_NULL; maxH = d.rt[t].hidden > maxH ? d.rt[t].hidden : maxH; }
        maxI = d.I > maxI ? d.I : maxI;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int npaths = want_kl ? 2 : 1;
    const int gblk = (maxI + CB - 1) / CB;
    const int S = Tz + (want_kl ? Tr : 0);                                        // launches after the init: one per transform
    struct Bufs { float *zbuf1, *h0, *h1, *lq0, *ldf, *ldz, *ldr, *ZF, *ZK, *ZR, *HS, *P[2]; int nblk, nblk_i; } B[LBBNN_MAX_LAYERS];
    // what launch s-1 (s = 0: the init) prepares for transform s: its rows' share of W_in (m z), for the paths transform s runs on
    auto next_of = [&](const lbbnn_dense_layer_t& d, const Bufs& b, int sidx) {
        NextA nx{};
        if (sidx >= S) return nx;
        const bool zph = sidx < Tz;
        const lbbnn_dense_transform_t& t = zph ? d.zt[sidx] : d.rt[sidx - Tz];
        nx.w_in = t.w_in; nx.mask[0] = t.mask_fwd; nx.mask[1] = t.mask_kl; nx.P = b.P[sidx & 1]; nx.hidden = t.hidden;
        nx.path_lo = zph ? 0 : 1; nx.npaths = zph ? npaths : 1;
        return nx;
    };
    InitBatch ib{};
    for (int k = 0; k < n; ++k) {
        const lbbnn_dense_layer_t& d = L[k];
        Bufs& b = B[k];
        b.nblk = (d.I + CB - 1) / CB; b.nblk_i = b.nblk;
        b.zbuf1 = d.work + d.I;                  // (work[0..I) was the path-0 scratch row of an earlier version)
        b.h0 = d.work + 2 * (size_t)d.I; b.h1 = b.h0 + HMAX; b.lq0 = b.h1 + HMAX;
        b.ldf = b.lq0 + b.nblk_i;
        b.ldz = b.ldf + (size_t)LBBNN_MAX_FLOW_T * b.nblk;
        b.ldr = b.ldz + (size_t)LBBNN_MAX_FLOW_T * b.nblk;
        b.P[0] = b.ldr + (size_t)LBBNN_MAX_FLOW_T * b.nblk; b.P[1] = b.P[0] + (size_t)2 * b.nblk * HMAX;
        b.ZF = d.save; b.ZK = d.save ? b.ZF + (size_t)(Tz + 1) * d.I : nullptr;
        b.ZR = d.save ? b.ZK + (size_t)(Tz + 1) * d.I : nullptr;
        b.HS = d.save ? b.ZR + (size_t)d.Tr * d.I : nullptr;
        InitArgs& ia = ib.l[k];
        ia.q0_mean = d.q0_mean; ia.q0_log_var = d.q0_log_var; ia.eps[0] = d.eps_fwd; ia.eps[1] = d.eps_kl; ia.rng = rng;
        ia.z[0] = d.save ? b.ZF : d.z_fwd; ia.z[1] = d.save ? b.ZK : b.zbuf1; ia.lq0_part = b.lq0; ia.I = d.I; ia.layer = d.layer_id & 63u; ia.npaths = npaths;
        ia.nx = next_of(d, b, 0);
        ia.draw = d.draw_masks ? 1 : 0;
        ia.Tm[0] = Tz; ia.Tm[1] = want_kl ? Tz : 0; ia.Tm[2] = want_kl ? Tr : 0;
        ia.nx_word = Tz > 0 ? 1 : 2;
        if (ia.draw) {
            for (int t = 0; t < Tz; ++t) { ia.mk[0][t] = const_cast<float*>(d.zt[t].mask_fwd); ia.mk[1][t] = const_cast<float*>(d.zt[t].mask_kl); }
            for (int t = 0; t < ia.Tm[2]; ++t) ia.mk[2][t] = const_cast<float*>(d.rt[t].mask_kl);
        }
    }
    if (phase != 2) hipLaunchKernelGGL(dense_init_kernel, dim3(gblk, npaths, n), dim3(NTC), 0, s, ib);

    for (int t = 0; t < S; ++t) {
        const bool zphase = t < Tz;
        if ((phase == 1 && !zphase) || (phase == 2 && zphase)) continue;
        StageBatch sb{};
        for (int k = 0; k < n; ++k) {
            const lbbnn_dense_layer_t& d = L[k];
            const Bufs& b = B[k];
            StageArgs& sa = sb.l[k];
            sa.tr = zphase ? d.zt[t] : d.rt[t - Tz];
            if (d.save) {
                const int tr_ = zphase ? t : t - Tz;
                sa.zin[0] = b.ZF + (size_t)t * d.I; sa.zout[0] = b.ZF + (size_t)(t + 1) * d.I;       // path 0: z phase only
                sa.zin[1] = zphase ? b.ZK + (size_t)t * d.I : (tr_ == 0 ? b.ZK + (size_t)Tz * d.I : b.ZR + (size_t)(tr_ - 1) * d.I);
                sa.zout[1] = zphase ? b.ZK + (size_t)(t + 1) * d.I : b.ZR + (size_t)tr_ * d.I;
                sa.h[0] = b.HS + ((size_t)t * 2 + 0) * 4 * HMAX; sa.h[1] = b.HS + ((size_t)t * 2 + 1) * 4 * HMAX;
                sa.keep_chain = 1;
                sa.zcopy[0] = (zphase && t == Tz - 1) ? d.z_fwd : nullptr;
            } else {
                sa.zin[0] = sa.zout[0] = d.z_fwd; sa.zin[1] = sa.zout[1] = b.zbuf1; sa.h[0] = b.h0; sa.h[1] = b.h1;
                sa.keep_chain = 0;
                sa.zcopy[0] = nullptr;
            }
            sa.zcopy[1] = (zphase && want_kl && t == Tz - 1) ? d.z_kl : nullptr;
            sa.ld_part[0] = b.ldf + (size_t)t * b.nblk;
            sa.ld_part[1] = zphase ? b.ldz + (size_t)t * b.nblk : b.ldr + (size_t)(t - Tz) * b.nblk;
            sa.I = d.I;
            sa.Pin = b.P[t & 1];
            sa.nx = next_of(d, b, t + 1);
            sa.path_lo = zphase ? 0 : 1;
            sa.npaths = zphase ? npaths : 1;
        }
        const int np = zphase ? npaths : 1;
        // the three middle matrices of a transform in LDS: 66 KB for the reference's H = 75, 117 KB for H = 100
        const size_t wl_bytes = (size_t)3 * maxH * maxH * sizeof(float);
        int use_lds = wl_bytes <= 144 * 1024 ? 1 : 0;
        static size_t raised = 0;
        if (use_lds && wl_bytes > 48 * 1024 && wl_bytes > raised) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(dense_stage_c_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)wl_bytes) == hipSuccess) raised = wl_bytes;
            else use_lds = 0;
        }
        hipLaunchKernelGGL(dense_stage_c_kernel, dim3(gblk, np, n), dim3(NTC), use_lds ? wl_bytes : 0, s, sb, use_lds);
    }
    FinishBatch fb{};
    for (int k = 0; k < n; ++k) {
        const lbbnn_dense_layer_t& d = L[k];
        const Bufs& b = B[k];
        if (phase != 2) {
            if (want_kl && Tz == 0) (void)hipMemcpyAsync(d.z_kl, d.save ? b.ZK : b.zbuf1, (size_t)d.I * sizeof(float), hipMemcpyDeviceToDevice, s);
            if (d.save && Tz == 0) (void)hipMemcpyAsync(d.z_fwd, b.ZF, (size_t)d.I * sizeof(float), hipMemcpyDeviceToDevice, s);
        }
        FinishArgs& fa = fb.l[k];
        fa.ldz = b.ldz; fa.ldr = b.ldr; fa.lq0 = b.lq0;
        fa.zr = !d.save ? b.zbuf1 : ((want_kl && Tr > 0) ? b.ZR + (size_t)(Tr - 1) * d.I : b.ZK + (size_t)Tz * d.I); fa.ldf = b.ldf; fa.scal = d.scal;
        fa.nblk = b.nblk; fa.nblk_i = b.nblk_i; fa.Tz = Tz; fa.Tr = want_kl ? Tr : 0; fa.I = d.I; fa.want_kl = want_kl;
    }
    if (phase != 1) hipLaunchKernelGGL(dense_finish_kernel, dim3(n), dim3(64), 0, s, fb);
    return (int)hipGetLastError();
}

extern "C" int lbbnn_layers_dense_flows(const lbbnn_dense_layer_t* layers, int n, const uint64_t* rng, void* stream) {
    return dense_flows_impl(layers, n, rng, stream);
}

extern "C" int lbbnn_layers_dense_flows_phase(const lbbnn_dense_layer_t* layers, int n, const uint64_t* rng, int phase, void* stream) {
    if (phase < 0 || phase > 2) return LBBNN_E_FLAGS;
    return dense_flows_impl(layers, n, rng, stream, phase);
}

extern "C" int lbbnn_mnf_flow_dense(const float* q0_mean, const float* q0_log_var,
                                    const lbbnn_dense_transform_t* zt, int Tz,
                                    const lbbnn_dense_transform_t* rt, int Tr,
                                    const float* eps_fwd, const float* eps_kl,
                                    const uint64_t* rng, uint32_t layer_id,
                                    float* z_fwd, float* z_kl, float* scal, float* work,
                                    int I, int want_kl, void* stream) {
    lbbnn_dense_layer_t d{};
    d.q0_mean = q0_mean; d.q0_log_var = q0_log_var; d.zt = zt; d.rt = rt; d.Tz = Tz; d.Tr = Tr;
    d.eps_fwd = eps_fwd; d.eps_kl = eps_kl; d.layer_id = layer_id; d.z_fwd = z_fwd; d.z_kl = z_kl; d.scal = scal; d.work = work;
    d.I = I; d.want_kl = want_kl; d.save = nullptr;
    return dense_flows_impl(&d, 1, rng, stream);
}
