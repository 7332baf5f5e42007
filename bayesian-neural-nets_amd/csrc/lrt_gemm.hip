// K2 -- the two activation-moment GEMMs of the local-reparameterisation layer, fused with the
// sampling epilogue, on gfx950 matrix cores (fp32-exact path: v_mfma_f32_16x16x4_f32).
//
//   mean[b,o] = sum_k x[b,k]   * e_w[o,k]   + bias_mean[o]        (torch.mm, LBBNN-GP-MF-LRT.py:172)
//   var [b,o] = sum_k x[b,k]^2 * var_w[o,k] (*var_scale[o]) + bias_var[o]      (…LRT.py:173)
//   out [b,o] = mean + sqrt(var) * eps[b,o]  (+ReLU)                           (…LRT.py:174-175)
//
// Design (MI355X-first, not a port of anything):
//  * One LDS image of the x tile feeds BOTH products: x^2 is formed in registers from the same
//    fragment, so the variance GEMM costs no extra HBM or LDS traffic.
//  * Orientation: the MFMA "A" operand is the weight tile (16 output features), "B" is the x tile
//    (16 batch rows).  The 16x16 accumulator then holds, per lane, 4 CONSECUTIVE output features of
//    one batch row, so the epilogue reads eps / bias and writes `out` as float4 (one 16-B access
//    per lane instead of four 4-B ones).
//  * K permutation: lane quarter q reads one float4 at k = 16c + 4q .. +3 from the [row][k] LDS
//    image and feeds element j to MFMA j; MFMA j therefore contracts k = 16c + 4q + j over q.
//    Both operands use the same map, so the dot product is exact and every LDS read is a
//    ds_read_b128 of the natural k-contiguous layout (no transposed image needed).
//  * fp32 MFMA runs at 64 FLOP/clk/SIMD, so the kernel is MFMA-issue bound: per 16-deep K chunk a
//    wave issues 80 MFMAs (2560 cycles) against 12 ds_read_b128; global->LDS staging of the next
//    chunk is issued before the MFMAs and written after them (one barrier per chunk).
//  * Roofline for this path: fp32 matrix peak 157.3 TFLOP/s (MI355X_MICROARCH.md).
#include <cstdlib>
#include "lbbnn_device.h"
#include "lbbnn_internal.h"
#include "kl_piggy.h"
#include "gemm_common.h"
#include "../../include/lbbnn.h"

namespace {

using namespace lbbnn;
typedef float floatx4 __attribute__((ext_vector_type(4)));

constexpr int BK = 16;        // K extent of one LDS chunk (4 MFMA k-steps)
constexpr int LDS_LD = 20;    // floats per LDS row: 16 + 4 pad (80 B keeps ds_read_b128 16-B aligned)

struct GemmArgs {
    const float* x; const float* e_w; const float* var_w;
    const float* bias_mean; const float* bias_var; const float* var_scale;
    const float* eps; const uint64_t* rng;
    float* out;
    float* std_out;          // optional (B,O): sqrt(var) before ReLU, for the backward pass
    long long row_offset;
    int ldx, ld, ldo, B, I, O;
    uint32_t rng_stream;
    int relu, log_softmax;
    int kchunk;              // split-K (bf16x3 kernel only): workgroup z contracts k in [z*kchunk, (z+1)*kchunk), 0 = off
    long long split_stride;  // ... and writes its partial product to out + z*split_stride
    // lbbnn_lrt_gemm_combine (mean-only): out = comb_add + 2 * comb_x (.) (x . e_w^T), the input-gradient combination
    // dX = G_m.W_m + 2 x (.) (G_v.W_v) fused into the second product's epilogue; NULL: plain output
    const float* comb_x; const float* comb_add;
    int ld_cx, ld_ca;
    // lbbnn_lrt_gemm_finalize: fin.n > 0 => the grid has one extra row of workgroups (blockIdx.y == gridDim.y - 1) whose
    // first workgroup does the KL finalize of the network (kl_piggy.h) while the tiles are computed; the rest of the row exits
    FinalizePiggy fin;
    // lbbnn_lrt_gemm_members: gridDim.z = members; member m = blockIdx.z is the same product on x + m*x_ms, e_w + m*w_ms
    // (var_w shared), out + m*o_ms, its noise drawn at Philox offset rng[1] + m*m_adv -- an ensemble of forwards
    // (test_ensemble, LBBNN-GP-MF-MNF.py:286-294) in one launch, every member bit-identical to its own launch
    int single16;                // LBBNN_F_SINGLE16 (host-side dispatch only)
    int members;
    long long x_ms, w_ms, o_ms;
    unsigned long long m_adv, m_off;     // m_off: filled in by member_view()
};

__device__ __forceinline__ GemmArgs member_view(const GemmArgs& in) {
    GemmArgs a = in;
    a.m_off = 0;
    if (in.members > 1) {
        const long long m = blockIdx.z;
        a.x = in.x + m * in.x_ms; a.e_w = in.e_w + m * in.w_ms; a.out = in.out + m * in.o_ms;
        a.m_off = (unsigned long long)m * in.m_adv;
    }
    return a;
}

struct EpiCtx { bool ovec; uint64_t seed, offs; };

template <bool MEAN_ONLY>
__device__ __forceinline__ EpiCtx make_epi_ctx(const GemmArgs& a) {
    EpiCtx c;
    c.ovec = ((a.O & 3) == 0) && ((a.ldo & 3) == 0) && ((reinterpret_cast<uintptr_t>(a.out) & 15u) == 0) &&
             (!a.eps || (reinterpret_cast<uintptr_t>(a.eps) & 15u) == 0);
    c.seed = 0; c.offs = 0;
    if (!MEAN_ONLY && !a.eps) { c.seed = a.rng[0]; c.offs = a.rng[1] + a.m_off; }
    return c;
}

// Per-output-feature constants of out[.][o..o+3], loaded once per o-tile (float4 when aligned).
struct OConst { float bm[4], bv[4], vs[4]; };

__device__ __forceinline__ void load4_or_fill(const float* p, int o, int O, float fill, float out[4]) {
    if (!p) { out[0] = out[1] = out[2] = out[3] = fill; return; }
    if (o + 3 < O && ((reinterpret_cast<uintptr_t>(p + o) & 15u) == 0)) {
        const float4 t = *reinterpret_cast<const float4*>(p + o);
        out[0] = t.x; out[1] = t.y; out[2] = t.z; out[3] = t.w;
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) out[r] = (o + r < O) ? p[o + r] : fill;
    }
}

__device__ __forceinline__ OConst load_oconst(const GemmArgs& a, int o) {
    OConst c;
    load4_or_fill(a.bias_mean, o, a.O, 0.f, c.bm);
    load4_or_fill(a.bias_var, o, a.O, 0.f, c.bv);
    load4_or_fill(a.var_scale, o, a.O, 1.f, c.vs);
    return c;
}

// Local-reparameterisation epilogue for out[b][o..o+3] (LBBNN-GP-MF-LRT.py:172-175): bias, variance
// scale/bias, eps (explicit -- already loaded by the caller into e_in -- or Philox with counter (row_offset + b, o/4)),
// sqrt, optional ReLU; cx_in / ca_in: the preloaded operands of the mean-only combine form (dX = ca + 2 cx * product).
template <bool MEAN_ONLY>
__device__ __forceinline__ void epilogue4(const GemmArgs& a, const EpiCtx& c, const OConst& oc, int b, int o,
                                          const floatx4& am, const floatx4& av, const float* e_in, const float* cx_in,
                                          const float* ca_in, float res[4], float sd[4]) {
    float e[4] = {0.f, 0.f, 0.f, 0.f};
    if (!MEAN_ONLY) {
        if (a.eps) {
#pragma unroll
            for (int r = 0; r < 4; ++r) e[r] = e_in[r];
        } else {
            philox_normal4(c.seed, c.offs, a.rng_stream, (uint64_t)(a.row_offset + b), (uint32_t)(o >> 2), e);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float mean = am[r] + oc.bm[r];
        sd[r] = 0.f;
        if (!MEAN_ONLY) { sd[r] = sqrt_hw(av[r] * oc.vs[r] + oc.bv[r]); mean += sd[r] * e[r]; }
        res[r] = a.relu ? fmaxf(mean, 0.f) : mean;
    }
    if (MEAN_ONLY && a.comb_x) {
#pragma unroll
        for (int r = 0; r < 4; ++r) res[r] = ca_in[r] + 2.f * cx_in[r] * res[r];
    }
}

__device__ __forceinline__ void load4_rows(const float* p, bool vec, int o, int O, float out[4]) {
    if (vec) { const float4 t = *reinterpret_cast<const float4*>(p); out[0] = t.x; out[1] = t.y; out[2] = t.z; out[3] = t.w; }
    else {
#pragma unroll
        for (int r = 0; r < 4; ++r) out[r] = (o + r < O) ? p[r] : 0.f;
    }
}

__device__ __forceinline__ void store4_rows(float* p, bool vec, int o, int O, const float v[4]) {
    if (vec) *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    else {
#pragma unroll
        for (int r = 0; r < 4; ++r) if (o + r < O) p[r] = v[r];
    }
}

// Epilogue of a wave's TO x TB accumulator tiles (lane: out[b][o .. o+3] of tile (i, j), o = o0 + 16 i + 4 q,
// b = brow0 + 16 j).  Two phases.  (1) EVERY load the epilogue needs -- the per-feature constants of the TO o-tiles, the
// explicit eps or the combine operands where the call has them -- is issued back to back, before the first store.  (2)
// noise, arithmetic and stores per tile, no load in between.  The former loop loaded each o-tile's constants inside the
// loop: `s_waitcnt vmcnt(0)` before their use also waited for the previous o-tile's STORES (vmcnt counts stores), five
// dependent round trips per workgroup at the point of the launch where nothing else is left to overlap them
// (tools/gemm_ksweep.py: 16.4 us of a launch did not depend on K).
template <int TO, int TB, bool MEAN_ONLY>
__device__ __forceinline__ void epilogue_tile(const GemmArgs& a, int o0, int q, int brow0,
                                              const floatx4 (&accm)[TO][TB], const floatx4 (&accv)[TO][TB]) {
    EpiCtx ec = make_epi_ctx<MEAN_ONLY>(a);
    OConst oc[TO];
#pragma unroll
    for (int i = 0; i < TO; ++i) {
        const int o = o0 + i * 16 + 4 * q;
        if (o < a.O) oc[i] = load_oconst(a, o);
    }
    const bool pre_eps = !MEAN_ONLY && a.eps != nullptr;
    const bool pre_comb = MEAN_ONLY && a.comb_x != nullptr;
    const bool comb_vec = pre_comb && ec.ovec && ((a.ld_cx | a.ld_ca) & 3) == 0 &&
                          ((reinterpret_cast<uintptr_t>(a.comb_x) | reinterpret_cast<uintptr_t>(a.comb_add)) & 15u) == 0;
    float pa[TO][TB][4], pb[TO][TB][4];       // explicit eps (pa) or the combine operands (pa = comb_x, pb = comb_add)
    if (pre_eps || pre_comb) {
#pragma unroll
        for (int i = 0; i < TO; ++i) {
            const int o = o0 + i * 16 + 4 * q;
#pragma unroll
            for (int j = 0; j < TB; ++j) {
                const int b = brow0 + j * 16;
                if (o >= a.O || b >= a.B) continue;
                if (pre_eps) load4_rows(a.eps + (size_t)b * a.O + o, ec.ovec, o, a.O, pa[i][j]);
                else {
                    load4_rows(a.comb_x + (size_t)b * a.ld_cx + o, comb_vec, o, a.O, pa[i][j]);
                    load4_rows(a.comb_add + (size_t)b * a.ld_ca + o, comb_vec, o, a.O, pb[i][j]);
                }
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < TO; ++i) {
        const int o = o0 + i * 16 + 4 * q;
        if (o >= a.O) continue;
#pragma unroll
        for (int j = 0; j < TB; ++j) {
            const int b = brow0 + j * 16;
            if (b >= a.B) continue;
            float res[4], sd[4];
            epilogue4<MEAN_ONLY>(a, ec, oc[i], b, o, accm[i][j], accv[i][j], pa[i][j], pa[i][j], pb[i][j], res, sd);
            store4_rows(a.out + (size_t)b * a.ldo + o, ec.ovec, o, a.O, res);
            if (!MEAN_ONLY && a.std_out) store4_rows(a.std_out + (size_t)b * a.O + o, ec.ovec, o, a.O, sd);
        }
    }
}

// TO x TB 16x16 tiles per wave (o x b), WB waves along b; all waves share the o extent.
//
// Staging is deliberately branch-free in the steady state: each thread owns NIT fixed (row, 4-float
// column) slots of the chunk image; its global pointers are computed ONCE (rows past B / O are clamped
// to the last valid row -- their accumulators are never stored) and advance by 16 floats per chunk.
// Only a K tail (I % 16 != 0) takes the guarded path, and only for the x rows (the weight operands
// are zero-padded to ld by the weight pass).
template <int TO, int TB, int WB, bool MEAN_ONLY, bool XVEC>
__global__ __launch_bounds__(WB * 64, 2) void lrt_gemm_f32_kernel(const GemmArgs a_in) {
    const GemmArgs a = member_view(a_in);
    constexpr int NT = WB * 64;
    constexpr int BN = TO * 16;          // output features per block
    constexpr int BM = TB * WB * 16;     // batch rows per block
    constexpr int NW = MEAN_ONLY ? 1 : 2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // [buf][ X: BM rows | Wm: BN rows | Wv: BN rows ] x LDS_LD
    constexpr int ROWS = BM + NW * BN;
    constexpr int BUF = ROWS * LDS_LD;

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int lr = lane & 15, q = lane >> 4;
    const int o0 = blockIdx.x * BN;
    const int b0 = blockIdx.y * BM;

    constexpr int SLOTS = ROWS * 4;
    constexpr int NIT = (SLOTS + NT - 1) / NT;
    const float* gp[NIT];     // this thread's source pointer per slot (chunk 0)
    int loff[NIT];            // LDS float offset of the slot inside a buffer
    int kcol[NIT];            // first k of the slot inside a chunk, or -1 for weight slots
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        int sl = tid + it * NT;
        if (sl >= SLOTS) sl -= SLOTS;              // surplus threads duplicate an early slot (same data, same address)
        const int row = sl >> 2, c4 = (sl & 3) << 2;
        loff[it] = row * LDS_LD + c4;
        if (row < BM) {
            const int bb = min(b0 + row, a.B - 1);
            gp[it] = a.x + (size_t)bb * a.ldx + c4;
            kcol[it] = c4;
        } else {
            const int wr = row - BM;
            const bool isv = (!MEAN_ONLY) && wr >= BN;
            const int oo = min(o0 + (isv ? wr - BN : wr), a.O - 1);
            gp[it] = (isv ? a.var_w : a.e_w) + (size_t)oo * a.ld + c4;
            kcol[it] = -1;
        }
    }
    float4 stage[NIT];
    // steady state: chunk c lies entirely inside I -> unconditional loads
    auto load_full = [&](int c) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const float* p = gp[it] + c * BK;
            if (XVEC) stage[it] = *reinterpret_cast<const float4*>(p);
            else if (kcol[it] < 0) stage[it] = *reinterpret_cast<const float4*>(p);
            else stage[it] = make_float4(p[0], p[1], p[2], p[3]);
        }
    };
    // the (single) K-tail chunk: x elements at k >= I read as zero; weight operands are zero-padded
    auto load_tail = [&](int c) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const float* p = gp[it] + c * BK;
            if (kcol[it] < 0) {
                stage[it] = *reinterpret_cast<const float4*>(p);
            } else {
                const int k = c * BK + kcol[it];
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (k + 0 < a.I) v.x = p[0];
                if (k + 1 < a.I) v.y = p[1];
                if (k + 2 < a.I) v.z = p[2];
                if (k + 3 < a.I) v.w = p[3];
                stage[it] = v;
            }
        }
    };
    auto store_chunk = [&](float* buf) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) *reinterpret_cast<float4*>(buf + loff[it]) = stage[it];
    };

    floatx4 accm[TO][TB], accv[TO][TB];
#pragma unroll
    for (int i = 0; i < TO; ++i)
#pragma unroll
        for (int j = 0; j < TB; ++j) { accm[i][j] = floatx4{0.f, 0.f, 0.f, 0.f}; accv[i][j] = floatx4{0.f, 0.f, 0.f, 0.f}; }

    const int nfull = a.I / BK;                       // chunks entirely inside I
    const int nchunks = (a.I + BK - 1) / BK;          // nfull (+1 guarded tail chunk)
    if (nfull > 0) load_full(0); else load_tail(0);
    store_chunk(smem);
    __syncthreads();

    // fragment read offsets (floats) inside a buffer
    const int xoff = (wv * TB * 16 + lr) * LDS_LD + 4 * q;
    const int woff = (BM + lr) * LDS_LD + 4 * q;

    // One chunk of MFMA work from the LDS image `cur`.  All 2*TO + TB fragment reads are issued back
    // to back (one LDS latency per chunk, counted lgkmcnt waits); the MFMAs then run k-outermost so
    // consecutive MFMAs never share an accumulator.
    auto compute = [&](const float* cur) {
        float xf[TB][4], xs[TB][4], wm[TO][4], wq[TO][4];
#pragma unroll
        for (int j = 0; j < TB; ++j) {
            const float4 t = *reinterpret_cast<const float4*>(cur + xoff + j * 16 * LDS_LD);
            xf[j][0] = t.x; xf[j][1] = t.y; xf[j][2] = t.z; xf[j][3] = t.w;
        }
#pragma unroll
        for (int i = 0; i < TO; ++i) {
            const float4 t = *reinterpret_cast<const float4*>(cur + woff + i * 16 * LDS_LD);
            wm[i][0] = t.x; wm[i][1] = t.y; wm[i][2] = t.z; wm[i][3] = t.w;
            if (!MEAN_ONLY) {
                const float4 u = *reinterpret_cast<const float4*>(cur + woff + (BN + i * 16) * LDS_LD);
                wq[i][0] = u.x; wq[i][1] = u.y; wq[i][2] = u.z; wq[i][3] = u.w;
            }
        }
#pragma unroll
        for (int j = 0; j < TB; ++j)
#pragma unroll
            for (int k = 0; k < 4; ++k) xs[j][k] = xf[j][k] * xf[j][k];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
            for (int i = 0; i < TO; ++i) {
#pragma unroll
                for (int j = 0; j < TB; ++j) {
                    accm[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wm[i][k], xf[j][k], accm[i][j], 0, 0, 0);
                    if (!MEAN_ONLY)
                        accv[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[i][k], xs[j][k], accv[i][j], 0, 0, 0);
                }
            }
        }
    };

    // steady state: both this chunk and the next are full -> no predicates anywhere in the loop
    int c = 0;
    for (; c + 1 < nfull; ++c) {
        load_full(c + 1);                                  // global loads in flight under the MFMAs
        __builtin_amdgcn_sched_barrier(0);                 // keep them ABOVE the MFMAs (hipcc sinks them otherwise)
        compute(smem + (c & 1) * BUF);
        __builtin_amdgcn_sched_barrier(0);
        store_chunk(smem + ((c & 1) ^ 1) * BUF);
        __syncthreads();
    }
    // last full chunk (prefetching the guarded K-tail chunk if there is one), then the tail chunk
    for (; c < nchunks; ++c) {
        const bool more = (c + 1) < nchunks;
        if (more) load_tail(c + 1);
        __builtin_amdgcn_sched_barrier(0);
        compute(smem + (c & 1) * BUF);
        __builtin_amdgcn_sched_barrier(0);
        if (more) store_chunk(smem + ((c & 1) ^ 1) * BUF);
        __syncthreads();
    }

    // ---- epilogue: lane holds out[b][o .. o+3] for each (i, j) tile
    epilogue_tile<TO, TB, MEAN_ONLY>(a, o0, q, b0 + wv * TB * 16 + lr, accm, accv);
}

// ------------------------------------------------------------------------------------------------
// LDS-DMA variant of the kernel above (the fast path: x 16-B aligned, I % 16 == 0).
//
// Measured on MI355X (tools/mfma_peak.hip): with this loop skeleton the register-staged
// `ds_write_b128` pass alone costs 12 % of MFMA throughput (146 -> 129 TFLOP/s) -- a store moves its
// address and data VGPRs to the LDS at ~13 cycles per wave-instruction and serialises every wave's
// write -> lgkmcnt(0) -> barrier -> read chain.  `global_load_lds_dwordx4` writes the chunk image
// straight into LDS: no staging VGPRs, no ds_write, and the loads are counted by vmcnt.
//
// LDS-DMA writes wave-uniform base + lane*16 B, so the image is lane-linear: rows of 64 B (16 floats,
// NO padding), one wave-instruction = 16 rows.  Unswizzled, a ds_read_b128 fragment read (lane ->
// row lr, 16-B slot q) would be 4-way bank conflicted; the fix goes on the per-lane SOURCE address
// (cdna guide rule 21): LDS slot q' of row r holds global slot q' ^ F[(r>>2)&3], F = {0,2,3,1}, and
// readers apply the same involution.  With that map each of ds_read_b128's four 16-lane groups
// touches 16 distinct 16-B slots of the 256-B bank row: conflict-free.
constexpr int DROW = 16;                     // floats per LDS row in the DMA image

__device__ __forceinline__ int swz(int rowgrp) { return (0x78 >> (2 * (rowgrp & 3))) & 3; }   // F = {0,2,3,1}

template <int TO, int TB, int WB, bool MEAN_ONLY>
__global__ __launch_bounds__(WB * 64, 2) void lrt_gemm_f32_dma_kernel(const GemmArgs a_in) {
    const GemmArgs a = member_view(a_in);
    constexpr int BN = TO * 16, BM = TB * WB * 16;
    constexpr int NW = MEAN_ONLY ? 1 : 2;
    constexpr int ROWS = BM + NW * BN;
    constexpr int NG = ROWS / 16;                        // 16-row DMA groups per chunk image
    constexpr int NPW = (NG + WB - 1) / WB;              // DMA instructions per wave per chunk
    constexpr int BUF = ROWS * DROW;                     // floats per buffer
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, q = lane >> 4;
    if (a.fin.n > 0 && blockIdx.y == gridDim.y - 1) {                          // the finalize row (uniform per workgroup)
        if (blockIdx.x == 0) kl_finalize_piggy<WB>(kernarg_as<GemmArgs>()->fin, smem);
        return;
    }
    int tox, tby;
    tile_of_block(tox, tby, a.fin.n > 0 ? 1 : 0);
    const int o0 = tox * BN;
    const int b0 = tby * BM;

    // ---- per-lane DMA sources: group g = wv + WB*u covers image rows [16g, 16g+16); this lane feeds
    // row 16g + (lane>>2), LDS slot lane&3, i.e. global slot (lane&3) ^ F[(lane>>4)&3].
    const int srow = lane >> 2;
    const int scol = ((lane & 3) ^ swz(lane >> 4)) << 2;
    const float* gp[NPW];
#pragma unroll
    for (int u = 0; u < NPW; ++u) {
        const int g = wv + WB * u;
        const int row = 16 * g + srow;
        if (row < BM) {
            gp[u] = a.x + (size_t)min(b0 + row, a.B - 1) * a.ldx + scol;
        } else {
            const int wr = row - BM;
            const bool isv = (!MEAN_ONLY) && wr >= BN;
            const int oo = min(o0 + (isv ? wr - BN : wr), a.O - 1);
            gp[u] = (isv ? a.var_w : a.e_w) + (size_t)oo * a.ld + scol;
        }
    }
    auto dma_chunk = [&](int c, float* buf) {
#pragma unroll
        for (int u = 0; u < NPW; ++u) {
            const int g = wv + WB * u;                   // wave-uniform
            if (g < NG)
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)(gp[u] + c * BK),
                    (__attribute__((address_space(3))) void*)(buf + g * 16 * DROW), 16, 0, 0);
        }
    };

    floatx4 accm[TO][TB], accv[TO][TB];
#pragma unroll
    for (int i = 0; i < TO; ++i)
#pragma unroll
        for (int j = 0; j < TB; ++j) { accm[i][j] = floatx4{0.f, 0.f, 0.f, 0.f}; accv[i][j] = floatx4{0.f, 0.f, 0.f, 0.f}; }

    // fragment read offsets: row (R + lr), slot q ^ F[(lr>>2)&3]   (R is a multiple of 16)
    const int fcol = (q ^ swz(lr >> 2)) << 2;
    const int xoff = (wv * TB * 16 + lr) * DROW + fcol;
    const int woff = (BM + lr) * DROW + fcol;

    float xf[TB][4], xs[TB][4], wm[TO][4], wq[TO][4];
    auto read_frags = [&](const float* cur) {
#pragma unroll
        for (int j = 0; j < TB; ++j) {
            const float4 t = *reinterpret_cast<const float4*>(cur + xoff + j * 16 * DROW);
            xf[j][0] = t.x; xf[j][1] = t.y; xf[j][2] = t.z; xf[j][3] = t.w;
        }
#pragma unroll
        for (int i = 0; i < TO; ++i) {
            const float4 t = *reinterpret_cast<const float4*>(cur + woff + i * 16 * DROW);
            wm[i][0] = t.x; wm[i][1] = t.y; wm[i][2] = t.z; wm[i][3] = t.w;
            if (!MEAN_ONLY) {
                const float4 u = *reinterpret_cast<const float4*>(cur + woff + (BN + i * 16) * DROW);
                wq[i][0] = u.x; wq[i][1] = u.y; wq[i][2] = u.z; wq[i][3] = u.w;
            }
        }
    };
    auto mfmas = [&]() {
#pragma unroll
        for (int j = 0; j < TB; ++j)
#pragma unroll
            for (int k = 0; k < 4; ++k) xs[j][k] = xf[j][k] * xf[j][k];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
            for (int i = 0; i < TO; ++i) {
#pragma unroll
                for (int j = 0; j < TB; ++j) {
                    accm[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wm[i][k], xf[j][k], accm[i][j], 0, 0, 0);
                    if (!MEAN_ONLY)
                        accv[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[i][k], xs[j][k], accv[i][j], 0, 0, 0);
                }
            }
        }
    };

    const int nchunks = a.I / BK;                        // host guarantees I % 16 == 0
    dma_chunk(0, smem);
    __syncthreads();                                     // vmcnt(0) + barrier: chunk 0 visible to all waves
    for (int c = 0; c < nchunks; ++c) {
        // Order matters: hipcc drains every in-flight LDS-DMA (vmcnt(0)) before a ds_read it cannot
        // disambiguate from the DMA destination, so the fragment reads of `cur` are ISSUED first, then
        // the DMA of the next chunk (which lands under the 80 MFMAs), then the MFMAs.
        read_frags(smem + (c & 1) * BUF);
        __builtin_amdgcn_sched_barrier(0);
        if (c + 1 < nchunks) dma_chunk(c + 1, smem + ((c & 1) ^ 1) * BUF);
        __builtin_amdgcn_sched_barrier(0);
        mfmas();
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();      // my DMAs landed (vmcnt(0)) + every wave done reading `cur` and done landing `nxt`
    }

    // ---- epilogue (identical to the register-staged kernel)
    epilogue_tile<TO, TB, MEAN_ONLY>(a, o0, q, b0 + wv * TB * 16 + lr, accm, accv);
}

// ------------------------------------------------------------------------------------------------
// Split-precision variant (LBBNN_F_SPLIT16): the same two GEMMs on v_mfma_f32_16x16x32_bf16 (16 cycles
// for 32 k, vs 2 x 4 x 32 cycles on the fp32 path = 8x fewer matrix-core cycles) without giving up the
// 1e-4 contract:
//   mean = x.e_w^T with x = xh + xl, e_w = wh + wl (bf16 each):  xh.wh + xh.wl + xl.wh   (fp32 accumulate;
//          the dropped xl.wl term is ~2^-16 relative; measured ~4e-6 relative on the layer output)
//   var  = x^2 . var_w^T the same way: s = x^2 = sh + sl, v = vh + vl:  sh.vh + sl.vh + sh.vl.  (A single
//          bf16 product was measured first: its 2^-9 roundings do not average out under the max-norm --
//          worst element 1.2e-4 at every K -- so the variance gets the 3-term split too; the kernel is
//          operand-delivery bound, MFMA utilisation ~27 %, so the two extra MFMAs per tile are hidden.)
// x stays fp32 in HBM and in LDS (the previous layer's output as it is); each lane splits its fragment in
// registers: xh = bits & 0xFFFF0000 (truncation, so xl = x - xh is exact), xl -> bf16 RNE; same for x^2
// (v_and, v_sub, v_perm, v_cvt_pk, v_mul: ~56 VALU per 8 values, hidden under the 60 MFMAs of the step).
// Weight operands come from lbbnn_weight_pass(LBBNN_F_SPLIT16) in the split layout of lbbnn_device.h: per row and
// 32-k chunk one 128-B line of units (hi k0-7 | lo k0-7 | hi k8-15 | lo k8-15 | ...), e_w and var_w each.
//
// K step = 32.  LDS image per step: X 128 rows x 128 B (fp32) | E 80 rows x 128 B | V 80 rows x 128 B, all filled by
// LDS-DMA in 1-KiB pieces of 8 rows x 128 B (delivery-only build: 48.9 us against 58.0 us with 16 x 64-B pieces
// from separate hi / lo planes).  Conflict-free swizzle on the source side, the same for all three regions:
// 16-B slot s of row r is stored at slot s ^ G(r & 15), G(r) = ((r>>1)&3)*2 + ((r>>3)&1)  (a ds_read_b128 group of
// 16 lanes -- the hardware's, e.g. lanes {0-3, 12-15, 20-27} -- reads units 2q / 2q+1 of 16 rows conflict-free;
// SQ_LDS_BANK_CONFLICT = 0.  A first weight layout [hi 64 B | lo 64 B] read as units q / 4+q showed 2-way conflicts.)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float floatx2 __attribute__((ext_vector_type(2)));
constexpr int BKS = 32;

__device__ __forceinline__ int swzx(int r) { return (((r >> 1) & 3) << 1) | ((r >> 3) & 1); }

// NP = 3: the split products above.  NP = 1 (LBBNN_F_SINGLE16): ONE bf16 product per moment -- xh.wh and sh.vh with RNE
// operands, the lo units of the operand lines unread -- the plain "bf16 MFMA" arithmetic BASELINE configs[1] names: a
// third of the matrix work and of the conversions, measured error 2e-3 relative on the mean GEMM (outside the 1e-4
// contract; its own tolerance in the tests), so never the default.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
// (the body is a __device__ function: with the LDS-DMA builtin fed from an array element directly inside a __global__ template,
// hipcc 7.2's host pass silently drops the kernel's host stub -- undefined symbol at load time)
template <int TO, int TB, int WB, bool MEAN_ONLY, int NP, bool F16>
__device__ __forceinline__ void lrt_gemm_bf16x3_body(const GemmArgs& a_in) {
    static_assert(!F16 || NP == 1, "fp16 operands exist in the single-product form only");
    const GemmArgs a = member_view(a_in);
    constexpr int BN = TO * 16, BM = TB * WB * 16;
    constexpr int NWR = MEAN_ONLY ? 1 : 2;               // weight regions: e_w (, var_w); a row = [hi 64 B | lo 64 B]
    constexpr int XB = BM * 128;                         // bytes of the X region
    constexpr int WRB = BN * 128;                        // bytes of one weight region
    constexpr int BUFB = XB + NWR * WRB;                 // bytes per buffer
    constexpr int NGX = BM / 8, NGW = BN / 8;            // 1-KiB DMA pieces: 8 rows x 128 B, for x and for the weights
    constexpr int NG = NGX + NWR * NGW;
    constexpr int NPW = (NG + WB - 1) / WB;
    extern __shared__ __attribute__((aligned(16))) char smc[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, q = lane >> 4;
    if (a.fin.n > 0 && blockIdx.y == gridDim.y - 1) {                          // the finalize row (uniform per workgroup)
        if (blockIdx.x == 0) kl_finalize_piggy<WB>(kernarg_as<GemmArgs>()->fin, reinterpret_cast<float*>(smc));
        return;
    }
    int tox, tby;
    tile_of_block(tox, tby, a.fin.n > 0 ? 1 : 0);
    const int o0 = tox * BN;
    const int b0 = tby * BM;
    // split-K: this workgroup's k range (kbeg is a multiple of 32, so every alignment below is unchanged)
    const int kbeg = a.kchunk ? (int)blockIdx.z * a.kchunk : 0;
    const int Iloc = a.kchunk ? min(a.I - kbeg, a.kchunk) : a.I;
    const char* const eb = reinterpret_cast<const char*>(a.e_w);       // rows of 4*ld bytes: per 32-k chunk [hi | lo]
    const char* const vb = reinterpret_cast<const char*>(a.var_w);
    // 16 B of zeros for x lanes past I in the K tail: the zero-filled hi tail of row 0's last chunk
    const char* const zsrc = eb + split_hi_index(0, a.I, a.ld) * 2;

    // LDS-DMA through buffer descriptors (buffer_load_dwordx4 ... offen lds): the per-lane byte offset of a piece is
    // computed once, the K step advances a SCALAR offset (x and weights both move 128 B per step), and an x lane past
    // I in the K tail is pointed out of range -- the descriptor's bounds check returns zeros.  (The flat-pointer form
    // needed a 64-bit VALU add per piece and step plus a zero-source redirect.)
    const unsigned xbytes = (unsigned)min((size_t)0x7FFFFFF0u, ((size_t)(a.B - 1) * a.ldx + a.I) * 4);
    const unsigned wbytes = (unsigned)min((size_t)0x7FFFFFF0u, (size_t)a.O * a.ld * 4);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t re = __builtin_amdgcn_make_buffer_rsrc((void*)a.e_w, 0, (int)wbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void*)(MEAN_ONLY ? a.e_w : a.var_w), 0, (int)wbytes, 0x00020000);
    (void)eb; (void)vb; (void)zsrc;
    int gv[NPW];              // per-lane byte offset of DMA piece wv + WB*u at this workgroup's first K step
    int kx[NPW];              // first k of this lane's x slot (for the tail), or -1
#pragma unroll
    for (int u = 0; u < NPW; ++u) {
        const int g = wv + WB * u;
        if (g < NGX) {
            const int row = 8 * g + (lane >> 3);
            const int slot = (lane & 7) ^ swzx(row & 15);
            gv[u] = (int)(((size_t)min(b0 + row, a.B - 1) * a.ldx + kbeg) * 4) + 16 * slot;
            kx[u] = 4 * slot;
        } else {
            const int gw = g - NGX, row = 8 * (gw % NGW) + (lane >> 3);
            const int slot = (lane & 7) ^ swzx(row & 15);        // units 2g = hi, 2g+1 = lo of k group g; same swizzle as the x rows
            gv[u] = (int)((size_t)min(o0 + row, a.O - 1) * a.ld * 4) + (kbeg >> 5) * 128 + 16 * slot;
            kx[u] = -1;
        }
    }
    const int nsteps = (Iloc + BKS - 1) / BKS;
    const bool has_tail = (Iloc % BKS) != 0;
    // The per-lane offsets of the K-tail step (x lanes past I pointed out of range) live in registers of their own, gvt[], and
    // the step picks the set by a SCALAR branch: an LDS-DMA reads its address VGPR only when the fill path takes the
    // instruction, which under load is 100+ cycles after its issue, and until then the hardware holds back any later write
    // to that register.  The former form computed `tail ? out_of_range : gv[u]` into ONE temporary per piece, so every piece
    // of a wave waited for the previous one to be taken (tools/gemm_stamps.py: ~125 cycles per piece, 1 100 per step).
    int gvt[NPW];
#pragma unroll
    for (int u = 0; u < NPW; ++u)
        gvt[u] = (kx[u] >= 0 && (nsteps - 1) * BKS + kx[u] >= Iloc) ? 0x7FFFFFF0 : gv[u];
    auto dma_pieces = [&](int c, char* buf, const int (&va)[NPW]) {
#pragma unroll
        for (int u = 0; u < NPW; ++u) {
            const int g = wv + WB * u;                   // wave-uniform
            if (g < NG) {
                const int loff = g < NGX ? g * 1024 : XB + (g - NGX) * 1024;
                auto* dst = (__attribute__((address_space(3))) void*)(buf + loff);
                if (g < NGX)                 __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, dst, 16, va[u], c * 128, 0, 0);
                else if (g - NGX < NGW)      __builtin_amdgcn_raw_ptr_buffer_load_lds(re, dst, 16, va[u], c * 128, 0, 0);
                else                         __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, dst, 16, va[u], c * 128, 0, 0);
            }
        }
    };
    auto dma_step = [&](int c, char* buf) {
        if (has_tail && c == nsteps - 1) dma_pieces(c, buf, gvt);
        else dma_pieces(c, buf, gv);
    };

    floatx4 accm[TO][TB], accv[TO][TB];
#pragma unroll
    for (int i = 0; i < TO; ++i)
#pragma unroll
        for (int j = 0; j < TB; ++j) { accm[i][j] = floatx4{0.f, 0.f, 0.f, 0.f}; accv[i][j] = floatx4{0.f, 0.f, 0.f, 0.f}; }

    // fragment read byte offsets
    const int gx = swzx(lr);
    const int xo0 = (wv * TB * 16 + lr) * 128 + 16 * ((2 * q) ^ gx);
    const int xo1 = (wv * TB * 16 + lr) * 128 + 16 * ((2 * q + 1) ^ gx);
    const int woh = XB + lr * 128 + 16 * ((2 * q) ^ gx);     // weight row lr: hi of this lane's k group = unit 2q, lo = unit 2q+1
    const int wol = XB + lr * 128 + 16 * ((2 * q + 1) ^ gx);

    float4 xr[TB][2];
    uint4 wh[TO], wl[TO], wvh[TO], wvl[TO];
    auto read_frags = [&](const char* cur) {
#pragma unroll
        for (int j = 0; j < TB; ++j) {
            xr[j][0] = *reinterpret_cast<const float4*>(cur + xo0 + j * 16 * 128);
            xr[j][1] = *reinterpret_cast<const float4*>(cur + xo1 + j * 16 * 128);
        }
#pragma unroll
        for (int i = 0; i < TO; ++i) {
            wh[i] = *reinterpret_cast<const uint4*>(cur + woh + i * 16 * 128);
            if (NP == 3) wl[i] = *reinterpret_cast<const uint4*>(cur + wol + i * 16 * 128);
            if (!MEAN_ONLY) {
                wvh[i] = *reinterpret_cast<const uint4*>(cur + WRB + woh + i * 16 * 128);
                if (NP == 3) wvl[i] = *reinterpret_cast<const uint4*>(cur + WRB + wol + i * 16 * 128);
            }
        }
    };
    auto mfmas = [&]() {
        bf16x8 xh[TB], xl[TB], sh[TB], sl[TB];
#pragma unroll
        for (int j = 0; j < TB; ++j) {
            const float v[8] = {xr[j][0].x, xr[j][0].y, xr[j][0].z, xr[j][0].w, xr[j][1].x, xr[j][1].y, xr[j][1].z, xr[j][1].w};
            uint32_t ph[4], pl[4], qh[4], ql[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (NP == 1) {                                                           // one RNE 16-bit value per element (unbiased)
                    const floatx2 hv = {v[2 * t], v[2 * t + 1]};
                    ph[t] = F16 ? __builtin_bit_cast(uint32_t, __builtin_convertvector(hv, f16x2))
                                : __builtin_bit_cast(uint32_t, __builtin_convertvector(hv, bf16x2));
                    pl[t] = 0; qh[t] = 0; ql[t] = 0;
                    if (!MEAN_ONLY) {
                        const floatx2 sv = {v[2 * t] * v[2 * t], v[2 * t + 1] * v[2 * t + 1]};
                        qh[t] = F16 ? __builtin_bit_cast(uint32_t, __builtin_convertvector(sv, f16x2))
                                    : __builtin_bit_cast(uint32_t, __builtin_convertvector(sv, bf16x2));
                    }
                    continue;
                }
                const uint32_t u0 = __float_as_uint(v[2 * t]), u1 = __float_as_uint(v[2 * t + 1]);
                ph[t] = __builtin_amdgcn_perm(u1, u0, 0x07060302);                       // {hi16(v1), hi16(v0)}
                const floatx2 lo = {v[2 * t] - __uint_as_float(u0 & 0xFFFF0000u), v[2 * t + 1] - __uint_as_float(u1 & 0xFFFF0000u)};
                pl[t] = __builtin_bit_cast(uint32_t, __builtin_convertvector(lo, bf16x2));
                if (!MEAN_ONLY) {
                    const float s0 = v[2 * t] * v[2 * t], s1 = v[2 * t + 1] * v[2 * t + 1];
                    const uint32_t w0 = __float_as_uint(s0), w1 = __float_as_uint(s1);
                    qh[t] = __builtin_amdgcn_perm(w1, w0, 0x07060302);
                    const floatx2 slo = {s0 - __uint_as_float(w0 & 0xFFFF0000u), s1 - __uint_as_float(w1 & 0xFFFF0000u)};
                    ql[t] = __builtin_bit_cast(uint32_t, __builtin_convertvector(slo, bf16x2));
                }
            }
            xh[j] = __builtin_bit_cast(bf16x8, make_uint4(ph[0], ph[1], ph[2], ph[3]));
            xl[j] = __builtin_bit_cast(bf16x8, make_uint4(pl[0], pl[1], pl[2], pl[3]));
            if (!MEAN_ONLY) {
                sh[j] = __builtin_bit_cast(bf16x8, make_uint4(qh[0], qh[1], qh[2], qh[3]));
                sl[j] = __builtin_bit_cast(bf16x8, make_uint4(ql[0], ql[1], ql[2], ql[3]));
            }
        }
        // b-tile outer: the second tile's conversions can issue under the first tile's MFMAs
#pragma unroll
        for (int j = 0; j < TB; ++j) {
#pragma unroll
            for (int i = 0; i < TO; ++i) {
                if (F16) {
                    accm[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wh[i]), __builtin_bit_cast(f16x8, xh[j]),
                                                                        accm[i][j], 0, 0, 0);
                    if (!MEAN_ONLY)
                        accv[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wvh[i]), __builtin_bit_cast(f16x8, sh[j]),
                                                                            accv[i][j], 0, 0, 0);
                    continue;
                }
                const bf16x8 ah = __builtin_bit_cast(bf16x8, wh[i]);
                const bf16x8 al = NP == 3 ? __builtin_bit_cast(bf16x8, wl[i]) : ah;
                accm[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, xh[j], accm[i][j], 0, 0, 0);
                if (NP == 3) {
                    accm[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, xh[j], accm[i][j], 0, 0, 0);
                    accm[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, xl[j], accm[i][j], 0, 0, 0);
                }
                if (!MEAN_ONLY) {
                    const bf16x8 bh = __builtin_bit_cast(bf16x8, wvh[i]);
                    accv[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, sh[j], accv[i][j], 0, 0, 0);
                    if (NP == 3) {
                        accv[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wvl[i]), sh[j], accv[i][j], 0, 0, 0);
                        accv[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, sl[j], accv[i][j], 0, 0, 0);
                    }
                }
            }
        }
    };

    dma_step(0, smc);
    __syncthreads();
    for (int c = 0; c < nsteps; ++c) {
        read_frags(smc + (c & 1) * BUFB);
        __builtin_amdgcn_sched_barrier(0);
        if (c + 1 < nsteps) dma_step(c + 1, smc + ((c & 1) ^ 1) * BUFB);
        __builtin_amdgcn_sched_barrier(0);
        mfmas();
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
    }

    GemmArgs ao = a;                                     // split-K: partial product z goes to its own output slab
    if (a.kchunk) ao.out = a.out + (size_t)blockIdx.z * a.split_stride;
    epilogue_tile<TO, TB, MEAN_ONLY>(ao, o0, q, b0 + wv * TB * 16 + lr, accm, accv);
}
template <int TO, int TB, int WB, bool MEAN_ONLY, int NP = 3, bool F16 = false>
__global__ __launch_bounds__(WB * 64, (TB * TO > 10) ? 1 : (WB == 8 ? 4 : 2)) void lrt_gemm_bf16x3_kernel(const GemmArgs a_in) {
    lrt_gemm_bf16x3_body<TO, TB, WB, MEAN_ONLY, NP, F16>(a_in);
}

// ------------------------------------------------------------------------------------------------
// (A 256 x 80 three-stage ring form of the kernel above -- one 8-wave workgroup per CU, LDS-DMA two K steps ahead, the two
// halves of the workgroup half a step apart -- was rebuilt in round 2 on today's piece shapes and measured again: bit-identical,
// 1.38-1.51 us per K step against 1.40-1.44 for the 128 x 80 kernel in four orderings of DMA / reads / MFMAs.  Source: commits
// 8e8de53 and its successor; numbers and the reading in DESIGN.md 7.7.)

// ------------------------------------------------------------------------------------------------
// Skinny-output variant (O <= 16: the 10-class head).  One 16(o) x 16(b) accumulator pair per wave;
// the 16 waves of a 1024-thread workgroup split K 16 ways (chunk c -> wave c mod 16), every wave
// issues ALL its global loads (x fragment straight to registers: the tile is read once, LDS staging
// would only add a round trip) before its first MFMA, then the 16 partial accumulators are summed
// through LDS in a fixed order (deterministic) and wave 0 runs the epilogue -- optionally fused with
// log_softmax over the row (LBBNN-GP-MF-LRT.py:210), whose <=16 logits sit on 4 lanes x 4 registers.
// HBM-bound: x is (B, I) fp32 read once.
constexpr int SK_WAVES = 16;
constexpr int SK_NCH = 5;      // chunks per wave per batch: covers I <= 16*16*5 = 1280 in one batch

template <bool MEAN_ONLY, bool XVEC>
__global__ __launch_bounds__(SK_WAVES * 64) void lrt_gemm_skinny_kernel(const GemmArgs a_in) {
    const GemmArgs a = member_view(a_in);
    __shared__ __attribute__((aligned(16))) float red[SK_WAVES][2][64][4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int lr = lane & 15, q = lane >> 4;
    constexpr int SKR = 16;
    const int b0 = blockIdx.x * SKR;
    const int b = b0 + lr;
    const bool brow = b < a.B && lr < SKR, orow = lr < a.O;
    floatx4 accm = {0.f, 0.f, 0.f, 0.f}, accv = {0.f, 0.f, 0.f, 0.f};
    // wave 0 runs the epilogue: what it loads there (per-feature constants, explicit eps / combine operands, the Philox
    // state) is requested HERE, ahead of the x stream, instead of as a dependent round trip after the reduction
    const EpiCtx ec = make_epi_ctx<MEAN_ONLY>(a);
    const int o = 4 * q;
    const bool live = brow && o < a.O;
    OConst oc;
    float pa[4] = {0.f, 0.f, 0.f, 0.f}, pb[4] = {0.f, 0.f, 0.f, 0.f};
    if (wv == 0 && live) {
        oc = load_oconst(a, o);
        if (!MEAN_ONLY && a.eps) load4_rows(a.eps + (size_t)b * a.O + o, ec.ovec, o, a.O, pa);
        if (MEAN_ONLY && a.comb_x) {
            const bool cvec = ec.ovec && ((a.ld_cx | a.ld_ca) & 3) == 0 &&
                              ((reinterpret_cast<uintptr_t>(a.comb_x) | reinterpret_cast<uintptr_t>(a.comb_add)) & 15u) == 0;
            load4_rows(a.comb_x + (size_t)b * a.ld_cx + o, cvec, o, a.O, pa);
            load4_rows(a.comb_add + (size_t)b * a.ld_ca + o, cvec, o, a.O, pb);
        }
    }
    const int nchunks = (a.I + BK - 1) / BK;
    for (int cb = wv; cb < nchunks; cb += SK_WAVES * SK_NCH) {
        float4 xf[SK_NCH], wm[SK_NCH], wvv[SK_NCH];
#pragma unroll
        for (int u = 0; u < SK_NCH; ++u) {
            const int c = cb + u * SK_WAVES;
            const int k = c * BK + 4 * q;
            float4 vx = make_float4(0.f, 0.f, 0.f, 0.f), vm = vx, vv = vx;
            if (c < nchunks) {
                if (brow) {
                    const float* p = a.x + (size_t)b * a.ldx + k;
                    if (XVEC) { if (k < a.I) vx = *reinterpret_cast<const float4*>(p); }
                    else {
                        if (k + 0 < a.I) vx.x = p[0];
                        if (k + 1 < a.I) vx.y = p[1];
                        if (k + 2 < a.I) vx.z = p[2];
                        if (k + 3 < a.I) vx.w = p[3];
                    }
                }
                if (orow) {
                    vm = *reinterpret_cast<const float4*>(a.e_w + (size_t)lr * a.ld + k);
                    if (!MEAN_ONLY) vv = *reinterpret_cast<const float4*>(a.var_w + (size_t)lr * a.ld + k);
                }
            }
            xf[u] = vx; wm[u] = vm; wvv[u] = vv;
        }
#pragma unroll
        for (int u = 0; u < SK_NCH; ++u) {
            accm = __builtin_amdgcn_mfma_f32_16x16x4f32(wm[u].x, xf[u].x, accm, 0, 0, 0);
            accm = __builtin_amdgcn_mfma_f32_16x16x4f32(wm[u].y, xf[u].y, accm, 0, 0, 0);
            accm = __builtin_amdgcn_mfma_f32_16x16x4f32(wm[u].z, xf[u].z, accm, 0, 0, 0);
            accm = __builtin_amdgcn_mfma_f32_16x16x4f32(wm[u].w, xf[u].w, accm, 0, 0, 0);
            if (!MEAN_ONLY) {
                accv = __builtin_amdgcn_mfma_f32_16x16x4f32(wvv[u].x, xf[u].x * xf[u].x, accv, 0, 0, 0);
                accv = __builtin_amdgcn_mfma_f32_16x16x4f32(wvv[u].y, xf[u].y * xf[u].y, accv, 0, 0, 0);
                accv = __builtin_amdgcn_mfma_f32_16x16x4f32(wvv[u].z, xf[u].z * xf[u].z, accv, 0, 0, 0);
                accv = __builtin_amdgcn_mfma_f32_16x16x4f32(wvv[u].w, xf[u].w * xf[u].w, accv, 0, 0, 0);
            }
        }
    }
    *reinterpret_cast<floatx4*>(&red[wv][0][lane][0]) = accm;
    if (!MEAN_ONLY) *reinterpret_cast<floatx4*>(&red[wv][1][lane][0]) = accv;
    __syncthreads();
    if (wv != 0) return;
    floatx4 sm = *reinterpret_cast<const floatx4*>(&red[0][0][lane][0]);
    floatx4 sv = {0.f, 0.f, 0.f, 0.f};
    if (!MEAN_ONLY) sv = *reinterpret_cast<const floatx4*>(&red[0][1][lane][0]);
#pragma unroll 3
    for (int w = 1; w < SK_WAVES; ++w) {
        sm += *reinterpret_cast<const floatx4*>(&red[w][0][lane][0]);
        if (!MEAN_ONLY) sv += *reinterpret_cast<const floatx4*>(&red[w][1][lane][0]);
    }
    float res[4] = {0.f, 0.f, 0.f, 0.f}, sd[4] = {0.f, 0.f, 0.f, 0.f};
    if (live) epilogue4<MEAN_ONLY>(a, ec, oc, b, o, sm, sv, pa, pa, pb, res, sd);
    if (a.log_softmax) {
        // the row's logits live on lanes lr, lr+16, lr+32, lr+48 (q = 0..3), 4 registers each
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 4; ++r) if (live && o + r < a.O) mx = fmaxf(mx, res[r]);
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float se = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) if (live && o + r < a.O) se += expf(res[r] - mx);
        se += __shfl_xor(se, 16, 64);
        se += __shfl_xor(se, 32, 64);
        const float lse = mx + logf(se);
#pragma unroll
        for (int r = 0; r < 4; ++r) res[r] -= lse;
    }
    if (live) {
        store4_rows(a.out + (size_t)b * a.ldo + o, ec.ovec, o, a.O, res);
        if (!MEAN_ONLY && a.std_out) store4_rows(a.std_out + (size_t)b * a.O + o, ec.ovec, o, a.O, sd);
    }
}

// *hosted (if non-NULL, and a.fin.n > 0): set to true when the launch carries the finalize row; otherwise the caller
// runs the finalize as a launch of its own
template <int TO, int TB, int WB>
int launch_cfg(GemmArgs& a, bool mean_only, bool xvec, hipStream_t s, bool* hosted) {
    constexpr int BN = TO * 16, BM = TB * WB * 16;
    dim3 grid((a.O + BN - 1) / BN, (a.B + BM - 1) / BM, a.members > 1 ? a.members : 1);
    dim3 block(WB * 64);
    const long nblocks = (long)grid.x * grid.y * grid.z;
    static const bool no_dma = getenv("LBBNN_GEMM_NO_DMA") != nullptr;     // A/B knob for bench sweeps
    const int fin_n = a.fin.n;
    a.fin.n = 0;
    if (xvec && (a.I % BK) == 0 && !no_dma) {
        const size_t l_full = lds_request(2u * (BM + 2 * BN) * DROW * sizeof(float), nblocks);
        const size_t l_mean = lds_request(2u * (BM + BN) * DROW * sizeof(float), nblocks);
        if (fin_n > 0 && hosted) {
            a.fin.n = fin_n;
            if (piggy_lds_bytes(a.fin) <= (mean_only ? l_mean : l_full)) { grid.y += 1; *hosted = true; }
            else a.fin.n = 0;
        }
        if (mean_only) return launch_one(lrt_gemm_f32_dma_kernel<TO, TB, WB, true>, grid, block, l_mean, s, a);
        return launch_one(lrt_gemm_f32_dma_kernel<TO, TB, WB, false>, grid, block, l_full, s, a);
    }
    const size_t lds_full = lds_request(2u * (BM + 2 * BN) * LDS_LD * sizeof(float), nblocks);
    const size_t lds_mean = lds_request(2u * (BM + BN) * LDS_LD * sizeof(float), nblocks);
    if (mean_only) {
        if (xvec) return launch_one(lrt_gemm_f32_kernel<TO, TB, WB, true, true>, grid, block, lds_mean, s, a);
        return launch_one(lrt_gemm_f32_kernel<TO, TB, WB, true, false>, grid, block, lds_mean, s, a);
    }
    if (xvec) return launch_one(lrt_gemm_f32_kernel<TO, TB, WB, false, true>, grid, block, lds_full, s, a);
    return launch_one(lrt_gemm_f32_kernel<TO, TB, WB, false, false>, grid, block, lds_full, s, a);
}

template <int TO, int TB, int WB>
int launch_split_cfg(GemmArgs& a, bool mean_only, hipStream_t s, bool* hosted) {
    constexpr int BN = TO * 16, BM = TB * WB * 16;
    dim3 grid((a.O + BN - 1) / BN, (a.B + BM - 1) / BM,
              a.kchunk ? (a.I + a.kchunk - 1) / a.kchunk : (a.members > 1 ? a.members : 1));
    dim3 block(WB * 64);
    const long nblocks = (long)grid.x * grid.y * grid.z;
    const size_t l_full = lds_request(2u * (BM * 128 + 2 * BN * 128), nblocks);
    const size_t l_mean = lds_request(2u * (BM * 128 + 1 * BN * 128), nblocks);
    const int fin_n = a.fin.n;
    a.fin.n = 0;
    if (fin_n > 0 && hosted && !a.kchunk) {
        a.fin.n = fin_n;
        if (piggy_lds_bytes(a.fin) <= (mean_only ? l_mean : l_full)) { grid.y += 1; *hosted = true; }
        else a.fin.n = 0;
    }
    if (a.single16 == 2 && !mean_only) return launch_one(lrt_gemm_bf16x3_kernel<TO, TB, WB, false, 1, true>, grid, block, l_full, s, a);
    if (a.single16 && !mean_only) return launch_one(lrt_gemm_bf16x3_kernel<TO, TB, WB, false, 1>, grid, block, l_full, s, a);
    if (mean_only) return launch_one(lrt_gemm_bf16x3_kernel<TO, TB, WB, true>, grid, block, l_mean, s, a);
    return launch_one(lrt_gemm_bf16x3_kernel<TO, TB, WB, false>, grid, block, l_full, s, a);
}

int launch_split(GemmArgs& a, bool mean_only, hipStream_t s, bool* hosted) {
    // 128x80 tile, 2 workgroups/CU, 2 LDS stages.  Variants measured and dropped: 256x80 3-stage ring
    // (one workgroup/CU), 128x160 with 4 or 8 waves, x split once per tile through LDS, x delivered pre-split.
    const long nz = a.kchunk ? (a.I + a.kchunk - 1) / a.kchunk : 1;
    const long blocks_big = (long)((a.O + 79) / 80) * ((a.B + 127) / 128) * nz;
    if (blocks_big >= 256 && a.B >= 96) return launch_split_cfg<5, 2, 4>(a, mean_only, s, hosted);
    return launch_split_cfg<5, 1, 2>(a, mean_only, s, hosted);
}

}  // namespace

static int lrt_gemm_impl(const float* x, int ldx, const void* e_w, const void* var_w, int ld,
                         const float* bias_mean, const float* bias_var, const float* var_scale,
                         const float* eps, const uint64_t* rng, uint32_t rng_stream, int64_t row_offset,
                         float* out, int ldo, float* std_out, int B, int I, int O, int flags, void* stream, int kchunk = 0,
                         const FinalizePiggy* fin = nullptr, bool* hosted = nullptr,
                         const float* comb_x = nullptr, int ld_cx = 0, const float* comb_add = nullptr, int ld_ca = 0,
                         int members = 1, long long x_ms = 0, long long w_ms = 0, long long o_ms = 0,
                         unsigned long long m_adv = 0) {
    if (B == 0 && I > 0 && O > 0) return 0;        // empty batch (torch.mm of 0 rows, LBBNN-GP-MF-LRT.py:172): nothing to do
    if (!x || !e_w || !out) return LBBNN_E_NULL;
    if (B <= 0 || I <= 0 || O <= 0 || ldx < I || ldo < O) return LBBNN_E_SHAPE;
    if (flags & ~(LBBNN_F_RELU | LBBNN_F_MEAN_ONLY | LBBNN_F_SPLIT16 | LBBNN_F_LOG_SOFTMAX | LBBNN_F_SINGLE16 | LBBNN_F_HALF16))
        return LBBNN_E_FLAGS;
    if ((flags & LBBNN_F_HALF16) && (!(flags & LBBNN_F_SINGLE16) || (flags & LBBNN_F_MEAN_ONLY))) return LBBNN_E_FLAGS;
    if ((flags & LBBNN_F_LOG_SOFTMAX) && (O > 16 || (flags & LBBNN_F_RELU))) return LBBNN_E_FLAGS;
    if ((flags & LBBNN_F_SINGLE16) && !(flags & LBBNN_F_SPLIT16)) return LBBNN_E_FLAGS;
    const bool split = (flags & LBBNN_F_SPLIT16) != 0;
    const bool mean_only = (flags & LBBNN_F_MEAN_ONLY) != 0;
    if (!mean_only && !var_w) return LBBNN_E_NULL;
    if (!mean_only && !eps && !rng) return LBBNN_E_NOISE;
    if (ld < I || (ld & 31)) return LBBNN_E_ALIGN;
    if ((reinterpret_cast<uintptr_t>(e_w) & 15u) || (var_w && (reinterpret_cast<uintptr_t>(var_w) & 15u))) return LBBNN_E_ALIGN;

    GemmArgs a;
    a.x = x; a.e_w = static_cast<const float*>(e_w); a.var_w = static_cast<const float*>(var_w);
    a.bias_mean = bias_mean; a.bias_var = bias_var; a.var_scale = var_scale;
    a.eps = eps; a.rng = rng; a.out = out; a.std_out = std_out; a.row_offset = row_offset;
    a.ldx = ldx; a.ld = ld; a.ldo = ldo; a.B = B; a.I = I; a.O = O;
    a.rng_stream = rng_stream; a.relu = (flags & LBBNN_F_RELU) ? 1 : 0;
    a.log_softmax = (flags & LBBNN_F_LOG_SOFTMAX) ? 1 : 0;
    a.kchunk = kchunk; a.split_stride = (long long)B * ldo;
    if (fin) a.fin = *fin; else a.fin = FinalizePiggy{};
    a.comb_x = comb_x; a.comb_add = comb_add; a.ld_cx = ld_cx; a.ld_ca = ld_ca;
    a.single16 = (flags & LBBNN_F_HALF16) ? 2 : ((flags & LBBNN_F_SINGLE16) ? 1 : 0);
    a.members = members; a.x_ms = x_ms; a.w_ms = w_ms; a.o_ms = o_ms; a.m_adv = m_adv; a.m_off = 0;
    if (members > 1 && (kchunk || fin || eps || std_out || comb_x)) return LBBNN_E_FLAGS;

    const bool xvec = ((I & 3) == 0) && ((ldx & 3) == 0) && ((reinterpret_cast<uintptr_t>(x) & 15u) == 0);
    hipStream_t s = static_cast<hipStream_t>(stream);
    // Tile choice: 80(o) x 128(b) fills the chip for the headline shapes (B=4096, O=1200 -> 15x32 = 480
    // workgroups, 2 resident per CU); small problems take a 80x32 tile for more workgroups; a skinny
    // output (O <= 16, the 10-class head) takes the split-K kernel above.
    if (split) {
        // split-precision operands: aligned x, I % 8 == 0, >= 16 B of zero tail when there is a K tail, O > 16
        const bool tail = (I % BKS) != 0;
        if (!xvec || (I & 7) || O <= 16 || (tail && (ld - I) < 8)) return LBBNN_E_ALIGN;
        // the split kernel addresses x and the operands through 32-bit buffer offsets
        if (((size_t)(B - 1) * ldx + I) * 4 >= 0x7FFFFFF0u || (size_t)O * ld * 4 >= 0x7FFFFFF0u) return LBBNN_E_SHAPE;
        return launch_split(a, mean_only, s, hosted);
    }
    if (O <= 16) {
        a.fin.n = 0;
        dim3 grid((B + 15) / 16, 1, a.members > 1 ? a.members : 1), block(SK_WAVES * 64);
        if (mean_only) {
            if (xvec) hipLaunchKernelGGL((lrt_gemm_skinny_kernel<true, true>), grid, block, 0, s, a);
            else      hipLaunchKernelGGL((lrt_gemm_skinny_kernel<true, false>), grid, block, 0, s, a);
        } else {
            if (xvec) hipLaunchKernelGGL((lrt_gemm_skinny_kernel<false, true>), grid, block, 0, s, a);
            else      hipLaunchKernelGGL((lrt_gemm_skinny_kernel<false, false>), grid, block, 0, s, a);
        }
        return (int)hipGetLastError();
    }
    const long blocks_big = (long)((O + 79) / 80) * ((B + 127) / 128);
    if (blocks_big >= 256) return launch_cfg<5, 2, 4>(a, mean_only, xvec, s, hosted);
    return launch_cfg<5, 1, 2>(a, mean_only, xvec, s, hosted);
}

extern "C" int lbbnn_lrt_gemm(const float* x, int ldx, const void* e_w, const void* var_w, int ld,
                              const float* bias_mean, const float* bias_var, const float* var_scale,
                              const float* eps, const uint64_t* rng, uint32_t rng_stream, int64_t row_offset,
                              float* out, int ldo, int B, int I, int O, int flags, void* stream) {
    return lrt_gemm_impl(x, ldx, e_w, var_w, ld, bias_mean, bias_var, var_scale, eps, rng, rng_stream, row_offset,
                         out, ldo, nullptr, B, I, O, flags, stream);
}

extern "C" int lbbnn_lrt_gemm_train(const float* x, int ldx, const void* e_w, const void* var_w, int ld,
                                    const float* bias_mean, const float* bias_var, const float* var_scale,
                                    const float* eps, const uint64_t* rng, uint32_t rng_stream, int64_t row_offset,
                                    float* out, int ldo, float* std_out, int B, int I, int O, int flags, void* stream) {
    if ((flags & LBBNN_F_MEAN_ONLY) && std_out) return LBBNN_E_FLAGS;
    // (log_softmax + std_out is the training forward of the 10-class head: the backward needs sqrt(var), not the logits)
    return lrt_gemm_impl(x, ldx, e_w, var_w, ld, bias_mean, bias_var, var_scale, eps, rng, rng_stream, row_offset,
                         out, ldo, std_out, B, I, O, flags, stream);
}

// lbbnn_lrt_gemm with the KL finalize of a whole network carried by one extra workgroup of the same launch (include/lbbnn.h)
static int gemm_finalize_impl(const float* x, int ldx, const void* e_w, const void* var_w, int ld,
                              const float* bias_mean, const float* bias_var, const float* var_scale,
                              const float* eps, const uint64_t* rng, uint32_t rng_stream, int64_t row_offset,
                              float* out, int ldo, float* std_out, int B, int I, int O, int flags,
                              const lbbnn_layer_desc_t* layers, int n, const uint64_t* fin_rng, float* kl_total,
                              uint64_t* rng_live, uint64_t advance, void* stream) {
    if ((flags & LBBNN_F_MEAN_ONLY) && std_out) return LBBNN_E_FLAGS;
    if (n == 0) {
        // nothing to finalize (a forward without KL): the plain GEMM, then the advance as the tiny launch it is
        const int rc = lrt_gemm_impl(x, ldx, e_w, var_w, ld, bias_mean, bias_var, var_scale, eps, rng, rng_stream, row_offset,
                                     out, ldo, std_out, B, I, O, flags, stream, 0);
        if (rc || !rng_live || !advance) return rc;
        return lbbnn_rng_advance(rng_live, advance, stream);
    }
    FinalizePiggy fin{};
    if (const int rc = fill_finalize_args(layers, n, fin_rng, fin.l, fin.active)) return rc;
    if (kl_total) for (int i = 0; i < n; ++i) if (!fin.active[i]) return LBBNN_E_NULL;   // a total needs every layer's KL
    fin.n = n; fin.total = kl_total;
    fin.rng_adv = advance ? rng_live : nullptr; fin.adv = advance;
    bool hosted = false;
    const int rc = lrt_gemm_impl(x, ldx, e_w, var_w, ld, bias_mean, bias_var, var_scale, eps, rng, rng_stream, row_offset,
                                 out, ldo, std_out, B, I, O, flags, stream, 0, &fin, &hosted);
    if (rc) return rc;
    if (hosted) return 0;
    // this GEMM's kernel cannot host the extra workgroup (small tile configuration, skinny output, LDS): same work, own launch
    return launch_kl_finalize_all(fin.l, fin.active, n, advance ? rng_live : nullptr, advance, kl_total,
                                  static_cast<hipStream_t>(stream));
}

extern "C" int lbbnn_lrt_gemm_finalize(const float* x, int ldx, const void* e_w, const void* var_w, int ld,
                                       const float* bias_mean, const float* bias_var, const float* var_scale,
                                       const float* eps, const uint64_t* rng, uint32_t rng_stream, int64_t row_offset,
                                       float* out, int ldo, float* std_out, int B, int I, int O, int flags,
                                       const lbbnn_layer_desc_t* layers, int n, const uint64_t* fin_rng, float* kl_total,
                                       void* stream) {
    if (n <= 0) return LBBNN_E_SHAPE;
    return gemm_finalize_impl(x, ldx, e_w, var_w, ld, bias_mean, bias_var, var_scale, eps, rng, rng_stream, row_offset, out, ldo,
                              std_out, B, I, O, flags, layers, n, fin_rng, kl_total, nullptr, 0, stream);
}

extern "C" int lbbnn_lrt_gemm_finalize_adv(const float* x, int ldx, const void* e_w, const void* var_w, int ld,
                                           const float* bias_mean, const float* bias_var, const float* var_scale,
                                           const float* eps, const uint64_t* rng, uint32_t rng_stream, int64_t row_offset,
                                           float* out, int ldo, float* std_out, int B, int I, int O, int flags,
                                           const lbbnn_layer_desc_t* layers, int n, const uint64_t* fin_rng, float* kl_total,
                                           uint64_t* rng_live, uint64_t advance, void* stream) {
    if (n < 0 || (n > 0 && !layers)) return LBBNN_E_SHAPE;
    if (advance && !rng_live) return LBBNN_E_NULL;
    return gemm_finalize_impl(x, ldx, e_w, var_w, ld, bias_mean, bias_var, var_scale, eps, rng, rng_stream, row_offset, out, ldo,
                              std_out, B, I, O, flags, layers, n, fin_rng, kl_total, rng_live, advance, stream);
}

// An ensemble of `members` forwards of one layer in ONE launch (include/lbbnn.h)
extern "C" int lbbnn_lrt_gemm_members(const float* x, int ldx, int64_t x_mstride, const void* e_w, int64_t w_mstride,
                                      const void* var_w, int ld, const float* bias_mean, const float* bias_var,
                                      const uint64_t* rng, uint32_t rng_stream, int64_t row_offset, uint64_t member_advance,
                                      float* out, int ldo, int64_t o_mstride, int B, int I, int O, int flags, int members,
                                      void* stream) {
    if (members < 1 || members > 65535) return LBBNN_E_SHAPE;
    if (x_mstride < 0 || w_mstride < 0 || o_mstride < (int64_t)B * ldo) return LBBNN_E_SHAPE;
    if ((x_mstride & 3) || (w_mstride & 3) || (o_mstride & 3)) return LBBNN_E_ALIGN;       // every member 16-B aligned (pad the stride)
    if (flags & LBBNN_F_SPLIT16)                                                           // 32-bit buffer offsets
        if (((size_t)(B - 1) * ldx + I) * 4 >= 0x7FFFFFF0u) return LBBNN_E_SHAPE;
    return lrt_gemm_impl(x, ldx, e_w, var_w, ld, bias_mean, bias_var, nullptr, nullptr, rng, rng_stream, row_offset, out, ldo,
                         nullptr, B, I, O, flags, stream, 0, nullptr, nullptr, nullptr, 0, nullptr, 0, members,
                         (long long)x_mstride, (long long)w_mstride, (long long)o_mstride, member_advance);
}

// Mean-only product with the input-gradient combination in its epilogue (include/lbbnn.h)
extern "C" int lbbnn_lrt_gemm_combine(const float* x, int ldx, const void* w_op, int ld, const float* comb_x, int ld_cx,
                                      const float* comb_add, int ld_ca, float* out, int ldo, int B, int I, int O, int flags,
                                      void* stream) {
    if (!comb_x || !comb_add) return LBBNN_E_NULL;
    if (ld_cx < O || ld_ca < O) return LBBNN_E_SHAPE;
    if (flags & ~LBBNN_F_SPLIT16) return LBBNN_E_FLAGS;
    if (O <= 16) return LBBNN_E_SHAPE;                       // (the skinny kernel has its own epilogue)
    return lrt_gemm_impl(x, ldx, w_op, nullptr, ld, nullptr, nullptr, nullptr, nullptr, nullptr, 0u, 0, out, ldo, nullptr,
                         B, I, O, flags | LBBNN_F_MEAN_ONLY, stream, 0, nullptr, nullptr, comb_x, ld_cx, comb_add, ld_ca);
}

// Split-K plain product on the bf16x3 kernel: out[z] = x[:, Kz] . w[:, Kz]^T for the k ranges Kz = [z*kchunk, (z+1)*kchunk).
extern "C" int lbbnn_matmul_splitk(const float* x, int ldx, const void* w_op, int ld, float* out, int ldo,
                                   int B, int I, int O, int kchunk, void* stream) {
    if (kchunk <= 0 || (kchunk & 31)) return LBBNN_E_ALIGN;
    return lrt_gemm_impl(x, ldx, w_op, nullptr, ld, nullptr, nullptr, nullptr, nullptr, nullptr, 0u, 0, out, ldo, nullptr,
                         B, I, O, LBBNN_F_MEAN_ONLY | LBBNN_F_SPLIT16, stream, kchunk);
}
