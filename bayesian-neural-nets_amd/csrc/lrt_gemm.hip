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
#include "lbbnn_device.h"
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
    long long row_offset;
    int ldx, ld, ldo, B, I, O;
    uint32_t rng_stream;
    int relu;
};

// TO x TB 16x16 tiles per wave (o x b), WB waves along b; all waves share the o extent.
template <int TO, int TB, int WB, bool MEAN_ONLY, bool XVEC>
__global__ __launch_bounds__(WB * 64, 2) void lrt_gemm_f32_kernel(const GemmArgs a) {
    constexpr int NT = WB * 64;
    constexpr int BN = TO * 16;          // output features per block
    constexpr int BM = TB * WB * 16;     // batch rows per block
    constexpr int NW = MEAN_ONLY ? 1 : 2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // [buf][ X: BM rows | Wm: BN rows | Wv: BN rows ] x LDS_LD
    constexpr int ROWS = BM + NW * BN;
    float* const buf0 = smem;
    float* const buf1 = smem + ROWS * LDS_LD;

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int lr = lane & 15, q = lane >> 4;
    const int o0 = blockIdx.x * BN;
    const int b0 = blockIdx.y * BM;

    // ---- staging plan: the block moves ROWS*4 float4 per chunk; slot s -> (row s>>2, col4 s&3)
    constexpr int SLOTS = ROWS * 4;
    constexpr int NIT = (SLOTS + NT - 1) / NT;
    float4 stage[NIT];

    auto load_chunk = [&](int k0) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int s = tid + it * NT;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (s < SLOTS) {
                const int row = s >> 2, k = k0 + ((s & 3) << 2);
                if (row < BM) {
                    const int b = b0 + row;
                    if (b < a.B) {
                        const float* p = a.x + (size_t)b * a.ldx + k;
                        if (XVEC) {
                            if (k < a.I) v = *reinterpret_cast<const float4*>(p);
                        } else {
                            if (k + 0 < a.I) v.x = p[0];
                            if (k + 1 < a.I) v.y = p[1];
                            if (k + 2 < a.I) v.z = p[2];
                            if (k + 3 < a.I) v.w = p[3];
                        }
                    }
                } else {
                    const int wr = row - BM;
                    const bool isv = (!MEAN_ONLY) && wr >= BN;
                    const int o = o0 + (isv ? wr - BN : wr);
                    // operands are zero-padded to ld (multiple of 32 >= I): no k guard needed
                    if (o < a.O) v = *reinterpret_cast<const float4*>((isv ? a.var_w : a.e_w) + (size_t)o * a.ld + k);
                }
            }
            stage[it] = v;
        }
    };
    auto store_chunk = [&](float* buf) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int s = tid + it * NT;
            if (s < SLOTS) *reinterpret_cast<float4*>(buf + (s >> 2) * LDS_LD + ((s & 3) << 2)) = stage[it];
        }
    };

    floatx4 accm[TO][TB], accv[TO][TB];
#pragma unroll
    for (int i = 0; i < TO; ++i)
#pragma unroll
        for (int j = 0; j < TB; ++j) { accm[i][j] = floatx4{0.f, 0.f, 0.f, 0.f}; accv[i][j] = floatx4{0.f, 0.f, 0.f, 0.f}; }

    const int nchunks = (a.I + BK - 1) / BK;
    load_chunk(0);
    store_chunk(buf0);
    __syncthreads();

    for (int c = 0; c < nchunks; ++c) {
        float* const cur = (c & 1) ? buf1 : buf0;
        float* const nxt = (c & 1) ? buf0 : buf1;
        const bool more = (c + 1) < nchunks;
        if (more) load_chunk((c + 1) * BK);          // global loads in flight under the MFMAs

        float4 xf[TB], xs[TB];
#pragma unroll
        for (int j = 0; j < TB; ++j) {
            xf[j] = *reinterpret_cast<const float4*>(cur + ((wv * TB + j) * 16 + lr) * LDS_LD + 4 * q);
            xs[j] = make_float4(xf[j].x * xf[j].x, xf[j].y * xf[j].y, xf[j].z * xf[j].z, xf[j].w * xf[j].w);
        }
#pragma unroll
        for (int i = 0; i < TO; ++i) {
            const float4 wm = *reinterpret_cast<const float4*>(cur + (BM + i * 16 + lr) * LDS_LD + 4 * q);
            float4 wvv = wm;
            if (!MEAN_ONLY) wvv = *reinterpret_cast<const float4*>(cur + (BM + BN + i * 16 + lr) * LDS_LD + 4 * q);
#pragma unroll
            for (int j = 0; j < TB; ++j) {
                accm[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wm.x, xf[j].x, accm[i][j], 0, 0, 0);
                accm[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wm.y, xf[j].y, accm[i][j], 0, 0, 0);
                accm[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wm.z, xf[j].z, accm[i][j], 0, 0, 0);
                accm[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wm.w, xf[j].w, accm[i][j], 0, 0, 0);
                if (!MEAN_ONLY) {
                    accv[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wvv.x, xs[j].x, accv[i][j], 0, 0, 0);
                    accv[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wvv.y, xs[j].y, accv[i][j], 0, 0, 0);
                    accv[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wvv.z, xs[j].z, accv[i][j], 0, 0, 0);
                    accv[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wvv.w, xs[j].w, accv[i][j], 0, 0, 0);
                }
            }
        }
        if (more) store_chunk(nxt);
        __syncthreads();
    }

    // ---- epilogue: lane holds out[b][o .. o+3] for each (i, j) tile
    const bool ovec = ((a.O & 3) == 0) && ((a.ldo & 3) == 0) &&
                      ((reinterpret_cast<uintptr_t>(a.out) & 15u) == 0) &&
                      (!a.eps || (reinterpret_cast<uintptr_t>(a.eps) & 15u) == 0);
    uint64_t seed = 0, offs = 0;
    if (!MEAN_ONLY && !a.eps) { seed = a.rng[0]; offs = a.rng[1]; }

#pragma unroll
    for (int j = 0; j < TB; ++j) {
        const int b = b0 + (wv * TB + j) * 16 + lr;
        if (b >= a.B) continue;
#pragma unroll
        for (int i = 0; i < TO; ++i) {
            const int o = o0 + i * 16 + 4 * q;
            if (o >= a.O) continue;
            float m[4] = {accm[i][j][0], accm[i][j][1], accm[i][j][2], accm[i][j][3]};
            float v[4] = {accv[i][j][0], accv[i][j][1], accv[i][j][2], accv[i][j][3]};
            float e[4] = {0.f, 0.f, 0.f, 0.f};
            if (!MEAN_ONLY) {
                if (a.eps) {
                    const float* ep = a.eps + (size_t)b * a.O + o;
                    if (ovec) { const float4 t = *reinterpret_cast<const float4*>(ep); e[0] = t.x; e[1] = t.y; e[2] = t.z; e[3] = t.w; }
                    else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) if (o + r < a.O) e[r] = ep[r];
                    }
                } else {
                    philox_normal4(seed, offs, a.rng_stream, (uint64_t)(a.row_offset + b), (uint32_t)(o >> 2), e);
                }
            }
            float res[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int oo = (o + r < a.O) ? o + r : a.O - 1;
                float mean = m[r] + (a.bias_mean ? a.bias_mean[oo] : 0.f);
                if (!MEAN_ONLY) {
                    float var = v[r];
                    if (a.var_scale) var *= a.var_scale[oo];
                    if (a.bias_var) var += a.bias_var[oo];
                    mean += sqrtf(var) * e[r];
                }
                res[r] = a.relu ? fmaxf(mean, 0.f) : mean;
            }
            float* op = a.out + (size_t)b * a.ldo + o;
            if (ovec) *reinterpret_cast<float4*>(op) = make_float4(res[0], res[1], res[2], res[3]);
            else {
#pragma unroll
                for (int r = 0; r < 4; ++r) if (o + r < a.O) op[r] = res[r];
            }
        }
    }
}

template <int TO, int TB, int WB>
int launch_cfg(const GemmArgs& a, bool mean_only, bool xvec, hipStream_t s) {
    constexpr int BN = TO * 16, BM = TB * WB * 16;
    dim3 grid((a.O + BN - 1) / BN, (a.B + BM - 1) / BM);
    dim3 block(WB * 64);
    const size_t lds_full = 2u * (BM + 2 * BN) * LDS_LD * sizeof(float);
    const size_t lds_mean = 2u * (BM + BN) * LDS_LD * sizeof(float);
    if (mean_only) {
        if (xvec) hipLaunchKernelGGL((lrt_gemm_f32_kernel<TO, TB, WB, true, true>), grid, block, lds_mean, s, a);
        else      hipLaunchKernelGGL((lrt_gemm_f32_kernel<TO, TB, WB, true, false>), grid, block, lds_mean, s, a);
    } else {
        if (xvec) hipLaunchKernelGGL((lrt_gemm_f32_kernel<TO, TB, WB, false, true>), grid, block, lds_full, s, a);
        else      hipLaunchKernelGGL((lrt_gemm_f32_kernel<TO, TB, WB, false, false>), grid, block, lds_full, s, a);
    }
    return (int)hipGetLastError();
}

}  // namespace

extern "C" int lbbnn_lrt_gemm(const float* x, int ldx, const void* e_w, const void* var_w, int ld,
                              const float* bias_mean, const float* bias_var, const float* var_scale,
                              const float* eps, const uint64_t* rng, uint32_t rng_stream, int64_t row_offset,
                              float* out, int ldo, int B, int I, int O, int flags, void* stream) {
    if (!x || !e_w || !out) return LBBNN_E_NULL;
    if (B <= 0 || I <= 0 || O <= 0 || ldx < I || ldo < O) return LBBNN_E_SHAPE;
    if (flags & ~(LBBNN_F_RELU | LBBNN_F_MEAN_ONLY | LBBNN_F_SPLIT16)) return LBBNN_E_FLAGS;
    if (flags & LBBNN_F_SPLIT16) return LBBNN_E_FLAGS;    // split-precision path: not in this build
    const bool mean_only = (flags & LBBNN_F_MEAN_ONLY) != 0;
    if (!mean_only && !var_w) return LBBNN_E_NULL;
    if (!mean_only && !eps && !rng) return LBBNN_E_NOISE;
    if (ld < I || (ld & 31)) return LBBNN_E_ALIGN;
    if ((reinterpret_cast<uintptr_t>(e_w) & 15u) || (var_w && (reinterpret_cast<uintptr_t>(var_w) & 15u))) return LBBNN_E_ALIGN;

    GemmArgs a;
    a.x = x; a.e_w = static_cast<const float*>(e_w); a.var_w = static_cast<const float*>(var_w);
    a.bias_mean = bias_mean; a.bias_var = bias_var; a.var_scale = var_scale;
    a.eps = eps; a.rng = rng; a.out = out; a.row_offset = row_offset;
    a.ldx = ldx; a.ld = ld; a.ldo = ldo; a.B = B; a.I = I; a.O = O;
    a.rng_stream = rng_stream; a.relu = (flags & LBBNN_F_RELU) ? 1 : 0;

    const bool xvec = ((I & 3) == 0) && ((ldx & 3) == 0) && ((reinterpret_cast<uintptr_t>(x) & 15u) == 0);
    hipStream_t s = static_cast<hipStream_t>(stream);
    // Tile choice: 80(o) x 128(b) fills the chip for the headline shapes (B=4096, O=1200 -> 15x32 = 480
    // workgroups, 2 resident per CU); small problems take a 80x32 tile for more workgroups; a skinny
    // output (O <= 16, the 10-class head) takes 16(o) x 64(b).
    if (O <= 16) return launch_cfg<1, 1, 4>(a, mean_only, xvec, s);
    const long blocks_big = (long)((O + 79) / 80) * ((B + 127) / 128);
    if (blocks_big >= 256) return launch_cfg<5, 2, 4>(a, mean_only, xvec, s);
    return launch_cfg<5, 1, 2>(a, mean_only, xvec, s);
}
