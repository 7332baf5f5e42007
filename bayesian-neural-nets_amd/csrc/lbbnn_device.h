// Device-side helpers shared by the gfx950 kernels: wave64/block reductions, Philox4x32-10,
// Box-Muller, and the accurate elementwise forms the KL terms need.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lbbnn {

constexpr int kWave = 64;   // CDNA wavefront

// ------------------------------------------------------------------------------------------ reductions
// Fixed-order sum over the 64 lanes of a wave; every lane ends with the full sum and the order of additions is
// independent of scheduling, so results are bitwise reproducible.  Within a row of 16 lanes the four butterfly steps
// are DPP register moves (quad_perm xor 1, xor 2, row_half_mirror, row_mirror: no LDS crossbar round trip, which is
// what __shfl_xor / ds_bpermute costs -- ~100 cycles per dependent step on these latency-bound single-workgroup
// chains); the four row sums are then combined from v_readlane (lanes 0, 16, 32, 48) as (r0 + r1) + (r2 + r3).
template <int CTRL>
__device__ __forceinline__ int dpp_mov(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, false); }

template <int CTRL>
__device__ __forceinline__ float dpp_get(float v) { return __builtin_bit_cast(float, dpp_mov<CTRL>(__builtin_bit_cast(int, v))); }
template <int CTRL>
__device__ __forceinline__ double dpp_get(double v) {
    const uint64_t b = __builtin_bit_cast(uint64_t, v);
    const uint32_t lo = (uint32_t)dpp_mov<CTRL>((int)(uint32_t)b), hi = (uint32_t)dpp_mov<CTRL>((int)(uint32_t)(b >> 32));
    return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ float lane_get(float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); }
__device__ __forceinline__ double lane_get(double v, int l) {
    const uint64_t b = __builtin_bit_cast(uint64_t, v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)b, l), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(b >> 32), l);
    return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}

template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
    v += dpp_get<0xB1>(v);      // quad_perm [1,0,3,2]   lane ^ 1
    v += dpp_get<0x4E>(v);      // quad_perm [2,3,0,1]   lane ^ 2
    v += dpp_get<0x141>(v);     // row_half_mirror       quad <-> neighbouring quad
    v += dpp_get<0x140>(v);     // row_mirror            half row <-> other half: every lane holds its row's sum
    return (lane_get(v, 0) + lane_get(v, 16)) + (lane_get(v, 32) + lane_get(v, 48));
}

// Block-wide sum for blockDim.x = NW*64 threads; `scratch` holds NW values of T in LDS.
// All threads get the result. Contains two barriers.
template <typename T, int NW>
__device__ __forceinline__ T block_sum(T v, T* scratch) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();                 // protect scratch from a previous use
    if (lane == 0) scratch[w] = v;
    __syncthreads();
    T s = scratch[0];
#pragma unroll
    for (int i = 1; i < NW; ++i) s += scratch[i];
    return s;
}

// ------------------------------------------------------------------------------------------ LDS-DMA prefetch
// Copy n floats global -> LDS with `global_load_lds_dword` (64 floats per wave-instruction, written at
// wave-uniform base + lane*4).  Fire-and-forget: the caller issues every vector it needs, then waits ONCE
// (dma_wait_all), so a latency-bound kernel pays a single memory latency for all of its inputs.
// The LDS vector must be padded to a multiple of 64 floats (tail lanes re-read element n-1).
__device__ __forceinline__ void dma_stage(float* lds_dst, const float* __restrict__ src, int n) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = blockDim.x >> 6;
    for (int c = wave; c * 64 < n; c += nw) {
        const int i = min(c * 64 + lane, n - 1);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + i),
                                         (__attribute__((address_space(3))) void*)(lds_dst + c * 64), 4, 0, 0);
    }
}
__device__ __forceinline__ void dma_wait_all() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}
__host__ __device__ __forceinline__ int pad64(int n) { return (n + 63) & ~63; }

// ------------------------------------------------------------------------------------------ split (bf16x3) operand layout
// LBBNN_F_SPLIT16 operands hold w = hi + lo (both bf16) as rows of 2*ld bf16 (= the 4*ld bytes of the fp32 row they
// replace).  Per 32-k chunk one 128-B line of eight 16-B units: unit 2g = hi of k in [8g, 8g+8), unit 2g+1 = lo of the
// same k (g = 0..3) -- an MFMA lane (k group g) needs exactly units 2g and 2g+1, the access shape of the fp32 x rows, so
// one LDS swizzle serves x and weights conflict-free.  One line carries everything the GEMM needs from a row for one K
// step, and an LDS-DMA piece is 8 rows x 128 B (measured: the delivery-only build of the GEMM takes 48.9 us with such
// pieces against 58.0 us with 16 rows x 64 B from two separate hi / lo planes).  ld is a multiple of 32; the tail k in
// [I, ld) is zero in both parts.
__host__ __device__ __forceinline__ size_t split_hi_index(size_t row, int k, int ld) {
    return row * (2 * (size_t)ld) + (size_t)(k >> 5) * 64 + (size_t)((k >> 3) & 3) * 16 + (size_t)(k & 7);
}
constexpr int kSplitLoOffset = 8;       // lo part of the same k: the next 16-B unit (+8 bf16)

// ------------------------------------------------------------------------------------------ x planes (LBBNN_F_XPLANES)
// fp32 rows -> fp16 hi | lo planes (include/lbbnn.h): one work item per 8 consecutive k -- two float4 in, the hi unit and
// the lo unit (16 B each) out; groups past I (the zero tail of the plane row) are written as zeros.  `first` / `stride`:
// the calling kernel's grid-stride loop.  I % 8 == 0.
struct FormatJob { const float* x; char* planes; int ldx, ldp, B, I; };

__device__ __forceinline__ void format_x_items(const FormatJob& j, size_t first, size_t stride) {
    typedef _Float16 fx_h2 __attribute__((ext_vector_type(2)));
    typedef float fx_f2 __attribute__((ext_vector_type(2)));
    const int groups = j.ldp >> 3;
    const size_t n = (size_t)j.B * groups;
    for (size_t t = first; t < n; t += stride) {
        const int b = (int)(t / groups), g = (int)(t % groups), k = 8 * g;
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (k < j.I) {
            const float4 f0 = *reinterpret_cast<const float4*>(j.x + (size_t)b * j.ldx + k);
            const float4 f1 = *reinterpret_cast<const float4*>(j.x + (size_t)b * j.ldx + k + 4);
            v[0] = f0.x; v[1] = f0.y; v[2] = f0.z; v[3] = f0.w; v[4] = f1.x; v[5] = f1.y; v[6] = f1.z; v[7] = f1.w;
        }
        uint32_t h[4], l[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const fx_f2 pv = {v[2 * u], v[2 * u + 1]};
            const fx_h2 ph = __builtin_convertvector(pv, fx_h2);
            const fx_f2 lv = {v[2 * u] - (float)ph[0], v[2 * u + 1] - (float)ph[1]};
            h[u] = __builtin_bit_cast(uint32_t, ph);
            l[u] = __builtin_bit_cast(uint32_t, __builtin_convertvector(lv, fx_h2));
        }
        char* p = j.planes + (size_t)b * j.ldp * 4 + (size_t)(k >> 5) * 128 + ((k >> 3) & 3) * 32;
        *reinterpret_cast<uint4*>(p) = make_uint4(h[0], h[1], h[2], h[3]);
        *reinterpret_cast<uint4*>(p + 16) = make_uint4(l[0], l[1], l[2], l[3]);
    }
}

// ------------------------------------------------------------------------------------------ kernel arguments as memory
// A by-value kernel-argument struct indexed with a RUNTIME index (bt.l[blockIdx.y], a.zf.u[t]) is copied to scratch
// memory by hipcc (measured: 872 B/lane in the flow kernel, every pointer fetch then a private-memory round trip on a
// latency-bound chain).  Reading the same bytes through the kernarg segment pointer keeps them scalar loads with a
// computed offset.  T must be the type of the kernel's FIRST parameter.
#define LBBNN_CONST_AS __attribute__((address_space(4)))
template <typename T>
__device__ __forceinline__ const LBBNN_CONST_AS T* kernarg_as() {
    return (const LBBNN_CONST_AS T*)__builtin_amdgcn_kernarg_segment_ptr();
}

// ------------------------------------------------------------------------------------------ elementwise
// softplus exactly as the reference spells it: log1p(exp(rho)) (LBBNN-GP-MF-LRT.py:81-82).
__device__ __forceinline__ float softplus_ref(float rho) { return log1pf(expf(rho)); }
// alpha = 1/(1+exp(-lambda)) (LBBNN-GP-MF-LRT.py:167)
__device__ __forceinline__ float sigmoid_ref(float l) { return 1.0f / (1.0f + expf(-l)); }

// Hardware-transcendental forms (v_exp_f32 / v_log_f32 / v_rcp_f32, ~1 ulp each) for the latency-bound tails where the
// libm sequences (30-50 instructions each) are the cost: log1p(exp(rho)) by a 5-term series while exp(rho) < 0.04,
// tanh through one exp.  Absolute errors ~1e-7, far inside the 1e-4 contract of the sums they feed.
__device__ __forceinline__ float softplus_fast(float rho) {
    const float y = __expf(rho);
    if (y < 0.04f) return y * (1.f + y * (-0.5f + y * (0.33333334f + y * (-0.25f + y * 0.2f))));
    return log1pf(y);
}
// The weight pass's gate and scale (round 3), shared by every kernel that must reproduce its operands bit for bit (K1's
// row / generic kernels, lbbnn_weight_operands_t).  Raw hardware forms: v_exp_f32 is 2^x, v_rcp_f32 ~1 ulp; the library
// forms spend 4-5 more instructions per call on denormal scaling that these arguments cannot need (exp(-lambda) underflowing
// to 0 / overflowing to inf gives alpha = 1 / 0, as the rounded reference value).  sigma's exp keeps the rounding of
// rho * log2(e) (5e-7 relative at rho = -9, doubled in var_w = sigma^2 alpha^2) out through the product's exact residual.
__device__ __forceinline__ float k1_exp_raw(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
__device__ __forceinline__ float k1_exp_acc(float x) {
    const float th = x * 1.4426950408889634f;
    float tl = __builtin_fmaf(x, 1.4426950408889634f, -th);
    tl = __builtin_fmaf(x, 1.9259629911e-8f, tl);                            // log2(e) - fl32(log2(e))
    const float e = __builtin_amdgcn_exp2f(th);
    return __builtin_fmaf(e, tl * 0.6931471805599453f, e);
}
__device__ __forceinline__ float k1_alpha(float lam) { return __builtin_amdgcn_rcpf(1.0f + k1_exp_raw(-lam)); }
// softplus from y = exp(rho): the 5-term log1p series while y < 0.04 (truncation < 1e-9 relative), libm beyond
__device__ __forceinline__ float k1_sigma_of(float y) {
    if (y < 0.04f)
        return y * __builtin_fmaf(y, __builtin_fmaf(y, __builtin_fmaf(y, __builtin_fmaf(y, 0.2f, -0.25f), 0.33333334f), -0.5f), 1.f);
    return log1pf(y);
}
__device__ __forceinline__ float k1_sigma(float rho) { return k1_sigma_of(k1_exp_acc(rho)); }
// log(softplus(rho)) from the same y: rho + (-y/2 + 5y^2/24 - y^3/8 + 251y^4/2880) in the series range (|next term| < 7e-9)
__device__ __forceinline__ float k1_log_sigma_of(float rho, float y, float sigma) {
    if (y < 0.04f)
        return __builtin_fmaf(y, __builtin_fmaf(y, __builtin_fmaf(y, __builtin_fmaf(y, 0.08715278f, -0.125f), 0.20833333f), -0.5f), rho);
    return __logf(sigma);
}

// sqrt of an activation variance in the GEMM epilogues: v_sqrt_f32 alone (1 ulp).  sqrtf compiles to 12 instructions (range
// scaling + a Newton step for the last half ulp) -- 40 of them per lane of a 128 x 80 tile; a variance below 2^-126 gives 0.
__device__ __forceinline__ float sqrt_hw(float x) { return __builtin_amdgcn_sqrtf(x); }

__device__ __forceinline__ float tanh_fast(float x) {
    const float t = __expf(-2.f * fabsf(x));
    const float r = (1.f - t) * __frcp_rn(1.f + t);
    return copysignf(r, x);
}

// ------------------------------------------------------------------------------------------ Philox4x32-10
struct Philox4 { uint32_t x, y, z, w; };

__device__ __forceinline__ Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                 uint32_t k0, uint32_t k1) {
    constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // one 32x32->64 multiply (v_mad_u64_u32) per product instead of a quarter-rate mul_hi + mul_lo pair
        const uint64_t p0 = (uint64_t)M0 * (uint64_t)c0, p1 = (uint64_t)M1 * (uint64_t)c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += W0; k1 += W1;
    }
    return Philox4{c0, c1, c2, c3};
}

// Four N(0,1) draws for counter (ctr0, ctr1) of stream `stream`; rng = {seed, offset}.
// Counter layout: (ctr0_lo, ctr0_hi ^ ctr1<<?, ...) kept simple and collision-free:
//   c0 = ctr0 low 32, c1 = ctr0 high 32, c2 = ctr1, c3 = stream ^ (offset_hi mixed into key)
// key = seed_lo ^ offset_lo*golden, seed_hi ^ offset_hi.  Distinct (seed, offset, stream, ctr) never collide
// within one (seed, offset) because the counter words are distinct; offset enters through the key.
__device__ __forceinline__ void philox_normal4(uint64_t seed, uint64_t offset, uint32_t stream,
                                               uint64_t ctr0, uint32_t ctr1, float out[4]) {
    const uint32_t k0 = (uint32_t)seed ^ ((uint32_t)offset * 0x9E3779B9u);
    const uint32_t k1 = (uint32_t)(seed >> 32) ^ (uint32_t)(offset >> 32) ^ ((uint32_t)offset >> 7);
    const Philox4 r = philox4x32_10((uint32_t)ctr0, (uint32_t)(ctr0 >> 32), ctr1, stream, k0, k1);
    // Box-Muller on two pairs; u in [2^-32, 1]: (x + 1) * 2^-32 avoids log(0) (and is never a denormal, never above 1: the
    // largest x rounds to 2^32).  Round 3: the hardware forms taken directly -- v_log_f32 is log2, v_sin/v_cos take their
    // argument in revolutions, v_sqrt_f32 at its native 1 ulp -- 24 VALU instructions after the Philox rounds where
    // sqrtf / __logf / __sincosf (range checks, a Newton step, denormal scaling none of which these arguments can need)
    // compiled to 82; every consumer draws through this one function, so explicit-noise == in-kernel-noise stays bitwise.
    const float u0 = ((float)r.x + 1.0f) * 2.3283064365386963e-10f;
    const float u1 = (float)r.y * 2.3283064365386963e-10f;
    const float u2 = ((float)r.z + 1.0f) * 2.3283064365386963e-10f;
    const float u3 = (float)r.w * 2.3283064365386963e-10f;
    const float ra = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u0));    // sqrt(-2 ln u0)
    const float rb = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u2));
    out[0] = ra * __builtin_amdgcn_cosf(u1); out[1] = ra * __builtin_amdgcn_sinf(u1);
    out[2] = rb * __builtin_amdgcn_cosf(u3); out[3] = rb * __builtin_amdgcn_sinf(u3);
}

// 128 raw bits for counter (ctr0, ctr1) of stream `stream` (same keying as philox_normal4): Bernoulli(0.5) masks.
__device__ __forceinline__ Philox4 philox_bits4(uint64_t seed, uint64_t offset, uint32_t stream, uint64_t ctr0, uint32_t ctr1) {
    const uint32_t k0 = (uint32_t)seed ^ ((uint32_t)offset * 0x9E3779B9u);
    const uint32_t k1 = (uint32_t)(seed >> 32) ^ (uint32_t)(offset >> 32) ^ ((uint32_t)offset >> 7);
    return philox4x32_10((uint32_t)ctr0, (uint32_t)(ctr0 >> 32), ctr1, stream, k0, k1);
}

}  // namespace lbbnn
