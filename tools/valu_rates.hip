// Issue rate of the VALU instructions the GEMM epilogue's noise is made of (gfx950): one wave per SIMD, N independent chains
// of one instruction, cycles per instruction from s_memtime.    hipcc --offload-arch=gfx950 -O3 tools/valu_rates.hip -o /tmp/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP 64
template <int OP>
__global__ void k(uint32_t* out, uint64_t* cyc, int iters) {
    uint32_t a[8];
    float f[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 2654435761u + i; f[i] = 0.5f + 1e-3f * (threadIdx.x + i); }
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < REP / 8; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) { const uint64_t p = (uint64_t)a[i] * 0xD2511F53u; a[i] = (uint32_t)(p >> 32) ^ (uint32_t)p; }   // v_mad_u64_u32 + xor
                if (OP == 1) a[i] = a[i] ^ (a[i] >> 3);                                                                    // shift + xor
                if (OP == 2) f[i] = __builtin_amdgcn_exp2f(f[i]);
                if (OP == 3) f[i] = __builtin_amdgcn_sinf(f[i]);
                if (OP == 4) f[i] = __builtin_amdgcn_sqrtf(f[i]);
                if (OP == 5) f[i] = __builtin_fmaf(f[i], 0.999f, 0.001f);
                if (OP == 6) a[i] = __umulhi(a[i], 0xD2511F53u);
                if (OP == 7) a[i] = a[i] * 0xD2511F53u;
            }
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    uint32_t s = 0; float g = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) { s ^= a[i]; g += f[i]; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s ^ __float_as_uint(g);
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[OP] = t1 - t0;
}

int main() {
    uint32_t* out; uint64_t* cyc;
    hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 64);
    const int iters = 200;
    const char* names[8] = {"v_mad_u64_u32 (+xor)", "v_lshr + v_xor", "v_exp_f32", "v_sin_f32", "v_sqrt_f32", "v_fma_f32", "v_mul_hi_u32", "v_mul_lo_u32"};
    for (int waves = 1; waves <= 2; ++waves) {
        printf("%d wave(s) per SIMD (block of %d threads on one CU); counter ticks per instruction per wave (s_memtime: 100 MHz ticks x 24 = shader clocks at 2.4 GHz)\n", waves, 256 * waves);
#define RUN(OP) { hipLaunchKernelGGL(k<OP>, dim3(1), dim3(256 * waves), 0, 0, out, cyc, iters); hipDeviceSynchronize(); \
                  uint64_t c; hipMemcpy(&c, cyc + OP, 8, hipMemcpyDeviceToHost); \
                  printf("  %-22s %8.3f ticks/instr\n", names[OP], (double)c / (iters * REP * (OP <= 1 ? 2 : 1))); }
        RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7)
    }
    return 0;
}
