// Lab (GPU box): LDS fragment-read rate of the split-precision GEMM's main loop, alone.
// Each wave issues the GEMM's 24 ds_read_b128 per K step (2 x 2 x-units of its 32 rows, 5 x 4 weight units of the 80 weight rows, the
// kernel's swizzle), waits for them, and repeats.  Nothing else runs.  Reported: cycles per step and bytes per clock per CU
// for 4 waves (one workgroup) and 8 waves (two workgroups, or one of 512 threads) per CU.
// Build: hipcc --offload-arch=gfx950 -O3 -w tools/lab/lds_rate.hip -o tools/lab/lds_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define DS_READ128(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(off) : "memory")

__device__ __forceinline__ int swzx(int r) { return (((r >> 1) & 3) << 1) | ((r >> 3) & 1); }

template <int WAVES, int MODE>
__device__ __forceinline__ void body(int iters, unsigned* sink, unsigned long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) char smc[];
    const int tid = threadIdx.x, lane = tid & 63, wv = (tid >> 6) & 3;
    const int lr = lane & 15, q = lane >> 4;
    constexpr int XB = 128 * 128, WRB = 80 * 128, BUFB = XB + 2 * WRB;
    for (int i = tid; i < 2 * BUFB / 4; i += WAVES * 64) reinterpret_cast<unsigned*>(smc)[i] = i;
    __syncthreads();
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smc;
    const int gx = swzx(lr);
    const unsigned half = (WAVES == 8 && (tid >> 8)) ? 0u : 0u;   // 8 waves of one workgroup read the same image (two b halves share w)
    const unsigned xo0 = lds0 + half + (wv * 32 + lr) * 128 + 16 * ((2 * q) ^ gx);
    const unsigned xo1 = lds0 + half + (wv * 32 + lr) * 128 + 16 * ((2 * q + 1) ^ gx);
    const unsigned woh = lds0 + XB + lr * 128 + 16 * ((2 * q) ^ gx);
    const unsigned wol = lds0 + XB + lr * 128 + 16 * ((2 * q + 1) ^ gx);
    u32x4 acc = {0, 0, 0, 0};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        const unsigned bo = (it & 1) * BUFB;
        u32x4 r[24];
        DS_READ128(r[0], xo0 + bo, 0); DS_READ128(r[1], xo1 + bo, 0);
        DS_READ128(r[2], xo0 + bo, 2048); DS_READ128(r[3], xo1 + bo, 2048);
#define W4(i) DS_READ128(r[4 + 4 * i], woh + bo, i * 2048); DS_READ128(r[5 + 4 * i], wol + bo, i * 2048); \
              DS_READ128(r[6 + 4 * i], woh + bo, WRB + i * 2048); DS_READ128(r[7 + 4 * i], wol + bo, WRB + i * 2048);
        W4(0) W4(1) W4(2) W4(3) W4(4)
        if (MODE == 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int k = 0; k < 24; ++k) asm volatile("" :: "v"(r[k]));
        if (MODE == 1 && (it & 3) == 3) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (acc.x == 0x12345u) sink[0] = acc.x;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int WAVES, int MODE>
__global__ __launch_bounds__(WAVES * 64) void lds_k(int iters, unsigned* sink, unsigned long long* cyc) { body<WAVES, MODE>(iters, sink, cyc); }

template <int WAVES, int MODE>
static void run(const char* name, int nwg, unsigned* sink, unsigned long long* cyc) {
    const int iters = 4000;
    const size_t lds = 2 * (128 * 128 + 2 * 80 * 128);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&lds_k<WAVES, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    lds_k<WAVES, MODE><<<dim3(nwg), dim3(WAVES * 64), lds, 0>>>(iters, sink, cyc);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    lds_k<WAVES, MODE><<<dim3(nwg), dim3(WAVES * 64), lds, 0>>>(iters, sink, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(nwg);
    hipMemcpy(h.data(), cyc, nwg * 8, hipMemcpyDeviceToHost);
    double cs = 0; for (auto v : h) cs += (double)v; cs /= nwg;
    const double waves_cu = (double)WAVES * nwg / 256.0;
    printf("%-44s | %7.1f us | %6.0f cycles per step (in-kernel) | %5.1f B/clk per CU | wall %.2f us per step\n", name, ms * 1e3,
           cs / iters, waves_cu * 24.0 * 1024.0 / (cs / iters), ms * 1e3 / iters);
}

int main() {
    unsigned* sink; unsigned long long* cyc;
    hipMalloc(&sink, 64); hipMalloc(&cyc, 4096 * 8);
    run<4, 0>("4 waves/CU (1 WG), wait every step", 256, sink, cyc);
    run<4, 1>("4 waves/CU (1 WG), wait every 4 steps", 256, sink, cyc);
    run<4, 0>("8 waves/CU (2 WG x 4), wait every step", 512, sink, cyc);
    run<4, 1>("8 waves/CU (2 WG x 4), wait every 4 steps", 512, sink, cyc);
    run<8, 0>("8 waves/CU (1 WG x 8), wait every step", 256, sink, cyc);
    return 0;
}
