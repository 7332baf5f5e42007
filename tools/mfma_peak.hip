// Microbenchmark: ceiling of v_mfma_f32_16x16x4_f32 with the GEMM's accumulator pattern (20 independent
// 16x16 accumulators, k-outermost order), no memory traffic.  hipcc --offload-arch=gfx950 -O3 -o mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, float a0, float b0) {
    floatx4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = floatx4{0.f, 0.f, 0.f, 0.f};
    float a = a0 + threadIdx.x, b = b0 + threadIdx.x * 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a + k, b + i, acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// Same MFMA count per iteration as the GEMM chunk (80), operands re-read from LDS every iteration the
// way the GEMM does (12 ds_read_b128 from a [row][20-float] image), optional barrier per iteration.
template <bool BARRIER, bool REREAD, int WRITES, bool GLOADS>
__global__ __launch_bounds__(256, 2) void lds_mfma_loop(float* out, int iters, const float* __restrict__ src) {
    constexpr int LD = 20, BM = 128, BN = 80;
    __shared__ __attribute__((aligned(16))) float sm[2 * (BM + 2 * BN) * LD];
    for (int i = threadIdx.x; i < 2 * (BM + 2 * BN) * LD; i += 256) sm[i] = (float)(i % 7) * 0.25f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, lr = lane & 15, q = lane >> 4;
    floatx4 accm[5][2], accv[5][2];
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) { accm[i][j] = floatx4{0.f, 0.f, 0.f, 0.f}; accv[i][j] = floatx4{0.f, 0.f, 0.f, 0.f}; }
    float xf[2][4], xs[2][4], wm[5][4], wq[5][4];
    auto rd = [&](int it) {
        const float* base = REREAD ? sm + (it & 1) * 4 : sm;     // address depends on `it`: no hoisting
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float4 t = *reinterpret_cast<const float4*>(base + ((wv * 2 + j) * 16 + lr) * LD + 4 * (q & 2));
            xf[j][0] = t.x; xf[j][1] = t.y; xf[j][2] = t.z; xf[j][3] = t.w;
        }
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const float4 t = *reinterpret_cast<const float4*>(base + (BM + i * 16 + lr) * LD + 4 * (q & 2));
            const float4 u = *reinterpret_cast<const float4*>(base + (BM + BN + i * 16 + lr) * LD + 4 * (q & 2));
            wm[i][0] = t.x; wm[i][1] = t.y; wm[i][2] = t.z; wm[i][3] = t.w;
            wq[i][0] = u.x; wq[i][1] = u.y; wq[i][2] = u.z; wq[i][3] = u.w;
        }
    };
    rd(0);
    float4 stage[5];
    const float* gp = src + ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    for (int it = 0; it < iters; ++it) {
        if (GLOADS) {
#pragma unroll
            for (int w = 0; w < 5; ++w) stage[w] = *reinterpret_cast<const float4*>(gp + (size_t)w * (1 << 20) + (it & 63) * 16);
        } else {
#pragma unroll
            for (int w = 0; w < 5; ++w) stage[w] = make_float4(it, w, 1.f, 2.f);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (REREAD) rd(it);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int k = 0; k < 4; ++k) xs[j][k] = xf[j][k] * xf[j][k];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int i = 0; i < 5; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    accm[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wm[i][k], xf[j][k], accm[i][j], 0, 0, 0);
                    accv[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[i][k], xs[j][k], accv[i][j], 0, 0, 0);
                }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int w = 0; w < WRITES; ++w)
            *reinterpret_cast<float4*>(sm + ((it & 1) ^ 1) * (BM + 2 * BN) * LD + ((threadIdx.x + w * 256) % 1152 >> 2) * LD + ((threadIdx.x & 3) << 2)) = stage[w];
        if (BARRIER) __syncthreads();
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) s += accm[i][j][0] + accv[i][j][1] + accm[i][j][2] + accv[i][j][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <bool BARRIER, bool REREAD, int WRITES, bool GLOADS>
void run_lds(int blocks_per_cu, int iters) {
    float* out; hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    float* src; hipMalloc(&src, (size_t)8 * (1 << 20) * sizeof(float) + (1 << 24)); hipMemset(src, 0, (size_t)8 * (1 << 20) * sizeof(float) + (1 << 24));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int grid = 256 * blocks_per_cu;
    lds_mfma_loop<BARRIER, REREAD, WRITES, GLOADS><<<grid, 256>>>(out, 10, src);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    lds_mfma_loop<BARRIER, REREAD, WRITES, GLOADS><<<grid, 256>>>(out, iters, src);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)grid * 4 * iters * 80.0 * (16 * 16 * 4 * 2);
    printf("lds_mfma barrier=%d reread=%d writes=%d gloads=%d blocks/CU=%d: %.3f ms  %.1f TFLOP/s\n", (int)BARRIER, (int)REREAD, WRITES, (int)GLOADS, blocks_per_cu, ms, flops / ms / 1e9);
    hipFree(out); hipFree(src);
}

template <int NACC>
void run(int blocks_per_cu, int iters) {
    float* out; hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int grid = 256 * blocks_per_cu;
    mfma_loop<NACC><<<grid, 256>>>(out, 10, 1.f, 2.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    mfma_loop<NACC><<<grid, 256>>>(out, iters, 1.f, 2.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)grid * 4 * iters * 4.0 * NACC * (16 * 16 * 4 * 2);
    printf("NACC=%d blocks/CU=%d: %.3f ms  %.1f TFLOP/s\n", NACC, blocks_per_cu, ms, flops / ms / 1e9);
    hipFree(out);
}

int main() {
    run<20>(2, 2000);
    for (int rep = 0; rep < 2; ++rep) {
        run_lds<true, true, 0, false>(2, 2000);
        run_lds<true, true, 5, false>(2, 2000);
        run_lds<true, true, 0, true>(2, 2000);
        run_lds<true, true, 5, true>(2, 2000);
        run_lds<true, true, 5, true>(1, 2000);
        run_lds<true, true, 5, true>(3, 2000);
    }
    return 0;
}
