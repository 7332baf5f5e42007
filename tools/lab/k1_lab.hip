// Lab (GPU box): what bounds the weight pass?  Streams three (O,I) fp32 arrays in and two out, with the row -> wave map of
// weight_rows_kernel, a grid-stride float4 map, and optional per-element arithmetic; each variant timed with the 48 MB
// working set (a) resident from the previous iteration and (b) flushed by a 512 MB memset between iterations.
// Build: hipcc --offload-arch=gfx950 -O3 tools/lab/k1_lab.hip -o tools/lab/k1_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ __forceinline__ float4 ld4(const float* p, size_t j) { return reinterpret_cast<const float4*>(p)[j]; }

template <int MATH>
__device__ __forceinline__ void elem(float mu, float rho, float lam, float& e, float& v) {
    if (MATH == 0) { e = mu + lam; v = rho; return; }
    const float alpha = __frcp_rn(1.0f + __expf(-lam));
    const float y = __expf(rho);
    const float sigma = y * (1.f + y * (-0.5f + y * (0.33333334f + y * (-0.25f + y * 0.2f))));
    e = mu * alpha;
    v = (sigma * sigma) * (alpha * alpha);
    if (MATH == 2) {
        const float one_m = 1.f - alpha;
        v += 1e-20f * (alpha * ((0.f - __logf(sigma)) - 0.5f + (__logf(alpha) + 3.f) + (sigma * sigma + mu * mu) * 0.5f) + one_m * (__logf(one_m) + 0.05f));
    }
}

// one wave per row, 4 rows per 256-thread workgroup (weight_rows_kernel's map)
template <int MATH, int STORE>
__global__ __launch_bounds__(256, 3) void rows_k(const float* mu, const float* rho, const float* lam, float* ew, float* vw, int O, int I) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int o = 4 * blockIdx.x + wv;
    if (o >= O) return;
    const int iq = I >> 2;
    const size_t ro = (size_t)o * iq;
    float4 m[5], r[5], l[5];
#pragma unroll
    for (int g = 0; g < 5; ++g) { const int j = lane + 64 * g; if (j < iq) { m[g] = ld4(mu, ro + j); r[g] = ld4(rho, ro + j); l[g] = ld4(lam, ro + j); } }
    float acc = 0.f;
#pragma unroll
    for (int g = 0; g < 5; ++g) {
        const int j = lane + 64 * g;
        if (j < iq) {
            float4 e, v;
            elem<MATH>(m[g].x, r[g].x, l[g].x, e.x, v.x); elem<MATH>(m[g].y, r[g].y, l[g].y, e.y, v.y);
            elem<MATH>(m[g].z, r[g].z, l[g].z, e.z, v.z); elem<MATH>(m[g].w, r[g].w, l[g].w, e.w, v.w);
            if (STORE) { reinterpret_cast<float4*>(ew)[ro + j] = e; reinterpret_cast<float4*>(vw)[ro + j] = v; }
            else acc += e.x + v.y + e.z + v.w;
        }
    }
    if (!STORE && acc == 1.2345f) ew[0] = acc;
}

// flat grid-stride float4 stream, UNR float4 per array in flight per thread
template <int MATH, int UNR>
__global__ __launch_bounds__(256) void flat_k(const float* mu, const float* rho, const float* lam, float* ew, float* vw, size_t n4) {
    const size_t stride = (size_t)gridDim.x * 256 * UNR;
    for (size_t base = (size_t)blockIdx.x * 256 * UNR + threadIdx.x; base < n4; base += stride) {
        float4 m[UNR], r[UNR], l[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) { const size_t j = base + 256 * u; if (j < n4) { m[u] = ld4(mu, j); r[u] = ld4(rho, j); l[u] = ld4(lam, j); } }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const size_t j = base + 256 * u;
            if (j < n4) {
                float4 e, v;
                elem<MATH>(m[u].x, r[u].x, l[u].x, e.x, v.x); elem<MATH>(m[u].y, r[u].y, l[u].y, e.y, v.y);
                elem<MATH>(m[u].z, r[u].z, l[u].z, e.z, v.z); elem<MATH>(m[u].w, r[u].w, l[u].w, e.w, v.w);
                reinterpret_cast<float4*>(ew)[j] = e; reinterpret_cast<float4*>(vw)[j] = v;
            }
        }
    }
}

int main() {
    const int O = 2410, I = 1200;                 // all rows of the headline net at the wide layers' width (~ same bytes)
    const size_t n = (size_t)O * I;
    float *mu, *rho, *lam, *ew, *vw; char* flush;
    hipMalloc(&mu, n * 4); hipMalloc(&rho, n * 4); hipMalloc(&lam, n * 4); hipMalloc(&ew, n * 4); hipMalloc(&vw, n * 4);
    hipMalloc(&flush, 512u << 20);
    std::vector<float> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = -4.5f + 0.001f * (float)(i % 997);
    hipMemcpy(mu, h.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(rho, h.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(lam, h.data(), n * 4, hipMemcpyHostToDevice);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto run = [&](const char* name, auto launch) {
        for (int mode = 0; mode < 2; ++mode) {
            float tot = 0.f; const int it = 30;
            for (int k = 0; k < it + 3; ++k) {
                if (mode) hipMemsetAsync(flush, k, 512u << 20, 0);
                hipEventRecord(a, 0); launch(); hipEventRecord(b, 0); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b); if (k >= 3) tot += ms;
            }
            printf("%-34s %s  %7.2f us  %6.2f TB/s (20 B/weight)\n", name, mode ? "flushed " : "resident", tot / it * 1e3, (double)n * 20 / (tot / it * 1e-3) / 1e12);
        }
    };
    const dim3 gr((O + 3) / 4), bl(256);
    run("rows  copy", [&] { rows_k<0, 1><<<gr, bl>>>(mu, rho, lam, ew, vw, O, I); });
    run("rows  copy, no store", [&] { rows_k<0, 0><<<gr, bl>>>(mu, rho, lam, ew, vw, O, I); });
    run("rows  operands math", [&] { rows_k<1, 1><<<gr, bl>>>(mu, rho, lam, ew, vw, O, I); });
    run("rows  operands + KL math", [&] { rows_k<2, 1><<<gr, bl>>>(mu, rho, lam, ew, vw, O, I); });
    run("flat  copy  x2  2048 wg", [&] { flat_k<0, 2><<<2048, bl>>>(mu, rho, lam, ew, vw, n / 4); });
    run("flat  copy  x4  1024 wg", [&] { flat_k<0, 4><<<1024, bl>>>(mu, rho, lam, ew, vw, n / 4); });
    run("flat  copy  x4  2048 wg", [&] { flat_k<0, 4><<<2048, bl>>>(mu, rho, lam, ew, vw, n / 4); });
    run("flat  KL math x4 1024 wg", [&] { flat_k<2, 4><<<1024, bl>>>(mu, rho, lam, ew, vw, n / 4); });
    run("flat  KL math x2 2048 wg", [&] { flat_k<2, 2><<<2048, bl>>>(mu, rho, lam, ew, vw, n / 4); });
    return 0;
}
