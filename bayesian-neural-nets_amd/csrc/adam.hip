// lbbnn_adam_step: multi-tensor Adam in one launch (see include/lbbnn.h).  HBM-bound elementwise: 16 B read + 12 B
// written per parameter.  Workgroup -> (tensor, 4096-element chunk) by a binary search over the per-tensor chunk
// prefix held in the kernel arguments (read through the kernarg pointer: scalar loads, no scratch copy).
#include <cmath>
#include "lbbnn_device.h"
#include "lbbnn_internal.h"

namespace {

using namespace lbbnn;
constexpr int CHUNK = 4096;     // elements per workgroup: 256 threads x 4 float4

struct AdamKArgs {
    lbbnn_adam_list_t l;
    int first[LBBNN_ADAM_MAX_TENSORS + 1];     // first workgroup of tensor i; first[n] = grid size
    float lr, b1, b2, eps, wd;
    const float* step;
};

__global__ __launch_bounds__(256) void adam_kernel(const AdamKArgs ka) {
    const LBBNN_CONST_AS AdamKArgs& a = *kernarg_as<AdamKArgs>();
    const int blk = blockIdx.x;
    int lo = 0, hi = a.l.n;                                   // largest i with first[i] <= blk
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (a.first[mid] <= blk) lo = mid; else hi = mid; }
    const int ti = lo;
    float* __restrict__ p = a.l.p[ti];
    const float* __restrict__ g = a.l.g[ti];
    float* __restrict__ m = a.l.m[ti];
    float* __restrict__ v = a.l.v[ti];
    const int64_t n = a.l.numel[ti];
    const int64_t base = (int64_t)(blk - a.first[ti]) * CHUNK;
    const float t = a.step[0] + 1.f;
    const float bc1 = 1.f - powf(a.b1, t), bc2s = sqrtf(1.f - powf(a.b2, t));
    const float step_size = a.lr / bc1, b1 = a.b1, b2 = a.b2, eps = a.eps, wd = a.wd;
    const bool vec = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                       reinterpret_cast<uintptr_t>(v)) & 15u) == 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int64_t i = base + (int64_t)(threadIdx.x + 256 * k) * 4;
        if (i >= n) break;
        float pp[4], gg[4], mm[4], vv[4];
        const int cnt = (int)((n - i) < 4 ? (n - i) : 4);
        if (vec && cnt == 4) {
            const float4 a0 = *reinterpret_cast<const float4*>(p + i), a1 = *reinterpret_cast<const float4*>(g + i);
            const float4 a2 = *reinterpret_cast<const float4*>(m + i), a3 = *reinterpret_cast<const float4*>(v + i);
            pp[0] = a0.x; pp[1] = a0.y; pp[2] = a0.z; pp[3] = a0.w;  gg[0] = a1.x; gg[1] = a1.y; gg[2] = a1.z; gg[3] = a1.w;
            mm[0] = a2.x; mm[1] = a2.y; mm[2] = a2.z; mm[3] = a2.w;  vv[0] = a3.x; vv[1] = a3.y; vv[2] = a3.z; vv[3] = a3.w;
        } else {
            for (int q = 0; q < 4; ++q) { const bool in = q < cnt; pp[q] = in ? p[i + q] : 0.f; gg[q] = in ? g[i + q] : 0.f; mm[q] = in ? m[i + q] : 0.f; vv[q] = in ? v[i + q] : 0.f; }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float gq = gg[q] + wd * pp[q];
            mm[q] = b1 * mm[q] + (1.f - b1) * gq;
            vv[q] = b2 * vv[q] + (1.f - b2) * gq * gq;
            const float denom = sqrtf(vv[q]) / bc2s + eps;
            pp[q] = pp[q] - step_size * (mm[q] / denom);
        }
        if (vec && cnt == 4) {
            *reinterpret_cast<float4*>(p + i) = make_float4(pp[0], pp[1], pp[2], pp[3]);
            *reinterpret_cast<float4*>(m + i) = make_float4(mm[0], mm[1], mm[2], mm[3]);
            *reinterpret_cast<float4*>(v + i) = make_float4(vv[0], vv[1], vv[2], vv[3]);
        } else {
            for (int q = 0; q < cnt; ++q) { p[i + q] = pp[q]; m[i + q] = mm[q]; v[i + q] = vv[q]; }
        }
    }
}

__global__ void adam_advance_kernel(float* step) { step[0] += 1.f; }

struct CopyKArgs { lbbnn_copy_list_t l; int first[LBBNN_ADAM_MAX_TENSORS + 1]; };

__global__ __launch_bounds__(256) void multi_copy_kernel(const CopyKArgs ka) {
    const LBBNN_CONST_AS CopyKArgs& a = *kernarg_as<CopyKArgs>();
    const int blk = blockIdx.x;
    int lo = 0, hi = a.l.n;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (a.first[mid] <= blk) lo = mid; else hi = mid; }
    float* __restrict__ d = a.l.dst[lo];
    const float* __restrict__ sp = a.l.src[lo];
    const int64_t n = a.l.numel[lo], base = (int64_t)(blk - a.first[lo]) * CHUNK;
    const bool vec = ((reinterpret_cast<uintptr_t>(d) | reinterpret_cast<uintptr_t>(sp)) & 15u) == 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int64_t i = base + (int64_t)(threadIdx.x + 256 * k) * 4;
        if (i >= n) break;
        if (vec && i + 4 <= n) {
            *reinterpret_cast<float4*>(d + i) = sp ? *reinterpret_cast<const float4*>(sp + i) : make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
            for (int64_t q = i; q < n && q < i + 4; ++q) d[q] = sp ? sp[q] : 0.f;
        }
    }
}

}  // namespace

extern "C" int lbbnn_adam_step(const lbbnn_adam_list_t* list, float lr, float beta1, float beta2, float eps, float weight_decay,
                               float* step, int advance, void* stream) {
    if (!list || !step) return LBBNN_E_NULL;
    if (list->n < 0 || list->n > LBBNN_ADAM_MAX_TENSORS) return LBBNN_E_SHAPE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (list->n > 0) {
        AdamKArgs ka;
        ka.l = *list;
        int nb = 0;
        for (int i = 0; i < list->n; ++i) {
            if (!list->p[i] || !list->g[i] || !list->m[i] || !list->v[i]) return LBBNN_E_NULL;
            if (list->numel[i] <= 0) return LBBNN_E_SHAPE;
            ka.first[i] = nb;
            nb += (int)((list->numel[i] + CHUNK - 1) / CHUNK);
        }
        for (int i = list->n; i <= LBBNN_ADAM_MAX_TENSORS; ++i) ka.first[i] = nb;
        ka.lr = lr; ka.b1 = beta1; ka.b2 = beta2; ka.eps = eps; ka.wd = weight_decay; ka.step = step;
        hipLaunchKernelGGL(adam_kernel, dim3(nb), dim3(256), 0, s, ka);
    }
    if (advance) hipLaunchKernelGGL(adam_advance_kernel, dim3(1), dim3(1), 0, s, step);
    return (int)hipGetLastError();
}

extern "C" int lbbnn_multi_copy(const lbbnn_copy_list_t* list, void* stream) {
    if (!list) return LBBNN_E_NULL;
    if (list->n < 0 || list->n > LBBNN_ADAM_MAX_TENSORS) return LBBNN_E_SHAPE;
    if (list->n == 0) return 0;
    CopyKArgs ka;
    ka.l = *list;
    int nb = 0;
    for (int i = 0; i < list->n; ++i) {
        if (!list->dst[i]) return LBBNN_E_NULL;
        if (list->numel[i] <= 0) return LBBNN_E_SHAPE;
        ka.first[i] = nb;
        nb += (int)((list->numel[i] + CHUNK - 1) / CHUNK);
    }
    for (int i = list->n; i <= LBBNN_ADAM_MAX_TENSORS; ++i) ka.first[i] = nb;
    hipLaunchKernelGGL(multi_copy_kernel, dim3(nb), dim3(256), 0, static_cast<hipStream_t>(stream), ka);
    return (int)hipGetLastError();
}
