// Pieces shared by the GEMM translation units (lrt_gemm.hip, lrt_gemm_f16.hip): the XCD-aware tile map, the residency cap
// through the dynamic-LDS request, and the launch helper.  Everything here has internal linkage per translation unit.
#pragma once
#include <cstdlib>
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

// XCD-aware tile assignment (cdna guide T1).  Workgroups are dealt round-robin over the 8 XCDs, each with its own L2;
// with the natural order the 15 workgroups that share one x tile land on 8 different L2s.  The linear id is remapped
// so that an XCD owns a 2-D block of the tile grid: the b tiles are cut into 4 groups, each group's tiles (o-major
// order) into two halves -- XCD (bg, half) then fetches a QUARTER of the x tiles and HALF of the weight tiles into its
// L2 (fabric traffic 4 W + 2 x per launch; a contiguous b-major run per XCD, the round-1 map, streams ALL weight tiles
// through every L2: 8 W + x = 113 MB at the 1200 x 1200 layer against 86 MB).  Falls back to the b-major run when the
// b-tile count is not a multiple of 4, and to the identity when the grid is not a multiple of 8.  Bijective in both
// forms; speed only, never correctness.
__device__ __forceinline__ void tile_of_block(int& ox, int& by, int extra_rows = 0) {
    const int nx = gridDim.x, ny = gridDim.y - extra_rows, n = nx * ny;
    int t = blockIdx.y * nx + blockIdx.x;
    if ((n & 7) == 0 && (ny & 3) == 0) {
        const int xcd = t & 7, s = t >> 3, per = n >> 3, q = ny >> 2;      // per: tiles per XCD, q: b tiles per group
        const int idx = (xcd & 1) * per + s;                              // position inside the b group, o-major
        ox = idx / q;
        by = (xcd >> 1) * q + idx % q;
        return;
    }
    if ((n & 7) == 0) t = (t & 7) * (n >> 3) + (t >> 3);
    ox = t % nx; by = t / nx;
}

// Workgroup residency matters more than anything else here: the kernel is MFMA-issue bound, so a CU
// that receives 3 workgroups takes 3x as long as one that receives 1, and the hardware dispatcher
// packs as many as fit (LDS 46 KB and 164 VGPRs admit 3).  We therefore cap residency through the
// dynamic-LDS request so that `nblocks` spread evenly: r = ceil(nblocks / 256 CUs) per CU (<= 3).
constexpr size_t kLdsPerCU = 160 * 1024;
constexpr int kNumCU = 256;

inline size_t lds_request(size_t needed, long nblocks) {
    long r = (nblocks + kNumCU - 1) / kNumCU;
    static const char* const res_env = getenv("LBBNN_GEMM_RESIDENCY");   // tuning knob (bench sweeps only), read once
    if (res_env) r = atol(res_env);
    if (r < 1) r = 1;
    if (r >= 3) return needed;                       // as many as fit
    const size_t cap = kLdsPerCU / (size_t)(r + 1) + 256;   // r fit, r+1 do not
    return needed > cap ? needed : cap;
}

template <typename K, typename A>
inline int launch_one(K kernel, dim3 grid, dim3 block, size_t lds, hipStream_t s, const A& a) {
    if (lds > 64 * 1024) {
        // above 64 KB the dynamic-LDS limit of the function has to be raised (host-side attribute, not a stream
        // op); done once per kernel instantiation and size (this template is instantiated per kernel type K).
        static size_t raised = 0;
        if (lds > raised) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return (int)e;
            raised = lds;
        }
    }
    hipLaunchKernelGGL(kernel, grid, block, lds, s, a);
    return (int)hipGetLastError();
}


}  // namespace
