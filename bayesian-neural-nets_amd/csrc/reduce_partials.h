// Second level of the two-level column sums (lbbnn_output_grad: Sum_b G_m / G_v; lbbnn_weight_pass_backward: dz_fwd, dz_kl,
// dr0_c): out[q][i] = Sum_b work[b * block_stride + q * q_stride + i].  A 1024-thread workgroup owns 64 columns; its 16 waves
// each add every 16th block (coalesced 256-B rows), then waves 0..nq-1 add the 16 partials in a fixed order.  ONE body for
// the stand-alone launches and for lbbnn_reduce_partials_batch, so deferring the sums does not change a bit of them.
#pragma once
#include "lbbnn_device.h"
#include "../../include/lbbnn.h"

namespace lbbnn {

__device__ __forceinline__ void reduce_partials_body(const float* __restrict__ work, long long block_stride, long long q_stride,
                                                     int nblk, int ncols, int nq, float* const (&out)[3], int colblock,
                                                     float (&part)[3][16][64]) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int i = colblock * 64 + lane;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    if (i < ncols)
        for (int b = w; b < nblk; b += 16) {
            const float* p = work + (size_t)b * block_stride + i;
            if (out[0]) s0 += p[0];
            if (nq > 1 && out[1]) s1 += p[q_stride];
            if (nq > 2 && out[2]) s2 += p[2 * q_stride];
        }
    part[0][w][lane] = s0; part[1][w][lane] = s1; part[2][w][lane] = s2;
    __syncthreads();
    if (w < nq && i < ncols && out[w]) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += part[w][k][lane];
        out[w][i] = s;
    }
}

}  // namespace lbbnn
