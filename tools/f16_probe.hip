// Probe (GPU box): does v_mfma_f32_16x16x32_f16 honour fp16 SUBNORMAL inputs, and which (row, k) does a lane's
// operand element feed?  Prints one line per check.  Build: hipcc --offload-arch=gfx950 -O2 tools/f16_probe.hip -o tools/f16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void probe(const _Float16* A, const _Float16* B, float* D) {
    // A: [16 m][32 k], B: [16 n][32 k] (both k-contiguous).  lane: row = lane%16, k group = lane/16 -> k = 8*(lane/16) .. +7
    const int lane = threadIdx.x, r = lane & 15, q = lane >> 4;
    h8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = A[r * 32 + 8 * q + j]; b[j] = B[r * 32 + 8 * q + j]; }
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
    // D[m][n]: lane holds m = 4*(lane/16) + j, n = lane%16
    for (int j = 0; j < 4; ++j) D[(4 * q + j) * 16 + r] = acc[j];
}

__global__ void cvt(const float* x, _Float16* hi, _Float16* lo, int n) {
    int i = threadIdx.x;
    if (i < n) { _Float16 h = (_Float16)x[i]; hi[i] = h; lo[i] = (_Float16)(x[i] - (float)h); }
}

int main() {
    _Float16 hA[512], hB[512];
    float hD[256];
    _Float16 *dA, *dB; float* dD;
    hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dD, sizeof(hD));
    // 1) layout: A[m][k] = m + 100 k (small ints, exact), B = one-hot per n at k = n
    for (int m = 0; m < 16; ++m) for (int k = 0; k < 32; ++k) { hA[m * 32 + k] = (_Float16)(float)(m + 1 + 0.25f * k); hB[m * 32 + k] = (_Float16)(k == (m * 2 + 1) ? 1.f : 0.f); }
    hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
    probe<<<1, 64>>>(dA, dB, dD); hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) { float want = m + 1 + 0.25f * (2 * n + 1); if (hD[m * 16 + n] != want) ++bad; }
    printf("layout D[m][n] = sum_k A[m][k] B[n][k] with lane(row=l%%16, k=8*(l/16)+j): mismatches %d\n", bad);
    // 2) subnormals: A = 2^-20 (fp16 subnormal: 16 ulp of 2^-24), B = 1 -> D = 32 * 2^-20 = 2^-15
    for (int i = 0; i < 512; ++i) { hA[i] = (_Float16)9.5367431640625e-07f; hB[i] = (_Float16)1.0f; }
    hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
    probe<<<1, 64>>>(dA, dB, dD); hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost);
    printf("subnormal A (2^-20) x 1: D[0][0] = %.9g (exact product sum 2^-15 = %.9g; 0 means inputs flushed)\n", hD[0], 3.0517578125e-05);
    // both subnormal-ish: A = 2^-20, B = 2^-20 -> 32 * 2^-40
    for (int i = 0; i < 512; ++i) hB[i] = (_Float16)9.5367431640625e-07f;
    hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
    probe<<<1, 64>>>(dA, dB, dD); hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost);
    printf("subnormal x subnormal: D = %.9g (exact 32*2^-40 = %.9g)\n", hD[0], 32.0 * 9.094947017729282e-13);
    // 3) f32 -> f16 conversion keeps subnormals (hi / lo split of small numbers)
    float hx[4] = {3.0e-6f, 0.0123456789f, 1.0f / 3.0f, 6.0e-8f}, *dx; _Float16 *dh, *dl, hh[4], hl[4];
    hipMalloc(&dx, 16); hipMalloc(&dh, 8); hipMalloc(&dl, 8);
    hipMemcpy(dx, hx, 16, hipMemcpyHostToDevice);
    cvt<<<1, 64>>>(dx, dh, dl, 4);
    hipMemcpy(hh, dh, 8, hipMemcpyDeviceToHost); hipMemcpy(hl, dl, 8, hipMemcpyDeviceToHost);
    for (int i = 0; i < 4; ++i) printf("split %.9g -> hi %.9g lo %.9g  residual %.3g\n", hx[i], (float)hh[i], (float)hl[i], hx[i] - (float)hh[i] - (float)hl[i]);
    return 0;
}
