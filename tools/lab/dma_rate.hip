// Lab (GPU box): how fast can ONE CU take L2-resident operand tiles in, by the access shape of the split-precision GEMM?
// Every wave issues pieces of 1 KiB (64 lanes x 16 B = 8 rows x 128 B at a row stride of RS bytes, the next piece 128 B further
// along the row: the GEMM's K loop), either as LDS-DMA (buffer_load_dwordx4 ... lds) into a per-wave ring in LDS, or as
// global_load_dwordx4 into registers.  Nothing consumes the data; a counted s_waitcnt vmcnt(INFL) after every issue caps the
// pieces a wave keeps in flight.  Workgroup w reads tile (w % T) of 128 rows, so tiles are shared between workgroups and stay
// in the XCD's L2 as the GEMM's x / weight tiles do.
// Build: hipcc --offload-arch=gfx950 -O3 tools/lab/dma_rate.hip -o tools/lab/dma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }

// MODE 0: LDS-DMA; MODE 1: loads to registers (U in flight = INFL)
template <int MODE, int WAVES, int INFL>
__device__ __forceinline__ void fill_body(const char* src, int rs, int ksteps, int iters, int T, unsigned* sink,
                                          unsigned long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) char smc[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = blockIdx.x % T;
    constexpr int PPS = 16 / (WAVES / 4 > 0 ? 1 : 1);      // pieces of one K step of a 128-row tile
    // wave wv issues pieces g = wv, wv + WAVES, ... of the 16 of a step (rows 8g .. 8g+7)
    constexpr int NPW = (PPS + WAVES - 1) / WAVES;
    const unsigned bytes = 0x7FFFFFF0u;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)bytes, 0x00020000);
    int off[NPW];
#pragma unroll
    for (int u = 0; u < NPW; ++u) {
        const int g = (wv + WAVES * u) % PPS;
        const int row = tile * 128 + 8 * g + (lane >> 3);
        off[u] = row * rs + 16 * (lane & 7);
    }
    const unsigned long long t0 = __builtin_readcyclecounter();
    unsigned acc = 0;
    if (MODE == 2 || MODE == 3 || MODE == 4 || MODE == 5 || MODE == 6) {
        // waves with odd SIMD-partner index run back-to-back MFMAs (matrix pipe busy on every SIMD), the others issue the pieces:
        // MODE 2 per-lane offsets in a VGPR (offen), MODE 3 no address VGPR at all (descriptor ADD_TID_ENABLE, stride 16: lane i
        // reads 16 B at base + soffset + 16 i -- a contiguous 1-KiB piece), MODE 4 = MODE 2 with the MFMA waves idle (control)
        typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
        typedef float floatx4 __attribute__((ext_vector_type(4)));
        const bool mf = wv >= WAVES / 2;
        if (mf && (MODE == 5 || MODE == 6)) {
            // partner waves stream ds_read_b128 (MODE 5: all the time, 24 reads then a wait -- the GEMM's fragment reads; MODE 6:
            // plus 60 MFMAs after every 24 reads, the GEMM's ratio)
            typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
            typedef float floatx4 __attribute__((ext_vector_type(4)));
            typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
            const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smc;
            const unsigned ra = lds0 + (lane & 15) * 128 + 16 * (lane >> 4);
            floatx4 c0 = {0, 0, 0, 0}, c1 = c0;
            bf16x8 a, b;
            for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(lane + i); b[i] = (__bf16)(float)(lane ^ i); }
            const int n = iters * ksteps * (MODE == 6 ? 1 : 3);
            for (int i = 0; i < n; ++i) {
                u32x4 r[24];
#pragma unroll
                for (int k = 0; k < 24; ++k) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r[k]) : "v"(ra), "i"((k % 16) * 2048) : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int k = 0; k < 24; ++k) asm volatile("" :: "v"(r[k]));
                if (MODE == 6) {
#pragma unroll
                    for (int k = 0; k < 30; ++k) {
                        c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
                        c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
                    }
                }
            }
            acc = __float_as_uint(c0[0] + c1[1]);
        } else if (mf) {
            if (MODE != 4) {
                floatx4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
                bf16x8 a, b;
                for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(lane + i); b[i] = (__bf16)(float)(lane ^ i); }
                const int n = iters * ksteps * 16 * 2;       // 16x16x32 MFMAs, 16 cycles each: about as long as the other waves' loop
                for (int i = 0; i < n; i += 4) {
                    c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
                    c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0);
                    c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0);
                }
                acc = __float_as_uint(c0[0] + c1[1] + c2[2] + c3[3]);
            }
        } else {
            constexpr int HW = WAVES / 2 > 0 ? WAVES / 2 : 1; constexpr int NPW2 = (16 + HW - 1) / HW;
            int off2[NPW2];
#pragma unroll
            for (int u = 0; u < NPW2; ++u) {
                const int g = (wv + HW * u) % 16;
                off2[u] = (tile * 128 + 8 * g + (lane >> 3)) * rs + 16 * (lane & 7);
            }
            // ADD_TID_ENABLE = bit 23 of dword 3, stride (dword 1 bits 29:16) = 16
            // (with ADD_TID_ENABLE the DATA_FORMAT bits of dword 3 extend the stride: they must be 0 here)
            const __amdgpu_buffer_rsrc_t rtid = __builtin_amdgcn_make_buffer_rsrc((void*)src, 16, (int)bytes, (1 << 23));
            int slot = 0;
            for (int it = 0; it < iters; ++it)
                for (int c = 0; c < ksteps; ++c) {
#pragma unroll
                    for (int u = 0; u < NPW2; ++u) {
                        auto* dst = (__attribute__((address_space(3))) void*)(smc + (wv * (INFL + 1) + slot) * 1024);
                        if (MODE == 3) __builtin_amdgcn_raw_ptr_buffer_load_lds(rtid, dst, 16, 0, (tile * 128 + 8 * u) * rs + c * 1024, 0, 0);
                        else           __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, dst, 16, off2[u], c * 128, 0, 0);
                        slot = slot == INFL ? 0 : slot + 1;
                        wait_vmcnt<INFL>();
                    }
                }
            wait_vmcnt<0>();
        }
    } else if (MODE == 0) {
        int slot = 0;
        for (int it = 0; it < iters; ++it)
            for (int c = 0; c < ksteps; ++c) {
#pragma unroll
                for (int u = 0; u < NPW; ++u) {
                    auto* dst = (__attribute__((address_space(3))) void*)(smc + (wv * (INFL + 1) + slot) * 1024);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, dst, 16, off[u], c * 128, 0, 0);
                    slot = slot == INFL ? 0 : slot + 1;
                    wait_vmcnt<INFL>();
                }
            }
        wait_vmcnt<0>();
    } else {
        const char* base = src;
        for (int it = 0; it < iters; ++it)
            for (int c = 0; c < ksteps; c += (INFL / NPW > 0 ? INFL / NPW : 1)) {
                uint4 r[INFL];
#pragma unroll
                for (int i = 0; i < INFL; ++i) {
                    const int cc = c + i / NPW, u = i % NPW;
                    r[i] = *reinterpret_cast<const uint4*>(base + off[u] + (cc < ksteps ? cc : 0) * 128);
                }
#pragma unroll
                for (int i = 0; i < INFL; ++i) acc ^= r[i].x ^ r[i].y ^ r[i].z ^ r[i].w;
            }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (acc == 0x12345u) sink[0] = acc;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;     // wave 0 is a DMA wave in every mode
}

template <int MODE, int WAVES, int INFL>
__global__ __launch_bounds__(WAVES * 64) void fill_k(const char* src, int rs, int ksteps, int iters, int T, unsigned* sink,
                                                     unsigned long long* cyc) {
    fill_body<MODE, WAVES, INFL>(src, rs, ksteps, iters, T, sink, cyc);
}

template <int MODE, int WAVES, int INFL>
static void run(const char* name, const char* src, int rs, int ksteps, int iters, int T, int nwg, unsigned* sink,
                unsigned long long* cyc) {
    const size_t lds = MODE != 1 ? ((size_t)WAVES * (INFL + 1) * 1024 < 36864 ? 36864 : (size_t)WAVES * (INFL + 1) * 1024) : 0;
    if (lds > 64 * 1024) hipFuncSetAttribute(reinterpret_cast<const void*>(&fill_k<MODE, WAVES, INFL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) fill_k<MODE, WAVES, INFL><<<dim3(nwg), dim3(WAVES * 64), lds, 0>>>(src, rs, ksteps, iters, T, sink, cyc);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        fill_k<MODE, WAVES, INFL><<<dim3(nwg), dim3(WAVES * 64), lds, 0>>>(src, rs, ksteps, iters, T, sink, cyc);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    std::vector<unsigned long long> h(nwg);
    hipMemcpy(h.data(), cyc, nwg * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double cs = 0; for (auto v : h) cs += (double)v; cs /= nwg;
    constexpr int NPW = MODE >= 2 ? (16 + WAVES / 2 - 1) / (WAVES / 2) / 2 : (16 + WAVES - 1) / WAVES;   // MODE >= 2: half the waves issue
    constexpr int SPB = (INFL / NPW > 0 ? INFL / NPW : 1);       // MODE 1 issues whole batches of SPB K steps
    const int ksi = MODE == 1 ? (ksteps + SPB - 1) / SPB * SPB : ksteps;
    const double pieces_wg = (double)WAVES * NPW * ksi * iters;
    const double bytes = pieces_wg * 1024.0 * nwg;
    const double us = best * 1e3;
    printf("%-34s nwg %4d T %3d | %8.1f us | %6.1f GB/s per CU (256) | %5.2f TB/s chip | in-kernel: %7.0f cycles, %5.1f cycles per piece per WG, %5.1f B/clk per WG\n",
           name, nwg, T, us, bytes / 256.0 / us * 1e-3, bytes / us * 1e-6, cs, cs / pieces_wg, pieces_wg * 1024.0 / cs);
}

int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int rs = 4800, ksteps = 37, rows = 4096;
    char* src; unsigned* sink; unsigned long long* cyc;
    hipMalloc(&src, (size_t)rows * rs + 4096);
    hipMemset(src, 1, (size_t)rows * rs + 4096);
    hipMalloc(&sink, 64); hipMalloc(&cyc, 4096 * 8);
    const int iters = 8;
    for (int T : {32, 4}) {
        printf("--- tiles shared: T = %d (footprint %.1f MB)\n", T, T * 128.0 * rs * 1e-6);
        run<0, 4, 3>("LDS-DMA 4 waves, 4 in flight/wave", src, rs, ksteps, iters, T, 256, sink, cyc);
        run<0, 4, 8>("LDS-DMA 4 waves, 9 in flight/wave", src, rs, ksteps, iters, T, 256, sink, cyc);
        run<0, 4, 15>("LDS-DMA 4 waves, 16 in flight/wave", src, rs, ksteps, iters, T, 256, sink, cyc);
        run<0, 4, 8>("LDS-DMA 2 WG/CU x 4 waves, 9 infl", src, rs, ksteps, iters, T, 512, sink, cyc);
        run<0, 4, 15>("LDS-DMA 2 WG/CU x 4 waves, 16 infl", src, rs, ksteps, iters, T, 512, sink, cyc);
        run<0, 8, 8>("LDS-DMA 8 waves, 9 in flight/wave", src, rs, ksteps, iters, T, 256, sink, cyc);
        run<0, 8, 15>("LDS-DMA 8 waves, 16 in flight/wave", src, rs, ksteps, iters, T, 256, sink, cyc);
        run<0, 1, 15>("LDS-DMA 1 wave, 16 in flight", src, rs, ksteps, iters, T, 256, sink, cyc);
        run<0, 2, 15>("LDS-DMA 2 waves, 16 in flight/wave", src, rs, ksteps, iters, T, 256, sink, cyc);
        run<4, 8, 15>("LDS-DMA 4 of 8 waves, others idle", src, rs, ksteps, iters, T, 256, sink, cyc);
        run<2, 8, 15>("LDS-DMA 4 of 8 waves, others MFMA", src, rs, ksteps, iters, T, 256, sink, cyc);
        run<3, 8, 15>("same, ADD_TID (no address VGPR)", src, rs, ksteps, iters, T, 256, sink, cyc);
        run<5, 8, 15>("LDS-DMA 4 of 8 waves, others ds_read_b128", src, rs, ksteps, iters, T, 256, sink, cyc);
        run<6, 8, 15>("LDS-DMA 4 of 8, others 24 reads + 60 MFMA", src, rs, ksteps, iters, T, 256, sink, cyc);
        run<1, 4, 8>("regs 4 waves, 8 in flight/wave", src, rs, ksteps, iters, T, 256, sink, cyc);
        run<1, 4, 16>("regs 4 waves, 16 in flight/wave", src, rs, ksteps, iters, T, 256, sink, cyc);
        run<1, 8, 16>("regs 8 waves, 16 in flight/wave", src, rs, ksteps, iters, T, 256, sink, cyc);
        run<1, 4, 16>("regs 2 WG/CU x 4 waves, 16 infl", src, rs, ksteps, iters, T, 512, sink, cyc);
        run<1, 4, 16>("regs 4 WG/CU x 4 waves, 16 infl", src, rs, ksteps, iters, T, 1024, sink, cyc);
    }
    return 0;
}
