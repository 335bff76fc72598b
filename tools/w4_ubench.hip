// What does a wave ALONE on its SIMD pay for the non-MFMA instructions of the 4-wave conv kernel (conv_dma4w_kernel.h)?  (not part of the library)
// One workgroup of 4 waves per CU (160 KB of LDS), every wave: ITER x [192 tied-accumulator MFMAs + fillers], fillers by mode:
//   0  none                                   1  NP LDS-DMA pieces, one behind every 8th MFMA, m0 saved / set / restored per piece (the library's dma16)
//   2  the same, m0 set per piece, no save / restore            3  m0 set once per FOUR pieces, the pieces 1 KB apart through the instruction offset
//   4  34 ds_read_b128 (18 up front, 4 per group)               5  modes 1 + 4          6  modes 3 + 4
// Prints cycles per 192-MFMA sub-stage (ideal 192 x 16 = 3072) from s_memtime (100 MHz ... the constant clock) and wall time.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/w4_ubench.hip -o tools/abl_w4_ubench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int NP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k(const char* src, unsigned bytes, int iters, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned long long v = (unsigned long long)src;
    const i32x4 q = {(int)(unsigned)v, (int)((unsigned)(v >> 32) & 0xFFFFu), (int)bytes, 0x00020000};
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + wave * (32 * 1024);
    const unsigned voff = (unsigned)((lane >> 2) * 64 + (lane & 3) * 16 + wave * 4096 + (blockIdx.x & 63) * 16384);
    f32x4 acc[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    u32x4 fa[10], fb[2][4];
#pragma unroll
    for (int i = 0; i < 10; ++i) fa[i] = u32x4{(unsigned)lane + i, 1u, 2u, 3u};
#pragma unroll
    for (int i = 0; i < 8; ++i) fb[i >> 2][i & 3] = u32x4{(unsigned)lane * 3 + i, 5u, 6u, 7u};
    const char* rd = smem + wave * (32 * 1024) + lane * 16;
    for (int it = 0; it < iters; ++it) {
        if (MODE >= 4) {
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[0][j] = *(const u32x4*)(rd + j * 1024);
#pragma unroll
            for (int r = 0; r < 10; ++r) fa[r] = *(const u32x4*)(rd + 8192 + r * 1024);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 6; ++g) {
            if (MODE >= 4 && g < 5) {
#pragma unroll
                for (int j = 0; j < 4; ++j) fb[(g + 1) & 1][j] = *(const u32x4*)(rd + (g + 1) * 4096 + j * 1024);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i * 8 + (g & 1) * 4 + j]) : "v"(fb[g & 1][j]), "v"(fa[i + (g >> 1)]));
                    const int n = g * 32 + i * 4 + j;
                    const int kk = (n - 4) >> 3;
                    const bool dma = (MODE == 1 || MODE == 2 || MODE == 3 || MODE == 5 || MODE == 6) && (n & 7) == 4 && kk < NP;
                    if (dma) {
                        const int soff = (it & 15) * 1024;
                        if (MODE == 1 || MODE == 5) {
                            unsigned keep;
                            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                                         : "=&s"(keep) : "v"(voff + kk * 1024), "s"(lds0 + kk * 1024), "s"(q), "s"(soff) : "memory");
                        } else if (MODE == 2) {
                            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, %3 offen lds"
                                         :: "v"(voff + kk * 1024), "s"(lds0 + kk * 1024), "s"(q), "s"(soff) : "memory");
                        } else {
                            // m0 once per four pieces; the instruction offset moves the LDS address AND the memory address by 1 KB per piece
                            if ((kk & 3) == 0) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" :: "s"(lds0 + kk * 1024) : "memory");
                            if ((kk & 3) == 0) asm volatile("buffer_load_dwordx4 %0, %1, %2 offen lds" :: "v"(voff + kk * 1024), "s"(q), "s"(soff) : "memory");
                            if ((kk & 3) == 1) asm volatile("buffer_load_dwordx4 %0, %1, %2 offen offset:1024 lds" :: "v"(voff + (kk - 1) * 1024), "s"(q), "s"(soff) : "memory");
                            if ((kk & 3) == 2) asm volatile("buffer_load_dwordx4 %0, %1, %2 offen offset:2048 lds" :: "v"(voff + (kk - 2) * 1024), "s"(q), "s"(soff) : "memory");
                            if ((kk & 3) == 3) asm volatile("buffer_load_dwordx4 %0, %1, %2 offen offset:3072 lds" :: "v"(voff + (kk - 3) * 1024), "s"(q), "s"(soff) : "memory");
                        }
                    }
                }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (MODE != 0 && MODE != 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 64; ++i) t += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (sink && t == 123.456f) sink[blockIdx.x] = t;
}

template <int MODE, int NP>
static void run(const char* src, unsigned bytes, const char* what) {
    const int iters = 2000, grid = 256;
    auto kern = k<MODE, NP>;
    const int lds = 160 * 1024;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, src, bytes, iters, (float*)nullptr);
    CK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int r = 0; r < 3; ++r) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, src, bytes, iters, (float*)nullptr);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
    }
    const double us_per = best * 1e3 / iters;
    printf("mode %d NP %2d %-58s %7.3f us per sub-stage = %6.0f cycles at 2.4 GHz (ideal 3072)  %6.0f TFLOP/s\n", MODE, NP, what, us_per, us_per * 2400.0,
           256.0 * 4 * 192 * 2.0 * 16 * 16 * 32 / us_per / 1e6);
}

int main() {
    char* src; const unsigned bytes = 8u << 20;
    CK(hipMalloc(&src, bytes)); CK(hipMemset(src, 1, bytes));
    run<0, 0>(src, bytes, "MFMAs only");
    run<1, 12>(src, bytes, "+ 12 DMA pieces, m0 save / set / restore");
    run<1, 18>(src, bytes, "+ 18 DMA pieces, m0 save / set / restore");
    run<2, 12>(src, bytes, "+ 12 DMA pieces, m0 set only");
    run<3, 12>(src, bytes, "+ 12 DMA pieces, m0 once per 4 (instruction offset)");
    run<3, 16>(src, bytes, "+ 16 DMA pieces, m0 once per 4 (instruction offset)");
    run<4, 0>(src, bytes, "+ 34 ds_read_b128");
    run<5, 12>(src, bytes, "+ 34 ds_read_b128 + 12 DMA pieces (save / restore)");
    run<6, 12>(src, bytes, "+ 34 ds_read_b128 + 12 DMA pieces (m0 once per 4)");
    return 0;
}
