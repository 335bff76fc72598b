// LDS-DMA (buffer_load_dwordx4 ... lds) throughput per CU as a function of the lane -> global address pattern of a 1 KB piece
// (not part of the library).  Each wave keeps DEPTH pieces in flight; the source region is small enough to stay in L2.
//   V0: 16 rows x 64 B (the conv kernels' 32-channel slabs), row stride RSTR bytes      V1: 8 rows x 128 B      V2: 4 rows x 256 B
//   V3: 1 KB contiguous      V4: V0 with every second row outside the buffer      V5: every lane outside the buffer
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/dma_ubench.hip -o tools/abl_dma_ubench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int V, int DEPTH>
__global__ __launch_bounds__(256, 2) void k(const char* src, unsigned bytes, int rstr, int iters, int region_rows, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned long long v = (unsigned long long)src;
    const i32x4 q = {(int)(unsigned)v, (int)((unsigned)(v >> 32) & 0xFFFFu), (int)bytes, 0x00020000};
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + wave * (DEPTH * 1024);
    constexpr unsigned OOB = 0xFFFF0000u;
    int row, col;
    if (V == 0 || V == 4) { row = lane >> 2; col = (lane & 3) * 16; }
    else if (V == 1) { row = lane >> 3; col = (lane & 7) * 16; }
    else if (V == 2) { row = lane >> 4; col = (lane & 15) * 16; }
    else { row = 0; col = lane * 16; }
    unsigned voff = (unsigned)(row * rstr + col);
    if (V == 4 && (row & 1)) voff = OOB;
    if (V == 5) voff = OOB;
    const int rows_per_piece = V == 0 || V == 4 ? 16 : V == 1 ? 8 : V == 2 ? 4 : 1;
    // every wave of every workgroup walks the same region (L2 hits) from its own start
    int r = ((blockIdx.x * 4 + wave) * 37) % region_rows;
    int kcol = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int soff = r * rstr + kcol;
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(voff), "s"(lds0 + d * 1024), "s"(q), "s"(soff) : "memory");
            r += rows_per_piece; if (r + rows_per_piece > region_rows) { r = 0; kcol = (kcol + 1024) % rstr; if (V != 3 && kcol + 256 > rstr) kcol = 0; }
        }
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH / 2) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (sink && threadIdx.x == 0 && iters < 0) sink[blockIdx.x] = *(float*)smem;
}

template <int V, int DEPTH>
static void run(const char* src, unsigned bytes, int rstr, int region_rows, const char* what) {
    const int iters = 400, grid = 512;
    auto kern = k<V, DEPTH>;
    const int lds = 4 * DEPTH * 1024 < 70 * 1024 ? 70 * 1024 : 4 * DEPTH * 1024;     // two workgroups per CU
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, src, bytes, rstr, iters, region_rows, (float*)nullptr);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, src, bytes, rstr, iters, region_rows, (float*)nullptr);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double pieces_per_cu = 5.0 * iters * DEPTH * 4 * 2;       // 2 workgroups x 4 waves per CU
    const double ns = ms * 1e6 / pieces_per_cu;
    printf("V%d depth %2d rstr %5d rows %5d %-34s: %6.2f ns per piece per CU (%5.1f cyc @2.1 GHz, %5.1f B/clk/CU, %5.2f TB/s chip)\n", V, DEPTH, rstr, region_rows, what, ns, ns * 2.1,
           1024.0 / (ns * 2.1), 1024.0 * 256 / ns / 1e3);
}

int main() {
    const unsigned bytes = 64u << 20;
    char* src; CK(hipMalloc(&src, bytes)); CK(hipMemset(src, 1, bytes));
    // region: region_rows x rstr bytes; 1024 rows x 1536 B = 1.5 MB (L2-resident per XCD)
    run<0, 8>(src, bytes, 1536, 1024, "16 rows x 64 B");
    run<1, 8>(src, bytes, 1536, 1024, "8 rows x 128 B");
    run<2, 8>(src, bytes, 1536, 1024, "4 rows x 256 B");
    run<3, 8>(src, bytes, 1536, 1024, "1 KB contiguous");
    run<4, 8>(src, bytes, 1536, 1024, "16 rows x 64 B, odd rows outside");
    run<5, 8>(src, bytes, 1536, 1024, "all lanes outside");
    run<0, 4>(src, bytes, 1536, 1024, "16 rows x 64 B");
    run<0, 16>(src, bytes, 1536, 1024, "16 rows x 64 B");
    run<1, 16>(src, bytes, 1536, 1024, "8 rows x 128 B");
    run<3, 16>(src, bytes, 1536, 1024, "1 KB contiguous");
    run<0, 8>(src, bytes, 1536, 8192, "16 rows x 64 B, 12 MB region");
    run<1, 8>(src, bytes, 1536, 8192, "8 rows x 128 B, 12 MB region");
    run<0, 8>(src, bytes, 256, 4096, "16 rows x 64 B, 256 B rows");
    return 0;
}
