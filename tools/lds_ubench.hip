// LDS read bandwidth per CU: every wave streams ds_read_b128 over a conflict-free 64-byte-row image (the conv kernel's A/B layout)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
template <int WIDTH>
__global__ __launch_bounds__(512) void k(unsigned long long* out, float* sink, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) ((float*)smem)[i] = i;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane & 15, u = lane >> 4;
    const int off = (wave * 64 + q) * 64 + ((u ^ ((q >> 1) & 2)) << 4);
    float acc = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            typedef unsigned u4 __attribute__((ext_vector_type(4)));
            typedef unsigned u2 __attribute__((ext_vector_type(2)));
            if (WIDTH == 16) { u4 v; asm volatile("ds_read_b128 %0, %1 offset:0" : "=v"(v) : "v"(off + (r & 3) * 1024)); asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory"); if (r == 15) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v)); acc += __uint_as_float(v.x ^ v.w); } }
            else { u2 v; asm volatile("ds_read_b64 %0, %1 offset:0" : "=v"(v) : "v"(off + (r & 3) * 1024)); asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory"); if (r == 15) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v)); acc += __uint_as_float(v.x ^ v.y); } }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    sink[blockIdx.x * 512 + threadIdx.x] = acc;
    if (lane == 0 && blockIdx.x == 0) out[wave] = t1 - t0;
}
int main() {
    unsigned long long* d; float* sink;
    CK(hipMalloc(&d, 64)); CK(hipMalloc(&sink, 256 * 512 * 4));
    const int iters = 1000;
    for (int nw = 1; nw <= 8; nw *= 2) {
        hipLaunchKernelGGL(k<16>, dim3(256), dim3(64 * nw), 65536, 0, d, sink, iters); CK(hipDeviceSynchronize());
        unsigned long long h[8]; CK(hipMemcpy(h, d, 64, hipMemcpyDeviceToHost));
        printf("b128 %d waves: %.1f ticks per ds_read_b128 per wave -> %.1f B/tick/CU\n", nw, (double)h[0] / iters / 16, nw * 1024.0 * 16 * iters / h[0]);
        hipLaunchKernelGGL(k<8>, dim3(256), dim3(64 * nw), 65536, 0, d, sink, iters); CK(hipDeviceSynchronize());
        CK(hipMemcpy(h, d, 64, hipMemcpyDeviceToHost));
        printf("b64  %d waves: %.1f ticks per ds_read_b64 per wave -> %.1f B/tick/CU\n", nw, (double)h[0] / iters / 16, nw * 512.0 * 16 * iters / h[0]);
    }
    return 0;
}
