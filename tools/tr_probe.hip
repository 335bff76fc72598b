// What ds_read_b64_tr_b16 returns (gfx950): LDS holds u16 element e at index e; every lane passes its own byte address; prints the four u16 each lane gets.
// build: hipcc --offload-arch=gfx950 -O2 tools/tr_probe.hip -o tools/abl_tr_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void probe(unsigned short* out, int mode) {
    __shared__ unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int l = threadIdx.x;
    // mode 0: the guide's layout: lane address = ((l & 15) + (l >> 4) * 64) elements?  mode 1: [row][16 cols] rows of 64 elements: lane i -> row (i>>2), cols 4*(i&3); group g -> rows 4g..
    unsigned addr;
    if (mode == 0) addr = (unsigned)(((l & 15) * 4 + (l >> 4) * 64) * 2);
    else addr = (unsigned)((((l >> 4) * 4 + ((l & 15) >> 2)) * 64 + (l & 3) * 4) * 2);
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned short*)lds;
    unsigned long long v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(base + addr) : "memory");
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)(v >> (16 * j));
}
int main() {
    unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
    unsigned short h[256];
    for (int mode = 0; mode < 2; ++mode) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, mode);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) { printf("lane %2d:", l); for (int j = 0; j < 4; ++j) printf(" %5d", h[l * 4 + j]); printf("%s", (l & 3) == 3 ? "\n" : "   "); }
    }
    return 0;
}
