// What a kernel boundary costs (not part of the library): back-to-back launches of an empty kernel on one stream, by grid / block / dynamic LDS size, and
// with a trailing store per thread (the end-of-kernel write-back has something to do).   build: hipcc --offload-arch=gfx950 -O3 tools/launch_ubench.hip -o tools/abl_launch
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
struct Args { float* p; int store; char pad[400]; };          // a ConvArgs-sized kernarg segment
__global__ void empty_kernel(const Args a) {
    extern __shared__ char smem[];
    if (a.store) a.p[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = 1.0f;
    if (a.store == 77) smem[threadIdx.x] = 1;
}
int main() {
    float* p; CK(hipMalloc(&p, (size_t)1 << 28));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute((const void*)empty_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int cfg[][4] = {{256, 512, 160, 0}, {256, 512, 0, 0}, {256, 512, 76, 0}, {512, 256, 76, 0}, {256, 64, 0, 0}, {2048, 64, 0, 0}, {1024, 512, 160, 0}, {1, 64, 0, 0},
                          {256, 512, 160, 1}, {2048, 64, 0, 1}, {16384, 256, 0, 1}};
    for (auto& c : cfg) {
        Args a{}; a.p = p; a.store = c[3];
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(empty_kernel, dim3(c[0]), dim3(c[1]), c[2] * 1024, 0, a);
        CK(hipDeviceSynchronize());
        const int it = 200;
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < it; ++i) hipLaunchKernelGGL(empty_kernel, dim3(c[0]), dim3(c[1]), c[2] * 1024, 0, a);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        // the same 200 launches as ONE hipGraph (stream capture): does the replayed graph shorten the boundary between dependent kernels?
        hipStream_t st; CK(hipStreamCreate(&st));
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        for (int i = 0; i < it; ++i) hipLaunchKernelGGL(empty_kernel, dim3(c[0]), dim3(c[1]), c[2] * 1024, st, a);
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float msg; CK(hipEventElapsedTime(&msg, e0, e1));
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g)); CK(hipStreamDestroy(st));
        printf("grid %5d x %3d threads, LDS %3d KB, store %d : %6.2f us per launch, %6.2f as a graph\n", c[0], c[1], c[2], c[3], ms / it * 1e3, msg / it * 1e3);
    }
    return 0;
}
