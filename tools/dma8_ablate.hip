// Ablation micro-benchmark of the 8x8 LDS-DMA conv kernel (not part of the library): phases switched off by -DWDM_D8ABL=<mask>
// (conv_dma8_kernel.h: 2 MFMAs, 16 fragment reads with 2, 4 halo DMA, 8 weight DMA).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DWDM_D8ABL=<m> -DBN8=<48|64> -DNI8=<2|4> -I wavedm_amd/csrc -I include tools/dma8_ablate.hip -o tools/abl_dma8_<m>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "conv_dma8_kernel.h"
using namespace wdm;
#ifndef BN8
#define BN8 48
#endif
#ifndef NI8
#define NI8 2
#endif
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 64, Cin = argc > 2 ? atoi(argv[2]) : 768, Cout = argc > 3 ? atoi(argv[3]) : 768;
    const int H = 8;
    const size_t nx = (size_t)B * H * H * Cin, ny = (size_t)B * H * H * Cout, nw = (size_t)9 * Cout * Cin;
    unsigned short *x, *y, *w; float* bias;
    CK(hipMalloc(&x, nx * 2)); CK(hipMalloc(&y, ny * 2)); CK(hipMalloc(&w, nw * 2)); CK(hipMalloc(&bias, Cout * 4));
    std::vector<unsigned short> hx(nx), hw(nw);
    srand(1);
    for (auto& v : hx) v = (unsigned short)(0x3f00 + (rand() & 0xff) + ((rand() & 1) << 15));
    for (auto& v : hw) v = (unsigned short)(0x3c00 + (rand() & 0xff) + ((rand() & 1) << 15));
    CK(hipMemcpy(x, hx.data(), nx * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(w, hw.data(), nw * 2, hipMemcpyHostToDevice));
    CK(hipMemset(bias, 0, Cout * 4));
    ConvArgs a{};
    a.x0 = x; a.C0 = Cin; a.xs0 = Cin; a.B = B; a.Hin = a.Win = a.Hout = a.Wout = H; a.Cin = Cin; a.Cout = Cout;
    a.w = w; a.w_tap_stride = (long long)Cout * Cin; a.w_row_stride = Cin; a.w_rows = Cout; a.bias = bias; a.alpha = 1.f;
    a.y = y; a.y_mode = Y_NHWC; a.y_s = Cout;
    a.x0_bytes = (unsigned)(nx * 2); a.w_bytes = (unsigned)(nw * 2);
    if (getenv("SM")) { a.w_tap_stride = (long long)Cout * 32; a.w_row_stride = 32; a.w_slab_stride = 9 * Cout * 32; }      // slab-major weights
    using C = ConvDma8Cfg<BN8, NI8>;
    auto kern = conv_dma8_kernel<BN8, NI8>;
    a.mtiles = (B + NI8 - 1) / NI8; a.ntiles = (Cout + C::BN - 1) / C::BN;
    const int gn = getenv("GN") ? atoi(getenv("GN")) : 1, gm = 8 / gn;          // N-tile groups per XCD (conv_dispatch.inc: launch_dma8)
    a.grid_gn = gn;
    const int grid = 8 * ((a.ntiles + gn - 1) / gn) * ((a.mtiles + gm - 1) / gm);
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(C::NTHREADS), C::LDS_BYTES, 0, a);
    CK(hipDeviceSynchronize());
    const int it = 20;
    float ms;
    if (getenv("COLD")) {
        // COLD=<n>: rotate over n copies of the weight tensor (n x 10-21 MB: with n = 24 a copy has left L2 and most of the Infinity Cache when its turn
        // comes again -- the situation inside the UNet, where a layer's weights were last read one forward earlier).  PF=1: a streaming read of the NEXT
        // copy (one dword per 128-byte line) runs as its own launch before each conv (what a prefetch could buy, upper bound)
        const int nc = atoi(getenv("COLD"));
        std::vector<unsigned short*> ws(nc);
        for (int c = 0; c < nc; ++c) { CK(hipMalloc(&ws[c], nw * 2)); CK(hipMemcpy(ws[c], w, nw * 2, hipMemcpyDeviceToDevice)); }
        CK(hipDeviceSynchronize());
        float tot = 0.f;
        for (int i = 0; i < 3 * nc; ++i) {
            a.w = ws[i % nc];
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(kern, dim3(grid), dim3(C::NTHREADS), C::LDS_BYTES, 0, a);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float t; CK(hipEventElapsedTime(&t, e0, e1));
            if (i >= nc) tot += t;
        }
        ms = tot / (2 * nc) * it;
    } else {
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < it; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(C::NTHREADS), C::LDS_BYTES, 0, a);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    }
    const double fl = 2.0 * B * H * H * Cout * 9.0 * Cin;
    printf("D8ABL=%2d BN=%d NI=%d B=%d %d->%d grid=%d : %7.1f us  %7.1f TFLOP/s (nominal)\n", WDM_D8ABL, BN8, NI8, B, Cin, Cout, grid, ms / it * 1e3, fl / (ms / it) / 1e9);
    return 0;
}
