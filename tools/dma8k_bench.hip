// conv_dma8k_kernel.h (K-split 128 x 96 tile, eight waves) against conv_dma8_kernel.h (128 x 48, four waves, two workgroups per CU) on one 8 x 8 layer:
// outputs compared (the two differ in summation order only), both timed warm and with the weights cold (COLD=<n> copies in rotation, as inside the UNet).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DWDM_D8KABL=<m> -DWDM_D8ABL=<m>] -I wavedm_amd/csrc -I include -I tools tools/dma8k_bench.hip -o /tmp/dma8k_bench
// run:   [SC=<shortcut channels>] [COLD=24] [GN=4] /tmp/dma8k_bench B Cin Cout
#include <hip/hip_runtime.h>
#include <math.h>
#include <string.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "conv_dma8_kernel.h"
#include "experiments/conv_dma8k_kernel.h"
using namespace wdm;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

static float bf2f(unsigned short v) { unsigned u = (unsigned)v << 16; float f; memcpy(&f, &u, 4); return f; }

template <class K>
static float time_kernel(K kern, int grid, int nthreads, int lds, ConvArgs a, const std::vector<unsigned short*>& ws) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(nthreads), lds, 0, a);
    CK(hipDeviceSynchronize());
    float ms = 0.f;
    if (ws.size() > 1) {
        const int nc = (int)ws.size();
        float tot = 0.f;
        for (int i = 0; i < 3 * nc; ++i) {
            a.w = ws[i % nc];
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(kern, dim3(grid), dim3(nthreads), lds, 0, a);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float t; CK(hipEventElapsedTime(&t, e0, e1));
            if (i >= nc) tot += t;
        }
        ms = tot / (2 * nc);
    } else {
        const int it = 20;
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < it; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(nthreads), lds, 0, a);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= it;
    }
    return ms * 1e3f;
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 64, Cin = argc > 2 ? atoi(argv[2]) : 768, Cout = argc > 3 ? atoi(argv[3]) : 768;
    const int SC = getenv("SC") ? atoi(getenv("SC")) : 0;
    const int H = 8;
    const size_t nx = (size_t)B * H * H * Cin, ny = (size_t)B * H * H * Cout, nw = (size_t)9 * Cout * Cin, nsx = (size_t)B * H * H * (SC ? SC : 64), nsw = (size_t)Cout * (SC ? SC : 64);
    unsigned short *x, *y0, *y1, *w, *sx, *sw; float* bias;
    CK(hipMalloc(&x, nx * 2)); CK(hipMalloc(&y0, ny * 2)); CK(hipMalloc(&y1, ny * 2)); CK(hipMalloc(&w, nw * 2)); CK(hipMalloc(&bias, Cout * 4));
    CK(hipMalloc(&sx, nsx * 2)); CK(hipMalloc(&sw, nsw * 2));
    std::vector<unsigned short> hx(nx), hw(nw), hsx(nsx), hsw(nsw);
    srand(1);
    for (auto& v : hx) v = (unsigned short)(0x3f00 + (rand() & 0xff) + ((rand() & 1) << 15));
    for (auto& v : hw) v = (unsigned short)(0x3c00 + (rand() & 0xff) + ((rand() & 1) << 15));
    for (auto& v : hsx) v = (unsigned short)(0x3f00 + (rand() & 0xff) + ((rand() & 1) << 15));
    for (auto& v : hsw) v = (unsigned short)(0x3c00 + (rand() & 0xff) + ((rand() & 1) << 15));
    CK(hipMemcpy(x, hx.data(), nx * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(w, hw.data(), nw * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(sx, hsx.data(), nsx * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(sw, hsw.data(), nsw * 2, hipMemcpyHostToDevice));
    CK(hipMemset(bias, 0, Cout * 4));
    ConvArgs a{};
    a.x0 = x; a.C0 = Cin; a.xs0 = Cin; a.B = B; a.Hin = a.Win = a.Hout = a.Wout = H; a.Cin = Cin; a.Cout = Cout;
    a.w = w; a.w_rows = Cout; a.bias = bias; a.alpha = 1.f;
    a.w_tap_stride = (long long)Cout * 32; a.w_row_stride = 32; a.w_slab_stride = 9 * Cout * 32;      // slab-major weights (the bytes are random either way)
    a.y_mode = Y_NHWC; a.y_s = Cout;
    a.x0_bytes = (unsigned)(nx * 2); a.w_bytes = (unsigned)(nw * 2);
    if (SC) {
        a.sx0 = sx; a.sC0 = SC; a.sC1 = 0; a.sxs0 = SC; a.sw = sw; a.sw_row_stride = SC; a.sw_rows = Cout;
        a.sx0_bytes = (unsigned)(nsx * 2); a.sw_bytes = (unsigned)(nsw * 2);
    }
    std::vector<unsigned short*> ws(1, w);
    if (getenv("COLD")) {
        const int nc = atoi(getenv("COLD"));
        ws.resize(nc);
        for (int c = 0; c < nc; ++c) { CK(hipMalloc(&ws[c], nw * 2)); CK(hipMemcpy(ws[c], w, nw * 2, hipMemcpyDeviceToDevice)); }
    }
    const int gn = getenv("GN") ? atoi(getenv("GN")) : 4;
    const double fl = 2.0 * B * H * H * Cout * (9.0 * Cin + SC);
    // ---- the shipped kernel
    {
        using C = ConvDma8Cfg<48, 2>;
        auto kern = conv_dma8_kernel<48, 2>;
        ConvArgs b = a; b.y = y0;
        b.mtiles = (B + 1) / 2; b.ntiles = Cout / 48; b.grid_gn = (b.ntiles % gn) ? 1 : gn;
        const int g = b.grid_gn, gm = 8 / g;
        const int grid = g == 1 ? 8 * b.ntiles * ((b.mtiles + 7) / 8) : 8 * (b.ntiles / g) * ((b.mtiles + gm - 1) / gm);
        CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
        const float us = time_kernel(kern, grid, C::NTHREADS, C::LDS_BYTES, b, ws);
        printf("dma8  128x48 4w : B=%d %d->%d%s grid=%d : %7.1f us  %7.1f TFLOP/s\n", B, Cin, Cout, SC ? " +1x1" : "", grid, us, fl / us / 1e6);
    }
    // ---- the K-split kernel
    {
        using C = ConvDma8kCfg;
        auto kern = conv_dma8k_kernel<__bf16>;
        ConvArgs b = a; b.y = y1;
        b.mtiles = (B + 1) / 2; b.ntiles = Cout / 96;
        const int gk = getenv("GNK") ? atoi(getenv("GNK")) : gn;
        b.grid_gn = (b.ntiles % gk) ? 1 : gk;
        const int g = b.grid_gn, gm = 8 / g;
        const int grid = g == 1 ? 8 * b.ntiles * ((b.mtiles + 7) / 8) : 8 * (b.ntiles / g) * ((b.mtiles + gm - 1) / gm);
        CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
        const float us = time_kernel(kern, grid, C::NTHREADS, C::LDS_BYTES, b, ws);
        printf("dma8k 128x96 8w : B=%d %d->%d%s grid=%d : %7.1f us  %7.1f TFLOP/s\n", B, Cin, Cout, SC ? " +1x1" : "", grid, us, fl / us / 1e6);
    }
    std::vector<unsigned short> h0(ny), h1(ny);
    CK(hipMemcpy(h0.data(), y0, ny * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(h1.data(), y1, ny * 2, hipMemcpyDeviceToHost));
    double mx = 0.0, md = 0.0; size_t nd = 0;
    for (size_t i = 0; i < ny; ++i) { const double r = bf2f(h0[i]), v = bf2f(h1[i]); mx = fmax(mx, fabs(r)); md = fmax(md, fabs(r - v)); nd += h0[i] != h1[i]; }
    printf("outputs: max |ref| %.4g, max |diff| %.4g (rel %.3g), %zu of %zu differ\n", mx, md, md / mx, nd, ny);
    return md / mx < 8e-3 ? 0 : 1;
}
