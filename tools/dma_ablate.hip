// Ablation micro-benchmark of the LDS-DMA 3x3 conv kernel (not part of the library): one layer shape, phases switched off by
// -DWDM_DABL=<mask> (conv_dma_kernel.h: 1 transform, 2 MFMAs, 4 halo DMA, 8 weight DMA, 16 fragment reads with 2).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DWDM_DABL=<m> -I wavedm_amd/csrc -I include tools/dma_ablate.hip -o tools/abl_dma_<m>
// run:   tools/abl_dma_<m> [B H Cin Cout pro]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "conv_dma_kernel.h"
#ifdef TILE32
#include "conv_dma_variants.h"
#endif
using namespace wdm;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 64, H = argc > 2 ? atoi(argv[2]) : 16, Cin = argc > 3 ? atoi(argv[3]) : 512, Cout = argc > 4 ? atoi(argv[4]) : 512;
    const int pro = argc > 5 ? atoi(argv[5]) : 1;
    const size_t nx = (size_t)B * H * H * Cin, ny = (size_t)B * H * H * Cout, nw = (size_t)9 * Cout * Cin;
    unsigned short *x, *y, *w; float *sc, *sh, *bias;
    CK(hipMalloc(&x, nx * 2)); CK(hipMalloc(&y, ny * 2)); CK(hipMalloc(&w, nw * 2));
    CK(hipMalloc(&sc, (size_t)B * Cin * 4)); CK(hipMalloc(&sh, (size_t)B * Cin * 4)); CK(hipMalloc(&bias, Cout * 4));
    std::vector<unsigned short> hx(nx), hw(nw);
    srand(1);
    for (auto& v : hx) v = (unsigned short)(0x3f00 + (rand() & 0xff) + ((rand() & 1) << 15));
    for (auto& v : hw) v = (unsigned short)(0x3c00 + (rand() & 0xff) + ((rand() & 1) << 15));
    CK(hipMemcpy(x, hx.data(), nx * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(w, hw.data(), nw * 2, hipMemcpyHostToDevice));
    std::vector<float> ones((size_t)B * Cin, -1.4426950408889634f);
    CK(hipMemcpy(sc, ones.data(), ones.size() * 4, hipMemcpyHostToDevice)); CK(hipMemset(sh, 0, ones.size() * 4)); CK(hipMemset(bias, 0, Cout * 4));
    ConvArgs a{};
    a.x0 = x; a.C0 = Cin; a.xs0 = Cin; a.B = B; a.Hin = a.Win = a.Hout = a.Wout = H; a.Cin = Cin; a.Cout = Cout;
    a.w = w; a.w_tap_stride = (long long)Cout * Cin; a.w_row_stride = Cin; a.w_rows = Cout; a.bias = bias; a.alpha = 1.f;
    a.pro = pro; a.scale = sc; a.shift = sh; a.y = y; a.y_mode = Y_NHWC; a.y_s = Cout;
    a.x0_bytes = (unsigned)(nx * 2); a.w_bytes = (unsigned)(nw * 2);
    if (getenv("SM")) { a.w_tap_stride = (long long)Cout * 32; a.w_row_stride = 32; a.w_slab_stride = 9 * Cout * 32; }      // slab-major weights
    if (getenv("FULL")) {       // what the model's layers also carry: GroupNorm partial statistics of the output, temb rows, a residual
        const int nslab = (H / 16) * (H / 16) * 4;
        float* st; float* temb; unsigned short* res;
        CK(hipMalloc(&st, (size_t)B * nslab * Cout * 16)); CK(hipMalloc(&temb, (size_t)B * Cout * 4)); CK(hipMalloc(&res, ny * 2));
        CK(hipMemset(temb, 0, (size_t)B * Cout * 4)); CK(hipMemset(res, 0, ny * 2));
        a.stats = st; a.stats_nslab = nslab; a.temb = temb; a.temb_ld = Cout; a.temb_per_image = 1;
        if (atoi(getenv("FULL")) > 1) { a.res = res; a.res_s = Cout; }
    }
#ifdef WDM_EPI_TS
    unsigned long long* ts; CK(hipMalloc(&ts, 128 * 8)); CK(hipMemset(ts, 0, 128 * 8)); a.ts = ts;
#endif
#ifdef TILE32              // the 512 x 128 tile of round 3 (tools/experiments/conv_dma_variants.h; add -I tools/experiments and include it)
    using C = ConvDmaVarCfgT<4, 2, 8, 4, 32, 3>;
    auto kern = conv_dma_var_kernel<4, 2, 8, 4, 32, 3>;
    a.mtiles = B * (H / 32) * (H / 16);
#else
    using C = ConvDmaCfg;
    auto kern = conv_dma_kernel<false>;
    a.mtiles = B * (H / 16) * (H / 16);
#endif
    a.ntiles = (Cout + C::BN - 1) / C::BN; a.grid_gn = 1;
    const int grid = 8 * a.ntiles * ((a.mtiles + 7) / 8);
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), C::LDS_BYTES, 0, a);
    CK(hipDeviceSynchronize());
    // NW=<n>: cycle through n copies of the weight tensor (and NX=<n> of the input) so that every launch finds them cold in L2 (n * size > 32 MB) or in
    // the Infinity Cache as well (> 256 MB): what a layer sees inside the model, where 312 MB of weights stream by between two uses
    const int NW = getenv("NW") ? atoi(getenv("NW")) : 1, NX = getenv("NX") ? atoi(getenv("NX")) : 1;
    std::vector<unsigned short*> ws(NW, w), xs(NX, x);
    for (int i = 1; i < NW; ++i) { CK(hipMalloc(&ws[i], nw * 2)); CK(hipMemcpy(ws[i], w, nw * 2, hipMemcpyDeviceToDevice)); }
    for (int i = 1; i < NX; ++i) { CK(hipMalloc(&xs[i], nx * 2)); CK(hipMemcpy(xs[i], x, nx * 2, hipMemcpyDeviceToDevice)); }
    const int it = getenv("IT") ? atoi(getenv("IT")) : 20;
    for (int i = 0; i < it; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), C::LDS_BYTES, 0, a);      // settle the clocks
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < it; ++i) { a.w = ws[i % NW]; a.x0 = xs[i % NX]; hipLaunchKernelGGL(kern, dim3(grid), dim3(512), C::LDS_BYTES, 0, a); }
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double fl = 2.0 * B * H * H * Cout * 9.0 * Cin;
    printf("NW=%d NX=%d DABL=%2d EABL=%d B=%d H=%d %d->%d pro=%d grid=%d : %7.1f us  %7.1f TFLOP/s (nominal)\n", NW, NX, WDM_DABL, WDM_EABL, B, H, Cin, Cout, pro, grid, ms / it * 1e3, fl / (ms / it) / 1e9);
#ifdef WDM_EPI_TS
    {
        unsigned long long h[128]; CK(hipMemcpy(h, a.ts, sizeof(h), hipMemcpyDeviceToHost));
        printf("stamps of workgroup %d (s_memtime ticks since kernel entry of wave 0): entry | main loop end | after barrier | tile written | rows stored | epilogue done | stores acknowledged | first stage ready | slab 1, 2, 3 start | setup done | first DMAs issued | table + DMAs landed\n", WDM_EPI_TS);
        for (int w = 0; w < 8; ++w) { printf("  wave %d:", w); for (int k = 0; k < 14; ++k) printf(" %8lld", (long long)(h[w * 16 + k] - h[0])); printf("\n"); }
    }
#endif
    return 0;
}
