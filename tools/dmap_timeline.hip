// s_memtime timeline of one persistent workgroup of conv_dmap_kernel.h (not part of the library).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DWDM_EPI_TS=8 -I wavedm_amd/csrc -I tools/experiments -I include tools/dmap_timeline.hip -o tools/abl_dmap_timeline
// run:   tools/abl_dmap_timeline [B H Cin Cout pro]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "conv_dmap_kernel.h"
using namespace wdm;
#ifndef PACKED_V
#define PACKED_V true
#endif
#define KERN conv_dmap_kernel<PACKED_V>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 64, H = argc > 2 ? atoi(argv[2]) : 64, Cin = argc > 3 ? atoi(argv[3]) : 128, Cout = argc > 4 ? atoi(argv[4]) : 128;
    const int pro = argc > 5 ? atoi(argv[5]) : 1;
    const size_t nx = (size_t)B * H * H * Cin, ny = (size_t)B * H * H * Cout, nw = (size_t)9 * Cout * Cin;
    unsigned short *x, *y, *w; float *sc, *sh, *bias, *st, *temb;
    CK(hipMalloc(&x, nx * 2)); CK(hipMalloc(&y, ny * 2)); CK(hipMalloc(&w, nw * 2));
    CK(hipMalloc(&sc, (size_t)B * Cin * 4)); CK(hipMalloc(&sh, (size_t)B * Cin * 4)); CK(hipMalloc(&bias, Cout * 4));
    std::vector<unsigned short> hx(nx), hw(nw);
    srand(1);
    for (auto& v : hx) v = (unsigned short)(0x3f00 + (rand() & 0xff) + ((rand() & 1) << 15));
    for (auto& v : hw) v = (unsigned short)(0x3c00 + (rand() & 0xff) + ((rand() & 1) << 15));
    CK(hipMemcpy(x, hx.data(), nx * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(w, hw.data(), nw * 2, hipMemcpyHostToDevice));
    std::vector<float> ones((size_t)B * Cin, -1.4426950408889634f);
    CK(hipMemcpy(sc, ones.data(), ones.size() * 4, hipMemcpyHostToDevice)); CK(hipMemset(sh, 0, ones.size() * 4)); CK(hipMemset(bias, 0, Cout * 4));
    const int nslab = (H / 16) * (H / 16) * 4;
    CK(hipMalloc(&st, (size_t)B * nslab * Cout * 16)); CK(hipMalloc(&temb, (size_t)B * Cout * 4)); CK(hipMemset(temb, 0, (size_t)B * Cout * 4));
    ConvArgs a{};
    a.x0 = x; a.C0 = Cin; a.xs0 = Cin; a.B = B; a.Hin = a.Win = a.Hout = a.Wout = H; a.Cin = Cin; a.Cout = Cout;
    a.w = w; a.w_tap_stride = (long long)Cout * 32; a.w_row_stride = 32; a.w_slab_stride = 9 * Cout * 32; a.w_rows = Cout; a.bias = bias; a.alpha = 1.f;
    a.pro = pro; a.scale = sc; a.shift = sh; a.y = y; a.y_mode = Y_NHWC; a.y_s = Cout;
    a.x0_bytes = (unsigned)(nx * 2); a.w_bytes = (unsigned)(nw * 2);
    a.stats = st; a.stats_nslab = nslab; a.temb = temb; a.temb_ld = Cout; a.temb_per_image = 1;
    unsigned long long* ts; CK(hipMalloc(&ts, (512 + 4 * 4096) * 8)); CK(hipMemset(ts, 0, (512 + 4 * 4096) * 8)); a.ts = ts;
    a.mtiles = B * (H / 16) * (H / 16); a.ntiles = (Cout + 127) / 128; a.grid_gn = 1;
    int grid = 8 * a.ntiles * ((a.mtiles + 7) / 8);
    if (grid > 256) grid = 256;
    CK(hipFuncSetAttribute((const void*)KERN, hipFuncAttributeMaxDynamicSharedMemorySize, ConvDmaPCfg::LDS_BYTES));
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(KERN, dim3(grid), dim3(512), ConvDmaPCfg::LDS_BYTES, 0, a);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(KERN, dim3(grid), dim3(512), ConvDmaPCfg::LDS_BYTES, 0, a);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("B=%d H=%d %d->%d pro=%d grid=%d: %.1f us per launch\n", B, H, Cin, Cout, pro, grid, ms / 20 * 1e3);
    unsigned long long h[8 * 64]; CK(hipMemcpy(h, ts, sizeof(h), hipMemcpyDeviceToHost));
    printf("stamps of workgroup %d, ticks since its first stamp; per tile: top | table+halo in | first weights in (K loop starts) | K loop done | shortcut done | next tile set up | epilogue done | barrier\n", WDM_EPI_TS);
    for (int wv = 0; wv < 8; wv += 3)
        for (int t = 0; t < 8; ++t) {
            if (!h[wv * 64 + t * 8]) continue;
            printf("  wave %d tile %d:", wv, t);
            for (int k = 0; k < 8; ++k) printf(" %8lld", h[wv * 64 + t * 8 + k] ? (long long)(h[wv * 64 + t * 8 + k] - h[0]) : -1LL);
            printf("\n");
        }
    auto report = [&](int grid) {
        std::vector<unsigned long long> g(4 * 4096);
        CK(hipMemcpy(g.data(), ts + 512, g.size() * 8, hipMemcpyDeviceToHost));
        unsigned long long r0 = ~0ull, r1 = 0;
        for (int b = 0; b < grid; ++b) { if (g[4 * b] < r0) r0 = g[4 * b]; if (g[4 * b + 1] > r1) r1 = g[4 * b + 1]; }
        printf("s_memrealtime (100 MHz) span over all workgroups: %llu ticks = %.1f us; per workgroup (start, end since the first start in us; s_memtime ticks of its life; implied clock GHz):\n", r1 - r0, (r1 - r0) / 100.0);
        double tk = 0, rt = 0;
        for (int b = 0; b < grid; ++b) { tk += (double)(g[4 * b + 3] - g[4 * b + 2]); rt += (double)(g[4 * b + 1] - g[4 * b]); }
        printf("  all %d workgroups: mean life %.0f s_memtime ticks = %.2f us, implied clock %.3f GHz\n", grid, tk / grid, rt / grid / 100.0, tk / (rt * 10.0));
        for (int b = 0; b < grid; b += grid / 12 + 1) printf("  wg %3d: %7.2f %7.2f  %8llu  %.3f\n", b, (g[4 * b] - r0) / 100.0, (g[4 * b + 1] - r0) / 100.0, g[4 * b + 3] - g[4 * b + 2], (g[4 * b + 3] - g[4 * b + 2]) / ((g[4 * b + 1] - g[4 * b]) * 10.0));
    };
    report(grid);
    {   // the non-persistent kernel on the same layer, same stamps
        const int grid1 = 8 * a.ntiles * ((a.mtiles + 7) / 8);
        auto k1 = conv_dma_kernel<false>;
        CK(hipFuncSetAttribute((const void*)k1, hipFuncAttributeMaxDynamicSharedMemorySize, ConvDmaCfg::LDS_BYTES));
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k1, dim3(grid1), dim3(512), ConvDmaCfg::LDS_BYTES, 0, a);
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k1, dim3(grid1), dim3(512), ConvDmaCfg::LDS_BYTES, 0, a);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("non-persistent conv_dma_kernel, grid %d: %.1f us per launch\n", grid1, ms / 20 * 1e3);
        report(grid1 > 4096 ? 4096 : grid1);
        CK(hipMemcpy(h, ts, sizeof(h), hipMemcpyDeviceToHost));
        printf("stamps of workgroup %d of conv_dma_kernel: entry | setup done | first DMAs issued | table + halo landed | first stage ready (K loop starts) | slab 1 | slab 2 | slab 3 | main loop end | after barrier | tile written | rows stored | epilogue done | stores acknowledged\n", WDM_EPI_TS);
        const int order[14] = {0, 11, 12, 13, 7, 8, 9, 10, 1, 2, 3, 4, 5, 6};
        for (int wv = 0; wv < 8; wv += 3) { printf("  wave %d:", wv); for (int k = 0; k < 14; ++k) printf(" %7lld", (long long)(h[wv * 16 + order[k]] - h[0])); printf("\n"); }
    }
    return 0;
}
