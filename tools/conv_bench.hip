// A/B harness for the main 3x3 conv kernels (not part of the library): runs conv_dma_kernel.h (reference) and conv_pp_kernel.h on the same
// random inputs for a list of layer shapes, compares outputs and GroupNorm partial statistics, and times both in interleaved rounds.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I wavedm_amd/csrc -I tools/experiments -I include -I tools tools/conv_bench.hip -o tools/abl_conv_bench
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <vector>
#include "conv_dma_kernel.h"
#include "experiments/conv_pp_kernel.h"
using namespace wdm;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

static float bf2f(unsigned short v) { unsigned u = (unsigned)v << 16; float f; memcpy(&f, &u, 4); return f; }
static unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (unsigned short)(u >> 16); }
static float frand() { return (float)rand() / RAND_MAX * 2.f - 1.f; }

struct Shape { int B, H, Cin, Cout, pro, res; };

template <class K> static float time_kernel(K kern, int grid, int lds, const ConvArgs& a, int it) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < it; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, 0, a);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return ms / it * 1e3f;
}

int main(int argc, char** argv) {
    std::vector<Shape> shapes = {
        {64, 16, 512, 512, 1, 0}, {64, 64, 128, 128, 1, 0}, {64, 32, 256, 256, 1, 1}, {64, 64, 256, 128, 1, 0}, {64, 16, 1280, 512, 1, 0},
        {64, 32, 768, 256, 1, 0}, {64, 64, 96, 128, 0, 0}, {3, 16, 512, 512, 1, 1}, {2, 32, 128, 256, 1, 0},
    };
    if (argc > 4) shapes = {{atoi(argv[1]), atoi(argv[2]), atoi(argv[3]), atoi(argv[4]), argc > 5 ? atoi(argv[5]) : 1, 0}};
    typedef void (*kern_t)(const ConvArgs);
    constexpr int NV = 5;
    kern_t kerns_p[NV] = {conv_dma_kernel<false>, conv_pp_kernel<0, true>, conv_pp_kernel<8, true>, conv_pp_kernel<10, true>, conv_pp_kernel<12, true>};
    kern_t kerns_n[NV] = {conv_dma_kernel<false>, conv_pp_kernel<0, false>, conv_pp_kernel<8, false>, conv_pp_kernel<8, false>, conv_pp_kernel<8, false>};
    const char* names[NV] = {"old", "pp0", "ls0", "ls2", "ls4"};
    int ldsb[NV] = {ConvDmaCfg::LDS_BYTES, ConvPPCfg::LDS_BYTES, ConvPPCfg::LDS_BYTES, ConvPPCfg::LDS_BYTES, ConvPPCfg::LDS_BYTES};
    for (int v = 0; v < NV; ++v) { CK(hipFuncSetAttribute((const void*)kerns_p[v], hipFuncAttributeMaxDynamicSharedMemorySize, ldsb[v])); CK(hipFuncSetAttribute((const void*)kerns_n[v], hipFuncAttributeMaxDynamicSharedMemorySize, ldsb[v])); }
    for (const Shape& sh : shapes) {
        const int B = sh.B, H = sh.H, Cin = sh.Cin, Cout = sh.Cout;
        kern_t* kerns = sh.pro ? kerns_p : kerns_n;
        const size_t nx = (size_t)B * H * H * Cin, ny = (size_t)B * H * H * Cout, nw = (size_t)9 * Cout * Cin;
        unsigned short *x, *w, *res, *y[NV]; float *sc, *shf, *bias, *st[NV];
        const int nslab = (H / 16) * (H / 16) * 4;
        CK(hipMalloc(&x, nx * 2)); CK(hipMalloc(&w, nw * 2)); CK(hipMalloc(&res, ny * 2));
        for (int i = 0; i < NV; ++i) { CK(hipMalloc(&y[i], ny * 2)); CK(hipMalloc(&st[i], (size_t)B * nslab * Cout * 16)); CK(hipMemset(y[i], 0, ny * 2)); }
        CK(hipMalloc(&sc, (size_t)B * Cin * 4)); CK(hipMalloc(&shf, (size_t)B * Cin * 4)); CK(hipMalloc(&bias, Cout * 4));
        std::vector<unsigned short> hx(nx), hw(nw), hr(ny);
        srand(1234 + H + Cin);
        for (auto& v : hx) v = f2bf(frand() * 2.f);
        const float ws = 1.f / sqrtf(9.f * Cin);
        for (auto& v : hw) v = f2bf(frand() * ws * 1.7f);
        for (auto& v : hr) v = f2bf(frand());
        std::vector<float> hsc((size_t)B * Cin), hsh((size_t)B * Cin), hb(Cout);
        for (auto& v : hsc) v = -1.4426950408889634f * (0.5f + 0.5f * fabsf(frand()));
        for (auto& v : hsh) v = -1.4426950408889634f * 0.3f * frand();
        for (auto& v : hb) v = 0.1f * frand();
        CK(hipMemcpy(x, hx.data(), nx * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(w, hw.data(), nw * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(res, hr.data(), ny * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(sc, hsc.data(), hsc.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(shf, hsh.data(), hsh.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(bias, hb.data(), Cout * 4, hipMemcpyHostToDevice));
        ConvArgs a{};
        a.x0 = x; a.C0 = Cin; a.xs0 = Cin; a.B = B; a.Hin = a.Win = a.Hout = a.Wout = H; a.Cin = Cin; a.Cout = Cout;
        a.w = w; a.w_tap_stride = (long long)Cout * Cin; a.w_row_stride = Cin; a.w_rows = Cout; a.bias = bias; a.alpha = 1.f;
        a.pro = sh.pro; a.scale = sc; a.shift = shf; a.y_mode = Y_NHWC; a.y_s = Cout;
        a.x0_bytes = (unsigned)(nx * 2); a.w_bytes = (unsigned)(nw * 2);
        if (sh.res) { a.res = res; a.res_s = Cout; }
        a.stats_nslab = nslab;
        a.mtiles = B * (H / 16) * (H / 16); a.ntiles = (Cout + 127) / 128; a.grid_gn = 1;
        const int grid = 8 * a.ntiles * ((a.mtiles + 7) / 8);
        ConvArgs aa[NV];
        for (int i = 0; i < NV; ++i) { aa[i] = a; aa[i].y = y[i]; aa[i].stats = st[i]; }
        for (int v = 0; v < NV; ++v) hipLaunchKernelGGL(kerns[v], dim3(grid), dim3(512), ldsb[v], 0, aa[v]);
        CK(hipDeviceSynchronize());
        std::vector<unsigned short> h0(ny), h1(ny);
        std::vector<float> s0((size_t)B * nslab * Cout * 4), s1(s0.size());
        CK(hipMemcpy(h0.data(), y[0], ny * 2, hipMemcpyDeviceToHost));
        CK(hipMemcpy(s0.data(), st[0], s0.size() * 4, hipMemcpyDeviceToHost));
        double worst[NV] = {0}, worst_s[NV] = {0}, amax = 0, csum = 0;
        for (size_t i = 0; i < ny; ++i) { amax = fmax(amax, fabs(bf2f(h0[i]))); csum += bf2f(h0[i]) * (double)((i % 251) + 1); }
        for (int v = 1; v < NV; ++v) {
            CK(hipMemcpy(h1.data(), y[v], ny * 2, hipMemcpyDeviceToHost));
            CK(hipMemcpy(s1.data(), st[v], s1.size() * 4, hipMemcpyDeviceToHost));
            size_t nbad = 0;
            for (size_t i = 0; i < ny; ++i) { const double d = fabs(bf2f(h0[i]) - bf2f(h1[i])); if (d > worst[v]) worst[v] = d; if (!(d <= 0.02 * amax)) ++nbad; }
            for (size_t i = 0; i < s0.size(); i += 4) {
                const double m0 = s0[i] + s0[i + 1] / s0[i + 3], m1 = s1[i] + s1[i + 1] / s1[i + 3];
                const double v0 = s0[i + 2] / s0[i + 3] - (s0[i + 1] / s0[i + 3]) * (s0[i + 1] / s0[i + 3]), v1 = s1[i + 2] / s1[i + 3] - (s1[i + 1] / s1[i + 3]) * (s1[i + 1] / s1[i + 3]);
                worst_s[v] = fmax(worst_s[v], fmax(fabs(m0 - m1), fabs(v0 - v1)));
            }
            if (nbad) printf("   !! %s: %zu outputs differ by more than 2%% of max\n", names[v], nbad);
        }
        float t[NV];
        for (int v = 0; v < NV; ++v) t[v] = 1e9f;
        for (int round = 0; round < 3; ++round)
            for (int v = 0; v < NV; ++v) t[v] = fminf(t[v], time_kernel(kerns[v], grid, ldsb[v], aa[v], 10));
        const double fl = 2.0 * B * H * H * Cout * 9.0 * Cin;
        printf("B=%2d %2dx%-2d %4d->%-4d pro=%d res=%d amax %.2f csum %.6g |", B, H, H, Cin, Cout, sh.pro, sh.res, amax, csum);
        for (int v = 0; v < NV; ++v) printf(" %s %6.1f us %5.0f TF (d %.4f s %.1e) |", names[v], t[v], fl / t[v] / 1e6, worst[v], worst_s[v]);
        printf("\n");
        CK(hipFree(x)); CK(hipFree(w)); CK(hipFree(res)); CK(hipFree(sc)); CK(hipFree(shf)); CK(hipFree(bias));
        for (int i = 0; i < NV; ++i) { CK(hipFree(y[i])); CK(hipFree(st[i])); }
    }
    return 0;
}
