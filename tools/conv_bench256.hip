// A/B harness for the LDS-DMA 3x3 conv kernels (not part of the library): conv_dma_kernel.h (256 x 128 tile, the reference) against the
// 256-column tilings of conv_dma256_kernel.h on the same random inputs -- outputs and GroupNorm partial statistics must be BIT-identical --
// and interleaved timing rounds.  Every layer carries what the model's layers carry: statistics, temb rows, optionally a residual or the fused
// 1x1 shortcut over a (concatenated) block input.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I wavedm_amd/csrc -I tools/experiments -I include tools/conv_bench256.hip -o tools/abl_conv_bench256
// run:   tools/abl_conv_bench256            (the model's layer shapes at batch 64)
//        tools/abl_conv_bench256 B H Cin Cout [pro] [shortcut channels]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "conv_dma_kernel.h"
#include "conv_dma256_kernel.h"
#include "conv_dma4w_kernel.h"
#include "conv_dmap_kernel.h"
#include "conv_dma2_kernel.h"
using namespace wdm;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

static float bf2f(unsigned short v) { unsigned u = (unsigned)v << 16; float f; memcpy(&f, &u, 4); return f; }
static unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (unsigned short)(u >> 16); }
static float frand() { return (float)rand() / RAND_MAX * 2.f - 1.f; }

struct Shape { int B, H, Cin, Cout, pro, res, sc; };      // sc: channels of the fused 1x1 shortcut's input (0 = none)
typedef void (*kern_t)(const ConvArgs);
struct Variant { const char* name; kern_t kern; int lds, th, bn, persist, threads = 512; };

static float time_kernel(const Variant& v, int grid, const ConvArgs& a, int it) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < it; ++i) hipLaunchKernelGGL(v.kern, dim3(grid), dim3(v.threads), v.lds, 0, a);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return ms / it * 1e3f;
}

int main(int argc, char** argv) {
    std::vector<Shape> shapes = {
        {64, 64, 128, 128, 1, 1, 0}, {64, 64, 128, 128, 1, 0, 0}, {64, 64, 128, 128, 1, 0, 256}, {64, 64, 256, 128, 1, 0, 0}, {64, 64, 384, 128, 1, 0, 0}, {64, 64, 96, 128, 0, 0, 0}, {9, 64, 128, 128, 1, 1, 0}, {17, 32, 64, 128, 1, 0, 128},
        {64, 32, 256, 256, 1, 0, 512}, {64, 32, 256, 256, 1, 1, 0}, {64, 32, 768, 256, 1, 0, 0}, {64, 32, 512, 256, 1, 0, 0}, {64, 32, 384, 256, 1, 0, 0},
        {64, 32, 128, 256, 1, 0, 0}, {64, 16, 512, 512, 1, 0, 1024}, {64, 16, 512, 512, 1, 1, 0}, {64, 16, 1280, 512, 1, 0, 0}, {64, 16, 1024, 512, 1, 0, 0},
        {64, 16, 768, 512, 1, 0, 0}, {64, 16, 256, 512, 1, 0, 0}, {3, 16, 512, 512, 1, 1, 0}, {2, 32, 128, 256, 1, 0, 192}, {5, 48, 64, 256, 0, 0, 0},
    };
    if (argc > 4) shapes = {{atoi(argv[1]), atoi(argv[2]), atoi(argv[3]), atoi(argv[4]), argc > 5 ? atoi(argv[5]) : 1, 0, argc > 6 ? atoi(argv[6]) : 0}};
    const int rounds = getenv("ROUNDS") ? atoi(getenv("ROUNDS")) : 3, iters = getenv("IT") ? atoi(getenv("IT")) : 10;
    using C128 = ConvDmaCfg;
    using C256 = ConvDma256Cfg<4, 2, 4, 8, 16>;
    using C256h = ConvDma256Cfg<2, 4, 4, 4, 8>;
    std::vector<Variant> vars = {
        {"t256x128", conv_dma_kernel<false>, C128::LDS_BYTES, 16, 128, 0},
        {"t256x128P", conv_dma_kernel<true>, C128::LDS_BYTES, 16, 128, 0},
        {"t256x256", conv_dma256_kernel<4, 2, 4, 8, 16, true>, C256::LDS_BYTES, 16, 256, 0},
        {"t256x256F", conv_dma256_kernel<4, 2, 4, 8, 16, false>, C256::LDS_BYTES, 16, 256, 0},
        {"t512x128", conv_dma256_kernel<8, 1, 4, 8, 32, true, false>, ConvDma256Cfg<8, 1, 4, 8, 32>::LDS_BYTES, 32, 128, 0},
        {"t512x128F", conv_dma256_kernel<8, 1, 4, 8, 32, false, false>, ConvDma256Cfg<8, 1, 4, 8, 32>::LDS_BYTES, 32, 128, 0},
        {"t512x128S", conv_dma256_kernel<8, 1, 4, 8, 32, true, true>, ConvDma256Cfg<8, 1, 4, 8, 32>::LDS_BYTES, 32, 128, 0},
        {"t512x128FS", conv_dma256_kernel<8, 1, 4, 8, 32, false, true>, ConvDma256Cfg<8, 1, 4, 8, 32>::LDS_BYTES, 32, 128, 0},
        {"w4_256x256", conv_dma4w_kernel<2, 2, 16, true>, ConvDma4wCfg<2, 2, 16>::LDS_BYTES, 16, 256, 0, 256},
        {"w4_256x256F", conv_dma4w_kernel<2, 2, 16, false>, ConvDma4wCfg<2, 2, 16>::LDS_BYTES, 16, 256, 0, 256},
        {"w4_512x128", conv_dma4w_kernel<4, 1, 32, true>, ConvDma4wCfg<4, 1, 32>::LDS_BYTES, 32, 128, 0, 256},
        {"w4_512x128F", conv_dma4w_kernel<4, 1, 32, false>, ConvDma4wCfg<4, 1, 32>::LDS_BYTES, 32, 128, 0, 256},
        {"persist1", conv_dmap_kernel<true>, ConvDmaPCfg::LDS_BYTES, 16, 128, 1},
        {"persistF", conv_dmap_kernel<false>, ConvDmaPCfg::LDS_BYTES, 16, 128, 1},
        {"two80", conv_dma2_kernel<true>, ConvDma2Cfg::LDS_BYTES, 16, 128, 0, 256},
        {"two80F", conv_dma2_kernel<false>, ConvDma2Cfg::LDS_BYTES, 16, 128, 0, 256},
    };
    if (getenv("ONLY")) { std::vector<Variant> keep = {vars[0]}; for (size_t i = 1; i < vars.size(); ++i) if (strstr(getenv("ONLY"), vars[i].name)) keep.push_back(vars[i]); vars = keep; }
    const int NCU = getenv("NCU") ? atoi(getenv("NCU")) : 256;
    const int NV = (int)vars.size();
    for (auto& v : vars) CK(hipFuncSetAttribute((const void*)v.kern, hipFuncAttributeMaxDynamicSharedMemorySize, v.lds));
    int bad_total = 0;
    for (const Shape& sh : shapes) {
        const int B = sh.B, H = sh.H, Cin = sh.Cin, Cout = sh.Cout;
        const int C0 = (Cin >= 256 && (Cin / 2) % 32 == 0 && sh.sc == 0 && Cin % 64 == 0) ? Cin / 2 + 32 * ((Cin / 64) % 2) : Cin;   // some layers read a concat [x0 | x1]
        const int C1 = Cin - C0;
        const size_t npx = (size_t)B * H * H, ny = npx * Cout, nw = (size_t)9 * Cout * Cin;
        unsigned short *x0, *x1 = nullptr, *w, *wsm, *res, *sx = nullptr, *sw = nullptr; float *sc, *shf, *bias, *sbias, *temb;
        std::vector<unsigned short*> y(NV); std::vector<float*> st(NV);
        const int nslab = (H / 16) * (H / 16) * 4;
        CK(hipMalloc(&x0, npx * C0 * 2)); if (C1) CK(hipMalloc(&x1, npx * C1 * 2));
        CK(hipMalloc(&w, nw * 2)); CK(hipMalloc(&wsm, nw * 2)); CK(hipMalloc(&res, ny * 2));
        for (int i = 0; i < NV; ++i) { CK(hipMalloc(&y[i], ny * 2)); CK(hipMalloc(&st[i], (size_t)B * nslab * Cout * 16)); CK(hipMemset(y[i], 0xff, ny * 2)); CK(hipMemset(st[i], 0xff, (size_t)B * nslab * Cout * 16)); }
        CK(hipMalloc(&sc, (size_t)B * Cin * 4)); CK(hipMalloc(&shf, (size_t)B * Cin * 4)); CK(hipMalloc(&bias, Cout * 4)); CK(hipMalloc(&sbias, Cout * 4)); CK(hipMalloc(&temb, (size_t)B * Cout * 4));
        std::vector<unsigned short> hx0(npx * C0), hx1(npx * C1), hw(nw), hwsm(nw), hr(ny);
        srand(1234 + H + Cin);
        for (auto& v : hx0) v = f2bf(frand() * 2.f);
        for (auto& v : hx1) v = f2bf(frand() * 2.f);
        const float ws = 1.f / sqrtf(9.f * Cin);
        for (auto& v : hw) v = f2bf(frand() * ws * 1.7f);
        for (int s = 0; s < Cin / 32; ++s) for (int t = 0; t < 9; ++t) for (int n = 0; n < Cout; ++n) for (int c = 0; c < 32; ++c)     // slab-major copy [slab][tap][row][32]
            hwsm[(((size_t)s * 9 + t) * Cout + n) * 32 + c] = hw[((size_t)t * Cout + n) * Cin + s * 32 + c];
        for (auto& v : hr) v = f2bf(frand());
        std::vector<float> hsc((size_t)B * Cin), hsh((size_t)B * Cin), hb(Cout), hsb(Cout), ht((size_t)B * Cout);
        for (auto& v : hsc) v = -1.4426950408889634f * (0.5f + 0.5f * fabsf(frand()));
        for (auto& v : hsh) v = -1.4426950408889634f * 0.3f * frand();
        for (auto& v : hb) v = 0.1f * frand();
        for (auto& v : hsb) v = 0.1f * frand();
        for (auto& v : ht) v = 0.2f * frand();
        CK(hipMemcpy(x0, hx0.data(), hx0.size() * 2, hipMemcpyHostToDevice)); if (C1) CK(hipMemcpy(x1, hx1.data(), hx1.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(w, hw.data(), nw * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(wsm, hwsm.data(), nw * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(res, hr.data(), ny * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(sc, hsc.data(), hsc.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(shf, hsh.data(), hsh.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(bias, hb.data(), Cout * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(sbias, hsb.data(), Cout * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(temb, ht.data(), ht.size() * 4, hipMemcpyHostToDevice));
        ConvArgs a{};
        a.x0 = x0; a.x1 = x1; a.C0 = C0; a.C1 = C1; a.xs0 = C0; a.xs1 = C1; a.B = B; a.Hin = a.Win = a.Hout = a.Wout = H; a.Cin = Cin; a.Cout = Cout;
        a.w_rows = Cout; a.bias = bias; a.alpha = 1.f; a.w_bytes = (unsigned)(nw * 2);
        if (getenv("PLAINW")) { a.w = w; a.w_tap_stride = (long long)Cout * Cin; a.w_row_stride = Cin; }
        else { a.w = wsm; a.w_tap_stride = (long long)Cout * 32; a.w_row_stride = 32; a.w_slab_stride = 9 * Cout * 32; }
        a.pro = sh.pro; a.scale = sc; a.shift = shf; a.y_mode = Y_NHWC; a.y_s = Cout;
        a.x0_bytes = (unsigned)(npx * C0 * 2); a.x1_bytes = (unsigned)(npx * C1 * 2);
        a.temb = temb; a.temb_ld = Cout; a.temb_per_image = 1;
        if (sh.res) { a.res = res; a.res_s = Cout; }
        if (sh.sc) {
            const size_t nsx = npx * sh.sc, nsw = (size_t)Cout * sh.sc;
            CK(hipMalloc(&sx, nsx * 2)); CK(hipMalloc(&sw, nsw * 2));
            std::vector<unsigned short> hsx(nsx), hsw(nsw);
            for (auto& v : hsx) v = f2bf(frand());
            for (auto& v : hsw) v = f2bf(frand() / sqrtf((float)sh.sc));
            CK(hipMemcpy(sx, hsx.data(), nsx * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(sw, hsw.data(), nsw * 2, hipMemcpyHostToDevice));
            // the shortcut input as a concat of two halves of one buffer (pixel stride sh.sc): exercises the sx1 path
            a.sx0 = sx; a.sx1 = sx + sh.sc / 2; a.sC0 = sh.sc / 2; a.sC1 = sh.sc - sh.sc / 2; a.sxs0 = sh.sc; a.sxs1 = sh.sc;
            a.sw = sw; a.sw_row_stride = sh.sc; a.sw_rows = Cout; a.sbias = sbias;
            a.sx0_bytes = (unsigned)(nsx * 2); a.sx1_bytes = (unsigned)(nsx * 2 - sh.sc); a.sw_bytes = (unsigned)(nsw * 2);
        }
        a.stats_nslab = nslab;
        std::vector<ConvArgs> aa(NV);
        std::vector<int> grid(NV);
        for (int i = 0; i < NV; ++i) {
            aa[i] = a; aa[i].y = y[i]; aa[i].stats = getenv("NOSTATS") ? nullptr : st[i];      // NOSTATS=1: what the statistics pass of the epilogue costs (timing only)
            aa[i].mtiles = B * (H / vars[i].th) * (H / 16); aa[i].ntiles = (Cout + vars[i].bn - 1) / vars[i].bn; aa[i].grid_gn = 1;
            grid[i] = 8 * aa[i].ntiles * ((aa[i].mtiles + 7) / 8);
            if (vars[i].persist && grid[i] > NCU) grid[i] = NCU;
        }
        std::vector<int> skip(NV, 0);
        for (int v = 0; v < NV; ++v) skip[v] = (vars[v].bn == 256 && Cout % 256 != 0) || (vars[v].th == 32 && H % 32 != 0) || (!strcmp(vars[v].name, "t512x128") && (sh.res || sh.sc)) || (!strcmp(vars[v].name, "t512x128F") && (!sh.res || sh.sc)) || (!strcmp(vars[v].name, "t512x128S") && (sh.res || !sh.sc)) || (!strcmp(vars[v].name, "t512x128FS") && !sh.sc) || (!strncmp(vars[v].name, "two80", 5) && Cin > 768) || (!strcmp(vars[v].name, "two80") && sh.res) || (!strcmp(vars[v].name, "two80F") && !sh.res)
#ifdef WDM_NO_PACK
            || !strcmp(vars[v].name, "persist1") || !strcmp(vars[v].name, "t256x256");
#else
            || (!strcmp(vars[v].name, "persist1") && sh.res) || (!strcmp(vars[v].name, "persistF") && !sh.res)
            || (!strcmp(vars[v].name, "w4_256x256") && sh.res) || (!strcmp(vars[v].name, "w4_256x256F") && !sh.res) || (!strcmp(vars[v].name, "w4_512x128") && sh.res) || (!strcmp(vars[v].name, "w4_512x128F") && !sh.res)
            || (!strcmp(vars[v].name, "t256x128P") && sh.res) || (!strcmp(vars[v].name, "t256x256") && sh.res) || (!strcmp(vars[v].name, "t256x256F") && !sh.res);
#endif
        for (int v = 0; v < NV; ++v) if (!skip[v]) hipLaunchKernelGGL(vars[v].kern, dim3(grid[v]), dim3(vars[v].threads), vars[v].lds, 0, aa[v]);
        CK(hipDeviceSynchronize());
        std::vector<unsigned short> h0(ny), h1(ny);
        std::vector<unsigned> s0((size_t)B * nslab * Cout * 4), s1(s0.size());
        CK(hipMemcpy(h0.data(), y[0], ny * 2, hipMemcpyDeviceToHost));
        CK(hipMemcpy(s0.data(), st[0], s0.size() * 4, hipMemcpyDeviceToHost));
        double amax = 0, csum = 0;
        for (size_t i = 0; i < ny; ++i) { amax = fmax(amax, fabs(bf2f(h0[i]))); csum += bf2f(h0[i]) * (double)((i % 251) + 1); }
        std::vector<size_t> nbad(NV, 0), nbad_s(NV, 0);
        for (int v = 1; v < NV; ++v) {
            if (skip[v]) continue;
            CK(hipMemcpy(h1.data(), y[v], ny * 2, hipMemcpyDeviceToHost));
            CK(hipMemcpy(s1.data(), st[v], s1.size() * 4, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < ny; ++i) if (h0[i] != h1[i]) ++nbad[v];
            for (size_t i = 0; i < s0.size(); ++i) if (s0[i] != s1[i]) { if (nbad_s[v] < 6 && getenv("DUMP")) { float f0, f1; memcpy(&f0, &s0[i], 4); memcpy(&f1, &s1[i], 4); const size_t e = i / 4; printf("   %s stats[%zu] img %zu slab %zu ch %zu word %zu: %.9g vs %.9g\n", vars[v].name, i, e / ((size_t)nslab * Cout), (e / Cout) % nslab, e % Cout, i % 4, f0, f1); } ++nbad_s[v]; }
            bad_total += (nbad[v] != 0) + (nbad_s[v] != 0);
        }
        std::vector<float> t(NV, 1e9f);
        for (int round = 0; round < rounds; ++round)
            for (int v = 0; v < NV; ++v) if (!skip[v] && !(v == 0 && NV > 1 && getenv("NOREF"))) t[v] = fminf(t[v], time_kernel(vars[v], grid[v], aa[v], iters));
        const double fl = 2.0 * B * H * (double)H * Cout * (9.0 * Cin + sh.sc);
        printf("B=%2d %2dx%-2d %4d->%-4d pro=%d res=%d sc=%-4d amax %.2f csum %.6g |", B, H, H, Cin, Cout, sh.pro, sh.res, sh.sc, amax, csum);
        for (int v = 0; v < NV; ++v) if (!skip[v]) printf(" %s wg %4d %6.1f us %5.0f TF (bad %zu / %zu) |", vars[v].name, grid[v], t[v], fl / t[v] / 1e6, nbad[v], nbad_s[v]);
        printf("\n");
        CK(hipFree(x0)); if (x1) CK(hipFree(x1)); CK(hipFree(w)); CK(hipFree(wsm)); CK(hipFree(res)); CK(hipFree(sc)); CK(hipFree(shf)); CK(hipFree(bias)); CK(hipFree(sbias)); CK(hipFree(temb));
        if (sx) { CK(hipFree(sx)); CK(hipFree(sw)); }
        for (int i = 0; i < NV; ++i) { CK(hipFree(y[i])); CK(hipFree(st[i])); }
    }
    printf(bad_total ? "MISMATCHES: %d\n" : "all variants bit-identical to the 256 x 128 tile (%d)\n", bad_total);
    return bad_total ? 1 : 0;
}
