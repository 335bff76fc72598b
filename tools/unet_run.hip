// Whole-UNet driver without Python (not part of the library): links libwavedm_hip.so, builds the raindrop_wavelet UNet through the C ABI, fills the packed
// weight buffer with small pseudo-random values and runs `iters` forward calls at batch B -- so that `rocprofv3 --pmc ...` (which crashes on the
// python process in this image) can count HBM traffic and SQ activity of EVERY kernel of the path under the real launch sequence: cold weights, real
// operand mix (residuals, shortcuts, statistics).  With PROF=1 it prints the library's own per-kernel table (launches, us, algorithmic flops / bytes)
// as JSON lines: the algorithmic side of the traffic ratios.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include tools/unet_run.hip -L wavedm_amd/csrc -lwavedm_hip -Wl,-rpath,'$ORIGIN/../wavedm_amd/csrc' -o tools/abl_unet_run
// run:   tools/abl_unet_run [B=64] [iters=3] [R=64]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "wavedm.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
#define WK(x) do { int rc = (x); if (rc != 0) { printf("%s -> %d: %s\n", #x, rc, wdm_last_error()); exit(1); } } while (0)

// words whose bf16 halves are ~ +-2^-7 ... 2^-5 and which are also small finite fp32 values (the packed buffer mixes bf16 matrices and fp32 vectors)
__global__ void fill_kernel(uint32_t* p, size_t n, uint32_t seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t h = (uint32_t)i * 2654435761u + seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        const uint32_t lo = 0x3c00u + (h & 0xffu) + ((h >> 8 & 1u) << 15), hi = 0x3c00u + (h >> 16 & 0xffu) + ((h >> 9 & 1u) << 15);
        p[i] = lo | (hi << 16);
    }
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 64, iters = argc > 2 ? atoi(argv[2]) : 3, R = argc > 3 ? atoi(argv[3]) : 64;
    wdm_handle* h; WK(wdm_create(0, &h));
    wdm_unet_config cfg; memset(&cfg, 0, sizeof(cfg));
    cfg.ch = 128; cfg.n_levels = 4; cfg.ch_mult[0] = 1; cfg.ch_mult[1] = 2; cfg.ch_mult[2] = 4; cfg.ch_mult[3] = 6;       // configs/raindrop_wavelet.yml
    cfg.num_res_blocks = 2; cfg.n_attn_res = 1; cfg.attn_resolutions[0] = 16; cfg.in_channels = 96; cfg.out_ch = 3; cfg.resolution = R;
    cfg.resamp_with_conv = 1; cfg.dtype = getenv("DTYPE") ? atoi(getenv("DTYPE")) : WDM_BF16;
    wdm_unet* u; WK(wdm_unet_create(h, &cfg, &u));
    const size_t pb = wdm_unet_packed_bytes(u);
    void* packed; CK(hipMalloc(&packed, pb));
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, (uint32_t*)packed, pb / 4, 12345u);
    WK(wdm_unet_set_packed(u, packed, pb));
    WK(wdm_unet_mark_loaded(u));
    const size_t wsb = wdm_unet_workspace_bytes(u, B);
    void* ws; CK(hipMalloc(&ws, wsb));
    const size_t es = cfg.dtype == WDM_BF16 ? 2 : 4;
    const size_t nx = (size_t)B * R * R * 96 * es;
    void* x96; CK(hipMalloc(&x96, nx));
    if (es == 2) hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, (uint32_t*)x96, nx / 4, 777u);
    else CK(hipMemset(x96, 0, nx));
    float* t; CK(hipMalloc(&t, 4)); const float t0 = 500.f; CK(hipMemcpy(t, &t0, 4, hipMemcpyHostToDevice));
    float* eps; CK(hipMalloc(&eps, (size_t)B * 3 * R * R * 4));
    CK(hipDeviceSynchronize());
    printf("UNet: %d params, packed %.0f MB, workspace %.0f MB, B = %d, R = %d\n", wdm_unet_num_params(u), pb / 1e6, wsb / 1e6, B, R);
    const bool prof = getenv("PROF") != nullptr;
    WK(wdm_unet_forward(u, x96, t, 1, B, eps, ws, wsb, nullptr));       // warm-up
    CK(hipDeviceSynchronize());
    if (prof) WK(wdm_prof_enable(1));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) WK(wdm_unet_forward(u, x96, t, 1, B, eps, ws, wsb, nullptr));
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%.3f ms per UNet call (%d calls)\n", ms / iters, iters);
    if (prof) {
        std::vector<wdm_prof_entry> ent(512);
        int n = 0;
        WK(wdm_prof_report(ent.data(), (int)ent.size(), &n));
        for (int i = 0; i < n; ++i)
            printf("{\"kernel\": \"%s\", \"launches\": %d, \"ms\": %.4f, \"flops\": %.6g, \"bytes\": %.6g}\n", ent[i].kernel, (int)ent[i].launches, ent[i].total_ms, ent[i].total_flops, ent[i].total_bytes);
    }
    std::vector<float> he(16);
    CK(hipMemcpy(he.data(), eps, 64, hipMemcpyDeviceToHost));
    printf("eps[0..3] = %g %g %g %g\n", he[0], he[1], he[2], he[3]);
    return 0;
}
