// Micro-benchmark: can VALU work hide behind MFMAs on a CDNA4 SIMD?  One workgroup per CU, W waves per SIMD.
//   mode 0: every wave issues NM MFMAs                      mode 1: every wave issues NV VALU ops (kind K)
//   mode 2: every wave interleaves 1 MFMA : R VALU          mode 3: even waves MFMA only, odd waves VALU only (same SIMD pairs)
// prints cycles per wave (s_memtime) for each mode
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int KIND> __device__ __forceinline__ void valu(float& a, float& b, float c) {
    if (KIND == 0) { a = __builtin_fmaf(a, c, b); }                                   // plain fma
    else if (KIND == 1) { a = __builtin_amdgcn_exp2f(a); }                            // transcendental
    else if (KIND == 2) { a = __builtin_amdgcn_rcpf(a); }
    else { typedef __attribute__((ext_vector_type(2))) float f2; f2 v = {a, b}; f2 w = {c, c}; v = __builtin_elementwise_fma(v, w, v); a = v.x; b = v.y; }   // packed
}

template <int MODE, int KIND, int R>
__global__ __launch_bounds__(512) void k(unsigned long long* out, float* sink, int iters) {
    const int wave = threadIdx.x >> 6;
    bf16x8 A, Bv;
    for (int i = 0; i < 8; ++i) { A[i] = (__bf16)(float)(threadIdx.x + i); Bv[i] = (__bf16)(float)(i + 1); }
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
    float v[8], w[8];
    for (int i = 0; i < 8; ++i) { v[i] = 0.5f + threadIdx.x * 1e-3f + i; w[i] = 1.0f; }
    const bool do_m = MODE == 0 || MODE == 2 || (MODE == 3 && (wave < 4));
    const bool do_v = MODE == 1 || MODE == 2 || (MODE == 3 && (wave >= 4));
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, Bv, acc[i], 0, 0, 0);
#pragma unroll
                for (int r = 0; r < R; ++r) valu<KIND>(v[(i + r) & 7], w[(i + r) & 7], 1.0001f);
            }
        } else {
            if (do_m) {
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, Bv, acc[i], 0, 0, 0);
            }
            if (do_v) {
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int r = 0; r < R; ++r) valu<KIND>(v[(i + r) & 7], w[(i + r) & 7], 1.0001f);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + v[i] + w[i];
    sink[blockIdx.x * 512 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) out[wave] = t1 - t0;
}

template <int MODE, int KIND, int R> void run(const char* name, unsigned long long* d, float* sink) {
    const int iters = 200000;
    hipLaunchKernelGGL((k<MODE, KIND, R>), dim3(256), dim3(512), 0, 0, d, sink, iters);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k<MODE, KIND, R>), dim3(256), dim3(512), 0, 0, d, sink, iters);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (MODE == 0) printf("   [MFMA-only kernel: %.3f ms -> %.0f TFLOP/s sustained, %.2f ticks/ns]\n", ms, 256.0 * 8 * iters * 8 * 16384.0 / (ms * 1e-3) / 1e12, 0.0);
    unsigned long long h[8];
    CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
    printf("%-34s per 8-MFMA-group (or 8*R VALU): wave0 %.1f  wave4 %.1f ticks\n", name, (double)h[0] / iters, (double)h[4] / iters);
}

int main() {
    unsigned long long* d; float* sink;
    CK(hipMalloc(&d, 64)); CK(hipMalloc(&sink, 256 * 512 * 4));
    run<0, 0, 1>("MFMA only (2 waves/SIMD)", d, sink);
    run<1, 0, 1>("fma only R=1", d, sink);
    run<1, 1, 1>("exp only R=1", d, sink);
    run<1, 3, 1>("pk_fma only R=1", d, sink);
    run<2, 0, 1>("same wave MFMA + 1 fma", d, sink);
    run<2, 0, 2>("same wave MFMA + 2 fma", d, sink);
    run<2, 0, 3>("same wave MFMA + 3 fma", d, sink);
    run<2, 1, 1>("same wave MFMA + 1 exp", d, sink);
    run<2, 1, 2>("same wave MFMA + 2 exp", d, sink);
    run<2, 3, 1>("same wave MFMA + 1 pk_fma", d, sink);
    run<3, 0, 1>("split waves MFMA | 1 fma", d, sink);
    run<3, 0, 3>("split waves MFMA | 3 fma", d, sink);
    run<3, 1, 1>("split waves MFMA | 1 exp", d, sink);
    run<3, 1, 2>("split waves MFMA | 2 exp", d, sink);
    run<3, 3, 2>("split waves MFMA | 2 pk_fma", d, sink);
    return 0;
}
