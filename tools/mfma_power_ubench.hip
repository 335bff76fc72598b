// Which MFMA shape does the board sustain better on RANDOM 16-bit operands (power-limited regime)?  (not part of the library)
// Every CU: 8 waves, each 4096 x [64 MFMA-units of work] from registers holding random bf16 / fp16 data; no memory traffic in the loop.
//   (round 5, second half: + an LDS-fed mode, kl<NA, NB>: the same MFMA stream with every operand a fresh ds_read_b128 fragment -- the roof of an LDS-fed kernel)
//   mode 0: v_mfma_f32_16x16x32_bf16   1: v_mfma_f32_32x32x16_bf16   2: v_mfma_f32_16x16x32_f16   3: v_mfma_f32_32x32x16_f16     ZERO=1: all-zero operands
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_power_ubench.hip -o tools/abl_mfma_power
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(512, 2) void k(const uint4* src, int iters, float* sink) {
    const int t = blockIdx.x * 512 + threadIdx.x;
    uint4 a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = src[(t * 16 + i) & 0xFFFFF]; b[i] = src[(t * 16 + 8 + i) & 0xFFFFF]; }
    float r = 0.f;
    if (MODE == 0 || MODE == 2) {
        f32x4 acc[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int rep = 0; rep < 4; ++rep)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    if (MODE == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[(i + rep) & 7]), __builtin_bit_cast(bf16x8, b[i & 7]), acc[i], 0, 0, 0);
                    else acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a[(i + rep) & 7]), __builtin_bit_cast(f16x8, b[i & 7]), acc[i], 0, 0, 0);
                }
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) r += acc[i][0] + acc[i][3];
    } else {
        f32x16 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int rep = 0; rep < 8; ++rep)                  // 32 MFMAs of 2 units each = 64 units
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (MODE == 1) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[(i + rep) & 7]), __builtin_bit_cast(bf16x8, b[(i * 2 + rep) & 7]), acc[i], 0, 0, 0);
                    else acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[(i + rep) & 7]), __builtin_bit_cast(f16x8, b[(i * 2 + rep) & 7]), acc[i], 0, 0, 0);
                }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) r += acc[i][0] + acc[i][15];
    }
    if (r == 123.456f) sink[t] = r;
}


// LDS-fed variant (round 5): the same MFMA stream, but every iteration's operands are fresh ds_read_b128 fragments of random data in LDS (64 KB per workgroup, two
// workgroups per CU; contiguous 1 KB per wave-instruction: conflict-free) -- NA + NB fragment reads for NA x NB MFMAs, the next iteration's reads issued before this
// iteration's MFMAs.  No barriers, no global traffic: what an LDS-fed 16x16x32 loop can sustain at 0.5 (4 + 4) or 0.375 (4 + 8) reads per MFMA.
template <int NA, int NB, bool F16>
__global__ __launch_bounds__(512, 2) void kl(const uint4* src, int iters, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 4096; i += 512) ((uint4*)smem)[i] = src[(blockIdx.x * 4096 + i) & 0xFFFFF];
    __syncthreads();
    f32x4 acc[NA][NB];
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    uint4 a[NA], b[NB], an[NA], bn[NB];
    auto ld = [&](int it, uint4 (&x)[NA], uint4 (&y)[NB]) __attribute__((always_inline)) {
        const int base = (it * (NA + NB) + wave * 5) * 1024 + lane * 16;
#pragma unroll
        for (int i = 0; i < NA; ++i) x[i] = *(const uint4*)(smem + ((base + i * 1024) & 0xFFFF));
#pragma unroll
        for (int j = 0; j < NB; ++j) y[j] = *(const uint4*)(smem + ((base + (NA + j) * 1024) & 0xFFFF));
    };
    ld(0, a, b);
    for (int it = 0; it < iters; ++it) {
        ld(it + 1, an, bn);
#pragma unroll
        for (int i = 0; i < NA; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                if (F16) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a[i]), __builtin_bit_cast(f16x8, b[j]), acc[i][j], 0, 0, 0);
                else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[i]), __builtin_bit_cast(bf16x8, b[j]), acc[i][j], 0, 0, 0);
            }
#pragma unroll
        for (int i = 0; i < NA; ++i) a[i] = an[i];
#pragma unroll
        for (int j = 0; j < NB; ++j) b[j] = bn[j];
    }
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) r += acc[i][j][0] + acc[i][j][3];
    if (r == 123.456f) sink[blockIdx.x * 512 + tid] = r;
}
template <int NA, int NB, bool F16>
static void run_l(const uint4* src, const char* what) {
    const int iters = 16384 / (NA * NB) * 4, grid = 256 * 2;
    CK(hipFuncSetAttribute((const void*)kl<NA, NB, F16>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((kl<NA, NB, F16>), dim3(grid), dim3(512), 65536, 0, src, iters, (float*)nullptr);
    CK(hipDeviceSynchronize());
    float best = 1e9f, sum = 0.f;
    for (int r = 0; r < 5; ++r) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((kl<NA, NB, F16>), dim3(grid), dim3(512), 65536, 0, src, iters, (float*)nullptr);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best; sum += ms;
    }
    const double flop = (double)grid * 8 * iters * (NA * NB) * 2.0 * 16 * 16 * 32;
    printf("%-44s best %7.3f ms  %6.0f TFLOP/s   mean %7.3f ms %6.0f TFLOP/s\n", what, best, flop / best / 1e9, sum / 5, flop / (sum / 5) / 1e9);
}

template <int MODE>
static void run(const uint4* src, const char* what) {
    const int iters = 4096, grid = 256 * 2;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(512), 0, 0, src, iters, (float*)nullptr);
    CK(hipDeviceSynchronize());
    float best = 1e9f, sum = 0.f;
    for (int r = 0; r < 5; ++r) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(512), 0, 0, src, iters, (float*)nullptr);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best; sum += ms;
    }
    const double flop = (double)grid * 8 * iters * 64 * 2.0 * 16 * 16 * 32;
    printf("%-28s best %7.3f ms  %6.0f TFLOP/s   mean %7.3f ms %6.0f TFLOP/s\n", what, best, flop / best / 1e9, sum / 5, flop / (sum / 5) / 1e9);
}

// HOLD=<seconds> MODE=reg|lds : one variant back to back for that long (scripts/mfma_power_probe.sh samples socket power and shader clock beside it)
template <class F>
static void hold(F launch, double flop_per_launch, double seconds, const char* what) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch(); CK(hipDeviceSynchronize());
    int n = 0; float total = 0.f;
    while (total < seconds * 1e3f) {
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < 20; ++i) launch();
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        total += ms; n += 20;
    }
    printf("%s: %d launches in %.2f s  %6.0f TFLOP/s\n", what, n, total / 1e3, flop_per_launch * n / total / 1e9);
}

int main() {
    const bool zero = getenv("ZERO") != nullptr;
    std::vector<unsigned short> h((size_t)(1 << 20) * 8);
    srand(7);
    for (auto& v : h) {                                    // bf16 / fp16 bit patterns of moderate magnitude, random signs and mantissas
        const unsigned short mant = rand() & 0x03ff, sign = (rand() & 1) << 15;
        v = zero ? 0 : (unsigned short)(sign | (14 + (rand() & 1)) << 10 | mant);       // fp16: exponent 14 / 15 (0.5 ... 2); as bf16 the same bits are tiny normal numbers with random mantissas
    }
    uint4* src; CK(hipMalloc(&src, h.size() * 2)); CK(hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    if (getenv("HOLD")) {
        const double sec = atof(getenv("HOLD"));
        const char* mode = getenv("MODE") ? getenv("MODE") : "lds";
        if (mode[0] == 'r') hold([&] { hipLaunchKernelGGL(k<0>, dim3(512), dim3(512), 0, 0, src, 4096, (float*)nullptr); }, 512.0 * 8 * 4096 * 64 * 2.0 * 16 * 16 * 32, sec, "register-resident 16x16x32 bf16");
        else {
            CK(hipFuncSetAttribute((const void*)kl<4, 8, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
            hold([&] { hipLaunchKernelGGL((kl<4, 8, false>), dim3(512), dim3(512), 65536, 0, src, 2048, (float*)nullptr); }, 512.0 * 8 * 2048 * 32 * 2.0 * 16 * 16 * 32, sec, "LDS-fed 16x16x32 bf16, 0.375 reads per MFMA");
        }
        return 0;
    }
    for (int rep = 0; rep < 2; ++rep) {
        run<0>(src, "16x16x32 bf16");
        run<1>(src, "32x32x16 bf16");
        run<2>(src, "16x16x32 f16");
        run<3>(src, "32x32x16 f16");
        run_l<4, 4, false>(src, "LDS-fed 16x16x32 bf16, 0.50 reads per MFMA");
        run_l<4, 8, false>(src, "LDS-fed 16x16x32 bf16, 0.375 reads per MFMA");
        run_l<4, 4, true>(src, "LDS-fed 16x16x32 f16, 0.50 reads per MFMA");
        run_l<4, 8, true>(src, "LDS-fed 16x16x32 f16, 0.375 reads per MFMA");
    }
    return 0;
}
