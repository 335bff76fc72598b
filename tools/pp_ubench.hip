// Micro-benchmark for the ping-pong conv kernel (conv_pp_kernel.h): what can the SECOND wave of a SIMD issue while the first one
// streams MFMAs?  One 512-thread workgroup per CU: waves 0-3 (one per SIMD) run an MFMA stream, waves 4-7 (their SIMD partners) run a
// stream of "filler" instructions (plain VALU / transcendental / ds_read_b128 / the GroupNorm+SiLU unit transform of the conv kernel).
// Printed per configuration: cycles per MFMA of the MFMA wave and cycles per filler of the filler wave, alone and side by side.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/pp_ubench.hip -o tools/abl_pp_ubench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// MK: 0 = 16x16x32 (8 independent accumulators), 1 = 32x32x16 (4 independent accumulators)
// FK: 0 v_fma, 1 v_exp, 2 ds_read_b128 (16 per wait), 3 GN+SiLU of one 16-byte unit (ds_read_b128 + ~48 VALU + ds_write_b128)
// ROLE: 0 both streams side by side, 1 MFMA waves only (partners idle at the barrier), 2 filler waves only, 3 ONE wave per SIMD interleaves 1 MFMA : R fillers
template <int MK, int FK, int ROLE, int R, int PRIO>
__global__ __launch_bounds__(512) void k(unsigned long long* out, float* sink, int iters) {
    __shared__ __attribute__((aligned(16))) char lds[65536];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    bf16x8 A, Bv;
    for (int i = 0; i < 8; ++i) { A[i] = (__bf16)(float)((threadIdx.x * 7 + i) % 13 - 6); Bv[i] = (__bf16)(float)(i - 3); }
    f32x4 a4[8]; f32x16 a16[4];
    for (int i = 0; i < 8; ++i) a4[i] = f32x4{0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) a16[i][j] = 0.f;
    float v[8], w[8];
    for (int i = 0; i < 8; ++i) { v[i] = 0.5f + threadIdx.x * 1e-3f + i; w[i] = 1.0f; }
    for (int i = threadIdx.x; i < 16384; i += 512) ((float*)lds)[i] = (float)(i & 255) * 0.01f;
    uint4 u4 = make_uint4(0, 0, 0, 0);
    const bool do_m = ROLE == 4 ? true : ROLE == 3 ? wave < 4 : (ROLE != 2 && wave < 4);
    const bool do_f = (ROLE == 3 || ROLE == 4) ? false : (ROLE != 1 && wave >= 4);
    __syncthreads();
    if (PRIO && do_m) __builtin_amdgcn_s_setprio(PRIO);
    auto filler = [&](int i) __attribute__((always_inline)) {
        if (FK == 0) v[i & 7] = __builtin_fmaf(v[i & 7], 1.0001f, w[i & 7]);
        else if (FK == 1) v[i & 7] = __builtin_amdgcn_exp2f(v[i & 7]);
        else if (FK == 4) { unsigned long long t2[2]; asm volatile("ds_read_b128 %0, %1" : "=v"(*(__attribute__((ext_vector_type(4))) unsigned*)t2) : "v"((unsigned)((lane * 16 + i * 1024) & 65535)) : "memory"); if ((i & 7) == 7) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
        else if (FK == 5) { const int q = i & 3; if (q < 2) v[i & 7] = __builtin_fmaf(v[i & 7], 1.0001f, w[i & 7]); else if (q == 2) v[i & 7] = __builtin_amdgcn_exp2f(v[i & 7]); else v[i & 7] = __builtin_amdgcn_rcpf(v[i & 7]); }
        else if (FK == 2) { const uint4 t = *(const uint4*)(lds + ((lane * 16 + i * 1024) & 65535)); u4.x ^= t.x; u4.y ^= t.y; u4.z ^= t.z; u4.w ^= t.w; }
        else {
            uint4* p = (uint4*)(lds + ((wave * 8192 + lane * 16 + (i & 7) * 1024) & 65535));
            const uint4 x = *p;
            float f[8] = {__uint_as_float(x.x << 16), __uint_as_float(x.x & 0xffff0000u), __uint_as_float(x.y << 16), __uint_as_float(x.y & 0xffff0000u),
                          __uint_as_float(x.z << 16), __uint_as_float(x.z & 0xffff0000u), __uint_as_float(x.w << 16), __uint_as_float(x.w & 0xffff0000u)};
            unsigned pk[4];
            for (int e = 0; e < 8; e += 2) {
                const f32x2 xx = {f[e], f[e + 1]}, s2 = {v[e], v[e + 1]}, h2 = {w[e], w[e + 1]};
                const f32x2 t = xx * s2 + h2;
                f32x2 d = {__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
                d = d + 1.0f;
                const f32x2 r = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
                const f32x2 y = (t * -0.6931471805599453f) * r;
                typedef __bf16 v2b __attribute__((ext_vector_type(2)));
                pk[e / 2] = __builtin_bit_cast(unsigned, __builtin_convertvector(y, v2b));
            }
            *p = make_uint4(pk[0], pk[1], pk[2], pk[3]);
        }
    };
    // time-bounded streams: every wave runs until the deadline and counts its iterations, so both streams overlap for the whole window
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long WINDOW = (unsigned long long)iters * 128;
    unsigned long long n_it = 0, t1 = t0;
    constexpr int NM = MK == 0 ? 8 : 4;
    if (do_m || do_f) {
        do {
#pragma unroll 1
            for (int it = 0; it < 8; ++it) {
                if (ROLE == 3 || ROLE == 4) {
#pragma unroll
                    for (int i = 0; i < NM; ++i) {
                        if (MK == 0) a4[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, Bv, a4[i], 0, 0, 0);
                        else a16[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, Bv, a16[i], 0, 0, 0);
#pragma unroll
                        for (int r = 0; r < R; ++r) filler(i * R + r);
                    }
                } else if (do_m) {
                    if (MK == 0) {
#pragma unroll
                        for (int i = 0; i < 8; ++i) a4[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, Bv, a4[i], 0, 0, 0);
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i) a16[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, Bv, a16[i], 0, 0, 0);
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 16; ++i) filler(i);
                }
            }
            n_it += 8;
            t1 = __builtin_amdgcn_s_memtime();
        } while (t1 - t0 < WINDOW);
    }
    float s = u4.x + u4.y + u4.z + u4.w;
    for (int i = 0; i < 8; ++i) s += a4[i][0] + v[i] + w[i];
    for (int i = 0; i < 4; ++i) s += a16[i][0] + a16[i][5];
    sink[blockIdx.x * 512 + threadIdx.x] = s + ((float*)lds)[threadIdx.x];
    if (lane == 0 && blockIdx.x == 3) { out[wave * 2] = t1 - t0; out[wave * 2 + 1] = n_it; }
}

template <int MK, int FK, int ROLE, int R, int PRIO> void run(const char* name, unsigned long long* d, float* sink) {
    const int iters = 20000;
    for (int rep = 0; rep < 2; ++rep) { CK(hipMemset(d, 0, 128)); hipLaunchKernelGGL((k<MK, FK, ROLE, R, PRIO>), dim3(256), dim3(512), 0, 0, d, sink, iters); CK(hipDeviceSynchronize()); }
    unsigned long long h[16];
    CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
    const int nm = MK == 0 ? 8 : 4;
    const double cm = h[1] ? (double)h[0] / h[1] / nm : 0.0, cf = h[9] ? (double)h[8] / h[9] / 16 : 0.0;
    if (ROLE == 3) printf("%-58s %7.1f cyc per MFMA (+%d fillers each)\n", name, cm, R);
    else if (ROLE == 4) printf("%-58s %7.1f cyc per MFMA per wave = %7.1f per SIMD-MFMA (+%d fillers each), wave4 %7.1f\n", name, cm, cm / 2, R, h[9] ? (double)h[8] / h[9] / nm : 0.0);
    else printf("%-58s MFMA wave %7.1f cyc/MFMA   filler wave %7.1f cyc/filler\n", name, cm, cf);
}

int main() {
    unsigned long long* d; float* sink;
    CK(hipMalloc(&d, 128)); CK(hipMemset(d, 0, 128)); CK(hipMalloc(&sink, 256 * 512 * 4));
    printf("== alone\n");
    run<0, 0, 1, 0, 0>("16x16x32 MFMA waves alone", d, sink);
    run<1, 0, 1, 0, 0>("32x32x16 MFMA waves alone", d, sink);
    run<0, 0, 2, 0, 0>("fma waves alone", d, sink);
    run<0, 1, 2, 0, 0>("exp waves alone", d, sink);
    run<0, 2, 2, 0, 0>("ds_read_b128 waves alone", d, sink);
    run<0, 3, 2, 0, 0>("GN+SiLU unit (1 filler = one 16-byte unit) alone", d, sink);
    printf("== side by side, 16x16x32\n");
    run<0, 0, 0, 0, 0>("16x16x32 | fma", d, sink);
    run<0, 1, 0, 0, 0>("16x16x32 | exp", d, sink);
    run<0, 2, 0, 0, 0>("16x16x32 | ds_read_b128", d, sink);
    run<0, 3, 0, 0, 0>("16x16x32 | GN+SiLU unit", d, sink);
    run<0, 3, 0, 0, 1>("16x16x32 prio1 | GN+SiLU unit", d, sink);
    printf("== side by side, 32x32x16\n");
    run<1, 0, 0, 0, 0>("32x32x16 | fma", d, sink);
    run<1, 1, 0, 0, 0>("32x32x16 | exp", d, sink);
    run<1, 2, 0, 0, 0>("32x32x16 | ds_read_b128", d, sink);
    run<1, 3, 0, 0, 0>("32x32x16 | GN+SiLU unit", d, sink);
    run<1, 0, 0, 0, 1>("32x32x16 prio1 | fma", d, sink);
    run<1, 2, 0, 0, 1>("32x32x16 prio1 | ds_read_b128", d, sink);
    run<1, 3, 0, 0, 1>("32x32x16 prio1 | GN+SiLU unit", d, sink);
    printf("== one wave per SIMD, 1 MFMA : R fillers\n");
    run<1, 0, 3, 4, 0>("32x32x16 + 4 fma", d, sink);
    run<1, 0, 3, 8, 0>("32x32x16 + 8 fma", d, sink);
    run<1, 5, 3, 4, 0>("32x32x16 + 4 mix(2 fma, exp, rcp)", d, sink);
    run<1, 5, 3, 8, 0>("32x32x16 + 8 mix", d, sink);
    run<1, 4, 3, 1, 0>("32x32x16 + 1 pure ds_read_b128", d, sink);
    run<1, 4, 3, 2, 0>("32x32x16 + 2 pure ds_read_b128", d, sink);
    run<1, 4, 3, 4, 0>("32x32x16 + 4 pure ds_read_b128", d, sink);
    printf("== BOTH waves of every SIMD, 1 MFMA : R fillers (cycles per MFMA seen by one wave; the SIMD issues two in that time)\n");
    run<1, 0, 4, 0, 0>("32x32x16 x2 waves, no fillers", d, sink);
    run<1, 0, 4, 2, 0>("32x32x16 x2 waves + 2 fma", d, sink);
    run<1, 0, 4, 4, 0>("32x32x16 x2 waves + 4 fma", d, sink);
    run<1, 0, 4, 8, 0>("32x32x16 x2 waves + 8 fma", d, sink);
    run<1, 5, 4, 2, 0>("32x32x16 x2 waves + 2 mix", d, sink);
    run<1, 5, 4, 4, 0>("32x32x16 x2 waves + 4 mix", d, sink);
    run<1, 5, 4, 8, 0>("32x32x16 x2 waves + 8 mix", d, sink);
    run<1, 4, 4, 1, 0>("32x32x16 x2 waves + 1 pure ds_read", d, sink);
    run<1, 4, 4, 2, 0>("32x32x16 x2 waves + 2 pure ds_read", d, sink);
    run<0, 0, 4, 0, 0>("16x16x32 x2 waves, no fillers", d, sink);
    run<0, 0, 4, 1, 0>("16x16x32 x2 waves + 1 fma", d, sink);
    run<0, 0, 4, 2, 0>("16x16x32 x2 waves + 2 fma", d, sink);
    run<0, 5, 4, 2, 0>("16x16x32 x2 waves + 2 mix", d, sink);
    run<0, 4, 4, 1, 0>("16x16x32 x2 waves + 1 pure ds_read", d, sink);
    printf("== side by side again, pure ds_read partner\n");
    run<1, 4, 0, 0, 0>("32x32x16 | pure ds_read_b128", d, sink);
    run<1, 5, 0, 0, 0>("32x32x16 | mix", d, sink);
    return 0;
}
