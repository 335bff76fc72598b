// Output side of DiffusiveRestoration.restore: image metrics and 8-bit conversion on the device, so that only
// two doubles and one byte per sample cross PCIe.
//   * wdm_image_sqdiff: per image, sum over pixels of (clamp(a)-clamp(b))^2 over the 3 channels (utils/metrics.py:7-11 torchPSNR)
//     and of (Y(a)-Y(b))^2 with Y = (24.966 c0 + 128.553 c1 + 65.481 c2 + 16)/255 (utils/metrics.py:30-51, :152-165: the
//     reference applies the "bgr" weights to the tensor's channels in storage order; so does this).  fp64 accumulation, one
//     workgroup per image, fixed reduction order.
//   * wdm_to_u8_hwc: torchvision.utils.save_image's quantisation x*255 + 0.5, clamp [0,255], truncate; NCHW f32 -> NHWC u8.
#include "common.h"

namespace wdm {

__global__ __launch_bounds__(1024) void image_sqdiff_kernel(const float* __restrict__ a, const float* __restrict__ b, int HW, double* __restrict__ out) {
    __shared__ double red[2][1024];
    const int img = blockIdx.x, tid = threadIdx.x;
    const float* pa = a + (long long)img * 3 * HW;
    const float* pb = b + (long long)img * 3 * HW;
    double s_rgb = 0.0, s_y = 0.0;
    for (int p = tid; p < HW; p += 1024) {
        const float a0 = pa[p], a1 = pa[HW + p], a2 = pa[2 * HW + p];
        const float b0 = pb[p], b1 = pb[HW + p], b2 = pb[2 * HW + p];
        const float d0 = fminf(fmaxf(a0, 0.f), 1.f) - fminf(fmaxf(b0, 0.f), 1.f);
        const float d1 = fminf(fmaxf(a1, 0.f), 1.f) - fminf(fmaxf(b1, 0.f), 1.f);
        const float d2 = fminf(fmaxf(a2, 0.f), 1.f) - fminf(fmaxf(b2, 0.f), 1.f);
        s_rgb += (double)d0 * d0 + (double)d1 * d1 + (double)d2 * d2;
        const double ya = (24.966 * (double)a0 + 128.553 * (double)a1 + 65.481 * (double)a2 + 16.0) / 255.0;
        const double yb = (24.966 * (double)b0 + 128.553 * (double)b1 + 65.481 * (double)b2 + 16.0) / 255.0;
        s_y += (ya - yb) * (ya - yb);
    }
    red[0][tid] = s_rgb; red[1][tid] = s_y;
    __syncthreads();
    for (int o = 512; o >= 1; o >>= 1) {
        if (tid < o) { red[0][tid] += red[0][tid + o]; red[1][tid] += red[1][tid + o]; }
        __syncthreads();
    }
    if (tid == 0) { out[img * 2] = red[0][0]; out[img * 2 + 1] = red[1][0]; }
}

__global__ __launch_bounds__(256) void to_u8_hwc_kernel(const float* __restrict__ x, uint8_t* __restrict__ y, int C, int HW, long long total) {
    for (long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(id % C);
        const long long bp = id / C;
        const long long b = bp / HW;
        const int p = (int)(bp - b * HW);
        float v = x[(b * C + c) * HW + p] * 255.0f + 0.5f;
        v = fminf(fmaxf(v, 0.f), 255.f);
        y[id] = (uint8_t)v;                       // truncation, like Tensor.to(torch.uint8)
    }
}

}  // namespace wdm

using namespace wdm;

extern "C" {

int wdm_image_sqdiff(wdm_handle* h, const float* a, const float* b, int B, int H, int W, double* sums, void* stream) {
    if (!h || !a || !b || !sums) WDM_FAIL(WDM_EINVAL, "wdm_image_sqdiff: null argument");
    if (B <= 0 || H <= 0 || W <= 0) WDM_FAIL(WDM_EINVAL, "wdm_image_sqdiff: bad size");
    hipLaunchKernelGGL(image_sqdiff_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, a, b, H * W, sums);
    WDM_HIP(hipGetLastError());
    return WDM_OK;
}

int wdm_to_u8_hwc(wdm_handle* h, const float* x, int B, int C, int H, int W, uint8_t* y, void* stream) {
    if (!h || !x || !y) WDM_FAIL(WDM_EINVAL, "wdm_to_u8_hwc: null argument");
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0) WDM_FAIL(WDM_EINVAL, "wdm_to_u8_hwc: bad size");
    const long long total = (long long)B * C * H * W;
    const long long nb = (total + 255) / 256;
    hipLaunchKernelGGL(to_u8_hwc_kernel, dim3((unsigned)(nb > 16384 ? 16384 : nb)), dim3(256), 0, (hipStream_t)stream, x, y, C, H * W, total);
    WDM_HIP(hipGetLastError());
    return WDM_OK;
}

}  // extern "C"
