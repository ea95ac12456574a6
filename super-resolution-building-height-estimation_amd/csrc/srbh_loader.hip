// srbh_loader.hip -- loader-side tensor math of the training tiles on the device.
//
// Stands in for the per-sample numpy/torch code of reference BH_loader.py:361-392 (myImageFloder_S12_globe.__getitem__):
//   normalise:  (img - min) / (max - min) per band, then clip to datarange      (:361-369; the nearest x4 up / x0.25 down
//               round trip at :353,365 is the identity for exact factors and is therefore not materialised)
//   labels   :  build = buildhir[height]; weight = heightweight[build]           (:373-375)
//               height_aggre = aggregate_torch(height, 0.25)                     (:386, aggregate_utils.py:29-41)
//               weight_aggre = heightweight[buildhir[height_aggre.long()]]       (:389-391)
// One thread per 4x4 label cell produces the 16 full-resolution outputs and the aggregated pair.
#include "srbh_internal.h"

namespace {
using namespace srbh;

__global__ void label_prep_kernel(const unsigned char* __restrict__ height, int B, int H, int W,
                                  const unsigned char* __restrict__ lut, const float* __restrict__ cw,
                                  long long* build, float* height_f, float* weight, float* height_aggre,
                                  float* weight_aggre) {
    const int oh = H / 4, ow = W / 4;
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long total = (long)B * oh * ow;
    if (idx >= total) return;
    const int ox = idx % ow;
    long r = idx / ow;
    const int oy = r % oh;
    const int b = r / oh;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const long row = ((long)b * H + oy * 4 + i) * W + ox * 4;
        const uchar4 h4 = *(const uchar4*)(height + row);
        const unsigned char hv[4] = {h4.x, h4.y, h4.z, h4.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int cls = lut[hv[j]];
            build[row + j] = cls;
            height_f[row + j] = (float)hv[j];
            weight[row + j] = cw[cls];
            s1 += (float)hv[j];
            s2 += 1.f;                       // (data >= 0) is always true for uint8 labels
        }
    }
    const float ha = s1 / (s2 + 1e-10f);
    height_aggre[idx] = ha;
    int hi = (int)ha;                         // .long(): truncation
    hi = hi < 0 ? 0 : (hi > 255 ? 255 : hi);
    weight_aggre[idx] = cw[lut[hi]];
}

__global__ void normalize_clamp_kernel(const float* __restrict__ src, float* __restrict__ dst, long n, int C, long hw,
                                       const float* __restrict__ mins, const float* __restrict__ ranges, float lo, float hi,
                                       int clamp) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const int c = (int)((i / hw) % C);
        float v = (src[i] - mins[c]) / ranges[c];
        if (clamp) v = v < lo ? lo : (v > hi ? hi : v);
        dst[i] = v;
    }
}

}  // namespace

extern "C" int srbh_label_prep(const unsigned char* height, int B, int H, int W, const unsigned char* buildhir_lut,
                               const float* class_weight, long long* build, float* height_f, float* weight,
                               float* height_aggre, float* weight_aggre, void* stream) {
    SRBH_REQUIRE(height && buildhir_lut && class_weight && build && height_f && weight && height_aggre && weight_aggre,
                 "srbh_label_prep: null pointer");
    SRBH_REQUIRE(B > 0 && H > 0 && W > 0 && H % 4 == 0 && W % 4 == 0, "srbh_label_prep: H, W must be multiples of 4");
    long total = (long)B * (H / 4) * (W / 4);
    hipLaunchKernelGGL(label_prep_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, height, B, H, W,
                       buildhir_lut, class_weight, build, height_f, weight, height_aggre, weight_aggre);
    SRBH_HIP(hipGetLastError());
    return SRBH_OK;
}

extern "C" int srbh_normalize_clamp(const float* src, float* dst, int B, int C, int H, int W, const float* mins,
                                    const float* ranges, float lo, float hi, int clamp, void* stream) {
    SRBH_REQUIRE(src && dst && mins && ranges && B > 0 && C > 0 && H > 0 && W > 0, "srbh_normalize_clamp: bad arguments");
    long n = (long)B * C * H * W;
    long blocks = (n + 255) / 256;
    hipLaunchKernelGGL(normalize_clamp_kernel, dim3(blocks < 4096 ? blocks : 4096), dim3(256), 0, (hipStream_t)stream, src,
                       dst, n, C, (long)H * W, mins, ranges, lo, hi, clamp);
    SRBH_HIP(hipGetLastError());
    return SRBH_OK;
}
