// srbh_aux.hip -- error plumbing, layout converters, weight packing and conv_first for libsrbh.
#include "srbh_internal.h"

namespace srbh {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int hip_fail(hipError_t e, const char* what) {
    set_error("HIP error %d (%s) in %s", (int)e, hipGetErrorString(e), what);
    return (int)e;
}

}  // namespace srbh

using namespace srbh;

extern "C" int srbh_version(void) { return 100; }
extern "C" const char* srbh_last_error(void) { return g_err; }

extern "C" size_t srbh_act16_bytes(int B, int C, int H, int W) {
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return 0;
    return act16_geo(B, (C + 31) / 32, H, W).total_b;
}

extern "C" size_t srbh_wpack16_bytes(int cout, int cin) {
    if (cout <= 0 || cin <= 0) return 0;
    return (size_t)((cin + 31) / 32) * 18 * ((cout + 31) / 32) * 1024;
}

namespace {

// ---- NCHW fp32 <-> ACT16 ----------------------------------------------------------------------
__global__ void nchw32_to_act16_kernel(const float* __restrict__ src, char* dst, int B, int C, int H, int W,
                                       int chunks, int row_b, int plane_b, long img_b) {
    // one thread per (b, chunk, y, x); writes the 32-channel record (zero for channels >= C)
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long total = (long)B * chunks * H * W;
    if (idx >= total) return;
    int x = idx % W;
    long r = idx / W;
    int y = r % H;
    r /= H;
    int ch = r % chunks;
    int b = r / chunks;
    _Float16* o = (_Float16*)(dst + b * img_b + (long)ch * plane_b + (long)(y + 1) * row_b + (x + 1) * PIX_B);
    for (int c = 0; c < 32; ++c) {
        int cc = ch * 32 + c;
        float v = cc < C ? src[(((long)b * C + cc) * H + y) * W + x] : 0.f;
        o[c] = (_Float16)v;
    }
}

__global__ void act16_to_nchw32_kernel(const char* src, float* __restrict__ dst, int B, int C, int H, int W,
                                       int row_b, int plane_b, long img_b) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long total = (long)B * C * H * W;
    if (idx >= total) return;
    int x = idx % W;
    long r = idx / W;
    int y = r % H;
    r /= H;
    int c = r % C;
    int b = r / C;
    const _Float16* s =
        (const _Float16*)(src + b * img_b + (long)(c >> 5) * plane_b + (long)(y + 1) * row_b + (x + 1) * PIX_B);
    dst[idx] = (float)s[c & 31];
}

// ---- OIHW fp32 -> WPACK16:  [chunk][tap][ks][mb][lane][8] ------------------------------------------
__global__ void pack_w_kernel(const float* __restrict__ w, _Float16* __restrict__ out, int cout, int cin, int nchunk,
                              int nmb) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;  // one per packed half
    long total = (long)nchunk * 18 * nmb * 512;
    if (idx >= total) return;
    int j = idx & 7;
    int lane = (idx >> 3) & 63;
    long f = idx >> 9;
    int mb = f % nmb;
    f /= nmb;
    int ks = f & 1;
    f >>= 1;
    int tap = f % 9;
    int chunk = f / 9;
    int oc = mb * 32 + (lane & 31);
    int ic = chunk * 32 + ks * 16 + (lane >> 5) * 8 + j;
    float v = (oc < cout && ic < cin) ? w[((long)oc * cin + ic) * 9 + tap] : 0.f;
    out[idx] = (_Float16)v;
}

// ---- conv_first: fp32 direct conv, few input channels ------------------------------------------------
// One thread per (pixel, group of 4 output channels); input taps are re-read through L1/L2 (tiny op).
__global__ void conv_first_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                  const float* __restrict__ bias, int B, int cin, int H, int W, float* ra, float* rb,
                                  float* rc, char* out16, int row_b, int plane_b, long img_b) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long total = (long)B * H * W * 16;
    if (idx >= total) return;
    int og = idx & 15;  // output channels og*4 .. +3
    long pix = idx >> 4;
    int xx = pix % W;
    long r = pix / W;
    int yy = r % H;
    int b = r / H;
    float acc[4];
    for (int q = 0; q < 4; ++q) acc[q] = bias ? bias[og * 4 + q] : 0.f;
    for (int ic = 0; ic < cin; ++ic) {
        const float* xp = x + ((long)b * cin + ic) * H * W;
        for (int ky = 0; ky < 3; ++ky) {
            int sy = yy + ky - 1;
            if (sy < 0 || sy >= H) continue;
            for (int kx = 0; kx < 3; ++kx) {
                int sx = xx + kx - 1;
                if (sx < 0 || sx >= W) continue;
                float v = xp[(long)sy * W + sx];
                for (int q = 0; q < 4; ++q) acc[q] = fmaf(v, w[((long)(og * 4 + q) * cin + ic) * 9 + ky * 3 + kx], acc[q]);
            }
        }
    }
    float4 v4 = make_float4(acc[0], acc[1], acc[2], acc[3]);
    if (ra) *(float4*)(ra + pix * 64 + og * 4) = v4;
    if (rb) *(float4*)(rb + pix * 64 + og * 4) = v4;
    if (rc) *(float4*)(rc + pix * 64 + og * 4) = v4;
    if (out16) {
        _Float16* o = (_Float16*)(out16 + b * img_b + (long)(og >> 3) * plane_b + (long)(yy + 1) * row_b +
                                  (xx + 1) * PIX_B + (og & 7) * 8);
        for (int q = 0; q < 4; ++q) o[q] = (_Float16)acc[q];
    }
}

}  // namespace

extern "C" int srbh_nchw32_to_act16(const float* src, void* dst, int B, int C, int H, int W, void* stream) {
    SRBH_REQUIRE(src && dst && B > 0 && C > 0 && H > 0 && W > 0, "srbh_nchw32_to_act16: bad arguments");
    int chunks = (C + 31) / 32;
    Act16Geo g = act16_geo(B, chunks, H, W);
    long total = (long)B * chunks * H * W;
    hipLaunchKernelGGL(nchw32_to_act16_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, src,
                       (char*)dst, B, C, H, W, chunks, g.row_b, g.plane_b, g.img_b);
    SRBH_HIP(hipGetLastError());
    return SRBH_OK;
}

extern "C" int srbh_act16_to_nchw32(const void* src, float* dst, int B, int C, int H, int W, void* stream) {
    SRBH_REQUIRE(src && dst && B > 0 && C > 0 && H > 0 && W > 0, "srbh_act16_to_nchw32: bad arguments");
    Act16Geo g = act16_geo(B, (C + 31) / 32, H, W);
    long total = (long)B * C * H * W;
    hipLaunchKernelGGL(act16_to_nchw32_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       (const char*)src, dst, B, C, H, W, g.row_b, g.plane_b, g.img_b);
    SRBH_HIP(hipGetLastError());
    return SRBH_OK;
}

extern "C" int srbh_pack_conv3x3_f16(const float* w, int cout, int cin, void* packed, void* stream) {
    SRBH_REQUIRE(w && packed && cout > 0 && cin > 0, "srbh_pack_conv3x3_f16: bad arguments");
    int nchunk = (cin + 31) / 32, nmb = (cout + 31) / 32;
    long total = (long)nchunk * 18 * nmb * 512;
    hipLaunchKernelGGL(pack_w_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, w,
                       (_Float16*)packed, cout, cin, nchunk, nmb);
    SRBH_HIP(hipGetLastError());
    return SRBH_OK;
}

extern "C" int srbh_conv_first_f32(const float* x, const float* w, const float* bias, int B, int cin, int H, int W,
                                   float* ra, float* rb, float* rc, void* out16, int out16_chunks_total,
                                   void* stream) {
    SRBH_REQUIRE(x && w && B > 0 && cin > 0 && H > 0 && W > 0, "srbh_conv_first_f32: bad arguments");
    SRBH_REQUIRE(!out16 || out16_chunks_total >= 2, "srbh_conv_first_f32: out16 needs >= 2 chunk planes");
    Act16Geo g = act16_geo(B, out16_chunks_total > 0 ? out16_chunks_total : 2, H, W);
    long total = (long)B * H * W * 16;
    hipLaunchKernelGGL(conv_first_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, w, bias, B,
                       cin, H, W, ra, rb, rc, (char*)out16, g.row_b, g.plane_b, g.img_b);
    SRBH_HIP(hipGetLastError());
    return SRBH_OK;
}
