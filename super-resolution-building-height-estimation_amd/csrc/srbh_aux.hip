// srbh_aux.hip -- error plumbing, layout converters, weight packing and conv_first for libsrbh.
#include "srbh_internal.h"

namespace srbh {

static thread_local char g_err[512] = "";

unsigned long long g_path_counters[PATH_N] = {};
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

__global__ void zero_words_kernel(unsigned* __restrict__ p, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = 0u;
}

int zero_async(void* p, size_t bytes, hipStream_t st) {
    if (!bytes) return SRBH_OK;
    if (!p || (bytes & 3) || ((uintptr_t)p & 3)) { set_error("zero_async: bad buffer"); return SRBH_ERR_ARG; }
    const size_t n = bytes / 4;
    const unsigned blocks = (unsigned)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(zero_words_kernel, dim3(blocks), dim3(256), 0, st, (unsigned*)p, n);
    SRBH_HIP(hipGetLastError());
    return SRBH_OK;
}

}  // namespace srbh

using namespace srbh;

extern "C" int srbh_version(void) { return 100; }
/* out[0..n): launches per form since the last reset (order: srbh.h SRBH_PATH_*); reset != 0 clears them.  Host-side counters, not
 * thread-safe beyond "approximately right": a diagnostic. */
extern "C" int srbh_path_counters(unsigned long long* out, int n, int reset) {
    for (int i = 0; i < n && i < srbh::PATH_N; ++i) if (out) out[i] = srbh::g_path_counters[i];
    if (reset) for (int i = 0; i < srbh::PATH_N; ++i) srbh::g_path_counters[i] = 0;
    return srbh::PATH_N;
}
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
template <int BF>
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
    if constexpr (BF) {          // bf16 (RNE) bits in the 16-bit slot
        const unsigned u = __builtin_bit_cast(unsigned, v);
        ((unsigned short*)out)[idx] = (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
    } else {
        out[idx] = (_Float16)v;
    }
}

// ---- NHWC fp32 [B][H][W][C] (C a multiple of 32) -> ACT16 chunk planes chunk0.. of a `chunks_total`-plane buffer, scaled, as fp16 or bf16
template <int BF>
__global__ void nhwc32_to_act16_kernel(const float* __restrict__ src, char* dst, int B, int C, int H, int W, int chunk0, float scale,
                                       int row_b, int plane_b, long img_b) {
    // one thread per (pixel, 8-channel octet): 32-byte read, 16-byte write
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int oct = C / 8;
    const long total = (long)B * H * W * oct;
    if (idx >= total) return;
    const int o8 = (int)(idx % oct);
    long r = idx / oct;
    const int x = (int)(r % W);
    r /= W;
    const int y = (int)(r % H);
    const int b = (int)(r / H);
    const float* s = src + (((long)b * H + y) * W + x) * C + o8 * 8;
    unsigned short hv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float v = s[j] * scale;
        if constexpr (BF) {
            const unsigned u = __builtin_bit_cast(unsigned, v);
            hv[j] = (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
        } else {
            const _Float16 h = (_Float16)v;
            hv[j] = __builtin_bit_cast(unsigned short, h);
        }
    }
    char* o = dst + (long)b * img_b + (long)(chunk0 + (o8 >> 2)) * plane_b + (long)(y + 1) * row_b + (x + 1) * PIX_B + (o8 & 3) * 16;
    uint4 pk = {(unsigned)hv[0] | ((unsigned)hv[1] << 16), (unsigned)hv[2] | ((unsigned)hv[3] << 16),
                (unsigned)hv[4] | ((unsigned)hv[5] << 16), (unsigned)hv[6] | ((unsigned)hv[7] << 16)};
    *(uint4*)o = pk;
}

// per-channel sums over (B, H, W) of `nchunk` ACT16 planes (bias gradients of the RRDBNet training path); fp16 or bf16 elements.
// A thread owns one 16-byte octet (8 channels) of the pixel records it walks: whole 64-byte records per 4 lanes.
template <int BF>
__global__ __launch_bounds__(256) void act16_channel_sum_kernel(const char* __restrict__ src, int B, int H, int W, int nchunk, int row_b,
                                                                 int plane_b, long img_b, float* __restrict__ out /* [nchunk*32], zeroed */) {
    __shared__ float red[32];
    if (threadIdx.x < 32) red[threadIdx.x] = 0.f;
    __syncthreads();
    const int oct = threadIdx.x & 3, pl = threadIdx.x >> 2;          // 64 pixel lanes x 4 octets
    const int chunk = blockIdx.y;
    const long npix = (long)B * H * W;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (long px = (long)blockIdx.x * 64 + pl; px < npix; px += (long)gridDim.x * 64) {
        const int x = (int)(px % W);
        const long r = px / W;
        const int y = (int)(r % H), b = (int)(r / H);
        const uint4 v = *(const uint4*)(src + (long)b * img_b + (long)chunk * plane_b + (long)(y + 1) * row_b + (x + 1) * PIX_B + oct * 16);
        const unsigned w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if constexpr (BF) {
                acc[2 * k] += __builtin_bit_cast(float, w4[k] << 16);
                acc[2 * k + 1] += __builtin_bit_cast(float, w4[k] & 0xffff0000u);
            } else {
                acc[2 * k] += (float)__builtin_bit_cast(_Float16, (unsigned short)(w4[k] & 0xffffu));
                acc[2 * k + 1] += (float)__builtin_bit_cast(_Float16, (unsigned short)(w4[k] >> 16));
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        float a = acc[k];
#pragma unroll
        for (int m = 4; m < 64; m <<= 1) a += __shfl_xor(a, m);       // over the 16 pixel lanes of this wave with the same octet
        if ((threadIdx.x & 63) < 4) atomicAdd(&red[oct * 8 + k], a);
    }
    __syncthreads();
    if (threadIdx.x < 32) atomicAdd(out + chunk * 32 + threadIdx.x, red[threadIdx.x]);
}

// ---- conv_first: fp32 direct conv, few input channels ------------------------------------------------
// Thread = (4 consecutive pixels of a row, 4 consecutive output channels); 16 lanes cover the 64 channels of a pixel
// group so every store instruction writes whole 256-byte pixel records.  Weights sit in LDS as [ic*9+tap][64] and are
// read as float4 broadcasts; each input value is loaded once and reused for the 4 channels x up to 3 pixels it touches.
template <int CIN>
__global__ __launch_bounds__(256) void conv_first_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, int B, int cin_rt, int H, int W,
                                                         float* ra, float* rb, float* rc, char* out16, int row_b,
                                                         int plane_b, long img_b) {
    extern __shared__ float s_w[];   // [cin*9][64]
    const int cin = CIN > 0 ? CIN : cin_rt;
    for (int u = threadIdx.x; u < cin * 9 * 64; u += 256) {
        const int oc = u & 63, k = u >> 6;            // k = ic*9 + tap
        s_w[u] = w[(long)oc * cin * 9 + k];
    }
    __syncthreads();
    const int wq = (W + 3) >> 2;                      // pixel groups per row
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)B * H * wq * 16;
    if (idx >= total) return;
    const int og = idx & 15;
    long r = idx >> 4;
    const int xg = r % wq; r /= wq;
    const int yy = r % H;
    const int b = r / H;
    const int x0 = xg * 4;
    float acc[4][4];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[p][q] = bias ? bias[og * 4 + q] : 0.f;
    for (int ic = 0; ic < cin; ++ic) {
        const float* xp = x + ((long)b * cin + ic) * H * W;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int sy = yy + ky - 1;
            float in[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const int sx = x0 + j - 1;
                in[j] = (sy >= 0 && sy < H && sx >= 0 && sx < W) ? xp[(long)sy * W + sx] : 0.f;
            }
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const float4 wv = *(const float4*)(s_w + (ic * 9 + ky * 3 + kx) * 64 + og * 4);
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const float v = in[p + kx];
                    acc[p][0] = fmaf(v, wv.x, acc[p][0]);
                    acc[p][1] = fmaf(v, wv.y, acc[p][1]);
                    acc[p][2] = fmaf(v, wv.z, acc[p][2]);
                    acc[p][3] = fmaf(v, wv.w, acc[p][3]);
                }
            }
        }
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int xx = x0 + p;
        if (xx >= W) break;
        const long pix = ((long)b * H + yy) * W + xx;
        const float4 v4 = make_float4(acc[p][0], acc[p][1], acc[p][2], acc[p][3]);
        if (ra) *(float4*)(ra + pix * 64 + og * 4) = v4;
        if (rb) *(float4*)(rb + pix * 64 + og * 4) = v4;
        if (rc) *(float4*)(rc + pix * 64 + og * 4) = v4;
        if (out16) {
            _Float16* o = (_Float16*)(out16 + b * img_b + (long)(og >> 3) * plane_b + (long)(yy + 1) * row_b +
                                      (xx + 1) * PIX_B + (og & 7) * 8);
            typedef _Float16 half4 __attribute__((ext_vector_type(4)));
            *(half4*)o = half4{(_Float16)acc[p][0], (_Float16)acc[p][1], (_Float16)acc[p][2], (_Float16)acc[p][3]};
        }
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
    hipLaunchKernelGGL(pack_w_kernel<0>, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, w,
                       (_Float16*)packed, cout, cin, nchunk, nmb);
    SRBH_HIP(hipGetLastError());
    return SRBH_OK;
}

extern "C" int srbh_pack_conv3x3_b16(const float* w, int cout, int cin, void* packed, void* stream) {
    SRBH_REQUIRE(w && packed && cout > 0 && cin > 0, "srbh_pack_conv3x3_b16: bad arguments");
    int nchunk = (cin + 31) / 32, nmb = (cout + 31) / 32;
    long total = (long)nchunk * 18 * nmb * 512;
    hipLaunchKernelGGL(pack_w_kernel<1>, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, w,
                       (_Float16*)packed, cout, cin, nchunk, nmb);
    SRBH_HIP(hipGetLastError());
    return SRBH_OK;
}

// ---- the same packs for MANY convs in one launch (a generator whose weights move every iteration: 351 forward packs, 345 gradient packs)
__global__ __launch_bounds__(256) void pack_w_many_kernel(const srbh_pack3x3_desc* __restrict__ table) {
    const srbh_pack3x3_desc d = table[blockIdx.y];
    const int nchunk = (d.cin + 31) / 32, nmb = (d.cout + 31) / 32;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (d.bias_dst && idx < d.cout) d.bias_dst[idx] = d.bias_src[idx];
    const long total = (long)nchunk * 18 * nmb * 512;
    if (idx >= total) return;
    const int j = idx & 7, lane = (idx >> 3) & 63;
    long f = idx >> 9;
    const int mb = f % nmb; f /= nmb;
    const int ks = f & 1; f >>= 1;
    const int tap = f % 9;
    const int chunk = (int)(f / 9);
    const int oc = mb * 32 + (lane & 31), ic = chunk * 32 + ks * 16 + (lane >> 5) * 8 + j;
    const float v = (oc < d.cout && ic < d.cin) ? d.w[((long)oc * d.cin + ic) * 9 + tap] : 0.f;
    if (d.bf16) {
        const unsigned u = __builtin_bit_cast(unsigned, v);
        ((unsigned short*)d.packed)[idx] = (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
    } else {
        ((_Float16*)d.packed)[idx] = (_Float16)v;
    }
}

extern "C" int srbh_pack_conv3x3_many(const srbh_pack3x3_desc* table_dev, int n, long max_elems, void* stream) {
    SRBH_REQUIRE(table_dev && n > 0 && max_elems > 0, "srbh_pack_conv3x3_many: bad arguments");
    hipLaunchKernelGGL(pack_w_many_kernel, dim3((unsigned)((max_elems + 255) / 256), n), dim3(256), 0, (hipStream_t)stream, table_dev);
    SRBH_HIP(hipGetLastError());
    return SRBH_OK;
}

extern "C" int srbh_nhwc32_to_act16(const float* src, void* dst, int B, int C, int H, int W, int chunks_total, int chunk0, float scale,
                                    int bf16, void* stream) {
    SRBH_REQUIRE(src && dst && B > 0 && C > 0 && (C & 31) == 0 && H > 0 && W > 0, "srbh_nhwc32_to_act16: bad arguments (C %% 32 == 0)");
    SRBH_REQUIRE(chunk0 >= 0 && chunk0 + C / 32 <= chunks_total, "srbh_nhwc32_to_act16: chunk range outside the buffer");
    const Act16Geo g = act16_geo(B, chunks_total, H, W);
    const long total = (long)B * H * W * (C / 8);
    if (bf16) hipLaunchKernelGGL(nhwc32_to_act16_kernel<1>, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, src, (char*)dst, B, C, H, W, chunk0, scale, g.row_b, g.plane_b, g.img_b);
    else hipLaunchKernelGGL(nhwc32_to_act16_kernel<0>, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, src, (char*)dst, B, C, H, W, chunk0, scale, g.row_b, g.plane_b, g.img_b);
    SRBH_HIP(hipGetLastError());
    return SRBH_OK;
}

__global__ void axpby_kernel(float4* __restrict__ dst, float a, const float4* __restrict__ x, float b, const float4* __restrict__ y, long n4) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float4 v = x[i];
        v.x *= a; v.y *= a; v.z *= a; v.w *= a;
        if (y) {
            const float4 w = y[i];
            v.x += b * w.x; v.y += b * w.y; v.z += b * w.z; v.w += b * w.w;
        }
        dst[i] = v;
    }
}

/* dst = a * x + b * y (y may be NULL; dst may alias x or y); n % 4 == 0, 16-byte aligned */
extern "C" int srbh_axpby_f32(float* dst, float a, const float* x, float b, const float* y, long n, void* stream) {
    SRBH_REQUIRE(dst && x && n > 0 && (n & 3) == 0, "srbh_axpby_f32: bad arguments");
    const long n4 = n / 4;
    const int blocks = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
    hipLaunchKernelGGL(axpby_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (float4*)dst, a, (const float4*)x, b, (const float4*)y, n4);
    SRBH_HIP(hipGetLastError());
    return SRBH_OK;
}

// g *= (y > 0 ? 1 : slope): the backward of y = leaky_relu(z) from the SAVED output (y > 0 <=> z > 0; torch's slope at z <= 0), in place, one pass
__global__ void lrelu_bwd_kernel(float4* __restrict__ g, const float4* __restrict__ y, float slope, long n4) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float4 v = g[i];
        const float4 w = y[i];
        v.x *= w.x > 0.f ? 1.f : slope; v.y *= w.y > 0.f ? 1.f : slope; v.z *= w.z > 0.f ? 1.f : slope; v.w *= w.w > 0.f ? 1.f : slope;
        g[i] = v;
    }
}

// adjoint of nearest-neighbour x2 on an NHWC fp32 tensor: out[b][y][x][c] = sum of the 2 x 2 block of g; one thread per 4 output channels
__global__ void up2_bwd_nhwc_kernel(const float4* __restrict__ g, float4* __restrict__ out, long n4, int Ho, int Wo, int C4) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4);
        long r = i / C4;
        const int x = (int)(r % Wo);
        r /= Wo;
        const int y = (int)(r % Ho);
        const long b = r / Ho;
        const long row = (long)2 * Wo * C4;
        const float4* q = g + ((b * 2 * Ho + 2 * y) * 2 * Wo + 2 * x) * C4 + c;
        const float4 a0 = q[0], a1 = q[C4], a2 = q[row], a3 = q[row + C4];
        out[i] = float4{(a0.x + a1.x) + (a2.x + a3.x), (a0.y + a1.y) + (a2.y + a3.y), (a0.z + a1.z) + (a2.z + a3.z), (a0.w + a1.w) + (a2.w + a3.w)};
    }
}

extern "C" int srbh_up2_bwd_nhwc_f32(const float* g, float* out, int B, int Ho, int Wo, int C, void* stream) {
    SRBH_REQUIRE(g && out && B > 0 && Ho > 0 && Wo > 0 && C > 0 && (C & 3) == 0, "srbh_up2_bwd_nhwc_f32: bad arguments (C %% 4 == 0)");
    const long n4 = (long)B * Ho * Wo * (C / 4);
    const int blocks = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
    hipLaunchKernelGGL(up2_bwd_nhwc_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float4*)g, (float4*)out, n4, Ho, Wo, C / 4);
    SRBH_HIP(hipGetLastError());
    return SRBH_OK;
}

// ---- F.interpolate(scale_factor=2, mode="bilinear", align_corners=False) on NHWC fp32 tensors (UNetDiscriminatorSN's up-sampling, SR/rrdbnet_arch.py:
// 285-297) and its adjoint.  In one dimension: out[2i] = 0.25 x[max(i-1, 0)] + 0.75 x[i], out[2i+1] = 0.75 x[i] + 0.25 x[min(i+1, n-1)]; the adjoint
// gathers dx[i] = 0.75 (g[2i] + g[2i+1]) + 0.25 (g[max(2i-1, 0)] + g[min(2i+2, 2n-1)]) -- no atomics (the stock backward scatters: 4.5 ms per trainer
// iteration at batch 8, profiles/r05cx).  One thread per 4 channels of an output element.
__global__ void bilinear2x_fwd_kernel(const float4* __restrict__ x, float4* __restrict__ out, long n4, int H, int W, int C4) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4);
        long r = i / C4;
        const int ox = (int)(r % (2 * W));
        r /= 2 * W;
        const int oy = (int)(r % (2 * H));
        const long b = r / (2 * H);
        const int iy = oy >> 1, ix = ox >> 1;
        const int y0 = (oy & 1) ? iy : max(iy - 1, 0), y1 = (oy & 1) ? min(iy + 1, H - 1) : iy;
        const int x0 = (ox & 1) ? ix : max(ix - 1, 0), x1 = (ox & 1) ? min(ix + 1, W - 1) : ix;
        const float wy0 = (oy & 1) ? 0.75f : 0.25f, wy1 = 1.f - wy0, wx0 = (ox & 1) ? 0.75f : 0.25f, wx1 = 1.f - wx0;
        const float4* p = x + b * H * W * C4 + c;
        const float4 a = p[((long)y0 * W + x0) * C4], bq = p[((long)y0 * W + x1) * C4], cq = p[((long)y1 * W + x0) * C4], d = p[((long)y1 * W + x1) * C4];
        float4 o;
        o.x = wy0 * (wx0 * a.x + wx1 * bq.x) + wy1 * (wx0 * cq.x + wx1 * d.x);
        o.y = wy0 * (wx0 * a.y + wx1 * bq.y) + wy1 * (wx0 * cq.y + wx1 * d.y);
        o.z = wy0 * (wx0 * a.z + wx1 * bq.z) + wy1 * (wx0 * cq.z + wx1 * d.z);
        o.w = wy0 * (wx0 * a.w + wx1 * bq.w) + wy1 * (wx0 * cq.w + wx1 * d.w);
        out[i] = o;
    }
}

__global__ void bilinear2x_bwd_kernel(const float4* __restrict__ g, float4* __restrict__ dx, long n4, int H, int W, int C4) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4);
        long r = i / C4;
        const int ix = (int)(r % W);
        r /= W;
        const int iy = (int)(r % H);
        const long b = r / H;
        const float4* p = g + b * 4 * H * W * C4 + c;
        const float wt[4] = {0.25f, 0.75f, 0.75f, 0.25f};
        float4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int gy = min(max(2 * iy - 1 + a, 0), 2 * H - 1);
            float4 row = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int gx = min(max(2 * ix - 1 + q, 0), 2 * W - 1);
                const float4 v = p[((long)gy * 2 * W + gx) * C4];
                row.x += wt[q] * v.x; row.y += wt[q] * v.y; row.z += wt[q] * v.z; row.w += wt[q] * v.w;
            }
            s.x += wt[a] * row.x; s.y += wt[a] * row.y; s.z += wt[a] * row.z; s.w += wt[a] * row.w;
        }
        dx[i] = s;
    }
}

extern "C" int srbh_bilinear2x_nhwc_f32(const float* x, float* out, int B, int H, int W, int C, int backward, void* stream) {
    SRBH_REQUIRE(x && out && B > 0 && H > 0 && W > 0 && C > 0 && (C & 3) == 0, "srbh_bilinear2x_nhwc_f32: bad arguments (C %% 4 == 0)");
    const long n4 = (long)B * H * W * (C / 4) * (backward ? 1 : 4);
    const int blocks = (int)((n4 + 255) / 256 < 8192 ? (n4 + 255) / 256 : 8192);
    if (backward) hipLaunchKernelGGL(bilinear2x_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float4*)x, (float4*)out, n4, H, W, C / 4);
    else hipLaunchKernelGGL(bilinear2x_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float4*)x, (float4*)out, n4, H, W, C / 4);
    SRBH_HIP(hipGetLastError());
    return SRBH_OK;
}

extern "C" int srbh_lrelu_bwd_f32(float* g, const float* y, float slope, long n, void* stream) {
    SRBH_REQUIRE(g && y && n > 0 && (n & 3) == 0, "srbh_lrelu_bwd_f32: bad arguments");
    const long n4 = n / 4;
    const int blocks = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
    hipLaunchKernelGGL(lrelu_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (float4*)g, (const float4*)y, slope, n4);
    SRBH_HIP(hipGetLastError());
    return SRBH_OK;
}

extern "C" int srbh_act16_channel_sum(const void* src, int B, int H, int W, int chunks_total, int chunk0, int nchunk, int bf16, float* out,
                                      void* stream) {
    SRBH_REQUIRE(src && out && B > 0 && H > 0 && W > 0 && nchunk > 0 && chunk0 >= 0 && chunk0 + nchunk <= chunks_total, "srbh_act16_channel_sum: bad arguments");
    const Act16Geo g = act16_geo(B, chunks_total, H, W);
    if (int rc = zero_async(out, (size_t)nchunk * 32 * sizeof(float), (hipStream_t)stream)) return rc;
    const long npix = (long)B * H * W;
    const int gx = (int)((npix + 255) / 256 < 1024 ? (npix + 255) / 256 : 1024);
    const char* base = (const char*)src + (long)chunk0 * g.plane_b;
    if (bf16) hipLaunchKernelGGL(act16_channel_sum_kernel<1>, dim3(gx, nchunk), dim3(256), 0, (hipStream_t)stream, base, B, H, W, nchunk, g.row_b, g.plane_b, g.img_b, out);
    else hipLaunchKernelGGL(act16_channel_sum_kernel<0>, dim3(gx, nchunk), dim3(256), 0, (hipStream_t)stream, base, B, H, W, nchunk, g.row_b, g.plane_b, g.img_b, out);
    SRBH_HIP(hipGetLastError());
    return SRBH_OK;
}

extern "C" int srbh_conv_first_f32(const float* x, const float* w, const float* bias, int B, int cin, int H, int W,
                                   float* ra, float* rb, float* rc, void* out16, int out16_chunks_total,
                                   void* stream) {
    SRBH_REQUIRE(x && w && B > 0 && cin > 0 && H > 0 && W > 0, "srbh_conv_first_f32: bad arguments");
    SRBH_REQUIRE(!out16 || out16_chunks_total >= 2, "srbh_conv_first_f32: out16 needs >= 2 chunk planes");
    Act16Geo g = act16_geo(B, out16_chunks_total > 0 ? out16_chunks_total : 2, H, W);
    SRBH_REQUIRE(cin <= 56, "srbh_conv_first_f32: at most 56 input channels (got %d)", cin);
    long total = (long)B * H * ((W + 3) / 4) * 16;
    const size_t lds = (size_t)cin * 9 * 64 * sizeof(float);
    if (cin == 3)
        hipLaunchKernelGGL(conv_first_kernel<3>, dim3((total + 255) / 256), dim3(256), lds, (hipStream_t)stream, x, w, bias,
                           B, cin, H, W, ra, rb, rc, (char*)out16, g.row_b, g.plane_b, g.img_b);
    else {
        if (lds > 65536)
            SRBH_ONCE_PER_DEVICE(SRBH_HIP(hipFuncSetAttribute((const void*)conv_first_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 56 * 9 * 64 * 4)));
        hipLaunchKernelGGL(conv_first_kernel<0>, dim3((total + 255) / 256), dim3(256), lds, (hipStream_t)stream, x, w, bias,
                           B, cin, H, W, ra, rb, rc, (char*)out16, g.row_b, g.plane_b, g.img_b);
    }
    SRBH_HIP(hipGetLastError());
    return SRBH_OK;
}
