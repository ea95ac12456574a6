// mfma_rate.hip -- issue rate of the legacy v_mfma_f32_16x16x16_f16 against gfx950's v_mfma_f32_16x16x32_f16 (and the bf16 pair), one wave per SIMD
// and four: does a K = 16 product cost the same issue time as a K = 32 one?  (Every head kernel uses the K = 16 shape: 16-channel chunks.)
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef short short4v __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void rate_kernel(float* out, int iters, unsigned long long* cyc) {
    floatx4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    const float s = (float)(threadIdx.x & 7) * 0.125f + 0.25f;
    half4 a4 = {(_Float16)s, (_Float16)(s * 0.5f), (_Float16)(-s), (_Float16)0.75f}, b4 = {(_Float16)0.5f, (_Float16)s, (_Float16)0.25f, (_Float16)(-0.5f)};
    half8 a8 = {a4[0], a4[1], a4[2], a4[3], a4[3], a4[2], a4[1], a4[0]}, b8 = {b4[0], b4[1], b4[2], b4[3], b4[1], b4[0], b4[3], b4[2]};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if constexpr (MODE == 0) acc[k] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc[k], 0, 0, 0);
                else if constexpr (MODE == 1) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, acc[k], 0, 0, 0);
                else if constexpr (MODE == 2) acc[k] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(short4v, a4), __builtin_bit_cast(short4v, b4), acc[k], 0, 0, 0);
                else acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a8), __builtin_bit_cast(bf16x8, b8), acc[k], 0, 0, 0);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
    out[blockIdx.x * 256 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
}

template <int MODE>
static void run(const char* name, int threads, double flop_per) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 1024 * 1024 * 4); hipMalloc(&cyc, 8);
    const int iters = 4000, blocks = 256 * 4;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(rate_kernel<MODE>, dim3(blocks), dim3(threads), 0, 0, out, 100, cyc);
    hipEventRecord(e0);
    hipLaunchKernelGGL(rate_kernel<MODE>, dim3(blocks), dim3(threads), 0, 0, out, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double n = (double)blocks * (threads / 64) * iters * 32;
    printf("%-34s %d waves/wg x %d wgs: %.3f ms  %.1f TF/s   %.1f shader cycles per MFMA and wave (one wave's view)\n", name, threads / 64, blocks, ms,
           n * flop_per / ms / 1e9, (double)c / (iters * 32.0));
    hipFree(out); hipFree(cyc);
}
int main() {
    run<0>("v_mfma_f32_16x16x16_f16 (legacy)", 256, 8192.0);
    run<1>("v_mfma_f32_16x16x32_f16 (gfx950)", 256, 16384.0);
    run<2>("v_mfma_f32_16x16x16_bf16 (legacy)", 256, 8192.0);
    run<3>("v_mfma_f32_16x16x32_bf16 (gfx950)", 256, 16384.0);
    return 0;
}
