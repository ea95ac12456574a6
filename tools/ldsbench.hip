// ldsbench.hip -- developer micro-benchmark: LDS read throughput of the access patterns used by the conv kernels.
// hipcc --offload-arch=gfx950 -O3 tools/ldsbench.hip -o tools/ldsbench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
// data patterns: 0 = the original smooth ramp, 1 = pseudo-random fp16 pairs in [-0.5, 0.5) (matrix-core power depends on
// operand toggling: the real kernels see the second kind)
__device__ __forceinline__ float pattern(int i, int rnd) {
    if (!rnd) return (float)(i & 1023) * 1e-3f;
    unsigned h = (unsigned)i * 2654435761u;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    const _Float16 a = (_Float16)(((h & 0xffff) / 65536.f) - 0.5f), b = (_Float16)(((h >> 16) / 65536.f) - 0.5f);
    typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
    const half2_t v = {a, b};
    return __builtin_bit_cast(float, v);
}
__global__ void fill_kernel(float* p, long n, int rnd) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = pattern((int)i, rnd);
}
static int g_rnd = 0;

// mode 0: conflict-free linear (lane*16); 1: the pixel-fragment pattern (64-B pixel records, XOR-swizzled k-slot);
// 2: same without the swizzle; 3: ds_read_b64 x2 of the pixel pattern
template <int MODE, int WITH_MFMA>
__global__ __launch_bounds__(256, 1) void lds_read_kernel(unsigned long long* out, int iters, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 160 * 1024 / 4; i += 256) ((float*)smem)[i] = (float)i;
    __syncthreads();
    const int l31 = lane & 31, hi = lane >> 5;
    const int wr = wave >> 1, wc = wave & 1;
    int off[6];
    for (int r = 0; r < 6; ++r) {
        const int pc = wc * 32 + l31;
        if (MODE == 0) off[r] = (wave * 6 + r) * 1024 + lane * 16;
        else if (MODE == 2) off[r] = (wr * 4 + r) * 66 * 64 + pc * 64 + hi * 16;
        else off[r] = (wr * 4 + r) * 66 * 64 + pc * 64 + ((hi ^ ((pc >> 2) & 3)) << 4);
    }
    half8 acc = {0, 0, 0, 0, 0, 0, 0, 0};
    floatx16 c = {0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            half8 v[6];
#pragma unroll
            for (int r = 0; r < 6; ++r) v[r] = *(const half8*)(smem + off[r] + ((it + u) & 1) * 32);
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                if (WITH_MFMA) {
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(v[r], v[(r + 1) % 6], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(v[(r + 2) % 6], v[r], c, 0, 0, 0);
                } else {
                    acc += v[r];
                }
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) out[blockIdx.x] = t1 - t0;
    float s = 0;
    for (int q = 0; q < 8; ++q) s += (float)acc[q];
    for (int q = 0; q < 16; ++q) s += c[q];
    if (s == 12345.678f) sink[0] = s;
}

template <int MODE, int WITH_MFMA>
static void run(const char* name) {
    unsigned long long* d;
    float* sink;
    hipMalloc(&d, 256 * 8);
    hipMalloc(&sink, 4);
    hipFuncSetAttribute((const void*)lds_read_kernel<MODE, WITH_MFMA>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((lds_read_kernel<MODE, WITH_MFMA>), dim3(256), dim3(256), 160 * 1024, 0, d, iters, sink);
    hipDeviceSynchronize();
    unsigned long long h[256];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    double avg = 0;
    for (int i = 0; i < 256; ++i) avg += (double)h[i];
    avg /= 256;
    const double reads_per_wave = (double)iters * 4 * 6;
    printf("%-44s %8.1f cycles per ds_read_b128 per wave  (CU: %.1f B/clk)%s\n", name, avg / reads_per_wave,
           4.0 * 1024.0 / (avg / reads_per_wave), WITH_MFMA ? "  [2 MFMA 32x32x16 per read]" : "");
    hipFree(d);
    hipFree(sink);
}

// the conv kernels' group shape: NR reads (6 pixel-fragment rows + 3*CB weight fragments) and NM = 12*CB MFMAs per group,
// operands double-buffered (the reads of group g+1 fly under the MFMAs of group g), ACC independent accumulators
template <int NR, int NM, int ACC>
__global__ __launch_bounds__(256, 1) void group_kernel(unsigned long long* out, int iters, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 160 * 1024 / 4; i += 256) ((float*)smem)[i] = (float)(i & 1023) * 1e-3f;
    __syncthreads();
    const int l31 = lane & 31, hi = lane >> 5, wr = wave >> 1, wc = wave & 1;
    int off[NR];
    for (int r = 0; r < NR; ++r) {
        const int pc = wc * 32 + l31;
        off[r] = r < 6 ? (wr * 4 + r) * 66 * 64 + pc * 64 + ((hi ^ ((pc >> 2) & 3)) << 4) : 45056 + (r - 6) * 1024 + lane * 16;
    }
    floatx16 c[ACC];
    for (int a = 0; a < ACC; ++a) c[a] = floatx16{0};
    half8 v[2][NR];
    for (int r = 0; r < NR; ++r) v[0][r] = *(const half8*)(smem + off[r]);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
            for (int r = 0; r < NR; ++r) v[(u + 1) & 1][r] = *(const half8*)(smem + off[r] + ((it + u) & 1) * 64);
#pragma unroll
            for (int m = 0; m < NM; ++m)
                c[m % ACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v[u][6 + m % (NR - 6)], v[u][m % 6], c[m % ACC], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, NR, 0);
#pragma unroll
            for (int k = 0; k < NR; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, NM - NR, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) out[blockIdx.x] = t1 - t0;
    float s = 0;
    for (int a = 0; a < ACC; ++a)
        for (int q = 0; q < 16; ++q) s += c[a][q];
    if (s == 12345.678f) sink[0] = s;
}

// the same groups arranged as the kernels' K steps: barrier, first group's reads exposed, 6 groups, (no DMA)
template <int NR, int NM, int ACC, int ND>
__global__ __launch_bounds__(256, 1) void step_kernel(unsigned long long* out, int iters, float* sink, const char* gsrc, unsigned long long dmask, int rnd) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 160 * 1024 / 4; i += 256) ((float*)smem)[i] = pattern(i, rnd);
    __syncthreads();
    const int l31 = lane & 31, hi = lane >> 5, wr = wave >> 1, wc = wave & 1;
    int off[NR];
    for (int r = 0; r < NR; ++r) {
        const int pc = wc * 32 + l31;
        off[r] = r < 6 ? (wr * 4 + r) * 66 * 64 + pc * 64 + ((hi ^ ((pc >> 2) & 3)) << 4) : 45056 + (r - 6) * 1024 + lane * 16;
    }
    floatx16 c[ACC];
    for (int a = 0; a < ACC; ++a) c[a] = floatx16{0};
    half8 v[2][NR];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const char* sb = smem + (it & 1) * 81920;
        const unsigned dstl = (unsigned)(unsigned long long)(__attribute__((address_space(3))) const char*)(smem + ((it + 1) & 1) * 81920) + wave * 1024;
        const unsigned long long gb = (unsigned long long)gsrc + (unsigned long long)blockIdx.x * 81920 + (it & 3) * 20480;
        const unsigned dl = __builtin_amdgcn_readfirstlane(dstl);
        const unsigned glo = __builtin_amdgcn_readfirstlane((unsigned)gb), ghi = __builtin_amdgcn_readfirstlane((unsigned)(gb >> 32));
        const unsigned long long gbu = ((unsigned long long)ghi << 32) | glo;
#pragma unroll
        for (int r = 0; r < NR; ++r) v[0][r] = *(const half8*)(sb + off[r]);
#pragma unroll
        for (int u = 0; u < 6; ++u) {
#pragma unroll
            for (int i = 0; i < ND; ++i)
                if (i * 3 / ND == u) {
                    unsigned long long sv;
                    asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %1\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                                 "global_load_lds_dwordx4 %3, %4 sc1\n\ts_mov_b64 exec, %0"
                                 : "=&s"(sv) : "s"(dmask), "s"(dl + i * 4096), "v"(lane * 16 + wave * 1024 + i * 4096), "s"(gbu) : "memory", "m0");
                }
            if (u + 1 < 6) {
#pragma unroll
                for (int r = 0; r < NR; ++r) v[(u + 1) & 1][r] = *(const half8*)(sb + off[r] + (u + 1) * 16);
            }
#pragma unroll
            for (int m = 0; m < NM; ++m)
                c[m % ACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v[u & 1][6 + m % (NR - 6)], v[u & 1][m % 6], c[m % ACC], 0, 0, 0);
            if (u == 0) __builtin_amdgcn_sched_group_barrier(0x100, NR, 0);
            if (u + 1 < 6) {
#pragma unroll
                for (int k = 0; k < NR; ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, NM - NR, 0);
            } else {
                __builtin_amdgcn_sched_group_barrier(0x008, NM, 0);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) out[blockIdx.x] = t1 - t0;
    float s = 0;
    for (int a = 0; a < ACC; ++a)
        for (int q = 0; q < 16; ++q) s += c[a][q];
    if (s == 12345.678f) sink[0] = s;
}

template <int NR, int NM, int ACC, int ND>
static void run_step(const char* name, unsigned long long dmask) {
    unsigned long long* d;
    float* sink;
    hipMalloc(&d, 256 * 8);
    hipMalloc(&sink, 4);
    hipFuncSetAttribute((const void*)step_kernel<NR, NM, ACC, ND>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int iters = 400;
    char* gsrc;
    hipMalloc(&gsrc, 256 * 81920 + (1 << 20));
    hipLaunchKernelGGL(fill_kernel, dim3(1024), dim3(256), 0, 0, (float*)gsrc, (long)(256 * 81920 + (1 << 20)) / 4, g_rnd);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((step_kernel<NR, NM, ACC, ND>), dim3(256), dim3(256), 160 * 1024, 0, d, iters, sink, gsrc, dmask, g_rnd);
    hipDeviceSynchronize();
    unsigned long long h[256];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    double avg = 0;
    for (int i = 0; i < 256; ++i) avg += (double)h[i];
    avg /= 256;
    printf("%-44s %8.1f cycles per step of %d MFMA (issue floor %d) -> %.0f %%\n", name, avg / iters, 6 * NM, 6 * NM * 32,
           100.0 * 6 * NM * 32 / (avg / iters));
    hipFree(d);
    hipFree(sink);
}


// 8 waves per workgroup (2 per SIMD), each owning 2 rows x 32 pixels: NP = 4 pixel-fragment rows + 3*CB weight fragments
// feed 6*CB MFMAs per group; same staging volume per step as step_kernel (ND DMA per wave)
template <int CB, int ND>
__global__ __launch_bounds__(512, 1) void step8_kernel(unsigned long long* out, int iters, float* sink, const char* gsrc, unsigned long long dmask, int rnd) {
    constexpr int NP = 4, NR = NP + 3 * CB, NM = 6 * CB, ACC = 2 * CB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 160 * 1024 / 4; i += 512) ((float*)smem)[i] = pattern(i, rnd);
    __syncthreads();
    const int l31 = lane & 31, hi = lane >> 5, wr = wave >> 1, wc = wave & 1;
    int off[NR];
    for (int r = 0; r < NR; ++r) {
        const int pc = wc * 32 + l31;
        off[r] = r < NP ? (wr * 2 + r) * 66 * 64 + pc * 64 + ((hi ^ ((pc >> 2) & 3)) << 4) : 45056 + (r - NP) * 1024 + lane * 16;
    }
    floatx16 c[ACC];
    for (int a = 0; a < ACC; ++a) c[a] = floatx16{0};
    half8 v[2][NR];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const char* sb = smem + (it & 1) * 81920;
        const unsigned dstl = (unsigned)(unsigned long long)(__attribute__((address_space(3))) const char*)(smem + ((it + 1) & 1) * 81920) + wave * 1024;
        const unsigned long long gb = (unsigned long long)gsrc + (unsigned long long)blockIdx.x * 81920 + (it & 3) * 20480;
        const unsigned dl = __builtin_amdgcn_readfirstlane(dstl);
        const unsigned glo = __builtin_amdgcn_readfirstlane((unsigned)gb), ghi = __builtin_amdgcn_readfirstlane((unsigned)(gb >> 32));
        const unsigned long long gbu = ((unsigned long long)ghi << 32) | glo;
#pragma unroll
        for (int r = 0; r < NR; ++r) v[0][r] = *(const half8*)(sb + off[r]);
#pragma unroll
        for (int u = 0; u < 6; ++u) {
#pragma unroll
            for (int i = 0; i < ND; ++i)
                if (i * 3 / ND == u) {
                    unsigned long long sv;
                    asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %1\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                                 "global_load_lds_dwordx4 %3, %4 sc1\n\ts_mov_b64 exec, %0"
                                 : "=&s"(sv) : "s"(dmask), "s"(dl + i * 8192), "v"(lane * 16 + wave * 1024 + i * 8192), "s"(gbu) : "memory", "m0");
                }
            if (u + 1 < 6) {
#pragma unroll
                for (int r = 0; r < NR; ++r) v[(u + 1) & 1][r] = *(const half8*)(sb + off[r] + (u + 1) * 16);
            }
#pragma unroll
            for (int m = 0; m < NM; ++m)
                c[m % ACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v[u & 1][NP + m % (NR - NP)], v[u & 1][m % NP], c[m % ACC], 0, 0, 0);
            if (u == 0) __builtin_amdgcn_sched_group_barrier(0x100, NR, 0);
            if (u + 1 < 6) {
                constexpr int K = NR < NM ? NR : NM;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                if constexpr (NM > NR) __builtin_amdgcn_sched_group_barrier(0x008, NM - NR, 0);
                if constexpr (NR > NM) __builtin_amdgcn_sched_group_barrier(0x100, NR - NM, 0);
            } else {
                __builtin_amdgcn_sched_group_barrier(0x008, NM, 0);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) out[blockIdx.x] = t1 - t0;
    float s = 0;
    for (int a = 0; a < ACC; ++a)
        for (int q = 0; q < 16; ++q) s += c[a][q];
    if (s == 12345.678f) sink[0] = s;
}

template <int CB, int ND>
static void run_step8(const char* name, unsigned long long dmask) {
    unsigned long long* d;
    float* sink;
    hipMalloc(&d, 256 * 8);
    hipMalloc(&sink, 4);
    hipFuncSetAttribute((const void*)step8_kernel<CB, ND>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int iters = 400;
    char* gsrc;
    hipMalloc(&gsrc, 256 * 81920 + (1 << 20));
    hipLaunchKernelGGL(fill_kernel, dim3(1024), dim3(256), 0, 0, (float*)gsrc, (long)(256 * 81920 + (1 << 20)) / 4, g_rnd);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((step8_kernel<CB, ND>), dim3(256), dim3(512), 160 * 1024, 0, d, iters, sink, gsrc, dmask, g_rnd);
    hipDeviceSynchronize();
    unsigned long long h[256];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    double avg = 0;
    for (int i = 0; i < 256; ++i) avg += (double)h[i];
    avg /= 256;
    printf("%-52s %8.1f cycles per step (per-SIMD MFMA issue floor %d) -> %.0f %%\n", name, avg / iters, 2 * 6 * 6 * CB * 32,
           100.0 * 2 * 6 * 6 * CB * 32 / (avg / iters));
    hipFree(d);
    hipFree(sink);
    hipFree(gsrc);
}


// 8 waves per workgroup (2 per SIMD), SPLIT-K / split-channel-block form: a wave keeps the 4-row x 32-pixel tile of the 4-wave
// kernel (6 pixel-fragment rows + 3 weight fragments feed 12 MFMAs per group: the same 0.75 LDS reads per MFMA), and the two
// waves that own the same pixels divide the work along K (cout 32: wave half h runs the 3 groups of k-step h) or along the
// output channel blocks (cout 64: wave half h runs all 6 groups for channel block h).  NG = groups per wave and step.
template <int NG, int ND>
__global__ __launch_bounds__(512, 1) void step8k_kernel(unsigned long long* out, int iters, float* sink, const char* gsrc, unsigned long long dmask, int rnd) {
    constexpr int NP = 6, NR = 9, NM = 12, ACC = 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 160 * 1024 / 4; i += 512) ((float*)smem)[i] = pattern(i, rnd);
    __syncthreads();
    const int l31 = lane & 31, hi = lane >> 5, q = wave & 3, h = wave >> 2, wr = q >> 1, wc = q & 1;
    int off[NR];
    for (int r = 0; r < NR; ++r) {
        const int pc = wc * 32 + l31;
        off[r] = r < NP ? (wr * 4 + r) * 66 * 64 + pc * 64 + (((hi + 2 * h) ^ ((pc >> 2) & 3)) << 4) : 45056 + (h * 9 + r - NP) * 1024 + lane * 16;
    }
    floatx16 c[ACC];
    for (int a = 0; a < ACC; ++a) c[a] = floatx16{0};
    half8 v[2][NR];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const char* sb = smem + (it & 1) * 81920;
        const unsigned dstl = (unsigned)(unsigned long long)(__attribute__((address_space(3))) const char*)(smem + ((it + 1) & 1) * 81920) + wave * 1024;
        const unsigned long long gb = (unsigned long long)gsrc + (unsigned long long)blockIdx.x * 81920 + (it & 3) * 20480;
        const unsigned dl = __builtin_amdgcn_readfirstlane(dstl);
        const unsigned glo = __builtin_amdgcn_readfirstlane((unsigned)gb), ghi = __builtin_amdgcn_readfirstlane((unsigned)(gb >> 32));
        const unsigned long long gbu = ((unsigned long long)ghi << 32) | glo;
#pragma unroll
        for (int r = 0; r < NR; ++r) v[0][r] = *(const half8*)(sb + off[r]);
#pragma unroll
        for (int u = 0; u < NG; ++u) {
#pragma unroll
            for (int i = 0; i < ND; ++i)
                if (i * 2 / ND == u) {
                    unsigned long long sv;
                    asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %1\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                                 "global_load_lds_dwordx4 %3, %4 sc1\n\ts_mov_b64 exec, %0"
                                 : "=&s"(sv) : "s"(dmask), "s"(dl + i * 8192), "v"(lane * 16 + wave * 1024 + i * 8192), "s"(gbu) : "memory", "m0");
                }
            if (u + 1 < NG) {
#pragma unroll
                for (int r = 0; r < NR; ++r) v[(u + 1) & 1][r] = *(const half8*)(sb + off[r] + (u + 1) * 64);
            }
#pragma unroll
            for (int m = 0; m < NM; ++m)
                c[m % ACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v[u & 1][NP + m % (NR - NP)], v[u & 1][m % NP], c[m % ACC], 0, 0, 0);
            if (u == 0) __builtin_amdgcn_sched_group_barrier(0x100, NR, 0);
            if (u + 1 < NG) {
#pragma unroll
                for (int k = 0; k < NR; ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, NM - NR, 0);
            } else {
                __builtin_amdgcn_sched_group_barrier(0x008, NM, 0);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) out[blockIdx.x] = t1 - t0;
    float s = 0;
    for (int a = 0; a < ACC; ++a)
        for (int q2 = 0; q2 < 16; ++q2) s += c[a][q2];
    if (s == 12345.678f) sink[0] = s;
}

template <int NG, int ND>
static void run_step8k(const char* name, unsigned long long dmask) {
    unsigned long long* d;
    float* sink;
    hipMalloc(&d, 256 * 8);
    hipMalloc(&sink, 4);
    hipFuncSetAttribute((const void*)step8k_kernel<NG, ND>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int iters = 400;
    char* gsrc;
    hipMalloc(&gsrc, 256 * 81920 + (1 << 20));
    hipLaunchKernelGGL(fill_kernel, dim3(1024), dim3(256), 0, 0, (float*)gsrc, (long)(256 * 81920 + (1 << 20)) / 4, g_rnd);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((step8k_kernel<NG, ND>), dim3(256), dim3(512), 160 * 1024, 0, d, iters, sink, gsrc, dmask, g_rnd);
    hipDeviceSynchronize();
    unsigned long long h[256];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    double avg = 0;
    for (int i = 0; i < 256; ++i) avg += (double)h[i];
    avg /= 256;
    printf("%-60s %8.1f cycles per step (per-SIMD MFMA issue floor %d) -> %.0f %%\n", name, avg / iters, 2 * NG * 12 * 32,
           100.0 * 2 * NG * 12 * 32 / (avg / iters));
    hipFree(d);
    hipFree(sink);
    hipFree(gsrc);
}

template <int NR, int NM, int ACC>
static void run_group(const char* name) {
    unsigned long long* d;
    float* sink;
    hipMalloc(&d, 256 * 8);
    hipMalloc(&sink, 4);
    hipFuncSetAttribute((const void*)group_kernel<NR, NM, ACC>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int iters = 1000;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((group_kernel<NR, NM, ACC>), dim3(256), dim3(256), 160 * 1024, 0, d, iters, sink);
    hipDeviceSynchronize();
    unsigned long long h[256];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    double avg = 0;
    for (int i = 0; i < 256; ++i) avg += (double)h[i];
    avg /= 256;
    printf("%-44s %8.1f cycles per group of %d MFMA (issue floor %d) -> %.0f %% of the MFMA rate\n", name, avg / (iters * 2.0), NM, NM * 32,
           100.0 * NM * 32 / (avg / (iters * 2.0)));
    hipFree(d);
    hipFree(sink);
}

int main(int argc, char** argv) {
    for (g_rnd = 0; g_rnd < 2; ++g_rnd) {
        printf("---- operand data: %s\n", g_rnd ? "pseudo-random fp16 in [-0.5, 0.5)" : "smooth ramp");
        run_step8k<3, 1>("8 waves split-K, cout 32 STEP: barrier + 3 groups, ~no DMA", 0ull);
        run_step8k<3, 8>("8 waves split-K, cout 32 STEP + 8 DMA/wave masked off", 0ull);
        run_step8k<3, 8>("8 waves split-K, cout 32 STEP + 8 DMA/wave (64 KiB/step)", ~0ull);
        run_step8k<6, 1>("8 waves split-cb, cout 64 STEP: barrier + 6 groups, ~no DMA", 0ull);
        run_step8k<6, 10>("8 waves split-cb, cout 64 STEP + 10 DMA/wave (80 KiB/step)", ~0ull);
        run_step8<1, 1>("8 waves, cout 32 STEP: barrier + 6 groups, ~no DMA", 0ull);
        run_step8<1, 8>("8 waves, cout 32 STEP + 8 DMA/wave (64 KiB/step)", ~0ull);
        run_step8<2, 1>("8 waves, cout 64 STEP: barrier + 6 groups, ~no DMA", 0ull);
        run_step8<2, 10>("8 waves, cout 64 STEP + 10 DMA/wave (80 KiB/step)", ~0ull);
        run_step<9, 12, 4, 1>("cout 32 STEP: barrier + 6 groups, ~no DMA", 0ull);
        run_step<9, 12, 4, 16>("cout 32 STEP + 16 DMA slots masked off", 0ull);
        run_step<9, 12, 4, 16>("cout 32 STEP + 16 DMA (64 KiB/step)", ~0ull);
        run_step<12, 24, 8, 1>("cout 64 STEP: barrier + 6 groups, ~no DMA", 0ull);
        run_step<12, 24, 8, 20>("cout 64 STEP + 20 DMA (80 KiB/step)", ~0ull);
    }
    g_rnd = 0;
    if (argc > 1) return 0;
    run_group<9, 12, 4>("cout 32 group: 9 reads + 12 MFMA, 4 acc");
    run_group<12, 24, 8>("cout 64 group: 12 reads + 24 MFMA, 8 acc");
    run_group<7, 6, 2>("variant 2 cout 32: 7 reads + 6 MFMA");
    run<0, 0>("linear lane*16");
    run<1, 0>("pixel records, swizzled");
    run<2, 0>("pixel records, unswizzled");
    run<0, 1>("linear + MFMA");
    run<1, 1>("pixel records swizzled + MFMA");
    return 0;
}
