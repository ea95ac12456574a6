// mfma_ceiling.hip -- what can v_mfma_f32_32x32x16_f16 deliver on THIS chip under ITS power cap?  (round-3 VERDICT, task 2(i))
//
// The trunk kernel's roofline is quoted against 2.5 PFLOP/s = 256 CUs x 4 SIMDs x 1024 FLOP/clk x 2.4 GHz.  A package that is
// power-limited never runs the matrix cores at 2.4 GHz on real data, so the reachable ceiling is lower; this tool measures it:
// one wave per SIMD (the trunk's occupancy), 256 workgroups, launches of a few ms back to back for seconds, operands held in
// registers (no LDS, no memory in the loop), wall clock by HIP events, shader clock from s_memtime / s_memrealtime inside the kernel.
//   mode mfma      : MFMAs only, 4 independent accumulators (back-to-back issue), operands cycle through 8 A x 8 B fragments
//   mode mfma+lds  : the same MFMAs with the trunk's LDS read mix (0.75 ds_read_b128 per MFMA, conflict-free pattern) feeding them
//   mode mfma+lds+wdma : the same plus the trunk's WEIGHT stream: 18 KiB per 72 MFMAs and workgroup by LDS-DMA (global_load_lds_dwordx4)
//                    from a 33 MB buffer every workgroup walks in step (69 RDBs x 479 KB of fp16 weights: L2 hits after the first
//                    workgroup of an XCD, as in the trunk).  256 workgroups x all weights is what ANY B = 32 launch with one 8 x 64
//                    tile per CU has to move through L2 -> LDS, so this is the ceiling of the tiling, not of the kernel
//   mode duty<p>   : MFMA bursts with idle gaps (s_sleep) so that the matrix core is busy ~p % of the cycles: what the chip gives
//                    back in clock when the kernel idles (the trunk: 59 % busy)
// operand data: zeros | random fp16 in [-0.5, 0.5) | a dump of real trunk activations / weights (tools/dump_trunk_operands.py)
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_ceiling.hip -o tools/mfma_ceiling ; tools/mfma_ceiling [acts.bin weights.bin]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstring>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

struct Out { unsigned long long cyc, real; };

// MODE 0: MFMA only; 1: MFMA + LDS reads (9 per 12 MFMAs); 2: duty-cycled MFMA (sleep after each burst of 32); 3: as 1 + the weight LDS-DMA stream
template <int MODE>
__global__ __launch_bounds__(256, 1) void ceiling_kernel(const half8* __restrict__ wsrc, const half8* __restrict__ asrc, int nfrag, int iters, int sleep_n,
                                                         Out* out, float* sink, const char* wstream = nullptr, long wstream_b = 0) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    half8 A[8], B[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        A[k] = wsrc[((blockIdx.x * 4 + wave) * 8 + k) % nfrag * 64 + lane];
        B[k] = asrc[((blockIdx.x * 4 + wave) * 8 + k) % nfrag * 64 + lane];
    }
    if (MODE == 1 || MODE == 3) {      // LDS filled with the activation data: 36 KiB of fragments per wave
        for (int i = tid; i < 144 * 64; i += 256) ((half8*)smem)[i] = asrc[(blockIdx.x * 144 + i / 64) % nfrag * 64 + (i & 63)];
        __syncthreads();
    }
    floatx16 c[4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int q = 0; q < 16; ++q) c[a][q] = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    if (MODE == 1 || MODE == 3) {
        // the trunk's group: 9 reads (6 pixel-row + 3 weight fragments) feed 12 MFMAs; double-buffered, one read per MFMA shadow
        half8 v[2][9];
        const char* base = smem + wave * 36 * 1024 + lane * 16;
#pragma unroll
        for (int r = 0; r < 9; ++r) v[0][r] = *(const half8*)(base + r * 1024);
        // MODE 3: position in the weight stream (an SGPR pair), advanced by one 18 KiB chunk per 72 MFMAs
        unsigned long long wpos = (unsigned long long)wstream, wend = wpos + (unsigned long long)wstream_b - 18 * 1024;
        const unsigned lds_w = __builtin_amdgcn_readfirstlane(144 * 1024 + wave * 1024);   // weight stage: 16 KiB behind the activation image
        const unsigned vo = lane * 16;
        const int wv = __builtin_amdgcn_readfirstlane(wave);
        constexpr int NG = MODE == 3 ? 6 : 4;                            // groups per iteration (MODE 3: one 72-MFMA step, as the trunk's)
        const int n_it = MODE == 3 ? iters * 4 / 6 : iters;
        for (int it = 0; it < n_it; ++it) {
#pragma unroll
            for (int u = 0; u < NG; ++u) {
#pragma unroll
                for (int m = 0; m < 12; ++m) {
                    c[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v[u & 1][6 + m / 4], v[u & 1][(m & 3) + m / 4], c[m & 3], 0, 0, 0);
                    if (m < 9) v[(u + 1) & 1][m] = *(const half8*)(base + ((u & 3) * 9 + m) * 1024);      // (immediate offsets, as in the trunk)
                    if (MODE == 3 && m == 10 && u < 5) {
                        // 18 KiB per workgroup and step = 4.5 KiB per wave: 4 full statements + 1 that only waves 0, 1 issue, one statement
                        // behind an MFMA of 5 of the 6 groups -- the trunk's own statement incl. its wait states, everything else compile-time
                        const unsigned long long sb = wpos + (u < 4 ? u * 4096 + wv * 1024 : 16384 + wv * 1024);
                        if (u < 4 || wv < 2)
                            asm volatile("s_mov_b32 m0, %0\n\ts_nop 3\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_w + (u & 3) * 4096), "v"(vo), "s"(sb) : "memory", "m0");
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (MODE == 3) {
                wpos = wpos + 18 * 1024 <= wend ? wpos + 18 * 1024 : (unsigned long long)wstream;
                asm volatile("s_waitcnt vmcnt(5)" ::: "memory");         // (at most one step of statements in flight, as behind the trunk's step barrier)
            }
        }
        if (MODE == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int m = 0; m < 48; ++m) c[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[m & 7], B[(m + m / 8) & 7], c[m & 3], 0, 0, 0);
            if (MODE == 2) {
                for (int s = 0; s < sleep_n; ++s) __builtin_amdgcn_s_sleep(8);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (tid == 0) { out[blockIdx.x].cyc = t1 - t0; out[blockIdx.x].real = r1 - r0; }
    float s = 0;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int q = 0; q < 16; ++q) s += c[a][q];
    if (s == 12345.678f) sink[0] = s;
}

// ---- operand-ORDER modes (round 5; round-4 VERDICT task 2(i)): the same back-to-back MFMA stream, registers only, 4 independent
// accumulators, but WHICH operand changes between consecutive MFMAs is the variable.  If the matrix core's input latches / operand
// delivery cost less when an operand repeats, the power-capped clock -- and so the ceiling -- moves with the order.
//   ORD 0: both change every MFMA (= "mfma only" above)      ORD 1/2/3: A (weights) held for 2 / 4 / 8 MFMAs, B changes every MFMA
//   ORD 4: B (pixels) held for 4, A changes every MFMA        ORD 5: snake -- consecutive MFMAs always share ONE operand, alternately A and B
//   ORD 6: the trunk's exact cout-32 group order: A[m / 4] (3 fragments), P[(m & 3) + m / 4] (6 fragments), 12 MFMAs, A held x4
//   ORD 7: the same 12 MFMAs B-stationary: pixel row r = i + dy held over its (up to 3) uses, A changes
// SHAPE 0: v_mfma_f32_32x32x16_f16 (16 acc registers, 32 cycles)   SHAPE 1: v_mfma_f32_16x16x32_f16 (4 acc registers, 16 cycles; same
// FLOPs per operand byte pair... half the FLOPs per instruction, so twice the operand traffic per FLOP)
typedef float floatx4c __attribute__((ext_vector_type(4)));
template <int ORD, int SHAPE>
__global__ __launch_bounds__(256, 1) void order_kernel(const half8* __restrict__ wsrc, const half8* __restrict__ asrc, int nfrag, int iters, Out* out, float* sink) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    half8 A[8], B[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        A[k] = wsrc[((blockIdx.x * 4 + wave) * 8 + k) % nfrag * 64 + lane];
        B[k] = asrc[((blockIdx.x * 4 + wave) * 8 + k) % nfrag * 64 + lane];
    }
    floatx16 c[4];
    floatx4c d[8];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int q = 0; q < 16; ++q) c[a][q] = 0.f;
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int q = 0; q < 4; ++q) d[a][q] = 0.f;
    auto idx = [](const int m, int& ia, int& ib) {
        if (ORD == 0) { ia = m & 7; ib = (m + m / 8) & 7; }
        else if (ORD == 1) { ia = (m / 2) & 7; ib = m & 7; }
        else if (ORD == 2) { ia = (m / 4) & 7; ib = m & 7; }
        else if (ORD == 3) { ia = (m / 8) & 7; ib = (m + m / 8) & 7; }
        else if (ORD == 4) { ia = m & 7; ib = (m / 4) & 7; }
        else if (ORD == 5) { ia = ((m + 1) / 2) & 7; ib = (m / 2) & 7; }
        else if (ORD == 6) { const int g = m % 12; ia = g / 4 + 3 * ((m / 12) & 1); ib = (g & 3) + g / 4; }
        else {   // ORD 7: rows r = 0..5 of a group, (dy, i = r - dy) for dy with 0 <= i <= 3
            constexpr int RR[12] = {0, 1, 1, 2, 2, 2, 3, 3, 3, 4, 4, 5}, DY[12] = {0, 0, 1, 0, 1, 2, 0, 1, 2, 1, 2, 2};
            const int g = m % 12; ib = RR[g]; ia = DY[g] + 3 * ((m / 12) & 1);
        }
    };
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 48; ++m) {
            int ia, ib;
            idx(m, ia, ib);
            if (SHAPE == 0) {
                // accumulator: ORD 6 / 7 use the trunk's (one per output row i); the others rotate 4
                int ac = m & 3;
                if (ORD == 6) ac = (m % 12) & 3;
                if (ORD == 7) { constexpr int RR[12] = {0, 1, 1, 2, 2, 2, 3, 3, 3, 4, 4, 5}, DY[12] = {0, 0, 1, 0, 1, 2, 0, 1, 2, 1, 2, 2}; ac = RR[m % 12] - DY[m % 12]; }
                c[ac] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[ia], B[ib], c[ac], 0, 0, 0);
            } else {
                d[m & 7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[ia], B[ib], d[m & 7], 0, 0, 0);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (tid == 0) { out[blockIdx.x].cyc = t1 - t0; out[blockIdx.x].real = r1 - r0; }
    float s = 0;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int q = 0; q < 16; ++q) s += c[a][q];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int q = 0; q < 4; ++q) s += d[a][q];
    if (s == 12345.678f) sink[0] = s;
}

template <int ORD, int SHAPE>
static void run_order(const char* name, const half8* dw, const half8* da, int nfrag, double seconds) {
    Out* dout; float* sink;
    hipMalloc(&dout, 256 * sizeof(Out)); hipMalloc(&sink, 4);
    const int iters = SHAPE == 0 ? 4000 : 8000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int launches = 0; float ms = 0;
    for (int phase = 0; phase < 2; ++phase) {
        hipEventRecord(e0);
        int n = 0; float acc = 0;
        do {
            for (int k = 0; k < 20; ++k) hipLaunchKernelGGL((order_kernel<ORD, SHAPE>), dim3(256), dim3(256), 0, 0, dw, da, nfrag, iters, dout, sink);
            n += 20;
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&acc, e0, e1);
        } while (acc < seconds * 500.0);
        launches = n; ms = acc;
    }
    Out h[256]; hipMemcpy(h, dout, sizeof(h), hipMemcpyDeviceToHost);
    double cyc = 0, real = 0;
    for (int i = 0; i < 256; ++i) { cyc += (double)h[i].cyc; real += (double)h[i].real; }
    cyc /= 256; real /= 256;
    const double per = SHAPE == 0 ? 32768.0 : 16384.0, cycm = SHAPE == 0 ? 32.0 : 16.0;
    const double mfmas = (double)iters * 48, flop = mfmas * per * 4 * 256;
    const double ms_l = ms / launches, tf = flop / (ms_l * 1e-3) / 1e12, mhz = cyc / real * 100.0;
    printf("%-46s %7.3f ms/launch  %7.1f TF/s = %.3f of 2500   sclk %6.0f MHz   matrix-core busy %5.1f %%\n", name, ms_l, tf, tf / 2500.0, mhz, 100.0 * mfmas * cycm / cyc);
    fflush(stdout);
    hipFree(dout); hipFree(sink);
}

static std::vector<_Float16> load_or_make(const char* path, size_t n, int kind, unsigned seed) {
    std::vector<_Float16> v(n);
    if (path) {
        FILE* f = fopen(path, "rb");
        if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(1); }
        std::vector<_Float16> raw;
        fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET);
        raw.resize(sz / 2);
        if (fread(raw.data(), 2, raw.size(), f) != raw.size()) { fprintf(stderr, "short read %s\n", path); exit(1); }
        fclose(f);
        for (size_t i = 0; i < n; ++i) v[i] = raw[i % raw.size()];
        return v;
    }
    unsigned h = seed;
    for (size_t i = 0; i < n; ++i) {
        h = h * 1664525u + 1013904223u;
        v[i] = kind == 0 ? (_Float16)0.f : (_Float16)(((h >> 8) & 0xffff) / 65536.f - 0.5f);
    }
    return v;
}

template <int MODE>
static void run(const char* name, const half8* dw, const half8* da, int nfrag, int sleep_n, double seconds) {
    Out* dout; float* sink;
    hipMalloc(&dout, 256 * sizeof(Out)); hipMalloc(&sink, 4);
    const int lds = (MODE == 1 || MODE == 3) ? 160 * 1024 : 0;
    static char* wstream = nullptr;
    const long wstream_b = 69L * 479232;                           // 69 RDBs x 9 x 26 624 fp16 weights
    if (MODE == 3 && !wstream) {                                   // the weight stream: the weight operand data repeated
        hipMalloc(&wstream, wstream_b + 65536);
        for (long o = 0; o < wstream_b; o += (long)nfrag * 1024) {
            const long nb = o + (long)nfrag * 1024 <= wstream_b ? (long)nfrag * 1024 : wstream_b - o;
            hipMemcpy(wstream + o, dw, nb, hipMemcpyDeviceToDevice);
        }
    }
    if (lds) hipFuncSetAttribute((const void*)ceiling_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const int iters = MODE == 2 ? 2000 : 4000;                    // 48 MFMAs x 32 cyc x 4000 = 6.1 M cycles ~ 3 ms
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    // warm the power state: run for `seconds`, time the second half
    int launches = 0; float ms = 0;
    for (int phase = 0; phase < 2; ++phase) {
        hipEventRecord(e0);
        int n = 0; float acc = 0;
        do {
            for (int k = 0; k < 20; ++k) hipLaunchKernelGGL((ceiling_kernel<MODE>), dim3(256), dim3(256), lds, 0, dw, da, nfrag, iters, sleep_n, dout, sink, (const char*)wstream, wstream_b);
            n += 20;
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&acc, e0, e1);
        } while (acc < seconds * 500.0);
        launches = n; ms = acc;
    }
    Out h[256]; hipMemcpy(h, dout, sizeof(h), hipMemcpyDeviceToHost);
    double cyc = 0, real = 0;
    for (int i = 0; i < 256; ++i) { cyc += (double)h[i].cyc; real += (double)h[i].real; }
    cyc /= 256; real /= 256;
    const double mfmas = MODE == 3 ? (double)(iters * 4 / 6) * 72 : (double)iters * 48, flop = mfmas * 32768.0 * 4 * 256;      // per launch: 4 waves x 256 workgroups
    const double ms_l = ms / launches, tf = flop / (ms_l * 1e-3) / 1e12, mhz = cyc / real * 100.0;
    printf("%-34s %7.3f ms/launch  %7.1f TF/s = %.3f of 2500   sclk %6.0f MHz   matrix-core busy %5.1f %% of kernel cycles   (kernel %0.3f ms by counters)\n",
           name, ms_l, tf, tf / 2500.0, mhz, 100.0 * mfmas * 32.0 / cyc, real / 100e6 * 1e3);
    fflush(stdout);
    hipFree(dout); hipFree(sink);
}

int main(int argc, char** argv) {
    const int nfrag = 4096;                                       // 4096 fragments x 64 lanes x 8 halfs = 4 MiB per operand set
    const size_t n = (size_t)nfrag * 64 * 8;
    const double seconds = getenv("CEIL_SECONDS") ? atof(getenv("CEIL_SECONDS")) : 6.0;
    for (int kind = 0; kind < 3; ++kind) {
        if (kind == 2 && argc < 3) break;
        const char* label = kind == 0 ? "zeros" : kind == 1 ? "random fp16 [-0.5,0.5)" : "real trunk dump";
        std::vector<_Float16> a = load_or_make(kind == 2 ? argv[1] : nullptr, n, kind, 1u), w = load_or_make(kind == 2 ? argv[2] : nullptr, n, kind, 7u);
        half8 *da, *dw;
        hipMalloc(&da, n * 2); hipMalloc(&dw, n * 2);
        hipMemcpy(da, a.data(), n * 2, hipMemcpyHostToDevice); hipMemcpy(dw, w.data(), n * 2, hipMemcpyHostToDevice);
        printf("---- operand data: %s\n", label);
        run<0>("mfma only", dw, da, nfrag, 0, seconds);
        run<1>("mfma + 0.75 ds_read_b128 / mfma", dw, da, nfrag, 0, seconds);
        run<3>("mfma + lds + weight LDS-DMA stream", dw, da, nfrag, 0, seconds);
        if (kind != 0 && !getenv("CEIL_SKIP_ORDER")) {
            run_order<0, 0>("order: 32x32x16 both operands change", dw, da, nfrag, seconds);
            run_order<1, 0>("order: 32x32x16 A held x2", dw, da, nfrag, seconds);
            run_order<2, 0>("order: 32x32x16 A held x4", dw, da, nfrag, seconds);
            run_order<3, 0>("order: 32x32x16 A held x8", dw, da, nfrag, seconds);
            run_order<4, 0>("order: 32x32x16 B held x4", dw, da, nfrag, seconds);
            run_order<5, 0>("order: 32x32x16 snake (share one operand)", dw, da, nfrag, seconds);
            run_order<6, 0>("order: 32x32x16 trunk cout-32 group (A x4)", dw, da, nfrag, seconds);
            run_order<7, 0>("order: 32x32x16 trunk group, B-stationary", dw, da, nfrag, seconds);
            run_order<0, 1>("order: 16x16x32 both operands change", dw, da, nfrag, seconds);
            run_order<2, 1>("order: 16x16x32 A held x4", dw, da, nfrag, seconds);
            run_order<4, 1>("order: 16x16x32 B held x4", dw, da, nfrag, seconds);
        }
        if (kind != 0 && !getenv("CEIL_SKIP_DUTY")) {
            run<2>("mfma bursts, sleep 1x", dw, da, nfrag, 1, seconds);
            run<2>("mfma bursts, sleep 2x", dw, da, nfrag, 2, seconds);
            run<2>("mfma bursts, sleep 4x", dw, da, nfrag, 4, seconds);
        }
        hipFree(da); hipFree(dw);
    }
    return 0;
}
