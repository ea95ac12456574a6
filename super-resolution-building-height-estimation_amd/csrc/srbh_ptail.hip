// srbh_ptail.hip -- persistent form of the 64 -> 64 channel 3x3 convs at up-sampled resolution
// (conv_up1 / conv_up2 behind the nearest-x2 read, conv_hr, conv_last's producer; reference SR/rrdbnet_arch.py:
// 234-239): 1 024 - 4 096 output tiles per launch, i.e. 4 - 16 per CU.
//
// The per-launch kernel (srbh_conv3x3_kernel.h) runs one tile per workgroup and one workgroup per CU (160 KiB LDS), so
// every tile pays a cold staging latency and its epilogue stores with idle matrix cores.  Here a workgroup walks a
// contiguous range of tiles:
//   * the two 36 KiB weight chunks are loaded ONCE per workgroup and stay resident (K = 64 input channels only);
//   * both input chunks of a tile are resident too, so the 288 MFMAs per wave of a tile run without any barrier or DMA;
//   * the epilogue stores straight from the MFMA D layout (v_permlane32_swap -> 16 B per lane, no LDS), which leaves
//     LDS free: the input tiles of the NEXT tile are DMA'd while the epilogue of the current one drains.
// Same arithmetic, same accumulation order as conv3x3_f16_kernel<2, UPS> (bit-identical results).
// (Round 6: the three phases of a tile run one after the other here -- ablation builds, profiles/r06ag_ptail_ablation.txt: of the 0.34 ms the
//  three tail convs take per 32 tiles the MFMAs are 0.11, the epilogue (800 VALU instructions + 48 KB of stores per wave) 0.12, the exposed part
//  of the input DMA 0.045.  A fully pipelined form was built: two accumulator sets, the epilogue of tile t - 1 interleaved unit by unit with the
//  MFMA groups of tile t's second chunk in one basic block, the input chunks requested half a tile ahead, counted waits; bit-identical, 45 tests.
//  It is 0.4 % faster on forward_feature and 0.6 % slower in the tiled prediction (profiles/r06ah_ab_ptail_pipe.txt): with the phases
//  overlapped the launch draws more power at once and the package clocks down -- the tail convs are energy-bound like the trunk,
//  not latency-bound.  Not kept.)
#include <stdlib.h>
#include "srbh_conv3x3_kernel.h"

namespace {
using namespace srbh;
using namespace srbh_k;

struct TParams {
    const char* in;             // first input plane (ACT16)
    long in_img_b;
    int in_plane_b, in_row_b;
    const char* w;              // WPACK16, 2 chunks x 36 KiB
    const float* bias;
    int H, W;                   // OUTPUT geometry
    int tiles_x, tiles_per_img, ntiles, tiles_per_wg;
    int lrelu;
    char* out16;                // ACT16 output (2 planes) or nullptr
    long out16_img_b;
    int out16_plane_b, out16_row_b;
    int out16_pix_b, out16_border;   // ACT16: 64-byte pixel records behind a 1-pixel border; NHWC16 (srbh_conv3x3_args::out16_nhwc): dense
                                     // fp16 [B][H][W][C] records of out16_pix_b bytes, no border, "plane" = 64 bytes (the next 32 channels)
    float* out32;               // fp32 NHWC (64 channels) output or nullptr
};

constexpr int W_RES_B = 2 * 36 * 1024;   // resident weights of both chunks

template <int UPS>
__global__ __launch_bounds__(256, 1) void ptail_kernel(const TParams p) {
    using G = TileGeo<UPS>;
    constexpr int IN_EX = G::UNITS * 16;   // exact input tile bytes (tail lanes of the last DMA instruction are masked)
    constexpr int CB = 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [weights 72 KiB][input chunk 0][input chunk 1]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int wr = wave >> 1, wc = wave & 1;

    int goff[G::NJ];
#pragma unroll
    for (int j = 0; j < G::NJ; ++j) {
        const int u0 = j * 256 + tid;
        const int u = u0 < G::UNITS ? u0 : 0;
        const int trow = u / (G::COLS * 4);
        const int rem = u - trow * (G::COLS * 4);
        const int pc = rem >> 2, ps = rem & 3;
        goff[j] = trow * p.in_row_b + pc * PIX_B + ((ps ^ ((pc >> 2) & 3)) << 4);
    }
    const unsigned long long tail_mask = __builtin_amdgcn_ballot_w64((G::NJ - 1) * 256 + tid < G::UNITS);
    int aoff[3][2];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
        const int pc = UPS ? (((wc * 32 + l31 + dx - 1) >> 1) + 1) : (wc * 32 + l31 + dx);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            aoff[dx][ks] = wr * (UPS ? 2 : 4) * G::ROW_B + pc * PIX_B + (((ks * 2 + hi) ^ ((pc >> 2) & 3)) << 4);
    }

    // 16 B per lane LDS-DMA under an explicit EXEC mask (see srbh_ptrunk.hip)
    auto dma16 = [&](const char* gaddr, const unsigned lds_off_v, const unsigned long long mask) {
        unsigned long long sv;
        const unsigned lds_off = __builtin_amdgcn_readfirstlane(lds_off_v);
        asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %1\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %3, off\n\ts_mov_b64 exec, %0"
                     : "=&s"(sv) : "s"(mask), "s"(lds_off), "v"(gaddr) : "memory", "m0");
    };
    auto tile_origin = [&](int t, int& img, int& Y0, int& X0) {
        img = t / p.tiles_per_img;
        const int trem = t - img * p.tiles_per_img;
        const int ty = trem / p.tiles_x;
        Y0 = ty * TILE_H;
        X0 = (trem - ty * p.tiles_x) * TILE_W;
    };
    auto stage_inputs = [&](int t) {
        int img, Y0, X0;
        tile_origin(t, img, Y0, X0);
        const char* src0 = p.in + (long)img * p.in_img_b + (long)(UPS ? (Y0 >> 1) : Y0) * p.in_row_b + (UPS ? (X0 >> 1) : X0) * PIX_B;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int j = 0; j < G::NJ; ++j)
                dma16(src0 + (long)c * p.in_plane_b + goff[j], W_RES_B + c * IN_EX + (j * 256 + wave * 64) * 16,
                      j < G::NJ - 1 ? ~0ull : tail_mask);
    };

    // epilogue stores per wave and tile when every tile is full (no store predicated off): see the wait at the top of the tile loop
    const int nst = ((p.H & (TILE_H - 1)) == 0 && (p.W & (TILE_W - 1)) == 0) ? (p.out32 ? 32 : 0) + (p.out16 ? 16 : 0) : 0;
    const int t0 = blockIdx.x * p.tiles_per_wg;
    const int t1 = (t0 + p.tiles_per_wg < p.ntiles) ? t0 + p.tiles_per_wg : p.ntiles;
    if (t0 >= t1) return;
    // resident weights: 72 fragments of 1 KiB, 18 per wave
#pragma unroll
    for (int k = 0; k < 18; ++k) dma16(p.w + (wave + 4 * k) * 1024 + lane * 16, (wave + 4 * k) * 1024, ~0ull);
    stage_inputs(t0);
    floatx4 bias4[CB][4];
#pragma unroll
    for (int mb = 0; mb < CB; ++mb)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            bias4[mb][g] = p.bias ? *(const floatx4*)(p.bias + mb * 32 + g * 8 + hi * 4) : floatx4{0.f, 0.f, 0.f, 0.f};

    for (int t = t0; t < t1; ++t) {
        int img, Y0, X0;
        tile_origin(t, img, Y0, X0);
        // this tile's inputs (and, first time, the weights) landed ...  From the second tile on the only operations YOUNGER than that DMA are the
        // previous tile's epilogue stores (issued behind stage_inputs): full tiles issue a fixed number of them per wave -- 32 (fp32 output) and /
        // or 16 (16-bit output) -- so the wait leaves exactly those in flight instead of waiting for their acknowledgement as well
        if (t == t0 || nst == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (nst == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else if (nst == 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
        __syncthreads();                                    // ... on every wave
        floatx16 acc[CB][4];
#pragma unroll
        for (int mb = 0; mb < CB; ++mb)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mb][i][r] = 0.f;
        constexpr int NREAD = G::NP + 3 * CB, NMFMA = 12 * CB;
        half8 P[2][G::NP];
        half8 A[2][3][CB];
        auto load_group = [&](int q, int set) {   // q = chunk * 6 + (ks * 3 + dx)
            const int c = q / 6, g = q - c * 6;
            const int ks = g / 3, dx = g - ks * 3;
            const char* sbi = smem + W_RES_B + c * IN_EX;
            const char* sbw = smem + c * (36 * 1024) + lane * 16;
#pragma unroll
            for (int r = 0; r < G::NP; ++r) P[set][r] = *(const half8*)(sbi + aoff[dx][ks] + r * G::ROW_B);
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int mb = 0; mb < CB; ++mb)
                    A[set][dy][mb] = *(const half8*)(sbw + ((((dy * 3 + dx) * 2 + ks) * CB + mb) << 10));
        };
        load_group(0, 0);
#pragma unroll
        for (int q = 0; q < 12; ++q) {
            if (q + 1 < 12) load_group(q + 1, (q + 1) & 1);
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int pr = UPS ? (((i + dy - 1) >> 1) + 1) : (i + dy);
#pragma unroll
                    for (int mb = 0; mb < CB; ++mb)
                        acc[mb][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[q & 1][dy][mb], P[q & 1][pr], acc[mb][i], 0, 0, 0);
                }
            if (q == 0) __builtin_amdgcn_sched_group_barrier(0x100, NREAD, 0);
            if (q + 1 < 12) {
#pragma unroll
                for (int k = 0; k < NREAD; ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, NMFMA - NREAD, 0);
            } else {
                __builtin_amdgcn_sched_group_barrier(0x008, NMFMA, 0);
            }
        }
        __syncthreads();                       // every wave is done reading the input tiles
        if (t + 1 < t1) stage_inputs(t + 1);   // ... so the next tile's inputs fly under this tile's epilogue

        // ---- epilogue, straight from the MFMA D layout: lane (l31, hi) holds for row i and channel group g the 4
        // consecutive channels 8g + 4hi + (0..3) of pixel l31
        const int X = X0 + wc * 32 + l31;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int Y = Y0 + wr * 4 + i;
            const bool valid = (Y < p.H) && (X < p.W);
            const long pix = ((long)img * p.H + Y) * p.W + X;
#pragma unroll
            for (int mb = 0; mb < CB; ++mb) {
                unsigned hp[4][2];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    floatx4 v;
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = acc[mb][i][g * 4 + q];
                    v += bias4[mb][g];
                    if (p.lrelu) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) v[q] = v[q] >= 0.f ? v[q] : v[q] * 0.2f;
                    }
                    if (p.out32 && valid) *(floatx4*)(p.out32 + pix * 64 + mb * 32 + g * 8 + hi * 4) = v;
                    half4 h4;
#pragma unroll
                    for (int q = 0; q < 4; ++q) h4[q] = (_Float16)v[q];
                    const uint2 u = __builtin_bit_cast(uint2, h4);
                    hp[g][0] = u.x;
                    hp[g][1] = u.y;
                }
                if (p.out16) {
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        auto s0 = __builtin_amdgcn_permlane32_swap(hp[2 * m][0], hp[2 * m + 1][0], false, false);
                        auto s1 = __builtin_amdgcn_permlane32_swap(hp[2 * m][1], hp[2 * m + 1][1], false, false);
                        typedef unsigned uintx4 __attribute__((ext_vector_type(4)));
                        const uintx4 raw = {s0[0], s1[0], s0[1], s1[1]};
                        if (valid)
                            *(uintx4*)(p.out16 + (long)img * p.out16_img_b + (long)mb * p.out16_plane_b + (long)(Y + p.out16_border) * p.out16_row_b +
                                       (X + p.out16_border) * p.out16_pix_b + m * 32 + hi * 16) = raw;
                    }
                }
            }
        }
    }
}

thread_local int g_ptail_wgs_cap = 0;      // srbh_ptail_wgs_cap: workgroups per launch (0: one per CU)

template <int UPS>
int launch(const TParams& p0, hipStream_t stream) {
    constexpr int LDS_B = W_RES_B + 2 * TileGeo<UPS>::UNITS * 16;
    static_assert(LDS_B <= 163840, "resident weights + both input chunks must fit the 160 KiB LDS");
    SRBH_ONCE_PER_DEVICE(SRBH_HIP(hipFuncSetAttribute((const void*)ptail_kernel<UPS>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_B)));
    static int ncu_of[64] = {0};   // CU count per device (queried once each)
    int dev = 0;
    SRBH_HIP(hipGetDevice(&dev));
    if (!ncu_of[dev & 63]) SRBH_HIP(hipDeviceGetAttribute(&ncu_of[dev & 63], hipDeviceAttributeMultiprocessorCount, dev));
    int ncu = ncu_of[dev & 63];
    // SRBH_PTAIL_WGS (harness knob, like SRBH_PT_IMAGES): fewer workgroups than CUs -- this kernel holds a whole CU's LDS per workgroup for
    // its entire walk, so a full grid lets no kernel of another stream in while it runs
    if (g_ptail_wgs_cap > 0 && g_ptail_wgs_cap < ncu) ncu = g_ptail_wgs_cap;
    if (const char* we = getenv("SRBH_PTAIL_WGS")) {          // (developer A/B aid: overrides the caller's cap)
        const int cap = atoi(we);
        if (cap > 0 && cap < ncu_of[dev & 63]) ncu = cap;
    }
    TParams p = p0;
    const int nwg = p.ntiles < ncu ? p.ntiles : ncu;
    p.tiles_per_wg = (p.ntiles + nwg - 1) / nwg;
    const int grid = (p.ntiles + p.tiles_per_wg - 1) / p.tiles_per_wg;
    hipLaunchKernelGGL(ptail_kernel<UPS>, dim3(grid), dim3(256), LDS_B, stream, p);
    SRBH_HIP(hipGetLastError());
    return SRBH_OK;
}

}  // namespace

/* Workgroups per launch of the persistent tail convs (conv_up1 / conv_up2 / conv_hr of srbh_rrdbnet_forward) on the calling host thread: 0 = one
 * per CU (default), n = at most n.  A workgroup of this kernel holds its CU's whole LDS for the entire walk, so a full grid lets no kernel of
 * another stream in: harness.TrainStep caps it for the feature prefetch it runs beside the training step.  Returns the previous value. */
extern "C" int srbh_ptail_wgs_cap(int cap) {
    const int prev = g_ptail_wgs_cap;
    g_ptail_wgs_cap = cap > 0 ? cap : 0;
    return prev;
}

namespace srbh {

// *used = 1 when the persistent form ran (64 -> 64 channels, no residual / skip epilogue), 0 when the caller must launch
// the per-tile kernel
int ptail_run(const srbh_conv3x3_args* a, hipStream_t stream, int* used) {
    *used = 0;
    const char* e = getenv("SRBH_PTAIL");
    if (e && e[0] == '0') return SRBH_OK;
    if (a->in_chunks != 2 || a->cout != 64 || a->res1 || a->res2 || a->skip) return SRBH_OK;
    if (a->out32 && a->out32_c != 64) return SRBH_OK;
    const int tiles_x = (a->W + TILE_W - 1) / TILE_W, tiles_y = (a->H + TILE_H - 1) / TILE_H;
    if (a->out16_nhwc && (!a->out16 || a->out32)) return SRBH_OK;
    if ((long)tiles_x * tiles_y * a->B < 512 && !a->out16_nhwc) return SRBH_OK;   // too few tiles per CU to amortise the resident weights (the NHWC16 store exists only here)
    const int inH = a->upsample2x ? a->H / 2 : a->H, inW = a->upsample2x ? a->W / 2 : a->W;
    const Act16Geo gi = act16_geo(a->B, a->in_chunks_total, inH, inW);
    TParams p{};
    p.in = (const char*)a->in + (long)a->in_chunk0 * gi.plane_b;
    p.in_img_b = gi.img_b;
    p.in_plane_b = gi.plane_b;
    p.in_row_b = gi.row_b;
    p.w = (const char*)a->w;
    p.bias = a->bias;
    p.H = a->H;
    p.W = a->W;
    p.tiles_x = tiles_x;
    p.tiles_per_img = tiles_x * tiles_y;
    p.ntiles = p.tiles_per_img * a->B;
    p.lrelu = a->lrelu;
    if (a->out16 && a->out16_nhwc) {
        const int C = a->out16_chunks_total * 32;
        p.out16 = (char*)a->out16 + (long)a->out16_chunk0 * 64;
        p.out16_pix_b = C * 2;
        p.out16_row_b = a->W * C * 2;
        p.out16_img_b = (long)a->H * a->W * C * 2;
        p.out16_plane_b = 64;
        p.out16_border = 0;
    } else if (a->out16) {
        const Act16Geo go = act16_geo(a->B, a->out16_chunks_total, a->H, a->W);
        p.out16_pix_b = PIX_B;
        p.out16_border = 1;
        p.out16 = (char*)a->out16 + (long)a->out16_chunk0 * go.plane_b;
        p.out16_img_b = go.img_b;
        p.out16_plane_b = go.plane_b;
        p.out16_row_b = go.row_b;
    }
    p.out32 = a->out32;
    const int rc = a->upsample2x ? launch<1>(p, stream) : launch<0>(p, stream);
    if (rc == SRBH_OK) *used = 1;
    return rc;
}

}  // namespace srbh
