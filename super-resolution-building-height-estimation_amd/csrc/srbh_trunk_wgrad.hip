// srbh_trunk_wgrad.hip -- weight and bias gradients of ALL dense blocks of the RRDB trunk in one launch (the gradient of the five convs
// of reference SR/rrdbnet_arch.py:136-167 w.r.t. their weights, 69 times, inside SR/rrdbnet_arch.py:538-592's l_g_total.backward()).
//
// What is given (the persistent forward / backward keep both, one buffer per RDB): D = the RDB's saved dense buffer, six fp16 chunk planes of 32
// channels [x | x | x1 | x2 | x3 | x4], and G = its gradient buffer, six bf16 planes [g5 | g5 | g4 | g3 | g2 | g1].  conv_k's weight gradient is
// dW[co][ci][ky][kx] = sum over pixels of g_k[y][x][co] * D[y + ky - 1][x + kx - 1][ci]: per pair (G plane, D plane) one 32 x 32 x 9 block, 26 pairs per
// RDB (conv5: 2 x 6, conv4: 5, conv3: 4, conv2: 3, conv1: 2).  The general-purpose kernel this replaces (hwgrad_b16_kernel, written for the head's 16
// channel tensors) gave every workgroup a 16 x 16 block: 345 launches per step, the same D tile staged (fp16 -> bf16, transposed) by cout/16 workgroups
// and the same G tile by cin/16 of them: 9.2 ms per generator step at batch 8, 25 ms at batch 24 -- 60-80 % of the step.
//
// Here a workgroup owns ONE pair and a contiguous range of 4-row tiles.  Per tile the two 32-channel tiles are staged channel-major in LDS (the
// matrix core wants 8 consecutive PIXELS of one channel per lane: K is the pixel index), x rounded fp16 -> bf16 (RNE) on the way as before, and each
// of the four matrix-core waves runs one row: 4 K-steps x 9 taps of v_mfma_f32_32x32x16_bf16 (A = g^T: 32 cout x 16 pixels, B = x shifted by the tap:
// 16 pixels x 32 cin; the three horizontal taps of a row come from one 16-byte read plus its two neighbour dwords and five v_alignbit).  Four more
// waves do the staging one tile ahead into the other LDS stage (see the kernel), the nine 32 x 32 accumulators stay in registers over the whole
// range, the four waves' sums meet in LDS once, and the partial block goes to the workspace; a second kernel adds the splits in a fixed order and
// scatters into the OIHW gradients.  The bias gradient (sum of g over pixels) rides along as a tenth accumulator against an all-ones B in the
// workgroups of D plane 0.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "srbh.h"
#include "srbh_internal.h"

using namespace srbh;

#ifndef TW_ABL
#define TW_ABL 0      // developer aid (tools/time_trunk_wgrad.py): 1 = no MFMA phase, 2 = no LDS stores, 4 = no global loads (timing only: wrong results)
#endif

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned uintx4 __attribute__((ext_vector_type(4)));
typedef unsigned uintx2 __attribute__((ext_vector_type(2)));
typedef _Float16 half8v __attribute__((ext_vector_type(8)));

constexpr int TW_W = 64, TW_H = 4;              // tile: 4 rows x 64 pixels (the image width); images are H % 8 == 0 tall
constexpr int RS = 80;                          // staged x row: image column c at position c + 8 (16-byte aligned operand reads), c = -4 .. 67
constexpr int XROWS = TW_H + 2;
constexpr int CSX = XROWS * RS * 2 + 16;        // bytes per staged x channel (244 dwords = 52 mod 64: 16 lanes x 4 dwords hit 64 banks)
constexpr int CSD = TW_H * TW_W * 2 + 16;       // bytes per staged g channel (132 dwords = 4 mod 64)
constexpr int X_B = 32 * CSX, D_B = 32 * CSD;
constexpr int STAGE_B = X_B + D_B;              // 48 128 B
constexpr int TWG_LDS_B = 2 * STAGE_B;          // two stages: 96 256 B, one workgroup of 8 waves per CU
constexpr int NPAIR = 26;
constexpr int XUNITS = XROWS * 16 * 4, DUNITS = TW_H * 16 * 4;      // 384 / 256 staging units of (4 pixels x 8 channels); image columns -1 and 64 are the planes' zero borders: never staged
static_assert(TWG_LDS_B >= 2 * 9216 * 4 + 512, "the cross-wave reduce uses the staging area (two waves' blocks at a time)");
static_assert(DUNITS == 256 && XUNITS <= 512, "one g unit and up to two x units per producer thread");

struct TWParams {
    const char* dense;        // forward buffers: RDB i at dense + i * dense_stride
    long dense_stride;
    const char* G;            // gradient buffers: the k-th RDB from the end at G + k * g_stride
    long g_stride;
    long img_b;
    int plane_b, row_b;
    int n_rdb, H, ntiles, tiles_per_img, tiles_per_split, nsplit, per_xcd;
    float* ws;                // [n_rdb][nsplit][26][9216]
    float* wsb;               // [n_rdb][nsplit][6][32]
};

// pair -> (G plane, D plane): conv5 = G planes 0, 1 x D planes 0..5; conv4 = G plane 2 x D 0..4; conv3: 3 x 0..3; conv2: 4 x 0..2; conv1: 5 x 0..1
__device__ __forceinline__ void pair_planes(int p, int& gp, int& dp) {
    if (p < 12) { gp = p / 6; dp = p - gp * 6; }
    else if (p < 17) { gp = 2; dp = p - 12; }
    else if (p < 21) { gp = 3; dp = p - 17; }
    else if (p < 24) { gp = 4; dp = p - 21; }
    else { gp = 5; dp = p - 24; }
}

// 512 threads: waves 0..3 are the matrix-core waves (one tile row each: 4 K-steps x 9 taps per tile, the 32 x 32 x 9 accumulators in registers),
// waves 4..7 the staging waves (global loads one tile ahead in registers, fp16 -> bf16 + 4-pixel transposes, ds_write into the OTHER LDS stage) --
// one per SIMD of each kind, so a SIMD's staging VALU work runs in the shadow of its MFMAs instead of in front of them (as ONE set of waves doing
// both in turn the kernel spent 4.4 us per 8-row tile against 1.1 us of matrix-core time).  One barrier per tile.  Where it stands (batch 24,
// tools/time_trunk_wgrad.py with -DTW_ABL builds): 4.5 ms per call; matrix-core waves alone 2.3 ms, staging waves alone 2.6 ms, nothing 0.47 ms
// (profiles/r05bv) -- the two kinds share each SIMD's VALU issue (transposes, conversions, operand shifts and swaps are ~350 VALU instructions per
// tile and SIMD beside 40 MFMAs), so the phases overlap only in part.  Moving the fp16 -> bf16 rounding of x from the staging to the matrix-core
// waves changed nothing (4.48 -> 4.55 ms; then 2.8 / 2.5 ms alone: profiles/r05ca): built, not kept.
// (Round 6: the staging stores are 2-way bank-conflicted -- a 16-lane pass holds channel octets 0 and 2 of four pixel groups, 8 CSX = 32 banks
//  apart twice; walking the pixel group fastest over the lanes makes them conflict-free and the global loads of a lane quad 256 B apart instead of
//  adjacent: 4.42-4.56 -> 4.61-4.67 ms at batch 24, 1.685 -> 1.75 at batch 8 (profiles/r06an_trunk_wgrad_remap.txt): not kept.)
__global__ __launch_bounds__(512, 1) void trunk_wgrad_kernel(const TWParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const bool producer = wave >= 4;
    const int ptid = tid & 255;
    // XCD-aware order: workgroup id b runs on XCD b % 8, which owns the contiguous range [x * per_xcd, (x + 1) * per_xcd) of (RDB, tile range, pair)
    // triples, pair fastest: the 26 workgroups that read the same twelve planes' tiles sit on one XCD at the same time (worth 4 % at batch 8)
    const int logical = (int)(blockIdx.x & 7) * p.per_xcd + (int)(blockIdx.x >> 3);
    if (logical >= NPAIR * p.nsplit * p.n_rdb) return;
    const int pair = logical % NPAIR, split = (logical / NPAIR) % p.nsplit, k = logical / (NPAIR * p.nsplit);
    int gp, dp;
    pair_planes(pair, gp, dp);
    const char* xpl = p.dense + (long)(p.n_rdb - 1 - k) * p.dense_stride + (long)dp * p.plane_b;
    const char* gpl = p.G + (long)k * p.g_stride + (long)gp * p.plane_b;
    const int t0 = split * p.tiles_per_split, t1 = min(t0 + p.tiles_per_split, p.ntiles);

    const bool with_bias = dp == 0;                 // (uniform)
    // The two kinds of waves run SEPARATE loops with the same number of barriers (an s_barrier counts waves, not program counters): the
    // accumulators are live only in the matrix-core branch and the staging registers only in the other, so the 256 registers a wave gets with
    // eight waves per CU hold 160 of the one or 96 of the other -- as one loop with a role test inside, both were live everywhere (274 spills).
    if (producer) {
        // ---- staging waves: x = 6 rows x 16 four-pixel groups x 4 channel octets = 384 units (two per thread, the second half idle), g = 4 x 16 x 4 = 256.
        // TWO register sets: a tile's loads are issued two iterations before its LDS stores -- one iteration, ~0.6 us of matrix-core time, did not
        // cover their latency (1.6 of the kernel's 5.5 ms at batch 24 were these waves waiting: tools/time_trunk_wgrad.py + -DTW_ABL)
        uintx4 xrA[2][4], drA[4], xrB[2][4], drB[4];
        auto load_tile = [&](const int t, uintx4 (&xr)[2][4], uintx4 (&dr)[4]) {
            const int img = t / p.tiles_per_img, Y0 = (t - img * p.tiles_per_img) * TW_H;
            const char* xb = xpl + (long)img * p.img_b + (long)Y0 * p.row_b;          // padded row Y0 = image row Y0 - 1
            const char* gb = gpl + (long)img * p.img_b + (long)(Y0 + 1) * p.row_b;
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int u = ptid + it * 256;
                const int c8 = u & 3, qq = u >> 2, r = qq >> 4, q = qq & 15;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    uintx4 v = {0u, 0u, 0u, 0u};
                    if (u < XUNITS) v = *(const uintx4*)(xb + (long)r * p.row_b + (4 * q + i + 1) * 64 + c8 * 16);      // image column 4 q + i
                    xr[it][i] = v;
                }
            }
            {
                const int c8 = ptid & 3, qq = ptid >> 2, r = qq >> 4, q = qq & 15;
#pragma unroll
                for (int i = 0; i < 4; ++i) dr[i] = *(const uintx4*)(gb + (long)r * p.row_b + (4 * q + i + 1) * 64 + c8 * 16);
            }
        };
        auto store_tile = [&](char* stage, const uintx4 (&xr)[2][4], const uintx4 (&dr)[4]) {
            char* s_x = stage;
            char* s_d = stage + X_B;
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int u = ptid + it * 256;
                if (u < XUNITS) {
                    const int c8 = u & 3, qq = u >> 2, r = qq >> 4, q = qq & 15;
                    half8v h[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) h[i] = __builtin_bit_cast(half8v, xr[it][i]);
                    char* o = s_x + (c8 * 8) * CSX + (r * RS + 4 * q + 8) * 2;
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        *(uintx2*)__builtin_assume_aligned(o + j * CSX, 8) = uintx2{bf16x2_rne((float)h[0][j], (float)h[1][j]), bf16x2_rne((float)h[2][j], (float)h[3][j])};
                }
            }
            {
                const int c8 = ptid & 3, qq = ptid >> 2, r = qq >> 4, q = qq & 15;
                char* o = s_d + (c8 * 8) * CSD + (r * TW_W + 4 * q) * 2;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const unsigned sel = (j & 1) ? 0x07060302u : 0x05040100u;
                    const unsigned lo = __builtin_amdgcn_perm(dr[1][j >> 1], dr[0][j >> 1], sel);
                    const unsigned hi2 = __builtin_amdgcn_perm(dr[3][j >> 1], dr[2][j >> 1], sel);
                    *(uintx2*)__builtin_assume_aligned(o + j * CSD, 8) = uintx2{lo, hi2};
                }
            }
        };
        if (t0 < t1) {
            load_tile(t0, xrA, drA);
            store_tile(smem, xrA, drA);
            if (t0 + 1 < t1) load_tile(t0 + 1, xrB, drB);
            if (t0 + 2 < t1) load_tile(t0 + 2, xrA, drA);
        }
        __syncthreads();
        // one tile per barrier; unrolled by two so that the register sets are named statically (tile t0 + r sits in set A for even r, B for odd r)
        auto step = [&](const int t, uintx4 (&xr)[2][4], uintx4 (&dr)[4]) {      // xr / dr: the set holding tile t + 1
            char* other = smem + (((t - t0) & 1) ^ 1) * STAGE_B;
            if (t + 1 < t1) {
                if (!(TW_ABL & 2)) store_tile(other, xr, dr);                        // tile t + 1 (loaded two iterations ago)
                if (t + 3 < t1 && !(TW_ABL & 4)) load_tile(t + 3, xr, dr);           // in flight for two iterations
            }
            __syncthreads();                                   // the stage the other waves read: read out; other: complete
        };
        for (int t = t0; t < t1; t += 2) {
            step(t, xrB, drB);
            if (t + 1 < t1) step(t + 1, xrA, drA);
        }
        __syncthreads();       // the three barriers of the accumulators' meeting in LDS (below), and the bias gradient's
        __syncthreads();
        __syncthreads();
        if (with_bias) __syncthreads();
        return;
    }

    // ---- matrix-core waves: wave w = row w of the tile
    floatx16 acc[9], accb;
#pragma unroll
    for (int tp = 0; tp < 9; ++tp)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[tp][r] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) accb[r] = 0.f;
    const unsigned one2 = 0x3f803f80u;              // two bf16 ones
    const uintx4 ones = {one2, one2, one2, one2};
    auto mfma_tile = [&](const char* stage) {
        const char* s_x = stage;
        const char* s_d = stage + X_B;
        bf16x8 a[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            a[g] = __builtin_bit_cast(bf16x8, *(const uintx4*)__builtin_assume_aligned(s_d + l31 * CSD + (wave * TW_W + g * 16 + hi * 8) * 2, 16));
            if (with_bias) accb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[g], __builtin_bit_cast(bf16x8, ones), accb, 0, 0, 0);
        }
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            // the row's 64 pixels of this lane's channel: four 16-byte reads.  The pixel in front of / behind a lane's eight sits in the PARTNER
            // lane of the other half-wave (same channel): v_permlane32_swap + a select, no LDS access -- as `ds_read_b32` of the neighbour dwords
            // (a per-channel stride of 244 dwords: 8-way bank conflicts) those 24 reads per tile cost as much LDS time as the 16 wide ones.
            // Image columns -1 and 64 are the planes' zero borders.
            uintx4 c[4];
#pragma unroll
            for (int g = 0; g < 4; ++g)
                c[g] = *(const uintx4*)__builtin_assume_aligned(s_x + l31 * CSX + ((wave + dy) * RS + g * 16 + hi * 8 + 8) * 2, 16);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const auto sp = __builtin_amdgcn_permlane32_swap(g > 0 ? c[g - 1][3] : 0u, c[g][3], false, false);      // {(a.lo, b.lo), (a.hi, b.hi)}
                const auto sn = __builtin_amdgcn_permlane32_swap(c[g][0], g < 3 ? c[g + 1][0] : 0u, false, false);
                const unsigned pv = hi ? sp[0] : sp[1];       // the dword whose HIGH half is the pixel in front of this lane's eight
                const unsigned nx = hi ? sn[0] : sn[1];       // the dword whose LOW half is the pixel behind them
                const uintx4 cur = c[g];
                const unsigned m1 = __builtin_amdgcn_alignbit(cur[1], cur[0], 16), m2 = __builtin_amdgcn_alignbit(cur[2], cur[1], 16),
                               m3 = __builtin_amdgcn_alignbit(cur[3], cur[2], 16);
                const uintx4 b0 = {__builtin_amdgcn_alignbit(cur[0], pv, 16), m1, m2, m3};
                const uintx4 b2 = {m1, m2, m3, __builtin_amdgcn_alignbit(nx, cur[3], 16)};
                acc[dy * 3 + 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[g], __builtin_bit_cast(bf16x8, b0), acc[dy * 3 + 0], 0, 0, 0);
                acc[dy * 3 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[g], __builtin_bit_cast(bf16x8, cur), acc[dy * 3 + 1], 0, 0, 0);
                acc[dy * 3 + 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[g], __builtin_bit_cast(bf16x8, b2), acc[dy * 3 + 2], 0, 0, 0);
            }
        }
    };
    __syncthreads();
    for (int t = t0; t < t1; ++t) {
        if (!(TW_ABL & 1)) mfma_tile(smem + ((t - t0) & 1) * STAGE_B);
        __syncthreads();
    }
    // ---- the four matrix-core waves' blocks meet in LDS: waves 2, 3 -> LDS, waves 0, 1 add; wave 1 -> LDS, wave 0 adds and writes the partial block
    // accumulator element r of lane (l31, hi) is (co = 8 (r >> 2) + 4 hi + (r & 3), ci = l31): stored as [tap][co][ci]
    float* red = (float*)smem;
    auto idx = [&](int tp, int r) { return tp * 1024 + (8 * (r >> 2) + 4 * hi + (r & 3)) * 32 + l31; };
    if (wave >= 2) {
#pragma unroll
        for (int tp = 0; tp < 9; ++tp)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[(wave - 2) * 9216 + idx(tp, r)] = acc[tp][r];
    }
    __syncthreads();
    if (wave < 2) {
#pragma unroll
        for (int tp = 0; tp < 9; ++tp)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tp][r] += red[wave * 9216 + idx(tp, r)];
    }
    __syncthreads();
    if (wave == 1) {
#pragma unroll
        for (int tp = 0; tp < 9; ++tp)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[idx(tp, r)] = acc[tp][r];
    }
    __syncthreads();
    if (wave == 0) {
        float* out = p.ws + (((long)k * p.nsplit + split) * NPAIR + pair) * 9216;
#pragma unroll
        for (int tp = 0; tp < 9; ++tp)
#pragma unroll
            for (int r = 0; r < 16; ++r) out[idx(tp, r)] = acc[tp][r] + red[idx(tp, r)];
    }
    if (with_bias) {       // every column of accb holds the same sums: column 0's lanes (l31 == 0) carry them
        float* rb = (float*)smem + 2 * 9216;                 // (behind what wave 0 may still be reading)
        if (l31 == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) rb[wave * 32 + 8 * (r >> 2) + 4 * hi + (r & 3)] = accb[r];
        }
        __syncthreads();
        if (tid < 32) p.wsb[(((long)k * p.nsplit + split) * 6 + gp) * 32 + tid] = rb[tid] + rb[32 + tid] + rb[64 + tid] + rb[96 + tid];
    }
}

// the splits in a fixed order -> OIHW gradients (dw_all: per RDB 239 616 floats, conv1..conv5) and db_all (per RDB 192 floats in G's channel order)
__global__ __launch_bounds__(256) void trunk_wgrad_reduce_kernel(const float* ws, const float* wsb, int nsplit, int n_rdb, float* dw_all, float* db_all) {
    const int pr = blockIdx.x, k = blockIdx.y, i = n_rdb - 1 - k;
    if (pr == NPAIR) {
        if (threadIdx.x < 192) {
            float s = 0.f;
            for (int sp = 0; sp < nsplit; ++sp) s += wsb[((long)k * nsplit + sp) * 192 + threadIdx.x];
            db_all[(long)i * 192 + threadIdx.x] = s;
        }
        return;
    }
    int gp, dp;
    pair_planes(pr, gp, dp);
    // conv of G plane gp: conv5 (planes 0, 1: output channels gp * 32 ..), conv4 (2), conv3 (3), conv2 (4), conv1 (5)
    const int conv = gp < 2 ? 4 : 5 - gp;                                   // 0-based conv index
    const int cin = 64 + 32 * conv;
    const long dwoff = conv == 0 ? 0 : conv == 1 ? 9L * 2048 : conv == 2 ? 9L * (2048 + 3072) : conv == 3 ? 9L * (2048 + 3072 + 4096) : 9L * (2048 + 3072 + 4096 + 5120);
    const int co0 = gp == 1 ? 32 : 0;
    float* dw = dw_all + (long)i * (9L * 26624) + dwoff;
    for (int e = threadIdx.x; e < 9216; e += 256) {
        float s = 0.f;
        for (int sp = 0; sp < nsplit; ++sp) s += ws[(((long)k * nsplit + sp) * NPAIR + pr) * 9216 + e];
        const int tp = e >> 10, co = (e >> 5) & 31, ci = e & 31;
        dw[((long)(co0 + co) * cin + dp * 32 + ci) * 9 + tp] = s;
    }
}

}  // namespace

// tile ranges per (RDB, pair): 4 (swept at batch 8 and 24, tools/time_trunk_wgrad.py with SRBH_TWG_SPLITS: every workgroup pays a cold first tile, the
// meeting of its four accumulator sets in LDS and 37 KB of partial sums)
static int twg_splits(int ntiles) {
    static const int want = getenv("SRBH_TWG_SPLITS") ? atoi(getenv("SRBH_TWG_SPLITS")) : 4;
    const int s = want > 0 ? want : 4;
    return ntiles < s ? ntiles : s;
}

extern "C" size_t srbh_trunk_wgrad_ws_bytes(int num_block, int B, int H, int W) {
    if (num_block <= 0 || B <= 0 || W != TW_W || H <= 0 || (H % 8) != 0) return 0;
    const int ntiles = B * (H / TW_H);
    const int nsplit = twg_splits(ntiles);
    return (size_t)num_block * 3 * nsplit * ((size_t)NPAIR * 9216 + 192) * sizeof(float);
}

extern "C" int srbh_trunk_wgrad(int num_block, const void* dense_all, size_t dense_stride, const void* G_all, size_t g_stride, int B, int H, int W,
                                float* dw_all, float* db_all, void* ws, void* stream) {
    SRBH_REQUIRE(num_block > 0 && dense_all && G_all && dw_all && db_all && ws && B > 0, "srbh_trunk_wgrad: bad arguments");
    SRBH_REQUIRE(srbh_trunk_wgrad_ws_bytes(num_block, B, H, W) > 0, "srbh_trunk_wgrad: 64-pixel-wide images, H %% 8 == 0 (srbh_trunk_wgrad_ws_bytes)");
    const Act16Geo g = act16_geo(B, 6, H, W);
    TWParams p;
    p.dense = (const char*)dense_all; p.dense_stride = (long)dense_stride;
    p.G = (const char*)G_all; p.g_stride = (long)g_stride;
    p.img_b = g.img_b; p.plane_b = g.plane_b; p.row_b = g.row_b;
    p.n_rdb = num_block * 3; p.H = H;
    p.tiles_per_img = H / TW_H;
    p.ntiles = B * p.tiles_per_img;
    p.nsplit = twg_splits(p.ntiles);
    p.tiles_per_split = (p.ntiles + p.nsplit - 1) / p.nsplit;
    p.ws = (float*)ws;
    p.wsb = (float*)ws + (size_t)p.n_rdb * p.nsplit * NPAIR * 9216;
    hipStream_t st = (hipStream_t)stream;
    SRBH_ONCE_PER_DEVICE(SRBH_HIP(hipFuncSetAttribute((const void*)trunk_wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, TWG_LDS_B)));
    const int total = NPAIR * p.nsplit * p.n_rdb;
    p.per_xcd = (total + 7) / 8;
    hipLaunchKernelGGL(trunk_wgrad_kernel, dim3(p.per_xcd * 8), dim3(512), TWG_LDS_B, st, p);
    SRBH_HIP(hipGetLastError());
    hipLaunchKernelGGL(trunk_wgrad_reduce_kernel, dim3(NPAIR + 1, p.n_rdb), dim3(256), 0, st, p.ws, p.wsb, p.nsplit, p.n_rdb, dw_all, db_all);
    SRBH_HIP(hipGetLastError());
    return SRBH_OK;
}
