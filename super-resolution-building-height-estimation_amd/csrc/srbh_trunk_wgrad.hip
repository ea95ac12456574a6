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
// Here a workgroup owns ONE pair and a contiguous range of 8-row tiles.  Per tile the two 32-channel tiles are staged channel-major in LDS (the
// matrix core wants 8 consecutive PIXELS of one channel per lane: K is the pixel index), x rounded fp16 -> bf16 (RNE) on the way as before, and every
// wave runs its two rows: 8 K-steps x 9 taps of v_mfma_f32_32x32x16_bf16 (A = g^T: 32 cout x 16 pixels, B = x shifted by the tap: 16 pixels x 32 cin;
// the three horizontal taps of a row come from one 16-byte read plus its two neighbour dwords and five v_alignbit).  The next tile's global loads are in
// flight under the MFMAs (registers), the nine 32 x 32 accumulators stay in registers over the whole range, the four waves' sums meet in LDS once, and
// the partial block goes to the workspace; a second kernel adds the splits in a fixed order and scatters into the OIHW gradients.  The bias gradient
// (sum of g over pixels) rides along as a tenth accumulator against an all-ones B in the workgroups of D plane 0.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "srbh.h"
#include "srbh_internal.h"

using namespace srbh;

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned uintx4 __attribute__((ext_vector_type(4)));
typedef unsigned uintx2 __attribute__((ext_vector_type(2)));
typedef _Float16 half8v __attribute__((ext_vector_type(8)));

constexpr int TW_W = 64, TW_H = 8;              // tile: 8 rows x 64 pixels (the image width)
constexpr int RS = 80;                          // staged x row: image column c at position c + 8 (16-byte aligned operand reads), c = -4 .. 67
constexpr int CSX = 10 * RS * 2 + 16;           // bytes per staged x channel (404 dwords = 20 mod 64: 16 lanes x 4 dwords hit 64 banks)
constexpr int CSD = TW_H * TW_W * 2 + 16;       // bytes per staged g channel (260 dwords = 4 mod 64)
constexpr int X_B = 32 * CSX, D_B = 32 * CSD;
constexpr int TWG_LDS_B = X_B + D_B;            // 84 992 B: one workgroup per CU
constexpr int NPAIR = 26;
static_assert(TWG_LDS_B >= 2 * 9216 * 4, "the cross-wave reduce uses the staging area (two waves' blocks at a time)");

struct TWParams {
    const char* dense;        // forward buffers: RDB i at dense + i * dense_stride
    long dense_stride;
    const char* G;            // gradient buffers: the k-th RDB from the end at G + k * g_stride
    long g_stride;
    long img_b;
    int plane_b, row_b;
    int n_rdb, H, ntiles, tiles_per_img, tiles_per_split, nsplit;
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

__global__ __launch_bounds__(256, 1) void trunk_wgrad_kernel(const TWParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* s_x = smem;
    char* s_d = smem + X_B;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    int gp, dp;
    pair_planes(blockIdx.x, gp, dp);
    const int split = blockIdx.y, k = blockIdx.z;
    const char* xpl = p.dense + (long)(p.n_rdb - 1 - k) * p.dense_stride + (long)dp * p.plane_b;
    const char* gpl = p.G + (long)k * p.g_stride + (long)gp * p.plane_b;
    const int t0 = split * p.tiles_per_split, t1 = min(t0 + p.tiles_per_split, p.ntiles);

    floatx16 acc[9], accb;
#pragma unroll
    for (int tp = 0; tp < 9; ++tp)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[tp][r] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) accb[r] = 0.f;
    const bool with_bias = dp == 0;                 // (uniform)
    const unsigned one2 = 0x3f803f80u;              // two bf16 ones
    const uintx4 ones = {one2, one2, one2, one2};

    // staging registers: x = 10 rows x 18 four-pixel groups x 4 channel octets = 720 units (3 per thread, the last partly idle), g = 8 x 16 x 4 = 512
    uintx4 xr[3][4], dr[2][4];
    auto load_tile = [&](const int t) {
        const int img = t / p.tiles_per_img, Y0 = (t - img * p.tiles_per_img) * TW_H;
        const char* xb = xpl + (long)img * p.img_b + (long)Y0 * p.row_b;          // padded row Y0 = image row Y0 - 1
        const char* gb = gpl + (long)img * p.img_b + (long)(Y0 + 1) * p.row_b;
#pragma unroll
        for (int it = 0; it < 3; ++it) {
            const int u = tid + it * 256;
            const int c8 = u & 3, qq = u >> 2, r = qq / 18, q = qq - r * 18;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = 4 * q - 4 + i;                                      // image column
                uintx4 v = {0u, 0u, 0u, 0u};
                if (u < 720 && c >= -1 && c <= TW_W) v = *(const uintx4*)(xb + (long)r * p.row_b + (c + 1) * 64 + c8 * 16);
                xr[it][i] = v;
            }
        }
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int u = tid + it * 256;
            const int c8 = u & 3, qq = u >> 2, r = qq >> 4, q = qq & 15;
#pragma unroll
            for (int i = 0; i < 4; ++i) dr[it][i] = *(const uintx4*)(gb + (long)r * p.row_b + (4 * q + i + 1) * 64 + c8 * 16);
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int it = 0; it < 3; ++it) {
            const int u = tid + it * 256;
            if (u < 720) {
                const int c8 = u & 3, qq = u >> 2, r = qq / 18, q = qq - r * 18;
                half8v h[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) h[i] = __builtin_bit_cast(half8v, xr[it][i]);
                char* o = s_x + (c8 * 8) * CSX + (r * RS + 4 * q + 4) * 2;
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    *(uintx2*)(o + j * CSX) = uintx2{bf16x2_rne((float)h[0][j], (float)h[1][j]), bf16x2_rne((float)h[2][j], (float)h[3][j])};
            }
        }
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int u = tid + it * 256;
            const int c8 = u & 3, qq = u >> 2, r = qq >> 4, q = qq & 15;
            char* o = s_d + (c8 * 8) * CSD + (r * TW_W + 4 * q) * 2;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const unsigned sel = (j & 1) ? 0x07060302u : 0x05040100u;
                const unsigned lo = __builtin_amdgcn_perm(dr[it][1][j >> 1], dr[it][0][j >> 1], sel);
                const unsigned hi2 = __builtin_amdgcn_perm(dr[it][3][j >> 1], dr[it][2][j >> 1], sel);
                *(uintx2*)(o + j * CSD) = uintx2{lo, hi2};
            }
        }
    };

    if (t0 < t1) load_tile(t0);
    for (int t = t0; t < t1; ++t) {
        __syncthreads();                                   // the previous tile's operand reads are done
        store_tile();
        __syncthreads();
        if (t + 1 < t1) load_tile(t + 1);                  // in flight under the MFMAs
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const int row = wave * 2 + (ks >> 2), g = ks & 3;
            const bf16x8 a = __builtin_bit_cast(bf16x8, *(const uintx4*)(s_d + l31 * CSD + (row * TW_W + g * 16 + hi * 8) * 2));
            if (with_bias) accb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8, ones), accb, 0, 0, 0);
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const char* bp = s_x + l31 * CSX + ((row + dy) * RS + g * 16 + hi * 8 + 8) * 2;
                const uintx4 cur = *(const uintx4*)bp;
                const unsigned pv = *(const unsigned*)(bp - 4), nx = *(const unsigned*)(bp + 16);
                const unsigned m1 = __builtin_amdgcn_alignbit(cur[1], cur[0], 16), m2 = __builtin_amdgcn_alignbit(cur[2], cur[1], 16),
                               m3 = __builtin_amdgcn_alignbit(cur[3], cur[2], 16);
                const uintx4 b0 = {__builtin_amdgcn_alignbit(cur[0], pv, 16), m1, m2, m3};
                const uintx4 b2 = {m1, m2, m3, __builtin_amdgcn_alignbit(nx, cur[3], 16)};
                acc[dy * 3 + 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8, b0), acc[dy * 3 + 0], 0, 0, 0);
                acc[dy * 3 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8, cur), acc[dy * 3 + 1], 0, 0, 0);
                acc[dy * 3 + 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8, b2), acc[dy * 3 + 2], 0, 0, 0);
            }
        }
    }
    // ---- the four waves' blocks meet in LDS: waves 2, 3 -> LDS, waves 0, 1 add; wave 1 -> LDS, wave 0 adds and writes the partial block
    // accumulator element r of lane (l31, hi) is (co = 8 (r >> 2) + 4 hi + (r & 3), ci = l31): stored as [tap][co][ci]
    float* red = (float*)smem;
    auto idx = [&](int tp, int r) { return tp * 1024 + (8 * (r >> 2) + 4 * hi + (r & 3)) * 32 + l31; };
    __syncthreads();
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
        float* out = p.ws + (((long)k * p.nsplit + split) * NPAIR + blockIdx.x) * 9216;
#pragma unroll
        for (int tp = 0; tp < 9; ++tp)
#pragma unroll
            for (int r = 0; r < 16; ++r) out[idx(tp, r)] = acc[tp][r] + red[idx(tp, r)];
    }
    if (with_bias) {       // every column of accb holds the same sums: column 0's lanes (l31 == 0) carry them
        __syncthreads();
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

extern "C" size_t srbh_trunk_wgrad_ws_bytes(int num_block, int B, int H, int W) {
    if (num_block <= 0 || B <= 0 || W != TW_W || H <= 0 || (H % TW_H) != 0) return 0;
    const int ntiles = B * (H / TW_H);
    const int nsplit = ntiles < 4 ? ntiles : 4;
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
    p.nsplit = p.ntiles < 4 ? p.ntiles : 4;
    p.tiles_per_split = (p.ntiles + p.nsplit - 1) / p.nsplit;
    p.ws = (float*)ws;
    p.wsb = (float*)ws + (size_t)p.n_rdb * p.nsplit * NPAIR * 9216;
    hipStream_t st = (hipStream_t)stream;
    SRBH_ONCE_PER_DEVICE(SRBH_HIP(hipFuncSetAttribute((const void*)trunk_wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, TWG_LDS_B)));
    hipLaunchKernelGGL(trunk_wgrad_kernel, dim3(NPAIR, p.nsplit, p.n_rdb), dim3(256), TWG_LDS_B, st, p);
    SRBH_HIP(hipGetLastError());
    hipLaunchKernelGGL(trunk_wgrad_reduce_kernel, dim3(NPAIR + 1, p.n_rdb), dim3(256), 0, st, p.ws, p.wsb, p.nsplit, p.n_rdb, dw_all, db_all);
    SRBH_HIP(hipGetLastError());
    return SRBH_OK;
}
