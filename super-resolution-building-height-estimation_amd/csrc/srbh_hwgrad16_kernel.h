// srbh_hwgrad16_kernel.h -- weight gradient of the head's dominant layer (3x3, 16 -> 16 channels, one fp32 NHWC source, bf16 operands)
// in the shape of hconv16_kernel (included by srbh_head_bwd.hip inside its anonymous namespace, after WGParams / bf16_pair).
//
// hwgrad_b16_kernel already walks tiles persistently, but one tile at a time: load -> LDS -> barrier -> MFMA with two workgroups per CU
// to overlap (0.43 of the HBM peak; issuing the next tile's loads early cost 281 registers with its 8-row tiles).  Here the tile is
// 4 x 64 (48 registers of loads in flight instead of 80), LDS holds two stages (25 KiB each: three workgroups per CU), and the
// global loads of tile k+1 fly while tile k is multiplied: one barrier per tile.  The nine 16x16 accumulators live in registers over
// the whole walk and are flushed once per workgroup into the same workspace / deterministic two-stage reduction as the other forms.
// Same operand rounding (bf16 RNE of the transformed input and of dY) and fp32 accumulation as hwgrad_b16_kernel; the summation
// order differs (tile shape), i.e. results agree to fp32 rounding, not bit for bit.
// Restrictions (host falls back otherwise): c0 = 16, c1 = 0, cout = 16, ksize 3, W % 64 == 0, H % 4 == 0, ld0 % 4 == 0.
struct WG16T {
    static constexpr int QX = 18;                     // staged 4-pixel groups per row: image columns X0-4 .. X0+67
    static constexpr int SX = 260;                    // dwords per staged X channel (6 rows x 18 quads x 2 = 216, padded to = 4 mod 64)
    static constexpr int SD = 132;                    // dwords per staged dY channel (4 rows x 16 quads x 2 = 128, padded)
    static constexpr int STAGE_DW = 16 * SX + 16 * SD;
    static constexpr int LDS_B = 2 * STAGE_DW * 4;    // 50 176 bytes (>= the 36 864-byte flush buffer)
};

// 16-bit tensors in memory (round 3, WGParams::io): XS = the source holds fp16 elements (a saved activation: widened, transformed by the
// pre-affine as before, rounded to bf16 while staged), DS = dY holds bf16 elements (an internal gradient tensor: its bits ARE the
// operand, no rounding).  A staging item is then an 8-byte load per pixel; the raw quads stay in registers until the LDS store.
// DS = 2: a NARROW fp32 dY (cout_total < 16: the 1- / 7-channel output convs conv_last, whose weight gradients used to fall back to the
// fp32-MFMA kernel at 0.10-0.13 of the HBM peak): channels >= cout_total are staged as zeros, the pixel records are not 16-byte
// aligned, so the quads are assembled from guarded scalar loads; dW rows >= cout_total come out zero and are never read.
template <int XS, int DS>
__global__ __launch_bounds__(256, 3) void hwgrad16_kernel(const WGParams p) {
    extern __shared__ __attribute__((aligned(16))) float wsm[];
    using G = WG16T;
    constexpr int QX = G::QX, SX = G::SX, SD = G::SD;
    unsigned* const s_base = (unsigned*)wsm;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, kk = lane >> 4;
    const int cg = tid & 3;
    const int t_end = min((int)(blockIdx.x & 7) * p.tiles_per_xcd + p.tiles_per_xcd, p.ntiles);
    const int t_first = (blockIdx.x & 7) * p.tiles_per_xcd + (blockIdx.x >> 3), t_step = gridDim.x >> 3;

    floatx4 psc = {1.f, 1.f, 1.f, 1.f}, psh = {0.f, 0.f, 0.f, 0.f};
    if (p.pre_scale) { psc = *(const floatx4*)(p.pre_scale + cg * 4); psh = *(const floatx4*)(p.pre_shift + cg * 4); }
    const bool pre_relu = p.pre_relu != 0;
    // staging items of this thread (tile independent): X window item it = (row xr, quad xq) for it < 2, one dY item (row dr, quad dq)
    constexpr int NIX = 2;                            // 6 * 18 * 4 = 432 items: the second iteration is partial
    int xoff[NIX], xlds[NIX];
    int xr[NIX], xq[NIX];
#pragma unroll
    for (int it = 0; it < NIX; ++it) {
        const int q = (tid + it * 256) >> 2;
        xr[it] = q / QX;
        xq[it] = q - xr[it] * QX;
        xoff[it] = (xr[it] * p.W + xq[it] * 4) * p.ld0 + cg * 4;
        xlds[it] = cg * 4 * SX + q * 2;
    }
    const bool x1_valid = tid + 256 < 6 * QX * 4;
    const int dq_ = tid >> 2;
    const int doff = ((dq_ >> 4) * p.W + (dq_ & 15) * 4) * p.cout_total + cg * 4;
    const int dlds = 16 * SX + cg * 4 * SD + dq_ * 2;

    floatx4 acc[9];
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) acc[tp] = floatx4{0.f, 0.f, 0.f, 0.f};
    typedef typename std::conditional<XS != 0, float2w, floatx4>::type lxv_t;
    typedef typename std::conditional<DS == 1, float2w, floatx4>::type ldv_t;
    lxv_t lx[NIX][4];
    ldv_t ld[4];
    unsigned okx = 0;
    auto issue = [&](const int t) {
        const int img = t / p.tiles_per_img;
        const int trem = t - img * p.tiles_per_img;
        const int ty = trem / p.tiles_x, tx = trem - ty * p.tiles_x;
        const int Y0 = ty * 4, X0 = tx * 64;
        const char* xp = (const char*)p.src0 + (((long)img * p.H + (Y0 - 1)) * p.W + (X0 - 4)) * p.ld0 * (XS ? 2 : 4);
        const char* dp = (const char*)p.dy + (((long)img * p.H + Y0) * p.W + X0) * p.cout_total * (DS == 1 ? 2 : 4);
        okx = 0;
#pragma unroll
        for (int it = 0; it < NIX; ++it) {
            bool ok = (unsigned)(Y0 - 1 + xr[it]) < (unsigned)p.H;
            if (xq[it] == 0) ok = ok && X0 > 0;
            if (xq[it] == QX - 1) ok = ok && X0 + 64 < p.W;
            if (it == 1) ok = ok && x1_valid;
#pragma unroll
            for (int i = 0; i < 4; ++i) lx[it][i] = lxv_t{};
            if (ok) {
#pragma unroll
                for (int i = 0; i < 4; ++i) lx[it][i] = *(const lxv_t*)(xp + (long)(xoff[it] + i * p.ld0) * (XS ? 2 : 4));
                okx |= 1u << it;
            }
        }
        if constexpr (DS == 2) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float* q = (const float*)dp + doff + i * p.cout_total;
                floatx4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (cg * 4 + j < p.cout_total) v[j] = q[j];
                ld[i] = v;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) ld[i] = *(const ldv_t*)(dp + (long)(doff + i * p.cout_total) * (DS == 1 ? 2 : 4));
        }
    };
    auto commit = [&](unsigned* stage) {
#pragma unroll
        for (int it = 0; it < NIX; ++it) {
            if (it == 0 || x1_valid) {
                floatx4 xv[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if constexpr (XS != 0) xv[i] = widen_h4(lx[it][i]);
                    else xv[i] = lx[it][i];
                }
                if (okx & (1u << it)) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        floatx4 a = xv[i] * psc + psh;
                        if (pre_relu) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) a[j] = fmaxf(a[j], 0.f);
                        }
                        xv[i] = a;
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    *(uint2w*)(stage + xlds[it] + j * SX) = uint2w{bf16_pair(xv[0][j], xv[1][j]), bf16_pair(xv[2][j], xv[3][j])};
            }
        }
        if constexpr (DS == 1) {      // bf16 in memory: channel j of the 4 pixels = 16-bit fields of the raw quads
#pragma unroll
            for (int j = 0; j < 4; ++j)
                *(uint2w*)(stage + dlds + j * SD) = uint2w{b16_field_pair(ld[0], ld[1], j), b16_field_pair(ld[2], ld[3], j)};
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                *(uint2w*)(stage + dlds + j * SD) = uint2w{bf16_pair(ld[0][j], ld[1][j]), bf16_pair(ld[2][j], ld[3][j])};
        }
    };

    if (t_first < t_end) issue(t_first);
    int buf = 0;
    const int abase = 16 * SX + l15 * SD + (wave * 16 + kk) * 2;            // A (dY) fragment: + g*8 dwords per 16-pixel group
    const int bbase = l15 * SX + (wave * QX + 1 + kk) * 2;                   // B (X) fragment: + dy*QX*2 + g*8
    for (int t = t_first; t < t_end; t += t_step, buf ^= 1) {
        unsigned* const stage = s_base + buf * G::STAGE_DW;
        commit(stage);
        if (t + t_step < t_end) issue(t + t_step);
        __syncthreads();           // stage `buf` complete; every wave is past the MFMAs of the tile before (other stage)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const uint2w a2 = *(const uint2w*)(stage + abase + g * 8);
            const short4w a = __builtin_bit_cast(short4w, a2);
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const unsigned* rp = stage + bbase + dy * QX * 2 + g * 8;
                const uint2w cur = *(const uint2w*)rp;
                const unsigned pv = rp[-1], nx = rp[2];
                const unsigned mid = __builtin_amdgcn_alignbit(cur[1], cur[0], 16);
                const uint2w b0 = {__builtin_amdgcn_alignbit(cur[0], pv, 16), mid};
                const uint2w b2 = {mid, __builtin_amdgcn_alignbit(nx, cur[1], 16)};
                acc[dy * 3 + 0] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, __builtin_bit_cast(short4w, b0), acc[dy * 3 + 0], 0, 0, 0);
                acc[dy * 3 + 1] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, __builtin_bit_cast(short4w, cur), acc[dy * 3 + 1], 0, 0, 0);
                acc[dy * 3 + 2] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, __builtin_bit_cast(short4w, b2), acc[dy * 3 + 2], 0, 0, 0);
            }
        }
    }
    // flush: D[row = oc = kk*4 + r][col = ci = l15]  (layout and workspace as hwgrad_b16_kernel: nob = nchunk = 1)
    __syncthreads();
    float* s_red = wsm;
#pragma unroll
    for (int tp = 0; tp < 9; ++tp)
#pragma unroll
        for (int r = 0; r < 4; ++r) s_red[((wave * 9 + tp) * 16 + kk * 4 + r) * 16 + l15] = acc[tp][r];
    __syncthreads();
    for (int u = tid; u < 9 * 256; u += 256) {
        const float v = s_red[u] + s_red[9 * 256 + u] + s_red[2 * 9 * 256 + u] + s_red[3 * 9 * 256 + u];
        p.ws[(long)blockIdx.x * (9 * 256) + u] = v;
    }
}
