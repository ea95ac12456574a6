// srbh_hconv_up_kernel.h -- the Upsampler's convolution (SR/HRfuse.py:17-44: conv3x3 16 -> 64 with bias, then PixelShuffle(2)) as a kernel
// of its own in the shape of hconv16_kernel (included by srbh_head.hip after it).
//
// Why: the hconv_f32_kernel template runs one 4 x 64 tile per workgroup with four output blocks and passes every 16-pixel group through an
// LDS slice to order the PixelShuffle store (16 scalar ds_writes + 4 reads + 2 waits per group): 0.45 ms per 128 tiles and head for 0.4 GB
// -- not byte-bound (the fp16 output of round 4 did not make it faster).  Here:
//   * persistent walk, two LDS stages, next tile's loads in flight under the MFMAs (hconv16_kernel's pipeline), weights in LDS (18 KB);
//   * NO LDS pass for the PixelShuffle: the weight rows are packed SUB-PIXEL-MAJOR (pixelshuffle2 == 2: packed row ob*16 + kk*4 + q holds
//     the conv channel (kk*4 + ob)*4 + q, i.e. output channel kk*4 + ob, sub-pixel q) -- a lane's four accumulator blocks then hold, for
//     each of its pixel's four sub-pixels, FOUR CONSECUTIVE output channels: one 16-byte (fp32) / 8-byte (fp16) store per sub-pixel, the
//     four kk lanes complete the output pixel's 16 channels, the dx = 0 / 1 stores of a wave fill alternate pixels of the same lines.
// Same products in the same order per output as the template (one 16-channel chunk, taps 0..8): bit-identical results.
// Restrictions (host: srbh_hconv_up_supported; anything else keeps the template with the standard pack): fp16 operands, c0 = 16, c1 = 0,
// cout = 64, no pre-affine / residual / statistics / post ops, W % 64 == 0, H % 4 == 0.
template <int S16, int O16>
__global__ __launch_bounds__(256, 2) void hconv_up_kernel(const HParams p) {
    constexpr int ROWS = 6, COLS = 66, NIT = (ROWS * COLS * 4 + 255) / 256;
    constexpr int STAGE_B = ROWS * COLS * 32;
    extern __shared__ __attribute__((aligned(16))) float hsm[];
    char* const s_base = (char*)hsm;
    char* const s_w = s_base + 2 * STAGE_B;                       // [9 taps][4 ob][64 lanes] 8 bytes
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, kk = lane >> 4;
    const int cg = tid & 3;
    const int t_end = min((int)(blockIdx.x & 7) * p.tiles_per_xcd + p.tiles_per_xcd, p.ntiles);
    const int t_first = (blockIdx.x & 7) * p.tiles_per_xcd + (blockIdx.x >> 3), t_step = gridDim.x >> 3;
    for (int u = tid; u < 36 * 64; u += 256) *(short4v*)(s_w + (long)u * 8) = ((const short4v*)p.w)[u];
    floatx4 e_bias[4];
#pragma unroll
    for (int ob = 0; ob < 4; ++ob) e_bias[ob] = p.bias ? *(const floatx4*)(p.bias + ob * 16 + kk * 4) : floatx4{0.f, 0.f, 0.f, 0.f};
    int uoff[NIT], ulds[NIT];
    unsigned urow = 0, ucol1 = 0;
    {
        int r = 0, col = tid >> 2;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            uoff[it] = (r * p.W + col) * p.ld0 + cg * 4;
            ulds[it] = (r * COLS + col) * 32 + ((cg ^ ((col >> 2) & 2)) << 3);
            urow |= (unsigned)r << (3 * it);
            if (col == 0) ucol1 |= 1u << it;
            if (col == COLS - 1) ucol1 |= 1u << (8 + it);
            const bool wrapped = col + 64 >= COLS;
            col += wrapped ? 64 - COLS : 64;
            r += wrapped ? 1 : 0;
        }
    }
    const bool last_unit = tid + (NIT - 1) * 256 < ROWS * COLS * 4;
    int bbase[3];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) bbase[dx] = (wave * COLS + dx + l15) * 32 + ((kk ^ ((((dx + l15) >> 3) & 1) << 1)) << 3);
    typedef typename std::conditional<S16 != 0, float2v, floatx4>::type ldv_t;
    ldv_t ld[NIT];
    auto issue = [&](const int t) {
        const int img = t / p.tiles_per_img;
        const int trem = t - img * p.tiles_per_img;
        const int ty = trem / p.tiles_x, tx = trem - ty * p.tiles_x;
        const int Y0 = ty * 4, X0 = tx * 64;
        const char* tp = (const char*)p.src0 + (((long)img * p.H + (Y0 - 1)) * p.W + (X0 - 1)) * p.ld0 * (S16 ? 2 : 4);
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int r = (urow >> (3 * it)) & 7;
            bool ok = (unsigned)(Y0 - 1 + r) < (unsigned)p.H;
            if ((ucol1 >> it) & 1) ok = ok && X0 > 0;
            if ((ucol1 >> (8 + it)) & 1) ok = ok && X0 + 64 < p.W;
            if (it == NIT - 1) ok = ok && last_unit;
            if constexpr (S16) ld[it] = float2v{0.f, 0.f};
            else ld[it] = floatx4{0.f, 0.f, 0.f, 0.f};
            if (ok) ld[it] = *(const ldv_t*)(tp + (long)uoff[it] * (S16 ? 2 : 4));
        }
    };
    auto commit = [&](char* stage) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            if (it < NIT - 1 || last_unit) {
                if constexpr (S16) {
                    *(float2v*)(stage + ulds[it]) = ld[it];
                } else {
                    const float t4[4] = {ld[it][0], ld[it][1], ld[it][2], ld[it][3]};
                    *(short4v*)(stage + ulds[it]) = round4<1>(t4);
                }
            }
        }
    };
    if (t_first < t_end) issue(t_first);
    __syncthreads();                   // the weights are in LDS
    int buf = 0;
    for (int t = t_first; t < t_end; t += t_step, buf ^= 1) {
        char* const stage = s_base + buf * STAGE_B;
        commit(stage);
        if (t + t_step < t_end) issue(t + t_step);
        __syncthreads();               // stage `buf` complete; every wave is past the MFMAs of the tile before (other stage)
        floatx4 acc[4][4];
#pragma unroll
        for (int ob = 0; ob < 4; ++ob)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[ob][i] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3, dx = tap - dy * 3;
            half4 wa[4];
#pragma unroll
            for (int ob = 0; ob < 4; ++ob) wa[ob] = *(const half4*)(s_w + ((long)(tap * 4 + ob) * 64 + lane) * 8);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const half4 b = *(const half4*)(stage + bbase[dx] + (dy * COLS + i * 16) * 32);
#pragma unroll
                for (int ob = 0; ob < 4; ++ob) acc[ob][i] = __builtin_amdgcn_mfma_f32_16x16x16f16(wa[ob], b, acc[ob][i], 0, 0, 0);
            }
        }
        const int img = t / p.tiles_per_img;
        const int trem = t - img * p.tiles_per_img;
        const int ty = trem / p.tiles_x, tx = trem - ty * p.tiles_x;
        const int Y = ty * 4 + wave;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int X = tx * 64 + i * 16 + l15;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const long o = ((((long)img * 2 * p.H + 2 * Y + (q >> 1)) * (2 * p.W) + 2 * X + (q & 1)) * 16 + kk * 4);
                const float t4[4] = {acc[0][i][q] + e_bias[0][q], acc[1][i][q] + e_bias[1][q], acc[2][i][q] + e_bias[2][q], acc[3][i][q] + e_bias[3][q]};
                if constexpr (O16) *(short4v*)((short*)p.out + o) = round4<1>(t4);
                else *(floatx4*)(p.out + o) = floatx4{t4[0], t4[1], t4[2], t4[3]};
            }
        }
    }
}
