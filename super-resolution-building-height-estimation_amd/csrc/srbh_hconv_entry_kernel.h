// srbh_hconv_entry_kernel.h -- the ENTRY of a BasicBlock with a downsample branch (SR/HRfuse.py:142-159: conv1 = 3x3 and
// downsample[0] = 1x1 over the SAME input, the widest tensor of each head: 64 channels in HRfeature, 16 + 16 concatenated in the fuse
// heads) as ONE pass in the shape of hconv16_kernel (included by srbh_head.hip after it).
//
// The 1x1 conv is the centre tap with other weights: one more MFMA per (tile, 16-channel chunk, 16-pixel group) on the B fragment the
// 3x3 already holds.  The input is read once instead of twice (the template ran two launches, each one workgroup per tile).  Walk,
// stages and statistics as hconv16_kernel; the pipeline unit is (tile, chunk): while chunk c is multiplied, chunk c+1 (or chunk 0 of the
// next tile) is in flight.  All chunks' weights sit in LDS (A fragments, 8 bytes per lane and tap: (9 + 1) x 512 bytes per chunk),
// loaded once per workgroup.  fp16 / bf16 operands, fp32 accumulate, same rounding as the template -> same numbers per output.
// Restrictions (host falls back to two template launches): 16 output channels each, c0 % 16 == 0, c1 % 16 == 0, <= 5 chunks, no
// pre-affine, W % 64 == 0, H % 4 == 0, 4-aligned strides.
struct EParams {
    HParams a;                 // conv1: sources, bias / post / relu, out, stats ; a.w = 3x3 pack
    const float* w2;           // 1x1 pack (srbh_hpack_conv_h16 of downsample[0])
    const float* bias2; const float* post2_scale; const float* post2_shift;
    float* out2; int out2_ld, out2_coff;
    double* stats2;
    int nchunk;
};

// O16 (compile-time): both outputs are 16-bit tensors (fp16 for OPT 1)
// S16 (compile-time, OPT 1): the sources hold fp16 elements (SRBH_IO_SRC0_H16 [| SRBH_IO_SRC1_H16]): staged verbatim -- 8-byte loads,
// no rounding, half the registers in flight (the fp32-source forms spill 24 VGPRs under this launch bound; this one does not)
#ifndef SRBH_ENTRY_WGS_PER_CU
#define SRBH_ENTRY_WGS_PER_CU 3      // (A/B aid: 2 = 256 registers per lane, no spills, but two workgroups per CU instead of three)
#endif
template <int OPT, int O16, int S16 = 0>
__global__ __launch_bounds__(256, SRBH_ENTRY_WGS_PER_CU) void hconv_entry_kernel(const EParams e) {
    static_assert(OPT == 1 || OPT == 2, "16-bit operand forms only");
    static_assert(S16 == 0 || OPT == 1, "16-bit sources: fp16 only (element type = operand type)");
    const HParams& p = e.a;
    constexpr int ROWS = 6, COLS = 66, NIT = (ROWS * COLS * 4 + 255) / 256;
    constexpr int STAGE_B = ROWS * COLS * 32;
    extern __shared__ __attribute__((aligned(16))) float hsm[];
    char* const s_base = (char*)hsm;                              // 2 stages, then the weights
    char* const s_w = s_base + 2 * STAGE_B;                       // [chunk][10 taps: 9 of conv1, then the 1x1][64 lanes] 8 bytes
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, kk = lane >> 4;
    const int cg = tid & 3;
    const int nchunk = e.nchunk;
    const int t_end = min((int)(blockIdx.x & 7) * p.tiles_per_xcd + p.tiles_per_xcd, p.ntiles);
    const int t_first = (blockIdx.x & 7) * p.tiles_per_xcd + (blockIdx.x >> 3), t_step = gridDim.x >> 3;

    for (int u = tid; u < nchunk * 10 * 64; u += 256) {
        const int c = u / 640, r = u - c * 640, tap = r >> 6, ln = r & 63;
        const short4v v = tap < 9 ? ((const short4v*)p.w)[(c * 9 + tap) * 64 + ln] : ((const short4v*)e.w2)[c * 64 + ln];
        *(short4v*)(s_w + (long)u * 8) = v;
    }
    const floatx4 e_bias = p.bias ? *(const floatx4*)(p.bias + kk * 4) : floatx4{0.f, 0.f, 0.f, 0.f};
    const floatx4 e_sc = p.post_scale ? *(const floatx4*)(p.post_scale + kk * 4) : floatx4{1.f, 1.f, 1.f, 1.f};
    const floatx4 e_sh = p.post_scale ? *(const floatx4*)(p.post_shift + kk * 4) : floatx4{0.f, 0.f, 0.f, 0.f};
    const floatx4 d_bias = e.bias2 ? *(const floatx4*)(e.bias2 + kk * 4) : floatx4{0.f, 0.f, 0.f, 0.f};
    const floatx4 d_sc = e.post2_scale ? *(const floatx4*)(e.post2_scale + kk * 4) : floatx4{1.f, 1.f, 1.f, 1.f};
    const floatx4 d_sh = e.post2_scale ? *(const floatx4*)(e.post2_shift + kk * 4) : floatx4{0.f, 0.f, 0.f, 0.f};
    int upix[NIT], ulds[NIT];       // window pixel offset r*W + col ; byte offset inside a stage (h16_off)
    unsigned urow = 0, ucol1 = 0;
    {
        int r = 0, col = tid >> 2;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            upix[it] = r * p.W + col;
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

    float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f}, dsum[4] = {0.f, 0.f, 0.f, 0.f}, dsq[4] = {0.f, 0.f, 0.f, 0.f};
    typedef typename std::conditional<S16 != 0, short4v, floatx4>::type ld_t;
    ld_t ld[NIT];
    unsigned okmask = 0;
    const int usafe = p.W + 1;          // window pixel offset of the tile's pixel (Y0, X0): always inside the image
    auto issue = [&](const int t, const int c) {
        const int img = t / p.tiles_per_img;
        const int trem = t - img * p.tiles_per_img;
        const int ty = trem / p.tiles_x, tx = trem - ty * p.tiles_x;
        const int Y0 = ty * 4, X0 = tx * 64;
        const bool in0 = c * 16 < p.c0;
        const int ldp = in0 ? p.ld0 : p.ld1;
        const long eoff = (((long)img * p.H + (Y0 - 1)) * p.W + (X0 - 1)) * ldp + cg * 4 + (in0 ? c * 16 : c * 16 - p.c0);      // in elements
        const float* tp = (in0 ? p.src0 : p.src1) + eoff;
        const short* tp16 = (const short*)(in0 ? p.src0 : p.src1) + eoff;
        unsigned m = 0;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int r = (urow >> (3 * it)) & 7;
            bool ok = (unsigned)(Y0 - 1 + r) < (unsigned)p.H;
            if ((ucol1 >> it) & 1) ok = ok && X0 > 0;
            if ((ucol1 >> (8 + it)) & 1) ok = ok && X0 + 64 < p.W;
            if (it == NIT - 1) ok = ok && last_unit;
            // UNCONDITIONAL loads (a unit outside the image reads the tile's own first pixel; `commit` zeroes it): a load behind a branch hides the
            // number of memory operations in flight from the compiler and turns every later counted wait into s_waitcnt vmcnt(0)
            // (csrc/srbh_hblock16_kernel.h); it also cost this kernel 24 spilled registers at its launch bound
            const int po = ok ? upix[it] : usafe;
            if constexpr (S16 != 0) ld[it] = *(const short4v*)(tp16 + po * ldp);
            else ld[it] = *(const floatx4*)(tp + po * ldp);
            m |= ok ? 1u << it : 0u;
        }
        okmask = m;
    };
    auto commit = [&](char* stage) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            if (it < NIT - 1 || last_unit) {
                const bool uok = (okmask >> it) & 1;
                if constexpr (S16 != 0) {
                    *(short4v*)(stage + ulds[it]) = uok ? ld[it] : short4v{0, 0, 0, 0};
                } else {
                    const float t4[4] = {ld[it][0], ld[it][1], ld[it][2], ld[it][3]};
                    *(short4v*)(stage + ulds[it]) = uok ? round4<OPT>(t4) : short4v{0, 0, 0, 0};
                }
            }
        }
    };

    // (values loaded once are consumed in front of the walk; the prefetch below is issued ALWAYS: constant counts -> counted waits)
    asm volatile("" ::"v"(e_bias), "v"(e_sc), "v"(e_sh), "v"(d_bias), "v"(d_sc), "v"(d_sh));
    if (t_first < t_end) issue(t_first, 0);
    __syncthreads();                   // the weights are in LDS
    int buf = 0;
    for (int t = t_first; t < t_end; t += t_step) {
        floatx4 acc[4], acd[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = acd[i] = floatx4{0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < nchunk; ++c, buf ^= 1) {
            char* const stage = s_base + buf * STAGE_B;
            commit(stage);
            if (c + 1 < nchunk) issue(t, c + 1);
            else issue(t + t_step < t_end ? t + t_step : t, 0);          // (behind the range's end: the tile again, never used)
            __syncthreads();           // stage `buf` complete; every wave is past the MFMAs of the unit before (other stage)
            const char* wc = s_w + ((long)c * 640 + lane) * 8;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int dy = tap / 3, dx = tap - dy * 3;
                const short4v wa = *(const short4v*)(wc + tap * 512);
                short4v wd;
                if (tap == 4) wd = *(const short4v*)(wc + 9 * 512);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const short4v b = *(const short4v*)(stage + bbase[dx] + (dy * COLS + i * 16) * 32);
                    if constexpr (OPT == 1) {
                        acc[i] = __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(half4, wa), __builtin_bit_cast(half4, b), acc[i], 0, 0, 0);
                        if (tap == 4) acd[i] = __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(half4, wd), __builtin_bit_cast(half4, b), acd[i], 0, 0, 0);
                    } else {
                        acc[i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(wa, b, acc[i], 0, 0, 0);
                        if (tap == 4) acd[i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(wd, b, acd[i], 0, 0, 0);
                    }
                }
            }
        }
        const int img = t / p.tiles_per_img;
        const int trem = t - img * p.tiles_per_img;
        const int ty = trem / p.tiles_x, tx = trem - ty * p.tiles_x;
        const long pix0 = ((long)img * p.H + ty * 4 + wave) * p.W + tx * 64 + l15;
        float* const o1 = p.out + pix0 * p.out_ld + p.out_coff + kk * 4;
        float* const o2 = e.out2 + pix0 * e.out2_ld + e.out2_coff + kk * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            floatx4 v = acc[i];
            if (p.bias) v += e_bias;
            if (p.post_scale) v = v * e_sc + e_sh;
            if (p.post_relu) {
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.f);
            }
            floatx4 d = acd[i];
            if (e.bias2) d += d_bias;
            if (e.post2_scale) d = d * d_sc + d_sh;
            if (p.stats) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    ssum[q] += v[q]; ssq[q] += v[q] * v[q];
                    dsum[q] += d[q]; dsq[q] += d[q] * d[q];
                }
            }
            if constexpr (O16 != 0) {
                const float t4[4] = {v[0], v[1], v[2], v[3]};
                *(short4v*)((char*)p.out + ((pix0 + i * 16) * p.out_ld + p.out_coff + kk * 4) * 2) = round4<OPT>(t4);
            } else {
                *(floatx4*)(o1 + i * 16 * p.out_ld) = v;
            }
            if constexpr (O16 != 0) {
                const float t4[4] = {d[0], d[1], d[2], d[3]};
                *(short4v*)((char*)e.out2 + ((pix0 + i * 16) * e.out2_ld + e.out2_coff + kk * 4) * 2) = round4<OPT>(t4);
            } else {
                *(floatx4*)(o2 + i * 16 * e.out2_ld) = d;
            }
        }
    }
    if (p.stats) {
        __syncthreads();
        float* red = hsm;                      // [4 waves][4 rows: sum1, sq1, sum2, sq2][16 channels]
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float v4[4] = {ssum[q], ssq[q], dsum[q], dsq[q]};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float a = v4[k];
#pragma unroll
                for (int m = 1; m < 16; m <<= 1) a += __shfl_xor(a, m);
                if (l15 == 0) red[(wave * 4 + k) * 16 + kk * 4 + q] = a;
            }
        }
        __syncthreads();
        if (tid < 64) {
            const int k = tid >> 4, oc = tid & 15;
            double v = 0.0;
#pragma unroll
            for (int w = 0; w < 4; ++w) v += (double)red[(w * 4 + k) * 16 + oc];
            double* slot = (k < 2 ? p.stats : e.stats2) + (long)(blockIdx.x % NSLOT) * 2 * 16;
            atomicAdd(slot + (k & 1) * 16 + oc, v);
        }
    }
}


// ---- HRfeature's entry: 64 fp16 channels = ONE 128-byte row per pixel (round 4).  hconv_entry_kernel<1, O16, 1> stages a (tile, chunk) unit
// from 8-byte loads: 32 bytes of every pixel row per unit, 16 different lines per load instruction, a barrier per chunk.  Here a lane loads
// 16 bytes, 8 lanes cover a pixel's whole row, ALL FOUR chunks of a tile go to LDS at once (4 x 12.4 KB per stage, two stages + 20 KB of
// weights = 119 KB: one workgroup per CU) and the pipeline unit is the tile: one barrier per tile, the next tile's 13 loads per thread in
// flight under 4 x 40 MFMAs.  Same staging layout per chunk (h16_off swizzle: a 16-byte unit is the granule pair the swizzle moves as a whole),
// same fragment reads, same MFMA order (chunk 0..3, tap 0..8) and epilogue as hconv_entry_kernel: bit-identical outputs; BatchNorm partial
// sums differ only by their order of addition (other grid).
// (Round 6: two chunks per v_mfma_f32_16x16x32_f16 -- 2 x 40 instead of 4 x 40 instructions per wave and tile -- was built and measured: this
//  kernel 1.21 -> ~1.0 ms per 256 tiles, tiled prediction +0.5 %, training step unchanged (profiles/r06aa_*).  NOT kept: the other summation
//  order ends the bit-identity with hconv_entry_kernel, i.e. of the fp16 feature hand-off with the fp32 one (tests/test_gpu_feature_h16.py),
//  which is worth more than 0.5 %.)
template <int O16>
__global__ __launch_bounds__(256, 1) void hconv_entry64_kernel(const EParams e) {
    const HParams& p = e.a;
    constexpr int ROWS = 6, COLS = 66, NPX = ROWS * COLS, NU = NPX * 8, NIT = (NU + 255) / 256, NC = 4;
    constexpr int CH_B = NPX * 32, STAGE_B = NC * CH_B;
    typedef unsigned uint4e __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) float hsm[];
    char* const s_base = (char*)hsm;
    char* const s_w = s_base + 2 * STAGE_B;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, kk = lane >> 4;
    const int t_end = min((int)(blockIdx.x & 7) * p.tiles_per_xcd + p.tiles_per_xcd, p.ntiles);
    const int t_first = (blockIdx.x & 7) * p.tiles_per_xcd + (blockIdx.x >> 3), t_step = gridDim.x >> 3;
    for (int u = tid; u < NC * 10 * 64; u += 256) {
        const int c = u / 640, r = u - c * 640, tap = r >> 6, ln = r & 63;
        const short4v v = tap < 9 ? ((const short4v*)p.w)[(c * 9 + tap) * 64 + ln] : ((const short4v*)e.w2)[c * 64 + ln];
        *(short4v*)(s_w + (long)u * 8) = v;
    }
    const floatx4 e_bias = p.bias ? *(const floatx4*)(p.bias + kk * 4) : floatx4{0.f, 0.f, 0.f, 0.f};
    const floatx4 e_sc = p.post_scale ? *(const floatx4*)(p.post_scale + kk * 4) : floatx4{1.f, 1.f, 1.f, 1.f};
    const floatx4 e_sh = p.post_scale ? *(const floatx4*)(p.post_shift + kk * 4) : floatx4{0.f, 0.f, 0.f, 0.f};
    const floatx4 d_bias = e.bias2 ? *(const floatx4*)(e.bias2 + kk * 4) : floatx4{0.f, 0.f, 0.f, 0.f};
    const floatx4 d_sc = e.post2_scale ? *(const floatx4*)(e.post2_scale + kk * 4) : floatx4{1.f, 1.f, 1.f, 1.f};
    const floatx4 d_sh = e.post2_scale ? *(const floatx4*)(e.post2_shift + kk * 4) : floatx4{0.f, 0.f, 0.f, 0.f};
    // staging unit `it` of this thread: window pixel (tid >> 3) + 32*it, 16-byte piece pc = tid & 7 (channels pc*8 .. pc*8+7)
    const int pc = tid & 7;
    int bbase[3];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) bbase[dx] = (wave * COLS + dx + l15) * 32 + ((kk ^ ((((dx + l15) >> 3) & 1) << 1)) << 3);
    float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f}, dsum[4] = {0.f, 0.f, 0.f, 0.f}, dsq[4] = {0.f, 0.f, 0.f, 0.f};
    uint4e ld[NIT];
    auto issue = [&](const int t) {
        const int img = t / p.tiles_per_img;
        const int trem = t - img * p.tiles_per_img;
        const int ty = trem / p.tiles_x, tx = trem - ty * p.tiles_x;
        const int Y0 = ty * 4, X0 = tx * 64;
        const short* tp = (const short*)p.src0 + (((long)img * p.H + (Y0 - 1)) * p.W + (X0 - 1)) * 64 + pc * 8;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int px = (tid >> 3) + 32 * it;
            const int r = px / COLS, col = px - r * COLS;
            const bool ok = px < NPX && (unsigned)(Y0 - 1 + r) < (unsigned)p.H && (unsigned)(X0 - 1 + col) < (unsigned)p.W;
            ld[it] = uint4e{0u, 0u, 0u, 0u};
            if (ok) ld[it] = *(const uint4e*)(tp + ((long)r * p.W + col) * 64);
        }
    };
    auto commit = [&](char* stage) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int px = (tid >> 3) + 32 * it;
            if (px < NPX) {
                const int col = px % COLS;
                *(uint4e*)(stage + (pc >> 1) * CH_B + px * 32 + (((pc & 1) ^ ((col >> 3) & 1)) << 4)) = ld[it];
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
        floatx4 acc[4], acd[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = acd[i] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const char* wc = s_w + ((long)c * 640 + lane) * 8;
            const char* sc = stage + c * CH_B;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int dy = tap / 3, dx = tap - dy * 3;
                const half4 wa = *(const half4*)(wc + tap * 512);
                half4 wd;
                if (tap == 4) wd = *(const half4*)(wc + 9 * 512);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const half4 b = *(const half4*)(sc + bbase[dx] + (dy * COLS + i * 16) * 32);
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x16f16(wa, b, acc[i], 0, 0, 0);
                    if (tap == 4) acd[i] = __builtin_amdgcn_mfma_f32_16x16x16f16(wd, b, acd[i], 0, 0, 0);
                }
            }
        }
        const int img = t / p.tiles_per_img;
        const int trem = t - img * p.tiles_per_img;
        const int ty = trem / p.tiles_x, tx = trem - ty * p.tiles_x;
        const long pix0 = ((long)img * p.H + ty * 4 + wave) * p.W + tx * 64 + l15;
        float* const o1 = p.out + pix0 * p.out_ld + p.out_coff + kk * 4;
        float* const o2 = e.out2 + pix0 * e.out2_ld + e.out2_coff + kk * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            floatx4 v = acc[i];
            if (p.bias) v += e_bias;
            if (p.post_scale) v = v * e_sc + e_sh;
            if (p.post_relu) {
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.f);
            }
            floatx4 d = acd[i];
            if (e.bias2) d += d_bias;
            if (e.post2_scale) d = d * d_sc + d_sh;
            if (p.stats) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    ssum[q] += v[q]; ssq[q] += v[q] * v[q];
                    dsum[q] += d[q]; dsq[q] += d[q] * d[q];
                }
            }
            if constexpr (O16 != 0) {
                const float t4[4] = {v[0], v[1], v[2], v[3]};
                *(short4v*)((char*)p.out + ((pix0 + i * 16) * p.out_ld + p.out_coff + kk * 4) * 2) = round4<1>(t4);
                const float u4[4] = {d[0], d[1], d[2], d[3]};
                *(short4v*)((char*)e.out2 + ((pix0 + i * 16) * e.out2_ld + e.out2_coff + kk * 4) * 2) = round4<1>(u4);
            } else {
                *(floatx4*)(o1 + i * 16 * p.out_ld) = v;
                *(floatx4*)(o2 + i * 16 * e.out2_ld) = d;
            }
        }
    }
    if (p.stats) {
        __syncthreads();
        float* red = hsm;                      // [4 waves][4 rows: sum1, sq1, sum2, sq2][16 channels]
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float v4[4] = {ssum[q], ssq[q], dsum[q], dsq[q]};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float a = v4[k];
#pragma unroll
                for (int m = 1; m < 16; m <<= 1) a += __shfl_xor(a, m);
                if (l15 == 0) red[(wave * 4 + k) * 16 + kk * 4 + q] = a;
            }
        }
        __syncthreads();
        if (tid < 64) {
            const int k = tid >> 4, oc = tid & 15;
            double v = 0.0;
#pragma unroll
            for (int w = 0; w < 4; ++w) v += (double)red[(w * 4 + k) * 16 + oc];
            double* slot = (k < 2 ? p.stats : e.stats2) + (long)(blockIdx.x % NSLOT) * 2 * 16;
            atomicAdd(slot + (k & 1) * 16 + oc, v);
        }
    }
}
