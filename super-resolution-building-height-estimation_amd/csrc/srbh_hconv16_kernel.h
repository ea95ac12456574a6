// srbh_hconv16_kernel.h -- the head's dominant convolution as its own kernel: 3x3, 16 -> 16 channels, one fp32 NHWC source, fp16 / bf16
// operands (included by srbh_head.hip inside its anonymous namespace, after HParams / h16_off / round4).
//
// Why a second kernel next to the hconv_f32_kernel template (round 2): the template runs ONE 4 x 64 tile per
// workgroup -- 16 384 workgroups of ~2.7 us for a B=64 layer -- and PMC / SQ counters showed it bound by that serial chain (launch,
// kernel-argument load, input load, LDS, MFMA, residual load, store), not by bytes: it moves exactly the algorithmic 537 MB at
// 3.4 TB/s, does not speed up when the bytes are halved, and a persistent walk bolted into the template blew its registers.  This
// kernel is written around the walk:
//   * a few workgroups per CU, each walks its XCD's contiguous run of tiles (halo rows from that XCD's L2);
//   * the weights (9 taps x 8 bytes per lane) are loaded once per workgroup;
//   * two LDS stages: while tile k is multiplied out of stage k&1 and stored, the global loads of tile k+1 are in flight (issued
//     right after tile k went to LDS) and the residual of tile k was requested before its MFMAs -- one barrier per tile;
//   * BatchNorm partial sums are kept in registers over the walk and flushed once per workgroup.
// Same arithmetic as hconv_f32_kernel<1, 3, 1, OPT> (same rounding, same MFMA order per pixel): results are bit-identical except the
// order in which the BatchNorm partial sums are added.  Restrictions (the host falls back to the template otherwise): c0 = 16,
// c1 = 0, cout = 16, W % 64 == 0, H % 4 == 0, 4-aligned strides, no PixelShuffle store, no second residual, no LeakyReLU.
//
// 16-bit tensors IN MEMORY (round 3; HParams::io_h16, element type = the operand type: fp16 for OPT 1, bf16 for OPT 2): S16 = the
// source holds 16-bit elements -- a staging unit is then ONE 8-byte load, and without a pre-affine it goes to LDS as it is (no VALU at
// all: the fp32 form spends ~10 VALU instructions per MFMA rounding); with a pre-affine (the producer's BatchNorm + ReLU) it is widened,
// transformed and rounded once.  SRBH_IO_RES1_H16 / SRBH_IO_OUT_H16 (run-time flags): the residual is read / the output written as
// 16-bit quads (one rounding in the epilogue; BatchNorm statistics are still taken from the fp32 values).  Used for the fp16
// activations of the inference chain and for the saved activations (fp16) / internal gradient tensors (bf16) of the training step.
// IO (compile-time, so that the all-fp32 form keeps its registers): bit 0 = the residual, bit 1 = the output is a 16-bit tensor
// BS (compile-time for the same reason: as a run-time flag the epilogue cost every instantiation 8-20 spilled registers):
// 0 = none, 1 = BatchNorm-backward sums with the c*ms + mh mask (c in the residual's registers)
// NIN (compile-time): 1 = NARROW input, c0 < 16 fp32 channels with pixel stride ld0 = any (the data gradients of the 1- / 7-channel output
//   convs: conv^T(dY[7], W) -- they ran the one-tile-per-workgroup template at 0.19-0.20 of the HBM roofline): a staging unit is four
//   scalar loads of the channels that exist, the rest of the 16-channel chunk is zero (the pack pads the weights the same way)
template <int OPT, int S16, int IO, int BS = 0, int NIN = 0>
__global__ __launch_bounds__(256, 3) void hconv16_kernel(const HParams p) {
    static_assert(OPT == 1 || OPT == 2, "16-bit operand forms only");
    static_assert(NIN == 0 || (S16 == 0 && BS == 0), "narrow input: fp32 source, plain epilogue");
    constexpr int ROWS = 6, COLS = 66, NIT = (ROWS * COLS * 4 + 255) / 256;        // 7 staging units per thread (the last one partial)
    constexpr int STAGE_B = ROWS * COLS * 32;                                       // 12 672 bytes per stage
    extern __shared__ __attribute__((aligned(16))) float hsm[];
    char* const s_base = (char*)hsm;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, kk = lane >> 4;
    const int cg = tid & 3;
    const int t_end = min((int)(blockIdx.x & 7) * p.tiles_per_xcd + p.tiles_per_xcd, p.ntiles);
    const int t_first = (blockIdx.x & 7) * p.tiles_per_xcd + (blockIdx.x >> 3), t_step = gridDim.x >> 3;

    // ---- per-thread constants of the walk
    short4v wa[9];
    {
        const short4v* wp = (const short4v*)p.w + lane;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) wa[tap] = wp[tap * 64];
    }
    floatx4 psc = {1.f, 1.f, 1.f, 1.f}, psh = {0.f, 0.f, 0.f, 0.f};
    if (p.pre_scale) { psc = *(const floatx4*)(p.pre_scale + cg * 4); psh = *(const floatx4*)(p.pre_shift + cg * 4); }
    const bool pre_relu = p.pre_relu != 0;
    const floatx4 e_bias = p.bias ? *(const floatx4*)(p.bias + kk * 4) : floatx4{0.f, 0.f, 0.f, 0.f};
    const floatx4 e_sc = p.post_scale ? *(const floatx4*)(p.post_scale + kk * 4) : floatx4{1.f, 1.f, 1.f, 1.f};
    const floatx4 e_sh = p.post_scale ? *(const floatx4*)(p.post_shift + kk * 4) : floatx4{0.f, 0.f, 0.f, 0.f};
    // staging unit `it` of this thread: window pixel (tid >> 2) + 64*it = (row ur[it], column uc[it]); tile-independent
    int uoff[NIT];          // element offset (r*W + col) * ld0 from the window origin
    int ulds[NIT];          // byte offset inside a stage (h16_off)
    unsigned urow = 0;      // 3 bits per unit: window row
    unsigned ucol1 = 0;     // bit it: the unit sits in window column 0 (left halo) ; bit 8+it: column 65 (right halo)
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
    // B-fragment reads: lane base per dx (the swizzle bit depends on (dx + l15) >> 3 only); (dy, 16-pixel group) are immediate offsets
    int bbase[3];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) bbase[dx] = (wave * COLS + dx + l15) * 32 + ((kk ^ ((((dx + l15) >> 3) & 1) << 1)) << 3);

    float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};
    floatx4 b_ms = {0.f, 0.f, 0.f, 0.f}, b_mh = {1.f, 1.f, 1.f, 1.f}, b_mean = {0.f, 0.f, 0.f, 0.f}, b_inv = {0.f, 0.f, 0.f, 0.f};
    if constexpr (BS != 0) {
        b_mean = *(const floatx4*)(p.bstat_mean + kk * 4);
        b_inv = *(const floatx4*)(p.bstat_invstd + kk * 4);
        if (p.bstat_ms) {               // (no mask given: every element passes, b_mh = 1 > 0)
            b_ms = *(const floatx4*)(p.bstat_ms + kk * 4);
            b_mh = *(const floatx4*)(p.bstat_mh + kk * 4);
        }
    }
    typedef typename std::conditional<S16 != 0, float2v, floatx4>::type ldv_t;      // a staging unit in flight: 8 or 16 bytes
    // loads in flight: the fp32 form holds ONE tile ahead (7 x 16 bytes per thread); a 16-bit source is half the bytes per tile, so at
    // the same depth only half the bytes were in flight per CU and the walk ran at the same ~85 us for half the traffic (load latency x
    // bytes in flight = bandwidth): the 16-bit forms therefore hold TWO tiles ahead in the same registers
    constexpr int DEPTH = S16 ? 2 : 1;
    ldv_t ld[DEPTH][NIT];
    unsigned okmask[DEPTH] = {};
    const bool has_pre = p.pre_scale != nullptr || pre_relu;
    constexpr bool res16 = (IO & 1) != 0, out16 = (IO & 2) != 0;
    auto issue = [&](auto slot_tag, const int t) {
        constexpr int SL = decltype(slot_tag)::value;
        const int img = t / p.tiles_per_img;
        const int trem = t - img * p.tiles_per_img;
        const int ty = trem / p.tiles_x, tx = trem - ty * p.tiles_x;
        const int Y0 = ty * 4, X0 = tx * 64;
        const char* tp = (const char*)p.src0 + (((long)img * p.H + (Y0 - 1)) * p.W + (X0 - 1)) * p.ld0 * (S16 ? 2 : 4);
        okmask[SL] = 0;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int r = (urow >> (3 * it)) & 7;
            bool ok = (unsigned)(Y0 - 1 + r) < (unsigned)p.H;
            if ((ucol1 >> it) & 1) ok = ok && X0 > 0;
            if ((ucol1 >> (8 + it)) & 1) ok = ok && X0 + 64 < p.W;
            if (it == NIT - 1) ok = ok && last_unit;
            if constexpr (S16) ld[SL][it] = float2v{0.f, 0.f};
            else ld[SL][it] = floatx4{0.f, 0.f, 0.f, 0.f};
            if (ok) {
                if constexpr (NIN != 0) {
                    const float* q = (const float*)tp + uoff[it];
                    floatx4 v4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (cg * 4 + j < p.c0) v4[j] = q[j];
                    ld[SL][it] = v4;
                } else {
                    ld[SL][it] = *(const ldv_t*)(tp + (long)uoff[it] * (S16 ? 2 : 4));
                }
                okmask[SL] |= 1u << it;
            }
        }
    };
    auto commit = [&](auto slot_tag, char* stage) {
        constexpr int SL = decltype(slot_tag)::value;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            if (it < NIT - 1 || last_unit) {
                if constexpr (S16) {
                    if (!has_pre) {          // 16-bit source, no transform: the quad goes to LDS as it is (zero padding = zero bits)
                        *(float2v*)(stage + ulds[it]) = ld[SL][it];
                        continue;
                    }
                }
                floatx4 a;
                if constexpr (S16) a = widen4<OPT>(ld[SL][it]);
                else a = ld[SL][it];
                if (okmask[SL] & (1u << it)) {
                    a = a * psc + psh;
                    if (pre_relu) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) a[j] = fmaxf(a[j], 0.f);
                    }
                }
                const float t4[4] = {a[0], a[1], a[2], a[3]};
                *(short4v*)(stage + ulds[it]) = round4<OPT>(t4);
            }
        }
    };

    using SL0 = std::integral_constant<int, 0>;
    using SL1 = std::integral_constant<int, DEPTH - 1>;
    if (t_first < t_end) issue(SL0{}, t_first);
    if (DEPTH == 2 && t_first + t_step < t_end) issue(SL1{}, t_first + t_step);
    auto tile = [&](auto slot_tag, const int t, const int buf) {
        char* const stage = s_base + buf * STAGE_B;
        commit(slot_tag, stage);
        const int img = t / p.tiles_per_img;
        const int trem = t - img * p.tiles_per_img;
        const int ty = trem / p.tiles_x, tx = trem - ty * p.tiles_x;
        const long pix0 = ((long)img * p.H + ty * 4 + wave) * p.W + tx * 64 + l15;
        // this tile's residual first, then the next tile's input: the epilogue can wait for the residual alone
        floatx4 rres[4];
        // (loading c after the MFMAs instead -- 16 fewer live registers across them -- measured level: 37.57 vs 37.59 ms per train step)
        if constexpr (BS == 1) { // backward-statistics epilogue: the BatchNorm input c rides in the residual's registers (host: no res1 then)
            const float* rp = p.bstat_c + pix0 * 16 + kk * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) rres[i] = *(const floatx4*)(rp + i * 16 * 16);
        }
        if (BS == 0 && p.res1) {
            if constexpr (res16) {
                const char* rp = (const char*)p.res1 + (pix0 * p.res1_ld + kk * 4) * 2;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float2v raw = *(const float2v*)(rp + (long)i * 16 * p.res1_ld * 2);
                    rres[i][0] = raw[0]; rres[i][1] = raw[1];          // (raw bits: widened in the epilogue, after the MFMAs)
                }
            } else {
                const float* rp = p.res1 + pix0 * p.res1_ld + kk * 4;
#pragma unroll
                for (int i = 0; i < 4; ++i) rres[i] = *(const floatx4*)(rp + i * 16 * p.res1_ld);
            }
        }
        if (t + DEPTH * t_step < t_end) issue(slot_tag, t + DEPTH * t_step);      // (into the registers `commit` has just emptied)
        __syncthreads();           // stage `buf` is complete; every wave is past the MFMAs of the tile before (they read the other stage)
        floatx4 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3, dx = tap - dy * 3;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const short4v b = *(const short4v*)(stage + bbase[dx] + (dy * COLS + i * 16) * 32);      // = h16_off(wave + dy, dx + i*16 + l15, kk)
                if constexpr (OPT == 1)
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(half4, wa[tap]), __builtin_bit_cast(half4, b), acc[i], 0, 0, 0);
                else
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(wa[tap], b, acc[i], 0, 0, 0);
            }
        }
        float* const o0 = p.out + pix0 * p.out_ld + p.out_coff + kk * 4;
        const bool vec_out = p.cout_store == 16 && (p.out_ld & 3) == 0 && (p.out_coff & 3) == 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            floatx4 v = acc[i];
            if (p.bias) v += e_bias;
            if (p.post_scale) v = v * e_sc + e_sh;
            if (BS == 0 && p.res1) {
                if constexpr (res16) v = v * p.res1_scale + widen4<OPT>(float2v{rres[i][0], rres[i][1]});
                else v = v * p.res1_scale + rres[i];
            }
            if (p.post_relu) {
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.f);
            }
            if constexpr (BS == 1) { // sum(dz), sum(dz * xhat) with dz = v where relu(bn(c)) is active (srbh_bn_bwd_reduce's sums)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float c = rres[i][q];
                    const float dz = fmaf(c, b_ms[q], b_mh[q]) > 0.f ? v[q] : 0.f;
                    ssum[q] += dz;
                    ssq[q] = fmaf(dz, (c - b_mean[q]) * b_inv[q], ssq[q]);
                }
            } else if (p.stats) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    ssum[q] += v[q];
                    ssq[q] += v[q] * v[q];
                }
            }
            if constexpr (out16) {     // (host: cout_store == 16, 4-aligned strides) one rounding here, none in the consumer
                const float t4[4] = {v[0], v[1], v[2], v[3]};
                *(short4v*)((char*)p.out + ((pix0 + i * 16) * p.out_ld + p.out_coff + kk * 4) * 2) = round4<OPT>(t4);
            } else if (vec_out) {
                *(floatx4*)(o0 + i * 16 * p.out_ld) = v;
            } else {                   // the 1- / 7-channel output convs (conv_last): scalar stores of the real channels
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (kk * 4 + q < p.cout_store) o0[i * 16 * p.out_ld + q] = v[q];
            }
        }
    };
    if constexpr (DEPTH == 1) {
        int buf = 0;
        for (int t = t_first; t < t_end; t += t_step, buf ^= 1) tile(SL0{}, t, buf);
    } else {
        for (int t = t_first; t < t_end;) {      // two tiles per trip: LDS stage and register slot are compile-time constants
            tile(SL0{}, t, 0);
            t += t_step;
            if (t >= t_end) break;
            tile(SL1{}, t, 1);
            t += t_step;
        }
    }
    if (p.stats) {
        __syncthreads();
        float* red = hsm;                      // [4 waves][2 moments][16 channels]
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float a = ssum[q], b = ssq[q];
#pragma unroll
            for (int m = 1; m < 16; m <<= 1) {
                a += __shfl_xor(a, m);
                b += __shfl_xor(b, m);
            }
            if (l15 == 0) {
                red[(wave * 2 + 0) * 16 + kk * 4 + q] = a;
                red[(wave * 2 + 1) * 16 + kk * 4 + q] = b;
            }
        }
        __syncthreads();
        if (tid < 32) {
            const int mom = tid >> 4, oc = tid & 15;
            double v = 0.0;
#pragma unroll
            for (int w = 0; w < 4; ++w) v += (double)red[(w * 2 + mom) * 16 + oc];
            double* slot = p.stats + (long)(blockIdx.x % NSLOT) * 2 * p.cout;
            atomicAdd(slot + mom * p.cout + oc, v);
        }
    }
}
