// srbh_ptrunk3_kernel.h -- persistent trunk kernel, variant 3: the protocol and data layout of ptrunk_kernel (one workgroup
// per (image, 8 rows), in-L2 halo exchange, resident plane 0, register-resident RDB stream), re-cut for the instruction
// stream.  Included by srbh_ptrunk.hip inside its anonymous namespace (shares PLayer / PParams / the LDS map).
//
// What the r01 kernel's ISA showed (one wave per SIMD: every instruction that is not in the shadow of an MFMA is wall clock):
//   * a step's ~20 LDS-DMA statements were emitted in clumps of 6-7 *in front of* a group's MFMAs, each clump idling the
//     matrix core for ~150 cycles; 4-6 of them were EXEC-masked no-ops that still cost their issue slots, and every step
//     recomputed its masks through v_cndmask / v_readfirstlane;
//   * ~70 instructions (descriptor arithmetic, mask selects, the first LDS reads, six DMAs) sat between the step barrier and
//     the first MFMA.
// Here the kind of staging a step performs is a COMPILE-TIME property (KIND), so a step contains exactly the DMA statements
// it needs and no mask arithmetic, and the order of its instructions is pinned with scheduling fences:
//   MFMA ; <= 2 ds_read of the next group  |  MFMA ; one DMA statement  -- every non-MFMA instruction sits behind an MFMA.
// The DMA statement carries its own wait states (M0 write -> LDS-DMA; a VALU-written scalar base -> VMEM), so it is
// hazard-proof by construction wherever the compiler puts the registers (tests/test_isa_hazards.py still scans the ISA).
// Restrictions (the host falls back to ptrunk_kernel otherwise): W == 64, H a multiple of 8 (full tiles: no store
// predication in the epilogues).  Same arithmetic in the same order as ptrunk_kernel: bit-identical results.
#pragma once

#ifndef P3_S1
#define P3_S1 1                 // 1: x's second chunk and X1 are staged from registers; 0: every plane by LDS-DMA (A/B aid)
#endif
#ifndef P3_X2REG
#define P3_X2REG 1              // 1: X2 is register-resident as well (3 more plane re-reads per RDB leave the fabric)
#endif
#ifndef P3_SKIPST
#define P3_SKIPST 1             // 1: a register-resident plane is stored to memory only where a neighbour reads it (its halo rows)
#endif
#ifndef P3_LAZYDRAIN
// 1 (round 3): a layer prologue waits only for what the layer's step 0 READS (the LDS-DMA issued during the previous layer's last
// step: `s_waitcnt vmcnt(<stores of the epilogue in between>)`), not for the epilogue's global stores to be acknowledged by L2 as
// well (r02: vmcnt(0), 1.1 k cycles per cout-32 prologue, 1.9 k behind the whole-plane stores of X3 / X4, with the matrix core idle);
// the stores drain under step 0's MFMAs and the progress counter is published behind the layer's FIRST top-of-step barrier instead
// (whose vmcnt(0) then covers them for free), in front of any neighbour check of that layer: every workgroup publishes layer L before
// it waits for layer L, so the exchange cannot deadlock.
#define P3_LAZYDRAIN 1
#endif
// The next group's LDS reads are spread EVENLY over the group's MFMA shadows (round 3: one read behind every MFMA of a cout-32 group, every
// second MFMA of conv5's) in the order of their FIRST USE, and the staging items share those shadows (LDS and VMEM are separate issue
// paths).  The r02 schedule packed a group's 9 (12) reads, two per shadow, behind its LAST 5 (6) MFMAs: all four waves march in lock-step,
// so every group boundary put 36 (48) ds_read_b128 into a 160 (192)-cycle window with the fragment the next group's FIRST MFMA needs
// issued 7th.

#ifndef P3_SEAM
// 1 (round 5): the RDB seam without a cold stage.  conv5's epilogue ALSO ds_writes the own rows of the new x's first chunk into the resident
// plane (phase-B stage 0, whose address range that is, is free during conv5's last step and epilogue), its last K step prefetches the next
// conv1's first 18 KiB of weights into the part of phase-B stage 0 behind the plane (= phase-A stage 0's INPUT area, which conv1's step 0
// does not read: it reads the resident plane), so that behind the neighbours' flags only the two halo rows are fetched (2 DMA statements
// instead of 11 + 5), and the own rows of that chunk go to memory only where a neighbour reads them.  Same arithmetic, same bits.
#define P3_SEAM 1
#endif

// The neighbour-progress check is a property of the WAVE, inside the step that fetches the halo rows (round 5; it used to be thread 0
// polling the two progress words between two steps -- an L2 round trip with the matrix core idle, an LDS word and two extra barriers, five
// times per RDB).  Every wave issues a slice of each halo DMA statement, so every wave can check for itself: the two polls go out at the top
// of the step as untracked loads, the step's other staging items (weights, the register-staged rows) and their MFMAs run, and the wave
// looks at the result -- `s_waitcnt vmcnt(<DMA statements issued since>)` -- only in front of the halo statements, which are the LAST
// staging items of such a step.  No LDS word, no barrier; a neighbour that is late makes the wave spin (bounded) where it stands.  A
// timed-out spin sets the error word and the wave CARRIES ON (uniform control flow, every later spin ends at once on the error word): the
// launch finishes with garbage that poison_on_error_kernel turns into NaN.
// The counted wait is only as good as the count: a VMEM statement whose EXEC mask is EMPTY for a wave never reaches vmcnt, so the window
// between the polls and the check may hold only statements every wave issues (see `stage_item`: the half-masked fifth weight statement
// sits behind the check).  With it inside the window waves 2, 3 counted one too many, looked at the second poll (the lower neighbour's --
// theirs) before it had landed and took whatever the register held: bit-identical in every quiet test and soak, 4 of 300 prefetches
// wrong (once NaN) when the trunk shared the chip with the training step (profiles/r05bo_soak_pipelined_variants.txt; fixed:
// profiles/r05bp_soak_fixed.txt, 1 500 of 1 500 identical; tests/test_gpu_pipeline.py keeps a short contended soak).

#ifndef P3_PREREAD
// 1 (round 5): the step barrier sits INSIDE the step, in front of the MFMA P3_PRE_AT of its last group, and the LDS reads of the NEXT step's
// first group ride in the shadows behind it.  With the barrier between two steps every step opened with its first group's reads (9 / 12
// ds_read_b128 per wave, four waves at once on a 128 B/clk port) in front of an idle matrix core.  At the barrier every wave has drained its
// own LDS-DMA (vmcnt(0)) and its LDS reads (lgkmcnt(0): the last group's operands are in registers), so the stage the next step reads is
// complete and the stage this step read is free for the next step's staging items -- the two facts the top-of-step barrier established.
// Steps of one layer only (the epilogue separates layers).
#define P3_PREREAD 1
#endif
#ifndef P3_TRAIN
// 1: the kernel also runs the SR-stage TRAINING forward (PParams.dense_stride / keep_all / out_pixel: one dense buffer per RDB, whole planes,
// fp32 output in pixel order); 0 compiles those run-time switches out (A/B aid: what they cost the inference launch)
#define P3_TRAIN 1
#endif
#ifndef P3_PRE_AT
#define P3_PRE_AT 2
#endif

// PROF = 1 (developer timeline, SRBH_PT_PROF): s_memtime stamps per layer in ptrunk_kernel's 6-slot format
// BW = 1 (round 5): the BACKWARD of the dense blocks on the same instruction stream.  The data gradient of an RDB is an RDB run on
// gradients (csrc/srbh_rrdbnet.hip, srbh_rrdbnet_trunk_train_backward: G = [g5 | g4 | g3 | g2 | g1], transposed + flipped weight slices stacked
// along K, the shapes of the forward exactly), so the launch walks the RDBs in reverse over a row of G buffers with three differences: bf16
// operands (v_mfma_f32_32x32x16_bf16: gradients need fp32's exponent range; the 16-bit planes are rounded RNE as the per-layer kernel does),
// the cout-32 epilogue multiplies by the LeakyReLU derivative read from the SAVED forward plane (y > 0 ? 1 : 0.2; pp.mask + rdb *
// pp.mask_stride, plane 5 - kk) instead of applying bias + LeakyReLU, and every plane is kept (the weight gradients read them).  The residual
// recurrences are the forward's own: with the stream held as 0.04 x the gradient, `x = 0.2 conv5 + x` is g5' = 0.2 (dx + cur) and the
// RRDB-closing `x = 0.2 x + x_rrdb` is the RRDB's skip connection (see srbh_rrdbnet_trunk_train_backward_persistent).
template <int PROF, int BW = 0>
__global__ __launch_bounds__(256, 1) void ptrunk3_kernel(const PParams pp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int wr = wave >> 1, wc = wave & 1;
    const int t = xcd_remap(blockIdx.x, pp.nblocks);
    const int img = t / pp.tiles_per_img;
    const int ty = t - img * pp.tiles_per_img;
    const int Y0 = ty * TILE_H;
    const int up = ty > 0 ? t - 1 : -1, dn = ty + 1 < pp.tiles_per_img ? t + 1 : -1;

    // ---- geometry that is identical for every layer (as ptrunk_kernel)
    int goff[G::NJ];
#pragma unroll
    for (int j = 0; j < G::NJ; ++j) {
        const int u0 = j * 256 + tid;
        const int u = u0 < G::UNITS ? u0 : 0;
        const int trow = u / (G::COLS * 4);
        const int rem = u - trow * (G::COLS * 4);
        const int pc = rem >> 2, ps = rem & 3;
        goff[j] = trow * pp.row_b + pc * PIX_B + ((ps ^ ((pc >> 2) & 3)) << 4);
    }
    const bool tail_ok = (G::NJ - 1) * 256 + tid < G::UNITS;
    int aoff[3][2];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
        const int pc = wc * 32 + l31 + dx;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            aoff[dx][ks] = wr * 4 * G::ROW_B + pc * PIX_B + (((ks * 2 + hi) ^ ((pc >> 2) & 3)) << 4);
    }
    const int woff = lane * 16;
    const long tile_off = (long)img * pp.img_b + (long)Y0 * pp.row_b;

    auto lds_addr = [](const char* p) { return (unsigned)(unsigned long long)(__attribute__((address_space(3))) const char*)p; };
    auto uni64 = [](unsigned long long m) {   // make uniformity visible to the compiler ("s" operands must be SGPRs)
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)m), hi32 = __builtin_amdgcn_readfirstlane((unsigned)(m >> 32));
        return ((unsigned long long)hi32 << 32) | lo;
    };
    const unsigned long long tail_mask = uni64(__builtin_amdgcn_ballot_w64(tail_ok));
    const unsigned long long w4_mask = uni64(wave < 2 ? ~0ull : 0ull);   // weight fragments 16, 17 of an 18-fragment chunk

    // ---- the LDS-DMA statement carries its own wait states: 5 between anything the compiler emitted in front of it and the
    // VMEM instruction (s_mov m0 + s_nop 3, or s_mov m0 + EXEC write + s_nop 2).  That covers "SALU writes M0 -> LDS-DMA" (1)
    // and "a VALU-written SGPR (v_readfirstlane, SGPR-spill reload) read as the scalar base" (5): hazard-proof by construction
    // wherever the register allocator reloads a base (it does: dropping the nops put a v_readlane 2 wait states in front of
    // a DMA, caught by tools/hazcheck.py).
    auto dma = [&](auto sc1_tag, const unsigned long long base, const unsigned voff, const unsigned lds_off) {
        if constexpr (decltype(sc1_tag)::value)
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 3\n\tglobal_load_lds_dwordx4 %1, %2 sc1" ::"s"(lds_off), "v"(voff), "s"(base) : "memory", "m0");
        else
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 3\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_off), "v"(voff), "s"(base) : "memory", "m0");
    };
    auto dma_masked = [&](auto sc1_tag, const unsigned long long base, const unsigned voff, const unsigned lds_off, const unsigned long long mask) {
        if constexpr (decltype(sc1_tag)::value)
            asm volatile("s_mov_b32 m0, %1\n\ts_mov_b64 exec, %0\n\ts_nop 2\n\tglobal_load_lds_dwordx4 %2, %3 sc1\n\ts_mov_b64 exec, -1"
                         ::"s"(mask), "s"(lds_off), "v"(voff), "s"(base) : "memory", "m0");
        else
            asm volatile("s_mov_b32 m0, %1\n\ts_mov_b64 exec, %0\n\ts_nop 2\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b64 exec, -1"
                         ::"s"(mask), "s"(lds_off), "v"(voff), "s"(base) : "memory", "m0");
    };
    using SC1 = std::true_type;
    using NOSC = std::false_type;
    constexpr int JORD[11] = {2, 3, 4, 5, 6, 7, 8, 0, 1, 9, 10};   // input DMA order: the two halo rows last
    static_assert(G::NJ == 11, "DMA order is written for 11 input instructions");
    // DMA statement d of a staging job (NIN input statements first, then NW weight statements)
    auto dma_item = [&](auto nin_tag, auto nw_tag, const int d, const unsigned long long ibase, const unsigned long long wbase,
                        const unsigned din_w, const unsigned dw_w) {
        constexpr int NIN = decltype(nin_tag)::value, NW = decltype(nw_tag)::value;
        if (d < NIN) {
            const int j = JORD[d];
            if (j < G::NJ - 1)
                dma(SC1{}, ibase, goff[j], din_w + j * 4096);
            else
                dma_masked(SC1{}, ibase, goff[j], din_w + j * 4096, uni64(tail_mask));   // (re-formed at the point of use: a long-lived 64-bit uniform may be parked in VGPRs)
        } else {
            const int k = d - NIN;
            if (NW == 5 && k == 4)
                dma_masked(NOSC{}, wbase + k * 4096, woff, dw_w + k * 4096, uni64(w4_mask));
            else
                dma(NOSC{}, wbase + k * 4096, woff, dw_w + k * 4096);
        }
    };
    // cold staging (kernel start and RDB seam: nothing could be prefetched)
    auto stage_cold = [&](const char* src, char* din, const char* wsrc, char* dw, auto nw_tag) {
        const unsigned long long ib = uni64((unsigned long long)src), wb = uni64((unsigned long long)(wsrc + wave * 1024));
        const unsigned din_w = __builtin_amdgcn_readfirstlane(lds_addr(din) + wave * 1024);
        const unsigned dw_w = __builtin_amdgcn_readfirstlane(lds_addr(dw) + wave * 1024);
        constexpr int NW = decltype(nw_tag)::value;
#pragma unroll
        for (int d = 0; d < 11 + NW; ++d) dma_item(std::integral_constant<int, 11>{}, nw_tag, d, ib, wb, din_w, dw_w);
    };

    // ---- neighbour progress (as ptrunk_kernel)
    int f_up = up < 0 ? 0x7fffffff : 0, f_dn = dn < 0 ? 0x7fffffff : 0;
    // per-wave progress check (f_up / f_dn are wave-uniform copies kept by every wave)
    int pq_up = 0, pq_dn = 0;
    const int inf_up = up < 0 ? 0x7fffffff : 0, inf_dn = dn < 0 ? 0x7fffffff : 0;
    auto poll_issue = [&]() {
        const int* qu = pp.prog + (up >= 0 ? up : t);
        const int* qd = pp.prog + (dn >= 0 ? dn : t);
        asm volatile("global_load_dword %0, %1, off sc1" : "=v"(pq_up) : "v"(qu) : "memory");
        asm volatile("global_load_dword %0, %1, off sc1" : "=v"(pq_dn) : "v"(qd) : "memory");
    };
    auto poll_spin = [&](const int need) {      // (rare: a neighbour is more than the poll's flight time behind)
        unsigned spins = 0;
        while (f_up < need || f_dn < need) {
            f_up = max(__builtin_amdgcn_readfirstlane(__hip_atomic_load(pp.prog + (up >= 0 ? up : t), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)), inf_up);
            f_dn = max(__builtin_amdgcn_readfirstlane(__hip_atomic_load(pp.prog + (dn >= 0 ? dn : t), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)), inf_dn);
            if (f_up >= need && f_dn >= need) break;
            __builtin_amdgcn_s_sleep(1);
            if (++spins > SPIN_LIMIT || __hip_atomic_load(pp.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                __hip_atomic_store(pp.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                f_up = f_dn = 0x7fffffff;       // carry on (garbage, reported): no wave of the launch waits again
                break;
            }
        }
    };
    auto poll_check = [&](auto k_tag, const int need) {   // k = VMEM instructions this wave issued behind poll_issue()
        constexpr int K = decltype(k_tag)::value;
        static_assert(K == 0 || K == 4 || K == 8 || K == 11 || K == 15, "vmcnt immediates of poll_check");
        if constexpr (K == 0) asm volatile("s_waitcnt vmcnt(0)" : "+v"(pq_up), "+v"(pq_dn)::"memory");
        else if constexpr (K == 4) asm volatile("s_waitcnt vmcnt(4)" : "+v"(pq_up), "+v"(pq_dn)::"memory");
        else if constexpr (K == 8) asm volatile("s_waitcnt vmcnt(8)" : "+v"(pq_up), "+v"(pq_dn)::"memory");
        else if constexpr (K == 11) asm volatile("s_waitcnt vmcnt(11)" : "+v"(pq_up), "+v"(pq_dn)::"memory");
        else asm volatile("s_waitcnt vmcnt(15)" : "+v"(pq_up), "+v"(pq_dn)::"memory");
        // (branch-free: a missing neighbour polled the workgroup's own word and is "infinitely far ahead")
        f_up = max(__builtin_amdgcn_readfirstlane(pq_up), inf_up);
        f_dn = max(__builtin_amdgcn_readfirstlane(pq_dn), inf_dn);
        if (min(f_up, f_dn) < need) poll_spin(need);
    };
    bool wt = true;
    {
        int my_xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(my_xcc));
        auto* word = (__attribute__((address_space(3))) int*)(smem);
        if (tid == 0) {
            __hip_atomic_store(pp.xcc + t, my_xcc + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int diff = pp.force_wt;
            for (int s = 0; s < 2 && !diff; ++s) {
                const int nb = s ? dn : up;
                if (nb < 0) continue;
                int v = 0;
                for (unsigned spins = 0; spins < SPIN_LIMIT; ++spins) {
                    v = __hip_atomic_load(pp.xcc + nb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (v) break;
                    __builtin_amdgcn_s_sleep(2);
                }
                if (v != my_xcc + 1) diff = 1;
            }
            *word = diff;
        }
        __syncthreads();
        wt = __builtin_amdgcn_readfirstlane(*word) != 0;
        __syncthreads();
    }
    auto publish = [&](int v) {
        if (tid == 0) {
            if (wt)
                __hip_atomic_store(pp.prog + t, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else
                __hip_atomic_store(pp.prog + t, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // reaches L2, no further
        }
    };
    bool pending_pub = false;
    int pub_val = 0;

    // ---- the RDB-level fp32 stream of this wave's 4 rows x 32 pixels x 64 channels: 128 registers per lane, conv5's
    // accumulator layout; loaded once from conv_first's pixel-order output
    floatx4 xres[2][4][4];
    const int X = wc * 32 + l31;
    {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int Y = Y0 + wr * 4 + i;
            const float* q = pp.xrr + (((long)img * pp.H + Y) * pp.W + X) * 64 + hi * 4;
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int g = 0; g < 4; ++g) xres[mb][i][g] = *(const floatx4*)(q + mb * 32 + g * 8);
        }
    }

    // ---- register-resident planes (S1).  The fp16 planes this workgroup re-reads most -- x's second chunk (5 reads per RDB) and
    // X1 (4 reads) -- stay in registers in MFMA B-fragment form (the very 16 bytes per lane the epilogues store), and are
    // staged into a ring slot with ds_write_b128 instead of being fetched back through L2 / MALL / HBM: per XCD the dense
    // planes of 32 workgroups (8 MB) overflow the 4 MB L2, so ~70 % of those re-reads were fabric traffic, and the steps
    // behind them ran at the landing time of their DMA, not at the matrix rate.  Only the two halo rows (the neighbours'
    // rows) still come by DMA.
    typedef unsigned uintx4 __attribute__((ext_vector_type(4)));
    uintx4 x1p[4][2], X1r[4][2], X2r[4][2];   // [row][k-step]: x chunk 1 (written by conv5's epilogue), X1 / X2 (conv1's / conv2's epilogue)
    const int soff = (wr * 4 + 1) * G::ROW_B + (X + 1) * PIX_B;                 // this lane's pixel record in a staged tile, row 0
    const int sswz[2] = {((0 + hi) ^ (((X + 1) >> 2) & 3)) << 4, ((2 + hi) ^ (((X + 1) >> 2) & 3)) << 4};
    // the zero border columns of the wave's 4 rows (a DMA-staged plane brings them along; a register-staged one must write them)
    // (lanes 0..19: tile rows 0..4 for the upper pair of waves, 5..9 for the lower one, i.e. the own rows plus one halo row)
    const int boff = (wr * 5 + (lane >> 2)) * G::ROW_B + (wc ? (G::COLS - 1) * PIX_B : 0) + (lane & 3) * 16;
    // halo-only DMA: the 64 real pixels of tile row 0 / row 9 are exactly one 256-lane statement each (4 KiB, lane-linear in
    // LDS behind the border pixel; the XOR swizzle is applied to the source address as everywhere)
    int hoff[2];
    {
        const int px = 1 + (tid >> 2), ps = tid & 3;
#pragma unroll
        for (int hrow = 0; hrow < 2; ++hrow)
            hoff[hrow] = (hrow ? G::ROWS - 1 : 0) * pp.row_b + px * PIX_B + ((ps ^ ((px >> 2) & 3)) << 4);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)      // x's second chunk as conv_first wrote it (fp16 plane 1 of dense[0])
#pragma unroll
        for (int m = 0; m < 2; ++m)
            x1p[i][m] = *(const uintx4*)(pp.dense[0] + tile_off + (long)pp.plane_b + (long)(wr * 4 + i + 1) * pp.row_b + (X + 1) * PIX_B + m * 32 + hi * 16);

    auto publish_pending = [&]() {
        if (pending_pub) {
            publish(pub_val);
            pending_pub = false;
        }
    };
    unsigned long long t_sync = 0, t_vm = 0;   // PROF: cycles in the top-of-step waits (own DMA landing / barrier)
    auto step_sync = [&]() {
        unsigned long long w0 = 0, w1 = 0;
        if (PROF) w0 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's LDS-DMA of the step has landed ...
        if (PROF) w1 = __builtin_amdgcn_s_memtime();
        __syncthreads();                                   // ... and everybody else's; all waves are past the previous step
        if (PROF) {
            t_vm += w1 - w0;
            t_sync += __builtin_amdgcn_s_memtime() - w1;
        }
    };

    half8 Pq[2][G::NP], Aq[2][3][2];   // operand fragments of two MFMA groups (kernel scope: group 0's are handed over a step boundary)
    // ---- one K step.  CB = cout/32 of the running layer.  What is staged for the NEXT step is a compile-time property:
    //   IN: 0 no input plane | 1 input plane by LDS-DMA (11 statements) | 2 input plane from registers (8 ds_write_b128 + 1
    //   border write + 2 halo DMA statements);   NW: weight DMA statements per wave (0 | 5 = 18 fragments | 9 = 36 fragments)
    auto run_step = [&](auto cb_tag, auto in_tag, auto nw_tag, floatx16 (&acc)[decltype(cb_tag)::value][4], const char* sbi, const char* sbw,
                        const char* nsrc, const char* nw, char* dst, const uintx4 (&rsrc)[4][2], auto flag_tag, const int need,
                        auto pre_tag, auto nxt_tag) {
        constexpr int CB = decltype(cb_tag)::value, IN = decltype(in_tag)::value, NW = decltype(nw_tag)::value;
        // FL: this step fetches rows of a neighbour (the plane it stages is new on them): polls at the top, the check in front of the
        // statements that carry those rows, which are then the LAST staging items of the step (need < 0: no check this time)
        constexpr bool FL = decltype(flag_tag)::value != 0;
        static_assert(!FL || IN != 0, "a flagged step stages an input plane");
        constexpr int NRD = G::NP + 3 * CB, NMF = 12 * CB;
        constexpr int NH = 2;                                           // halo DMA statements of a register-staged plane
        constexpr int NDMA = (IN == 1 ? 11 : IN == 2 ? NH : 0) + NW;    // DMA statements
        constexpr int ND = NDMA + (IN == 2 ? 9 : 0);                    // + LDS stores of a register-staged plane
        static_assert(ND <= 6 * 12, "the staging items of a step must fit its MFMA shadows");
        // LDS reads of a group in the order of their first use by the MFMA sequence below (m = dy-major, then row, then channel block):
        // r < NP: pixel-row fragment P[r]; r = NP + dy * CB + mb: weight fragment A[dy][mb]
        constexpr int ORD1[9] = {6, 0, 1, 2, 3, 7, 4, 8, 5};                   // CB = 1: A0 P0 P1 P2 P3 A1 P4 A2 P5
        constexpr int ORD2[12] = {6, 0, 7, 1, 2, 3, 8, 9, 4, 10, 11, 5};       // CB = 2: A00 P0 A01 P1 P2 P3 A10 A11 P4 A20 A21 P5
        static_assert(G::NP == 6, "read order tables are written for 6 pixel-row fragments");
        constexpr int RSTRIDE = NMF / NRD;                                      // shadows per read: 1 (CB = 1), 2 (CB = 2)
        const unsigned long long ibase = uni64((unsigned long long)nsrc);
        const unsigned long long wbase = uni64((unsigned long long)(nw + wave * 1024));
        const unsigned din_w = __builtin_amdgcn_readfirstlane(lds_addr(dst) + wave * 1024);
        const unsigned dw_w = din_w + IN_EX;
        constexpr bool PRE = decltype(pre_tag)::value != 0;    // group 0's operands were read by the previous step
        constexpr bool NXT = decltype(nxt_tag)::value != 0;    // this step holds the barrier and reads the next step's group 0 (same layer)
        constexpr int AT = P3_PRE_AT < NMF - NRD ? P3_PRE_AT : NMF - NRD;   // MFMA of the last group the barrier sits in front of
        static_assert(AT >= 0, "the next step's first reads must fit behind the barrier");
        half8 (&P)[2][G::NP] = Pq;
        half8 (&A)[2][3][2] = Aq;
        auto read_from = [&](const char* bi, const char* bw, const int g, const int r, const int set) {   // LDS read r (0..NRD-1) of group g
            const int ks = g / 3, dx = g - ks * 3;
            if (r < G::NP) {
                P[set][r] = *(const half8*)(bi + aoff[dx][ks] + r * G::ROW_B);
            } else {
                const int q = r - G::NP, dy = q / CB, mb = q - dy * CB;
                A[set][dy][mb] = *(const half8*)(bw + woff + ((((dy * 3 + dx) * 2 + ks) * CB + mb) << 10));
            }
        };
        auto read_item = [&](const int g, const int r, const int set) { read_from(sbi, sbw, g, r, set); };
        auto stage_item = [&](const int d) {
            // Flagged steps count the VMEM instructions between the polls and the check (poll_check's vmcnt immediate), so
            // every statement in that window must be one that EVERY wave issues: a statement whose EXEC mask is empty for a
            // wave (the 5th weight statement for waves 2, 3; the tail input statement) is dropped by the hardware without
            // touching vmcnt, the wave's count is then one short and the check reads the second poll before it has landed
            // (found by tools/soak_pipelined.py: rare stale halo rows when the trunk shares the chip).  The last weight
            // statement therefore goes behind the check, with the statements of the neighbours' rows.
            if constexpr (FL && IN == 2) {            // weights but the last, the own rows from registers, the border, [check], the last weights, the neighbours' rows
                if (d < NW - 1) {
                    dma_item(std::integral_constant<int, 0>{}, nw_tag, d, ibase, wbase, din_w, dw_w);
                } else if (d < NW + 7) {
                    const int e = d - (NW - 1), i = e >> 1, m = e & 1;
                    *(uintx4*)(dst + soff + sswz[m] + i * G::ROW_B) = rsrc[i][m];
                } else if (d == NW + 7) {
                    if (lane < 20) *(uintx4*)(dst + boff) = uintx4{0u, 0u, 0u, 0u};
                } else if (d == NW + 8) {
                    if (need >= 0) poll_check(std::integral_constant<int, NW - 1>{}, need);
                    dma_item(std::integral_constant<int, 0>{}, nw_tag, NW - 1, ibase, wbase, din_w, dw_w);
                } else {
                    const int h = d - NW - 9;
                    dma(SC1{}, ibase, hoff[h], din_w + (h ? G::ROWS - 1 : 0) * G::ROW_B + PIX_B);
                }
                return;
            }
            if constexpr (FL && IN == 1) {            // weights but the last, the statements of rows 1..8, [check], the last weights, the four statements holding rows 0 and 9
                if (d < NW - 1) {
                    dma_item(std::integral_constant<int, 11>{}, nw_tag, d + 11, ibase, wbase, din_w, dw_w);
                } else if (d < NW + 6) {
                    dma_item(std::integral_constant<int, 11>{}, nw_tag, d - (NW - 1), ibase, wbase, din_w, dw_w);
                } else if (d == NW + 6) {
                    if (need >= 0) poll_check(std::integral_constant<int, NW + 6>{}, need);
                    dma_item(std::integral_constant<int, 11>{}, nw_tag, NW - 1 + 11, ibase, wbase, din_w, dw_w);
                } else {
                    dma_item(std::integral_constant<int, 11>{}, nw_tag, d - NW, ibase, wbase, din_w, dw_w);
                }
                return;
            }
            if constexpr (IN == 2) {
                if (d < NH) {           // the neighbours' rows
                    dma(SC1{}, ibase, hoff[d], din_w + (d ? G::ROWS - 1 : 0) * G::ROW_B + PIX_B);
                } else if (d < NH + NW) {
                    dma_item(std::integral_constant<int, 0>{}, nw_tag, d - NH, ibase, wbase, din_w, dw_w);
                } else if (d < NH + NW + 8) {
                    const int e = d - NH - NW, i = e >> 1, m = e & 1;
                    *(uintx4*)(dst + soff + sswz[m] + i * G::ROW_B) = rsrc[i][m];
                } else {
                    if (lane < 20) *(uintx4*)(dst + boff) = uintx4{0u, 0u, 0u, 0u};
                }
            } else {
                dma_item(std::integral_constant<int, IN == 1 ? 11 : 0>{}, nw_tag, d, ibase, wbase, din_w, dw_w);
            }
        };
        if constexpr (FL) {
            if (need >= 0) poll_issue();          // (never a poll in flight that no wait covers: its late landing would hit a reallocated register)
        }
        if constexpr (!PRE) {
#pragma unroll
            for (int r = 0; r < NRD; ++r) read_item(0, CB == 1 ? ORD1[r % 9] : ORD2[r % 12], 0);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 6; ++g) {
#pragma unroll
            for (int m = 0; m < NMF; ++m) {
                const int dy = m / (4 * CB), rem = m - dy * 4 * CB, i = rem / CB, mb = rem - i * CB;
                if constexpr (NXT) {
                    if (g == 5 && m == AT) {             // the step barrier (see P3_PREREAD)
                        step_sync();
                        publish_pending();               // (the layer's first barrier carries the lazy publication of the previous layer's output)
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                if constexpr (BW)
                    acc[mb][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A[g & 1][dy][mb]), __builtin_bit_cast(bf16x8, P[g & 1][i + dy]), acc[mb][i], 0, 0, 0);
                else
                    acc[mb][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[g & 1][dy][mb], P[g & 1][i + dy], acc[mb][i], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                const int k = m / RSTRIDE;
                if (g + 1 < 6 && m % RSTRIDE == 0 && k < NRD) read_item(g + 1, CB == 1 ? ORD1[k % 9] : ORD2[k % 12], (g + 1) & 1);
                if constexpr (NXT) {                 // the next step's group 0 (it reads what this step staged: dst, dst + IN_EX), one read per shadow
                    const int k2 = m - AT;
                    if (g == 5 && k2 >= 0 && k2 < NRD) read_from(dst, dst + IN_EX, 0, CB == 1 ? ORD1[k2 % 9] : ORD2[k2 % 12], 0);
                }
                // staging: one item per shadow for a cout-32 group (12), every other shadow of conv5's (the ones without a read)
                const int sm = CB == 1 ? m : (m % 2 ? m / 2 : -1);
                if (sm >= 0) {
                    const int d = g * 12 + sm;
                    if (d < ND) stage_item(d);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    using C1 = std::integral_constant<int, 1>;
    using C2 = std::integral_constant<int, 2>;
    using I0 = std::integral_constant<int, 0>;   // IN: none / DMA / registers
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, P3_S1 ? 2 : 1>;
    using I2X = std::integral_constant<int, (P3_S1 && P3_X2REG) ? 2 : 1>;   // X2's staging
    using W0 = std::integral_constant<int, 0>;   // NW: weight DMA statements per wave
    using W5 = std::integral_constant<int, 5>;
    using W9 = std::integral_constant<int, 9>;
    using F0 = std::integral_constant<int, 0>;   // flag_tag: no neighbour rows in what the step stages / the step checks the neighbours' progress
    using F1 = std::integral_constant<int, 1>;
    using Q0 = std::integral_constant<int, 0>;   // pre_tag / nxt_tag
    using Q1 = std::integral_constant<int, 1>;

    // The next layer's bias is requested BEFORE the epilogue's stores (an asm load: the compiler's own wait for a load it tracks would be
    // vmcnt(0) -- it does not count the asm stores issued behind it), so that `vmcnt(NST)` in the prologue covers it together with the
    // step-0 DMA and leaves exactly the epilogue's NST stores in flight
    auto bias_request = [&](const float* bias, const int nb) {
        float v;
        const float* q = bias + (tid < nb ? tid : 0);
        asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(q) : "memory");
        return v;
    };
    // ---- layer prologue: wait for what step 0 reads, barrier, bias into LDS (the first layer of an RDB: everything has drained at the seam)
    auto prologue_cold = [&](const float* bias, const int nb, const int bias_lds) {
        float bias_v = 0.f;
        if (tid < nb) bias_v = bias[tid];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid < nb) ((float*)(smem + bias_lds))[tid] = bias_v;
        if (pending_pub) {
            publish(pub_val);
            pending_pub = false;
        }
    };
    auto prologue = [&](auto nst_tag, const float bias_v, const int nb, const int bias_lds) {
        constexpr int NST = decltype(nst_tag)::value;
        if constexpr (NST == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if constexpr (NST == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid < nb) ((float*)(smem + bias_lds))[tid] = bias_v;
    };
    // ---- epilogue of a cout-32 layer: bias, leaky ReLU, fp16, straight from the MFMA D layout
    auto epi32 = [&](floatx16 (&acc)[1][4], char* oplane, uintx4 (&keep)[4][2], const bool halo_only, const char* mplane = nullptr) {
        // BW: the saved forward plane's 16-byte records of this lane's four rows (same addresses as the stores below, in the mask buffer): all
        // eight loads go out in front of the arithmetic (the operand registers of the K loop are dead here)
        uintx4 mk[4][2];
        if constexpr (BW) {
#pragma unroll
            for (int io = 0; io < 4; ++io) {
                const int i = io == 0 ? 0 : io == 1 ? 3 : io - 1;
#pragma unroll
                for (int m = 0; m < 2; ++m)
                    mk[i][m] = *(const uintx4*)(mplane + (long)(Y0 + wr * 4 + i + 1) * pp.row_b + (X + 1) * PIX_B + m * 32 + hi * 16);
            }
        }
        floatx4 bias4[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) bias4[g] = *(const floatx4*)((const float*)(smem + A_BIAS_OFF) + g * 8 + hi * 4);
        // rows 0 and 3 first: one of them is the row a neighbour reads (its store is the one the publication waits for)
#pragma unroll
        for (int io = 0; io < 4; ++io) {
            const int i = io == 0 ? 0 : io == 1 ? 3 : io - 1;
            const int Y = Y0 + wr * 4 + i;
            const bool st = !halo_only || (i == 0 && wr == 0) || (i == 3 && wr == 1);   // (wave-uniform)
            unsigned hp[4][2];
            if constexpr (BW) {
                // the mask back in the accumulator's layout (permlane32_swap is its own inverse), then g *= (y > 0 ? 1 : 0.2) and bf16 (RNE)
                unsigned mq[4][2];
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    auto u0 = __builtin_amdgcn_permlane32_swap(mk[i][m][0], mk[i][m][2], false, false);
                    auto u1 = __builtin_amdgcn_permlane32_swap(mk[i][m][1], mk[i][m][3], false, false);
                    mq[2 * m][0] = u0[0];
                    mq[2 * m + 1][0] = u0[1];
                    mq[2 * m][1] = u1[0];
                    mq[2 * m + 1][1] = u1[1];
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const half4 mh = __builtin_bit_cast(half4, uint2{mq[g][0], mq[g][1]});
                    unsigned r[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float w = acc[0][i][g * 4 + q];
                        w = (float)mh[q] > 0.f ? w : w * 0.2f;
                        const unsigned u = __builtin_bit_cast(unsigned, w);
                        r[q] = (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
                    }
                    hp[g][0] = r[0] | (r[1] << 16);
                    hp[g][1] = r[2] | (r[3] << 16);
                }
            } else {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                floatx4 w;
#pragma unroll
                for (int q = 0; q < 4; ++q) w[q] = acc[0][i][g * 4 + q];
                w += bias4[g];
                const floatx4 ws = w * 0.2f;
#pragma unroll
                for (int q = 0; q < 4; ++q) asm("v_max_f32 %0, %1, %2" : "=v"(w[q]) : "v"(w[q]), "v"(ws[q]));
                half4 h4;
#pragma unroll
                for (int q = 0; q < 4; ++q) h4[q] = (_Float16)w[q];
                const uint2 u = __builtin_bit_cast(uint2, h4);
                hp[g][0] = u.x;
                hp[g][1] = u.y;
            }
            }
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                auto s0 = __builtin_amdgcn_permlane32_swap(hp[2 * m][0], hp[2 * m + 1][0], false, false);
                auto s1 = __builtin_amdgcn_permlane32_swap(hp[2 * m][1], hp[2 * m + 1][1], false, false);
                const uintx4 raw = {s0[0], s1[0], s0[1], s1[1]};
                keep[i][m] = raw;
                char* o = oplane + (long)(Y + 1) * pp.row_b + (X + 1) * PIX_B + m * 32 + hi * 16;
                if (st) {
                    if (wt)
                        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(o), "v"(raw) : "memory");
                    else
                        asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(o), "v"(raw) : "memory");
                }
            }
        }
    };
    // one row of x (channel block mb) -> 16-bit fragments, kept (mb == 1), written into the next RDB's resident plane (mb == 0, to_lds) and
    // stored where somebody reads them from memory
    auto x_row_out = [&](const int mb, const int i, char* obase, const bool st, const bool to_lds = false) {
        const int Y = Y0 + wr * 4 + i;
        unsigned hp[4][2];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if constexpr (BW) {       // bf16, RNE (as srbh_nhwc32_to_act16's bf16 form)
                unsigned r[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float xv = xres[mb][i][g][q];       // (a copy: __builtin_bit_cast applied to the vector ELEMENT read element 0 for every q)
                    const unsigned u = __builtin_bit_cast(unsigned, xv);
                    r[q] = (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
                }
                hp[g][0] = r[0] | (r[1] << 16);
                hp[g][1] = r[2] | (r[3] << 16);
            } else {
                half4 h4;
#pragma unroll
                for (int q = 0; q < 4; ++q) h4[q] = (_Float16)xres[mb][i][g][q];
                const uint2 u = __builtin_bit_cast(uint2, h4);
                hp[g][0] = u.x;
                hp[g][1] = u.y;
            }
        }
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            auto s0 = __builtin_amdgcn_permlane32_swap(hp[2 * m][0], hp[2 * m + 1][0], false, false);
            auto s1 = __builtin_amdgcn_permlane32_swap(hp[2 * m][1], hp[2 * m + 1][1], false, false);
            const uintx4 raw = {s0[0], s1[0], s0[1], s1[1]};
            if (mb == 1) x1p[i][m] = raw;
            if (P3_SEAM && mb == 0 && to_lds) *(uintx4*)(smem + soff + sswz[m] + i * G::ROW_B) = raw;      // the next RDB's resident plane, own rows
            char* o = obase + (long)mb * pp.plane_b + (long)(Y + 1) * pp.row_b + (X + 1) * PIX_B + m * 32 + hi * 16;
            if (st) {
                if (wt)
                    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(o), "v"(raw) : "memory");
                else
                    asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(o), "v"(raw) : "memory");
            }
        }
    };
    // ---- epilogue of conv5: x = 0.2 (acc + bias) + x in registers, fp16 copy out.  Every third RDB also closes an RRDB
    // (x = 0.2 x + x_rrdb, the RRDB-level stream lives in memory): that is a SECOND pass over the registers behind a uniform
    // branch -- as a flag inside one body hipcc if-converted it into 128 selects, as two bodies the 160 live registers met in
    // phis and spilled.  In an RRDB-closing RDB the first pass stores nothing (its fp16 values are not final).
    auto epi64 = [&](floatx16 (&acc)[2][4], char* obase, const bool r2, const bool r2_pixel, const bool x1_halo_only, const bool keep, const bool out_px) {
        floatx4 bias4[2][4];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int g = 0; g < 4; ++g) bias4[mb][g] = *(const floatx4*)((const float*)(smem + B_BIAS_OFF) + mb * 32 + g * 8 + hi * 4);
        const int frag_lane = wc * 2048 + lane * 4;   // fragment order: instruction (mb, g) owns 1 KiB, lane l its 16 B
        // The RRDB-level stream of an RRDB-closing RDB: its 8 loads per row are issued TWO ROWS AHEAD of their use (two register
        // slots of 32): rows 0 and 3 during the first pass -- they land under its arithmetic -- and rows 1, 2 as the slots free
        // up.  (Loading each row pair right before its use cost 17 k cycles per RRDB-closing epilogue against 3.6 k for a plain
        // one: two fully exposed memory latencies plus a wasted fp16 pass.  All four rows at once by LDS-DMA into the idle stage
        // was built and is slower -- the burst of all 256 workgroups is bandwidth-bound either way: profiles/r05an_ab_r2lds.txt.  Round 6: the
        // compiler's waits for these slots come out as vmcnt(1) / vmcnt(0) because the asm stores in between are invisible to it; asm loads with
        // manual counted waits (vmcnt(8) / (16)) shortened this epilogue by ~800 cycles and lengthened the seam by as much -- the stores have to
        // be acknowledged before the publication anyway: 3.727 vs 3.736 ms over four interleaved runs, profiles/r06s_*; not kept.)
        floatx4 a2[2][2][4];
        auto load_a2 = [&](const int slot, const int i) {
            const long rowb = ((long)img * pp.H + Y0 + wr * 4 + i) * pp.W * 64;
            const float* q2 = pp.xrr + rowb + (r2_pixel ? X * 64 + hi * 4 : frag_lane);
            const int sm = r2_pixel ? 32 : 1024, sg = r2_pixel ? 8 : 256;
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int g = 0; g < 4; ++g) a2[slot][mb][g] = *(const floatx4*)(q2 + mb * sm + g * sg);
        };
        // rows 0 and 3 first: one of them is the row a neighbour reads (its stores are what the publication waits for)
#pragma unroll
        for (int io = 0; io < 4; ++io) {
            const int i = io == 0 ? 0 : io == 1 ? 3 : io - 1;
            // (a slot's loads go out right behind the first pass of its row: the row's 32 accumulator registers are dead by then,
            //  so the prefetch costs no registers at the kernel's pressure peak; they land under the first pass of rows 1, 2)
            if (r2 && io == 1) load_a2(0, 0);
            if (r2 && io == 2) load_a2(1, 3);
            const bool halo = (i == 0 && wr == 0) || (i == 3 && wr == 1);   // (wave-uniform)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    floatx4 tt;
#pragma unroll
                    for (int q = 0; q < 4; ++q) tt[q] = acc[mb][i][g * 4 + q];
                    tt += bias4[mb][g];
                    xres[mb][i][g] = tt * 0.2f + xres[mb][i][g];
                }
                if (!r2) x_row_out(mb, i, obase, keep || (!P3_SEAM && mb == 0) || !x1_halo_only || halo, x1_halo_only);   // (RRDB-closing: the fp16 values are not final yet)
            }
        }
        if (r2) {
            auto close_row = [&](const int slot, const int i) {
                const bool halo = (i == 0 && wr == 0) || (i == 3 && wr == 1);
                const long rowb = ((long)img * pp.H + Y0 + wr * 4 + i) * pp.W * 64;
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) xres[mb][i][g] = xres[mb][i][g] * 0.2f + a2[slot][mb][g];
                    x_row_out(mb, i, obase, keep || (!P3_SEAM && mb == 0) || !x1_halo_only || halo, x1_halo_only);
                }
                // the RRDB-level stream goes back to memory (fragment order): private to this workgroup
                // (out_px: behind the LAST RDB of a training forward it is the trunk's output: to `xr`, pixel order = NHWC)
                const float* q = (out_px ? pp.xr : pp.xrr) + rowb;
                const unsigned vo = out_px ? (unsigned)(X * 64 + hi * 4) * 4u : (unsigned)frag_lane * 4u;
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const unsigned long long sb = uni64((unsigned long long)(q + (out_px ? mb * 32 + g * 8 : mb * 1024 + g * 256)));
                        asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2\n\ts_nop 1" ::"v"(vo), "v"(xres[mb][i][g]), "s"(sb) : "memory");
                    }
            };
            close_row(0, 0);
            load_a2(0, 1);
            close_row(1, 3);
            load_a2(1, 2);
            close_row(0, 1);
            close_row(1, 2);
        }
    };

    // ---- prologue of the launch: conv1 of RDB 0 reads conv_first's output (no flag needed)
    char* dcur = pp.dense[0] + tile_off;    // tile origin of plane 0 in the running RDB's dense buffer
    char* dnxt = pp.dense[1] + tile_off;
    const char* mcur = BW ? pp.mask + tile_off : nullptr;   // the saved forward planes of the RDB whose gradient is running (BW)
    // (P3_SEAM: conv1's step-0 weights live in phase-A stage 0's INPUT area -- step 0 reads the resident plane, not that area -- which is where
    //  conv5's last step can prefetch them at the seam)
    stage_cold(dcur, smem, pp.layers[0].w, smem + stage_off(1, 0) + (P3_SEAM ? 0 : IN_EX), W5{});
    const int nrdb = pp.nlayers / 5;
    for (int rdb = 0; rdb < nrdb; ++rdb) {
        const PLayer* T = pp.layers + rdb * 5;
        const int L0 = rdb * 5;
        int gs = 0;
        // ---------------- conv1..conv4 (cout 32, plane 0 resident, stages of IN_EX + 18 KiB).  One body per layer (kk is a compile-time
        // constant; conv1..3 and conv4 stage different things in their last steps: compiled as two variants of ONE layer body the
        // accumulators of the two last-step variants met in a phi, i.e. 64 v_accvgpr_mov per layer)
        float next_bias = 0.f;   // the next layer's bias element of this thread, requested ahead of the epilogue
        auto layerA = [&](auto kk_tag, auto last_nw_tag) {
            constexpr int kk = decltype(kk_tag)::value;
            const int L = L0 + kk, n = kk + 2;
            const char* wl = T[kk].w;
            unsigned long long p0 = 0, p1 = 0, p2 = 0;
            if (PROF) p0 = __builtin_amdgcn_s_memtime();
            if (P3_LAZYDRAIN && kk > 0) {
                // stores of the previous layer's epilogue still in flight per wave: X1 / X2 go out as one halo row (2 x 16 B), X3 whole (8)
                constexpr bool prev_halo_only = P3_SKIPST && P3_S1 && (kk - 1 == 0 || (P3_X2REG && kk - 1 == 1));
                prologue(std::integral_constant<int, prev_halo_only ? 2 : 8>{}, next_bias, 32, A_BIAS_OFF);
            } else {
                prologue_cold(T[kk].bias, 32, A_BIAS_OFF);
            }
            if (PROF) p1 = __builtin_amdgcn_s_memtime();
            t_sync = 0;
            t_vm = 0;
            floatx16 acc[1][4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[0][i][r] = 0.f;
            // step 0: resident plane 0; stages chunk 1 (x's second half) from registers.  Its barrier (inside its last MFMA group) is the
            // layer's FIRST: it carries the lazy publication of the previous layer's output (its vmcnt(0) covers the epilogue stores that
            // drained under step 0), in front of any neighbour check of this layer
            run_step(C1{}, I2{}, W5{}, acc, smem, smem + stage_off(1, gs & 1) + ((P3_SEAM && kk == 0) ? 0 : IN_EX), dcur + (long)pp.plane_b, wl + 18 * 1024,
                     smem + stage_off(1, (gs + 1) & 1), x1p, F0{}, 0, Q0{}, Q1{});
            ++gs;
            // The only NEW input plane of conv2..4 is the last chunk (index kk + 1): its halo rows are fetched during step kk, which checks
            // the neighbours' progress itself (conv1's inputs were verified at the seam).
            if (n >= 3) {      // step 1: chunk 1; stages chunk 2 (X1) from registers
                const char* st = smem + stage_off(1, gs & 1);
                run_step(C1{}, I2{}, W5{}, acc, st, st + IN_EX, dcur + 2l * pp.plane_b, wl + 2 * (18 * 1024),
                         smem + stage_off(1, (gs + 1) & 1), X1r, std::integral_constant<int, kk == 1>{}, L, Q1{}, Q1{});
                ++gs;
            }
            if (n >= 4) {      // step 2: chunk 2; stages chunk 3 (X2) from registers
                const char* st = smem + stage_off(1, gs & 1);
                run_step(C1{}, I2X{}, W5{}, acc, st, st + IN_EX, dcur + 3l * pp.plane_b, wl + 3 * (18 * 1024),
                         smem + stage_off(1, (gs + 1) & 1), X2r, std::integral_constant<int, kk == 2>{}, L, Q1{}, Q1{});
                ++gs;
            }
            if (n >= 5) {      // step 3 (conv4): chunk 3; stages chunk 4 (X3) by DMA
                const char* st = smem + stage_off(1, gs & 1);
                run_step(C1{}, I1{}, W5{}, acc, st, st + IN_EX, dcur + 4l * pp.plane_b, wl + 4 * (18 * 1024),
                         smem + stage_off(1, (gs + 1) & 1), x1p, F1{}, L, Q1{}, Q1{});
                ++gs;
            }
            {
                const char* st = smem + stage_off(1, gs & 1);
                // (the last step reads chunk kk + 1: x's second half, X1, X2 -- register-resident -- or X3)
                if constexpr (decltype(last_nw_tag)::value == 5)   // conv1..3: the next layer's step 0 reads the resident plane: weights only
                    run_step(C1{}, I0{}, W5{}, acc, st, st + IN_EX, nullptr, T[kk + 1].w, smem + stage_off(1, (gs + 1) & 1), x1p, F0{}, 0, Q1{}, Q0{});
                else              // conv4: conv5's chunk 0 IS the resident plane (phase-B stage 0 starts at the same address): 36 KiB of weights
                    run_step(C1{}, I0{}, W9{}, acc, st, st + IN_EX, nullptr, T[4].w, smem + stage_off(2, 0), x1p, F0{}, 0, Q1{}, Q0{});
                ++gs;
            }
            if (P3_LAZYDRAIN) next_bias = bias_request(T[kk + 1].bias, kk == 3 ? 64 : 32);   // (older than the epilogue's stores)
            if (PROF) p2 = __builtin_amdgcn_s_memtime();
            uintx4 kept[4][2];
            const bool halo_only = P3_SKIPST && P3_S1 && (kk == 0 || (P3_X2REG && kk == 1)) && !(P3_TRAIN && pp.keep_all);   // (training forward: whole planes)
            epi32(acc, dcur + (long)(2 + kk) * pp.plane_b - (long)Y0 * pp.row_b, kept, halo_only,
                  BW ? mcur + (long)(5 - kk) * pp.plane_b - (long)Y0 * pp.row_b : nullptr);
            if (kk == 0) {
#pragma unroll
                for (int io = 0; io < 4; ++io)
#pragma unroll
                    for (int m = 0; m < 2; ++m) X1r[io][m] = kept[io][m];
            }
            if (kk == 1) {
#pragma unroll
                for (int io = 0; io < 4; ++io)
#pragma unroll
                    for (int m = 0; m < 2; ++m) X2r[io][m] = kept[io][m];
            }
            pending_pub = true;   // published behind the next top-of-step barrier (whose vmcnt(0) covers these stores)
            pub_val = L + 1;
            if (PROF && tid == 0) {
                unsigned long long* q = pp.prof + ((long)blockIdx.x * pp.nlayers + L) * 6;
                q[0] = p1; q[1] = p2; q[2] = __builtin_amdgcn_s_memtime(); q[3] = p1 - p0; q[4] = t_sync; q[5] = t_vm;
            }
        };
        layerA(std::integral_constant<int, 0>{}, W5{});
        layerA(std::integral_constant<int, 1>{}, W5{});
        layerA(std::integral_constant<int, 2>{}, W5{});
        layerA(std::integral_constant<int, 3>{}, W9{});
        // ---------------- conv5 (cout 64, stages of IN_EX + 36 KiB, residual epilogue)
        {
            const int L = L0 + 4;
            const char* wl = T[4].w;
            unsigned long long p0 = 0, p1 = 0, p2 = 0, p3 = 0;
            if (PROF) p0 = __builtin_amdgcn_s_memtime();
            if (P3_LAZYDRAIN) {
                prologue(std::integral_constant<int, 8>{}, next_bias, 64, B_BIAS_OFF);   // (X4 went out whole: 8 stores per wave)
            } else {
                prologue_cold(T[4].bias, 64, B_BIAS_OFF);
            }
            if (PROF) p1 = __builtin_amdgcn_s_memtime();
            t_sync = 0;
            t_vm = 0;
            floatx16 acc[2][4];
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mb][i][r] = 0.f;
            {   // chunk 0 = the resident plane (in place: phase-B stage 0); stages chunk 1 from registers
                const char* st = smem + stage_off(2, 0);
                run_step(C2{}, I2{}, W9{}, acc, st, st + IN_EX, dcur + (long)pp.plane_b, wl + 36 * 1024, smem + stage_off(2, 1), x1p, F0{}, 0, Q0{}, Q1{});
            }
            {   // chunk 1; stages chunk 2 (X1) from registers
                const char* st = smem + stage_off(2, 1);
                run_step(C2{}, I2{}, W9{}, acc, st, st + IN_EX, dcur + 2l * pp.plane_b, wl + 2 * (36 * 1024), smem + stage_off(2, 0), X1r, F0{}, 0, Q1{}, Q1{});
            }
            {   // chunk 2; stages chunk 3 (X2) from registers
                const char* st = smem + stage_off(2, 0);
                run_step(C2{}, I2X{}, W9{}, acc, st, st + IN_EX, dcur + 3l * pp.plane_b, wl + 3 * (36 * 1024), smem + stage_off(2, 1), X2r, F0{}, 0, Q1{}, Q1{});
            }
            {   // chunks 3, 4; stage X3, X4 by DMA (X4 is conv4's output on the neighbours: checked by the step that fetches it).  ONE step body run
                // twice, as a do-while: the `for` form and two explicit bodies both cost 30 spilled VGPRs (and the two bodies 144 more MFMAs of code)
                int c = 3;
                do {
                    const char* st = smem + stage_off(2, c & 1);
                    run_step(C2{}, I1{}, W9{}, acc, st, st + IN_EX, dcur + (long)(c + 1) * pp.plane_b, wl + (long)(c + 1) * (36 * 1024),
                             smem + stage_off(2, (c + 1) & 1), x1p, F1{}, c == 4 ? L : -1, Q1{}, Q1{});
                } while (++c < 5);
            }
            {
                const char* st = smem + stage_off(2, 1);
                if constexpr (P3_SEAM) {
                    // the next conv1's first weight chunk -> smem + IN_EX (dst = smem: run_step puts weights at dst + IN_EX); the last RDB has no
                    // successor: it prefetches its own first chunk again (a select, not a branch: one step body), nobody reads it
                    const char* nw5 = rdb + 1 < nrdb ? T[5].w : T[0].w;
                    run_step(C2{}, I0{}, W5{}, acc, st, st + IN_EX, nullptr, nw5, smem, x1p, F0{}, 0, Q1{}, Q0{});
                } else {
                    run_step(C2{}, I0{}, W0{}, acc, st, st + IN_EX, nullptr, nullptr, smem, x1p, F0{}, 0, Q1{}, Q0{});
                }
            }
            const bool r2 = (rdb % 3) == 2;
            if (PROF) p2 = __builtin_amdgcn_s_memtime();
            epi64(acc, dnxt - (long)Y0 * pp.row_b, r2, rdb == 2, P3_SKIPST && P3_S1 && rdb + 1 < nrdb, P3_TRAIN && pp.keep_all != 0,
                  P3_TRAIN && pp.out_pixel != 0 && rdb + 1 == nrdb);   // (the last x goes out whole: conv_body reads it)
            if (PROF) p3 = __builtin_amdgcn_s_memtime();
            // RDB seam: the next conv1's first chunk is THIS layer's output on the neighbours: publish now, then wait for them
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            publish(L + 1);
            if (rdb + 1 < nrdb) {
                poll_issue();
                poll_check(std::integral_constant<int, 0>{}, L + 1);
                if constexpr (P3_SEAM) {
                    // own rows: written by the epilogue; weights: prefetched by the last step; left: the neighbours' two rows
                    const unsigned long long ib = uni64((unsigned long long)dnxt);
                    const unsigned din_w = __builtin_amdgcn_readfirstlane(lds_addr(smem) + wave * 1024);
                    dma(SC1{}, ib, hoff[0], din_w + PIX_B);
                    dma(SC1{}, ib, hoff[1], din_w + (G::ROWS - 1) * G::ROW_B + PIX_B);
                } else {
                    stage_cold(dnxt, smem, T[5].w, smem + stage_off(1, 0) + IN_EX, W5{});
                }
            }
            if (PROF && tid == 0) {
                unsigned long long* q = pp.prof + ((long)blockIdx.x * pp.nlayers + L) * 6;
                q[0] = p1; q[1] = p2; q[2] = p3; q[3] = (p1 - p0) | ((unsigned long long)(__builtin_amdgcn_s_memtime() - p3) << 32); q[4] = t_sync; q[5] = t_vm;
            }
            if constexpr (BW) mcur += pp.mask_stride;
            if (P3_TRAIN && pp.dense_stride) {      // (training forward: one dense buffer per RDB)
                dcur = dnxt;
                dnxt = dnxt + pp.dense_stride;
            } else {
                char* tmp = dcur;
                dcur = dnxt;
                dnxt = tmp;
            }
        }
    }
}
