// sz_nn_split.hip — split-precision tower: the reference-precision inference path on the matrix cores
// (/root/reference/network.py:36-83 BasicBlock, :105-137 stem/tower, :176-184 forward).
//
// network.py is fp32 end to end; bf16 MFMA operands keep 8 bits of mantissa.  Here every operand is carried as TWO bf16 numbers,
// x = hi + lo with hi = bf16(x), lo = bf16(x - hi) (16 bits of mantissa), and a product is three MFMAs with f32 accumulation:
//     w*x ~= w_hi*x_hi + w_lo*x_hi + w_hi*x_lo          (the dropped lo*lo term is 2^-16 relative)
// about 100x closer to the fp32 network than the bf16 tower, and it reproduces the fp32 network's SEARCH results (tests).
//
// Round-3 design (k_tower_split<WGB>), built around what bounded the round-2 kernel (one board per workgroup, three separate passes of the
// bf16 K loop: every weight fragment fetched once per 16 MFMAs = 64 B/clk/CU of L2 traffic, the most a CU gets):
//   * persistent, one workgroup (4 waves, one per SIMD) per CU takes a tile of WGB boards through all 39 convolutions;
//   * TWO boards per workgroup at large batches: activations live in LDS as ONE hi image and ONE lo image (2 x 70 KB) that every layer
//     overwrites in place — K loop, barrier, epilogue, barrier — which is what lets two boards fit; the residual x never goes through
//     LDS: every lane keeps the f32 values of its own accumulator elements in registers (128 VGPRs; exact f32, not hi + lo);
//   * the three products are FUSED per k-step: a wave loads its 4 w_hi and 4 w_lo fragments once and issues 96 MFMAs with them
//     (48 with one board), so the weight stream is 21 B/clk/CU and the LDS fragment reads are 16 per 96 MFMAs (the bf16 tower: 35 B/clk, 8 per 32);
//   * the weights of the whole tower are ONE stream in k-step order [conv][tap][k32]{hi 16 KB, lo 16 KB} (sz_nn_tower_split's `w_stream`):
//     a 2-slot register ring runs one k-step (1,536 matrix-pipe cycles) ahead straight across convolution and tile boundaries;
//   * position tile j = board row j of both boards (tile_row), so the border-row tiles under the dy = -1 / +1 taps are skipped (8.3 % of the MFMAs);
//   * every memory instruction sits alone in an MFMA gap, pinned with sched_barrier (as in k_tower16_bf16).
// WGB = 1 (n_boards <= #CUs: search of a few positions, the tail of a self-play run) runs the same code on one board per workgroup; every output
// element is accumulated in the same order in both forms (taps, k-steps, hi*hi / lo*hi / hi*lo), so a board's result does not depend on the batch size.
#include "sz_nn_common.h"

#define NN_MAX_CONVS_SPLIT 129                             // the k-step offset of the weight stream is a 32-bit byte offset: 36 + 72*128 k-steps x 32 KiB
#ifndef SP_PF
#define SP_PF 1                                            // weight prefetch distance in k-steps (1,536 matrix-pipe cycles each with two boards).  -DSP_PF=2 (measured, profiles/r03q_pf.txt):
                                                           // K loop unchanged (108.3k vs 108.4k cycles: the loop does not wait for weights), 30 registers more spilled around the epilogues
#endif
#define SP_WSCALE_LOG2 10                                   // f16 operands: the weights are packed times 2^10 (exact), results scaled back
#ifndef SP_EXPLICIT_WAIT
#define SP_EXPLICIT_WAIT 1
#endif
#define SP_RING (SP_PF == 1 ? 2 : 4)                       // named ring slots (compile-time indices ks & (SP_RING-1)); SP_PF + 1 of them are live at any time
#define SP_KSTEP_U4 2048                                   // uint4 per k-step of the weight stream: 16 co tiles x 64 lanes hi, then the same for lo

template <int WGB> struct SplitGeom {
    static constexpr int PITCH = NN_COUT * 2 + NN_PAD16;                      // 544 B per image row (position): conflict-free ds_read_b128
    static constexpr int IMG = WGB * 64 * PITCH + NN_ZERO16;                  // one image incl. its zero region
    static constexpr int NJ = 4 * WGB;                                        // position tiles (16) per workgroup
    static constexpr int TAB = 2 * IMG;                                       // byte offset of the tap address table [9][NJ][64] int
    static constexpr int SCR = 2 * IMG + 9 * NJ * 64 * 4;                     // byte offset of the heads' scratch: [WGB][64] position sums, [4 waves][2] maxima
    static constexpr int LDS_BYTES = SCR + 1024;
    static constexpr int PITCH_IN = 128 * 2 + NN_PAD16;                       // the stem's input image (128 channels), staged where the lo image lives
};

// One convolution's K loop on hi/lo operands.  B images: hi at byte offset offH of `lds`, lo at offL (BLO = false: the input is exact in bf16 —
// the 0/1 planes of the stem — and only w_hi*x + w_lo*x are formed).  Weight stream: global k-steps ks_base .. ks_base + 9*CIN/32 - 1; on entry
// the ring holds k-steps ks_base .. ks_base + SP_PF - 1; on exit it holds k-steps ks_after .. ks_after + SP_PF - 1 (the next convolution's first ones, or the
// stem's of the next tile), fetched under this convolution's last k-steps.
template <int CIN, int WGB, bool BLO, bool TAB, int ABL = 0 /* timing ablation (diagnostic build): 1 = no weight loads, 2 = no LDS fragment reads in the loop */,
          int NTAPS = 9 /* 1: a 1x1 convolution on the same images (conv_p1 of the fused heads) */,
          class E = ElemBF16 /* element of the hi / lo operands: bf16 (inference), or f16 with power-of-two operand scaling (training convolutions) */,
          int NI = 4 /* channel tiles (16) per wave: 4 = the workgroup covers all 256 output channels; 2 = half of them (training convolution at small batches) */>
__device__ __forceinline__ void split_kloop(const unsigned char* lds, const int offH, const int offL, const int* addr_tab, const WSrc& wr,
                                            const uint32_t ks_base, const uint32_t ks_after, const float* __restrict__ bias,
                                            f32x4 (&acc)[NI][4 * WGB], uint4 (&ring)[SP_RING][2 * NI], const int co_tile0 = -1 /* first channel tile of this wave; default wave*NI */) {
    constexpr int PITCH = CIN * 2 + NN_PAD16;
    constexpr int KSTEPS = CIN / 32;
    constexpr int NJ = 4 * WGB, G = WGB;                   // position tiles; groups of 4 tiles per k-step
    constexpr int NPROD = BLO ? 3 : 2;
    constexpr bool SKIPROWS = WGB == 2 && NN_ROWSKIP && NTAPS == 9;
    static_assert(NTAPS == 9 || NTAPS == 1, "3x3 or 1x1");
    static_assert(KSTEPS % SP_RING == 0, "ring slots must be compile-time indices");
    static_assert(8 + 2 * NI <= NPROD * NI * 3, "every memory instruction needs an MFMA gap of its own");
    int lane = threadIdx.x & 63;
    asm volatile("" : "+v"(lane));                         // opaque per call: hipcc otherwise hoists the stem's 72 tap addresses out of the tile loop and spills them
    const int wave = threadIdx.x >> 6;
    const int p16 = lane & 15, kg = lane >> 4;
    const int ct0 = co_tile0 < 0 ? wave * NI : co_tile0;
    const uint32_t wlane = (uint32_t)(ct0 * 64 + lane) * 16u;
    {
        f32x4 binit[NI];
#pragma unroll
        for (int i = 0; i < NI; i++) binit[i] = *(const f32x4*)(bias + (ct0 + i) * 16 + 4 * kg);
#pragma unroll
        for (int i = 0; i < NI; i++)
#pragma unroll
            for (int j = 0; j < NJ; j++) acc[i][j] = binit[i];
    }
    // The stem's input image (C_in = 128, row pitch 288 B) is staged with the same chunk swizzle as the 256-channel images, so an entry of their tap table
    // converts to it without coordinate arithmetic: valid entry row*544 + 16*chunk -> entry - 256*floor(entry / 544); zero-region entry: the same slot behind
    // the 288-B rows (the arithmetic form cost ~50 VALU per tile and tap: the stem K loop ran at half speed)
    constexpr int TABPITCH = NN_COUT * 2 + NN_PAD16;
    auto tap_addr = [&](int tap, int j) -> int {
        if constexpr (TAB) {
            const int rel = addr_tab[((NTAPS == 1 ? 4 : tap) * NJ + j) * 64 + lane];             // 1x1: the centre tap of the 3x3 table
            if constexpr (PITCH == TABPITCH) return rel;
            else {
                const int row = (int)__umulhi((unsigned)rel, 7895161u);                            // floor(rel / 544), exact for rel < 2^20
                return rel >= WGB * 64 * TABPITCH ? rel - WGB * 64 * (TABPITCH - PITCH) : rel - row * (TABPITCH - PITCH);
            }
        } else return conv_tap_addr16<PITCH, NTAPS, WGB>(tap, j, p16, kg);
    };
    const int lds_base = (int)(uint32_t)(uintptr_t)lds;
    auto abs_addr = [&](int rel) -> int {
        int a = lds_base + offH + rel;
        asm volatile("" : "+v"(a));                        // opaque: the sum stays in the VGPR, the k offset rides in the ds_read immediate
        return a;
    };
    auto LD = [](int addr) -> bf16x8 { return *(const __attribute__((address_space(3))) bf16x8*)(uint32_t)addr; };
    int bcurH[NJ], bcurL[BLO ? NJ : 1], bnxt[NJ];
    bf16x8 bH[2][4], bL[4];
#pragma unroll
    for (int j = 0; j < NJ; j++) {
        bnxt[j] = tap_addr(0, j);
        bcurH[j] = abs_addr(bnxt[j]);
        if constexpr (BLO) bcurL[j] = bcurH[j] + (offL - offH);
    }
#pragma unroll
    for (int m = 0; m < 4; m++)
        if (!(SKIPROWS && m == 0)) bH[0][m] = LD(bcurH[m]);                  // tap 0 looks one row up: tile 0 is idle
    // SK: 1 = position tile 0 idle under this tap (dy = -1), 2 = the last position tile idle (dy = +1)
    auto tap_body = [&](const int tap, auto skip_tag) {
        constexpr int SK = decltype(skip_tag)::value;
        if (tap + 1 < NTAPS) {
#pragma unroll
            for (int j = 0; j < NJ; j++) bnxt[j] = tap_addr(tap + 1, j);
        }
#pragma unroll
        for (int kc = 0; kc < KSTEPS; kc++) {
            const int slot = kc & (SP_RING - 1);           // ks_base and tap*KSTEPS are multiples of SP_RING
            // the k-step SP_PF ahead: this convolution's, or (from its last SP_PF k-steps) the first ones of what follows it in the stream
            const uint32_t ks_next = (kc + SP_PF >= KSTEPS && tap == NTAPS - 1) ? ks_after + (uint32_t)(kc + SP_PF - KSTEPS) : ks_base + (uint32_t)(tap * KSTEPS + kc + SP_PF);
#pragma unroll
            for (int g = 0; g < G; g++) {
                const int buf = (kc * G + g) & 1;
#pragma unroll
                for (int prod = 0; prod < NPROD; prod++) {
#pragma unroll
                    for (int i = 0; i < NI; i++) {
                        const bf16x8 a = __builtin_bit_cast(bf16x8, ring[slot][prod == 1 ? NI + i : i]);
#pragma unroll
                        for (int m = 0; m < 4; m++) {
                            const bool idle_g = SKIPROWS && ((SK == 1 && g == 0) || (SK == 2 && g == G - 1));     // this group has an idle position tile (border row)
                            const bool idle = idle_g && m == (SK == 1 ? 0 : 3);
                            if (idle) continue;
                            acc[i][g * 4 + m] = E::mfma(a, prod == 2 ? bL[m] : bH[buf][m], acc[i][g * 4 + m]);
                            // one memory instruction per MFMA gap; q counts the MFMAs actually issued in this group (a skipped tile has no gap of its own)
                            const int NR = idle_g ? 3 : 4, q = (prod * NI + i) * NR + (idle_g && SK == 1 ? m - 1 : m);
                            if (q < 4) {
                                // lo fragments of THIS group (used by its third product, 32 gaps on)
                                if constexpr (BLO && !(ABL & 2)) {
                                    const bool idl = SKIPROWS && ((SK == 1 && g == 0 && q == 0) || (SK == 2 && g == G - 1 && q == 3));
                                    if (!idl) bL[q] = LD(bcurL[g * 4 + q] + kc * 64);
                                }
                            } else if (q < 8) {
                                // hi fragments of the NEXT group
                                const int mm = q - 4;
                                if (ABL & 2) {
                                } else if (g + 1 < G) {
                                    if (!(SKIPROWS && SK == 2 && g + 1 == G - 1 && mm == 3)) bH[buf ^ 1][mm] = LD(bcurH[(g + 1) * 4 + mm] + kc * 64);
                                } else if (kc + 1 < KSTEPS) {
                                    if (!(SKIPROWS && SK == 1 && mm == 0)) bH[buf ^ 1][mm] = LD(bcurH[mm] + (kc + 1) * 64);
                                } else if (tap + 1 < NTAPS) {
                                    if (!(SKIPROWS && mm == 0 && tap + 1 < 3)) bH[buf ^ 1][mm] = LD(abs_addr(bnxt[mm]));
                                }
                            } else if (q < 8 + 2 * NI && g == 0) {
                                // weights of the next k-step (this convolution's, the next convolution's, or the next tile's stem)
                                const int f = q - 8;
                                if (!(ABL & 1)) ring[(kc + SP_PF) & (SP_RING - 1)][f] = ld_wfrag(wr, (size_t)ks_next * SP_KSTEP_U4 + (f >= NI ? SP_KSTEP_U4 / 2 : 0), wlane + (f % NI) * 1024);
                            }
#if SP_EXPLICIT_WAIT
                            // one explicit wait in a gap that carries no memory instruction, instead of hipcc's counted wait in front of every first use
                            // (sz_nn.hip NN_EXPLICIT_WAIT): before the third product for the lo fragments, at the end of a group for the next group's hi
                            // fragments, at the end of a k-step also for the next k-step's weights (SP_PF - 1 k-steps of 8 loads stay in flight)
                            if (q == 2 * NI * NR - 1 && NPROD == 3) __builtin_amdgcn_s_waitcnt(0xC07F);                          // lgkmcnt(0)
                            if (q == NPROD * NI * NR - 1) {
                                if (g == G - 1) __builtin_amdgcn_s_waitcnt(0x0070 | ((SP_PF - 1) * 2 * NI));                       // vmcnt(2*NI * (SP_PF - 1)) lgkmcnt(0)
                                else __builtin_amdgcn_s_waitcnt(0xC07F);                                                         // lgkmcnt(0)
                            }
#endif
                            asm volatile("" ::: "memory");
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
            }
        }
        if (NTAPS > 1) {
#pragma unroll
            for (int j = 0; j < NJ; j++) {
                bcurH[j] = abs_addr(bnxt[j]);
                if constexpr (BLO) bcurL[j] = bcurH[j] + (offL - offH);
            }
        }
    };
    using SK0 = std::integral_constant<int, 0>;
    if constexpr (SKIPROWS) {
        using SK1 = std::integral_constant<int, 1>; using SK2 = std::integral_constant<int, 2>;
        for (int tap = 0; tap < 3; tap++) tap_body(tap, SK1{});
        for (int tap = 3; tap < 6; tap++) tap_body(tap, SK0{});
        for (int tap = 6; tap < 9; tap++) tap_body(tap, SK2{});
    } else {
        for (int tap = 0; tap < NTAPS; tap++) tap_body(tap, SK0{});
    }
}

// The residual lives in the ACCUMULATION half of the register file between the epilogues (gfx950: 256 arch VGPRs + 256 AGPRs per lane at one wave per
// SIMD).  Left to itself hipcc kept the accumulators and the K loop's operands in the arch VGPRs and sent most of the 128 residual values to scratch
// memory: the conv2 epilogue was a chain of scratch_load -> s_waitcnt vmcnt(0), 35k cycles per convolution (in-kernel stamps, profiles/r03b_split_stamps.txt).
// A value written by v_accvgpr_write has an AGPR-class live range, so it stays there across the K loop.
__device__ __forceinline__ float to_agpr(float v) { float a; asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(a) : "v"(v)); return a; }
__device__ __forceinline__ float from_agpr(float a) { float v; asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(a)); return v; }
// ReLU on the f32 bit pattern: a negative float is a negative int32 (v_max_i32: one instruction, no canonicalisation; -0 -> +0)
__device__ __forceinline__ float relu_f32(float v) { const int i = __builtin_bit_cast(int, v); return __builtin_bit_cast(float, i > 0 ? i : 0); }

// Epilogue of one convolution: every lane turns its own accumulator elements (4 channels of one position per tile) into the hi / lo images.
//   MODE 0 (stem)  : x = relu(acc)         -> xres, images
//   MODE 1 (conv1) : t = relu(acc)         -> images            (network.py:70-72)
//   MODE 2 (conv2) : x = relu(acc + xres)  -> xres, images      (network.py:74-81; the residual is the exact f32 value, kept in registers)
// E = ElemF16: the weights (and the bias the accumulators started at) were packed times 2^SP_WSCALE_LOG2: the accumulator is scaled back first (exact)
template <int WGB, int MODE, class E = ElemBF16>
__device__ __forceinline__ void split_epilogue(unsigned char* hi_img, unsigned char* lo_img, const f32x4 (&acc)[4][4 * WGB], float (&xres)[4][4 * WGB][4]) {
    constexpr bool SCALED = std::is_same<E, ElemF16>::value;
    constexpr float USC = 1.0f / (float)(1 << SP_WSCALE_LOG2);
    constexpr int PITCH = NN_COUT * 2 + NN_PAD16;
    int lane = threadIdx.x & 63;
    asm volatile("" : "+v"(lane));                         // opaque per call: the store addresses are not hoisted out of the tile loop (and spilled)
    const int wave = threadIdx.x >> 6;
    const int p16 = lane & 15, kg = lane >> 4;
#pragma unroll
    for (int j = 0; j < 4 * WGB; j++) {
        const int row = tile_row<WGB>(j, p16);
        const int rowoff = row * PITCH, g = (row >> 2) & 1;          // g: the images' chunk swizzle (conv_tap_addr16 SWZ)
#pragma unroll
        for (int ip = 0; ip < 4; ip += 2) {
            uint2 h[2], l[2];
            // the AGPR moves are volatile asm (they keep their program order): all eight reads first, then the arithmetic, then all eight writes, so that
            // no instruction waits for the one in front of it (value by value the chain read -> add -> max -> write ran at 6.8 cycles per instruction)
            float r[2][4];
            f32x4 v[2];
#pragma unroll
            for (int d = 0; d < 2; d++)
#pragma unroll
                for (int k = 0; k < 4; k++) r[d][k] = MODE == 2 ? from_agpr(xres[ip + d][j][k]) : 0.f;
#pragma unroll
            for (int d = 0; d < 2; d++) {
                v[d] = acc[ip + d][j];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    if (SCALED) v[d][k] *= USC;
                    if (MODE == 2) v[d][k] += r[d][k];
                    v[d][k] = relu_f32(v[d][k]);
                }
            }
#pragma unroll
            for (int d = 0; d < 2; d++)
#pragma unroll
                for (int k = 0; k < 4; k++)
                    if (MODE != 1) xres[ip + d][j][k] = to_agpr(v[d][k]);
#pragma unroll
            for (int d = 0; d < 2; d++) {
                h[d].x = E::pack2(v[d][0], v[d][1]); h[d].y = E::pack2(v[d][2], v[d][3]);
                l[d].x = E::pack2(v[d][0] - E::lo(h[d].x), v[d][1] - E::hi(h[d].x));
                l[d].y = E::pack2(v[d][2] - E::lo(h[d].y), v[d][3] - E::hi(h[d].y));
            }
            // The lane holds 4 channels (8 B) of tile ip and 4 of tile ip + 1; its partner in the neighbouring lane quarter (lane ^ 16) holds the adjacent 4 of each.
            // v_permlane16_swap trades them so that an even quarter ends up with 8 consecutive channels of tile ip and an odd one with 8 of tile ip + 1: one
            // 16-byte store per lane and image instead of two 8-byte ones, and (with the chunk swizzle) no bank conflict: the 8-byte stores were 4-way
            // conflicted and made the epilogue LDS-bound (4,096 LDS cycles per convolution and CU against 4.5-5.9k exposed; profiles/r03p: 2.45e8 conflict cycles).
            const auto hx = __builtin_amdgcn_permlane16_swap(h[0].x, h[1].x, false, false), hy = __builtin_amdgcn_permlane16_swap(h[0].y, h[1].y, false, false);
            const auto lx = __builtin_amdgcn_permlane16_swap(l[0].x, l[1].x, false, false), ly = __builtin_amdgcn_permlane16_swap(l[0].y, l[1].y, false, false);
            const int it = ip + (kg & 1);                                        // the tile this lane now holds 8 channels of
            const int chunk = ((wave * 4 + it) * 16 + 4 * (kg & 2)) >> 3;        // their 16-byte chunk inside the row
            const int off = rowoff + ((chunk ^ g) << 4);
            *(uint4*)(hi_img + off) = make_uint4(hx[0], hy[0], hx[1], hy[1]);
            *(uint4*)(lo_img + off) = make_uint4(lx[0], ly[0], lx[1], ly[1]);
        }
    }
}

// ---- fused heads (network.py:141-174) on the tile while x is still in LDS as hi / lo images -------------------------------------------------
struct SplitHeadsParams {
    const uint4* w_p2;       // conv_p2 (73 -> 80 channels), sz_nn_pack_split_head: [8 k32-steps]{hi: 5 co tiles x 64 lanes, lo: the same} uint4
    const float* b_p2;       // [73]
    const float* wv;         // [256] conv_v1 with v_norm folded
    float bv;
    float* probs;            // [n_boards][4672] f32, the reference's flatten order plane*64 + position
    float* v1_out;           // [n_boards][64] relu(bn(conv_v1(x))) (the 64 -> 256 -> 1 MLP is k_value_head)
    int do_softmax;
};

// value : conv_v1 (256 -> 1) + ReLU per position, one f32 fma chain over the channels in ascending order
// policy: t = relu(bn(conv_p1(x))) on the tower's K loop (1x1) -> hi / lo images over x -> conv_p2 (three products again) + bias -> softmax over the
//         board's 4672 logits.
// Every sum runs in an order that does not depend on the workgroup form: per position over (channel tile, register) in the lane, the two shuffles over
// the lane's channel quarter, then the 64 positions of the board one after the other from LDS.  So a board's probabilities are the same bit for bit
// whether it ran alone in a workgroup, beside another board, at batch 1 or 4096 (the self-play records of a game do not depend on the batch it ran in).
template <int WGB, class E = ElemBF16>
__device__ __forceinline__ void split_heads_tail(unsigned char* lds, const int* addr_tab, const WSrc& wr, const uint32_t ks_p1, const float* __restrict__ bias_p1,
                                                 const SplitHeadsParams& hp, f32x4 (&acc)[4][4 * WGB], uint4 (&ring)[SP_RING][8], float (&xres)[4][4 * WGB][4],
                                                 const int board0, const int n_boards) {
    using GEO = SplitGeom<WGB>;
    constexpr int NJ = GEO::NJ, NT = WGB, PITCH = GEO::PITCH;          // NT: position tiles per wave in conv_p2
    unsigned char* imgH = lds;
    unsigned char* imgL = lds + GEO::IMG;
    float* psum = (float*)(lds + GEO::SCR);                            // [WGB][64]
    float* red = psum + WGB * 64;                                      // [4 waves][2 boards]
    int lane = threadIdx.x & 63;
    asm volatile("" : "+v"(lane));
    const int wave = threadIdx.x >> 6, p16 = lane & 15, kg = lane >> 4;
    {   // value conv: one image row (board*64 + position) per thread
        const int row = wave * 64 + lane;
        if (row < WGB * 64) {
            const unsigned char* ph = imgH + row * PITCH;
            const unsigned char* pl = imgL + row * PITCH;
            const int g = (row >> 2) & 1;                  // the images' chunk swizzle
            float sv = 0.f;
#pragma unroll 4
            for (int c8 = 0; c8 < 32; c8++) {
                const uint4 h = *(const uint4*)(ph + ((c8 ^ g) << 4)), l = *(const uint4*)(pl + ((c8 ^ g) << 4));
                const float4 w0 = ((const float4*)hp.wv)[c8 * 2], w1 = ((const float4*)hp.wv)[c8 * 2 + 1];
                sv = __builtin_fmaf(w0.x, E::lo(h.x) + E::lo(l.x), sv); sv = __builtin_fmaf(w0.y, E::hi(h.x) + E::hi(l.x), sv);
                sv = __builtin_fmaf(w0.z, E::lo(h.y) + E::lo(l.y), sv); sv = __builtin_fmaf(w0.w, E::hi(h.y) + E::hi(l.y), sv);
                sv = __builtin_fmaf(w1.x, E::lo(h.z) + E::lo(l.z), sv); sv = __builtin_fmaf(w1.y, E::hi(h.z) + E::hi(l.z), sv);
                sv = __builtin_fmaf(w1.z, E::lo(h.w) + E::lo(l.w), sv); sv = __builtin_fmaf(w1.w, E::hi(h.w) + E::hi(l.w), sv);
            }
            const int board = board0 + (row >> 6);
            if (board < n_boards) hp.v1_out[(size_t)board * 64 + (row & 63)] = fmaxf(sv + hp.bv, 0.f);
        }
    }
    split_kloop<256, WGB, true, true, 0, 1, E>(lds, 0, GEO::IMG, addr_tab, wr, ks_p1, 0u, bias_p1, acc, ring);     // fetches the next tile's first stem k-step on its way out
    __syncthreads();                                                   // every wave is done reading x
    split_epilogue<WGB, 1, E>(imgH, imgL, acc, xres);                  // t over x
    __syncthreads();
    // conv_p2: wave w owns position tiles w*NT .. w*NT + NT - 1, all 5 channel tiles; K = 256
    f32x4 pa[5][NT];
#pragma unroll
    for (int i = 0; i < 5; i++)
#pragma unroll
        for (int t = 0; t < NT; t++) pa[i][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    int baddr[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) baddr[t] = addr_tab[(4 * NJ + wave * NT + t) * 64 + lane];       // centre tap: the lane's own row of tile wave*NT + t, + 16*kg
#pragma unroll 2
    for (int kc = 0; kc < 8; kc++) {
        bf16x8 ah[5], al[5], bh[NT], bl[NT];
#pragma unroll
        for (int i = 0; i < 5; i++) {
            ah[i] = __builtin_bit_cast(bf16x8, hp.w_p2[(size_t)(kc * 10 + i) * 64 + lane]);
            al[i] = __builtin_bit_cast(bf16x8, hp.w_p2[(size_t)(kc * 10 + 5 + i) * 64 + lane]);
        }
#pragma unroll
        for (int t = 0; t < NT; t++) { bh[t] = *(const bf16x8*)(imgH + baddr[t] + kc * 64); bl[t] = *(const bf16x8*)(imgL + baddr[t] + kc * 64); }
#pragma unroll
        for (int prod = 0; prod < 3; prod++)
#pragma unroll
            for (int i = 0; i < 5; i++)
#pragma unroll
                for (int t = 0; t < NT; t++)
                    pa[i][t] = E::mfma(prod == 1 ? al[i] : ah[i], prod == 2 ? bl[t] : bh[t], pa[i][t]);
    }
    __syncthreads();                                                   // every wave is done reading t: the next tile may stage its planes
    // lane holds the logits of positions tile_row(wave*NT + t, p16), channels i*16 + 4*kg + r
    float mx = -3.0e38f;
#pragma unroll
    for (int i = 0; i < 5; i++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int co = i * 16 + 4 * kg + r;
            const float bb = co < 73 ? hp.b_p2[co] : 0.f;
#pragma unroll
            for (int t = 0; t < NT; t++) {
                if (std::is_same<E, ElemF16>::value) pa[i][t][r] *= 1.0f / (float)(1 << SP_WSCALE_LOG2);       // conv_p2's weights were packed times 2^SP_WSCALE_LOG2
                pa[i][t][r] += bb;
                if (co < 73) mx = fmaxf(mx, pa[i][t][r]);
            }
        }
    const int lboard = WGB == 2 ? (p16 >> 3) : 0;                      // the lane's board inside the workgroup (tile_row: lanes 8..15 of a tile are board 1)
    float inv = 1.0f;
    if (hp.do_softmax) {                                               // uniform across the workgroup
        // maximum over the board (exact in any order): lanes of the same board, then the four waves
        mx = fmaxf(mx, __shfl_xor(mx, 1)); mx = fmaxf(mx, __shfl_xor(mx, 2)); mx = fmaxf(mx, __shfl_xor(mx, 4));
        if (WGB == 1) mx = fmaxf(mx, __shfl_xor(mx, 8));
        mx = fmaxf(mx, __shfl_xor(mx, 16)); mx = fmaxf(mx, __shfl_xor(mx, 32));
        if ((lane & (WGB == 2 ? 0x37 : 0x3F)) == 0) red[wave * 2 + lboard] = mx;
        __syncthreads();
        mx = fmaxf(fmaxf(red[0 + lboard], red[2 + lboard]), fmaxf(red[4 + lboard], red[6 + lboard]));
#pragma unroll
        for (int t = 0; t < NT; t++) {
            float ps = 0.f;
#pragma unroll
            for (int i = 0; i < 5; i++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int co = i * 16 + 4 * kg + r;
                    const float e = co < 73 ? __expf(pa[i][t][r] - mx) : 0.f;
                    pa[i][t][r] = e;
                    ps += e;
                }
            ps += __shfl_xor(ps, 16);
            ps += __shfl_xor(ps, 32);
            const int row = tile_row<WGB>(wave * NT + t, p16);
            if (kg == 0) psum[row] = ps;                               // row = board*64 + position
        }
        __syncthreads();
        float tot = 0.f;
#pragma unroll 8
        for (int p = 0; p < 64; p++) tot += psum[lboard * 64 + p];
        inv = 1.0f / tot;
    }
    if (board0 + lboard < n_boards) {
        float* pb = hp.probs + (size_t)(board0 + lboard) * 4672;
#pragma unroll
        for (int t = 0; t < NT; t++) {
            const int pos = tile_row<WGB>(wave * NT + t, p16) & 63;
#pragma unroll
            for (int i = 0; i < 5; i++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int co = i * 16 + 4 * kg + r;
                    if (co < 73) pb[co * 64 + pos] = pa[i][t][r] * inv;
                }
        }
    }
}

// planes : bit-packed [n_boards][1 KiB] (SZ_NN_IN_BITS) or bf16 NHWC [n_boards][64][128]
// wstream: the whole tower's weights in k-step order (see the header); bias [n_convs][256] f32 (BatchNorm folded)
// out    : the tower activation, f32 NHWC [n_boards][64][256]
// MODE: 0 = shipped; 1 = diagnostic build with s_memtime stamps around the phases of convolutions 7 and 8 of a workgroup's second tile (tools/split_stamps.py; the
// stamps go to a buffer of their own); 2 / 3 / 4 = stamps + K loop without weight loads / without LDS fragment reads / without both (results garbage)
#define SPSTAMP(k) do { if (MODE != 0 && stamp_now) { unsigned long long _t = __builtin_amdgcn_s_memtime(); if ((threadIdx.x & 63) == 0) stamps[(size_t)(blockIdx.x * 4 + (threadIdx.x >> 6)) * 16 + (k)] = _t; } } while (0)
// HEADS: both heads run on the tile at the end (split_heads_tail; hp.probs / hp.v1_out instead of `out`, which may then be NULL)
// E: operand element — ElemBF16 (hi + lo bf16: 16 bits of mantissa) or ElemF16 (hi + lo f16: 22 bits, fp32's class; weights and biases packed times 2^SP_WSCALE_LOG2 so that
// the lo parts of weights down to 1e-3 stay normal numbers; activations of this network are O(1) and need no scaling)
template <int WGB, int MODE, bool HEADS, class E = ElemBF16>
__global__ __launch_bounds__(256, 1) void k_tower_split(const uint16_t* __restrict__ planes, const uint4* __restrict__ wstream, const float* __restrict__ bias,
                                                         float* __restrict__ out, int n_boards, int n_blocks, int flags, unsigned long long* __restrict__ stamps,
                                                         const SplitHeadsParams hp) {
    constexpr int ABL = MODE >= 2 ? MODE - 1 : 0;
    using GEO = SplitGeom<WGB>;
    constexpr int NJ = GEO::NJ;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* imgH = lds;
    unsigned char* imgL = lds + GEO::IMG;
    for (int c = threadIdx.x; c < NN_ZERO16 / 16; c += 256) {
        *(uint4*)(imgH + WGB * 64 * GEO::PITCH + c * 16) = make_uint4(0, 0, 0, 0);
        *(uint4*)(imgL + WGB * 64 * GEO::PITCH + c * 16) = make_uint4(0, 0, 0, 0);
    }
    int* addr_tab = (int*)(lds + GEO::TAB);
    for (int e = threadIdx.x; e < 9 * NJ * 64; e += 256)
        addr_tab[e] = conv_tap_addr16<GEO::PITCH, 9, WGB, true>(e / (NJ * 64), (e >> 6) % NJ, e & 15, (e >> 4) & 3);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n_tiles = (n_boards + WGB - 1) / WGB;
    const int n_convs = 1 + 2 * n_blocks;
    const WSrc wr = wfrag_rsrc(wstream);
    f32x4 acc[4][NJ];
    float xres[4][NJ][4];                                              // the residual x (exact f32), one value per AGPR
    uint4 ring[SP_RING][8];
    {   // the first SP_PF k-steps of the stem
        const uint32_t wlane = (uint32_t)((wave * 4) * 64 + lane) * 16u;
#pragma unroll
        for (int k = 0; k < SP_PF; k++)
#pragma unroll
            for (int f = 0; f < 8; f++) ring[k][f] = ld_wfrag(wr, (size_t)k * SP_KSTEP_U4 + (f >= 4 ? SP_KSTEP_U4 / 2 : 0), wlane + (f & 3) * 1024);
    }
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int board0 = tile * WGB;
        // nobody reads the images here: the previous tile ended with epilogue + barrier, its output went out from registers
        if (flags & SZ_NN_IN_BITS) stage_tile_bits<WGB, NN_PAD16, E, true>(imgL, planes, board0, n_boards);
        else stage_tile<128, WGB, NN_PAD16, false, true>(imgL, planes, board0, n_boards, false);
        __syncthreads();
        uint32_t ks = 0;
        const uint32_t ks_p1 = 36u + 72u * (uint32_t)(n_convs - 1);    // conv_p1's 8 k-steps follow the tower's in the stream
        const uint32_t ks_end = HEADS ? ks_p1 : 0u;                    // what the tower's last convolution prefetches: conv_p1, or the next tile's stem
        split_kloop<128, WGB, false, true, 0, 9, E>(lds, GEO::IMG, GEO::IMG, addr_tab, wr, ks, n_convs > 1 ? 36u : ks_end, bias, acc, ring);
        ks += 36;
        __syncthreads();                                               // every wave is done reading the planes
        split_epilogue<WGB, 0, E>(imgH, imgL, acc, xres);
        __syncthreads();
        for (int c = 1; c < n_convs; c++) {
            const bool stamp_now = MODE != 0 && (c == 7 || c == 8) && tile == (int)(blockIdx.x + gridDim.x);
            const int sb = (c & 1) ? 0 : 5;
            SPSTAMP(sb + 0);
            split_kloop<256, WGB, true, true, ABL, 9, E>(lds, 0, GEO::IMG, addr_tab, wr, ks, c + 1 < n_convs ? ks + 72u : ks_end, bias + c * NN_COUT, acc, ring);
            ks += 72;
            SPSTAMP(sb + 1);
            __syncthreads();                                           // every wave is done reading the images: they are rewritten in place
            SPSTAMP(sb + 2);
            if (c & 1) split_epilogue<WGB, 1, E>(imgH, imgL, acc, xres);
            else split_epilogue<WGB, 2, E>(imgH, imgL, acc, xres);
            SPSTAMP(sb + 3);
            __syncthreads();
            SPSTAMP(sb + 4);
            if (MODE != 0 && stamp_now && c == 8 && (threadIdx.x & 63) == 0) stamps[(size_t)(blockIdx.x * 4 + (threadIdx.x >> 6)) * 16 + 10] = __builtin_amdgcn_s_memrealtime();
        }
        if constexpr (HEADS) split_heads_tail<WGB, E>(lds, addr_tab, wr, ks_p1, bias + n_convs * NN_COUT, hp, acc, ring, xres, board0, n_boards);
        // tower output straight from the registers (exact f32): lane = 4 channels of one position per tile, 64-byte pieces
        const int p16 = lane & 15, kg = lane >> 4;
        if (!HEADS || out)
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const int row = tile_row<WGB>(j, p16);
            if (board0 + (row >> 6) < n_boards) {
                float* dst = out + ((size_t)board0 * 64 + row) * NN_COUT;
#pragma unroll
                for (int i = 0; i < 4; i++)
                    *(f32x4*)(dst + (wave * 4 + i) * 16 + 4 * kg) = f32x4{from_agpr(xres[i][j][0]), from_agpr(xres[i][j][1]), from_agpr(xres[i][j][2]), from_agpr(xres[i][j][3])};
            }
        }
    }
}

// =================================================================================================================
// One 3x3 convolution (256 -> 256 channels, padding 1, no bias) on f32 NCHW tensors at the reference's precision class — for the TRAINING step
// (train_RL.py:103-122 runs network.py's fp32 convolutions forward and backward through MIOpen: at batch 128 its fp32 kernels take 89-105 us per convolution,
// 51 % of an optimiser step).  Same machinery as the inference tower: a workgroup takes one board, stages it as hi / lo images in LDS, runs split_kloop
// (three MFMAs per product, f32 accumulate) over a 72-k-step weight stream and writes its 64 channels x 64 positions per wave back as f32.
// Forward: y = conv(x, w).  Backward-data is the same kernel on the gradient with the weights transposed and flipped (sz_nn_pack_conv_split_dev, transposed = 1).
// Operand element E:
//   ElemBF16  hi + lo bf16: 16 bits of mantissa at any magnitude (4.5e-6 from fp64 per convolution);
//   ElemF16   hi + lo f16: 22 bits — fp32's own class (measured: see tools/trainconv_probe.py) — made range-safe by power-of-two scaling, which is exact: the
//             weights are packed times 2^SP_WSCALE_LOG2 (|w| up to 63 stays finite, weights down to 1e-3 keep a normal lo part), every BOARD is scaled by its own
//             2^k so that its largest magnitude lands in [2^11, 2^12) (found in-kernel from the values the threads already hold: gradients of 1e-7 and
//             activations of 1e+3 are treated alike), and the output is multiplied by 2^-(k + SP_WSCALE_LOG2).  Elements far below a board's maximum have a
//             subnormal lo part: their error is bounded by 2^-37 of the maximum, absolutely.
// x, y: [n_boards][256][8][8] f32;  w_stream: 72 k-steps x {hi 16 KB, lo 16 KB} from sz_nn_pack_conv_split_dev.
// COSPLIT = 2: two workgroups per board, 128 output channels each (batches of at most #CUs / 2 boards: a batch-128 train step would otherwise leave half the chip idle)
// Where the time goes at 128 boards (tools/trainconv_time.py with the SZ_CONV_ABL builds, back-to-back launches: 30.9 us): without the K loop 8.4 us, without the LDS stage
// writes or with a quarter of the stores -1.4 us each, none of the three 5.5 us.  The K loop's 22.5 us for 13 us of MFMAs is the weight stream: 256 workgroups x 1.18 MB =
// 302 MB per launch out of the L2s, 13 TB/s (64 boards on half of the CUs: 23.7 us; fetching three k-steps ahead instead of one changes nothing: bandwidth, not latency).
#ifndef SZ_CONV_ABL
#define SZ_CONV_ABL 0                                        // timing builds (results garbage): 1 = no K loop, 2 = no LDS stage writes, 4 = a quarter of the stores
#endif
template <class E, int COSPLIT>
__global__ __launch_bounds__(256, 1) void k_conv3x3_split_f32(const float* __restrict__ x, const uint4* __restrict__ wstream, const float* __restrict__ zero_bias,
                                                               float* __restrict__ y, int n_boards, unsigned int* __restrict__ amax_bits /* optional: atomicMax of the f32 bit
                                                               pattern of max |x| over the whole tensor (what the weight-gradient kernel scales by) */) {
    constexpr int WGB = 1;
    constexpr bool SCALED = std::is_same<E, ElemF16>::value;
    using GEO = SplitGeom<WGB>;
    constexpr int NJ = GEO::NJ, PITCH = GEO::PITCH, NI = 4 / COSPLIT;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* imgH = lds;
    unsigned char* imgL = lds + GEO::IMG;
    float* wmax = (float*)(lds + GEO::SCR);                            // [4 waves]: the board's largest magnitude
    for (int c = threadIdx.x; c < NN_ZERO16 / 16; c += 256) {
        *(uint4*)(imgH + WGB * 64 * PITCH + c * 16) = make_uint4(0, 0, 0, 0);
        *(uint4*)(imgL + WGB * 64 * PITCH + c * 16) = make_uint4(0, 0, 0, 0);
    }
    int* addr_tab = (int*)(lds + GEO::TAB);
    for (int e = threadIdx.x; e < 9 * NJ * 64; e += 256)
        addr_tab[e] = conv_tap_addr16<PITCH, 9, WGB, true>(e / (NJ * 64), (e >> 6) % NJ, e & 15, (e >> 4) & 3);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const WSrc wr = wfrag_rsrc(wstream);
    f32x4 acc[NI][NJ];
    uint4 ring[SP_RING][2 * NI];
    const int half = COSPLIT == 2 ? (int)(blockIdx.x & 1) : 0, ct0 = half * 8 + wave * NI;      // this wave's first channel tile
    {
        const uint32_t wlane = (uint32_t)(ct0 * 64 + lane) * 16u;
#pragma unroll
        for (int k = 0; k < SP_PF; k++)
#pragma unroll
            for (int f = 0; f < 2 * NI; f++) ring[k][f] = ld_wfrag(wr, (size_t)k * SP_KSTEP_U4 + (f >= NI ? SP_KSTEP_U4 / 2 : 0), wlane + (f % NI) * 1024);
    }
    for (int board = blockIdx.x / COSPLIT; board < n_boards; board += gridDim.x / COSPLIT) {
        __syncthreads();                                               // the previous board's images are fully read
        // stage: the board is 256 channels x 64 positions of f32, channel-major; a wave instruction reads 1 KiB = 4 channels x 64 positions, lane l holds positions
        // 4*(l & 15) .. +3 of channel 4*q + (l >> 4); every value goes to its (position row, channel) slot of the hi and of the lo image (swizzled chunks)
        const float4* src = (const float4*)(x + (size_t)board * 256 * 64);
        float4 v[16];
#pragma unroll
        for (int t = 0; t < 16; t++) v[t] = src[(wave + 4 * t) * 64 + lane];
        float sx = 1.f, unscale = 1.f;
        if (SCALED) {
            float m = 0.f;
#pragma unroll
            for (int t = 0; t < 16; t++) m = fmaxf(fmaxf(m, fmaxf(fabsf(v[t].x), fabsf(v[t].y))), fmaxf(fabsf(v[t].z), fabsf(v[t].w)));
            for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
            if (lane == 0) wmax[wave] = m;
            __syncthreads();
            m = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
            if (amax_bits && threadIdx.x == 0 && half == 0) atomicMax(amax_bits, __builtin_bit_cast(unsigned int, m));
            // 2^k with max * 2^k in [2^11, 2^12): from the exponent field (an all-zero or non-finite board keeps scale 1)
            const int ex = (int)((__builtin_bit_cast(uint32_t, m) >> 23) & 0xFF);
            const int k = (ex == 0 || ex == 255) ? 0 : 11 - (ex - 127);
            const int kc = k < -100 ? -100 : (k > 100 ? 100 : k);
            sx = __builtin_bit_cast(float, (uint32_t)(127 + kc) << 23);
            unscale = __builtin_bit_cast(float, (uint32_t)(127 - kc - SP_WSCALE_LOG2) << 23);
        }
#pragma unroll
        for (int t = 0; t < 16; t++) {
            const int q = wave + 4 * t;
            const int c = q * 4 + (lane >> 4), p0 = (lane & 15) * 4;
            const float vv[4] = {v[t].x * sx, v[t].y * sx, v[t].z * sx, v[t].w * sx};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int row = p0 + k;
                const int off = row * PITCH + (((c >> 3) ^ ((row >> 2) & 1)) << 4) + (c & 7) * 2;
                const uint32_t hb = E::pack2(vv[k], 0.f) & 0xFFFFu;
                const uint32_t lb = E::pack2(vv[k] - E::lo(hb), 0.f) & 0xFFFFu;
                if (!(SZ_CONV_ABL & 2) || hb == 0x1234u) {
                    *(uint16_t*)(imgH + off) = (uint16_t)hb;
                    *(uint16_t*)(imgL + off) = (uint16_t)lb;
                }
            }
        }
        __syncthreads();
        if (SZ_CONV_ABL & 1) {
#pragma unroll
            for (int i = 0; i < NI; i++)
#pragma unroll
                for (int j = 0; j < NJ; j++) acc[i][j] = f32x4{sx, sx, sx, sx};
        } else
        split_kloop<256, WGB, true, true, 0, 9, E, NI>(lds, 0, GEO::IMG, addr_tab, wr, 0u, 0u, zero_bias, acc, ring, ct0);      // the stream restarts for the next board
        // acc tile (i, j): lane (p16, kg) holds channels (wave*4 + i)*16 + 4*kg + r, r = 0..3, of position j*16 + p16
        float* dst = y + (size_t)board * 256 * 64;
        const int p16 = lane & 15, kg = lane >> 4;
#pragma unroll
        for (int i = 0; i < NI; i++)
#pragma unroll
            for (int j = 0; j < NJ; j++)
#pragma unroll
                for (int r = 0; r < ((SZ_CONV_ABL & 4) ? 1 : 4); r++) dst[(size_t)((ct0 + i) * 16 + 4 * kg + r) * 64 + j * 16 + p16] = SCALED ? acc[i][j][r] * unscale : acc[i][j][r];
    }
}

// torch conv weight [256 co][256 ci][3][3] f32 (device) -> the 72-k-step hi / lo fragment stream of k_conv3x3_split_f32 (device), one thread per element.
// transposed = 1: the stream of the backward-data convolution, W'[ci][co][tap] = w[co][ci][8 - tap].  ElemF16: times 2^SP_WSCALE_LOG2.
template <class E>
__global__ __launch_bounds__(256) void k_pack_conv_split(const float* __restrict__ w, uint16_t* __restrict__ stream, int transposed, unsigned int* __restrict__ zero_u32,
                                                         uint16_t* __restrict__ stream_t /* optional: the second half of the grid packs the transposed stream here, and zero_u32 has two slots */) {
    int idx = blockIdx.x * 256 + threadIdx.x;                         // over 72 k-steps x 16 tiles x 64 lanes x 8 elements (twice with stream_t)
    if (zero_u32 && idx == 0) { zero_u32[0] = 0u; if (stream_t) zero_u32[1] = 0u; }      // the maximum slots of the convolutions that follow on this stream
    if (stream_t && idx >= 72 * 16 * 64 * 8) { idx -= 72 * 16 * 64 * 8; stream = stream_t; transposed = 1; }
    if (idx >= 72 * 16 * 64 * 8) return;
    const int e = idx & 7, l = (idx >> 3) & 63, tile = (idx >> 9) & 15, ks = idx >> 13;
    const int tap = ks >> 3, k32 = ks & 7;
    const int co = tile * 16 + (l & 15), ci = k32 * 32 + 8 * (l >> 4) + e;
    float v = transposed ? w[((size_t)ci * 256 + co) * 9 + (8 - tap)] : w[((size_t)co * 256 + ci) * 9 + tap];
    if (std::is_same<E, ElemF16>::value) v *= (float)(1 << SP_WSCALE_LOG2);
    const uint32_t h = E::pack2(v, 0.f) & 0xFFFFu;
    const uint32_t lo = E::pack2(v - E::lo(h), 0.f) & 0xFFFFu;
    uint16_t* rec = stream + (size_t)ks * SP_KSTEP_U4 * 8;
    rec[((size_t)tile * 64 + l) * 8 + e] = (uint16_t)h;
    rec[(size_t)SP_KSTEP_U4 * 4 + ((size_t)tile * 64 + l) * 8 + e] = (uint16_t)lo;
}

// =================================================================================================================
// Weight gradient of that convolution: dW[co][ci][tap] = sum over boards and positions of gy[b][co][pos] * x[b][ci][pos + off(tap)], on hi + lo f16 operands
// (three MFMAs per product, f32 accumulate; both tensors scaled by ONE power of two each, taken from the maxima the forward / backward-data kernels left behind).
// The reduction dimension of the MFMA is the POSITION: a k-step is 32 positions = 4 board rows, a lane's 8 k-elements are one board row.
//   grid = 4 co blocks x 4 ci blocks x KG board groups; a workgroup accumulates its 64 co x 64 ci x 9 taps over its boards (wave = 2 co tiles x 2 ci tiles x 9 taps:
//   144 accumulator registers) and writes ONE partial, [KG][tap][co][ci]; k_wgrad_reduce sums the KG partials into dW [co][ci][tap].
//   gy fragments (A operand, rows = co) come straight from global memory (a lane's 8 floats are contiguous); the x block is staged per board in LDS as f16 hi / lo in
//   three column-shifted copies (dx = -1, 0, +1) with a zero row above and below, so that every tap's B fragment is one aligned ds_read_b128:
//   Xs[dx][ci tile][row -1..8]{hi: ci 0..15 x 16 B, lo: ci 0..15 x 16 B}, row pitch 512 B: the 16 lanes of a ds_read_b128 / ds_write_b128 group touch 16 consecutive
//   16-byte slots of one row.  SQ_LDS_BANK_CONFLICT = 0 (rocprofv3 --pmc on tools/wgrad_time.py; at row pitches 528 / 544 / 576 / 640 B: 45 % of the LDS-active cycles,
//   at the same launch time).
#define WG_ROWPITCH 512
#define WG_TILE_BYTES (10 * WG_ROWPITCH)
#define WG_LDS_BYTES (3 * 4 * WG_TILE_BYTES)
__device__ __forceinline__ float pow2_from_amax_bits(unsigned int bits, int target_log2, int* k_out) {
    const int ex = (int)((bits >> 23) & 0xFF);
    int k = (ex == 0 || ex == 255) ? 0 : target_log2 - (ex - 127);
    k = k < -100 ? -100 : (k > 100 ? 100 : k);
    *k_out = k;
    return __builtin_bit_cast(float, (uint32_t)(127 + k) << 23);
}
__global__ __launch_bounds__(256, 1) void k_wgrad3x3_split(const float* __restrict__ gy, const float* __restrict__ x, const unsigned int* __restrict__ amax_gy,
                                                            const unsigned int* __restrict__ amax_x, float* __restrict__ part, int n_boards, int KG) {
    using E = ElemF16;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int cob = blockIdx.x & 3, cib = (blockIdx.x >> 2) & 3, kgrp = blockIdx.x >> 4;
    const int chunk = (n_boards + KG - 1) / KG, b0 = kgrp * chunk, b1 = min(n_boards, b0 + chunk);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wr = wave >> 1, wc = wave & 1;
    const int n = lane & 15, kg = lane >> 4;
    int kgy, kx;
    const float sgy = pow2_from_amax_bits(*amax_gy, 11, &kgy), sx = pow2_from_amax_bits(*amax_x, 11, &kx);
    const float unscale_g = __builtin_bit_cast(float, (uint32_t)(127 - kgy) << 23), unscale_x = __builtin_bit_cast(float, (uint32_t)(127 - kx) << 23);
    for (int c = threadIdx.x; c < WG_LDS_BYTES / 16; c += 256) *(uint4*)(lds + c * 16) = make_uint4(0, 0, 0, 0);      // zero rows / edge columns stay zero
    f32x4 acc[2][2][9];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
            for (int t = 0; t < 9; t++) acc[a][c][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    // staging role: thread -> (ci = lane of the block's 64, rows 2q, 2q+1 with q = wave): the 16 lanes of a ds_write_b128 group write 16 consecutive 16-byte slots of one
    // row (with ci = tid >> 2, q = tid & 3 a group's lanes were spread over four rows)
    const int sci = threadIdx.x & 63, sq = threadIdx.x >> 6;
    const int sbase = (sci >> 4) * WG_TILE_BYTES + (sci & 15) * 16;
    // Software pipeline over the boards: the global loads of board b + 1 (x rows for the stage, gy rows for the A fragments) are issued before board b's MFMAs and
    // consumed after them, so their latency hides behind 216 MFMAs per wave instead of standing in front of every board (49.7 -> 41.8 us at 128 boards).
    // (Tried on top and slower, profiles/r03zzk_wgrad_variants.txt: a second stage buffer with the conversion moved behind the MFMAs and the accumulator chains
    // interleaved by hand, 47.8 us; 64 x 32 blocks with 8 board groups — half the partial traffic, 6.8 instead of 11.6 us in the reduction — 49.7 us; operands read pre-split from the
    // convolutions instead of converted here, chains interleaved by hand: no change.)
    float4 xq[4], gq[2][2][2];
    auto load_board = [&](int bb) {
        const float4* src = (const float4*)(x + ((size_t)bb * 256 + cib * 64 + sci) * 64 + sq * 16);
#pragma unroll
        for (int i = 0; i < 4; i++) xq[i] = src[i];
#pragma unroll
        for (int cot = 0; cot < 2; cot++)
#pragma unroll
            for (int kh = 0; kh < 2; kh++) {
                const int co = cob * 64 + (2 * wr + cot) * 16 + n;
                const float4* g = (const float4*)(gy + ((size_t)bb * 256 + co) * 64 + (4 * kh + kg) * 8);
                gq[cot][kh][0] = g[0]; gq[cot][kh][1] = g[1];
            }
    };
    if (b0 < b1) load_board(b0);
    for (int b = b0; b < b1; b++) {
        __syncthreads();                                               // the previous board's fragments are read
        {
            const float4 v0 = xq[0], v1 = xq[1], v2 = xq[2], v3 = xq[3];
            const float rows[2][8] = {{v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w}, {v2.x, v2.y, v2.z, v2.w, v3.x, v3.y, v3.z, v3.w}};
#pragma unroll
            for (int rr = 0; rr < 2; rr++) {
                uint32_t h[8], l[8];
#pragma unroll
                for (int c = 0; c < 8; c++) {
                    const float sv = rows[rr][c] * sx;
                    h[c] = E::pack2(sv, 0.f) & 0xFFFFu;
                    l[c] = E::pack2(sv - E::lo(h[c]), 0.f) & 0xFFFFu;
                }
                const int row = 2 * sq + rr + 1;                       // +1: the zero row above
#pragma unroll
                for (int dxi = 0; dxi < 3; dxi++) {                    // copy dxi holds x[col + (dxi - 1)] at col
                    uint32_t ph[4], pl[4];
#pragma unroll
                    for (int p = 0; p < 4; p++) {
                        const int c0 = 2 * p + dxi - 1, c1 = c0 + 1;
                        ph[p] = ((c0 >= 0 && c0 < 8) ? h[c0] : 0u) | (((c1 >= 0 && c1 < 8) ? h[c1] : 0u) << 16);
                        pl[p] = ((c0 >= 0 && c0 < 8) ? l[c0] : 0u) | (((c1 >= 0 && c1 < 8) ? l[c1] : 0u) << 16);
                    }
                    unsigned char* dst = lds + dxi * 4 * WG_TILE_BYTES + sbase + row * WG_ROWPITCH;
                    *(uint4*)dst = make_uint4(ph[0], ph[1], ph[2], ph[3]);
                    *(uint4*)(dst + 256) = make_uint4(pl[0], pl[1], pl[2], pl[3]);
                }
            }
        }
        // A fragments: gy rows of this wave's two co tiles, both k halves
        bf16x8 ah[2][2], al[2][2];
#pragma unroll
        for (int cot = 0; cot < 2; cot++)
#pragma unroll
            for (int kh = 0; kh < 2; kh++) {
                const float4 g0 = gq[cot][kh][0], g1 = gq[cot][kh][1];
                const float f[8] = {g0.x * sgy, g0.y * sgy, g0.z * sgy, g0.w * sgy, g1.x * sgy, g1.y * sgy, g1.z * sgy, g1.w * sgy};
                uint32_t hh[4], ll[4];
#pragma unroll
                for (int p = 0; p < 4; p++) {
                    hh[p] = E::pack2(f[2 * p], f[2 * p + 1]);
                    ll[p] = E::pack2(f[2 * p] - E::lo(hh[p]), f[2 * p + 1] - E::hi(hh[p]));
                }
                ah[cot][kh] = __builtin_bit_cast(bf16x8, make_uint4(hh[0], hh[1], hh[2], hh[3]));
                al[cot][kh] = __builtin_bit_cast(bf16x8, make_uint4(ll[0], ll[1], ll[2], ll[3]));
            }
        if (b + 1 < b1) load_board(b + 1);                             // in flight under this board's MFMAs
        __syncthreads();
        // 36 groups (k half, ci tile, tap) of two fragment reads + six MFMAs.  The fragments are fetched WG_AHEAD groups ahead into a register ring and the schedule is
        // pinned: left to itself hipcc issues a group's two ds_read_b128 right in front of its MFMAs and waits for them — the LDS latency stood exposed in front of every
        // group of 96 MFMA cycles (24.5 us of MFMA loop for 13 us of MFMAs; tools/wgrad_time.py with the WG_ABL builds).
        constexpr int WG_AHEAD = 2, NGRP = 36;
        bf16x8 bh[WG_AHEAD + 1], bl[WG_AHEAD + 1];
        auto fetch = [&](int g) {
            const int kh = g / 18, cit = (g / 9) & 1, tap = g % 9, dy = tap / 3 - 1, dxi = tap % 3;
            const unsigned char* src = lds + (dxi * 4 + 2 * wc + cit) * WG_TILE_BYTES + (4 * kh + kg + dy + 1) * WG_ROWPITCH + n * 16;
            bh[g % (WG_AHEAD + 1)] = *(const bf16x8*)src; bl[g % (WG_AHEAD + 1)] = *(const bf16x8*)(src + 256);
        };
#pragma unroll
        for (int g = 0; g < WG_AHEAD; g++) fetch(g);
#pragma unroll
        for (int g = 0; g < NGRP; g++) {
            if (g + WG_AHEAD < NGRP) fetch(g + WG_AHEAD);
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            const int kh = g / 18, cit = (g / 9) & 1, tap = g % 9, sl = g % (WG_AHEAD + 1);
#pragma unroll
            for (int cot = 0; cot < 2; cot++) {
                acc[cot][cit][tap] = E::mfma(ah[cot][kh], bh[sl], acc[cot][cit][tap]);
                acc[cot][cit][tap] = E::mfma(al[cot][kh], bh[sl], acc[cot][cit][tap]);
                acc[cot][cit][tap] = E::mfma(ah[cot][kh], bl[sl], acc[cot][cit][tap]);
            }
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // the partial: [kgrp][tap][co][ci]; D layout: lane column n = ci, rows 4*kg + r = co
#pragma unroll
    for (int cot = 0; cot < 2; cot++)
#pragma unroll
        for (int cit = 0; cit < 2; cit++)
#pragma unroll
            for (int tap = 0; tap < 9; tap++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int co = cob * 64 + (2 * wr + cot) * 16 + 4 * kg + r, ci = cib * 64 + (2 * wc + cit) * 16 + n;
                    part[(((size_t)kgrp * 9 + tap) * 256 + co) * 256 + ci] = acc[cot][cit][tap][r] * unscale_g * unscale_x;
                }
}
// dW[co][ci][tap] = sum over the KG partials [kg][tap][co][ci]; one thread per (co, ci)
__global__ __launch_bounds__(256) void k_wgrad_reduce(const float* __restrict__ part, float* __restrict__ dw, int KG) {
    const int idx = blockIdx.x * 256 + threadIdx.x;                   // co * 256 + ci
    float s[9];
#pragma unroll
    for (int t = 0; t < 9; t++) s[t] = 0.f;
    for (int k = 0; k < KG; k++)
#pragma unroll
        for (int t = 0; t < 9; t++) s[t] += part[((size_t)k * 9 + t) * 65536 + idx];
#pragma unroll
    for (int t = 0; t < 9; t++) dw[(size_t)idx * 9 + t] = s[t];
}

static unsigned long long* g_split_stamps = nullptr;
static int g_split_mode = 1;

extern "C" {

// diagnostic: device buffer of 16 u64 per wave (256 workgroups x 4 waves) that receives the phase stamps of the stamped build; NULL = shipped kernel
int sz_nn_debug_split_stamps(void* dev_buffer, int32_t mode) { g_split_stamps = (unsigned long long*)dev_buffer; g_split_mode = mode; return SZ_OK; }

// Number of uint16 (bf16) elements of the weight stream of a tower with n_blocks BasicBlocks.
int64_t sz_nn_split_stream_elems(int32_t n_blocks) {
    if (n_blocks < 0) return SZ_ERR_INVALID;
    return (int64_t)(36 + 72 * 2 * (int64_t)n_blocks + 8) * SP_KSTEP_U4 * 8;      // stem, 2 convolutions per block, conv_p1 (1x1: 8 k-steps) for the fused heads
}

// Host-side packing of ONE convolution into its place in the weight stream.
//   w_in  : [256 co][cin_real][ksize][ksize] f32 (BatchNorm folded by the caller)
//   conv  : 0 = stem (cin_real 119, padded to 128: 36 k-steps), c >= 1: the c-th 256-channel 3x3 convolution (72 k-steps each); with ksize 1 and
//           conv = 1 + 2*n_blocks: conv_p1 of the policy head (8 k-steps behind the tower's)
//   stream: the whole stream (sz_nn_split_stream_elems elements); k-step record = 16 co tiles x 64 lanes x 8 bf16 of w_hi, then of w_lo,
//           fragment order of sz_nn_pack_weights16: lane l, elem e <- w[co = tile*16 + (l&15)][ci = k32*32 + 8*(l>>4) + e]
//   w_hi = bf16(w) (round to nearest even), w_lo = bf16(w - w_hi)
static int pack_split_stream_impl(const float* w_in, int32_t cin_real, int32_t ksize, int32_t conv, uint16_t* stream, bool f16);
int sz_nn_pack_split_stream(const float* w_in, int32_t cin_real, int32_t ksize, int32_t conv, uint16_t* stream) { return pack_split_stream_impl(w_in, cin_real, ksize, conv, stream, false); }
// the same for the f16-operand kernels (SZ_NN_F16): elements are f16 and the weights are multiplied by 2^10 first (the caller scales the biases alike)
int sz_nn_pack_split_stream_f16(const float* w_in, int32_t cin_real, int32_t ksize, int32_t conv, uint16_t* stream) { return pack_split_stream_impl(w_in, cin_real, ksize, conv, stream, true); }
}  // extern "C"
static int pack_split_stream_impl(const float* w_in, int32_t cin_real, int32_t ksize, int32_t conv, uint16_t* stream, bool f16) {
    if (!w_in || !stream || conv < 0 || conv > NN_MAX_CONVS_SPLIT || (ksize != 1 && ksize != 3) || (ksize == 1 && !(conv & 1))) return SZ_ERR_INVALID;
    const int cin_padded = conv == 0 ? 128 : 256, taps = ksize * ksize;
    if (cin_real <= 0 || cin_real > cin_padded) return SZ_ERR_INVALID;
    const int ksteps = cin_padded / 32;
    const size_t ks0 = conv == 0 ? 0 : 36 + (size_t)(conv - 1) * 72;
    const float wscale = f16 ? (float)(1 << SP_WSCALE_LOG2) : 1.f;
    auto rne = [f16](float v) -> uint16_t { return f16 ? ElemF16::from_float(v) : ElemBF16::from_float(v); };
    auto back = [f16](uint16_t h) -> float { if (f16) { _Float16 x; memcpy(&x, &h, 2); return (float)x; } const uint32_t hu = (uint32_t)h << 16; float hf; memcpy(&hf, &hu, 4); return hf; };
    for (int t = 0; t < taps; t++)
        for (int k = 0; k < ksteps; k++) {
            uint16_t* rec = stream + (ks0 + (size_t)t * ksteps + k) * SP_KSTEP_U4 * 8;
            for (int tile = 0; tile < 16; tile++)
                for (int l = 0; l < 64; l++)
                    for (int e = 0; e < 8; e++) {
                        const int co = tile * 16 + (l & 15), ci = k * 32 + 8 * (l >> 4) + e;
                        const float v = (ci < cin_real ? w_in[((size_t)co * cin_real + ci) * taps + t] : 0.f) * wscale;
                        const uint16_t h = rne(v);
                        rec[((size_t)tile * 64 + l) * 8 + e] = h;
                        rec[(size_t)SP_KSTEP_U4 * 4 + ((size_t)tile * 64 + l) * 8 + e] = rne(v - back(h));
                    }
        }
    return SZ_OK;
}
extern "C" {

// host: conv_p2.weight [73][256] f32 -> [8 k32-steps]{hi: 5 co tiles x 64 lanes x 8, lo: the same} bf16 (channels 73..79 zero), 8*2*5*64*8 elements
static int pack_split_head_impl(const float* w_in, uint16_t* out, bool f16);
int sz_nn_pack_split_head(const float* w_in, uint16_t* out) { return pack_split_head_impl(w_in, out, false); }
int sz_nn_pack_split_head_f16(const float* w_in, uint16_t* out) { return pack_split_head_impl(w_in, out, true); }
}  // extern "C"
static int pack_split_head_impl(const float* w_in, uint16_t* out, bool f16) {
    if (!w_in || !out) return SZ_ERR_INVALID;
    const float wscale = f16 ? (float)(1 << SP_WSCALE_LOG2) : 1.f;
    auto rne = [f16](float v) -> uint16_t { return f16 ? ElemF16::from_float(v) : ElemBF16::from_float(v); };
    auto back = [f16](uint16_t h) -> float { if (f16) { _Float16 x; memcpy(&x, &h, 2); return (float)x; } const uint32_t hu = (uint32_t)h << 16; float hf; memcpy(&hf, &hu, 4); return hf; };
    for (int ks = 0; ks < 8; ks++)
        for (int tile = 0; tile < 5; tile++)
            for (int l = 0; l < 64; l++)
                for (int e = 0; e < 8; e++) {
                    const int co = tile * 16 + (l & 15), ci = ks * 32 + 8 * (l >> 4) + e;
                    const float v = (co < 73 ? w_in[(size_t)co * 256 + ci] : 0.f) * wscale;
                    const uint16_t h = rne(v);
                    out[((((size_t)ks * 2 + 0) * 5 + tile) * 64 + l) * 8 + e] = h;
                    out[((((size_t)ks * 2 + 1) * 5 + tile) * 64 + l) * 8 + e] = rne(v - back(h));
                }
    return SZ_OK;
}
extern "C" {

}  // extern "C"

static int launch_split_one(const void* planes, const void* w_stream, const float* bias, int32_t n_blocks, float* out, int32_t n_boards, int32_t flags, void* stream,
                            const SplitHeadsParams* heads);
// As in sz_nn_tower_bf16: above 2 x #CUs boards a last round of at most #CUs boards is a launch of its own in the one-board form (768 boards: 512 in two-board tiles + 256
// with a CU each instead of a second two-board round on half of the CUs); per-board results are identical in both forms.
static int launch_split(const void* planes, const void* w_stream, const float* bias, int32_t n_blocks, float* out, int32_t n_boards, int32_t flags, void* stream,
                        const SplitHeadsParams* heads) {
    const int n_cu = device_cus(), rem = n_boards % (2 * n_cu);
    if (n_boards > 2 * n_cu && rem > 0 && rem <= n_cu && !(flags & (SZ_NN_SPLIT_WGB1 | SZ_NN_SPLIT_WGB2)) && !g_split_stamps) {
        const int head = n_boards - rem;
        const size_t plane_bytes = (flags & SZ_NN_IN_BITS) ? 64 * sizeof(uint4) : (size_t)64 * 128 * 2;
        const int rc = launch_split_one(planes, w_stream, bias, n_blocks, out, head, flags | SZ_NN_SPLIT_WGB2, stream, heads);
        if (rc != SZ_OK) return rc;
        SplitHeadsParams hp2;
        if (heads) { hp2 = *heads; hp2.probs += (size_t)head * 4672; hp2.v1_out += (size_t)head * 64; }
        return launch_split_one((const unsigned char*)planes + head * plane_bytes, w_stream, bias, n_blocks, out ? out + (size_t)head * 64 * 256 : nullptr, rem,
                                flags | SZ_NN_SPLIT_WGB1, stream, heads ? &hp2 : nullptr);
    }
    return launch_split_one(planes, w_stream, bias, n_blocks, out, n_boards, flags, stream, heads);
}
static int launch_split_one(const void* planes, const void* w_stream, const float* bias, int32_t n_blocks, float* out, int32_t n_boards, int32_t flags, void* stream,
                            const SplitHeadsParams* heads) {
    static bool attr_flags[NN_MAX_DEVICES] = {};
    bool& attr_set = attr_flags[current_device_slot()];
    if (!attr_set) {
#define SPLIT_ATTR(W, M, H) HIPCHK(hipFuncSetAttribute((const void*)k_tower_split<W, M, H>, hipFuncAttributeMaxDynamicSharedMemorySize, SplitGeom<W>::LDS_BYTES))
        SPLIT_ATTR(1, 0, false); SPLIT_ATTR(2, 0, false); SPLIT_ATTR(1, 0, true); SPLIT_ATTR(2, 0, true);
        SPLIT_ATTR(2, 1, false); SPLIT_ATTR(2, 2, false); SPLIT_ATTR(2, 3, false); SPLIT_ATTR(2, 4, false);
#undef SPLIT_ATTR
#define SPLIT_ATTR16(W, H) HIPCHK(hipFuncSetAttribute((const void*)k_tower_split<W, 0, H, ElemF16>, hipFuncAttributeMaxDynamicSharedMemorySize, SplitGeom<W>::LDS_BYTES))
        SPLIT_ATTR16(1, false); SPLIT_ATTR16(2, false); SPLIT_ATTR16(1, true); SPLIT_ATTR16(2, true);
#undef SPLIT_ATTR16
        attr_set = true;
    }
    const int n_cu = device_cus();
    const bool one = (flags & SZ_NN_SPLIT_WGB1) || (n_boards <= n_cu && !(flags & SZ_NN_SPLIT_WGB2));
    SplitHeadsParams hp;
    memset(&hp, 0, sizeof hp);
    if (heads) hp = *heads;
    const int n_tiles = one ? n_boards : (n_boards + 1) / 2;
    const dim3 grid(n_tiles < n_cu ? n_tiles : n_cu);
#define SPLIT_LAUNCH(W, M, H, ST) hipLaunchKernelGGL((k_tower_split<W, M, H>), grid, dim3(256), SplitGeom<W>::LDS_BYTES, (hipStream_t)stream, (const uint16_t*)planes, \
                                                     (const uint4*)w_stream, bias, out, n_boards, n_blocks, (int)flags, ST, hp)
#define SPLIT_LAUNCH16(W, H) hipLaunchKernelGGL((k_tower_split<W, 0, H, ElemF16>), grid, dim3(256), SplitGeom<W>::LDS_BYTES, (hipStream_t)stream, (const uint16_t*)planes, \
                                                  (const uint4*)w_stream, bias, out, n_boards, n_blocks, (int)flags, (unsigned long long*)nullptr, hp)
    if (flags & SZ_NN_F16) {
        if (heads) { if (one) SPLIT_LAUNCH16(1, true); else SPLIT_LAUNCH16(2, true); }
        else { if (one) SPLIT_LAUNCH16(1, false); else SPLIT_LAUNCH16(2, false); }
    } else if (heads) {
        if (one) SPLIT_LAUNCH(1, 0, true, (unsigned long long*)nullptr); else SPLIT_LAUNCH(2, 0, true, (unsigned long long*)nullptr);
    } else if (one) SPLIT_LAUNCH(1, 0, false, (unsigned long long*)nullptr);
    else if (!g_split_stamps) SPLIT_LAUNCH(2, 0, false, (unsigned long long*)nullptr);
    else if (g_split_mode == 2) SPLIT_LAUNCH(2, 2, false, g_split_stamps);
    else if (g_split_mode == 3) SPLIT_LAUNCH(2, 3, false, g_split_stamps);
    else if (g_split_mode == 4) SPLIT_LAUNCH(2, 4, false, g_split_stamps);
    else SPLIT_LAUNCH(2, 1, false, g_split_stamps);
#undef SPLIT_LAUNCH
#undef SPLIT_LAUNCH16
    HIPCHK(hipGetLastError());
    return SZ_OK;
}

extern "C" {

// Split-precision tower (k_tower_split): stem + n_blocks BasicBlocks in one persistent launch.
//   w_stream: device buffer built with sz_nn_pack_split_stream; bias: device [1 + 2*n_blocks (+ 1: conv_p1)][256] f32; out: device [n_boards][64][256] f32 NHWC.
int sz_nn_tower_split(const void* planes, const void* w_stream, const float* bias, int32_t n_blocks, float* out, int32_t n_boards, int32_t flags, void* stream) {
    if (!planes || !w_stream || !bias || !out || n_boards <= 0 || n_blocks < 0 || 1 + 2 * n_blocks > NN_MAX_CONVS_SPLIT) return SZ_ERR_INVALID;
    StreamDeviceGuard _guard(stream);
    return launch_split(planes, w_stream, bias, n_blocks, out, n_boards, flags, stream, nullptr);
}

int sz_nn_value_mlp(const float* v1, const float* fc1_w_t, const float* fc1_b, const float* fc2_w, float fc2_b, float* value, int32_t n_boards, void* stream);

// The whole network at the reference's precision class in two launches (network.py:176-192): the tower as above with BOTH heads fused onto each tile while it
// is still in LDS (conv_p1 -> conv_p2 -> softmax on hi + lo operands; conv_v1 in f32), then the 64 -> 256 -> 1 value MLP (sz_nn_value_mlp).
//   w_stream / bias: as for sz_nn_tower_split, with conv_p1 (p_norm1 folded) packed as convolution 1 + 2*n_blocks (ksize 1) and its bias as the last row;
//   w_p2_packed: sz_nn_pack_split_head(conv_p2.weight); b_p2 [73]; wv [256], bv: conv_v1 with v_norm folded; fc1_w_t [64][256], fc1_b [256], fc2_w [256], fc2_b;
//   probs [n_boards][4672] f32 (softmax iff do_softmax, else logits), value [n_boards], v1_scratch [n_boards][64] f32; tower_out: optional f32 [n_boards][64][256].
int sz_nn_forward_split(const void* planes, const void* w_stream, const float* bias, int32_t n_blocks, const void* w_p2_packed, const float* b_p2, const float* wv, float bv,
                        const float* fc1_w_t, const float* fc1_b, const float* fc2_w, float fc2_b, float* probs, float* value, float* v1_scratch, float* tower_out,
                        int32_t n_boards, int32_t do_softmax, int32_t flags, void* stream) {
    if (!planes || !w_stream || !bias || !w_p2_packed || !b_p2 || !wv || !fc1_w_t || !fc1_b || !fc2_w || !probs || !value || !v1_scratch || n_boards <= 0 || n_blocks < 0 ||
        1 + 2 * n_blocks > NN_MAX_CONVS_SPLIT) return SZ_ERR_INVALID;
    StreamDeviceGuard _guard(stream);
    SplitHeadsParams hp;
    hp.w_p2 = (const uint4*)w_p2_packed; hp.b_p2 = b_p2; hp.wv = wv; hp.bv = bv; hp.probs = probs; hp.v1_out = v1_scratch; hp.do_softmax = do_softmax;
    const int rc = launch_split(planes, w_stream, bias, n_blocks, tower_out, n_boards, flags, stream, &hp);
    if (rc != SZ_OK) return rc;
    return sz_nn_value_mlp(v1_scratch, fc1_w_t, fc1_b, fc2_w, fc2_b, value, n_boards, stream);
}

// Training-step convolutions at the reference's precision class (k_conv3x3_split_f32).  x, y: device [n_boards,256,8,8] f32 (NCHW, contiguous); w_stream: device buffer
// of 72*2048*16 bytes written by sz_nn_pack_conv_split_dev; zero256: device [256] f32 zeros (the convolutions of network.py:28,30 have no bias); f16: hi + lo f16
// operands with power-of-two scaling (fp32's class) instead of hi + lo bf16.
int sz_nn_conv3x3_split_f32(const float* x, const void* w_stream, const float* zero256, float* y, int32_t n_boards, int32_t f16, void* amax_bits, void* stream) {
    if (!x || !w_stream || !zero256 || !y || n_boards <= 0) return SZ_ERR_INVALID;
    StreamDeviceGuard _guard(stream);
    static bool attr_flags[NN_MAX_DEVICES] = {};
    bool& attr_set = attr_flags[current_device_slot()];
    if (!attr_set) {
        HIPCHK(hipFuncSetAttribute((const void*)k_conv3x3_split_f32<ElemBF16, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, SplitGeom<1>::LDS_BYTES));
        HIPCHK(hipFuncSetAttribute((const void*)k_conv3x3_split_f32<ElemF16, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, SplitGeom<1>::LDS_BYTES));
        HIPCHK(hipFuncSetAttribute((const void*)k_conv3x3_split_f32<ElemBF16, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, SplitGeom<1>::LDS_BYTES));
        HIPCHK(hipFuncSetAttribute((const void*)k_conv3x3_split_f32<ElemF16, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, SplitGeom<1>::LDS_BYTES));
        attr_set = true;
    }
    const int n_cu = device_cus();
    const bool two = 2 * n_boards <= n_cu;                               // two workgroups per board while that fits in one wave of workgroups
    const dim3 grid(two ? 2 * n_boards : (n_boards < n_cu ? n_boards : n_cu));
#define CONV_LAUNCH(E_, CS_, AM_) hipLaunchKernelGGL((k_conv3x3_split_f32<E_, CS_>), grid, dim3(256), SplitGeom<1>::LDS_BYTES, (hipStream_t)stream, x, (const uint4*)w_stream, zero256, y, n_boards, AM_)
    if (f16) { if (two) CONV_LAUNCH(ElemF16, 2, (unsigned int*)amax_bits); else CONV_LAUNCH(ElemF16, 1, (unsigned int*)amax_bits); }
    else { if (two) CONV_LAUNCH(ElemBF16, 2, (unsigned int*)nullptr); else CONV_LAUNCH(ElemBF16, 1, (unsigned int*)nullptr); }
#undef CONV_LAUNCH
    HIPCHK(hipGetLastError());
    return SZ_OK;
}
// w: device [256,256,3,3] f32 (a torch conv weight); w_stream: device, 72*2048*16 bytes; transposed = 1 packs the backward-data convolution's weights;
// f16 must match the convolution call's; zero_u32: optional device uint32 set to 0 (the amax_bits slot of the convolution that follows on the same stream).
int sz_nn_pack_conv_split_dev(const float* w, int32_t transposed, int32_t f16, void* w_stream, void* zero_u32, void* stream) {
    if (!w || !w_stream) return SZ_ERR_INVALID;
    StreamDeviceGuard _guard(stream);
    if (f16) hipLaunchKernelGGL(k_pack_conv_split<ElemF16>, dim3(72 * 16 * 64 * 8 / 256), dim3(256), 0, (hipStream_t)stream, w, (uint16_t*)w_stream, (int)transposed, (unsigned int*)zero_u32, (uint16_t*)nullptr);
    else hipLaunchKernelGGL(k_pack_conv_split<ElemBF16>, dim3(72 * 16 * 64 * 8 / 256), dim3(256), 0, (hipStream_t)stream, w, (uint16_t*)w_stream, (int)transposed, (unsigned int*)zero_u32, (uint16_t*)nullptr);
    HIPCHK(hipGetLastError());
    return SZ_OK;
}
// Both streams of one convolution in ONE launch (a train step packs every convolution's weights for its forward and for its backward-data pass): w_stream for the
// forward convolution, w_stream_t for backward-data (each 72*2048*16 bytes); zero_u32: optional device uint32[2], both set to 0 (the amax_bits slots of the two convolutions).
int sz_nn_pack_conv_split_both(const float* w, int32_t f16, void* w_stream, void* w_stream_t, void* zero_u32, void* stream) {
    if (!w || !w_stream || !w_stream_t) return SZ_ERR_INVALID;
    StreamDeviceGuard _guard(stream);
    if (f16) hipLaunchKernelGGL(k_pack_conv_split<ElemF16>, dim3(2 * 72 * 16 * 64 * 8 / 256), dim3(256), 0, (hipStream_t)stream, w, (uint16_t*)w_stream, 0, (unsigned int*)zero_u32, (uint16_t*)w_stream_t);
    else hipLaunchKernelGGL(k_pack_conv_split<ElemBF16>, dim3(2 * 72 * 16 * 64 * 8 / 256), dim3(256), 0, (hipStream_t)stream, w, (uint16_t*)w_stream, 0, (unsigned int*)zero_u32, (uint16_t*)w_stream_t);
    HIPCHK(hipGetLastError());
    return SZ_OK;
}

// Weight gradient of the 3x3 convolution (k_wgrad3x3_split + k_wgrad_reduce): gy, x device [n_boards,256,8,8] f32; amax_gy / amax_x: device uint32 holding the f32 bit
// pattern of max |gy| / max |x| (left behind by sz_nn_conv3x3_split_f32's amax_bits output on the same tensors); part: device scratch of 16*9*256*256 f32; dw: device
// [256,256,3,3] f32 (overwritten).
int sz_nn_wgrad3x3_split_f32(const float* gy, const float* x, const void* amax_gy, const void* amax_x, float* part, float* dw, int32_t n_boards, void* stream) {
    if (!gy || !x || !amax_gy || !amax_x || !part || !dw || n_boards <= 0) return SZ_ERR_INVALID;
    StreamDeviceGuard _guard(stream);
    static bool attr_flags[NN_MAX_DEVICES] = {};
    bool& attr_set = attr_flags[current_device_slot()];
    if (!attr_set) {
        HIPCHK(hipFuncSetAttribute((const void*)k_wgrad3x3_split, hipFuncAttributeMaxDynamicSharedMemorySize, WG_LDS_BYTES));
        attr_set = true;
    }
    const int KG = n_boards < 16 ? n_boards : 16;
    hipLaunchKernelGGL(k_wgrad3x3_split, dim3(16 * KG), dim3(256), WG_LDS_BYTES, (hipStream_t)stream, gy, x, (const unsigned int*)amax_gy, (const unsigned int*)amax_x, part, n_boards, KG);
    hipLaunchKernelGGL(k_wgrad_reduce, dim3(256), dim3(256), 0, (hipStream_t)stream, (const float*)part, dw, KG);
    HIPCHK(hipGetLastError());
    return SZ_OK;
}

// One call per direction of a training convolution (the host side of a train step is as busy as the GPU: every ctypes call and tensor allocation counts).
// forward: pack (both streams when w_stream_t is given, else the forward one) + convolution; amax2: optional device uint32[2] = {max|x|, max|gy|} bit patterns
// (zeroed here, slot 0 filled by this convolution, slot 1 by the backward call).
int sz_nn_conv3x3_train_fwd(const float* x, const float* w, int32_t f16, void* w_stream, void* w_stream_t, const float* zero256, float* y, int32_t n_boards, void* amax2, void* stream) {
    if (!x || !w || !w_stream || !zero256 || !y || n_boards <= 0) return SZ_ERR_INVALID;
    const int rc = w_stream_t ? sz_nn_pack_conv_split_both(w, f16, w_stream, w_stream_t, amax2, stream) : sz_nn_pack_conv_split_dev(w, 0, f16, w_stream, amax2, stream);
    if (rc != SZ_OK) return rc;
    return sz_nn_conv3x3_split_f32(x, w_stream, zero256, y, n_boards, f16, f16 ? amax2 : nullptr, stream);
}
// backward: gx = backward-data convolution of gy on w_stream_t (packed by the forward call), then dw = weight gradient (f16 operands; needs amax2 from the forward call).
// gx == NULL or dw == NULL skips that part; dw without gx needs amax2[1] = max|gy| filled by the caller.
int sz_nn_conv3x3_train_bwd(const float* gy, const float* x, const void* w_stream_t, const float* zero256, float* gx, void* amax2, float* part, float* dw, int32_t n_boards,
                            int32_t f16, void* stream) {
    if (!gy || n_boards <= 0 || (!gx && !dw)) return SZ_ERR_INVALID;
    if (gx) {
        if (!w_stream_t || !zero256) return SZ_ERR_INVALID;
        const int rc = sz_nn_conv3x3_split_f32(gy, w_stream_t, zero256, gx, n_boards, f16, (f16 && amax2) ? (void*)((unsigned int*)amax2 + 1) : nullptr, stream);
        if (rc != SZ_OK) return rc;
    }
    if (dw) {
        if (!x || !amax2 || !part || !f16) return SZ_ERR_INVALID;
        return sz_nn_wgrad3x3_split_f32(gy, x, (const unsigned int*)amax2 + 1, amax2, part, dw, n_boards, stream);
    }
    return SZ_OK;
}

}  // extern "C"
