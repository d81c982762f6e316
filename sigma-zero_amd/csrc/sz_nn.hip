// sz_nn.hip — hand-written CDNA4 MFMA kernels for the policy/value network of the reference
// (/root/reference/network.py:36-83 BasicBlock, :105-137 policyNN stem/tower, :141-174 heads).
//
//   k_tower16_bf16  THE SHIPPED PATH: stem + all BasicBlocks in one persistent launch (16x16x32 MFMA; one workgroup per CU takes
//                   2-board tiles through all 39 convolutions with the activations resident in LDS) — see its own header below
//   k_heads16_bf16  both heads from one read of the tower output (+ k_value_head for the 64->256->1 MLP)
//   building blocks / cross-checks, per-layer launches:
//   k_conv_bf16, k_conv16_bf16    one fused layer : conv3x3(pad 1) / conv1x1 + folded BatchNorm + bias (+ residual) (+ ReLU)
//   k_block_bf16, k_block16_bf16  one fused BasicBlock : relu(bn2(conv2(relu(bn1(conv1(x))))) + x), the intermediate activation
//                 never leaves the CU (it is written to LDS in exactly the layout the second conv reads)
// The notes below describe the first (32x32x16) per-layer kernels; the 16x16x32 path has its own section further down.
// NHWC bf16 in/out, f32 accumulate, C_out = 256, C_in in {128 (zero-padded 119-plane stem), 256}; one launch
// replaces MIOpen's igemm + batch_norm + clamp + add kernels of the torch graph.
//
// MI355X mapping (not a warp-tiling port):
//   * one workgroup = 4 waves = WGB boards (2 by default -> 68 KB of LDS -> TWO workgroups per CU, so one
//     workgroup's HBM phases hide partly under the other's MFMA phase; the first co-resident pair is phase-staggered);
//   * the boards' activations (WGB*64 positions x C_in) are loaded ONCE from HBM into LDS (row pitch
//     C_in*2+16 B: conflict-free ds_read_b128) and stay resident for all 9 taps — the im2col matrix is
//     never formed; a tap is just a per-lane row address (off-board taps read a zero row);
//   * D = W x Act^T on v_mfma_f32_32x32x16_bf16 with the WEIGHTS as the A operand (rows = out channels) and
//     activations as the B operand (cols = positions); every wave owns a distinct quarter of the output
//     channels (2 channel tiles x 2*WGB position tiles), so each weight fragment is fetched by one wave only;
//   * weights are pre-packed on the host in exact fragment order [tap][kstep][co_tile][lane][8]: a weight
//     fragment is one fully coalesced 1 KiB global_load_dwordx4 from L2 (1.18 MB/layer stays L2-resident),
//     prefetched 3 k-steps ahead through a 4-deep register ring (8-deep measured no faster); activations are
//     double-buffered one k-step ahead; the order is pinned with sched_barrier so the compiler's waits become
//     counted vmcnt/lgkmcnt: the K loop has NO workgroup barrier and no exposed memory latency;
//   * epilogue through LDS: (acc + bias) -> bf16 -> [pos][co] image, then whole 16-byte chunks are moved with
//     coalesced residual reads and stores (scattered 8-byte stores from the accumulator layout cost 20 %).
// `flags` bit 0 = ReLU; higher bits are timing-ablation / A-B switches used by tools/conv_bench.py only
// (2/4/8 skip load/store/K loop, 16 = 4-board workgroups, 32/64 + bits 8..15 = phase stagger, 0x10000 = no stagger,
// 0x200000/0x400000/0x800000 = streaming (non-temporal) tile loads / output stores / residual re-read in the fused block:
// measured with tools/block_ab.py, all-streaming +3 % slower (the residual re-read then misses), stores or residual alone within noise).
#include "sz_nn_common.h"

// K loop of the persistent tower: ONE explicit s_waitcnt in the last MFMA gap of every half-step (it carries no memory instruction) for everything the next
// half-step consumes — lgkmcnt(0) for its activation fragments, plus vmcnt(8) at the end of a k-step for the next k-step's weights (two k-steps of loads stay in
// flight).  hipcc otherwise puts a counted wait in front of each first use, i.e. into the very gaps that also issue a load: 1,276 -> 523 s_waitcnt in the kernel,
// BasicBlock 81,440 -> 80,196 cycles, launch 7.34 -> 7.24 ms (same box, interleaved: profiles/r03u_explicit_wait_ab.txt).  0 = off (A/B), 1 = LDS only.
#ifndef NN_EXPLICIT_WAIT
#define NN_EXPLICIT_WAIT 2
#endif
#ifndef NN_TAPGAP
#define NN_TAPGAP 1                                        // tap bookkeeping (table reads, row addresses) inside MFMA gaps; 0 = between the taps (A/B)
#endif


// ---- the K loop: acc[i][j] += W[tap,k] x Act[tap,k]^T over all taps and channels ----------------------------
// Software pipeline, pinned with sched_barrier so that hipcc cannot sink the prefetches to their uses:
//   issue { weights of k-step ks+PF (L2 -> ring), activations of ks+1 (LDS -> bfrag) } ; NI*NJ MFMAs of ks.
template <int CIN, int NTAPS, int WGB, bool PROBE16 = false>
__device__ __forceinline__ void conv_kloop(const unsigned char* lds, const uint4* __restrict__ w, f32x16 (&acc)[NN_NI][2 * WGB], bool skip) {
    constexpr int PITCH = CIN * 2 + 16;
    constexpr int KSTEPS = CIN / 16;
    constexpr int ZERO_ROW = WGB * 64;
    constexpr int NI = NN_NI, NJ = 2 * WGB;
    constexpr int RING = 4, PF = RING - 1;
    constexpr int TOTAL_KS = NTAPS * KSTEPS;
    constexpr int W_KSTEP_STRIDE = 8 * 64;                 // uint4 per (tap,kstep)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int p32 = lane & 31, h = lane >> 5;              // lane owns position p32 of each 32-position tile; h selects k 8..15
    const uint4* wbase = w + (size_t)(wave * NI) * 64 + lane;
#pragma unroll
    for (int i = 0; i < NI; i++)
#pragma unroll
        for (int j = 0; j < NJ; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    uint4 aring[RING][NI];
#pragma unroll
    for (int s = 0; s < PF; s++)
#pragma unroll
        for (int i = 0; i < NI; i++) aring[s][i] = wbase[(size_t)s * W_KSTEP_STRIDE + i * 64];
    auto tap_addr = [&](int tap, int j) -> int {           // LDS byte address of this lane's activation row
        const int dy = (NTAPS == 9) ? tap / 3 - 1 : 0, dx = (NTAPS == 9) ? tap % 3 - 1 : 0;
        int pos = (j & 1) * 32 + p32;                      // position inside its board (board = j >> 1)
        int y = (pos >> 3) + dy, x = (pos & 7) + dx;
        bool ok = (unsigned)y < 8u && (unsigned)x < 8u;
        int row = ok ? ((j >> 1) * 64 + y * 8 + x) : ZERO_ROW;
        return row * PITCH + h * 16;
    };
    int bcur[NJ], bnxt[NJ];
    bf16x8 bfrag[2][NJ];
#pragma unroll
    for (int j = 0; j < NJ; j++) { bcur[j] = tap_addr(0, j); bnxt[j] = bcur[j]; bfrag[0][j] = *(const bf16x8*)(lds + bcur[j]); }
    for (int tap = 0; tap < (skip ? 0 : NTAPS); tap++) {
        if (tap + 1 < NTAPS) {
#pragma unroll
            for (int j = 0; j < NJ; j++) bnxt[j] = tap_addr(tap + 1, j);
        }
#pragma unroll
        for (int kc = 0; kc < KSTEPS; kc++) {
            const int ks = tap * KSTEPS + kc;
            if (ks + PF < TOTAL_KS) {
#pragma unroll
                for (int i = 0; i < NI; i++) aring[(kc + PF) & (RING - 1)][i] = wbase[(size_t)(ks + PF) * W_KSTEP_STRIDE + i * 64];
            }
            if (kc + 1 < KSTEPS) {
#pragma unroll
                for (int j = 0; j < NJ; j++) bfrag[(kc + 1) & 1][j] = *(const bf16x8*)(lds + bcur[j] + (kc + 1) * 32);
            } else if (tap + 1 < NTAPS) {
#pragma unroll
                for (int j = 0; j < NJ; j++) bfrag[(kc + 1) & 1][j] = *(const bf16x8*)(lds + bnxt[j]);
            }
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < NI; i++) {
                bf16x8 a = __builtin_bit_cast(bf16x8, aring[kc & (RING - 1)][i]);
#pragma unroll
                for (int j = 0; j < NJ; j++) {
                    if constexpr (PROBE16) {
                        // timing probe only (numerically meaningless): the same operands through twice as many 16x16x32 MFMAs
#pragma unroll
                        for (int q = 0; q < 2; q++) {
                            f32x4 c4 = {acc[i][j][8 * q], acc[i][j][8 * q + 1], acc[i][j][8 * q + 2], acc[i][j][8 * q + 3]};
                            c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bfrag[kc & 1][j], c4, 0, 0, 0);
                            acc[i][j][8 * q] = c4[0]; acc[i][j][8 * q + 1] = c4[1]; acc[i][j][8 * q + 2] = c4[2]; acc[i][j][8 * q + 3] = c4[3];
                        }
                    } else {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bfrag[kc & 1][j], acc[i][j], 0, 0, 0);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int j = 0; j < NJ; j++) bcur[j] = bnxt[j];
    }
}

// ---- (acc + bias) [-> ReLU] -> bf16 -> LDS image [pos][co], pitch 528 B (= the C_in=256 activation layout) ---
template <int WGB>
__device__ __forceinline__ void acc_to_lds(unsigned char* lds, const f32x16 (&acc)[NN_NI][2 * WGB], const float* __restrict__ bias, bool relu) {
    constexpr int OPITCH = NN_COUT * 2 + 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int p32 = lane & 31, h = lane >> 5;
#pragma unroll
    for (int j = 0; j < 2 * WGB; j++) {
        const int row = (j >> 1) * 64 + (j & 1) * 32 + p32;
#pragma unroll
        for (int i = 0; i < NN_NI; i++) {
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int co = (wave * NN_NI + i) * 32 + 8 * g + 4 * h;
                f32x4 b4 = *(const f32x4*)(bias + co);
                float v0 = acc[i][j][4 * g + 0] + b4[0], v1 = acc[i][j][4 * g + 1] + b4[1];
                float v2 = acc[i][j][4 * g + 2] + b4[2], v3 = acc[i][j][4 * g + 3] + b4[3];
                if (relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
                uint2 o; o.x = pack_bf16x2(v0, v1); o.y = pack_bf16x2(v2, v3);
                *(uint2*)(lds + row * OPITCH + co * 2) = o;
            }
        }
    }
}

// ---- LDS image -> (+ residual) -> (ReLU) -> coalesced 16-byte NHWC stores ------------------------------------
// The residual is added to the bf16-rounded conv+bias value (torch's own bf16 graph rounds there too).
template <int WGB, int PAD = 16, bool NT = false, bool NTS = NT>
__device__ __forceinline__ void lds_to_out(const unsigned char* lds, const uint16_t* __restrict__ res, uint16_t* __restrict__ out, int board0, int n_boards, bool relu) {
    constexpr int OPITCH = NN_COUT * 2 + PAD;
    constexpr int OUT_CHUNKS = WGB * 64 * 32;              // 16-byte chunks of the output tile
    const int tid = threadIdx.x;
    const int valid = min(WGB, n_boards - board0) * 64 * 32;
    const uint4* res4 = res ? (const uint4*)(res + (size_t)board0 * 64 * NN_COUT) : nullptr;
    uint4* out4 = (uint4*)(out + (size_t)board0 * 64 * NN_COUT);
#pragma unroll 4
    for (int c = tid; c < OUT_CHUNKS; c += 256) {
        if (c >= valid) break;
        uint4 v = *(const uint4*)(lds + (c >> 5) * OPITCH + (c & 31) * 16);
        if (!res4 && !relu) {                              // plain copy (the persistent tower's output): element type does not matter
            if (NTS) st_stream(out4 + c, v); else out4[c] = v;
            continue;
        }
        float f[8] = {bf16_lo(v.x), bf16_hi(v.x), bf16_lo(v.y), bf16_hi(v.y), bf16_lo(v.z), bf16_hi(v.z), bf16_lo(v.w), bf16_hi(v.w)};
        if (res4) {
            uint4 r = NT ? ld_stream(res4 + c) : res4[c];
            f[0] += bf16_lo(r.x); f[1] += bf16_hi(r.x); f[2] += bf16_lo(r.y); f[3] += bf16_hi(r.y);
            f[4] += bf16_lo(r.z); f[5] += bf16_hi(r.z); f[6] += bf16_lo(r.w); f[7] += bf16_hi(r.w);
        }
        if (relu) {
#pragma unroll
            for (int k = 0; k < 8; k++) f[k] = fmaxf(f[k], 0.f);
        }
        uint4 o;
        o.x = pack_bf16x2(f[0], f[1]); o.y = pack_bf16x2(f[2], f[3]); o.z = pack_bf16x2(f[4], f[5]); o.w = pack_bf16x2(f[6], f[7]);
        if (NTS) st_stream(out4 + c, o); else out4[c] = o;
    }
}

// =================================================================================================================
// 16x16x32 MFMA path.  Same data movement and the same number of matrix-pipe cycles as the 32x32x16 path, but the chip
// holds a ~15 % higher clock on this shape under bf16 load (measured with the timing probe in tools/conv_bench.py:
// 0.212 vs 0.250 ms per conv at B = 4096; MI355X_MICROARCH.md "DVFS give-back" item 7).
//   wave tile 64 channels x WGB*64 positions = 4 channel tiles(16) x 4*WGB position tiles(16), 4 acc regs each;
//   one k-step = 32 channels; it is executed as two half-steps over the position halves (16 MFMAs = 256 cycles each):
//   weights (4 fragments per k-step) ride a 2-deep ring one k-step ahead, activations are double-buffered one
//   half-step ahead.  Weight order: [tap][k32][co_tile16][lane][8] (sz_nn_pack_weights16).
// =================================================================================================================

// first PF k-steps of a convolution's weight stream into the ring (issued early, e.g. under the previous layer's epilogue)
template <int RING>
__device__ __forceinline__ void conv_prefetch16(const uint4* __restrict__ w, uint4 (&aring)[RING][4]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t wlane = (uint32_t)((wave * 4) * 64 + lane) * 16u;
    const WSrc wr = wfrag_rsrc(w);
#pragma unroll
    for (int s = 0; s < RING - 1; s++)
#pragma unroll
        for (int i = 0; i < 4; i++) aring[s][i] = ld_wfrag(wr, (size_t)s * (16 * 64), wlane + i * 1024);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}


template <int CIN, int NTAPS, int WGB, int RING = 2, bool PREFETCHED = false, int ABL = 0 /* timing ablation: 1 = no weight loads, 2 = no LDS reads in the loop */,
          bool ACCUM = false /* add onto the accumulators as they are: no bias, no initialisation */,
          class E = ElemBF16 /* operand element: ElemBF16 or ElemF16 (sz_nn_common.h) */,
          class EPI = std::nullptr_t /* callable (p, stage): stage -1..3 of the epilogue of accumulator tile p = i*NH + j of the FIRST position half (EpiTile16 /
                                        EpiResidual16); when given, the LAST tap runs its two position halves one after the other and the epilogue of the first
                                        half rides in the MFMA gaps of the second */>
__device__ __forceinline__ void conv_kloop16(const unsigned char* lds, const uint4* __restrict__ w, f32x4 (&acc)[4][4 * WGB], bool skip, bool wprobe = false,
                                             uint4 (*ring_in)[4] = nullptr, const int img_off = 0 /* byte offset of the image inside `lds` */,
                                             const float* __restrict__ bias = nullptr /* accumulators start at the bias (C layout: channel = 16*tile + 4*(lane>>4) + reg) */,
                                             const int* addr_tab = nullptr /* optional LDS table [NTAPS][NJ][64] of conv_tap_addr16 values: a tap's addresses
                                                                              are then 8 ds_read_b32 instead of ~50 VALU instructions of coordinate arithmetic */,
                                             EPI epi0 = EPI()) {
    constexpr int PITCH = CIN * 2 + NN_PAD16;             // 34 slots of 16 B per row: (2p + kg) mod 16 is a permutation per lane group
    constexpr int KSTEPS = CIN / 32;                       // k32-steps per tap
    constexpr int NI = 4, NJ = 4 * WGB, NH = NJ / 2;       // channel tiles, position tiles, position tiles per half-step
    constexpr int TOTAL_KS = NTAPS * KSTEPS;
    const int W_KSTEP_STRIDE = wprobe ? 0 : 16 * 64;       // uint4 per (tap,k32); 0 = timing probe: every k-step re-reads the same (L1-hot) fragments
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int p16 = lane & 15, kg = lane >> 4;             // lane owns position p16 of each 16-position tile; kg selects k 8kg..8kg+7
    const uint32_t wlane = (uint32_t)((wave * NI) * 64 + lane) * 16u;   // the lane's constant byte offset inside a k-step's 16 fragments
    const WSrc wr = wfrag_rsrc(w);
    // The accumulators start at the bias: the MFMAs of the very first k-step take the bias quad as their C operand (tap 0 is peeled
    // off the tap loop for that), so no accumulator is ever initialised separately (128 v_accvgpr writes per convolution otherwise).
    f32x4 binit[NI];
#pragma unroll
    for (int i = 0; i < NI; i++) binit[i] = bias ? *(const f32x4*)(bias + (wave * NI + i) * 16 + 4 * kg) : f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr bool PEEL = RING == 4;                       // 512-register tower only: the peeled copy costs the 256-register kernels spills
    // 2-board tower: position tile 0 (board row 0) is idle under taps 0..2 (they read row -1), tile NJ-1 (row 7) under taps 6..8: no fragment
    // loads, no MFMAs for them there (tile_row above).  ABL & 4 switches it off (A/B timing build).
    constexpr bool SKIPROWS = PEEL && NTAPS == 9 && WGB == 2 && !(ABL & 4) && NN_ROWSKIP;
    if ((skip || !PEEL) && !ACCUM) {
#pragma unroll
        for (int i = 0; i < NI; i++)
#pragma unroll
            for (int j = 0; j < NJ; j++) acc[i][j] = binit[i];
    } else if (SKIPROWS && !ACCUM) {
#pragma unroll
        for (int i = 0; i < NI; i++) acc[i][0] = binit[i];   // tile 0 takes no MFMA under the first tap, so the bias cannot ride in as its C operand
    }
    constexpr int PF = RING - 1;                           // weight prefetch distance in k-steps (512 matrix-pipe cycles each)
    static_assert(KSTEPS % RING == 0, "ring slots must be compile-time indices");
    uint4 aring[RING][NI];
#pragma unroll
    for (int s = 0; s < PF; s++)
#pragma unroll
        for (int i = 0; i < NI; i++) {
            if constexpr (PREFETCHED) aring[s][i] = ring_in[s][i];
            else aring[s][i] = ld_wfrag(wr, (size_t)s * W_KSTEP_STRIDE, wlane + i * 1024);
        }
    // the image's offset is folded into the per-lane row address, so the k offset still fits the 16-bit immediate of ds_read for the
    // second image of the persistent tower, which sits beyond 64 KB (otherwise every read pays a v_add)
    auto tap_addr = [&](int tap, int j) -> int {
        // Off-board taps read zeros.  ds_read_b128 is conflict-free when the 16 lanes of a group hit 16 different 16-byte slots mod 256 B;
        // a valid lane's slot is (2*row + kg + 4*kc) mod 16 and the rows of a group are consecutive, which makes that a permutation.  An
        // off-board lane therefore reads the zero region (768 B, 256-B aligned) at the slot its VIRTUAL row would have had, instead of
        // one shared zero row that collides with some valid lane's slot.
        // returned WITHOUT the image offset: a table value must not be touched where it is loaded (that would drain the LDS queue at the
        // head of the tap); abs_addr() adds the offset where the address is first used, at the end of the tap
        return addr_tab ? addr_tab[(tap * NJ + j) * 64 + lane] : conv_tap_addr16<PITCH, NTAPS, WGB>(tap, j, p16, kg);
    };
    // The stem (C_in = 128: row pitch 288 B) has no table of its own (LDS is full); given the 256-channel images' table it converts an entry instead of
    // redoing the coordinate arithmetic (~50 VALU per tile and tap: the stem's K loop took 33k cycles for 17k of MFMAs, tools/tower_stamps.py tile stamps):
    //   valid entry  row*544 + 16*kg  ->  row*288 + 16*kg = entry - 256*floor(entry / 544);   zero-region entry: same slot behind the 288-B rows
    constexpr int TABPITCH = NN_COUT * 2 + NN_PAD16;
    constexpr bool CONVERT = PITCH != TABPITCH;                 // only together with addr_tab
    auto tab_convert = [&](int rel) -> int {
        if constexpr (!CONVERT) return rel;
        else {
            const int row = (int)__umulhi((unsigned)rel, 7895161u);        // floor(rel / 544), exact for rel < 2^20 (7895161 = ceil(2^32 / 544))
            return rel >= WGB * 64 * TABPITCH ? rel - WGB * 64 * (TABPITCH - PITCH) : rel - row * (TABPITCH - PITCH);
        }
    };
    // The LDS base of the dynamic shared array is a link-time constant (0 here) that hipcc cannot fold early: reading through `lds + offset` cost one
    // v_add_u32 v, 0, v in front of EVERY ds_read_b128 (64 per tap).  The base therefore goes into the row address once per tap, together with the
    // image offset, and the fragments are read through LDS-address-space pointers built from that integer: address + immediate offset, no VALU.
    const int lds_base = (int)(uint32_t)(uintptr_t)lds;
    auto abs_addr = [&](int rel) -> int {
        int a = lds_base + img_off + rel;
        asm volatile("" : "+v"(a));                        // opaque: keeps the sum inside the VGPR (hipcc otherwise re-associates it into a per-read v_add)
        return a;
    };
    auto LD = [](int addr) -> bf16x8 { return *(const __attribute__((address_space(3))) bf16x8*)(uint32_t)addr; };
    int bcur[NJ], bnxt[NJ];
    bf16x8 bfrag[2][NH];
#pragma unroll
    for (int j = 0; j < NJ; j++) { bnxt[j] = addr_tab ? tab_convert(tap_addr(0, j)) : tap_addr(0, j); bcur[j] = abs_addr(bnxt[j]); }
#pragma unroll
    for (int j = 0; j < NH; j++)
        if (!(SKIPROWS && j == 0)) bfrag[0][j] = LD(bcur[j]);
    auto tap_body = [&](const int tap, auto first_tag, auto skip_tag) {
        constexpr bool FIRST = decltype(first_tag)::value;
        constexpr int SK = decltype(skip_tag)::value;          // SKIPROWS: 1 = position tile 0 idle under this tap (dy = -1), 2 = the last position tile idle (dy = +1)
        // Tap bookkeeping inside MFMA gaps (persistent tower with the address table): the next tap's NJ table entries are read in the load-free gaps of the
        // first k-step's second half-step, the NJ row addresses are formed in those of the last k-step's — one instruction per gap — instead of a
        // block of ~15 instructions between two taps with the matrix pipe idle (NN_TAPGAP=0: A/B).
        constexpr bool TAPGAP = NN_TAPGAP && PEEL && NN_ILV && NH + NJ <= NI * (NH - 1);
        const bool in_gaps = TAPGAP && addr_tab != nullptr;
        if (tap + 1 < NTAPS && !in_gaps) {
#pragma unroll
            for (int j = 0; j < NJ; j++) bnxt[j] = addr_tab ? tab_convert(tap_addr(tap + 1, j)) : tap_addr(tap + 1, j);
        }
#pragma unroll
        for (int kc = 0; kc < KSTEPS; kc++) {
            const int ks = tap * KSTEPS + kc;
#pragma unroll
            for (int hs = 0; hs < 2; hs++) {
                if (!NN_ILV) {
                    if (hs == 0 && ks + PF < TOTAL_KS) {        // weights PF k-steps ahead (slot freed by the previous half-step)
#pragma unroll
                        for (int i = 0; i < NI; i++) aring[(kc + PF) & (RING - 1)][i] = ld_wfrag(wr, (size_t)(ks + PF) * W_KSTEP_STRIDE, wlane + i * 1024);
                    }
                    // activations of the next half-step
                    if (hs == 0) {
#pragma unroll
                        for (int j = 0; j < NH; j++) if (!(SK == 2 && j == NH - 1)) bfrag[1][j] = LD(bcur[NH + j] + kc * 64);
                    } else if (kc + 1 < KSTEPS) {
#pragma unroll
                        for (int j = 0; j < NH; j++) if (!(SK == 1 && j == 0)) bfrag[0][j] = LD(bcur[j] + (kc + 1) * 64);
                    } else if (tap + 1 < NTAPS) {
#pragma unroll
                        for (int j = 0; j < NH; j++) if (!(SK == 1 && j == 0 && tap + 1 < 3)) bfrag[0][j] = LD(abs_addr(bnxt[j]));
                    }
                    asm volatile("" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int i = 0; i < NI; i++) {
                    bf16x8 a = __builtin_bit_cast(bf16x8, aring[kc & (RING - 1)][i]);
#pragma unroll
                    for (int j = 0; j < NH; j++) {
                        const bool idle_hs = (SK == 1 && hs == 0) || (SK == 2 && hs == 1);            // this half-step has an idle position tile (border row)
                        const bool idle = idle_hs && j == (SK == 1 ? 0 : NH - 1);
                        if (!idle) acc[i][hs * NH + j] = E::mfma(a, bfrag[hs][j], (FIRST && kc == 0) ? binit[i] : acc[i][hs * NH + j]);
                        if (NN_ILV && !idle) {
                            // one memory instruction per MFMA gap (an MFMA leaves 8 of its 16 cycles for other issue): first the next
                            // half-step's activations (LDS), then - in the first half-step - the weights PF k-steps ahead (L2).  m counts the MFMAs
                            // actually issued (a skipped border tile has no gap: its memory instruction would land in its neighbour's)
                            const int NR = idle_hs ? NH - 1 : NH, m = i * NR + (idle_hs && SK == 1 ? j - 1 : j);
                            if (m < NH) {
                                if ((ABL & 2) || (SK == 2 && hs == 0 && m == NH - 1)) {}
                                else if (hs == 0) bfrag[1][m] = LD(bcur[NH + m] + kc * 64);
                                else if (kc + 1 < KSTEPS) { if (!(SK == 1 && m == 0)) bfrag[0][m] = LD(bcur[m] + (kc + 1) * 64); }
                                else if (tap + 1 < NTAPS) { if (!(SK == 1 && m == 0 && tap + 1 < 3)) bfrag[0][m] = LD(abs_addr(bnxt[m])); }   // tile 0 is needed again from tap 3 on
                            } else if (m < NH + NI && hs == 0) {
                                if (ABL & 1) {}
                                else if (ks + PF < TOTAL_KS)
                                    aring[(kc + PF) & (RING - 1)][m - NH] = ld_wfrag(wr, (size_t)(ks + PF) * W_KSTEP_STRIDE, wlane + (m - NH) * 1024);
                            } else if (TAPGAP && hs == 1 && m >= NH && m < NH + NJ) {
                                if (in_gaps) {
                                    if (kc == 0) bnxt[m - NH] = tap_addr(tap + 1 < NTAPS ? tap + 1 : tap, m - NH);
                                    else if (CONVERT && kc == 1) bnxt[m - NH] = tab_convert(bnxt[m - NH]);
                                    else if (kc == KSTEPS - 1) bcur[m - NH] = abs_addr(bnxt[m - NH]);
                                }
                            }
#if NN_EXPLICIT_WAIT
                            // last gap of a half-step: one wait for everything the next half-step consumes (see NN_EXPLICIT_WAIT)
                            if (PEEL && m == NI * NR - 1) {
                                if (NN_EXPLICIT_WAIT >= 2 && hs == 1) __builtin_amdgcn_s_waitcnt(0x0078);      // vmcnt(8) lgkmcnt(0): the next k-step's weights (two k-steps of loads stay in flight)
                                else __builtin_amdgcn_s_waitcnt(0xC07F);                                     // lgkmcnt(0)
                            }
#endif
                            asm volatile("" ::: "memory");
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (!in_gaps) {
#pragma unroll
            for (int j = 0; j < NJ; j++) bcur[j] = abs_addr(bnxt[j]);
        }
    };
    constexpr bool SPLIT = !std::is_same<EPI, std::nullptr_t>::value && NTAPS == 9 && NN_ILV && PEEL && KSTEPS % RING == 0;
    // Last tap with its position halves in sequence (SPLIT): phase A finishes the accumulators of the first half (board 0), phase B runs the second
    // half and carries, in two of every 16 MFMA gaps, one tile of the first half's epilogue (accumulator read-out, bf16 pack, ReLU, LDS write) —
    // work that is otherwise exposed after the K loop.  Phase B reads the tap's weights a second time (virtual k-steps TOTAL_KS .. TOTAL_KS+KSTEPS-1
    // of the ring); activations alternate between the two fragment buffers by k-step parity.
    auto last_tap_split = [&]() {
        if constexpr (SPLIT) {
            constexpr int tap = NTAPS - 1;
#pragma unroll
            for (int kc = 0; kc < KSTEPS; kc++) {                            // phase A
                const int ks = tap * KSTEPS + kc;
#pragma unroll
                for (int i = 0; i < NI; i++) {
                    bf16x8 a = __builtin_bit_cast(bf16x8, aring[kc & (RING - 1)][i]);
#pragma unroll
                    for (int j = 0; j < NH; j++) {
                        acc[i][j] = E::mfma(a, bfrag[kc & 1][j], acc[i][j]);
                        const int m = i * NH + j;
                        if (m < NH) {
                            if (ABL & 2) {}
                            else if (kc + 1 < KSTEPS) bfrag[(kc + 1) & 1][m] = LD(bcur[m] + (kc + 1) * 64);
                            else if (!(SKIPROWS && m == NH - 1)) bfrag[(kc + 1) & 1][m] = LD(bcur[NH + m]);            // first fragments of phase B
                        } else if (m < NH + NI) {
                            const int vks = ks + PF, wks = vks < TOTAL_KS ? vks : vks - KSTEPS;              // phase B re-reads this tap's weights
                            if (!(ABL & 1)) aring[(kc + PF) & (RING - 1)][m - NH] = ld_wfrag(wr, (size_t)wks * W_KSTEP_STRIDE, wlane + (m - NH) * 1024);
                        } else if (kc == KSTEPS - 1 && (m == 2 * NH || m == 3 * NH)) {
                            epi0(m == 2 * NH ? 0 : 1, -1);                                                     // stage -1: operand prefetch for the first two tiles
                        }
#if NN_EXPLICIT_WAIT >= 2
                        if (m == NI * NH - 1) __builtin_amdgcn_s_waitcnt(0x0078);                              // vmcnt(8) lgkmcnt(0), as in tap_body
#endif
                        asm volatile("" ::: "memory");
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
#pragma unroll
            for (int kc = 0; kc < KSTEPS; kc++) {                            // phase B
#pragma unroll
                for (int i = 0; i < NI; i++) {
                    bf16x8 a = __builtin_bit_cast(bf16x8, aring[kc & (RING - 1)][i]);
#pragma unroll
                    for (int j = 0; j < NH; j++) {
                        if (!(SKIPROWS && j == NH - 1))                                           // the last tap looks one row down: board row 7 reads only zeros
                            acc[i][NH + j] = E::mfma(a, bfrag[kc & 1][j], acc[i][NH + j]);
                        const int m = i * NH + j;
                        if (m < NH) {
                            if (!(ABL & 2) && kc + 1 < KSTEPS && !(SKIPROWS && m == NH - 1)) bfrag[(kc + 1) & 1][m] = LD(bcur[NH + m] + (kc + 1) * 64);
                        } else if (m < NH + NI) {
                            if (!(ABL & 1) && kc + PF < KSTEPS)
                                aring[(kc + PF) & (RING - 1)][m - NH] = ld_wfrag(wr, (size_t)(tap * KSTEPS + kc + PF) * W_KSTEP_STRIDE, wlane + (m - NH) * 1024);
                        } else {
                            // 16 tiles of the first half over 8 k-steps: two per k-step, each in four stages of one or two instructions
                            // (gaps 8..11 and 12..15), so that a gap never carries more than an MFMA leaves free
                            const int p = 2 * kc + (m >= 3 * NH ? 1 : 0);
                            if (p < NI * NH) epi0(p, (m - 2 * NH) & 3);
                        }
                        asm volatile("" ::: "memory");
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        }
    };
    if (!skip) {
        using SK0 = std::integral_constant<int, 0>;
        if constexpr (SKIPROWS) {                              // taps 0..2 look one row up (tile 0 idle), 3..5 stay in the row, 6..8 look one row down (last tile idle)
            using SK1 = std::integral_constant<int, 1>; using SK2 = std::integral_constant<int, 2>;
            tap_body(0, std::integral_constant<bool, !ACCUM>{}, SK1{});
            for (int tap = 1; tap < 3; tap++) tap_body(tap, std::false_type{}, SK1{});
            for (int tap = 3; tap < 6; tap++) tap_body(tap, std::false_type{}, SK0{});
            for (int tap = 6; tap < (SPLIT ? NTAPS - 1 : NTAPS); tap++) tap_body(tap, std::false_type{}, SK2{});
            last_tap_split();
        } else if constexpr (PEEL) {
            tap_body(0, std::integral_constant<bool, !ACCUM>{}, SK0{});
            for (int tap = 1; tap < (SPLIT ? NTAPS - 1 : NTAPS); tap++) tap_body(tap, std::false_type{}, SK0{});
            last_tap_split();
        } else {
            for (int tap = 0; tap < NTAPS; tap++) tap_body(tap, std::false_type{}, SK0{});
        }
    }
}

// K loop of the ONE-board workgroup form (256 -> 256 channels, 3x3, persistent tower at up to #CUs boards): the same k-step order and the same MFMAs as conv_kloop16
// — a board's result is bit-identical in both forms — but scheduled for four position tiles.  conv_kloop16 splits a k-step into two half-steps of NH position tiles; with
// NH = 2 that is 8 MFMAs (128 cycles) between a fragment's ds_read_b128 and its use, a wait twice per k-step, and no room for the tap bookkeeping in the load-free gaps
// (stamped: 24.2k cycles for 18.4k of MFMAs, 22.0k with every load removed; tools/tower_stamps.py 128 1 2 3 4).  Here a k-step is ONE phase of 16 MFMAs:
//   gaps 0..3   the next k-step's four activation fragments (LDS; double-buffered by k-step parity: a whole k-step = 256 cycles ahead of their use)
//   gaps 4..7   the weights PF = 3 k-steps ahead (L2, through the buffer descriptor)
//   gaps 8..11  tap bookkeeping: k-step 0 reads the next tap's table entries, the last k-step forms the row addresses
//   gap 15      one wait for everything the next k-step consumes.
// ring_in: the first PF k-steps of weights (conv_prefetch16 under the previous epilogue).  ABL: 1 = no weight loads, 2 = no LDS reads (timing builds, results garbage).
#ifndef NN_ONE_RING
#define NN_ONE_RING 4                                      // weight ring of the one-board K loop: 4 or 8 slots (prefetch distance 3 or 7 k-steps of 256 cycles)
#endif
#ifndef NN_ONE_WHOT
#define NN_ONE_WHOT 0                                      // timing probe: 1 = every k-step re-reads the first k-step's (cache-hot) weights; results garbage
#endif
template <int ABL = 0, class E = ElemBF16>
__device__ __forceinline__ void conv_kloop16_one(const unsigned char* lds, const uint4* __restrict__ w, f32x4 (&acc)[4][4], uint4 (*ring_in)[4] /* [NN_ONE_RING - 1][4] */, const int img_off,
                                                 const float* __restrict__ bias, const int* addr_tab) {
    constexpr int KSTEPS = NN_COUT / 32, NI = 4, NJ = 4, NTAPS = 9, RING = NN_ONE_RING, PF = RING - 1, TOTAL_KS = NTAPS * KSTEPS;
    constexpr int W_KSTEP_STRIDE = NN_ONE_WHOT ? 0 : 16 * 64;
    static_assert(KSTEPS % RING == 0, "ring slots must be compile-time indices");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, kg = lane >> 4;
    const uint32_t wlane = (uint32_t)((wave * NI) * 64 + lane) * 16u;
    const WSrc wr = wfrag_rsrc(w);
    f32x4 binit[NI];
#pragma unroll
    for (int i = 0; i < NI; i++) binit[i] = *(const f32x4*)(bias + (wave * NI + i) * 16 + 4 * kg);
    uint4 aring[RING][NI];
#pragma unroll
    for (int s_ = 0; s_ < PF; s_++)
#pragma unroll
        for (int i = 0; i < NI; i++) aring[s_][i] = ring_in[s_][i];
    const int lds_base = (int)(uint32_t)(uintptr_t)lds;
    auto abs_addr = [&](int rel) -> int {
        int a = lds_base + img_off + rel;
        asm volatile("" : "+v"(a));                        // see conv_kloop16: keeps base + offset + row in one VGPR, the k offset in the ds_read immediate
        return a;
    };
    auto LD = [](int addr) -> bf16x8 { return *(const __attribute__((address_space(3))) bf16x8*)(uint32_t)addr; };
    int bcur[NJ], bnxt[NJ];
    bf16x8 bfrag[2][NJ];
#pragma unroll
    for (int j = 0; j < NJ; j++) { bnxt[j] = addr_tab[j * 64 + lane]; bcur[j] = abs_addr(bnxt[j]); }
#pragma unroll
    for (int j = 0; j < NJ; j++) bfrag[0][j] = LD(bcur[j]);
    auto tap_body = [&](const int tap, auto first_tag) {
        constexpr bool FIRST = decltype(first_tag)::value;
#pragma unroll
        for (int kc = 0; kc < KSTEPS; kc++) {
            const int ks = tap * KSTEPS + kc;
#pragma unroll
            for (int i = 0; i < NI; i++) {
                const bf16x8 a = __builtin_bit_cast(bf16x8, aring[kc & (RING - 1)][i]);
#pragma unroll
                for (int j = 0; j < NJ; j++) {
                    acc[i][j] = E::mfma(a, bfrag[kc & 1][j], (FIRST && kc == 0) ? binit[i] : acc[i][j]);
                    const int m = i * NJ + j;
                    if (m < NJ) {
                        if (ABL & 2) {}
                        else if (kc + 1 < KSTEPS) bfrag[(kc + 1) & 1][m] = LD(bcur[m] + (kc + 1) * 64);
                        else if (tap + 1 < NTAPS) bfrag[(kc + 1) & 1][m] = LD(abs_addr(bnxt[m]));
                    } else if (m < NJ + NI) {
                        if (!(ABL & 1) && ks + PF < TOTAL_KS)
                            aring[(kc + PF) & (RING - 1)][m - NJ] = ld_wfrag(wr, (size_t)(ks + PF) * W_KSTEP_STRIDE, wlane + (m - NJ) * 1024);
                    } else if (m < 2 * NJ + NI) {
                        if (kc == 0) bnxt[m - NJ - NI] = addr_tab[((tap + 1 < NTAPS ? tap + 1 : tap) * NJ + (m - NJ - NI)) * 64 + lane];
                        else if (kc == KSTEPS - 1) bcur[m - NJ - NI] = abs_addr(bnxt[m - NJ - NI]);
                    } else if (m == NI * NJ - 1) {
                        if (RING == 4) __builtin_amdgcn_s_waitcnt(0x0078);  // vmcnt(8) lgkmcnt(0): the next k-step's weights and activations (two k-steps of weight loads stay in flight)
                        else __builtin_amdgcn_s_waitcnt(0x4078);                // vmcnt(24) lgkmcnt(0): six k-steps of weight loads stay in flight (vmcnt = simm16[15:14]:[3:0])
                    }
                    asm volatile("" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    };
    tap_body(0, std::true_type{});
    for (int tap = 1; tap < NTAPS; tap++) tap_body(tap, std::false_type{});
}

// Byte offset of the 8-byte epilogue slot (image row, 4 channels from co) inside a 16x16x32-path image.  The 16 lanes of a ds_write_b64 group hold 16 rows
// at one channel offset; at the 544-B pitch 8 B x (68 row) mod 128 B takes 4 values, so the store is 4-way bank-conflicted (SQ_LDS_BANK_CONFLICT: 17 % of the
// LDS-array cycles of the tower).  NN_EPI_NOCONFLICT=1 is a TIMING-ONLY A/B build (results garbage): the same stores and residual reads go to conflict-free
// addresses (the 16 lanes contiguous), which prices the conflicts in in-kernel cycles (tools/tower_stamps.py; profiles/r03g_*).
#ifndef NN_EPI_NOCONFLICT
#define NN_EPI_NOCONFLICT 0
#endif
__device__ __forceinline__ int epi_slot16(int row, int co) {
#if NN_EPI_NOCONFLICT
    return ((row >> 3) & 7) * 8192 + (co >> 2) * 128 + (((row >> 6) << 3) | (row & 7)) * 8;
#else
    return row * (NN_COUT * 2 + NN_PAD16) + co * 2;
#endif
}

// Staged epilogue functors for conv_kloop16's split last tap: tile p = i*4 + j (channel tile i, position tile j of the first half) in stages
//   -1: operand prefetch (residual only)   0: accumulator read-out   1, 2: bf16 pack (+ residual) + ReLU of one register pair each   3: LDS write
template <int WGB, class E = ElemBF16> struct EpiTile16 {
    unsigned char* img; const f32x4 (&acc)[4][4 * WGB]; f32x4 tv; uint2 o;
    __device__ __forceinline__ EpiTile16(unsigned char* img_, const f32x4 (&acc_)[4][4 * WGB]) : img(img_), acc(acc_) {}
    __device__ __forceinline__ void operator()(int p, int st) {
        const int i = p / (2 * WGB), j = p % (2 * WGB);
        if (st == 0) tv = acc[i][j];
        else if (st == 1) o.x = relu_bf16x2(E::pack2(tv[0], tv[1]));
        else if (st == 2) o.y = relu_bf16x2(E::pack2(tv[2], tv[3]));
        else if (st == 3) {
            const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
            const int row = tile_row<WGB>(j, lane & 15), co = (wave * 4 + i) * 16 + 4 * (lane >> 4);
            *(uint2*)(img + epi_slot16(row, co)) = o;
        }
    }
};
template <int WGB, class E = ElemBF16> struct EpiResidual16 {
    unsigned char* img; const f32x4 (&acc)[4][4 * WGB]; f32x4 tv; uint2 o; uint2 rr[3];     // residual operands ride two tiles ahead of their use
    __device__ __forceinline__ EpiResidual16(unsigned char* img_, const f32x4 (&acc_)[4][4 * WGB]) : img(img_), acc(acc_) {}
    __device__ __forceinline__ uint2* slot(int p) const {
        const int i = p / (2 * WGB), j = p % (2 * WGB);
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int row = tile_row<WGB>(j, lane & 15), co = (wave * 4 + i) * 16 + 4 * (lane >> 4);
        return (uint2*)(img + epi_slot16(row, co));
    }
    __device__ __forceinline__ void operator()(int p, int st) {
        const int i = p / (2 * WGB), j = p % (2 * WGB);
        if (st == -1) rr[p % 3] = *slot(p);
        else if (st == 0) { tv = acc[i][j]; if (p + 2 < 8 * WGB) rr[(p + 2) % 3] = *slot(p + 2); }
        else if (st == 1) o.x = relu_bf16x2(E::pack2(tv[0] + E::lo(rr[p % 3].x), tv[1] + E::hi(rr[p % 3].x)));
        else if (st == 2) o.y = relu_bf16x2(E::pack2(tv[2] + E::lo(rr[p % 3].y), tv[3] + E::hi(rr[p % 3].y)));
        else if (st == 3) *slot(p) = o;
    }
};

// epilogue of ONE accumulator tile (channel tile i, position tile j): relu?(acc) -> bf16 -> LDS image (the accumulators started at the bias)
template <int WGB, class E = ElemBF16>
__device__ __forceinline__ void acc_tile_to_lds16(unsigned char* lds, const f32x4 (&acc)[4][4 * WGB], int i, int j, bool relu) {
    constexpr int OPITCH = NN_COUT * 2 + NN_PAD16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = tile_row<WGB>(j, lane & 15), co = (wave * 4 + i) * 16 + 4 * (lane >> 4);
    const f32x4 v = acc[i][j];
    uint2 o;
    o.x = E::pack2(v[0], v[1]);
    o.y = E::pack2(v[2], v[3]);
    if (relu) { o.x = relu_bf16x2(o.x); o.y = relu_bf16x2(o.y); }
    *(uint2*)(lds + epi_slot16(row, co)) = o;
}
// the same with the residual: x <- relu(acc + x) in place on the LDS image (f32 add, one bf16 rounding)
template <int WGB, class E = ElemBF16>
__device__ __forceinline__ void acc_tile_residual16(unsigned char* xlds, const f32x4 (&acc)[4][4 * WGB], int i, int j) {
    constexpr int OPITCH = NN_COUT * 2 + NN_PAD16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = tile_row<WGB>(j, lane & 15), co = (wave * 4 + i) * 16 + 4 * (lane >> 4);
    const f32x4 v = acc[i][j];
    uint2* px = (uint2*)(xlds + epi_slot16(row, co));
    const uint2 r = *px;
    uint2 o;
    o.x = relu_bf16x2(E::pack2(v[0] + E::lo(r.x), v[1] + E::hi(r.x)));
    o.y = relu_bf16x2(E::pack2(v[2] + E::lo(r.y), v[3] + E::hi(r.y)));
    *px = o;
}

// bias == nullptr: the accumulators already started at the bias (conv_kloop16's `bias` argument)
template <int WGB, class E = ElemBF16>
__device__ __forceinline__ void acc_to_lds16(unsigned char* lds, const f32x4 (&acc)[4][4 * WGB], const float* __restrict__ bias, bool relu) {
    constexpr int OPITCH = NN_COUT * 2 + NN_PAD16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int p16 = lane & 15, kg = lane >> 4;
#pragma unroll
    for (int j = 0; j < 4 * WGB; j++) {
        const int row = tile_row<WGB>(j, p16);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int co = (wave * 4 + i) * 16 + 4 * kg;    // C layout: col = lane&15 (position), row = 4*(lane>>4) + reg (channel)
            f32x4 v = acc[i][j];
            if (bias) v += *(const f32x4*)(bias + co);      // (no "+ 0.f" otherwise: it is not a no-op for -0 and would stay in the code)
            uint2 o;
            o.x = E::pack2(v[0], v[1]);
            o.y = E::pack2(v[2], v[3]);
            if (relu) { o.x = relu_bf16x2(o.x); o.y = relu_bf16x2(o.y); }
            *(uint2*)(lds + row * OPITCH + co * 2) = o;
        }
    }
}

// Phase stagger (speed only, never correctness): the two workgroups that share a CU are dispatched together and
// would run load / MFMA / store phases in lock-step.  Measured: workgroups b and b + #CUs share a CU (round-robin
// dispatch); HW_ID.WAVE_ID bit 0 selects the same set.  Delaying one of the first pair keeps later rounds out of phase.
__device__ __forceinline__ void phase_stagger(int flags, int n_cu) {
    const int stagger = (flags >> 8) & 0xFF;               // sleep units (~8k cycles each); 0 = off
    if (!stagger) return;
    bool second = false;
    if (flags & 32) second = ((int)blockIdx.x >= n_cu && (int)blockIdx.x < 2 * n_cu);
    if (flags & 64) second = ((int)blockIdx.x < 2 * n_cu) && ((__builtin_amdgcn_s_getreg(0x1804) & 1) != 0);
    if (__builtin_amdgcn_readfirstlane((int)second))
        for (int i = 0; i < stagger; i++) __builtin_amdgcn_s_sleep(127);
}

// in  : [n_boards][64][CIN]  bf16 (NHWC)            w : packed fragments (see sz_nn_pack_weights)
// out : [n_boards][64][256]  bf16 (NHWC)            bias : [256] f32 (BN folded)      res : optional residual
template <int CIN, int NTAPS, int WGB /* boards per workgroup: 4 -> 1 workgroup/CU, 2 -> 2 workgroups/CU */>
__global__ __launch_bounds__(256, (WGB == 2 ? 2 : 1)) void k_conv_bf16(const uint16_t* __restrict__ in, const uint4* __restrict__ w, const float* __restrict__ bias,
                                                      const uint16_t* __restrict__ res, uint16_t* __restrict__ out, int n_boards, int flags, int n_cu) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int board0 = blockIdx.x * WGB;
    stage_tile<CIN, WGB>(lds, in, board0, n_boards, (flags & 2) != 0);
    __syncthreads();
    if (WGB == 2) phase_stagger(flags, n_cu);
    f32x16 acc[NN_NI][2 * WGB];
    if (CIN == 256 && NTAPS == 9 && WGB == 2 && (flags & 0x20000)) conv_kloop<CIN, NTAPS, WGB, true>(lds, w, acc, (flags & 8) != 0);
    else conv_kloop<CIN, NTAPS, WGB>(lds, w, acc, (flags & 8) != 0);
    __syncthreads();                                       // all waves are done reading the activation tile
    if (!((flags & 4) && acc[0][0][0] != 12345.f)) acc_to_lds<WGB>(lds, acc, bias, false);
    __syncthreads();
    if (!(flags & 4)) lds_to_out<WGB>(lds, res, out, board0, n_boards, (flags & 1) != 0);
}

// One whole BasicBlock (network.py:66-83) per launch: out = relu(conv2(relu(conv1(x)+b1)) + b2 + x).
// conv1's result is written to LDS in the very layout conv2 reads, so it never travels to HBM: one tile load and one
// tile store per block instead of two of each, and half the launches.
template <int WGB>
__global__ __launch_bounds__(256, (WGB == 2 ? 2 : 1)) void k_block_bf16(const uint16_t* __restrict__ in, const uint4* __restrict__ w1, const float* __restrict__ b1,
                                                       const uint4* __restrict__ w2, const float* __restrict__ b2, uint16_t* __restrict__ out,
                                                       int n_boards, int flags, int n_cu) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int board0 = blockIdx.x * WGB;
    stage_tile<256, WGB>(lds, in, board0, n_boards, false);
    __syncthreads();
    if (WGB == 2) phase_stagger(flags, n_cu);
    f32x16 acc[NN_NI][2 * WGB];
    conv_kloop<256, 9, WGB>(lds, w1, acc, false);
    __syncthreads();
    acc_to_lds<WGB>(lds, acc, b1, true);                   // t = relu(bn1(conv1(x))), bf16, in place of x
    __syncthreads();
    conv_kloop<256, 9, WGB>(lds, w2, acc, false);
    __syncthreads();
    acc_to_lds<WGB>(lds, acc, b2, false);
    __syncthreads();
    lds_to_out<WGB>(lds, in, out, board0, n_boards, true); // + x (re-read, still L2/MALL-resident), ReLU
}

template <int CIN, int NTAPS, int WGB>
__global__ __launch_bounds__(256, (WGB == 2 ? 2 : 1)) void k_conv16_bf16(const uint16_t* __restrict__ in, const uint4* __restrict__ w, const float* __restrict__ bias,
                                                        const uint16_t* __restrict__ res, uint16_t* __restrict__ out, int n_boards, int flags, int n_cu) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int board0 = blockIdx.x * WGB;
    if (CIN == 128 && (flags & SZ_NN_IN_BITS)) stage_tile_bits<WGB, NN_PAD16>(lds, in, board0, n_boards);
    else stage_tile<CIN, WGB, NN_PAD16>(lds, in, board0, n_boards, (flags & 2) != 0);
    __syncthreads();
    if (WGB == 2) phase_stagger(flags, n_cu);
    f32x4 acc[4][4 * WGB];
    conv_kloop16<CIN, NTAPS, WGB>(lds, w, acc, (flags & 8) != 0, (flags & 0x100000) != 0, nullptr, 0, bias);
    __syncthreads();
    if (!((flags & 4) && acc[0][0][0] != 12345.f)) acc_to_lds16<WGB>(lds, acc, nullptr, false);
    __syncthreads();
    if (!(flags & 4)) lds_to_out<WGB, NN_PAD16>(lds, res, out, board0, n_boards, (flags & 1) != 0);
}

template <int WGB>
__global__ __launch_bounds__(256, (WGB == 2 ? 2 : (WGB == 1 ? 3 : 1))) void k_block16_bf16(const uint16_t* __restrict__ in, const uint4* __restrict__ w1, const float* __restrict__ b1,
                                                         const uint4* __restrict__ w2, const float* __restrict__ b2, uint16_t* __restrict__ out,
                                                         int n_boards, int flags, int n_cu) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int board0 = blockIdx.x * WGB;
    const bool nt = (flags & 0x200000) != 0;               // A/B: streaming tile loads/stores
    if (nt) stage_tile<256, WGB, NN_PAD16, true>(lds, in, board0, n_boards, false);
    else stage_tile<256, WGB, NN_PAD16>(lds, in, board0, n_boards, false);
    __syncthreads();
    if (WGB == 2) phase_stagger(flags, n_cu);
    f32x4 acc[4][4 * WGB];
    conv_kloop16<256, 9, WGB>(lds, w1, acc, false, false, nullptr, 0, b1);
    __syncthreads();
    acc_to_lds16<WGB>(lds, acc, nullptr, true);
    __syncthreads();
    conv_kloop16<256, 9, WGB>(lds, w2, acc, false, false, nullptr, 0, b2);
    __syncthreads();
    acc_to_lds16<WGB>(lds, acc, nullptr, false);
    __syncthreads();
    if (nt) lds_to_out<WGB, NN_PAD16, true>(lds, in, out, board0, n_boards, true);
    else if ((flags & 0xC00000) == 0xC00000) lds_to_out<WGB, NN_PAD16, true, true>(lds, in, out, board0, n_boards, true);
    else if (flags & 0x400000) lds_to_out<WGB, NN_PAD16, false, true>(lds, in, out, board0, n_boards, true);    // A/B: streaming stores only
    else if (flags & 0x800000) lds_to_out<WGB, NN_PAD16, true, false>(lds, in, out, board0, n_boards, true);    // A/B: streaming residual re-read only
    else lds_to_out<WGB, NN_PAD16>(lds, in, out, board0, n_boards, true);
}

// =================================================================================================================
// Heads (network.py:141-174).  Small, memory-bound kernels that replace ~12 torch launches per forward.
//   k_policy_head : logits[73][64] = Wp2 x t^T + b  (t = relu(bn(conv_p1(x))) from k_conv16<256,1>), softmax over the
//                   4672 logits, written as f32 in the reference's flatten order [plane*64 + pos].  One wave per board:
//                   5 channel tiles(16) x 4 position tiles(16) of v_mfma_f32_16x16x32_bf16, K = 256.
//   k_value_head  : v = tanh(fc2(relu(fc1(relu(bn(conv_v1(x)))))));  one wave per board, lane = position.
// =================================================================================================================
__global__ __launch_bounds__(256) void k_policy_head(const uint16_t* __restrict__ t, const uint4* __restrict__ w, const float* __restrict__ bias,
                                                      float* __restrict__ probs, int n_boards, int do_softmax) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int board = blockIdx.x * 4 + wave;
    if (board >= n_boards) return;
    const int p16 = lane & 15, kg = lane >> 4;
    f32x4 acc[5][4];
#pragma unroll
    for (int i = 0; i < 5; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const uint16_t* tb = t + (size_t)board * 64 * 256;
#pragma unroll 2
    for (int kc = 0; kc < 8; kc++) {
        bf16x8 a[5], b[4];
#pragma unroll
        for (int i = 0; i < 5; i++) a[i] = __builtin_bit_cast(bf16x8, w[(size_t)(kc * 5 + i) * 64 + lane]);
#pragma unroll
        for (int j = 0; j < 4; j++) b[j] = *(const bf16x8*)(tb + (size_t)(j * 16 + p16) * 256 + kc * 32 + kg * 8);
#pragma unroll
        for (int i = 0; i < 5; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    // lane holds logits for position j*16+p16 and channels i*16 + 4*kg + r
    float mx = -3.0e38f;
#pragma unroll
    for (int i = 0; i < 5; i++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int co = i * 16 + 4 * kg + r;
            const float bv = co < 73 ? bias[co] : 0.f;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                acc[i][j][r] += bv;
                if (co < 73) mx = fmaxf(mx, acc[i][j][r]);
            }
        }
    float sum = 0.f;
    if (do_softmax) {
        for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
#pragma unroll
        for (int i = 0; i < 5; i++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int co = i * 16 + 4 * kg + r;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    float e = co < 73 ? __expf(acc[i][j][r] - mx) : 0.f;
                    acc[i][j][r] = e;
                    sum += e;
                }
            }
        for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off);
    }
    const float inv = do_softmax ? 1.0f / sum : 1.0f;
    float* pb = probs + (size_t)board * 4672;
#pragma unroll
    for (int i = 0; i < 5; i++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int co = i * 16 + 4 * kg + r;
            if (co < 73) {
#pragma unroll
                for (int j = 0; j < 4; j++) pb[co * 64 + j * 16 + p16] = acc[i][j][r] * inv;
            }
        }
}

template <int VH_BOARDS_PER_WAVE /* 4 at large batches (fc_v1 staging amortised over 16 boards), 1 at small ones (4x the workgroups) */>
__global__ __launch_bounds__(256) void k_value_head(const uint16_t* __restrict__ x, const float* __restrict__ wv, float bv, const float* __restrict__ fc1_w /*[64][256]*/,
                                                     const float* __restrict__ fc1_b, const float* __restrict__ fc2_w, float fc2_b, float* __restrict__ value, int n_boards,
                                                     const float* __restrict__ v1_in /* optional: relu(bn(conv_v1)) [n_boards][64] from k_heads16_bf16 */) {
    // workgroup = 4 waves x 4 boards each; fc_v1's 64 KB weight matrix is staged in LDS once per workgroup
    extern __shared__ __attribute__((aligned(16))) unsigned char vh_lds[];
    float* w1s = (float*)vh_lds;                            // [64][256]
    float* v1s = w1s + 64 * 256;                            // [4 waves][4 boards][64] (direct-x mode uses [4 waves][64])
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int c = threadIdx.x; c < 64 * 256 / 4; c += 256) ((float4*)w1s)[c] = ((const float4*)fc1_w)[c];
    float wreg[8], b1r[4], w2r[4];
#pragma unroll
    for (int k = 0; k < 8; k++) wreg[k] = v1_in ? 0.f : wv[(lane & 31) * 8 + k];
#pragma unroll
    for (int q = 0; q < 4; q++) { b1r[q] = fc1_b[lane + 64 * q]; w2r[q] = fc2_w[lane + 64 * q]; }
    __syncthreads();
    if (v1_in) {
        // conv_v1 outputs given (fused heads kernel): the wave's 4 boards go through fc_v1 together, so every weight read from LDS
        // serves 4 boards
        const int board0 = (blockIdx.x * 4 + wave) * VH_BOARDS_PER_WAVE;
        if (board0 >= n_boards) return;
        float* vw = v1s + wave * (VH_BOARDS_PER_WAVE * 64);
#pragma unroll
        for (int bi = 0; bi < VH_BOARDS_PER_WAVE; bi++) vw[bi * 64 + lane] = (board0 + bi < n_boards) ? v1_in[(size_t)(board0 + bi) * 64 + lane] : 0.f;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_s_waitcnt(0xC07F);                 // lgkmcnt(0): the wave's own LDS writes have landed
        float h[VH_BOARDS_PER_WAVE][4];
#pragma unroll
        for (int bi = 0; bi < VH_BOARDS_PER_WAVE; bi++)
#pragma unroll
            for (int q = 0; q < 4; q++) h[bi][q] = b1r[q];
#pragma unroll 4
        for (int p = 0; p < 64; p++) {
            float w[4];
#pragma unroll
            for (int q = 0; q < 4; q++) w[q] = w1s[p * 256 + lane + 64 * q];
#pragma unroll
            for (int bi = 0; bi < VH_BOARDS_PER_WAVE; bi++) {
                const float vp = vw[bi * 64 + p];
#pragma unroll
                for (int q = 0; q < 4; q++) h[bi][q] += vp * w[q];
            }
        }
#pragma unroll
        for (int bi = 0; bi < VH_BOARDS_PER_WAVE; bi++) {
            float part = 0.f;
#pragma unroll
            for (int q = 0; q < 4; q++) part += fmaxf(h[bi][q], 0.f) * w2r[q];
            for (int off = 32; off >= 1; off >>= 1) part += __shfl_xor(part, off);
            if (lane == 0 && board0 + bi < n_boards) value[board0 + bi] = tanhf(part + fc2_b);
        }
        return;
    }
    for (int bi = 0; bi < VH_BOARDS_PER_WAVE; bi++) {
        const int board = (blockIdx.x * 4 + wave) * VH_BOARDS_PER_WAVE + bi;
        if (board >= n_boards) break;
        // coalesced: the board's 32 KB tile is read as 32 contiguous 1 KiB pieces; in piece i lane l holds channels
        // (l&31)*8..+7 of position 2i + (l>>5); the 32 lanes of a position are summed with an xor butterfly
        const uint4* tile = (const uint4*)(x + (size_t)board * 64 * 256);
        float dot = 0.f;                                    // lane p ends up with the conv_v1 output of position p
#pragma unroll 8
        for (int i = 0; i < (v1_in ? 0 : 32); i++) {
            uint4 v = tile[i * 64 + lane];
            float part = bf16_lo(v.x) * wreg[0] + bf16_hi(v.x) * wreg[1] + bf16_lo(v.y) * wreg[2] + bf16_hi(v.y) * wreg[3] +
                         bf16_lo(v.z) * wreg[4] + bf16_hi(v.z) * wreg[5] + bf16_lo(v.w) * wreg[6] + bf16_hi(v.w) * wreg[7];
            for (int off = 16; off >= 1; off >>= 1) part += __shfl_xor(part, off);
            const float lo = __shfl(part, 0), hi = __shfl(part, 32);
            if (lane == 2 * i) dot = lo;
            if (lane == 2 * i + 1) dot = hi;
        }
        v1s[wave * 64 + lane] = v1_in ? v1_in[(size_t)board * 64 + lane] : fmaxf(dot + bv, 0.f);       // relu(bn(conv_v1)) for position `lane`
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_s_waitcnt(0xC07F);                 // lgkmcnt(0): the wave's own LDS writes have landed
        float h[4] = {b1r[0], b1r[1], b1r[2], b1r[3]};
        for (int p = 0; p < 64; p++) {
            const float vp = v1s[wave * 64 + p];
#pragma unroll
            for (int q = 0; q < 4; q++) h[q] += vp * w1s[p * 256 + lane + 64 * q];
        }
        float part = 0.f;
#pragma unroll
        for (int q = 0; q < 4; q++) part += fmaxf(h[q], 0.f) * w2r[q];
        for (int off = 32; off >= 1; off >>= 1) part += __shfl_xor(part, off);
        if (lane == 0) value[board] = tanhf(part + fc2_b);
    }
}

// Both heads from ONE pass over the tower output x (network.py:141-174): the workgroup stages its 2 boards of x in LDS once, then
//   value : conv_v1 (256 -> 1, v_norm folded, f32 weights) + ReLU per position -> v1_out [n_boards][64] (the 64->256->1 MLP is k_value_head)
//   policy: t = relu(bn(conv_p1(x)))  (1x1, the tower's own K loop)  -> bf16 into LDS over x -> conv_p2 (256 -> 73) + bias -> softmax over
//           the board's 4672 logits -> probs f32 in the reference's flatten order [plane*64 + pos].
// Replaces k_conv16<256,1> + k_policy_head + the x-reading part of k_value_head: x is read once (134 MB at B = 4096) instead of
// three passes over 134 MB plus a 134 MB intermediate written and re-read.
template <class E /* operand element of x, t and the packed weights: ElemBF16 or ElemF16 */>
__global__ __launch_bounds__(256, 2) void k_heads16_bf16(const uint16_t* __restrict__ x, const uint4* __restrict__ w_p1, const float* __restrict__ b_p1,
                                                          const uint4* __restrict__ w_p2, const float* __restrict__ b_p2, const float* __restrict__ wv, float bv,
                                                          float* __restrict__ probs, float* __restrict__ v1_out, int n_boards, int do_softmax) {
    constexpr int WGB = 2, PITCH = NN_COUT * 2 + NN_PAD16;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    float* red = (float*)(lds + WGB * 64 * PITCH + NN_ZERO16);    // [4 waves][2]: softmax max / sum exchange between the two waves of a board
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int p16 = lane & 15, kg = lane >> 4;
    const int board0 = blockIdx.x * WGB;
    stage_tile<256, WGB, NN_PAD16>(lds, x, board0, n_boards, false);
    const float4 wv4 = ((const float4*)wv)[lane];          // conv_v1 weights of channels 4*lane .. 4*lane+3
    __syncthreads();
    {   // value conv: wave w owns rows 32w .. 32w+31 (row = board*64 + position).  Every lane forms its 4-channel partial of all 32
        // rows, then a halving butterfly (16+8+4+2+1+1 = 32 shuffles instead of 32 x 6) leaves row l>>1 in lane l.
        float part[32];
#pragma unroll
        for (int r = 0; r < 32; r++) {
            const uint2 v = *(const uint2*)(lds + (wave * 32 + r) * PITCH + lane * 8);
            part[r] = E::lo(v.x) * wv4.x + E::hi(v.x) * wv4.y + E::lo(v.y) * wv4.z + E::hi(v.y) * wv4.w;
        }
#pragma unroll
        for (int m = 32, n = 16; m >= 2; m >>= 1, n >>= 1) {
            const bool up = (lane & m) != 0;
#pragma unroll
            for (int k = 0; k < n; k++) {
                const float mine = up ? part[n + k] : part[k], send = up ? part[k] : part[n + k];
                part[k] = mine + __shfl_xor(send, m);
            }
        }
        const float tot = part[0] + __shfl_xor(part[0], 1);
        const int row = wave * 32 + (lane >> 1), board = board0 + (row >> 6);
        if (!(lane & 1) && board < n_boards) v1_out[(size_t)board * 64 + (row & 63)] = fmaxf(tot + bv, 0.f);
    }
    f32x4 acc[4][4 * WGB];
    conv_kloop16<256, 1, WGB, 2, false, 0, false, E>(lds, w_p1, acc, false, false, nullptr, 0, b_p1);
    __syncthreads();
    acc_to_lds16<WGB, E>(lds, acc, nullptr, true);            // t over x, in the layout the MFMA B operand is read from
    __syncthreads();
    // policy logits: wave -> board wave>>1, position tiles 2*(wave&1) + {0,1}; 5 channel tiles (73 padded to 80), K = 256
    const int pboard = wave >> 1, j0 = (wave & 1) * 2;
    f32x4 pa[5][2];
#pragma unroll
    for (int i = 0; i < 5; i++) { pa[i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; pa[i][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll 2
    for (int kc = 0; kc < 8; kc++) {
        bf16x8 a[5], b[2];
#pragma unroll
        for (int i = 0; i < 5; i++) a[i] = __builtin_bit_cast(bf16x8, w_p2[(size_t)(kc * 5 + i) * 64 + lane]);
#pragma unroll
        for (int j = 0; j < 2; j++) b[j] = *(const bf16x8*)(lds + (pboard * 64 + (j0 + j) * 16 + p16) * PITCH + kc * 64 + kg * 16);
#pragma unroll
        for (int i = 0; i < 5; i++)
#pragma unroll
            for (int j = 0; j < 2; j++) pa[i][j] = E::mfma(a[i], b[j], pa[i][j]);
    }
    // lane holds logits of positions (j0+j)*16 + p16, channels i*16 + 4*kg + r
    float mx = -3.0e38f;
#pragma unroll
    for (int i = 0; i < 5; i++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int co = i * 16 + 4 * kg + r;
            const float bb = co < 73 ? b_p2[co] : 0.f;
#pragma unroll
            for (int j = 0; j < 2; j++) {
                pa[i][j][r] += bb;
                if (co < 73) mx = fmaxf(mx, pa[i][j][r]);
            }
        }
    float inv = 1.0f;
    if (do_softmax) {                                       // uniform across the workgroup
        for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
        if (lane == 0) red[wave * 2] = mx;
        __syncthreads();
        mx = fmaxf(red[wave * 2], red[(wave ^ 1) * 2]);
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 5; i++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int co = i * 16 + 4 * kg + r;
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    const float e = co < 73 ? __expf(pa[i][j][r] - mx) : 0.f;
                    pa[i][j][r] = e;
                    sum += e;
                }
            }
        for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off);
        if (lane == 0) red[wave * 2 + 1] = sum;
        __syncthreads();
        inv = 1.0f / (red[(wave & 2) * 2 + 1] + red[((wave & 2) + 1) * 2 + 1]);      // same operand order in both waves of a board
    }
    if (board0 + pboard < n_boards) {
        float* pb = probs + (size_t)(board0 + pboard) * 4672;
#pragma unroll
        for (int i = 0; i < 5; i++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int co = i * 16 + 4 * kg + r;
                if (co < 73) {
#pragma unroll
                    for (int j = 0; j < 2; j++) pb[co * 64 + (j0 + j) * 16 + p16] = pa[i][j][r] * inv;
                }
            }
    }
}

// x <- relu(acc + bias + x) in place on the LDS image (f32 add, one bf16 rounding): every lane owns its 4 channels x 1 position.
// bias == nullptr: the accumulators already started at the bias.
template <int WGB>
__device__ __forceinline__ void acc_residual_inplace16(unsigned char* xlds, const f32x4 (&acc)[4][4 * WGB], const float* __restrict__ bias) {
    constexpr int OPITCH = NN_COUT * 2 + NN_PAD16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int p16 = lane & 15, kg = lane >> 4;
#pragma unroll
    for (int j = 0; j < 4 * WGB; j++) {
        const int row = tile_row<WGB>(j, p16);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int co = (wave * 4 + i) * 16 + 4 * kg;
            f32x4 v = acc[i][j];
            if (bias) v += *(const f32x4*)(bias + co);
            uint2* px = (uint2*)(xlds + row * OPITCH + co * 2);
            const uint2 r = *px;
            uint2 o;
            o.x = relu_bf16x2(pack_bf16x2(v[0] + bf16_lo(r.x), v[1] + bf16_hi(r.x)));
            o.y = relu_bf16x2(pack_bf16x2(v[2] + bf16_lo(r.y), v[3] + bf16_hi(r.y)));
            *px = o;
        }
    }
}

// =================================================================================================================
// Whole tower in ONE persistent launch (network.py:176-184: stem + 19 BasicBlocks).  A workgroup (4 waves, one per
// SIMD, one workgroup per CU) takes a 2-board tile through all 39 convolutions: the running activation x and the
// block-internal activation t live in two LDS images (2 x 70 KB), so between the NHWC planes read by the stem and the
// tower output nothing but WEIGHTS moves: no per-layer tile loads/stores (5 GB of HBM traffic per forward at B=4096),
// no launch boundaries, the residual is added in f32 from LDS.  With the whole register file per wave the weight ring is
// 4 deep (3 k-steps = 1536 matrix-pipe cycles ahead), tap addresses come from a table in LDS, the first MFMAs of a convolution take
// the bias as their C operand.  In-kernel phase timing (tools/tower_stamps.py): 91.4k cycles per BasicBlock against 73.7k MFMA cycles
// = 80.7 % matrix-pipe busy; forward at B = 4096 7.3-7.6 ms vs 7.6-7.8 ms with per-block launches on the same box, B = 512 0.96 vs
// 1.23 ms (an 8-wave, two-waves-per-SIMD variant lost 7 %).  FastPolicyNet uses it at every batch size.
// =================================================================================================================
// Phase stagger of the four waves of a persistent-tower workgroup.  After a barrier the waves run the K loop in lock-step, and every wave issues its
// LDS fragment reads in the same 4 of 16 MFMA gaps of a half-step: 16 KiB wanted in 64 cycles from a 128 B/clk LDS.  Delaying wave w by w x 64 cycles
// spreads the four read windows over the 256-cycle half-step.  NN_STAGGER=0 builds without it (A/B).
#ifndef NN_STAGGER
#define NN_STAGGER 1
#endif
__device__ __forceinline__ void wave_stagger() {
#if NN_STAGGER
    const int wave = threadIdx.x >> 6;
    if (wave == 1) __builtin_amdgcn_s_sleep(1 * NN_STAGGER);
    else if (wave == 2) __builtin_amdgcn_s_sleep(2 * NN_STAGGER);
    else if (wave == 3) __builtin_amdgcn_s_sleep(3 * NN_STAGGER);
#endif
}

#define NN_MAX_CONVS 40
struct TowerParams {
    const uint4* w[NN_MAX_CONVS];                          // [0] stem (C_in 128), then conv1, conv2 of each block (16x16x32 fragment order)
    const float* b[NN_MAX_CONVS];
    unsigned long long* pace;                              // 8 arrival counters, one per XCD residue class of blockIdx (XCD-paced tile rounds); NULL = off (SZ_NN_PACE=0)
    unsigned long long pace_base;                          // arrivals per counter before this launch
};

// STAMP = diagnostic build (tools/tower_stamps.py): s_memtime stamps around the phases of block 3 of a workgroup's second tile (its only tile at small batches) go to
// a buffer of their own; the shipped instantiation (STAMP = false) executes no stamp.
// tile-level stamps (second tile of a workgroup) go behind the block stamps: slot 8192 + wave*8 + k
#define TILESTAMP(k) do { if (STAMP_ && tile == stamp_tile) { unsigned long long _t = __builtin_amdgcn_s_memtime(); if ((threadIdx.x & 63) == 0) stamps[8192 + (size_t)(blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + (k)] = _t; } } while (0)
#define TSTAMP(k) do { if (STAMP_ && stamp_now) { unsigned long long _t = __builtin_amdgcn_s_memtime(); if ((threadIdx.x & 63) == 0) stamps[(size_t)(blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + (k)] = _t; } } while (0)
// WGB = 1: ONE board per workgroup, for batches of at most #CUs boards (a search of one position, the tail of a self-play run): half the latency per forward
// of a 2-board tile with an empty half.  Same accumulation order per output element, so a board's result does not depend on the form.
template <int MODE /* 0 = shipped; 1 = stamps; 2/3/4 = stamps + K-loop ablation 1/2/3 (results garbage) */, class E = ElemBF16 /* operand element: bf16 or f16 */,
          int WGB = 2 /* boards per workgroup */>
__global__ __launch_bounds__(256, 1) void k_tower16_bf16(const uint16_t* __restrict__ planes, TowerParams prm, uint16_t* __restrict__ out, int n_boards, int n_blocks, int flags,
                                                          unsigned long long* __restrict__ stamps) {
    constexpr bool STAMP_ = MODE != 0;
    constexpr int ABL = MODE == 5 ? 4 : (MODE >= 2 ? MODE - 1 : 0);      // 5: stamps with the border-row skipping switched off (A/B of SKIPROWS)
    constexpr int IMG = WGB * 64 * (NN_COUT * 2 + NN_PAD16) + NN_ZERO16;  // one activation image incl. its zero region
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* bufX = lds;
    unsigned char* bufT = lds + IMG;
    const int n_tiles = (n_boards + WGB - 1) / WGB;
    const int stamp_tile = (int)blockIdx.x + (n_tiles > (int)gridDim.x ? (int)gridDim.x : 0);      // diagnostic builds: a workgroup's second tile, or its only one
    // zero rows of both 256-channel images (row index 128); the stem's own zero row is rewritten by stage_tile
    for (int c = threadIdx.x; c < NN_ZERO16 / 16; c += 256) {
        *(uint4*)(bufX + WGB * 64 * (NN_COUT * 2 + NN_PAD16) + c * 16) = make_uint4(0, 0, 0, 0);
        *(uint4*)(bufT + WGB * 64 * (NN_COUT * 2 + NN_PAD16) + c * 16) = make_uint4(0, 0, 0, 0);
    }
    // tap address table of the 256-channel images (9 taps x 8 position tiles x 64 lanes), built once per workgroup
    int* addr_tab = (int*)(lds + 2 * IMG);
    for (int e = threadIdx.x; e < 9 * 4 * WGB * 64; e += 256)
        addr_tab[e] = conv_tap_addr16<NN_COUT * 2 + NN_PAD16, 9, WGB>(e / (4 * WGB * 64), (e >> 6) % (4 * WGB), e & 15, (e >> 4) & 3);
    f32x4 acc[4][4 * WGB];
    constexpr int TRING = WGB == 1 ? NN_ONE_RING : 4;
    uint4 ring[TRING][4];                                              // next convolution's first weight fragments, fetched under the current epilogue
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int board0 = tile * WGB;
        TILESTAMP(0);
        conv_prefetch16<4>(prm.w[0], (uint4 (&)[4][4])ring);
        if (prm.pace && threadIdx.x == 0) {
            // XCD-paced tile rounds: the 32 workgroups that share an XCD (blockIdx mod 8, round-robin dispatch) start every tile round together, so
            // that a layer's weights are fetched into the XCD's 4 MB L2 once per round and hit by the other 31: FETCH_SIZE 2.3-4.5e6 -> 1.43e6 KB raw
            // per launch = 8 XCDs x 8 rounds x 45 MB, the minimum of this design; launch time unchanged.  Bounded wait (<= 512 x 128 cycles): a
            // workgroup that is not joined in time goes on alone, so a wrong mapping assumption or a shared GPU costs time, never progress.
            const int grp = gridDim.x >> 3, round = (tile - blockIdx.x) / gridDim.x;
            unsigned long long* c = prm.pace + (blockIdx.x & 7);
            const unsigned long long target = prm.pace_base + (unsigned long long)(round + 1) * grp;
            atomicAdd(c, 1ULL);
            for (int spins = 0; spins < 512 && __hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target; spins++) __builtin_amdgcn_s_sleep(2);
        }
        __syncthreads();                                               // previous tile's output image fully read
        TILESTAMP(1);
        if (flags & SZ_NN_IN_BITS) stage_tile_bits<WGB, NN_PAD16, E>(bufT, planes, board0, n_boards);
        else stage_tile<128, WGB, NN_PAD16>(bufT, planes, board0, n_boards, false);
        __syncthreads();
        TILESTAMP(2);
        conv_kloop16<128, 9, WGB, 4, true, 0, false, E>(lds, prm.w[0], acc, false, false, ring, IMG, prm.b[0], addr_tab);  // stem (reads bufT): x = relu(bn(conv1(planes)))
        TILESTAMP(3);
        if (n_blocks > 0) conv_prefetch16<TRING>(prm.w[1], ring);
        acc_to_lds16<WGB, E>(bufX, acc, nullptr, true);
        __syncthreads();
        TILESTAMP(4);
        for (int blk = 0; blk < n_blocks; blk++) {
            const bool stamp_now = STAMP_ && blk == 3 && tile == stamp_tile;
            TSTAMP(0);
            wave_stagger();
            auto epi_t = [&](int i, int j) { acc_tile_to_lds16<WGB, E>(bufT, acc, i, j, true); };          // t = relu(bn1(conv1(x))); bufT is idle
            if constexpr (WGB == 2) conv_kloop16<256, 9, WGB, 4, true, ABL, false, E>(bufX, prm.w[1 + 2 * blk], acc, false, false, ring, 0, prm.b[1 + 2 * blk], addr_tab, EpiTile16<WGB, E>(bufT, acc));
            else conv_kloop16_one<ABL, E>(bufX, prm.w[1 + 2 * blk], acc, ring, 0, prm.b[1 + 2 * blk], addr_tab);
            TSTAMP(1);
            conv_prefetch16<TRING>(prm.w[2 + 2 * blk], ring);
#pragma unroll
            for (int j = (WGB == 2 ? 2 * WGB : 0); j < 4 * WGB; j++)           // second position half; the first went out under the last tap (2-board form)
#pragma unroll
                for (int i = 0; i < 4; i++) epi_t(i, j);
            TSTAMP(2);
            __syncthreads();
            TSTAMP(3);
            wave_stagger();
            auto epi_x = [&](int i, int j) { acc_tile_residual16<WGB, E>(bufX, acc, i, j); };             // x = relu(bn2(conv2(t)) + x): own elements only, nobody reads bufX now
            if constexpr (WGB == 2) conv_kloop16<256, 9, WGB, 4, true, ABL, false, E>(lds, prm.w[2 + 2 * blk], acc, false, false, ring, IMG, prm.b[2 + 2 * blk], addr_tab, EpiResidual16<WGB, E>(bufX, acc));   // reads bufT
            else conv_kloop16_one<ABL, E>(lds, prm.w[2 + 2 * blk], acc, ring, IMG, prm.b[2 + 2 * blk], addr_tab);
            TSTAMP(4);
            if (blk + 1 < n_blocks) conv_prefetch16<TRING>(prm.w[3 + 2 * blk], ring);
#pragma unroll
            for (int j = (WGB == 2 ? 2 * WGB : 0); j < 4 * WGB; j++)
#pragma unroll
                for (int i = 0; i < 4; i++) epi_x(i, j);
            TSTAMP(5);
            __syncthreads();
            TSTAMP(6);
            if (STAMP_ && stamp_now && (threadIdx.x & 63) == 0) stamps[(size_t)(blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + 7] = __builtin_amdgcn_s_memrealtime();
        }
        TILESTAMP(5);
        lds_to_out<WGB, NN_PAD16>(bufX, nullptr, out, board0, n_boards, false);
        TILESTAMP(6);
    }
}


static int default_flags(int flags, int wgb) {
    if (wgb == 2 && !(flags & (32 | 64 | 0xFF00)) && !(flags & 0x10000)) flags |= 32 | (3 << 8);   // stagger the first co-resident pair
    return flags;
}

template <int CIN, int NTAPS, int WGB> static int launch_conv(const void* in, const void* w, const float* bias, const void* res, void* out, int n_boards, int flags, hipStream_t s) {
    constexpr int PITCH = CIN * 2 + 16;
    const size_t lds_in = (size_t)(WGB * 64 + 1) * PITCH, lds_out = (size_t)(WGB * 64) * (NN_COUT * 2 + 16);
    const size_t lds = lds_in > lds_out ? lds_in : lds_out;
    static bool attr_flags[NN_MAX_DEVICES] = {};                       // function attributes are per device
    bool& attr_set = attr_flags[current_device_slot()];
    if (!attr_set) {
        HIPCHK(hipFuncSetAttribute((const void*)k_conv_bf16<CIN, NTAPS, WGB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    const int grid = (n_boards + WGB - 1) / WGB;
    hipLaunchKernelGGL((k_conv_bf16<CIN, NTAPS, WGB>), dim3(grid), dim3(256), lds, s, (const uint16_t*)in, (const uint4*)w, bias, (const uint16_t*)res,
                       (uint16_t*)out, n_boards, default_flags(flags, WGB), device_cus());
    HIPCHK(hipGetLastError());
    return SZ_OK;
}

template <int WGB> static int launch_block(const void* in, const void* w1, const float* b1, const void* w2, const float* b2, void* out, int n_boards, int flags, hipStream_t s) {
    const size_t lds = (size_t)(WGB * 64 + 1) * (256 * 2 + 16);
    static bool attr_flags[NN_MAX_DEVICES] = {};                       // function attributes are per device
    bool& attr_set = attr_flags[current_device_slot()];
    if (!attr_set) {
        HIPCHK(hipFuncSetAttribute((const void*)k_block_bf16<WGB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    const int grid = (n_boards + WGB - 1) / WGB;
    hipLaunchKernelGGL((k_block_bf16<WGB>), dim3(grid), dim3(256), lds, s, (const uint16_t*)in, (const uint4*)w1, b1, (const uint4*)w2, b2, (uint16_t*)out,
                       n_boards, default_flags(flags, WGB), device_cus());
    HIPCHK(hipGetLastError());
    return SZ_OK;
}

template <int CIN, int NTAPS> static int launch_conv16(const void* in, const void* w, const float* bias, const void* res, void* out, int n_boards, int flags, hipStream_t s) {
    constexpr int WGB = 2, PITCH = CIN * 2 + NN_PAD16;
    const size_t lds_in = (size_t)(WGB * 64) * PITCH + NN_ZERO16, lds_out = (size_t)(WGB * 64) * (NN_COUT * 2 + NN_PAD16);
    const size_t lds = lds_in > lds_out ? lds_in : lds_out;
    static bool attr_flags[NN_MAX_DEVICES] = {};                       // function attributes are per device
    bool& attr_set = attr_flags[current_device_slot()];
    if (!attr_set) {
        HIPCHK(hipFuncSetAttribute((const void*)k_conv16_bf16<CIN, NTAPS, WGB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    hipLaunchKernelGGL((k_conv16_bf16<CIN, NTAPS, WGB>), dim3((n_boards + WGB - 1) / WGB), dim3(256), lds, s, (const uint16_t*)in, (const uint4*)w, bias,
                       (const uint16_t*)res, (uint16_t*)out, n_boards, default_flags(flags, WGB), device_cus());
    HIPCHK(hipGetLastError());
    return SZ_OK;
}
template <int WGB> static int launch_block16(const void* in, const void* w1, const float* b1, const void* w2, const float* b2, void* out, int n_boards, int flags, hipStream_t s) {
    const size_t lds = (size_t)(WGB * 64) * (256 * 2 + NN_PAD16) + NN_ZERO16;
    static bool attr_flags[NN_MAX_DEVICES] = {};                       // function attributes are per device
    bool& attr_set = attr_flags[current_device_slot()];
    if (!attr_set) {
        HIPCHK(hipFuncSetAttribute((const void*)k_block16_bf16<WGB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    hipLaunchKernelGGL((k_block16_bf16<WGB>), dim3((n_boards + WGB - 1) / WGB), dim3(256), lds, s, (const uint16_t*)in, (const uint4*)w1, b1, (const uint4*)w2, b2,
                       (uint16_t*)out, n_boards, default_flags(flags, WGB), device_cus());
    HIPCHK(hipGetLastError());
    return SZ_OK;
}

// host-side weight packing into MFMA A-fragment order of the 16x16x32 path, for either operand element
template <class E> static int pack_weights16_impl(const float* w_in, int32_t cin_real, int32_t cin_padded, int32_t ksize, uint16_t* out) {
    if (!w_in || !out || (ksize != 1 && ksize != 3) || cin_padded % 32 || cin_real > cin_padded) return SZ_ERR_INVALID;
    const int taps = ksize * ksize, ksteps = cin_padded / 32;
    for (int t = 0; t < taps; t++)
        for (int ks = 0; ks < ksteps; ks++)
            for (int tile = 0; tile < 16; tile++)
                for (int l = 0; l < 64; l++)
                    for (int j = 0; j < 8; j++) {
                        int co = tile * 16 + (l & 15), ci = ks * 32 + 8 * (l >> 4) + j;
                        float v = (ci < cin_real) ? w_in[((size_t)co * cin_real + ci) * taps + t] : 0.f;
                        out[((((size_t)t * ksteps + ks) * 16 + tile) * 64 + l) * 8 + j] = E::from_float(v);
                    }
    return SZ_OK;
}
template <class E> static int pack_head16_impl(const float* w_in, uint16_t* out) {
    if (!w_in || !out) return SZ_ERR_INVALID;
    for (int ks = 0; ks < 8; ks++)
        for (int tile = 0; tile < 5; tile++)
            for (int l = 0; l < 64; l++)
                for (int j = 0; j < 8; j++) {
                    int co = tile * 16 + (l & 15), ci = ks * 32 + 8 * (l >> 4) + j;
                    float v = co < 73 ? w_in[(size_t)co * 256 + ci] : 0.f;
                    out[(((size_t)ks * 5 + tile) * 64 + l) * 8 + j] = E::from_float(v);
                }
    return SZ_OK;
}

extern "C" {

// Fused conv (+folded BN) + bias (+ residual) (+ ReLU), NHWC bf16, C_out = 256.
//   ksize 3: 3x3 pad 1 (network.py:17-29 conv3x3) ; ksize 1: 1x1 (network.py:32-34 conv1x1)
//   cin: 128 (stem, channels >= 119 are zero; with SZ_NN_W16 | SZ_NN_IN_BITS `in` is the bit-packed plane image) or 256.
int sz_nn_conv_bf16(const void* in, const void* w_packed, const float* bias, const void* residual, void* out,
                    int32_t n_boards, int32_t cin, int32_t ksize, int32_t relu, void* stream) {
    if (!in || !w_packed || !bias || !out || n_boards <= 0) return SZ_ERR_INVALID;
    StreamDeviceGuard _guard(stream);
    hipStream_t s = (hipStream_t)stream;
    if ((relu & SZ_NN_IN_BITS) && !((relu & SZ_NN_W16) && ksize == 3 && cin == 128)) return SZ_ERR_INVALID;
    if (relu & SZ_NN_W16) {                                // weights packed for the 16x16x32 path (sz_nn_pack_weights16)
        if (ksize == 3 && cin == 256) return launch_conv16<256, 9>(in, w_packed, bias, residual, out, n_boards, relu, s);
        if (ksize == 3 && cin == 128) return launch_conv16<128, 9>(in, w_packed, bias, residual, out, n_boards, relu, s);
        if (ksize == 1 && cin == 256) return launch_conv16<256, 1>(in, w_packed, bias, residual, out, n_boards, relu, s);
        return SZ_ERR_INVALID;
    }
    const bool wg4 = (relu & 16) != 0;                   // A/B switch: 4-board workgroups (1 per CU)
    if (ksize == 3 && cin == 256) return wg4 ? launch_conv<256, 9, 4>(in, w_packed, bias, residual, out, n_boards, relu, s)
                                             : launch_conv<256, 9, 2>(in, w_packed, bias, residual, out, n_boards, relu, s);
    if (ksize == 3 && cin == 128) return launch_conv<128, 9, 2>(in, w_packed, bias, residual, out, n_boards, relu, s);
    if (ksize == 1 && cin == 256) return launch_conv<256, 1, 2>(in, w_packed, bias, residual, out, n_boards, relu, s);
    return SZ_ERR_INVALID;
}

// One BasicBlock (network.py:36-83) in one launch: out = relu(conv3x3(relu(conv3x3(in, w1) + b1), w2) + b2 + in).
// in/out [n_boards,64,256] bf16 NHWC (out must not alias in), weights from sz_nn_pack_weights, biases [256] f32.
int sz_nn_block_bf16(const void* in, const void* w1_packed, const float* bias1, const void* w2_packed, const float* bias2, void* out,
                     int32_t n_boards, int32_t flags, void* stream) {
    if (!in || !w1_packed || !bias1 || !w2_packed || !bias2 || !out || in == out || n_boards <= 0) return SZ_ERR_INVALID;
    StreamDeviceGuard _guard(stream);
    if (flags & SZ_NN_W16) return (flags & 0x80000) ? launch_block16<1>(in, w1_packed, bias1, w2_packed, bias2, out, n_boards, flags, (hipStream_t)stream)   // A/B: 1-board workgroups, 3-4 per CU
                                                    : launch_block16<2>(in, w1_packed, bias1, w2_packed, bias2, out, n_boards, flags, (hipStream_t)stream);
    return launch_block<2>(in, w1_packed, bias1, w2_packed, bias2, out, n_boards, flags, (hipStream_t)stream);
}

// Weight packing for the 16x16x32 path: [taps][cin/32 k-steps][16 co tiles][64 lanes][8] bf16;
//   lane l, elem j <- w[co = tile*16 + (l&15)][ci = kstep*32 + 8*(l>>4) + j]
int sz_nn_pack_weights16(const float* w_in, int32_t cin_real, int32_t cin_padded, int32_t ksize, uint16_t* out) {
    return pack_weights16_impl<ElemBF16>(w_in, cin_real, cin_padded, ksize, out);
}
// the same fragment order with f16 elements (SZ_NN_F16 kernels)
int sz_nn_pack_weights16_f16(const float* w_in, int32_t cin_real, int32_t cin_padded, int32_t ksize, uint16_t* out) {
    return pack_weights16_impl<ElemF16>(w_in, cin_real, cin_padded, ksize, out);
}

// diagnostic: device buffer of 8 u64 per wave (256 workgroups x 4 waves) that receives the phase stamps of the STAMP build; NULL = off
static unsigned long long* g_tower_stamps = nullptr;
static int g_tower_mode = 1;
int sz_nn_debug_tower_stamps(void* dev_buffer, int32_t mode) { g_tower_stamps = (unsigned long long*)dev_buffer; g_tower_mode = mode; return SZ_OK; }

// Whole tower (stem + n_blocks BasicBlocks) in one persistent launch.  planes [n_boards,64,128] bf16 (NHWC, 119 real channels),
// out [n_boards,64,256] bf16.  w/b: n_convs = 1 + 2*n_blocks device pointers each (weights from sz_nn_pack_weights16, stem with
// cin_padded 128; biases [256] f32 with BatchNorm folded), given as HOST arrays of device pointers.
static int tower_launch(const void* planes, const void* const* w_packed, const float* const* bias, int32_t n_blocks, void* out, int32_t n_boards, int32_t flags, void* stream);
int sz_nn_tower_bf16(const void* planes, const void* const* w_packed, const float* const* bias, int32_t n_blocks, void* out, int32_t n_boards, int32_t flags, void* stream) {
    if (!planes || !w_packed || !bias || !out || n_boards <= 0 || n_blocks < 0 || 1 + 2 * n_blocks > NN_MAX_CONVS) return SZ_ERR_INVALID;
    StreamDeviceGuard _guard(stream);
    // A last round of at most #CUs boards runs in the one-board form as a launch of its own: 768 boards on 256 CUs = one round of 256 two-board tiles + 256 boards with a
    // CU each (0.93 + 0.56 ms) instead of a second two-board round on half of the CUs (0.93 + 0.72 ms).  Per-board results are identical in both forms.
    const int n_cu = device_cus(), rem = n_boards % (2 * n_cu);
    if (n_boards > 2 * n_cu && rem > 0 && rem <= n_cu && !(flags & (SZ_NN_TOWER_WGB1 | SZ_NN_TOWER_WGB2)) && !g_tower_stamps) {
        const int head = n_boards - rem;
        const size_t plane_bytes = (flags & SZ_NN_IN_BITS) ? 64 * sizeof(uint4) : (size_t)64 * 128 * 2, out_bytes = (size_t)64 * NN_COUT * 2;
        const int rc = tower_launch(planes, w_packed, bias, n_blocks, out, head, flags | SZ_NN_TOWER_WGB2, stream);
        if (rc != SZ_OK) return rc;
        return tower_launch((const unsigned char*)planes + head * plane_bytes, w_packed, bias, n_blocks, (unsigned char*)out + head * out_bytes, rem, flags | SZ_NN_TOWER_WGB1, stream);
    }
    return tower_launch(planes, w_packed, bias, n_blocks, out, n_boards, flags, stream);
}
static int tower_launch(const void* planes, const void* const* w_packed, const float* const* bias, int32_t n_blocks, void* out, int32_t n_boards, int32_t flags, void* stream) {
    TowerParams prm;
    memset(&prm, 0, sizeof prm);
    for (int i = 0; i < 1 + 2 * n_blocks; i++) {
        if (!w_packed[i] || !bias[i]) return SZ_ERR_INVALID;
        prm.w[i] = (const uint4*)w_packed[i]; prm.b[i] = bias[i];
    }
    const size_t lds = 2 * ((size_t)(2 * 64) * (NN_COUT * 2 + NN_PAD16) + NN_ZERO16) + 9 * 8 * 64 * sizeof(int);   // two images + tap address table
    const size_t lds1 = 2 * ((size_t)64 * (NN_COUT * 2 + NN_PAD16) + NN_ZERO16) + 9 * 4 * 64 * sizeof(int);         // the one-board form
    static bool attr_flags[NN_MAX_DEVICES] = {};                       // function attributes are per device
    bool& attr_set = attr_flags[current_device_slot()];
    if (!attr_set) {
        HIPCHK(hipFuncSetAttribute((const void*)k_tower16_bf16<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIPCHK(hipFuncSetAttribute((const void*)k_tower16_bf16<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIPCHK(hipFuncSetAttribute((const void*)k_tower16_bf16<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIPCHK(hipFuncSetAttribute((const void*)k_tower16_bf16<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIPCHK(hipFuncSetAttribute((const void*)k_tower16_bf16<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIPCHK(hipFuncSetAttribute((const void*)k_tower16_bf16<5>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIPCHK(hipFuncSetAttribute((const void*)k_tower16_bf16<0, ElemF16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIPCHK(hipFuncSetAttribute((const void*)k_tower16_bf16<1, ElemF16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIPCHK(hipFuncSetAttribute((const void*)k_tower16_bf16<0, ElemBF16, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
        HIPCHK(hipFuncSetAttribute((const void*)k_tower16_bf16<0, ElemF16, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
        HIPCHK(hipFuncSetAttribute((const void*)k_tower16_bf16<1, ElemBF16, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
        HIPCHK(hipFuncSetAttribute((const void*)k_tower16_bf16<2, ElemBF16, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
        HIPCHK(hipFuncSetAttribute((const void*)k_tower16_bf16<3, ElemBF16, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
        HIPCHK(hipFuncSetAttribute((const void*)k_tower16_bf16<4, ElemBF16, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
        attr_set = true;
    }
    const int n_cu = device_cus();
    // one board per workgroup while every board can have a CU of its own; flags SZ_NN_TOWER_WGB1 / _WGB2 force a form (tests)
    const bool one = (flags & SZ_NN_TOWER_WGB1) || (n_boards <= n_cu && !(flags & SZ_NN_TOWER_WGB2));
    const int n_tiles = one ? n_boards : (n_boards + 1) / 2;
    const dim3 grid(n_tiles < n_cu ? n_tiles : n_cu);
    unsigned long long pace_add = 0, *pace_slot = nullptr;     // the arrival counters advance only when the launch went out (a failed launch leaves them consistent)
    {   // XCD-paced tile rounds (see the kernel); SZ_NN_PACE=0 switches them off.  The counters are per device and not thread-safe: one engine thread per GPU
        // (the C ABI's contract); a wrong base costs a bounded wait, never a wrong result.
        static int pace_on = -1;
        static unsigned long long* pace_buf[NN_MAX_DEVICES] = {};
        static unsigned long long pace_total[NN_MAX_DEVICES] = {};
        if (pace_on < 0) { const char* ev = getenv("SZ_NN_PACE"); pace_on = (ev && ev[0] == '0') ? 0 : 1; }
        if (pace_on && grid.x % 8 == 0 && n_tiles % (int)grid.x == 0) {
            const int slot = current_device_slot();
            if (!pace_buf[slot]) { HIPCHK(hipMalloc(&pace_buf[slot], 8 * sizeof(unsigned long long))); HIPCHK(hipMemset(pace_buf[slot], 0, 8 * sizeof(unsigned long long))); }
            prm.pace = pace_buf[slot]; prm.pace_base = pace_total[slot];
            pace_add = (unsigned long long)(n_tiles / (int)grid.x) * (grid.x / 8); pace_slot = &pace_total[slot];
        }
    }
#define TOWER_LAUNCH(M) hipLaunchKernelGGL(k_tower16_bf16<M>, grid, dim3(256), lds, (hipStream_t)stream, (const uint16_t*)planes, prm, (uint16_t*)out, n_boards, n_blocks, (int)flags, g_tower_stamps)
#define TOWER_LAUNCH1(M) hipLaunchKernelGGL((k_tower16_bf16<M, ElemBF16, 1>), grid, dim3(256), lds1, (hipStream_t)stream, (const uint16_t*)planes, prm, (uint16_t*)out, n_boards, n_blocks, (int)flags, g_tower_stamps)
    if (one && g_tower_stamps && !(flags & SZ_NN_F16)) {               // diagnostic builds of the one-board form (bf16 operands): stamps, K-loop ablations 2-4
        if (g_tower_mode == 2) TOWER_LAUNCH1(2); else if (g_tower_mode == 3) TOWER_LAUNCH1(3); else if (g_tower_mode == 4) TOWER_LAUNCH1(4); else TOWER_LAUNCH1(1);
    }
    else if (one && (flags & SZ_NN_F16))
        hipLaunchKernelGGL((k_tower16_bf16<0, ElemF16, 1>), grid, dim3(256), lds1, (hipStream_t)stream, (const uint16_t*)planes, prm, (uint16_t*)out, n_boards, n_blocks, (int)flags, g_tower_stamps);
    else if (one)
        hipLaunchKernelGGL((k_tower16_bf16<0, ElemBF16, 1>), grid, dim3(256), lds1, (hipStream_t)stream, (const uint16_t*)planes, prm, (uint16_t*)out, n_boards, n_blocks, (int)flags, g_tower_stamps);
    else if ((flags & SZ_NN_F16) && g_tower_stamps)
        hipLaunchKernelGGL((k_tower16_bf16<1, ElemF16>), grid, dim3(256), lds, (hipStream_t)stream, (const uint16_t*)planes, prm, (uint16_t*)out, n_boards, n_blocks, (int)flags, g_tower_stamps);
    else if (flags & SZ_NN_F16)
        hipLaunchKernelGGL((k_tower16_bf16<0, ElemF16>), grid, dim3(256), lds, (hipStream_t)stream, (const uint16_t*)planes, prm, (uint16_t*)out, n_boards, n_blocks, (int)flags, g_tower_stamps);
    else if (!g_tower_stamps) TOWER_LAUNCH(0);
    else if (g_tower_mode == 2) TOWER_LAUNCH(2);
    else if (g_tower_mode == 3) TOWER_LAUNCH(3);
    else if (g_tower_mode == 4) TOWER_LAUNCH(4);
    else if (g_tower_mode == 5) TOWER_LAUNCH(5);
    else TOWER_LAUNCH(1);
#undef TOWER_LAUNCH
#undef TOWER_LAUNCH1
    HIPCHK(hipGetLastError());
    if (pace_slot) *pace_slot += pace_add;
    return SZ_OK;
}

// Counter calibration (tools/fetch_calib.py, MI355X_MICROARCH.md "calibrate on a known byte count in your own access pattern"): read `n16` 16-byte
// elements exactly once, grid-strided, 16 B per lane per instruction, with flat global loads (mode 0) or through a buffer descriptor the way the
// tower streams its weights (mode 1); the xor of everything goes to sink so nothing is optimised away.
__global__ __launch_bounds__(256) void k_stream_read(const uint4* __restrict__ src, size_t n16, int mode, uint4* __restrict__ sink) {
    uint4 acc = make_uint4(0, 0, 0, 0);
    const size_t stride = (size_t)gridDim.x * 256;
    if (mode == 0) {
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) { const uint4 v = src[i]; acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
    } else {
        // 1 GiB windows: a raw buffer's offsets are 32 bit
        for (size_t base = 0; base < n16; base += ((size_t)1 << 26)) {
            const size_t cnt = n16 - base < ((size_t)1 << 26) ? n16 - base : ((size_t)1 << 26);
            const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)(src + base), 0, 0x7FFFFFFF, 0x00020000);
            for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < cnt; i += stride) {
                const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)(threadIdx.x * 16), (int)((i - threadIdx.x) * 16), 0);
                acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
            }
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9E3779B9u) sink[0] = acc;
}
int sz_debug_stream_read(const void* src, uint64_t bytes, int32_t mode, void* sink16, void* stream) {
    if (!src || !sink16 || bytes < 16) return SZ_ERR_INVALID;
    StreamDeviceGuard _guard(stream);
    hipLaunchKernelGGL(k_stream_read, dim3(device_cus() * 8), dim3(256), 0, (hipStream_t)stream, (const uint4*)src, (size_t)(bytes / 16), (int)mode, (uint4*)sink16);
    HIPCHK(hipGetLastError());
    return SZ_OK;
}

// Policy head (network.py:141-154 after conv_p1): t [n_boards,64,256] bf16 -> probs [n_boards,4672] f32 (softmax iff do_softmax).
// w_packed from sz_nn_pack_head16 (73 output channels padded to 80), bias [73] f32.
int sz_nn_policy_head_bf16(const void* t, const void* w_packed, const float* bias, float* probs, int32_t n_boards, int32_t do_softmax, void* stream) {
    if (!t || !w_packed || !bias || !probs || n_boards <= 0) return SZ_ERR_INVALID;
    StreamDeviceGuard _guard(stream);
    hipLaunchKernelGGL(k_policy_head, dim3((n_boards + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)t, (const uint4*)w_packed, bias, probs, n_boards, do_softmax);
    HIPCHK(hipGetLastError());
    return SZ_OK;
}
// Value head (network.py:156-174): x [n_boards,64,256] bf16 (tower output) -> value [n_boards] f32.
// wv [256] f32 and bv: conv_v1 with v_norm folded; fc1_w_t [64][256] f32 (fc_v1.weight transposed), fc1_b [256], fc2_w [256], fc2_b.
int sz_nn_value_head_bf16(const void* x, const float* wv, float bv, const float* fc1_w_t, const float* fc1_b, const float* fc2_w, float fc2_b,
                          float* value, int32_t n_boards, void* stream) {
    if (!x || !wv || !fc1_w_t || !fc1_b || !fc2_w || !value || n_boards <= 0) return SZ_ERR_INVALID;
    StreamDeviceGuard _guard(stream);
    const size_t lds = (64 * 256 + 16 * 64) * sizeof(float);
    static bool attr_flags[NN_MAX_DEVICES] = {};                       // function attributes are per device
    bool& attr_set = attr_flags[current_device_slot()];
    if (!attr_set) {
        HIPCHK(hipFuncSetAttribute((const void*)k_value_head<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    const int per_wg = 4 * 4;
    hipLaunchKernelGGL(k_value_head<4>, dim3((n_boards + per_wg - 1) / per_wg), dim3(256), lds, (hipStream_t)stream, (const uint16_t*)x, wv, bv, fc1_w_t, fc1_b, fc2_w, fc2_b, value, n_boards, (const float*)nullptr);
    HIPCHK(hipGetLastError());
    return SZ_OK;
}
// Both heads from ONE read of the tower output (see k_heads16_bf16).  w_p1_packed: conv_p1 + p_norm1 folded, sz_nn_pack_weights16
// (cin_padded 256, ksize 1); w_p2_packed: sz_nn_pack_head16; v1_scratch: [n_boards*64] f32 device scratch.
int sz_nn_heads_bf16(const void* x, const void* w_p1_packed, const float* b_p1, const void* w_p2_packed, const float* b_p2, const float* wv, float bv,
                     const float* fc1_w_t, const float* fc1_b, const float* fc2_w, float fc2_b, float* probs, float* value, float* v1_scratch,
                     int32_t n_boards, int32_t do_softmax, void* stream) {
    if (!x || !w_p1_packed || !b_p1 || !w_p2_packed || !b_p2 || !wv || !fc1_w_t || !fc1_b || !fc2_w || !probs || !value || !v1_scratch || n_boards <= 0) return SZ_ERR_INVALID;
    StreamDeviceGuard _guard(stream);
    const size_t lds_h = (size_t)(2 * 64) * (256 * 2 + NN_PAD16) + NN_ZERO16 + 64, lds_v = (64 * 256 + 16 * 64) * sizeof(float);
    static bool attr_flags[NN_MAX_DEVICES] = {};                       // function attributes are per device
    bool& attr_set = attr_flags[current_device_slot()];
    if (!attr_set) {
        HIPCHK(hipFuncSetAttribute((const void*)k_heads16_bf16<ElemBF16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_h));
        HIPCHK(hipFuncSetAttribute((const void*)k_heads16_bf16<ElemF16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_h));
        HIPCHK(hipFuncSetAttribute((const void*)k_value_head<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_v));
        HIPCHK(hipFuncSetAttribute((const void*)k_value_head<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_v));
        attr_set = true;
    }
    if (do_softmax & SZ_NN_F16)
        hipLaunchKernelGGL(k_heads16_bf16<ElemF16>, dim3((n_boards + 1) / 2), dim3(256), lds_h, (hipStream_t)stream, (const uint16_t*)x, (const uint4*)w_p1_packed, b_p1,
                           (const uint4*)w_p2_packed, b_p2, wv, bv, probs, v1_scratch, n_boards, do_softmax & 1);
    else
        hipLaunchKernelGGL(k_heads16_bf16<ElemBF16>, dim3((n_boards + 1) / 2), dim3(256), lds_h, (hipStream_t)stream, (const uint16_t*)x, (const uint4*)w_p1_packed, b_p1,
                           (const uint4*)w_p2_packed, b_p2, wv, bv, probs, v1_scratch, n_boards, do_softmax & 1);
    HIPCHK(hipGetLastError());
    if (n_boards > 2048)
        hipLaunchKernelGGL(k_value_head<4>, dim3((n_boards + 15) / 16), dim3(256), lds_v, (hipStream_t)stream, (const uint16_t*)x, wv, bv, fc1_w_t, fc1_b, fc2_w, fc2_b,
                           value, n_boards, (const float*)v1_scratch);
    else
        hipLaunchKernelGGL(k_value_head<1>, dim3((n_boards + 3) / 4), dim3(256), lds_v, (hipStream_t)stream, (const uint16_t*)x, wv, bv, fc1_w_t, fc1_b, fc2_w, fc2_b,
                           value, n_boards, (const float*)v1_scratch);
    HIPCHK(hipGetLastError());
    return SZ_OK;
}
// Value MLP alone (network.py:162-172): v1 [n_boards][64] f32 = relu(bn(conv_v1(x))) -> fc_v1 -> ReLU -> fc_v2 -> tanh -> value [n_boards].  f32 throughout;
// the second launch of sz_nn_heads_bf16 and of sz_nn_forward_split.
int sz_nn_value_mlp(const float* v1, const float* fc1_w_t, const float* fc1_b, const float* fc2_w, float fc2_b, float* value, int32_t n_boards, void* stream) {
    if (!v1 || !fc1_w_t || !fc1_b || !fc2_w || !value || n_boards <= 0) return SZ_ERR_INVALID;
    StreamDeviceGuard _guard(stream);
    const size_t lds_v = (64 * 256 + 16 * 64) * sizeof(float);
    static bool attr_flags[NN_MAX_DEVICES] = {};
    bool& attr_set = attr_flags[current_device_slot()];
    if (!attr_set) {
        HIPCHK(hipFuncSetAttribute((const void*)k_value_head<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_v));
        HIPCHK(hipFuncSetAttribute((const void*)k_value_head<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_v));
        attr_set = true;
    }
    if (n_boards > 2048)
        hipLaunchKernelGGL(k_value_head<4>, dim3((n_boards + 15) / 16), dim3(256), lds_v, (hipStream_t)stream, (const uint16_t*)nullptr, (const float*)nullptr, 0.f, fc1_w_t, fc1_b, fc2_w, fc2_b,
                           value, n_boards, v1);
    else
        hipLaunchKernelGGL(k_value_head<1>, dim3((n_boards + 3) / 4), dim3(256), lds_v, (hipStream_t)stream, (const uint16_t*)nullptr, (const float*)nullptr, 0.f, fc1_w_t, fc1_b, fc2_w, fc2_b,
                           value, n_boards, v1);
    HIPCHK(hipGetLastError());
    return SZ_OK;
}
// host: conv_p2.weight [73][256] f32 -> [8 k32-steps][5 co tiles][64 lanes][8] bf16 (channels 73..79 zero)
int sz_nn_pack_head16(const float* w_in, uint16_t* out) { return pack_head16_impl<ElemBF16>(w_in, out); }
int sz_nn_pack_head16_f16(const float* w_in, uint16_t* out) { return pack_head16_impl<ElemF16>(w_in, out); }

// Host-side weight packing into MFMA A-fragment order.
//   w_in : [256 co][cin_real][k][k] f32 (torch conv weight, BN already folded by the caller)
//   out  : [taps][cin/16 ksteps][8 co tiles][64 lanes][8] bf16 ;  lane l, elem j <- w[co = tile*32 + (l&31)][ci = kstep*16 + 8*(l>>5) + j]
int sz_nn_pack_weights(const float* w_in, int32_t cin_real, int32_t cin_padded, int32_t ksize, uint16_t* out) {
    if (!w_in || !out || (ksize != 1 && ksize != 3) || cin_padded % 16 || cin_real > cin_padded) return SZ_ERR_INVALID;
    const int taps = ksize * ksize, ksteps = cin_padded / 16;
    for (int t = 0; t < taps; t++)
        for (int ks = 0; ks < ksteps; ks++)
            for (int tile = 0; tile < 8; tile++)
                for (int l = 0; l < 64; l++)
                    for (int j = 0; j < 8; j++) {
                        int co = tile * 32 + (l & 31), ci = ks * 16 + 8 * (l >> 5) + j;
                        float v = (ci < cin_real) ? w_in[((size_t)co * cin_real + ci) * taps + t] : 0.f;
                        uint32_t u; memcpy(&u, &v, 4);
                        uint32_t r = u + 0x7FFFu + ((u >> 16) & 1u);            // RNE (weights are finite)
                        out[((((size_t)t * ksteps + ks) * 8 + tile) * 64 + l) * 8 + j] = (uint16_t)(r >> 16);
                    }
    return SZ_OK;
}

}  // extern "C"
