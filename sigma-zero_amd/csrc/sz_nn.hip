// sz_nn.hip — hand-written CDNA4 MFMA convolution for the policy/value tower of the reference
// (/root/reference/network.py:36-83 BasicBlock, :105-137 policyNN stem/tower).
//
// conv3x3(pad 1) / conv1x1 over 8x8 boards, NHWC bf16, C_out = 256, C_in in {128 (zero-padded stem), 256},
// with BatchNorm folded into weights/bias (eval mode) and bias + ReLU (+ residual add) fused in the
// epilogue — one launch replaces MIOpen's igemm + batch_norm + clamp + add kernels of a BasicBlock half.
//
// MI355X mapping (not a warp-tiling port):
//   * one workgroup = 4 waves = WGB boards (2 by default -> 68 KB of LDS -> TWO workgroups per CU, so one
//     workgroup's HBM phases (tile load, output store) hide under the other's MFMA phase);
//   * the boards' activations (WGB*64 positions x C_in) are loaded ONCE from HBM into LDS (row pitch
//     C_in*2+16 B: conflict-free ds_read_b128) and stay resident for all 9 taps — the im2col matrix is
//     never formed; a tap is just a per-lane row address (off-board taps read a zero row);
//   * D = W x Act^T on v_mfma_f32_32x32x16_bf16 with the WEIGHTS as the A operand (rows = out channels) and
//     activations as the B operand (cols = positions); every wave owns a distinct quarter of the output
//     channels (2 channel tiles x 2*WGB position tiles), so each weight fragment is fetched by one wave only;
//   * weights are pre-packed on the host in exact fragment order [tap][kstep][co_tile][lane][8]: a weight
//     fragment is one fully coalesced 1 KiB global_load_dwordx4 from L2 (1.18 MB/layer stays L2-resident),
//     prefetched 3 k-steps ahead through a 4-deep register ring (8-deep measured no faster); activations are double-buffered one k-step
//     ahead; the order is pinned with sched_barrier so the compiler's waits become counted vmcnt/lgkmcnt:
//     the K loop has NO workgroup barrier and no exposed memory latency;
//   * epilogue through LDS: (acc + bias) -> bf16 -> [pos][co] image, then whole 16-byte chunks are moved with
//     coalesced residual reads and stores (scattered 8-byte stores from the accumulator layout cost 20 %).
// Measured on MI355X, B = 4096 boards: 0.247 ms per 3x3 conv (1.25 PFLOP/s), 0.280 ms with residual.
// `relu` bit 0 = ReLU; higher bits are timing-ablation / A-B switches used by tools/conv_bench.py only
// (2/4/8 skip load/store/K loop, 16 = 4-board workgroups, 32/64 + bits 8..15 = phase stagger, 0x10000 = no stagger).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/sigmazero.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define NN_WG_BOARDS 4
#define NN_COUT 256

__device__ __forceinline__ uint32_t pack_bf16x2(float a, float b) {
    // round-to-nearest-even via the hardware convert (v_cvt_pk_bf16_f32)
    __bf16 x = (__bf16)a, y = (__bf16)b;
    uint16_t xb = __builtin_bit_cast(uint16_t, x), yb = __builtin_bit_cast(uint16_t, y);
    return (uint32_t)xb | ((uint32_t)yb << 16);
}
__device__ __forceinline__ float bf16_lo(uint32_t v) { return __builtin_bit_cast(float, v << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t v) { return __builtin_bit_cast(float, v & 0xFFFF0000u); }

// in  : [n_boards][64][CIN]  bf16 (NHWC)            w : packed fragments (see sz_nn_pack_weights)
// out : [n_boards][64][256]  bf16 (NHWC)            bias : [256] f32 (BN folded)
// res : optional residual, same layout as out; relu: apply max(0, .) last
template <int CIN, int NTAPS, int WGB /* boards per workgroup: 4 -> 1 workgroup/CU, 2 -> 2 workgroups/CU */>
__global__ __launch_bounds__(256, (WGB == 2 ? 2 : 1)) void k_conv_bf16(const uint16_t* __restrict__ in, const uint4* __restrict__ w, const float* __restrict__ bias,
                                                      const uint16_t* __restrict__ res, uint16_t* __restrict__ out, int n_boards, int relu, int n_cu) {
    constexpr int PITCH = CIN * 2 + 16;                  // bytes per position row in LDS
    constexpr int KSTEPS = CIN / 16;                     // k-steps (16 channels) per tap
    constexpr int ZERO_ROW = WGB * 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int board0 = blockIdx.x * WGB;

    // ---- stage the 4 boards' activations into LDS (coalesced 16-B loads), plus one zero row ----
    {
        constexpr int CHUNKS_PER_POS = CIN / 8;           // 16-B chunks per position
        constexpr int TOTAL = WGB * 64 * CHUNKS_PER_POS;
        const uint4* src = (const uint4*)(in + (size_t)board0 * 64 * CIN);
        const int valid_boards = min(WGB, n_boards - board0);
        const int valid_chunks = valid_boards * 64 * CHUNKS_PER_POS;
        // all loads of a thread are issued before the first LDS write (32 x 16 B in flight per lane: the
        // accumulators are not live yet, so the registers are free) -> one HBM latency per workgroup, not eight
        constexpr int PER_THREAD = TOTAL / 256;
        uint4 stage[PER_THREAD];
#pragma unroll
        for (int i = 0; i < PER_THREAD; i++) {
            const int c = tid + i * 256;
            stage[i] = (c < valid_chunks && !(relu & 2)) ? src[c] : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < PER_THREAD; i++) {
            const int c = tid + i * 256;
            const int pos = c / CHUNKS_PER_POS, ch = c % CHUNKS_PER_POS;
            *(uint4*)(lds + pos * PITCH + ch * 16) = stage[i];
        }
        for (int c = tid; c < PITCH / 16; c += 256) *(uint4*)(lds + ZERO_ROW * PITCH + c * 16) = make_uint4(0, 0, 0, 0);
    }
    __syncthreads();

    // Phase stagger (speed only, never correctness): the two workgroups that share a CU are dispatched together
    // and would run load / MFMA / store phases in lock-step, leaving the matrix pipe idle during both HBM phases.
    // Delaying ONE of the first pair by about half a K loop keeps them out of phase for every later round.
    if (WGB == 2) {
        const int stagger = (relu >> 8) & 0xFF;            // sleep units (x ~8k cycles); 0 = off
        if (stagger) {
            bool second = false;
            // measured: workgroups b and b + #CUs share a CU (round-robin dispatch); HW_ID.WAVE_ID bit 0 selects the
            // same set.  A wrong guess only costs the sleep, never correctness.
            if (relu & 32) second = ((int)blockIdx.x >= n_cu && (int)blockIdx.x < 2 * n_cu);
            if (relu & 64) second = ((int)blockIdx.x < 2 * n_cu) && ((__builtin_amdgcn_s_getreg(0x1804) & 1) != 0);
            second = __builtin_amdgcn_readfirstlane((int)second) != 0;
            if (second)
                for (int i = 0; i < stagger; i++) __builtin_amdgcn_s_sleep(127);
        }
    }

    // wave tiling: NI channel tiles x NJ position tiles of 32x32.  NI=2, NJ=8: every wave owns a distinct quarter of
    // the output channels for all 256 positions, so each weight fragment is fetched from L2 by exactly ONE wave of
    // the workgroup (half the L2 traffic of a 2x2 wave grid); activations are re-read from LDS, which has headroom.
    constexpr int NI = 2, NJ = 2 * WGB;
    f32x16 acc[NI][NJ];
#pragma unroll
    for (int i = 0; i < NI; i++)
#pragma unroll
        for (int j = 0; j < NJ; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    // per-lane geometry: lane owns position p32 of each 32-position tile; half h selects k 8..15
    const int p32 = lane & 31, h = lane >> 5;
    // weight fragment stream of this wave: [tap][kstep][co_tile 0..7][lane] (uint4 = 8 bf16)
    const uint4* wbase = w + (size_t)(wave * NI) * 64 + lane;
    constexpr int W_KSTEP_STRIDE = 8 * 64;                // uint4 per (tap,kstep)

    constexpr int RING = 4;                                // weight ring depth (k-steps); 8 measured no faster
    uint4 aring[RING][NI];
    constexpr int TOTAL_KS = NTAPS * KSTEPS;
    constexpr int PF = RING - 1;                           // weight prefetch distance in k-steps
#pragma unroll
    for (int s = 0; s < PF; s++)
#pragma unroll
        for (int i = 0; i < NI; i++) aring[s][i] = wbase[(size_t)s * W_KSTEP_STRIDE + i * 64];

    // LDS byte address of this lane's activation row for (tap, position tile j); off-board taps -> zero row
    auto tap_addr = [&](int tap, int j) -> int {
        const int dy = (NTAPS == 9) ? tap / 3 - 1 : 0, dx = (NTAPS == 9) ? tap % 3 - 1 : 0;
        int pos = (j & 1) * 32 + p32;                     // position inside its board (board = j >> 1)
        int y = (pos >> 3) + dy, x = (pos & 7) + dx;
        bool ok = (unsigned)y < 8u && (unsigned)x < 8u;
        int row = ok ? ((j >> 1) * 64 + y * 8 + x) : ZERO_ROW;
        return row * PITCH + h * 16;
    };
    int bcur[NJ], bnxt[NJ];
    bf16x8 bfrag[2][NJ];                                   // activations double-buffered one k-step ahead
#pragma unroll
    for (int j = 0; j < NJ; j++) { bcur[j] = tap_addr(0, j); bnxt[j] = bcur[j]; bfrag[0][j] = *(const bf16x8*)(lds + bcur[j]); }

    // Software pipeline, pinned with sched_barrier so that hipcc cannot sink the prefetches to their uses:
    //   issue { weights of k-step ks+PF (L2 -> ring), activations of ks+1 (LDS -> bfrag) } ; 16 MFMAs of ks.
    // The compiler's own s_waitcnt then becomes counted: loads stay in flight under the MFMAs.
    for (int tap = 0; tap < ((relu & 8) ? 0 : NTAPS); tap++) {
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
                for (int j = 0; j < NJ; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bfrag[kc & 1][j], acc[i][j], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int j = 0; j < NJ; j++) bcur[j] = bnxt[j];
    }

    // ---- epilogue ----------------------------------------------------------------------------------
    // The accumulator layout (lane = position, 4 channels per register quad) would give 8-byte stores scattered
    // over 32 rows per instruction — measured at ~115 cycles per wave-instruction in the texture-address path,
    // 20 % of the kernel.  So: (acc + bias) -> bf16 -> LDS image [pos][co] (the activation tile is dead now), then
    // every thread moves whole 16-byte chunks: LDS read, coalesced residual read, add, ReLU, coalesced store
    // (one wave-instruction = 2 complete 512-byte rows).  The residual is added to the bf16-rounded conv+bias
    // value (torch's own bf16 graph rounds there too).
    constexpr int OPITCH = NN_COUT * 2 + 16;               // output image pitch (independent of C_in)
    __syncthreads();                                       // all waves are done reading the activation tile
    if (!((relu & 4) && acc[0][0][0] != 12345.f)) {
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const int row = (j >> 1) * 64 + (j & 1) * 32 + p32;
#pragma unroll
            for (int i = 0; i < NI; i++) {
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const int co = (wave * NI + i) * 32 + 8 * g + 4 * h;
                    f32x4 b4 = *(const f32x4*)(bias + co);
                    uint2 o;
                    o.x = pack_bf16x2(acc[i][j][4 * g + 0] + b4[0], acc[i][j][4 * g + 1] + b4[1]);
                    o.y = pack_bf16x2(acc[i][j][4 * g + 2] + b4[2], acc[i][j][4 * g + 3] + b4[3]);
                    *(uint2*)(lds + row * OPITCH + co * 2) = o;
                }
            }
        }
    }
    __syncthreads();
    if (!(relu & 4)) {
        constexpr int OUT_CHUNKS = WGB * 64 * 32;          // 16-byte chunks of the output tile
        const int valid = min(WGB, n_boards - board0) * 64 * 32;
        const uint4* res4 = res ? (const uint4*)(res + (size_t)board0 * 64 * NN_COUT) : nullptr;
        uint4* out4 = (uint4*)(out + (size_t)board0 * 64 * NN_COUT);
#pragma unroll 4
        for (int c = tid; c < OUT_CHUNKS; c += 256) {
            if (c >= valid) break;
            uint4 v = *(const uint4*)(lds + (c >> 5) * OPITCH + (c & 31) * 16);
            float f[8] = {bf16_lo(v.x), bf16_hi(v.x), bf16_lo(v.y), bf16_hi(v.y), bf16_lo(v.z), bf16_hi(v.z), bf16_lo(v.w), bf16_hi(v.w)};
            if (res4) {
                uint4 r = res4[c];
                f[0] += bf16_lo(r.x); f[1] += bf16_hi(r.x); f[2] += bf16_lo(r.y); f[3] += bf16_hi(r.y);
                f[4] += bf16_lo(r.z); f[5] += bf16_hi(r.z); f[6] += bf16_lo(r.w); f[7] += bf16_hi(r.w);
            }
            if (relu & 1) {
#pragma unroll
                for (int k = 0; k < 8; k++) f[k] = fmaxf(f[k], 0.f);
            }
            uint4 o;
            o.x = pack_bf16x2(f[0], f[1]); o.y = pack_bf16x2(f[2], f[3]); o.z = pack_bf16x2(f[4], f[5]); o.w = pack_bf16x2(f[6], f[7]);
            out4[c] = o;
        }
    }
}

#define HIPCHK(x) do { hipError_t _e = (x); if (_e != hipSuccess) { fprintf(stderr, "[sigmazero] HIP error %s at %s:%d\n", hipGetErrorString(_e), __FILE__, __LINE__); return SZ_ERR_HIP; } } while (0)

template <int CIN, int NTAPS, int WGB> static int launch_conv(const void* in, const void* w, const float* bias, const void* res, void* out, int n_boards, int relu, hipStream_t s) {
    constexpr int PITCH = CIN * 2 + 16;
    const size_t lds_in = (size_t)(WGB * 64 + 1) * PITCH, lds_out = (size_t)(WGB * 64) * (NN_COUT * 2 + 16);
    const size_t lds = lds_in > lds_out ? lds_in : lds_out;
    static bool attr_set = false;
    static int n_cu = 256;
    if (!attr_set) {
        HIPCHK(hipFuncSetAttribute((const void*)k_conv_bf16<CIN, NTAPS, WGB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        int dev = 0; hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) n_cu = prop.multiProcessorCount;
        attr_set = true;
    }
    if (WGB == 2 && !(relu & (32 | 64 | 0xFF00)) && !(relu & 0x10000)) relu |= 32 | (3 << 8);   // default: stagger the first co-resident pair
    const int grid = (n_boards + WGB - 1) / WGB;
    hipLaunchKernelGGL((k_conv_bf16<CIN, NTAPS, WGB>), dim3(grid), dim3(256), lds, s, (const uint16_t*)in, (const uint4*)w, bias, (const uint16_t*)res, (uint16_t*)out, n_boards, relu, n_cu);
    HIPCHK(hipGetLastError());
    return SZ_OK;
}

extern "C" {

// Fused conv (+folded BN) + bias (+ residual) (+ ReLU), NHWC bf16, C_out = 256.
//   ksize 3: 3x3 pad 1 (network.py:17-29 conv3x3) ; ksize 1: 1x1 (network.py:32-34 conv1x1)
//   cin: 128 (stem, channels >= 119 are zero) or 256.
int sz_nn_conv_bf16(const void* in, const void* w_packed, const float* bias, const void* residual, void* out,
                    int32_t n_boards, int32_t cin, int32_t ksize, int32_t relu, void* stream) {
    if (!in || !w_packed || !bias || !out || n_boards <= 0) return SZ_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    const bool wg4 = (relu & 16) != 0;                   // debug/A-B switch: 4-board workgroups (1 per CU)
    if (ksize == 3 && cin == 256) return wg4 ? launch_conv<256, 9, 4>(in, w_packed, bias, residual, out, n_boards, relu, s)
                                             : launch_conv<256, 9, 2>(in, w_packed, bias, residual, out, n_boards, relu, s);
    if (ksize == 3 && cin == 128) return launch_conv<128, 9, 2>(in, w_packed, bias, residual, out, n_boards, relu, s);
    if (ksize == 1 && cin == 256) return launch_conv<256, 1, 2>(in, w_packed, bias, residual, out, n_boards, relu, s);
    return SZ_ERR_INVALID;
}

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
