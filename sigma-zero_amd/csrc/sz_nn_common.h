// sz_nn_common.h — device helpers shared by the network kernels (sz_nn.hip: bf16 tower, heads, per-layer kernels; sz_nn_split.hip: the
// split-precision tower): vector types, bf16 packing, LDS staging of the input planes, the weight-fragment stream through a buffer
// descriptor, the position-tile / tap-address mapping of the 16x16x32 MFMA path, and the host-side launch plumbing.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <type_traits>
#include "../../include/sigmazero.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

// streaming (non-temporal) 16-byte accesses for activation tiles: they are touched once per launch and should not evict
// the layer's weights (2.4 MB per block, re-read by every workgroup) from the XCD's L2
__device__ __forceinline__ uint4 ld_stream(const uint4* p) {
    u32x4 v = __builtin_nontemporal_load((const u32x4*)p);
    return make_uint4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void st_stream(uint4* p, uint4 v) {
    u32x4 t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, (u32x4*)p);
}

#define NN_COUT 256
#define NN_NI 2                                            // channel tiles (32) per wave
#define NN_PAD16 32                                        // LDS row padding of the 16x16x32 path (bytes)
#define NN_ZERO16 768                                      // zero region behind the rows of a 16x16x32-path image (see conv_kloop16 tap_addr)
#ifndef NN_ILV
#define NN_ILV 1                                           // K loop of the 16x16x32 path: memory instructions interleaved into the MFMA gaps (0 = issued in front of each group)
#endif

typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) short i16x2;
__device__ __forceinline__ uint32_t pack_bf16x2(float a, float b) {
    f32x2 v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));       // ONE v_cvt_pk_bf16_f32 (round-to-nearest-even)
}
// ReLU on a packed bf16 pair: a negative bf16 is a negative int16, so max(x, 0) per 16-bit half is the ReLU (v_pk_max_i16; -0 -> +0)
__device__ __forceinline__ uint32_t relu_bf16x2(uint32_t x) {
    i16x2 v = __builtin_bit_cast(i16x2, x), z = {0, 0};
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(v, z));
}
__device__ __forceinline__ float bf16_lo(uint32_t v) { return __builtin_bit_cast(float, v << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t v) { return __builtin_bit_cast(float, v & 0xFFFF0000u); }

// ---- operand element of the 16x16x32 MFMA path: bf16 (8 bits of mantissa, f32's range) or f16 (11 bits, range 6e-5 .. 65504: post-BatchNorm
// activations and BatchNorm-folded weights are O(1), comfortably inside).  Same MFMA cycles, same LDS / weight bytes, one pack instruction per pair either
// way (v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32, both round to nearest even), two to expand a pair; ReLU on the packed pair is v_pk_max_i16 for both (sign bit).
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
struct ElemBF16 {
    static constexpr uint32_t ONE = 0x3F80u;
    static __device__ __forceinline__ uint32_t pack2(float a, float b) { return pack_bf16x2(a, b); }
    static __device__ __forceinline__ float lo(uint32_t v) { return bf16_lo(v); }
    static __device__ __forceinline__ float hi(uint32_t v) { return bf16_hi(v); }
    static __device__ __forceinline__ f32x4 mfma(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
    static uint16_t from_float(float v) { uint32_t u; memcpy(&u, &v, 4); return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16); }      // host, RNE (finite input)
};
struct ElemF16 {
    static constexpr uint32_t ONE = 0x3C00u;
    static __device__ __forceinline__ uint32_t pack2(float a, float b) { f32x2 v = {a, b}; return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2)); }
    static __device__ __forceinline__ float lo(uint32_t v) { return (float)__builtin_bit_cast(f16x2, v)[0]; }
    static __device__ __forceinline__ float hi(uint32_t v) { return (float)__builtin_bit_cast(f16x2, v)[1]; }
    static __device__ __forceinline__ f32x4 mfma(bf16x8 a, bf16x8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
    static uint16_t from_float(float v) { const _Float16 h = (_Float16)v; uint16_t u; memcpy(&u, &h, 2); return u; }                          // host, RNE
};

// ---- stage WGB boards' activations (NHWC rows of C_in bf16) into LDS, plus one zero row ---------------------
template <int CIN, int WGB, int PAD = 16, bool NT = false, bool SWZ = false /* chunk swizzle of the split-precision images: chunk ^ ((position >> 2) & 1) */>
__device__ __forceinline__ void stage_tile(unsigned char* lds, const uint16_t* __restrict__ in, int board0, int n_boards, bool skip) {
    constexpr int PITCH = CIN * 2 + PAD;
    constexpr int CHUNKS_PER_POS = CIN / 8;                // 16-B chunks per position
    constexpr int TOTAL = WGB * 64 * CHUNKS_PER_POS;
    constexpr int PER_THREAD = TOTAL / 256;
    const int tid = threadIdx.x;
    const uint4* src = (const uint4*)(in + (size_t)board0 * 64 * CIN);
    const int valid_chunks = min(WGB, n_boards - board0) * 64 * CHUNKS_PER_POS;
    // all loads of a thread are issued before the first LDS write (the accumulators are not live yet, so the
    // registers are free) -> one HBM latency per workgroup
    uint4 stage[PER_THREAD];
#pragma unroll
    for (int i = 0; i < PER_THREAD; i++) {
        const int c = tid + i * 256;
        stage[i] = (c < valid_chunks && !skip) ? (NT ? ld_stream(src + c) : src[c]) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < PER_THREAD; i++) {
        const int c = tid + i * 256;
        *(uint4*)(lds + (c / CHUNKS_PER_POS) * PITCH + ((c % CHUNKS_PER_POS) ^ (SWZ ? ((c / CHUNKS_PER_POS) >> 2) & 1 : 0)) * 16) = stage[i];
    }
    for (int c = tid; c < (PAD == NN_PAD16 ? NN_ZERO16 : PITCH) / 16; c += 256) *(uint4*)(lds + WGB * 64 * PITCH + c * 16) = make_uint4(0, 0, 0, 0);
}

// ---- stage WGB boards from the engine's bit-packed planes (SZ_PLANES_NHWC128_BITS: 1 KiB per board) ----------
// uint4 l of a board (l = psub*16 + cq): byte q = channels cq*8..cq*8+7 of position q*4 + psub.  Two threads share a
// uint4 (q 0..7 / 8..15); each expands 8 bytes to 8 chunks of 8 elements (1.0 = E::ONE) and writes them to its LDS rows.
template <int WGB, int PAD, class E = ElemBF16, bool SWZ = false>
__device__ __forceinline__ void stage_tile_bits(unsigned char* lds, const uint16_t* __restrict__ in, int board0, int n_boards) {
    constexpr int PITCH = 128 * 2 + PAD;
    const uint4* src = (const uint4*)in + (size_t)board0 * 64;
    for (int t = threadIdx.x; t < WGB * 128; t += 256) {
        const int board = t >> 7, half = (t >> 6) & 1, l = t & 63, psub = l >> 4, cq = l & 15;
        uint4 v = (board0 + board < n_boards) ? src[board * 64 + l] : make_uint4(0, 0, 0, 0);
        const uint32_t w0 = half ? v.z : v.x, w1 = half ? v.w : v.y;
#pragma unroll
        for (int qq = 0; qq < 8; qq++) {
            const uint32_t byte = ((qq < 4 ? w0 : w1) >> ((qq & 3) * 8)) & 0xFFu;
            uint4 o;
            o.x = ((byte & 1) ? E::ONE : 0u) | ((byte & 2) ? (E::ONE << 16) : 0u);
            o.y = ((byte & 4) ? E::ONE : 0u) | ((byte & 8) ? (E::ONE << 16) : 0u);
            o.z = ((byte & 16) ? E::ONE : 0u) | ((byte & 32) ? (E::ONE << 16) : 0u);
            o.w = ((byte & 64) ? E::ONE : 0u) | ((byte & 128) ? (E::ONE << 16) : 0u);
            const int pos = (half * 8 + qq) * 4 + psub;
            *(uint4*)(lds + (board * 64 + pos) * PITCH + (cq ^ (SWZ ? (pos >> 2) & 1 : 0)) * 16) = o;
        }
    }
    for (int c = threadIdx.x; c < NN_ZERO16 / 16; c += 256) *(uint4*)(lds + WGB * 64 * PITCH + c * 16) = make_uint4(0, 0, 0, 0);
}

// One weight fragment through a buffer descriptor: UNIFORM base (kernel argument -> 4 SGPRs) + uniform element offset (soffset, SGPR)
// + the lane's constant 32-bit byte offset (voffset, one VGPR).  With flat global_load hipcc carried a 64-bit per-lane pointer per
// weight stream and hoisted a dozen of them out of the tile loop of the persistent tower: those were its 24 spilled VGPRs (100 B/lane of
// scratch, reloaded with vmcnt(0) waits once per tile — never inside a K loop, but serialising the first prefetch of every tile).
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
#ifndef NN_WBUF
#define NN_WBUF 1                                          // 0: A/B build with flat global loads for the weight stream (SIGMAZERO_EXTRA_FLAGS=-DNN_WBUF=0)
#endif
#ifndef NN_ROWSKIP
#define NN_ROWSKIP 1                                       // 0: A/B build that multiplies the all-zero border-row tiles too
#endif
struct WSrc { __amdgpu_buffer_rsrc_t r; const uint4* p; };
__device__ __forceinline__ WSrc wfrag_rsrc(const uint4* __restrict__ w) {
    WSrc s;
    s.r = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, 0x7FFFFFFF, 0x00020000);   // raw buffer, no stride; weights of one conv are < 2 MB
    s.p = w;
    return s;
}
__device__ __forceinline__ uint4 ld_wfrag(const WSrc& s, size_t uniform_off, uint32_t lane_bytes) {
#if NN_WBUF
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(s.r, (int)lane_bytes, (int)(uniform_off * 16), 0);
    return make_uint4(v.x, v.y, v.z, v.w);
#else
    return *(const uint4*)((const char*)(s.p + uniform_off) + lane_bytes);
#endif
}

// Which 16 positions form MFMA position tile j: image row (= board*64 + position) of the tile's column p16.
//   2-board workgroups: tile j = board ROW j of BOTH boards (lanes 0..7 board 0, 8..15 board 1).  Under the three taps that look one row up (dy = -1)
//     tile 0 reads nothing but off-board zeros, under the three that look down tile 7 does: those 6 of 72 tap-tiles are skipped outright
//     (conv_kloop16 SKIPROWS, 8.3 % of the MFMAs).  Eight consecutive lanes still read eight consecutive image rows: ds_read_b128 stays conflict-free.
//   1-board workgroups: tile j = rows 2j, 2j+1 of the board (no tile is ever entirely off the board).
template <int WGB>
__device__ __forceinline__ int tile_row(int j, int p16) {
    if constexpr (WGB == 2) return (p16 >> 3) * 64 + j * 8 + (p16 & 7);
    else return (j >> 2) * 64 + (j & 3) * 16 + p16;
}

// LDS byte offset (inside its image) that lane (p16, kg) reads for position tile j under tap `tap` — see conv_kloop16 for the zero region
// SWZ (the split-precision images): the 16-byte chunk index of a row's data is XOR-ed with bit 2 of the row's position (chunk ^ ((row >> 2) & 1), i.e. the two
// lane quarters kg, kg^1 trade places on rows 4..7 of every 8): it makes the epilogue's ds_write_b128 (8 lanes = 8 consecutive rows at one channel offset)
// conflict-free and leaves ds_read_b128 conflict-free (slot 2*row + (kg ^ g) is still a permutation per lane group).
template <int PITCH, int NTAPS, int WGB, bool SWZ = false>
__device__ __forceinline__ int conv_tap_addr16(int tap, int j, int p16, int kg) {
    const int dy = (NTAPS == 9) ? tap / 3 - 1 : 0, dx = (NTAPS == 9) ? tap % 3 - 1 : 0;
    const int row0 = tile_row<WGB>(j, p16), pos = row0 & 63;   // position inside its board (board = row0 >> 6)
    const int y = (pos >> 3) + dy, x = (pos & 7) + dx;
    const bool ok = (unsigned)y < 8u && (unsigned)x < 8u;
    const int vrow = pos + 8 * dy + dx;
    return ok ? ((row0 >> 6) * 64 + y * 8 + x) * PITCH + (SWZ ? (kg ^ ((x >> 2) & 1)) : kg) * 16 : WGB * 64 * PITCH + ((2 * vrow + kg) & 15) * 16;
}

#define HIPCHK(x) do { hipError_t _e = (x); if (_e != hipSuccess) { fprintf(stderr, "[sigmazero] HIP error %s at %s:%d\n", hipGetErrorString(_e), __FILE__, __LINE__); return SZ_ERR_HIP; } } while (0)

// the network entry points carry no engine handle: they run on the device that owns the caller's stream (a rank whose current
// device is another GPU would otherwise launch into the wrong context) and restore the caller's current device on return
struct StreamDeviceGuard {
    int prev = -1; bool changed = false;
    explicit StreamDeviceGuard(void* stream) {
        hipDevice_t sd = 0;
        if (!stream || hipStreamGetDevice((hipStream_t)stream, &sd) != hipSuccess) return;     // the null stream belongs to the current device
        if (hipGetDevice(&prev) == hipSuccess && prev != (int)sd) changed = (hipSetDevice((int)sd) == hipSuccess);
    }
    ~StreamDeviceGuard() { if (changed) (void)hipSetDevice(prev); }
};

#define NN_MAX_DEVICES 16
static inline int current_device_slot() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
    return dev % NN_MAX_DEVICES;
}
static inline int device_cus() {
    static int n_cu[NN_MAX_DEVICES] = {};
    int& n = n_cu[current_device_slot()];
    if (!n) {
        int dev = 0; hipDeviceProp_t prop;
        n = 256;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) n = prop.multiProcessorCount;
    }
    return n;
}
