// sz_engine.hip — batched MCTS self-play engine for MI355X (gfx950), hand-written HIP.
//
// Replaces, for thousands of concurrent boards, the per-game Python tree of the reference:
//   MCTS0.search            /root/reference/mcts.py:39-122
//   Node.select/get_ucb     /root/reference/mctsnode.py:23-37
//   Node.expand             /root/reference/mctsnode.py:39-54
//   Node.backpropagate      /root/reference/mctsnode.py:56-63
//   ChessTensor.move_piece / get_representation / get_value_and_terminated
//                           /root/reference/chess_tensor.py:88-172
//   actionsToTensor / tensorToAction (legal-move mask, child order)  chess_tensor.py:190-410
//   play_game's sampling + bookkeeping                               /root/reference/sim.py:46-97
//
// Execution model: ONE WAVEFRONT (64 lanes) OWNS ONE BOARD.
//   * tree = flat SoA in HBM per board: EdgeStat{W f64, N i32, P f32} (16 B, one dwordx4 per child,
//     a node's children are contiguous => coalesced span reads), EdgeMeta (16 B, read only for the
//     selected child), position records (80 B) for visited nodes only;
//   * select: lanes = children, UCB in the reference's exact fp32 op order, wave64 butterfly argmax
//     with lowest-index tie-break (torch.argmax semantics);
//   * move generation: lanes = squares (sz_chess.h), 73 ballots produce the legal-move mask directly
//     in action-index order, so children come out sorted with prefix popcounts (no sort);
//   * expand: masked renormalise with a fixed summation order (lane-strided partials + xor butterfly),
//     bit-identical to oracle/oc_mcts.c; backprop: the descent path sits in LDS, one lane per level,
//     no atomics (one leaf per board per step) => bitwise deterministic;
//   * encode: history bitboards staged in LDS; NCHW f32/bf16 (one 8-cell row per lane per store), NHWC bf16, or the
//     bit-packed NHWC image of 1 KiB per board (one 16-byte store per lane) that the MFMA stem expands itself.
// No MFMA here by design: this is latency/HBM-bound integer work; the network is the MFMA consumer.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "sz_chess.h"
#include "../../include/sigmazero.h"

// ------------------------------------------------------------------------------------------------
// device data
// ------------------------------------------------------------------------------------------------
struct alignas(16) EdgeStat { double W; int N; float P; };
struct alignas(16) EdgeMeta { int first; int node; unsigned short n; unsigned short action; signed char term; signed char tval; unsigned short pad; };

enum { ST_ACTIVE = 1, ST_PENDING = 2, ST_DONE = 4, ST_GAMEOVER = 8, ST_ERROR = 16, ST_SEARCHING = 32 };

struct alignas(16) Ctl {
    int status, n_nodes, n_edges, sims_done;
    int pend_node, pend_depth, game_ply, err;
    unsigned long long n_expand, n_term, sum_depth, sum_k;
    int max_edges, game_result, reuse_ready, pad1;      // reuse_ready: the store holds the subtree of the move just played (sz_config.reuse_subtree)
};

struct View {
    int B, S, n_cap, e_cap, p_cap, learning, chess960, planes_dtype;
    float c_puct, noise;
    unsigned long long* dbg;    // diagnostic (sz_debug_step_stamps): 8 u64 per board, s_memtime at the phase boundaries of k_search_step; NULL = off
    int reuse;                  // NON-REFERENCE option (sz_config.reuse_subtree): keep the chosen child's subtree as the next search's tree
    int* nmap;                  // reuse only: [B][n_cap] scratch map node id -> new index of the edge that owns it, during k_play's compaction
    const int* slot;            // optional (sz_compact): row of board b in the network batch (planes / policy / value); NULL = identity
    const float* root_gamma;    // optional (non-reference) true Dirichlet root noise: [B][SZ_MAX_MOVES] Gamma(alpha,1) draws; NULL = reference behaviour
    SzPos* npos; SzPos* ring; EdgeStat* es; EdgeMeta* em; int* gpath; u64* pmask; Ctl* ctl;
    // per-ply training record
    uint8_t* rec_planes; int* rec_action; int* rec_visits; int* rec_nchild; uint8_t* rec_colour; int* rec_chosen;
    uint8_t* rec_over; int8_t* rec_result; uint8_t* rec_active;
};

#define PMASK_STRIDE 80     // u64 words per board (73 used)
#define LDS_HIST_WORDS 64   // 8 history slots x 8 words
#define LDS_MASK_WORDS 80

// ------------------------------------------------------------------------------------------------
// wave helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int uni(int x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ u64 uni64(u64 x) {
    u32 lo = (u32)__builtin_amdgcn_readfirstlane((int)(u32)x), hi = (u32)__builtin_amdgcn_readfirstlane((int)(u32)(x >> 32));
    return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ SzPos load_pos(const SzPos* ptr) {
    SzPos p = *ptr;
    for (int k = 0; k < 6; k++) p.pc[k] = uni64(p.pc[k]);
    p.white = uni64(p.white); p.meta = uni64(p.meta); p.key = uni64(p.key); p.castling = uni64(p.castling);
    return p;
}
__device__ __forceinline__ float wave_sum_butterfly(float v) {
    for (int off = 32; off >= 1; off >>= 1) v = v + __shfl_xor(v, off);
    return v;
}
__device__ __forceinline__ size_t planes_board_bytes(int dtype) {
    return dtype == SZ_PLANES_F32 ? (size_t)SZ_NUM_PLANES * 64 * 4 : (dtype == SZ_PLANES_BF16 ? (size_t)SZ_NUM_PLANES * 64 * 2 :
           (dtype == SZ_PLANES_NHWC128_BITS ? (size_t)1024 : (size_t)64 * 128 * 2));
}
// lane-indexed ballot (bit v = view square v) -> real-square bitboard
__device__ __forceinline__ u64 view_to_squares(u64 m, int white) {
    return white ? __builtin_bswap64(m) : __builtin_bswap64(sz_brev(m));
}

// ------------------------------------------------------------------------------------------------
// Node.get_ucb (mctsnode.py:33-37) in torch's float32 operation order; see oracle/oc_mcts.c:oc_ucb.
// Compiled with -ffp-contract=off: each operator is one IEEE rounding.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float ucb_value(int vc, double wsum, float prior, float sqrt_parent, float c) {
    float vsum = (float)wsum;
    float t1 = (float)vc + 1e-6f;
    float q = 1.0f - ((vsum / t1) + 1.0f) / 2.0f;
    float r = 1.0f / (float)(vc + 1);
    float u = ((r * sqrt_parent) * c) * prior;
    return q + u;
}

// Node.select (mctsnode.py:23-31): argmax of get_ucb over the n contiguous children, first maximum wins (torch.argmax).
// Lanes = children; wave64 xor-butterfly with lowest-index tie-break.  ucb_out (optional, test hook) receives every child's value.
__device__ __forceinline__ int wave_select_child(const EdgeStat* ch, int n, int parentN, float c_puct, float* ucb_out) {
    const int lane = lane_id();
    const float sq = (float)sqrt((double)parentN);                      // math.sqrt(self.visit_count) in double, then a float32 operand
    float best = 0.f; int bi = 0x7fffffff;
    for (int c = lane; c < n; c += 64) {
        EdgeStat s = ch[c];
        float u = ucb_value(s.N, s.W, s.P, sq, c_puct);
        if (ucb_out) ucb_out[c] = u;
        if (bi == 0x7fffffff || u > best) { best = u; bi = c; }
    }
    for (int off = 32; off >= 1; off >>= 1) {
        float ob = __shfl_xor(best, off); int oi = __shfl_xor(bi, off);
        if (oi != 0x7fffffff && (bi == 0x7fffffff || ob > best || (ob == best && oi < bi))) { best = ob; bi = oi; }
    }
    return uni(bi);
}

struct BoardPtrs {
    SzPos* npos; SzPos* ring; EdgeStat* es; EdgeMeta* em; int* gpath; u64* pmask; Ctl* ctl;
};
__device__ __forceinline__ BoardPtrs board_ptrs(const View& v, int b) {
    BoardPtrs p;
    p.npos = v.npos + (size_t)b * v.n_cap;
    p.ring = v.ring + (size_t)b * SZ_RING;
    p.es = v.es + (size_t)b * v.e_cap;
    p.em = v.em + (size_t)b * v.e_cap;
    p.gpath = v.gpath + (size_t)b * v.p_cap;
    p.pmask = v.pmask + (size_t)b * PMASK_STRIDE;
    p.ctl = v.ctl + b;
    return p;
}

// position `j` plies before a node at tree depth D whose root sits at game ply root_ply
__device__ __forceinline__ const SzPos* ancestor_ptr(const BoardPtrs& bp, const int* path, int D, int root_ply, int j) {
    int dj = D - j;
    if (dj >= 0) return bp.npos + bp.em[path[dj]].node;
    int ply = root_ply + dj;
    if (ply < 0 || -dj >= SZ_RING) return nullptr;
    return bp.ring + (ply & (SZ_RING - 1));
}

// ------------------------------------------------------------------------------------------------
// Move generation for position X (uniform), one lane per square.  Fills mask[0..72] in LDS and
// returns n_legal / ep_legal / checkers (uniform).
// ------------------------------------------------------------------------------------------------
__device__ void wave_movegen(const SzPos& X, int chess960, u64* mask_lds, int& n_legal, int& ep_legal, u64& checkers) {
    const int lane = lane_id();
    SzInfo I = sz_info(X);
    const int s = lane ^ sz_view_flip(I.white);
    const bool need = (I.need >> s) & 1;
    bool att = false;
    if (need) att = sz_danger_at(X, I, s);
    u64 danger = view_to_squares(__ballot(att), I.white);
    u64 T;
    if (s == I.ksq) T = sz_king_targets(X, I, danger, chess960);
    else T = sz_piece_targets(X, I, s);
    const bool is_pawn = (X.pc[SZ_P] >> s) & 1;
    int ep = szm_ep(X.meta);
    ep_legal = (ep >= 0) ? (__ballot(is_pawn && ((T >> ep) & 1)) != 0) : 0;
    // legal-move mask in action-index order (chess_tensor.py:190-218): 73 ballots over the lane's plane bits (sz_lane_plane_bits)
    const SzPlaneBits pb = sz_lane_plane_bits(T, is_pawn, lane, I.white);
    int n = 0;
#pragma unroll
    for (int pl = 0; pl < SZ_MASK_WORDS; pl++) {
        const uint32_t word = pl < 56 ? pb.q[pl / 7] : (pl < 64 ? pb.kn : pb.up);
        const int sh = pl < 56 ? pl % 7 : (pl < 64 ? pl - 56 : (pl - 64) % 3);
        const u64 w = __ballot((word >> sh) & 1u);
        if (lane == 0) mask_lds[pl] = w;
        n += __popcll(w);
    }
    n_legal = n;
    checkers = I.checkers;
}

// earlier occurrences of X (key) within the reversible window: Board.is_repetition walk-back, lanes = plies
__device__ int wave_count_reps(const BoardPtrs& bp, const int* path, int D, int root_ply, const SzPos& X) {
    if (szm_irrev(X.meta)) return 0;
    const int lane = lane_id();
    int window = szm_half(X.meta);
    if (window > SZ_RING - 1) window = SZ_RING - 1;
    int reps = 0;
    for (int base = 0; base < window; base += 64) {
        int j = base + lane + 1;
        bool in = j <= window;
        const SzPos* a = in ? ancestor_ptr(bp, path, D, root_ply, j) : nullptr;
        bool stop = in && (a == nullptr);
        bool eq = false;
        if (a) { u64 k = a->key, m = a->meta; eq = (k == X.key); stop = szm_irrev(m) != 0; }
        u64 eqm = __ballot(eq), stm = __ballot(stop || !in);
        if (stm) {
            int sidx = __builtin_ctzll(stm);
            u64 keep = (sidx >= 63) ? ~0ULL : ((2ULL << sidx) - 1);
            reps += __popcll(eqm & keep);
            break;
        }
        reps += __popcll(eqm);
    }
    return reps > 4 ? 4 : reps;
}

// stage the 8 history positions (leaf + 7 ancestors) as 8x8 u64 words in LDS
__device__ void wave_load_history(const BoardPtrs& bp, const int* path, int D, int root_ply, const SzPos& X, u64* hist_lds) {
    const int lane = lane_id();
    const int t = lane >> 3, w = lane & 7;
    u64 val = 0;
    if (t == 0) {
        val = w == 0 ? X.pc[0] : w == 1 ? X.pc[1] : w == 2 ? X.pc[2] : w == 3 ? X.pc[3] : w == 4 ? X.pc[4] : w == 5 ? X.pc[5] : w == 6 ? X.white : X.meta;
    } else {
        const SzPos* a = ancestor_ptr(bp, path, D, root_ply, t);
        if (a) val = ((const u64*)a)[w];
    }
    hist_lds[lane] = val;
}

// chess_tensor.py:131-142 get_representation for the leaf whose history sits in hist_lds.
// Every lane owns (plane, row): 8 cells = one 32-byte (f32) or 16-byte (bf16) store, fully coalesced.
__device__ void wave_encode(const u64* hist_lds, const SzPos& X, void* out_board, int dtype, uint8_t* packed_out) {
    const int lane = lane_id();
    const int vw = szm_turn(X.meta);
    if (dtype == SZ_PLANES_NHWC128_BITS && out_board) {
        // bit-packed NHWC image, 1 KiB per board = ONE 16-byte store per lane: lane l = psub*16 + cq owns channels
        // cq*8..cq*8+7; byte q of its uint4 = those 8 channel bits at position q*4 + psub (view order).  The stem
        // kernel (csrc/sz_nn.hip stage_tile_bits) expands the bits to bf16 while staging its LDS tile.
        const int cq = lane & 15, psub = lane >> 4;
        uint32_t w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int c = cq * 8 + k;
            u64 bb = (c < 112) ? sz_hist_plane(hist_lds + (c / 14) * 8, c % 14, vw) : (c < SZ_NUM_PLANES ? sz_aux_plane(X, c - 112) : 0ULL);
            // view position p <-> board square p ^ sz_view_flip: ^56 (white) = byte swap, ^7 (black) = bit reversal + byte swap
            bb = __builtin_bswap64(vw ? bb : __brevll(bb));
            bb >>= psub;
            const uint32_t lo = (uint32_t)bb, hi = (uint32_t)(bb >> 32);
#pragma unroll
            for (int q = 0; q < 8; q++) {
                w[q >> 2] |= ((lo >> (4 * q)) & 1u) << ((q & 3) * 8 + k);
                w[2 + (q >> 2)] |= ((hi >> (4 * q)) & 1u) << ((q & 3) * 8 + k);
            }
        }
        ((uint4*)out_board)[lane] = make_uint4(w[0], w[1], w[2], w[3]);
        if (!packed_out) return;
        out_board = nullptr;
    }
    if (dtype == SZ_PLANES_NHWC128_BF16 && out_board) {
        // NHWC image [64 positions][128 channels] bf16 = 16 KB = 16 wave-instructions of 1 KiB, each fully contiguous:
        // in store q lane l covers position q*4 + (l>>4), channels (l&15)*8 .. +7.  A lane's 8 channel bitboards do not
        // depend on q, so they are formed once (8 planes per lane instead of 119) and only shifted per position.
        const int cq = lane & 15, psub = lane >> 4, flip = sz_view_flip(vw);
        u64 bb[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int c = cq * 8 + k;
            bb[k] = (c < 112) ? sz_hist_plane(hist_lds + (c / 14) * 8, c % 14, vw) : (c < SZ_NUM_PLANES ? sz_aux_plane(X, c - 112) : 0ULL);
        }
        uint4* dst = (uint4*)out_board + lane;
#pragma unroll 4
        for (int q = 0; q < 16; q++) {
            const int sq = (q * 4 + psub) ^ flip;
            uint4 o;
            o.x = (((bb[0] >> sq) & 1) ? 0x3F80u : 0u) | (((bb[1] >> sq) & 1) ? 0x3F800000u : 0u);
            o.y = (((bb[2] >> sq) & 1) ? 0x3F80u : 0u) | (((bb[3] >> sq) & 1) ? 0x3F800000u : 0u);
            o.z = (((bb[4] >> sq) & 1) ? 0x3F80u : 0u) | (((bb[5] >> sq) & 1) ? 0x3F800000u : 0u);
            o.w = (((bb[6] >> sq) & 1) ? 0x3F80u : 0u) | (((bb[7] >> sq) & 1) ? 0x3F800000u : 0u);
            dst[q * 64] = o;
        }
        if (!packed_out) return;
        out_board = nullptr;
    }
    const int r = lane & 7;
    for (int i = 0; i < 15; i++) {
        int c = i * 8 + (lane >> 3);
        if (c >= SZ_NUM_PLANES) break;
        u64 bb = (c < 112) ? sz_hist_plane(hist_lds + (c / 14) * 8, c % 14, vw) : sz_aux_plane(X, c - 112);
        uint32_t bits = sz_row_bits(bb, r, vw);
        if (packed_out) packed_out[c * 8 + r] = (uint8_t)bits;
        if (!out_board) continue;
        if (dtype == SZ_PLANES_F32) {
            float4* dst = (float4*)out_board + (size_t)(c * 8 + r) * 2;
            float4 a, b2;
            a.x = (bits & 1) ? 1.f : 0.f; a.y = (bits & 2) ? 1.f : 0.f; a.z = (bits & 4) ? 1.f : 0.f; a.w = (bits & 8) ? 1.f : 0.f;
            b2.x = (bits & 16) ? 1.f : 0.f; b2.y = (bits & 32) ? 1.f : 0.f; b2.z = (bits & 64) ? 1.f : 0.f; b2.w = (bits & 128) ? 1.f : 0.f;
            dst[0] = a; dst[1] = b2;
        } else {
            uint4 o;                                       // bf16 1.0 = 0x3F80
            o.x = ((bits & 1) ? 0x3F80u : 0u) | ((bits & 2) ? 0x3F800000u : 0u);
            o.y = ((bits & 4) ? 0x3F80u : 0u) | ((bits & 8) ? 0x3F800000u : 0u);
            o.z = ((bits & 16) ? 0x3F80u : 0u) | ((bits & 32) ? 0x3F800000u : 0u);
            o.w = ((bits & 64) ? 0x3F80u : 0u) | ((bits & 128) ? 0x3F800000u : 0u);
            ((uint4*)out_board)[c * 8 + r] = o;
        }
    }
}

// Node.backpropagate: one lane per level of the descent path (path[0] = root edge ... path[d] = leaf edge)
__device__ __forceinline__ void wave_backprop(EdgeStat* es, const int* path, int d, double v) {
    for (int j = lane_id(); j <= d; j += 64) {
        EdgeStat* e = es + path[j];
        double sv = ((d - j) & 1) ? -v : v;
        e->W = e->W + sv;
        e->N = e->N + 1;
    }
    __threadfence_block();
}

// make a new position from `parent` by action index and finish it (movegen, key, repetition, terminal).
// D = tree depth of the new position; path[0..D-1] are the edges above it.
__device__ SzPos wave_create_position(const BoardPtrs& bp, const SzPos& parent, int action, int chess960, const int* path, int D,
                                      int root_ply, u64* mask_lds) {
    int from, to, promo;
    sz_action_decode(parent, action, from, to, promo);
    SzPos X = sz_make_move(parent, from, to, promo, chess960);
    int n_legal, ep_legal; u64 checkers;
    wave_movegen(X, chess960, mask_lds, n_legal, ep_legal, checkers);
    X.key = sz_hash_key(X, ep_legal);
    int reps = wave_count_reps(bp, path, D, root_ply, X);
    X.meta = sz_finish_meta(X, checkers, n_legal, ep_legal, reps);
    return X;
}

// ------------------------------------------------------------------------------------------------
// kernel: start games (ChessTensor.start_board)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_new_games(View v, const int* scharnagl, const uint8_t* active) {
    extern __shared__ u64 lds64[];
    const int b = blockIdx.x;
    BoardPtrs bp = board_ptrs(v, b);
    if (active && !active[b]) return;
    int n = scharnagl ? scharnagl[b] : -1;
    SzPos X = sz_startpos(n);
    int n_legal, ep_legal; u64 checkers;
    wave_movegen(X, v.chess960, lds64 + LDS_HIST_WORDS, n_legal, ep_legal, checkers);
    X.key = sz_hash_key(X, ep_legal);
    X.meta = sz_finish_meta(X, checkers, n_legal, ep_legal, 0);
    if (lane_id() == 0) {
        bp.ring[0] = X;
        Ctl c = *bp.ctl;                                  // the cumulative counters (n_expand, n_term, sum_depth, sum_k, max_edges) survive a refill
        c.status = ST_ACTIVE; c.n_nodes = c.n_edges = c.sims_done = 0; c.pend_node = c.pend_depth = c.game_ply = c.err = 0; c.game_result = 0; c.reuse_ready = 0;
        *bp.ctl = c;
    }
}

__global__ void k_set_active(View v, const uint8_t* active) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= v.B) return;
    Ctl* c = v.ctl + b;
    int st = c->status & ~(ST_ACTIVE);
    if (active[b]) st |= ST_ACTIVE;
    c->status = st;
}

// network-batch rows for the boards that will search (active, game not over, no error), in board order: one workgroup, chunked scan
__global__ __launch_bounds__(1024) void k_compact(View v, int* slot, int* n_live) {
    __shared__ int cnt[1024];
    const int t = threadIdx.x, per = (v.B + 1023) / 1024, lo = t * per, hi = min(v.B, lo + per);
    int c = 0;
    for (int b = lo; b < hi; b++) { const int st = v.ctl[b].status; c += ((st & ST_ACTIVE) && !(st & (ST_GAMEOVER | ST_ERROR))) ? 1 : 0; }
    cnt[t] = c;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {                     // inclusive Hillis-Steele scan over the 1024 chunk counts
        int add = t >= off ? cnt[t - off] : 0;
        __syncthreads();
        cnt[t] += add;
        __syncthreads();
    }
    int base = cnt[t] - c;
    for (int b = lo; b < hi; b++) {
        const int st = v.ctl[b].status;
        const bool live = (st & ST_ACTIVE) && !(st & (ST_GAMEOVER | ST_ERROR));
        slot[b] = live ? base++ : -1;
    }
    if (t == 1023) *n_live = cnt[1023];
}

// finish an uploaded game record (status only; the ring was copied by the host)
__global__ void k_after_upload(View v, int b, int ply) {
    Ctl* c = v.ctl + b;
    Ctl z = *c;                                           // cumulative counters kept, per-game fields reset
    z.status = ST_ACTIVE; z.n_nodes = z.n_edges = z.sims_done = 0; z.pend_node = z.pend_depth = z.err = 0; z.game_result = 0; z.game_ply = ply; z.reuse_ready = 0;
    *c = z;
}

// ------------------------------------------------------------------------------------------------
// kernel: search begin — create roots (mcts.py:43-46), root movegen + terminal test, encode root planes
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_search_begin(View v, void* planes) {
    extern __shared__ u64 lds64[];
    u64* hist = lds64; u64* mask = lds64 + LDS_HIST_WORDS; int* path = (int*)(lds64 + LDS_HIST_WORDS + LDS_MASK_WORDS);
    const int b = blockIdx.x, lane = lane_id();
    BoardPtrs bp = board_ptrs(v, b);
    int status = uni(bp.ctl->status);
    if (v.rec_active && lane == 0) v.rec_active[b] = 0;
    if (!(status & ST_ACTIVE) || (status & (ST_GAMEOVER | ST_ERROR))) {
        if (lane == 0) bp.ctl->status = status & ~(ST_PENDING | ST_DONE | ST_SEARCHING);
        return;
    }
    if (v.slot && uni(v.slot[b]) < 0) {
        // the board became live after the last sz_compact: it has no row in the network batch.  Refuse loudly instead of searching nothing.
        if (lane == 0) { bp.ctl->status = (status & ~(ST_PENDING | ST_DONE | ST_SEARCHING)) | ST_ERROR; if (!bp.ctl->err) bp.ctl->err = SZ_ERR_STATE; }
        return;
    }
    const int root_ply = uni(bp.ctl->game_ply);
    SzPos X = load_pos(bp.ring + (root_ply & (SZ_RING - 1)));
    path[0] = 0;
    if (v.reuse && uni(bp.ctl->reuse_ready) && uni(bp.ctl->n_nodes) <= v.S && 2 * (long long)uni(bp.ctl->n_edges) <= v.e_cap && v.S > 0) {
        // NON-REFERENCE option: the store already holds the subtree below the move just played (k_play compacted it to the front: edge 0 /
        // node 0 = that child, now the root, with its visit count and value sum).  The search continues on it: nothing to expand here, so
        // the board does not wait for the network; sz_search_begin's descent-only launch selects its first leaf.
        wave_load_history(bp, path, 0, root_ply, X, hist);
        __syncthreads();
        wave_encode(hist, X, nullptr, v.planes_dtype, v.rec_planes ? v.rec_planes + (size_t)b * SZ_NUM_PLANES * 8 : nullptr);   // training record of the root only
        if (lane == 0) {
            bp.npos[0] = X;
            Ctl* c = bp.ctl;
            c->status = (status & ST_ACTIVE) | ST_SEARCHING; c->sims_done = 0; c->pend_node = -1; c->pend_depth = 0; c->reuse_ready = 0;
            if (v.rec_colour) v.rec_colour[b] = (uint8_t)szm_turn(X.meta);
        }
        return;
    }
    if (lane == 0) {
        bp.ctl->reuse_ready = 0;
        bp.npos[0] = X;
        EdgeStat s; s.W = 0.0; s.N = 1; s.P = 0.f;                 // root.visit_count = 1 (mcts.py:46)
        bp.es[0] = s;
        EdgeMeta m; m.first = -1; m.node = 0; m.n = 0; m.action = 0; m.term = (signed char)szm_term(X.meta);
        m.tval = (signed char)(szm_loss(X.meta) ? -1 : 0); m.pad = 0;
        bp.em[0] = m;
        bp.gpath[0] = 0;
    }
    int st = (status & ST_ACTIVE) | ST_SEARCHING;
    if (szm_term(X.meta) || v.S <= 0) {
        // a terminal root: every simulation re-visits it (mcts.py:104-109); children stay empty
        if (lane == 0) {
            int S = v.S > 0 ? v.S : 0;
            double tv = szm_loss(X.meta) ? -1.0 : 0.0;
            bp.es[0].W = tv * (double)S; bp.es[0].N = 1 + S;
            Ctl* c = bp.ctl; c->status = st | ST_DONE; c->n_nodes = 1; c->n_edges = 1; c->sims_done = S;
            c->n_term += (unsigned long long)S; c->pend_node = -1; c->pend_depth = 0;
        }
        return;
    }
    int n_legal, ep_legal; u64 checkers;
    wave_movegen(X, v.chess960, mask, n_legal, ep_legal, checkers);
    __syncthreads();
    for (int i = lane; i < SZ_MASK_WORDS; i += 64) bp.pmask[i] = mask[i];
    wave_load_history(bp, path, 0, root_ply, X, hist);
    __syncthreads();
    const int row = v.slot ? uni(v.slot[b]) : b;              // network batch row of this board (compacted batches: live boards first)
    wave_encode(hist, X, (planes && row >= 0) ? (char*)planes + (size_t)row * planes_board_bytes(v.planes_dtype) : nullptr, v.planes_dtype,
                v.rec_planes ? v.rec_planes + (size_t)b * SZ_NUM_PLANES * 8 : nullptr);
    if (lane == 0) {
        Ctl* c = bp.ctl;
        c->status = st | ST_PENDING; c->n_nodes = 1; c->n_edges = 1; c->sims_done = 0; c->pend_node = 0; c->pend_depth = 0;
        if (v.rec_colour) v.rec_colour[b] = (uint8_t)szm_turn(X.meta);
    }
}

// ------------------------------------------------------------------------------------------------
// kernel: one lock-step iteration (expand + backprop of the evaluated leaf, then select / move /
// terminal test / encode of the next one)
// ------------------------------------------------------------------------------------------------
#define STEP_STAMP(k) do { if (v.dbg) { unsigned long long _t = __builtin_amdgcn_s_memtime(); if (lane_id() == 0) v.dbg[(size_t)blockIdx.x * 8 + (k)] = _t; } } while (0)
__global__ __launch_bounds__(64, 4) void k_search_step(View v, const float* __restrict__ policy, const float* __restrict__ value, void* planes) {
    extern __shared__ u64 lds64[];
    u64* hist = lds64; u64* mask = lds64 + LDS_HIST_WORDS; int* path = (int*)(lds64 + LDS_HIST_WORDS + LDS_MASK_WORDS);
    const int b = blockIdx.x, lane = lane_id();
    BoardPtrs bp = board_ptrs(v, b);
    int status = uni(bp.ctl->status);
    if (!(status & ST_SEARCHING) || (status & (ST_DONE | ST_ERROR))) return;
    int n_nodes = uni(bp.ctl->n_nodes), n_edges = uni(bp.ctl->n_edges), sims = uni(bp.ctl->sims_done);
    const int root_ply = uni(bp.ctl->game_ply);
    const int row = v.slot ? uni(v.slot[b]) : b;              // network batch row of this board
    if (row < 0) return;
    if ((status & ST_PENDING) && !policy) return;             // descent-only launch (sz_search_begin with reuse): boards that already wait for the network sit it out
    unsigned long long n_expand = 0, n_term = 0, sum_depth = 0, sum_k = 0;
    int err = 0;
    STEP_STAMP(0);

    if (status & ST_PENDING) {
        // ---- mcts.py:77-109 for the leaf evaluated by the network -------------------------------
        const int d = uni(bp.ctl->pend_depth), node = uni(bp.ctl->pend_node);
        for (int j = lane; j <= d; j += 64) path[j] = bp.gpath[j];
        for (int i = lane; i < SZ_MASK_WORDS; i += 64) mask[i] = bp.pmask[i];
        __syncthreads();
        const float* pol = policy + (size_t)row * SZ_NUM_ACTIONS;
        // masked sum in the fixed order: per-lane partial over planes ascending, then xor butterfly.
        // Only planes that hold a legal move are fetched (~20 of 73), and in groups of 16 INDEPENDENT loads: a load-wait-add chain per
        // plane cost one full memory latency each, and fetching all 73 planes made the kernel bandwidth-bound at 4096 boards.  The
        // value and action of every legal move are stashed in LDS in ascending action order, so the second half works on the K <= 218
        // children directly (lane = child: coalesced child records) instead of walking the planes again.
        const int path_alloc = v.p_cap > 256 ? v.p_cap : 256;
        float* sval = (float*)(path + path_alloc);                          // [<= 218] policy value of legal move c
        unsigned short* sact = (unsigned short*)(sval + 220);               // [<= 218] its action index
        u64 ne0 = __ballot(mask[lane] != 0);                                // planes 0..63 with at least one legal move
        u64 ne1 = __ballot(lane < SZ_MASK_WORDS - 64 && mask[64 + (lane < SZ_MASK_WORDS - 64 ? lane : 0)] != 0);   // planes 64..72
        constexpr int PCH = 16;
        float acc = 0.0f;
        int n_moves = 0;
        while (ne0 | ne1) {
            int pls[PCH];
#pragma unroll
            for (int k = 0; k < PCH; k++) {                                  // next PCH non-empty planes, ascending (uniform)
                int pl = -1;
                if (ne0) { pl = __builtin_ctzll(ne0); ne0 &= ne0 - 1; }
                else if (ne1) { pl = 64 + __builtin_ctzll(ne1); ne1 &= ne1 - 1; }
                pls[k] = pl;
            }
            float pv[PCH];
#pragma unroll
            for (int k = 0; k < PCH; k++) pv[k] = pol[(pls[k] >= 0 ? pls[k] : 0) * 64 + lane];     // issued back to back; plane 0 stands in for "none"
#pragma unroll
            for (int k = 0; k < PCH; k++) {
                const int pl = pls[k];
                if (pl >= 0) {                                               // uniform
                    const bool mine = (mask[pl] >> lane) & 1;
                    if (mine) acc = acc + pv[k];
                    const u64 mm = __ballot(mine);
                    if (mine) {
                        const int r = n_moves + __popcll(mm & ((1ULL << lane) - 1));
                        sval[r] = pv[k];
                        sact[r] = (unsigned short)(pl * 64 + lane);
                    }
                    n_moves += __popcll(mm);
                }
            }
        }
        const float total = wave_sum_butterfly(acc);
        __syncthreads();                                                     // one wave per workgroup: makes the LDS stash visible
        const int first = n_edges;
        int kept = 0;
        const int leaf_edge = path[d];
        for (int base = 0; base < n_moves; base += 64) {
            const int c = base + lane;
            const bool mine = c < n_moves;
            float p = 0.0f;
            if (mine) p = sval[c] / total;                                  // policy /= torch.sum(policy)
            const bool keep = mine && !(p == 0.0f);                         // policy.nonzero() (NaN stays)
            const u64 km = __ballot(keep);
            if (keep) {
                if (v.learning && !v.root_gamma) p = (0.75f * p) + (0.25f * v.noise);   // (1-eps)*probs + eps*noise
                const int slot = first + kept + __popcll(km & ((1ULL << lane) - 1));
                if (slot < v.e_cap) {
                    EdgeStat s; s.W = 0.0; s.N = 0; s.P = p;
                    bp.es[slot] = s;
                    EdgeMeta m; m.first = -1; m.node = -1; m.n = 0; m.action = sact[c]; m.term = 0; m.tval = 0; m.pad = 0;
                    bp.em[slot] = m;
                }
            }
            kept += __popcll(km);
        }
        if (first + kept > v.e_cap) { err = SZ_ERR_CAPACITY; kept = 0; }
        if (v.learning && v.root_gamma && d == 0 && kept > 0) {
            // non-reference option (sz_set_root_noise): AlphaZero's root-only noise.  The K Gamma(alpha,1) draws of this board,
            // normalised over its K children, are one Dirichlet(alpha) sample of dimension K; inner nodes keep their priors.
            const float* g = v.root_gamma + (size_t)b * SZ_MAX_MOVES;
            __threadfence_block();                                      // the children were written by other lanes of this wave
            float gs = 0.f;
            for (int c = lane; c < kept; c += 64) gs = gs + g[c];
            gs = wave_sum_butterfly(gs);
            for (int c = lane; c < kept; c += 64) bp.es[first + c].P = (0.75f * bp.es[first + c].P) + (0.25f * (g[c] / gs));
        }
        if (lane == 0) { bp.em[leaf_edge].first = first; bp.em[leaf_edge].n = (unsigned short)kept; }
        n_edges += kept;
        (void)node;
        const double val = (double)value[row];                          // node.value = value.item()
        wave_backprop(bp.es, path, d, val);
        sims++; n_expand++; sum_depth += d; sum_k += kept;
        status &= ~ST_PENDING;
    }
    STEP_STAMP(1);

    // ---- next simulation(s): mcts.py:49-64 ------------------------------------------------------
    while (sims < v.S && !err) {
        int d = 0, cur = 0;
        path[0] = 0;
        EdgeMeta m = bp.em[0];
        int parentN = bp.es[0].N;
        m.first = uni(m.first); int mn = uni((int)m.n); parentN = uni(parentN);
        while (mn > 0) {                                                // Node.select
            int bi = wave_select_child(bp.es + m.first, mn, parentN, v.c_puct, nullptr);
            cur = m.first + bi;
            d++;
            if (d >= v.p_cap) { err = SZ_ERR_CAPACITY; break; }
            path[d] = cur;
            m = bp.em[cur];
            parentN = uni(bp.es[cur].N);
            m.first = uni(m.first); mn = uni((int)m.n);
        }
        if (err) break;
        STEP_STAMP(2);
        int node = uni(m.node);
        if (node >= 0) {
            // visited leaf without children: a terminal position (mcts.py:104-109)
            double tv = (double)(int)m.tval;
            wave_backprop(bp.es, path, d, tv);
            sims++; n_term++; sum_depth += d;
            continue;
        }
        // first visit: node.game = deepcopy(parent.game); move_piece(action)   (mcts.py:57-59)
        if (n_nodes >= v.n_cap) { err = SZ_ERR_CAPACITY; break; }
        const int parent_node = uni(bp.em[path[d - 1]].node);
        SzPos P = load_pos(bp.npos + parent_node);
        SzPos X = wave_create_position(bp, P, (int)uni((int)m.action), v.chess960, path, d, root_ply, mask);
        STEP_STAMP(3);
        const int nid = n_nodes++;
        const int is_term = szm_term(X.meta);
        if (lane == 0) {
            bp.npos[nid] = X;
            EdgeMeta* em = bp.em + cur;
            em->node = nid; em->term = (signed char)is_term; em->tval = (signed char)(szm_loss(X.meta) ? -1 : 0);
        }
        __threadfence_block();
        if (is_term) {                                                  // get_value_and_terminated -> (0|-1, True)
            double tv = szm_loss(X.meta) ? -1.0 : 0.0;
            wave_backprop(bp.es, path, d, tv);
            sims++; n_term++; sum_depth += d;
            continue;
        }
        // non-terminal leaf: hand it to the network
        __syncthreads();
        for (int i = lane; i < SZ_MASK_WORDS; i += 64) bp.pmask[i] = mask[i];
        for (int j = lane; j <= d; j += 64) bp.gpath[j] = path[j];
        wave_load_history(bp, path, d, root_ply, X, hist);
        __syncthreads();
        wave_encode(hist, X, (char*)planes + (size_t)row * planes_board_bytes(v.planes_dtype), v.planes_dtype, nullptr);
        status |= ST_PENDING;
        if (lane == 0) { bp.ctl->pend_node = nid; bp.ctl->pend_depth = d; }
        STEP_STAMP(4);
        break;
    }
    if (sims >= v.S && !(status & ST_PENDING)) status |= ST_DONE;
    if (err) status |= ST_ERROR;
    if (lane == 0) {
        Ctl* c = bp.ctl;
        c->status = status; c->n_nodes = n_nodes; c->n_edges = n_edges; c->sims_done = sims;
        c->n_expand += n_expand; c->n_term += n_term; c->sum_depth += sum_depth; c->sum_k += sum_k;
        if (n_edges > c->max_edges) c->max_edges = n_edges;
        if (err && !c->err) c->err = err;
    }
}

// ------------------------------------------------------------------------------------------------
// kernel: root readout (mcts.py:113-122)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_root_children(View v, int* action, int* visits, int* n_child, float* prior, double* wsum) {
    const int b = blockIdx.x, lane = lane_id();
    BoardPtrs bp = board_ptrs(v, b);
    int status = bp.ctl->status;
    int n = 0, first = 0;
    if (status & ST_SEARCHING) { EdgeMeta m = bp.em[0]; n = m.n; first = m.first; }
    if (lane == 0) n_child[b] = n;
    for (int c = lane; c < SZ_MAX_CHILDREN; c += 64) {
        size_t o = (size_t)b * SZ_MAX_CHILDREN + c;
        if (c < n) {
            EdgeStat s = bp.es[first + c];
            action[o] = bp.em[first + c].action; visits[o] = s.N;
            if (prior) prior[o] = s.P;
            if (wsum) wsum[o] = s.W;
        } else {
            action[o] = -1; visits[o] = 0;
            if (prior) prior[o] = 0.f;
            if (wsum) wsum[o] = 0.0;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// kernel: sample + play (sim.py:63-76) and game-over test (sim.py:46, 86-97)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_play(View v, const double* uniforms) {
    extern __shared__ u64 lds64[];
    u64* mask = lds64 + LDS_HIST_WORDS; int* path = (int*)(lds64 + LDS_HIST_WORDS + LDS_MASK_WORDS);
    int* vis_lds = path + 8;
    const int b = blockIdx.x, lane = lane_id();
    BoardPtrs bp = board_ptrs(v, b);
    int status = uni(bp.ctl->status);
    if (!(status & ST_SEARCHING) || !(status & ST_DONE) || (status & (ST_ERROR | ST_GAMEOVER))) return;
    EdgeMeta rm = bp.em[0];
    const int n = uni((int)rm.n), first = uni(rm.first);
    const int root_ply = uni(bp.ctl->game_ply);
    int total = 0;
    for (int c = lane; c < SZ_MAX_CHILDREN; c += 64) {
        int nv = 0, act = -1;
        if (c < n) { nv = bp.es[first + c].N; act = bp.em[first + c].action; }
        vis_lds[c] = nv;
        total += nv;
        if (v.rec_action) { v.rec_action[(size_t)b * SZ_MAX_CHILDREN + c] = act; v.rec_visits[(size_t)b * SZ_MAX_CHILDREN + c] = nv; }
    }
    for (int off = 32; off >= 1; off >>= 1) total += __shfl_xor(total, off);
    __syncthreads();
    if (n == 0 || total == 0) {                                          // ZeroDivisionError in the reference (mcts.py:118-120)
        if (lane == 0) { bp.ctl->status = status | ST_ERROR; if (!bp.ctl->err) bp.ctl->err = SZ_ERR_ZERO_VISITS; }
        return;
    }
    // np.random.choice: cdf = cumsum(p); cdf /= cdf[-1]; idx = searchsorted(cdf, u, 'right')   (sequential f64, like numpy)
    int chosen = 0;
    if (lane == 0) {
        const double u = uniforms[b];
        double acc = 0.0;
        for (int c = 0; c < n; c++) acc += (double)vis_lds[c] / (double)total;
        const double last = acc;
        acc = 0.0;
        int idx = 0;
        for (int c = 0; c < n; c++) { acc += (double)vis_lds[c] / (double)total; if (acc / last <= u) idx = c + 1; }
        if (idx >= n) idx = n - 1;
        if (u < 0.0) {                                                  // greedy: max(action_probs, key=...) of eval.py:92-94 — first maximum
            idx = 0;
            for (int c = 1; c < n; c++) if (vis_lds[c] > vis_lds[idx]) idx = c;
        }
        chosen = idx;
    }
    chosen = uni(chosen);
    const int action = uni((int)bp.em[first + chosen].action);
    SzPos R = load_pos(bp.npos);
    path[0] = 0;
    SzPos X = wave_create_position(bp, R, action, v.chess960, path, 1, root_ply, mask);
    const int over = szm_term(X.meta);
    int result = 0;
    if (over && szm_loss(X.meta)) result = szm_turn(X.meta) ? -1 : 1;      // side to move is mated
    if (lane == 0) {
        bp.ring[(root_ply + 1) & (SZ_RING - 1)] = X;
        Ctl* c = bp.ctl;
        c->game_ply = root_ply + 1;
        c->status = (status & ST_ACTIVE) | (over ? ST_GAMEOVER : 0);
        c->game_result = result;
        if (v.rec_nchild) { v.rec_nchild[b] = n; v.rec_chosen[b] = action; v.rec_over[b] = (uint8_t)over; v.rec_result[b] = (int8_t)result; v.rec_active[b] = 1; }
    }
    if (!v.reuse || over) return;
    // ---- NON-REFERENCE option: keep the subtree of the move just played (the reference builds a fresh tree per ply, sim.py:53) ------------
    // Spans are bump-allocated in node-creation order and a node is created after its parent, so walking the nodes in id order visits the kept
    // spans in ascending address order: every span moves DOWN (destination <= source), in place, no second buffer.  nmap[nid] = new index of the
    // edge that owns node nid (-1 = not in the kept subtree); the chosen child becomes edge 0 / node 0.
    __syncthreads();
    const int n_nodes = uni(bp.ctl->n_nodes);
    const EdgeMeta cm = bp.em[first + chosen];
    const int nid_c = uni(cm.node), n_c = uni((int)cm.n);
    if (nid_c < 0 || n_c == 0) return;                               // never visited, or a leaf: nothing to keep (reuse_ready stays 0)
    int* nmap = v.nmap + (size_t)b * v.n_cap;
    for (int i = lane; i < n_nodes; i += 64) nmap[i] = -1;
    const EdgeStat cs = bp.es[first + chosen];
    __threadfence_block();
    if (lane == 0) { bp.es[0] = cs; bp.em[0] = cm; nmap[nid_c] = 0; }
    __threadfence_block();
    int cursor = 1, new_nid = 0;
    for (int nid = nid_c; nid < n_nodes; nid++) {
        const int e_new = uni(nmap[nid]);
        if (e_new < 0) continue;
        EdgeMeta m = bp.em[e_new];                                   // already moved; `first` still points at the old span, which no copy has reached yet
        const int ofirst = uni(m.first), on = uni((int)m.n);
        for (int base = 0; base < on; base += 64) {
            const int k = base + lane;
            EdgeStat s2; EdgeMeta m2;
            if (k < on) { s2 = bp.es[ofirst + k]; m2 = bp.em[ofirst + k]; }
            __threadfence_block();
            if (k < on) {
                bp.es[cursor + k] = s2; bp.em[cursor + k] = m2;
                if (m2.node >= 0) nmap[m2.node] = cursor + k;
            }
            __threadfence_block();
        }
        if (lane == 0) {
            SzPos pz = bp.npos[nid];
            bp.npos[new_nid] = pz;
            bp.em[e_new].first = on > 0 ? cursor : -1;
            bp.em[e_new].node = new_nid;
        }
        __threadfence_block();
        cursor += on;
        new_nid++;
    }
    if (lane == 0) { Ctl* c = bp.ctl; c->n_nodes = new_nid; c->n_edges = cursor; c->reuse_ready = 1; }
}

// ------------------------------------------------------------------------------------------------
// test hook (sz_debug_select): the device's Node.select / get_ucb on caller-supplied children, one wave per case
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_debug_select(const int* offsets, const int* vc, const double* wsum, const float* prior,
                                                     const int* parent_visits, const float* c_puct, float* ucb_out, int* argmax_out) {
    extern __shared__ u64 lds64[];
    EdgeStat* ch = (EdgeStat*)lds64;                       // the case's children staged as the SoA records the search reads
    const int cs = blockIdx.x, lo = offsets[cs], n = offsets[cs + 1] - lo;
    if (n <= 0 || n > SZ_MAX_CHILDREN) { if (lane_id() == 0) argmax_out[cs] = -1; return; }
    for (int c = lane_id(); c < n; c += 64) { EdgeStat s; s.W = wsum[lo + c]; s.N = vc[lo + c]; s.P = prior[lo + c]; ch[c] = s; }
    __syncthreads();
    const int bi = wave_select_child(ch, n, parent_visits[cs], c_puct[cs], ucb_out + lo);
    if (lane_id() == 0) argmax_out[cs] = bi;
}

// ------------------------------------------------------------------------------------------------
// host side: the C ABI
// ------------------------------------------------------------------------------------------------
struct sz_engine {
    sz_config cfg;
    View v;
    size_t lds_bytes;
    std::vector<void*> allocs;
    int* d_scharnagl; uint8_t* d_active;
    int* d_slot; int* d_nlive;
};

#define HIPCHK(x) do { hipError_t _e = (x); if (_e != hipSuccess) { fprintf(stderr, "[sigmazero] HIP error %s at %s:%d\n", hipGetErrorString(_e), __FILE__, __LINE__); return SZ_ERR_HIP; } } while (0)

// every entry point runs with the engine's device current and restores the caller's afterwards (one engine per GPU per process;
// a rank whose torch current device is not the engine's must not see it changed under its feet)
struct DeviceGuard {
    int prev = -1; bool changed = false;
    explicit DeviceGuard(int dev) { if (hipGetDevice(&prev) == hipSuccess && prev != dev) changed = (hipSetDevice(dev) == hipSuccess); }
    ~DeviceGuard() { if (changed) (void)hipSetDevice(prev); }
};
#define ENGINE_GUARD(e) DeviceGuard _guard((e)->cfg.device)

template <typename T> static int dalloc(sz_engine* e, T** p, size_t count) {
    void* q = nullptr;
    hipError_t err = hipMalloc(&q, count * sizeof(T));
    if (err != hipSuccess) { fprintf(stderr, "[sigmazero] hipMalloc(%zu bytes) failed: %s\n", count * sizeof(T), hipGetErrorString(err)); return SZ_ERR_HIP; }
    e->allocs.push_back(q);
    *p = (T*)q;
    return SZ_OK;
}

extern "C" {

const char* sz_error_string(int code) {
    switch (code) {
        case SZ_OK: return "ok";
        case SZ_ERR_INVALID: return "invalid argument or illegal move";
        case SZ_ERR_HIP: return "HIP runtime error";
        case SZ_ERR_CAPACITY: return "node/edge capacity exhausted on a board";
        case SZ_ERR_NO_DEVICE: return "no HIP device: the engine has no CPU fallback";
        case SZ_ERR_STATE: return "call sequence violated";
        case SZ_ERR_ZERO_VISITS: return "root children have no visits (num_searches == 1 divides by zero in the reference)";
        default: return "unknown error";
    }
}

int sz_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int sz_create(const sz_config* cfg, sz_engine** out) {
    if (!cfg || !out || cfg->n_boards <= 0 || cfg->num_searches < 0) return SZ_ERR_INVALID;
    if (sz_device_count() <= 0) return SZ_ERR_NO_DEVICE;
    if (cfg->device < 0 || cfg->device >= sz_device_count()) return SZ_ERR_INVALID;
    DeviceGuard _guard(cfg->device);
    sz_engine* e = new sz_engine();
    e->cfg = *cfg;
    View& v = e->v;
    memset(&v, 0, sizeof v);
    v.B = cfg->n_boards; v.S = cfg->num_searches;
    v.reuse = cfg->reuse_subtree ? 1 : 0;
    v.n_cap = (v.reuse ? 2 : 1) * cfg->num_searches + 2;     // reuse: up to num_searches kept nodes + num_searches new ones
    if (cfg->edges_per_board > 0) {
        v.e_cap = cfg->edges_per_board;
    } else {
        // default: the worst case (every expansion creates the maximum of 218 children) when it fits in half of the free HBM, so that no
        // position can overflow a search (4096 boards x 800 searches: 22.9 GB of the 288 GB); otherwise what fits, at least 64 per search
        const long long worst = (long long)(v.reuse ? 2 : 1) * cfg->num_searches * SZ_MAX_MOVES + 2, floor_cap = (long long)cfg->num_searches * 64 + 256;
        size_t free_b = 0, total_b = 0;
        long long fit = floor_cap;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess)
            fit = (long long)(free_b / 2 / ((size_t)cfg->n_boards * (sizeof(EdgeStat) + sizeof(EdgeMeta))));
        long long cap = worst < fit ? worst : fit;
        if (cap < floor_cap) cap = floor_cap;
        v.e_cap = (int)cap;
    }
    if (v.e_cap < SZ_MAX_CHILDREN + 2) v.e_cap = SZ_MAX_CHILDREN + 2;
    v.p_cap = v.n_cap;
    v.learning = cfg->learning; v.chess960 = cfg->chess960; v.planes_dtype = cfg->planes_dtype;
    v.c_puct = cfg->c_puct; v.noise = cfg->noise_value; v.root_gamma = nullptr; v.dbg = nullptr;
    e->lds_bytes = (LDS_HIST_WORDS + LDS_MASK_WORDS) * 8 + (size_t)(v.p_cap > 256 ? v.p_cap : 256) * 4 + 220 * 4 + 220 * 2;   // + expand stash
    if (e->lds_bytes > 64 * 1024) { delete e; return SZ_ERR_INVALID; }
    const size_t B = v.B;
    int rc = SZ_OK;
    if ((rc = dalloc(e, &v.npos, B * v.n_cap)) || (rc = dalloc(e, &v.ring, B * SZ_RING)) || (rc = dalloc(e, &v.es, B * v.e_cap)) ||
        (rc = dalloc(e, &v.em, B * v.e_cap)) || (rc = dalloc(e, &v.gpath, B * v.p_cap)) || (rc = dalloc(e, &v.pmask, B * PMASK_STRIDE)) ||
        (rc = dalloc(e, &v.ctl, B)) || (rc = dalloc(e, &v.rec_planes, B * SZ_NUM_PLANES * 8)) ||
        (rc = dalloc(e, &v.rec_action, B * SZ_MAX_CHILDREN)) || (rc = dalloc(e, &v.rec_visits, B * SZ_MAX_CHILDREN)) ||
        (rc = dalloc(e, &v.rec_nchild, B)) || (rc = dalloc(e, &v.rec_colour, B)) || (rc = dalloc(e, &v.rec_chosen, B)) ||
        (rc = dalloc(e, &v.rec_over, B)) || (rc = dalloc(e, &v.rec_result, B)) || (rc = dalloc(e, &v.rec_active, B)) ||
        (v.reuse && (rc = dalloc(e, &v.nmap, B * v.n_cap))) ||
        (rc = dalloc(e, &e->d_scharnagl, B)) || (rc = dalloc(e, &e->d_active, B)) || (rc = dalloc(e, &e->d_slot, B)) || (rc = dalloc(e, &e->d_nlive, 1))) {
        sz_destroy(e);
        return rc;
    }
    HIPCHK(hipMemset(v.ctl, 0, B * sizeof(Ctl)));
    HIPCHK(hipMemset(v.ring, 0, B * SZ_RING * sizeof(SzPos)));
    HIPCHK(hipMemset(v.rec_active, 0, B));
    HIPCHK(hipDeviceSynchronize());
    *out = e;
    return SZ_OK;
}

int sz_destroy(sz_engine* e) {
    if (!e) return SZ_OK;
    ENGINE_GUARD(e);
    for (void* p : e->allocs) (void)hipFree(p);
    delete e;
    return SZ_OK;
}

int sz_new_games(sz_engine* e, const int32_t* scharnagl, const uint8_t* active, void* stream) {
    if (!e) return SZ_ERR_INVALID;
    ENGINE_GUARD(e);
    hipStream_t s = (hipStream_t)stream;
    const size_t B = e->v.B;
    std::vector<int> sch(B, -1);
    if (scharnagl) for (size_t i = 0; i < B; i++) sch[i] = scharnagl[i];
    HIPCHK(hipMemcpyAsync(e->d_scharnagl, sch.data(), B * sizeof(int), hipMemcpyHostToDevice, s));
    if (active) HIPCHK(hipMemcpyAsync(e->d_active, active, B, hipMemcpyHostToDevice, s));
    HIPCHK(hipStreamSynchronize(s));                    // host staging buffers go out of scope
    hipLaunchKernelGGL(k_new_games, dim3(e->v.B), dim3(64), e->lds_bytes, s, e->v, e->d_scharnagl, active ? e->d_active : nullptr);
    HIPCHK(hipGetLastError());
    return SZ_OK;
}

int sz_debug_step_stamps(sz_engine* e, void* dev_buffer) {
    if (!e) return SZ_ERR_INVALID;
    ENGINE_GUARD(e);
    e->v.dbg = (unsigned long long*)dev_buffer;
    return SZ_OK;
}

int sz_compact(sz_engine* e, int32_t enable, int32_t* n_live_out, void* stream) {
    if (!e) return SZ_ERR_INVALID;
    ENGINE_GUARD(e);
    if (!enable) { e->v.slot = nullptr; if (n_live_out) *n_live_out = e->v.B; return SZ_OK; }
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_compact, dim3(1), dim3(1024), 0, s, e->v, e->d_slot, e->d_nlive);
    HIPCHK(hipGetLastError());
    int n = 0;
    HIPCHK(hipMemcpyAsync(&n, e->d_nlive, sizeof n, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    e->v.slot = e->d_slot;
    if (n_live_out) *n_live_out = n;
    return SZ_OK;
}

int sz_set_root_noise(sz_engine* e, const float* gamma_dev) {
    if (!e) return SZ_ERR_INVALID;
    ENGINE_GUARD(e);
    // a reused root (sz_config.reuse_subtree) was expanded as an inner node, on un-noised priors, and is never expanded again: root noise would silently
    // apply to the first ply of a game only.  The two non-reference options are therefore mutually exclusive.
    if (gamma_dev && e->v.reuse) return SZ_ERR_STATE;
    e->v.root_gamma = gamma_dev;
    return SZ_OK;
}

int sz_set_active(sz_engine* e, const uint8_t* active, void* stream) {
    if (!e || !active) return SZ_ERR_INVALID;
    ENGINE_GUARD(e);
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(hipMemcpyAsync(e->d_active, active, e->v.B, hipMemcpyHostToDevice, s));
    HIPCHK(hipStreamSynchronize(s));
    hipLaunchKernelGGL(k_set_active, dim3((e->v.B + 255) / 256), dim3(256), 0, s, e->v, e->d_active);
    HIPCHK(hipGetLastError());
    return SZ_OK;
}

int sz_upload_game(sz_engine* e, int32_t board, const void* ring, int32_t ply, void* stream) {
    if (!e || !ring || board < 0 || board >= e->v.B || ply < 0) return SZ_ERR_INVALID;
    ENGINE_GUARD(e);
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(hipMemcpyAsync(e->v.ring + (size_t)board * SZ_RING, ring, SZ_RING * sizeof(SzPos), hipMemcpyHostToDevice, s));
    HIPCHK(hipStreamSynchronize(s));
    hipLaunchKernelGGL(k_after_upload, dim3(1), dim3(1), 0, s, e->v, board, ply);
    HIPCHK(hipGetLastError());
    return SZ_OK;
}

int sz_search_begin(sz_engine* e, void* planes_dev, void* stream) {
    if (!e) return SZ_ERR_INVALID;
    ENGINE_GUARD(e);
    hipLaunchKernelGGL(k_search_begin, dim3(e->v.B), dim3(64), e->lds_bytes, (hipStream_t)stream, e->v, planes_dev);
    HIPCHK(hipGetLastError());
    if (e->v.reuse && planes_dev) {
        // boards that continue on a kept subtree have no root to evaluate: one descent-only launch selects their first leaf (boards whose fresh
        // root waits for the network sit it out), so that every board enters the first network call with a real position
        hipLaunchKernelGGL(k_search_step, dim3(e->v.B), dim3(64), e->lds_bytes, (hipStream_t)stream, e->v, (const float*)nullptr, (const float*)nullptr, planes_dev);
        HIPCHK(hipGetLastError());
    }
    return SZ_OK;
}

int sz_search_step(sz_engine* e, const float* policy_dev, const float* value_dev, void* planes_dev, void* stream) {
    if (!e || !policy_dev || !value_dev || !planes_dev) return SZ_ERR_INVALID;
    ENGINE_GUARD(e);
    hipLaunchKernelGGL(k_search_step, dim3(e->v.B), dim3(64), e->lds_bytes, (hipStream_t)stream, e->v, policy_dev, value_dev, planes_dev);
    HIPCHK(hipGetLastError());
    return SZ_OK;
}

int sz_get_stats(sz_engine* e, sz_stats* out, void* stream) {
    if (!e || !out) return SZ_ERR_INVALID;
    ENGINE_GUARD(e);
    hipStream_t s = (hipStream_t)stream;
    std::vector<Ctl> h(e->v.B);
    HIPCHK(hipMemcpyAsync(h.data(), e->v.ctl, h.size() * sizeof(Ctl), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    memset(out, 0, sizeof *out);
    for (const Ctl& c : h) {
        out->expansions += c.n_expand; out->terminal_hits += c.n_term; out->sum_depth += c.sum_depth; out->sum_children += c.sum_k;
        if ((uint64_t)c.max_edges > out->max_edges_used) out->max_edges_used = c.max_edges;
        if (c.status & ST_PENDING) out->boards_pending++;
        if (c.status & ST_DONE) out->boards_done++;
        if (c.status & ST_ERROR) { out->boards_error++; if (!out->first_error) out->first_error = c.err; }
    }
    out->simulations = out->expansions + out->terminal_hits;
    return SZ_OK;
}

int sz_root_children(sz_engine* e, int32_t* action_dev, int32_t* visits_dev, int32_t* n_child_dev, float* prior_dev, double* value_sum_dev, void* stream) {
    if (!e || !action_dev || !visits_dev || !n_child_dev) return SZ_ERR_INVALID;
    ENGINE_GUARD(e);
    hipLaunchKernelGGL(k_root_children, dim3(e->v.B), dim3(64), 0, (hipStream_t)stream, e->v, action_dev, visits_dev, n_child_dev, prior_dev, value_sum_dev);
    HIPCHK(hipGetLastError());
    return SZ_OK;
}

int sz_play(sz_engine* e, const double* uniforms_dev, void* stream) {
    if (!e || !uniforms_dev) return SZ_ERR_INVALID;
    ENGINE_GUARD(e);
    hipLaunchKernelGGL(k_play, dim3(e->v.B), dim3(64), e->lds_bytes, (hipStream_t)stream, e->v, uniforms_dev);
    HIPCHK(hipGetLastError());
    return SZ_OK;
}

int sz_fetch_ply(sz_engine* e, uint8_t* packed_planes, int32_t* action, int32_t* visits, int32_t* n_child, uint8_t* colour, int32_t* chosen,
                 uint8_t* game_over, int8_t* result, uint8_t* active, void* stream) {
    if (!e) return SZ_ERR_INVALID;
    ENGINE_GUARD(e);
    hipStream_t s = (hipStream_t)stream;
    const size_t B = e->v.B;
    const View& v = e->v;
    if (packed_planes) HIPCHK(hipMemcpyAsync(packed_planes, v.rec_planes, B * SZ_NUM_PLANES * 8, hipMemcpyDeviceToHost, s));
    if (action) HIPCHK(hipMemcpyAsync(action, v.rec_action, B * SZ_MAX_CHILDREN * 4, hipMemcpyDeviceToHost, s));
    if (visits) HIPCHK(hipMemcpyAsync(visits, v.rec_visits, B * SZ_MAX_CHILDREN * 4, hipMemcpyDeviceToHost, s));
    if (n_child) HIPCHK(hipMemcpyAsync(n_child, v.rec_nchild, B * 4, hipMemcpyDeviceToHost, s));
    if (colour) HIPCHK(hipMemcpyAsync(colour, v.rec_colour, B, hipMemcpyDeviceToHost, s));
    if (chosen) HIPCHK(hipMemcpyAsync(chosen, v.rec_chosen, B * 4, hipMemcpyDeviceToHost, s));
    if (game_over) HIPCHK(hipMemcpyAsync(game_over, v.rec_over, B, hipMemcpyDeviceToHost, s));
    if (result) HIPCHK(hipMemcpyAsync(result, v.rec_result, B, hipMemcpyDeviceToHost, s));
    if (active) HIPCHK(hipMemcpyAsync(active, v.rec_active, B, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return SZ_OK;
}

int sz_debug_pending(sz_engine* e, uint64_t* mask, int32_t* depth, int32_t* n_nodes, int32_t* n_edges, int32_t* status, void* stream) {
    if (!e) return SZ_ERR_INVALID;
    ENGINE_GUARD(e);
    hipStream_t s = (hipStream_t)stream;
    const size_t B = e->v.B;
    std::vector<Ctl> h(B);
    std::vector<u64> pm(mask ? B * PMASK_STRIDE : 0);
    HIPCHK(hipMemcpyAsync(h.data(), e->v.ctl, B * sizeof(Ctl), hipMemcpyDeviceToHost, s));
    if (mask) HIPCHK(hipMemcpyAsync(pm.data(), e->v.pmask, pm.size() * 8, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    for (size_t b = 0; b < B; b++) {
        if (depth) depth[b] = h[b].pend_depth;
        if (n_nodes) n_nodes[b] = h[b].n_nodes;
        if (n_edges) n_edges[b] = h[b].n_edges;
        if (status) status[b] = h[b].status;
        if (mask) memcpy(mask + b * SZ_MASK_WORDS, pm.data() + b * PMASK_STRIDE, SZ_MASK_WORDS * 8);
    }
    return SZ_OK;
}

int sz_debug_position(sz_engine* e, int32_t board, void* pos_out, int32_t* ply, void* stream) {
    if (!e || board < 0 || board >= e->v.B || !pos_out) return SZ_ERR_INVALID;
    ENGINE_GUARD(e);
    hipStream_t s = (hipStream_t)stream;
    Ctl c;
    HIPCHK(hipMemcpyAsync(&c, e->v.ctl + board, sizeof c, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipMemcpyAsync(pos_out, e->v.ring + (size_t)board * SZ_RING + (c.game_ply & (SZ_RING - 1)), sizeof(SzPos), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (ply) *ply = c.game_ply;
    return SZ_OK;
}

int sz_debug_tree(sz_engine* e, int32_t board, int32_t max_nodes, int32_t* depth, int32_t* action, int32_t* visits, double* value_sum,
                  float* prior, int32_t* n_out, void* stream) {
    if (!e || board < 0 || board >= e->v.B || !n_out) return SZ_ERR_INVALID;
    ENGINE_GUARD(e);
    hipStream_t s = (hipStream_t)stream;
    Ctl c;
    HIPCHK(hipMemcpyAsync(&c, e->v.ctl + board, sizeof c, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    *n_out = 0;
    if (!(c.status & ST_SEARCHING) || c.n_edges <= 0) return SZ_OK;
    std::vector<EdgeStat> es(c.n_edges);
    std::vector<EdgeMeta> em(c.n_edges);
    HIPCHK(hipMemcpyAsync(es.data(), e->v.es + (size_t)board * e->v.e_cap, es.size() * sizeof(EdgeStat), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(em.data(), e->v.em + (size_t)board * e->v.e_cap, em.size() * sizeof(EdgeMeta), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    // depth-first in child order = the order a recursive walk over Node.children visits the reference's tree; edge 0 is the root itself
    std::vector<std::pair<int, int>> stack;                // (edge, depth)
    int n = 0;
    if (max_nodes > 0 && depth) { depth[0] = -1; action[0] = -1; visits[0] = es[0].N; value_sum[0] = es[0].W; prior[0] = es[0].P; }
    n = 1;
    for (int k = (int)em[0].n - 1; k >= 0 && em[0].first >= 0; k--) stack.push_back({em[0].first + k, 0});
    while (!stack.empty()) {
        auto [ed, d] = stack.back(); stack.pop_back();
        if (n < max_nodes && depth) { depth[n] = d; action[n] = em[ed].action; visits[n] = es[ed].N; value_sum[n] = es[ed].W; prior[n] = es[ed].P; }
        n++;
        if (em[ed].first >= 0) for (int k = (int)em[ed].n - 1; k >= 0; k--) stack.push_back({em[ed].first + k, d + 1});
    }
    *n_out = n;
    return SZ_OK;
}

int sz_debug_select(const int32_t* offsets_dev, const int32_t* vc_dev, const double* value_sum_dev, const float* prior_dev, const int32_t* parent_visits_dev,
                    const float* c_dev, float* ucb_out_dev, int32_t* argmax_out_dev, int32_t n_cases, void* stream) {
    if (!offsets_dev || !vc_dev || !value_sum_dev || !prior_dev || !parent_visits_dev || !c_dev || !ucb_out_dev || !argmax_out_dev || n_cases <= 0) return SZ_ERR_INVALID;
    hipLaunchKernelGGL(k_debug_select, dim3(n_cases), dim3(64), SZ_MAX_CHILDREN * sizeof(EdgeStat), (hipStream_t)stream, offsets_dev, vc_dev, value_sum_dev,
                       prior_dev, parent_visits_dev, c_dev, ucb_out_dev, argmax_out_dev);
    HIPCHK(hipGetLastError());
    return SZ_OK;
}

}  // extern "C"
