/* sigmazero.h — C ABI of the MI355X-native batched MCTS self-play engine.
 *
 * The reference (DidItWork/Sigma-Zero) has NO native/FFI layer: its boundary for this path is the
 * Python call surface.  This header is therefore the boundary a maintainer binds
 * from Python with ctypes (INTEGRATION.md shows the stub); every entry point names the reference
 * code it replaces.  Plain pointers and sizes only — no torch types.
 *
 * Conventions: every function returns SZ_OK (0) or a negative SZ_ERR_*; the engine owns all of its
 * device memory; tensors handed in by pointer stay owned by the caller; all device work is enqueued
 * on the caller's stream (pass torch.cuda.current_stream().cuda_stream), which must belong to the
 * engine's device (sz_config.device); every call makes that device current for its duration and
 * restores the caller's current device; one engine per GPU per process; not thread-safe.
 */
#ifndef SIGMAZERO_H
#define SIGMAZERO_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SZ_RING 256              /* per-game position history kept for repetition / history planes */
#define SZ_PLANES 119            /* chess_tensor.py:31-35  M*T+L = 14*8+7 */
#define SZ_ACTIONS 4672          /* 73*8*8, chess_tensor.py:197 */
#define SZ_MAX_MOVES 218
#define SZ_POS_BYTES 80

enum {
    SZ_OK = 0,
    SZ_ERR_INVALID = -1,         /* bad argument / illegal move (ValueError("Invalid move"), chess_tensor.py:91-92) */
    SZ_ERR_HIP = -2,             /* a HIP runtime call failed */
    SZ_ERR_CAPACITY = -3,        /* a board ran out of child slots (edges_per_board too small) */
    SZ_ERR_NO_DEVICE = -4,       /* no MI355X visible: the product path never falls back to the CPU */
    SZ_ERR_STATE = -5,           /* call sequence violated (e.g. step before begin) */
    SZ_ERR_ZERO_VISITS = -6      /* num_searches == 1: the reference raises ZeroDivisionError (mcts.py:118-120) */
};

/* network-input layouts written by the engine:
 *   F32 / BF16     : [n_boards,119,8,8] NCHW, the reference's get_representation() layout
 *   NHWC128_BF16   : [n_boards,64,128] position-major, channels 119..127 zero — input of sz_nn_conv_bf16 (stem)
 *   NHWC128_BITS   : the same image bit-packed, 1 KiB per board (16x fewer bytes than bf16): [n_boards][64 lanes] uint4,
 *                    lane l = psub*16 + cq; byte q (little-endian) of its 16 bytes holds channels cq*8..cq*8+7 (bit k =
 *                    channel cq*8+k) of position q*4 + psub.  Consumed by the stem with SZ_NN_IN_BITS. */
enum { SZ_PLANES_F32 = 0, SZ_PLANES_BF16 = 1, SZ_PLANES_NHWC128_BF16 = 2, SZ_PLANES_NHWC128_BITS = 3 };

/* ------------------------------------------------------------------ engine (HIP, gfx950) */

typedef struct sz_engine sz_engine;

typedef struct sz_config {
    int32_t n_boards;            /* concurrent self-play boards on this GPU */
    int32_t num_searches;        /* args['num_searches'], mcts.py:49 */
    float   c_puct;              /* args['C'], mctsnode.py:37 */
    int32_t learning;            /* search(..., learning=...), mcts.py:91 */
    float   noise_value;         /* value of the degenerate Dirichlet draw of mcts.py:93 (1-2^-24 under torch 2.10) */
    int32_t chess960;            /* boards spell castling king-takes-rook (chess.Board(chess960=True)) */
    int32_t edges_per_board;     /* child slots per board per search; 0 = worst case num_searches*218+2 when that fits in half of the free
                                  * HBM (no position can overflow), else what fits, at least num_searches*64+256 */
    int32_t planes_dtype;        /* SZ_PLANES_F32 / SZ_PLANES_BF16: element type of the network input */
    int32_t device;              /* HIP device ordinal */
    int32_t reuse_subtree;       /* NON-REFERENCE option, 0 = off (default, the reference builds a fresh tree per ply: sim.py:53).  != 0: sz_play keeps the
                                  * subtree below the move it plays (compacted in place) and the next sz_search_begin continues on it — the new root starts
                                  * with that child's visit count, value sum and children — whenever the kept part is at most num_searches nodes and half
                                  * of the child slots; otherwise, and after sz_new_games / sz_upload_game, the search starts fresh.  Doubles the stores. */
} sz_config;

typedef struct sz_stats {
    uint64_t simulations;        /* executions of the mcts.py:49 loop body, all boards */
    uint64_t expansions;         /* non-terminal leaves evaluated by the network (mcts.py:66-102) */
    uint64_t terminal_hits;      /* simulations that ended in a terminal leaf (mcts.py:104-106) */
    uint64_t sum_depth;          /* sum of leaf depths (edges from the root) */
    uint64_t sum_children;       /* sum of K over expansions */
    uint64_t max_edges_used;     /* high-water mark of child slots on any board */
    int32_t  boards_pending;     /* boards waiting for a network evaluation */
    int32_t  boards_done;        /* boards whose search finished */
    int32_t  boards_error;       /* boards with a sticky error flag */
    int32_t  first_error;        /* SZ_ERR_* of the first failing board, or 0 */
} sz_stats;

/* MCTS0.__init__ (mcts.py:30-37): allocate the SoA node store for n_boards trees. */
int sz_create(const sz_config* cfg, sz_engine** out);
int sz_destroy(sz_engine* e);

/* ChessTensor.__init__/start_board (chess_tensor.py:31-35,65-86) for every board with active[b] != 0
 * (NULL = all): scharnagl[b] >= 0 -> Board.from_chess960_pos(n), < 0 -> chess.Board().
 * Host arrays. */
int sz_new_games(sz_engine* e, const int32_t* scharnagl, const uint8_t* active, void* stream);

/* Put an arbitrary live game on one board: `ring` is the SZ_RING*SZ_POS_BYTES history exported by
 * szh_export (the root game object of mcts.py:43, which search() uses un-copied). */
int sz_upload_game(sz_engine* e, int32_t board, const void* ring, int32_t ply, void* stream);
/* mark boards (in)active for the next searches; host array of n_boards bytes */
int sz_set_active(sz_engine* e, const uint8_t* active, void* stream);

/* Batch compaction for ragged self-play (games of one call end at different plies, sim.py:46): with enable != 0 the boards that will
 * search next (active, game not over, no error) are numbered 0..n_live-1 in board order, and from then on board b reads its policy / value
 * from ROW slot(b) of policy_dev / value_dev and writes its network input to ROW slot(b) of planes_dev — so the network runs on n_live
 * rows instead of n_boards.  Call between searches, after sz_play / sz_new_games / sz_set_active changed the live set (synchronises the
 * stream to return n_live).  enable == 0 restores the identity mapping (row = board), the default.  Results per board do not depend on
 * the mapping.  Per-board outputs (sz_root_children, sz_fetch_ply, sz_play's uniforms) stay indexed by board.  A board that becomes live
 * without a new sz_compact has no row: sz_search_begin flags it SZ_ERR_STATE (sticky, sz_get_stats().first_error). */
int sz_compact(sz_engine* e, int32_t enable, int32_t* n_live_out, void* stream);

/* First half of MCTS0.search (mcts.py:43-75): create the roots (visit_count = 1), test them for
 * termination, and write the network input of every root into planes_dev [n_boards,119,8,8]. */
int sz_search_begin(sz_engine* e, void* planes_dev, void* stream);

/* One lock-step iteration over all boards: consume model(x, inference=True) for the pending leaves
 * (policy_dev [n_boards,4672] f32 probabilities, value_dev [n_boards] f32) = masked renormalise,
 * noise mix, Node.expand, Node.backpropagate (mcts.py:77-109, mctsnode.py:39-63); then run
 * Node.select (mctsnode.py:23-37) down to the next leaf of each board, play the move
 * (mcts.py:57-59), test termination (terminal leaves are backed up on the spot and selection
 * repeats), and encode the next network input into planes_dev. */
int sz_search_step(sz_engine* e, const float* policy_dev, const float* value_dev, void* planes_dev, void* stream);

/* Counters (synchronises the stream). */
int sz_get_stats(sz_engine* e, sz_stats* out, void* stream);

/* mcts.py:113-122 readout: for each board the root children in ascending action-index order.
 * Device pointers: action [n_boards,218] int32, visits [n_boards,218] int32, n_child [n_boards] int32;
 * optional (may be NULL) prior [n_boards,218] f32 and value_sum [n_boards,218] f64. */
int sz_root_children(sz_engine* e, int32_t* action_dev, int32_t* visits_dev, int32_t* n_child_dev,
                     float* prior_dev, double* value_sum_dev, void* stream);

/* sim.py:68-76: sample a move per board from the visit distribution exactly like
 * np.random.choice(keys, p=visits/sum) given the uniform it draws (uniforms_dev [n_boards] f64),
 * record the training sample of this ply, play the move into the board's game and test
 * board.is_game_over() (sim.py:46).  A negative uniform selects the most visited child instead
 * (max(action_probs, key=...) of eval.py:92-94 / play.py:40-41: first maximum in action-index order). */
int sz_play(sz_engine* e, const double* uniforms_dev, void* stream);

/* Copy this ply's training records to host arrays (synchronises; any pointer may be NULL):
 *  packed_planes [n_boards,119,8] uint8   root get_representation(), bit j of byte = column j
 *                                         (the (119,8) format of generate_training_supervised.py:91)
 *  action/visits [n_boards,218] int32, n_child [n_boards] int32        sim.py:72 'actions'
 *  colour [n_boards] uint8 (board.turn at the root, sim.py:73), chosen [n_boards] int32 (action index),
 *  game_over [n_boards] uint8, result [n_boards] int8 (+1 = '1-0', -1 = '0-1', 0 draw; sim.py:86-92),
 *  active [n_boards] uint8 (whether the record is valid for this ply). */
int sz_fetch_ply(sz_engine* e, uint8_t* packed_planes, int32_t* action, int32_t* visits, int32_t* n_child,
                 uint8_t* colour, int32_t* chosen, uint8_t* game_over, int8_t* result, uint8_t* active, void* stream);

/* NON-REFERENCE option, off by default: true AlphaZero root noise.  gamma_dev: device array [n_boards][SZ_MAX_MOVES]
 * f32 of Gamma(alpha,1) draws, read when a search's ROOT is expanded (the step after sz_search_begin): root prior k becomes
 * 0.75*p_k + 0.25*g_k/sum_{j<K} g_j (one Dirichlet(alpha) sample over the K legal moves); inner nodes get no noise.  NULL restores
 * the reference behaviour (mcts.py:91-98: the constant noise_value at every expansion).  Needs learning = 1.  Refused (SZ_ERR_STATE) on an engine created
 * with reuse_subtree: a reused root is never expanded again, the noise would silently reach the first ply of a game only. */
int sz_set_root_noise(sz_engine* e, const float* gamma_dev);

/* diagnostic only: with a device buffer of n_boards*8 uint64, sz_search_step records s_memtime at its phase boundaries per board
 * (0 start, 1 after expand+backprop, 2 after select, 3 after move/movegen/repetition/terminal, 4 after encode); NULL = off (default) */
int sz_debug_step_stamps(sz_engine* e, void* dev_buffer);

/* test / debug readback of the pending leaves: legal-move mask [n_boards,73] uint64 (bit v of word p =
 * action p*64+v), leaf depth, node count, edge count, status per board (host pointers, may be NULL). */
int sz_debug_pending(sz_engine* e, uint64_t* mask, int32_t* depth, int32_t* n_nodes, int32_t* n_edges,
                     int32_t* status, void* stream);
/* test / inspection: the WHOLE tree of one board after a search, depth-first in child order (= a recursive walk over Node.children
 * of mctsnode.py:7-18).  Row 0 is the root itself (depth -1, action -1); rows 1.. are its descendants: depth (0 = root child),
 * action index (action_taken), visit_count, value_sum (f64), prior (f32).  Host arrays of max_nodes entries (may be NULL to count);
 * *n_out = number of nodes in the tree. */
int sz_debug_tree(sz_engine* e, int32_t board, int32_t max_nodes, int32_t* depth, int32_t* action, int32_t* visits, double* value_sum,
                  float* prior, int32_t* n_out, void* stream);
/* test: the DEVICE code of Node.select / Node.get_ucb (mctsnode.py:23-37) on caller-supplied children, one wavefront per case.
 * Case c owns children offsets[c]..offsets[c+1] (<= SZ_MAX_MOVES each) of vc (visit_count), value_sum (f64), prior (f32); parent_visits[c],
 * c_puct[c].  Writes every child's UCB value and the selected child per case.  All device pointers. */
int sz_debug_select(const int32_t* offsets_dev, const int32_t* vc_dev, const double* value_sum_dev, const float* prior_dev,
                    const int32_t* parent_visits_dev, const float* c_dev, float* ucb_out_dev, int32_t* argmax_out_dev, int32_t n_cases, void* stream);
/* copy one board's current game position record (SZ_POS_BYTES) to the host */
int sz_debug_position(sz_engine* e, int32_t board, void* pos_out, int32_t* ply, void* stream);

/* ------------------------------------------------------------------ network tower (MFMA, gfx950) */

/* One fused layer of policyNN's tower (network.py:36-83 BasicBlock halves, :105 stem, :141 conv_p1):
 * conv (3x3 pad 1 or 1x1) with BatchNorm folded + bias (+ residual) (+ ReLU); NHWC bf16 in/out, f32 accumulate.
 * in [n_boards,64,cin] (cin 128 or 256), out/residual [n_boards,64,256], bias [256] f32, w_packed from sz_nn_pack_weights. */
int sz_nn_conv_bf16(const void* in, const void* w_packed, const float* bias, const void* residual, void* out,
                    int32_t n_boards, int32_t cin, int32_t ksize, int32_t relu, void* stream);
/* flag for `relu`/`flags` of the two entry points below: the weights were packed by sz_nn_pack_weights16 and the
 * 16x16x32 MFMA kernels are used (same results; the chip holds a higher clock on that shape) */
#define SZ_NN_W16 0x40000
/* flag for the stem (cin 128, ksize 3, SZ_NN_W16) and for sz_nn_tower_bf16: `in` is the engine's SZ_PLANES_NHWC128_BITS image */
#define SZ_NN_IN_BITS 0x1000000
/* One whole BasicBlock (network.py:36-83) in one launch: out = relu(conv3x3(relu(conv3x3(in,w1)+b1),w2)+b2+in); the
 * intermediate activation stays in LDS.  in/out [n_boards,64,256] bf16 NHWC, out != in. */
int sz_nn_block_bf16(const void* in, const void* w1_packed, const float* bias1, const void* w2_packed, const float* bias2, void* out,
                     int32_t n_boards, int32_t flags, void* stream);
/* host: torch conv weight [256,cin_real,k,k] f32 -> MFMA fragment order [k*k][cin_padded/16][8][64][8] bf16 */
int sz_nn_pack_weights(const float* w_in, int32_t cin_real, int32_t cin_padded, int32_t ksize, uint16_t* out);

int sz_nn_pack_weights16(const float* w_in, int32_t cin_real, int32_t cin_padded, int32_t ksize, uint16_t* out);
/* f16 operands instead of bf16 for sz_nn_tower_bf16 (in `flags`) and sz_nn_heads_bf16 (or-ed into `do_softmax`): v_mfma_f32_16x16x32_f16 runs at the bf16
 * rate and f16 keeps 11 bits of mantissa instead of 8; activations and BatchNorm-folded weights of this network are O(1), far inside f16's range.  The
 * weights are then packed with the _f16 packers, the tower output / heads input [n_boards,64,256] holds f16 bit patterns. */
#define SZ_NN_F16 0x8000000
int sz_nn_pack_weights16_f16(const float* w_in, int32_t cin_real, int32_t cin_padded, int32_t ksize, uint16_t* out);
int sz_nn_pack_head16_f16(const float* w_in, uint16_t* out);

/* The whole tower (network.py:176-184: stem conv + n_blocks BasicBlocks) in ONE persistent launch: a workgroup keeps its
 * boards' activations in LDS through all 1 + 2*n_blocks convolutions; only weights stream.  planes [n_boards,64,128] bf16
 * (SZ_PLANES_NHWC128_BF16; or _BITS with SZ_NN_IN_BITS), out [n_boards,64,256] bf16; w_packed / bias: HOST arrays of 1 + 2*n_blocks DEVICE pointers
 * (sz_nn_pack_weights16 order; [0] = stem packed with cin_padded = 128).
 * The workgroups that share an XCD start their tile rounds together (bounded wait on one arrival counter per XCD, allocated on first use), so a layer's
 * weights cross the fabric once per XCD and round; environment SZ_NN_PACE=0 switches that off.  Results do not depend on it. */
/* Two boards per workgroup when n_boards exceeds the number of CUs, else one (half the latency of a forward at small batches: a search of one position, the
 * tail of a self-play run); a board's result does not depend on the form.  SZ_NN_TOWER_WGB1 / _WGB2 force one (tests). */
#define SZ_NN_TOWER_WGB1 0x10000000
#define SZ_NN_TOWER_WGB2 0x20000000
int sz_nn_tower_bf16(const void* planes, const void* const* w_packed, const float* const* bias, int32_t n_blocks, void* out,
                     int32_t n_boards, int32_t flags /* SZ_NN_IN_BITS | SZ_NN_F16 | SZ_NN_TOWER_WGB* */, void* stream);
/* The same tower at the REFERENCE's precision class (network.py:176-184 is fp32 end to end) on the matrix cores: every operand carried as two bf16
 * numbers (hi = bf16(x), lo = bf16(x - hi): 16 bits of mantissa), every product as three MFMAs (hi*hi + lo*hi + hi*lo) with f32 accumulation, bias /
 * residual / ReLU in f32.  w_stream: ONE device buffer with the weights of the whole tower in k-step order, built on the host with
 * sz_nn_pack_split_stream (sz_nn_split_stream_elems(n_blocks) bf16 elements); bias: device [1 + 2*n_blocks][256] f32 (BatchNorm folded);
 * out [n_boards,64,256] F32 NHWC (the heads then run in fp32).  Two boards per workgroup when n_boards exceeds the number of CUs, else one; a
 * board's result does not depend on which (SZ_NN_SPLIT_WGB1 / _WGB2 force one form: tests).  About 100x closer to the fp32 network than
 * sz_nn_tower_bf16 at about 2.5x its time. */
#define SZ_NN_SPLIT_WGB1 0x2000000
#define SZ_NN_SPLIT_WGB2 0x4000000
int sz_nn_tower_split(const void* planes, const void* w_stream, const float* bias, int32_t n_blocks, float* out,
                      int32_t n_boards, int32_t flags /* SZ_NN_IN_BITS | SZ_NN_SPLIT_WGB* */, void* stream);
/* host: number of bf16 elements of the weight stream; one convolution (conv 0 = stem, cin_real 119; conv c >= 1: the c-th 256-channel 3x3
 * convolution in forward order) from the torch weight [256,cin_real,3,3] f32 into its place in the stream */
int64_t sz_nn_split_stream_elems(int32_t n_blocks);
int sz_nn_pack_split_stream(const float* w_in, int32_t cin_real, int32_t ksize /* 3; 1 for conv_p1 */, int32_t conv, uint16_t* stream);
/* The WHOLE network (network.py:176-192) at that precision class in two launches: the tower above with both heads fused onto each tile while it is still in
 * LDS — conv_p1 (packed into the stream as convolution 1 + 2*n_blocks with ksize 1, its folded bias as the last row of `bias`) -> ReLU -> conv_p2
 * (w_p2_packed from sz_nn_pack_split_head, 8*2*5*64*8 bf16 elements; b_p2 [73]) -> softmax over the 4672 logits, all on hi + lo operands; conv_v1 (wv [256],
 * bv: v_norm folded) in f32 — then the value MLP (sz_nn_value_mlp).  probs [n_boards,4672] f32 in the reference's flatten order (logits if !do_softmax),
 * value [n_boards], v1_scratch [n_boards*64] f32, tower_out: NULL or [n_boards,64,256] f32.  Every reduction runs in an order that does not depend on the
 * batch size or on the neighbours of a board: a position's policy and value are the same bit for bit at batch 1 and at batch 4096. */
int sz_nn_forward_split(const void* planes, const void* w_stream, const float* bias, int32_t n_blocks, const void* w_p2_packed, const float* b_p2,
                        const float* wv, float bv, const float* fc1_w_t, const float* fc1_b, const float* fc2_w, float fc2_b, float* probs, float* value,
                        float* v1_scratch, float* tower_out, int32_t n_boards, int32_t do_softmax, int32_t flags, void* stream);
int sz_nn_pack_split_head(const float* w_in /* conv_p2.weight [73,256] */, uint16_t* out);
/* SZ_NN_F16 in `flags` of sz_nn_tower_split / sz_nn_forward_split: hi + lo F16 operands (22 bits of mantissa: fp32's own class, logits 5e-7 from fp64) instead of hi + lo
 * bf16 (16 bits).  The stream and the head weights then come from the _f16 packers, which multiply the weights by 2^10 (exact; keeps the lo parts of weights down to 1e-3
 * normal f16 numbers); the caller passes the per-convolution biases (incl. conv_p1's) multiplied by 2^10 as well; the kernels scale every accumulator back. */
int sz_nn_pack_split_stream_f16(const float* w_in, int32_t cin_real, int32_t ksize, int32_t conv, uint16_t* stream);
int sz_nn_pack_split_head_f16(const float* w_in, uint16_t* out);
/* Training-step convolutions (train_RL.py:103-122 runs network.py:28,30's 3x3 convolutions forward and backward in fp32) at the reference's precision class on the
 * matrix cores: y = conv3x3(x, w), padding 1, no bias, 256 -> 256 channels; x, y device [n_boards,256,8,8] f32 NCHW.  w_stream (72*2048*16 bytes, device) comes from
 * sz_nn_pack_conv_split_dev(w [256,256,3,3] f32 device, transposed): transposed = 0 for the forward convolution, 1 for backward-data (the same kernel applied to the
 * output gradient).  zero256: device [256] f32 zeros.  f16 = 1 (both calls alike): hi + lo f16 operands (22 bits of mantissa: fp32's class) with exact power-of-two
 * scaling of the weights and of every board; f16 = 0: hi + lo bf16 (16 bits). */
int sz_nn_conv3x3_split_f32(const float* x, const void* w_stream, const float* zero256, float* y, int32_t n_boards, int32_t f16, void* amax_bits, void* stream);
/* weight gradient of the same convolution on hi + lo f16 operands: dw[co][ci][tap] = sum_b,pos gy[b][co][pos] * x[b][ci][pos + off(tap)].  amax_gy / amax_x: device uint32
 * with the f32 bit pattern of max|gy| / max|x| — the optional `amax_bits` output (f16 = 1; atomicMax into a zeroed word) of sz_nn_conv3x3_split_f32 run on those tensors;
 * part: device scratch, 16*9*256*256 f32; dw: device [256,256,3,3] f32, overwritten. */
int sz_nn_wgrad3x3_split_f32(const float* gy, const float* x, const void* amax_gy, const void* amax_x, float* part, float* dw, int32_t n_boards, void* stream);
int sz_nn_pack_conv_split_dev(const float* w, int32_t transposed, int32_t f16, void* w_stream, void* zero_u32 /* optional: device uint32 set to 0 = the amax_bits slot of the convolution that follows */, void* stream);
/* both streams of one convolution in one launch: w_stream (forward) and w_stream_t (backward-data), 72*2048*16 bytes each; zero_u32: optional device uint32[2], both set to 0 */
int sz_nn_pack_conv_split_both(const float* w, int32_t f16, void* w_stream, void* w_stream_t, void* zero_u32, void* stream);
/* One call per direction of a training convolution.  forward: pack (both streams when w_stream_t != NULL) + y = conv3x3(x, w); amax2: optional device uint32[2],
 * zeroed, slot 0 receives max|x|.  backward: gx = backward-data convolution of gy on w_stream_t (NULL: skipped; slot 1 of amax2 receives max|gy|), then dw = weight gradient
 * (NULL: skipped; f16 operands only; part: scratch as in sz_nn_wgrad3x3_split_f32). */
int sz_nn_conv3x3_train_fwd(const float* x, const float* w, int32_t f16, void* w_stream, void* w_stream_t, const float* zero256, float* y, int32_t n_boards, void* amax2, void* stream);
int sz_nn_conv3x3_train_bwd(const float* gy, const float* x, const void* w_stream_t, const float* zero256, float* gx, void* amax2, float* part, float* dw, int32_t n_boards,
                            int32_t f16, void* stream);
/* Train-mode BatchNorm2d + optional skip connection + ReLU in one launch per direction (network.py:62-83 under train_RL.py:103-122; csrc/sz_train.hip):
 * y = relu(bn(x) [+ residual]) with batch statistics over (boards, 8, 8) per channel, running statistics updated in place with `momentum` (NULL: not tracked),
 * save_mean / save_invstd [channels] kept for backward.  x, y, residual, gy, dx, dres: device [n_boards, channels, 8, 8] f32; dres NULL without a residual. */
int sz_bn_act_train_fwd(const float* x, const float* gamma, const float* beta, float* running_mean, float* running_var, float momentum, float eps, const float* residual,
                        float* y, float* save_mean, float* save_invstd, int32_t n_boards, int32_t channels, void* stream);
int sz_bn_act_train_bwd(const float* gy, const float* x, const float* y, const float* gamma, const float* save_mean, const float* save_invstd, float* dx, float* dres,
                        float* dgamma, float* dbeta, int32_t n_boards, int32_t channels, void* stream);
/* value MLP alone (network.py:162-172): v1 [n_boards,64] f32 = relu(bn(conv_v1(x))) -> fc_v1 -> ReLU -> fc_v2 -> tanh -> value [n_boards] */
int sz_nn_value_mlp(const float* v1, const float* fc1_w_t, const float* fc1_b, const float* fc2_w, float fc2_b, float* value, int32_t n_boards, void* stream);
/* diagnostic only: device buffer of 256*4*16 uint64; sz_nn_tower_split then launches its stamped build (tools/split_stamps.py); NULL = shipped kernel */
int sz_nn_debug_split_stamps(void* dev_buffer, int32_t mode /* 1 = stamps; 2/3/4 = stamps + no weight loads / no LDS reads / neither (timing only) */);
/* diagnostic only: when set to a device buffer of 256*4*16 uint64, sz_nn_tower_bf16 launches its stamped build, which records
 * s_memtime at the phase boundaries of one block (tools/tower_stamps.py); NULL switches back to the shipped kernel */
int sz_nn_debug_tower_stamps(void* dev_buffer, int32_t mode /* 1 = stamps; 2/3/4 = stamps + no weight loads / no LDS reads / neither (timing only) */);
/* counter calibration only (tools/fetch_calib.py): reads `bytes` of src exactly once, 16 B per lane, with flat global loads (mode 0) or buffer loads (mode 1) */
int sz_debug_stream_read(const void* src, uint64_t bytes, int32_t mode, void* sink16, void* stream);
/* Heads of policyNN (network.py:141-174) as two small kernels:
 *  policy: t = relu(bn(conv_p1(x))) [n_boards,64,256] bf16 -> conv_p2 + bias -> (softmax) -> probs [n_boards,4672] f32 in the
 *          reference's flatten order (plane*64 + row*8 + col); w_packed from sz_nn_pack_head16(conv_p2.weight [73,256]);
 *  value : x [n_boards,64,256] bf16 -> conv_v1 (+v_norm folded: wv[256], bv) -> ReLU -> fc_v1 -> ReLU -> fc_v2 -> tanh -> [n_boards]. */
int sz_nn_policy_head_bf16(const void* t, const void* w_packed, const float* bias, float* probs, int32_t n_boards, int32_t do_softmax, void* stream);
int sz_nn_value_head_bf16(const void* x, const float* wv, float bv, const float* fc1_w_t, const float* fc1_b, const float* fc2_w, float fc2_b,
                          float* value, int32_t n_boards, void* stream);
int sz_nn_pack_head16(const float* w_in, uint16_t* out);
/* Both heads from ONE read of the tower output x (replaces sz_nn_conv_bf16(ksize 1) + the two calls above): conv_p1 -> LDS -> conv_p2 ->
 * softmax -> probs, and conv_v1 -> value MLP -> value.  w_p1_packed: conv_p1 with p_norm1 folded, sz_nn_pack_weights16(cin_padded 256,
 * ksize 1); v1_scratch: [n_boards*64] f32 device scratch (conv_v1 outputs between the two internal launches). */
int sz_nn_heads_bf16(const void* x, const void* w_p1_packed, const float* b_p1, const void* w_p2_packed, const float* b_p2, const float* wv, float bv,
                     const float* fc1_w_t, const float* fc1_b, const float* fc2_w, float fc2_b, float* probs, float* value, float* v1_scratch,
                     int32_t n_boards, int32_t do_softmax, void* stream);

const char* sz_error_string(int code);
int sz_device_count(void);

/* ------------------------------------------------------------------ host mirror (no GPU) */

/* One game = the reference's ChessTensor object (chess_tensor.py:30-188), rules from the same
 * __host__ __device__ code the kernels run.  Piece codes follow python-chess: N=2 B=3 R=4 Q=5. */
typedef struct szh_game szh_game;

szh_game* szh_game_new(int chess960, int scharnagl);                 /* ChessTensor(chess960) ; :69 / :71 */
szh_game* szh_game_from_fen(const char* fen, int chess960);
szh_game* szh_game_copy(const szh_game* g);                          /* copy.deepcopy(game), mcts.py:58 */
void      szh_game_free(szh_game* g);
int  szh_legal_actions(const szh_game* g, int32_t* idx);             /* actionsToTensor(get_valid_moves()) support, ascending */
int  szh_action_to_move(const szh_game* g, int idx, int32_t* from, int32_t* to, int32_t* promo);  /* tensorToAction, :309-410 */
int  szh_move_to_action(int from, int to, int promo, int white);    /* actionToTensor, :221-306 */
int  szh_push_action(szh_game* g, int idx);                          /* move_piece, :88-129 */
int  szh_push_move(szh_game* g, int from, int to, int promo);
void szh_status(const szh_game* g, int32_t* out12);
void szh_planes(const szh_game* g, uint8_t* out /* 119*64 */);        /* get_representation, :131-142 */
uint64_t szh_perft(szh_game* g, int depth);
void szh_bitboards(const szh_game* g, uint64_t* out10);
int  szh_export(const szh_game* g, void* ring_out, int32_t* ply, int32_t* chess960);
int  szh_is_chess960(const szh_game* g);
int  szh_plane_bits_mismatches(const szh_game* g);                   /* test hook: kernels' bit-parallel mask extraction vs its definition; 0 = identical */

#ifdef __cplusplus
}
#endif
#endif
