// sz_host.cpp — host-side mirror of the reference's per-game objects (C ABI, no GPU needed).
//
// The reference keeps one `ChessTensor` (chess_tensor.py:30-188) per game on the CPU; callers
// (sim.py:36-76, play.py, eval.py) push moves into it and hand it to MCTS0.  This file provides
// that object for the drop-in Python surface: same rules code as the device kernels (sz_chess.h is
// __host__ __device__), 64 "lanes" emulated by a loop.  The search itself never runs here: a game is
// exported with szh_export() and uploaded to the HIP engine (sz_engine.hip).
#include "sz_chess.h"
#include "../../include/sigmazero.h"
#include <cstdlib>
#include <cstring>
#include <cctype>

struct szh_game {
    SzPos ring[SZ_RING];     // position after t plies lives at ring[t & (SZ_RING-1)]
    int ply;                 // plies played since the object was created
    int chess960;
};

namespace {

struct Gen {
    u64 targets[64];         // per real square
    u64 mask[SZ_MASK_WORDS]; // legal-move mask in action-index order
    int n_legal;
    int ep_legal;
    u64 checkers;
};

void generate(const SzPos& p, int chess960, Gen& g) {
    SzInfo I = sz_info(p);
    u64 danger = 0;
    for (u64 n = I.need; n; n &= n - 1) { int s = sz_lsb(n); if (sz_danger_at(p, I, s)) danger |= sz_bit(s); }
    int ep = szm_ep(p.meta);
    g.ep_legal = 0;
    for (int s = 0; s < 64; s++) {
        u64 t = (s == I.ksq && (p.pc[SZ_K] & I.us & sz_bit(s))) ? sz_king_targets(p, I, danger, chess960) : sz_piece_targets(p, I, s);
        g.targets[s] = t;
        if (ep >= 0 && (p.pc[SZ_P] & sz_bit(s)) && (t & sz_bit(ep))) g.ep_legal = 1;
    }
    int flip = sz_view_flip(I.white);
    g.n_legal = 0;
    for (int pl = 0; pl < SZ_MASK_WORDS; pl++) {
        u64 w = 0;
        for (int v = 0; v < 64; v++) {
            int s = v ^ flip;
            if (g.targets[s] && sz_lane_plane_bit(g.targets[s], (p.pc[SZ_P] >> s) & 1, v, pl, I.white)) w |= sz_bit(v);
        }
        g.mask[pl] = w;
        g.n_legal += sz_pop(w);
    }
    g.checkers = I.checkers;
}

// earlier occurrences of g->ring[ply] inside the reversible window (Board.is_repetition walk-back)
int count_reps(const szh_game* g, int ply, u64 key, int irrev_self) {
    if (irrev_self) return 0;
    int reps = 0;
    for (int j = 1; j <= ply && j < SZ_RING; j++) {
        const SzPos& a = g->ring[(ply - j) & (SZ_RING - 1)];
        if (a.key == key) reps++;
        if (szm_irrev(a.meta)) break;
    }
    return reps > 4 ? 4 : reps;
}

void finish(szh_game* g, int ply) {
    SzPos& p = g->ring[ply & (SZ_RING - 1)];
    Gen gen;
    generate(p, g->chess960, gen);
    p.key = sz_hash_key(p, gen.ep_legal);
    int reps = count_reps(g, ply, p.key, szm_irrev(p.meta));
    p.meta = sz_finish_meta(p, gen.checkers, gen.n_legal, gen.ep_legal, reps);
}

u64 perft_rec(szh_game* g, int depth) {
    const SzPos p = g->ring[g->ply & (SZ_RING - 1)];
    Gen gen;
    generate(p, g->chess960, gen);
    if (depth <= 1) return depth == 1 ? (u64)gen.n_legal : 1;
    u64 total = 0;
    for (int pl = 0; pl < SZ_MASK_WORDS; pl++)
        for (u64 w = gen.mask[pl]; w; w &= w - 1) {
            int from, to, promo;
            sz_action_decode(p, pl * 64 + sz_lsb(w), from, to, promo);
            g->ply++;
            g->ring[g->ply & (SZ_RING - 1)] = sz_make_move(p, from, to, promo, g->chess960);
            total += perft_rec(g, depth - 1);
            g->ply--;
        }
    return total;
}

// Board.clean_castling_rights() for a position without history
u64 clean_castling(const SzPos& p, int chess960) {
    u64 all = sz_all(p), blk = all & ~p.white;
    u64 c = p.castling & p.pc[SZ_R];
    u64 wc = c & SZ_RANK1 & p.white, bc = c & SZ_RANK8 & blk;
    u64 wk = p.white & p.pc[SZ_K] & SZ_RANK1, bk = blk & p.pc[SZ_K] & SZ_RANK8;
    if (!chess960) {
        wc &= sz_bit(0) | sz_bit(7); bc &= sz_bit(56) | sz_bit(63);
        if (!(wk & sz_bit(4))) wc = 0;
        if (!(bk & sz_bit(60))) bc = 0;
        return wc | bc;
    }
    if (!wk) wc = 0;
    if (!bk) bc = 0;
    u64 wa = wc & (~wc + 1), ba = bc & (~bc + 1);
    u64 wh = wc ? sz_bit(sz_msb(wc)) : 0, bh = bc ? sz_bit(sz_msb(bc)) : 0;
    if (wa && sz_msb(wa) > sz_msb(wk)) wa = 0;
    if (wh && sz_msb(wh) < sz_msb(wk)) wh = 0;
    if (ba && sz_msb(ba) > sz_msb(bk)) ba = 0;
    if (bh && sz_msb(bh) < sz_msb(bk)) bh = 0;
    return wa | wh | ba | bh;
}

}  // namespace

extern "C" {

szh_game* szh_game_new(int chess960, int scharnagl) {
    szh_game* g = (szh_game*)calloc(1, sizeof(szh_game));
    g->chess960 = chess960 ? 1 : 0;
    g->ply = 0;
    g->ring[0] = sz_startpos(chess960 ? scharnagl : -1);
    finish(g, 0);
    return g;
}

szh_game* szh_game_from_fen(const char* fen, int chess960) {
    szh_game* g = (szh_game*)calloc(1, sizeof(szh_game));
    g->chess960 = chess960 ? 1 : 0;
    SzPos p; memset(&p, 0, sizeof p);
    int r = 7, f = 0;
    const char* c = fen;
    for (; *c && *c != ' '; c++) {
        if (*c == '/') { r--; f = 0; }
        else if (isdigit((unsigned char)*c)) f += *c - '0';
        else {
            int k = -1;
            switch (tolower((unsigned char)*c)) { case 'p': k = SZ_P; break; case 'n': k = SZ_N; break; case 'b': k = SZ_B; break;
                case 'r': k = SZ_R; break; case 'q': k = SZ_Q; break; case 'k': k = SZ_K; break; }
            if (k >= 0 && f < 8 && r >= 0) { p.pc[k] |= sz_bit(r * 8 + f); if (isupper((unsigned char)*c)) p.white |= sz_bit(r * 8 + f); }
            f++;
        }
    }
    while (*c == ' ') c++;
    int white = (*c != 'b');
    while (*c && *c != ' ') c++;
    while (*c == ' ') c++;
    u64 blk = sz_all(p) & ~p.white;
    for (; *c && *c != ' '; c++) {
        if (*c == '-') continue;
        int w = isupper((unsigned char)*c) != 0;
        char flag = (char)tolower((unsigned char)*c);
        u64 back = w ? SZ_RANK1 : SZ_RANK8, side = w ? p.white : blk;
        u64 rooks = side & p.pc[SZ_R] & back, king = side & p.pc[SZ_K] & back;
        if (flag == 'q') p.castling |= (king && rooks && sz_lsb(rooks) < sz_lsb(king)) ? (rooks & (~rooks + 1)) : (SZ_FILEA & back);
        else if (flag == 'k') p.castling |= (king && rooks && sz_msb(king) < sz_msb(rooks)) ? sz_bit(sz_msb(rooks)) : (SZ_FILEH & back);
        else if (flag >= 'a' && flag <= 'h') p.castling |= (SZ_FILEA << (flag - 'a')) & back;
    }
    while (*c == ' ') c++;
    int ep = -1;
    if (*c && *c != '-') ep = (c[0] - 'a') + 8 * (c[1] - '1');
    while (*c && *c != ' ') c++;
    while (*c == ' ') c++;
    int half = 0;
    if (*c) { half = atoi(c); }
    if (half > 255) half = 255;
    p.castling = clean_castling(p, g->chess960);
    p.meta = ((u64)(ep + 1) << SZM_EP_SHIFT) | ((u64)white << SZM_TURN_BIT) | ((u64)half << SZM_HALF_SHIFT) | ((u64)1 << SZM_IRREV_BIT);
    g->ring[0] = p;
    g->ply = 0;
    finish(g, 0);
    return g;
}

szh_game* szh_game_copy(const szh_game* s) { szh_game* g = (szh_game*)malloc(sizeof(szh_game)); memcpy(g, s, sizeof(szh_game)); return g; }
void szh_game_free(szh_game* g) { free(g); }

int szh_legal_actions(const szh_game* g, int32_t* idx) {
    Gen gen;
    const SzPos& p = g->ring[g->ply & (SZ_RING - 1)];
    generate(p, g->chess960, gen);
    int n = 0;
    for (int pl = 0; pl < SZ_MASK_WORDS; pl++)
        for (u64 w = gen.mask[pl]; w; w &= w - 1) idx[n++] = pl * 64 + sz_lsb(w);
    return n;
}

int szh_action_to_move(const szh_game* g, int idx, int32_t* from, int32_t* to, int32_t* promo) {
    int f, t, pr;
    if (idx < 0 || idx >= SZ_NUM_ACTIONS || !sz_action_decode(g->ring[g->ply & (SZ_RING - 1)], idx, f, t, pr)) return SZ_ERR_INVALID;
    *from = f; *to = t; *promo = pr ? pr + 1 : 0;     // python-chess piece codes: N=2 .. Q=5
    return SZ_OK;
}

int szh_move_to_action(int from, int to, int promo, int white) {
    return sz_action_index(from, to, promo ? promo - 1 : 0, white);
}

int szh_push_action(szh_game* g, int idx) {
    if (idx < 0 || idx >= SZ_NUM_ACTIONS) return SZ_ERR_INVALID;
    Gen gen;
    const SzPos p = g->ring[g->ply & (SZ_RING - 1)];
    generate(p, g->chess960, gen);
    if (!((gen.mask[idx >> 6] >> (idx & 63)) & 1)) return SZ_ERR_INVALID;       // ValueError("Invalid move"), chess_tensor.py:91-92
    int from, to, promo;
    sz_action_decode(p, idx, from, to, promo);
    g->ply++;
    g->ring[g->ply & (SZ_RING - 1)] = sz_make_move(p, from, to, promo, g->chess960);
    finish(g, g->ply);
    return SZ_OK;
}

int szh_push_move(szh_game* g, int from, int to, int promo) {
    int white = szm_turn(g->ring[g->ply & (SZ_RING - 1)].meta);
    int idx = sz_action_index(from, to, promo ? promo - 1 : 0, white);
    if (idx < 0) return SZ_ERR_INVALID;
    // a queen promotion must be spelled as one; a bare pawn move to the last rank is not a legal move
    const SzPos& p = g->ring[g->ply & (SZ_RING - 1)];
    bool pawn_last = (p.pc[SZ_P] & sz_bit(from)) && ((to >> 3) == (white ? 7 : 0));
    if (pawn_last != (promo != 0)) return SZ_ERR_INVALID;
    return szh_push_action(g, idx);
}

// status words: [0] turn, [1] ply, [2] halfmove, [3] ep square, [4] terminal, [5] value (0/-1), [6] in check,
// [7] reps, [8] n_legal, [9] outcome kind (1 mate, 2 insufficient, 3 stalemate, 4 seventy-five, 5 fivefold), [10] ep legal, [11] castling flags
void szh_status(const szh_game* g, int32_t* out) {
    const SzPos& p = g->ring[g->ply & (SZ_RING - 1)];
    out[0] = szm_turn(p.meta); out[1] = g->ply; out[2] = szm_half(p.meta); out[3] = szm_ep(p.meta);
    out[4] = szm_term(p.meta); out[5] = szm_loss(p.meta) ? -1 : 0; out[6] = szm_check(p.meta);
    out[7] = szm_reps(p.meta); out[8] = szm_nlegal(p.meta);
    int kind = 0;
    if (szm_term(p.meta)) {
        if (szm_loss(p.meta)) kind = 1;
        else if (sz_insufficient(p)) kind = 2;
        else if (szm_nlegal(p.meta) == 0) kind = 3;
        else if (szm_half(p.meta) >= 150) kind = 4;
        else kind = 5;
    }
    out[9] = kind; out[10] = szm_eplegal(p.meta); out[11] = sz_castling_flags(p);
}

// get_representation(): uint8 [119][8][8]
void szh_planes(const szh_game* g, uint8_t* out) {
    const SzPos& leaf = g->ring[g->ply & (SZ_RING - 1)];
    int vw = szm_turn(leaf.meta);
    for (int c = 0; c < SZ_NUM_PLANES; c++) {
        u64 bb = 0;
        if (c < 112) {
            int t = c / 14, k = c % 14;
            if (t <= g->ply && t < SZ_RING) {
                const SzPos& h = g->ring[(g->ply - t) & (SZ_RING - 1)];
                bb = sz_hist_plane((const u64*)&h, k, vw);
            }
        } else bb = sz_aux_plane(leaf, c - 112);
        for (int r = 0; r < 8; r++) {
            uint8_t bits = sz_row_bits(bb, r, vw);
            for (int j = 0; j < 8; j++) out[c * 64 + r * 8 + j] = (bits >> j) & 1;
        }
    }
}

uint64_t szh_perft(szh_game* g, int depth) { return perft_rec(g, depth); }

void szh_bitboards(const szh_game* g, uint64_t* out10) { memcpy(out10, &g->ring[g->ply & (SZ_RING - 1)], sizeof(SzPos)); }

// raw history for upload into the engine: copies min(ply+1, SZ_RING) records, newest last; returns count
int szh_export(const szh_game* g, void* ring_out /* SZ_RING * 80 bytes */, int32_t* ply, int32_t* chess960) {
    memcpy(ring_out, g->ring, sizeof(g->ring));
    *ply = g->ply; *chess960 = g->chess960;
    return SZ_OK;
}

int szh_is_chess960(const szh_game* g) { return g->chess960; }

// test hook: the bit-parallel plane extraction the kernels use (sz_lane_plane_bits) against its definition (sz_lane_plane_bit) on the
// current position's legal-target sets; returns the number of differing (view square, plane) pairs (0 = identical)
int szh_plane_bits_mismatches(const szh_game* g) {
    const SzPos& p = g->ring[g->ply & (SZ_RING - 1)];
    Gen gen;
    generate(p, g->chess960, gen);
    const int white = szm_turn(p.meta), flip = sz_view_flip(white);
    int bad = 0;
    for (int v = 0; v < 64; v++) {
        const int s = v ^ flip;
        const u64 T = gen.targets[s];
        const bool is_pawn = (p.pc[SZ_P] >> s) & 1;
        const SzPlaneBits pb = sz_lane_plane_bits(T, is_pawn, v, white);
        for (int pl = 0; pl < SZ_MASK_WORDS; pl++) {
            const uint32_t word = pl < 56 ? pb.q[pl / 7] : (pl < 64 ? pb.kn : pb.up);
            const int sh = pl < 56 ? pl % 7 : (pl < 64 ? pl - 56 : (pl - 64) % 3);
            const bool fast = (word >> sh) & 1u, ref = T && sz_lane_plane_bit(T, is_pawn, v, pl, white);
            bad += fast != ref;
        }
    }
    return bad;
}

}  // extern "C"
