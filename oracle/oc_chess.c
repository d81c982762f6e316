/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see oc_chess.h for scope and parity status).
 * Restatement of python-chess 1.10.0 rules used by /root/reference/chess_tensor.py.
 * Deliberately simple: ray loops instead of lookup tables, pseudo-legal generation
 * filtered by make-move + king-safety.  The product HIP/host code uses a different
 * technique (pins / check masks, hyperbola quintessence) so the two implementations
 * cross-validate each other.
 */
#include "oc_chess.h"
#include <stdlib.h>
#include <string.h>
#include <ctype.h>

#define BIT(s) (1ULL << (s))
#define RANK_1 0x00000000000000FFULL
#define RANK_2 0x000000000000FF00ULL
#define RANK_3 0x0000000000FF0000ULL
#define RANK_4 0x00000000FF000000ULL
#define RANK_5 0x000000FF00000000ULL
#define RANK_6 0x0000FF0000000000ULL
#define RANK_7 0x00FF000000000000ULL
#define RANK_8 0xFF00000000000000ULL
#define FILE_A 0x0101010101010101ULL
#define DARK_SQUARES  0xAA55AA55AA55AA55ULL
#define LIGHT_SQUARES 0x55AA55AA55AA55AAULL

static inline int lsb(u64 b) { return __builtin_ctzll(b); }
static inline int msb(u64 b) { return 63 - __builtin_clzll(b); }
static inline int popcnt(u64 b) { return __builtin_popcountll(b); }
static inline int sq_file(int s) { return s & 7; }
static inline int sq_rank(int s) { return s >> 3; }

/* ------------------------------------------------------------------ attacks */

static u64 step_attacks(int sq, const int (*d)[2], int n) {
    u64 a = 0;
    for (int i = 0; i < n; i++) {
        int f = sq_file(sq) + d[i][0], r = sq_rank(sq) + d[i][1];
        if (f >= 0 && f < 8 && r >= 0 && r < 8) a |= BIT(r * 8 + f);
    }
    return a;
}
static u64 slide_attacks(int sq, u64 occ, const int (*d)[2], int n) {
    u64 a = 0;
    for (int i = 0; i < n; i++) {
        int f = sq_file(sq) + d[i][0], r = sq_rank(sq) + d[i][1];
        while (f >= 0 && f < 8 && r >= 0 && r < 8) {
            u64 b = BIT(r * 8 + f);
            a |= b;
            if (occ & b) break;
            f += d[i][0]; r += d[i][1];
        }
    }
    return a;
}
static const int KN_D[8][2] = {{1,2},{2,1},{2,-1},{1,-2},{-1,-2},{-2,-1},{-2,1},{-1,2}};
static const int KG_D[8][2] = {{1,0},{1,1},{0,1},{-1,1},{-1,0},{-1,-1},{0,-1},{1,-1}};
static const int RK_D[4][2] = {{1,0},{-1,0},{0,1},{0,-1}};
static const int BP_D[4][2] = {{1,1},{1,-1},{-1,1},{-1,-1}};

static u64 knight_att(int s) { return step_attacks(s, KN_D, 8); }
static u64 king_att(int s)   { return step_attacks(s, KG_D, 8); }
static u64 rook_att(int s, u64 occ)   { return slide_attacks(s, occ, RK_D, 4); }
static u64 bishop_att(int s, u64 occ) { return slide_attacks(s, occ, BP_D, 4); }
/* BB_PAWN_ATTACKS[color][sq]: squares attacked by a pawn of `color` standing on sq */
static u64 pawn_att(int color, int s) {
    static const int W[2][2] = {{-1,1},{1,1}}, B[2][2] = {{-1,-1},{1,-1}};
    return step_attacks(s, color == OC_WHITE ? W : B, 2);
}

/* Board.attackers_mask(color, square, occupied) */
static u64 attackers_mask(const oc_pos *p, int color, int sq, u64 occ) {
    u64 qr = p->bb[OC_QUEEN] | p->bb[OC_ROOK], qb = p->bb[OC_QUEEN] | p->bb[OC_BISHOP];
    u64 a = (king_att(sq) & p->bb[OC_KING]) | (knight_att(sq) & p->bb[OC_KNIGHT]) |
            (rook_att(sq, occ) & qr) | (bishop_att(sq, occ) & qb) |
            (pawn_att(!color, sq) & p->bb[OC_PAWN]);
    return a & p->occ_co[color];
}
static int is_attacked_by(const oc_pos *p, int color, int sq) {
    return attackers_mask(p, color, sq, p->occupied) != 0;
}

static int piece_type_at(const oc_pos *p, int sq) {
    u64 m = BIT(sq);
    if (!(p->occupied & m)) return 0;
    for (int t = OC_PAWN; t <= OC_KING; t++) if (p->bb[t] & m) return t;
    return 0;
}
int oc_piece_at(const oc_board *b, int sq, int *color) {
    int t = piece_type_at(&b->cur, sq);
    if (t && color) *color = (b->cur.occ_co[OC_WHITE] & BIT(sq)) ? OC_WHITE : OC_BLACK;
    return t;
}
static void remove_piece_at(oc_pos *p, int sq) {
    u64 m = ~BIT(sq);
    for (int t = OC_PAWN; t <= OC_KING; t++) p->bb[t] &= m;
    p->occ_co[0] &= m; p->occ_co[1] &= m; p->occupied &= m;
}
static void set_piece_at(oc_pos *p, int sq, int type, int color) {
    remove_piece_at(p, sq);
    u64 m = BIT(sq);
    p->bb[type] |= m; p->occ_co[color] |= m; p->occupied |= m;
}

/* Board.attacks_mask(square) */
static u64 attacks_mask(const oc_pos *p, int sq) {
    u64 m = BIT(sq);
    if (p->bb[OC_PAWN] & m) return pawn_att((p->occ_co[OC_WHITE] & m) ? OC_WHITE : OC_BLACK, sq);
    if (p->bb[OC_KNIGHT] & m) return knight_att(sq);
    if (p->bb[OC_KING] & m) return king_att(sq);
    u64 a = 0;
    if ((p->bb[OC_BISHOP] | p->bb[OC_QUEEN]) & m) a |= bishop_att(sq, p->occupied);
    if ((p->bb[OC_ROOK] | p->bb[OC_QUEEN]) & m) a |= rook_att(sq, p->occupied);
    return a;
}

/* ------------------------------------------------------- castling rights */

/* Board.clean_castling_rights() evaluated from scratch (used when the stack is empty) */
static u64 clean_castling_rights_scratch(const oc_pos *p, int chess960) {
    u64 castling = p->castling_rights & p->bb[OC_ROOK];
    u64 wc = castling & RANK_1 & p->occ_co[OC_WHITE];
    u64 bc = castling & RANK_8 & p->occ_co[OC_BLACK];
    if (!chess960) {
        wc &= (BIT(0) | BIT(7));
        bc &= (BIT(56) | BIT(63));
        if (!(p->occ_co[OC_WHITE] & p->bb[OC_KING] & BIT(4))) wc = 0;
        if (!(p->occ_co[OC_BLACK] & p->bb[OC_KING] & BIT(60))) bc = 0;
        return wc | bc;
    }
    u64 wk = p->occ_co[OC_WHITE] & p->bb[OC_KING] & RANK_1;
    u64 bk = p->occ_co[OC_BLACK] & p->bb[OC_KING] & RANK_8;
    if (!wk) wc = 0;
    if (!bk) bc = 0;
    u64 wa = wc & (~wc + 1), ba = bc & (~bc + 1);
    u64 wh = wc ? BIT(msb(wc)) : 0, bh = bc ? BIT(msb(bc)) : 0;
    if (wa && msb(wa) > msb(wk)) wa = 0;
    if (wh && msb(wh) < msb(wk)) wh = 0;
    if (ba && msb(ba) > msb(bk)) ba = 0;
    if (bh && msb(bh) < msb(bk)) bh = 0;
    return ba | bh | wa | wh;
}
/* Board.clean_castling_rights(): stored rights are trusted once a move was pushed */
static u64 clean_castling_rights(const oc_pos *p, int chess960, int stack_len) {
    if (stack_len) return p->castling_rights;
    return clean_castling_rights_scratch(p, chess960);
}

static int has_side_castling_rights(const oc_board *b, int color, int kingside) {
    const oc_pos *p = &b->cur;
    u64 backrank = color == OC_WHITE ? RANK_1 : RANK_8;
    u64 king_mask = p->bb[OC_KING] & p->occ_co[color] & backrank;
    if (!king_mask) return 0;
    u64 cr = clean_castling_rights(p, b->chess960, b->n_stack) & backrank;
    while (cr) {
        u64 rook = cr & (~cr + 1);
        if (kingside ? (rook > king_mask) : (rook < king_mask)) return 1;
        cr &= cr - 1;
    }
    return 0;
}
int oc_has_kingside_castling_rights(const oc_board *b, int color) { return has_side_castling_rights(b, color, 1); }
int oc_has_queenside_castling_rights(const oc_board *b, int color) { return has_side_castling_rights(b, color, 0); }

/* --------------------------------------------------------------- make move */

static oc_move to_chess960(const oc_pos *p, oc_move m) {
    if (m.from == 4 && (p->bb[OC_KING] & BIT(4))) {
        if (m.to == 6 && !(p->bb[OC_ROOK] & BIT(6))) m.to = 7;
        else if (m.to == 2 && !(p->bb[OC_ROOK] & BIT(2))) m.to = 0;
    } else if (m.from == 60 && (p->bb[OC_KING] & BIT(60))) {
        if (m.to == 62 && !(p->bb[OC_ROOK] & BIT(62))) m.to = 63;
        else if (m.to == 58 && !(p->bb[OC_ROOK] & BIT(58))) m.to = 56;
    }
    return m;
}
static oc_move from_chess960(const oc_pos *p, int chess960, oc_move m) {
    if (!chess960 && m.promo == 0) {
        if (m.from == 4 && (p->bb[OC_KING] & BIT(4))) {
            if (m.to == 7) m.to = 6; else if (m.to == 0) m.to = 2;
        } else if (m.from == 60 && (p->bb[OC_KING] & BIT(60))) {
            if (m.to == 63) m.to = 62; else if (m.to == 56) m.to = 58;
        }
    }
    return m;
}

/* Board.is_zeroing(move) on position p (before the move) */
static int is_zeroing(const oc_pos *p, oc_move m) {
    u64 touched = BIT(m.from) ^ BIT(m.to);
    return (touched & p->bb[OC_PAWN]) || (touched & p->occ_co[!p->turn]);
}

/* the state transition of Board.push (without the stack bookkeeping).
   `m` is in internal form (castling = king takes own rook). */
static void pos_push(oc_pos *p, oc_move m, int chess960, int stack_len) {
    p->castling_rights = clean_castling_rights(p, chess960, stack_len);
    int ep_square = p->ep_square;
    p->ep_square = -1;
    p->halfmove_clock += 1;
    if (p->turn == OC_BLACK) p->fullmove_number += 1;
    if (is_zeroing(p, m)) p->halfmove_clock = 0;

    u64 from_bb = BIT(m.from), to_bb = BIT(m.to);
    int piece_type = piece_type_at(p, m.from);
    remove_piece_at(p, m.from);
    int capture_square = m.to;
    int captured = piece_type_at(p, capture_square);

    p->castling_rights &= ~to_bb & ~from_bb;
    if (piece_type == OC_KING) p->castling_rights &= ~(p->turn == OC_WHITE ? RANK_1 : RANK_8);

    if (piece_type == OC_PAWN) {
        int diff = m.to - m.from;
        if (diff == 16 && sq_rank(m.from) == 1) p->ep_square = m.from + 8;
        else if (diff == -16 && sq_rank(m.from) == 6) p->ep_square = m.from - 8;
        else if (m.to == ep_square && (abs(diff) == 7 || abs(diff) == 9) && !captured) {
            int down = p->turn == OC_WHITE ? -8 : 8;
            capture_square = ep_square + down;
            remove_piece_at(p, capture_square);
        }
    }
    if (m.promo) piece_type = m.promo;

    int castling = piece_type == OC_KING && (p->occ_co[p->turn] & to_bb);
    if (castling) {
        int a_side = sq_file(m.to) < sq_file(m.from);
        remove_piece_at(p, m.from);
        remove_piece_at(p, m.to);
        int base = p->turn == OC_WHITE ? 0 : 56;
        if (a_side) { set_piece_at(p, base + 2, OC_KING, p->turn); set_piece_at(p, base + 3, OC_ROOK, p->turn); }
        else        { set_piece_at(p, base + 6, OC_KING, p->turn); set_piece_at(p, base + 5, OC_ROOK, p->turn); }
    } else {
        set_piece_at(p, m.to, piece_type, p->turn);
    }
    p->turn = !p->turn;
}

void oc_push(oc_board *b, oc_move m_ext) {
    if (b->n_stack == b->cap) {
        b->cap = b->cap ? b->cap * 2 : 64;
        b->stack = (oc_pos *)realloc(b->stack, sizeof(oc_pos) * b->cap);
        b->moves = (oc_move *)realloc(b->moves, sizeof(oc_move) * b->cap);
    }
    oc_move m = to_chess960(&b->cur, m_ext);
    b->stack[b->n_stack] = b->cur;
    b->moves[b->n_stack] = from_chess960(&b->cur, b->chess960, m);
    pos_push(&b->cur, m, b->chess960, b->n_stack);
    b->n_stack++;
}
void oc_pop(oc_board *b) {
    b->n_stack--;
    b->cur = b->stack[b->n_stack];
}

/* ------------------------------------------------------- move generation */

typedef struct { oc_move m[OC_MAX_MOVES]; int n; } mlist;
static void add(mlist *l, int f, int t, int pr) { l->m[l->n].from = (int8_t)f; l->m[l->n].to = (int8_t)t; l->m[l->n].promo = (int8_t)pr; l->n++; }

static u64 between_incl_path(int a, int b) {           /* chess.between(a, b) for same-rank squares */
    u64 r = 0;
    int lo = a < b ? a : b, hi = a < b ? b : a;
    for (int s = lo + 1; s < hi; s++) r |= BIT(s);
    return r;
}
static int attacked_for_king(const oc_pos *p, u64 path, u64 occ) {
    while (path) { int s = lsb(path); path &= path - 1; if (attackers_mask(p, !p->turn, s, occ)) return 1; }
    return 0;
}

/* Board.generate_castling_moves() — yields moves in the board's external form */
static void gen_castling(const oc_board *b, mlist *l) {
    const oc_pos *p = &b->cur;
    u64 backrank = p->turn == OC_WHITE ? RANK_1 : RANK_8;
    u64 king = p->occ_co[p->turn] & p->bb[OC_KING] & backrank;
    king &= (~king + 1);
    if (!king) return;
    int base = p->turn == OC_WHITE ? 0 : 56;
    u64 bb_c = BIT(base + 2), bb_d = BIT(base + 3), bb_f = BIT(base + 5), bb_g = BIT(base + 6);
    u64 cands = clean_castling_rights(p, b->chess960, b->n_stack) & backrank;
    while (cands) {                                       /* scan_reversed: order irrelevant for results */
        int cand = msb(cands); cands &= ~BIT(cand);
        u64 rook = BIT(cand);
        int a_side = rook < king;
        u64 king_to = a_side ? bb_c : bb_g, rook_to = a_side ? bb_d : bb_f;
        u64 king_path = between_incl_path(msb(king), msb(king_to));
        u64 rook_path = between_incl_path(cand, msb(rook_to));
        if (!(((p->occupied ^ king ^ rook) & (king_path | rook_path | king_to | rook_to)) ||
              attacked_for_king(p, king_path | king, p->occupied ^ king) ||
              attacked_for_king(p, king_to, p->occupied ^ king ^ rook ^ rook_to))) {
            oc_move m = { (int8_t)msb(king), (int8_t)cand, 0 };
            m = from_chess960(p, b->chess960, m);
            add(l, m.from, m.to, 0);
        }
    }
}

/* Board.generate_pseudo_legal_ep() */
static void gen_pseudo_ep(const oc_pos *p, mlist *l) {
    if (p->ep_square < 0) return;
    if (BIT(p->ep_square) & p->occupied) return;
    u64 capturers = p->bb[OC_PAWN] & p->occ_co[p->turn] & pawn_att(!p->turn, p->ep_square) &
                    (p->turn == OC_WHITE ? RANK_5 : RANK_4);
    while (capturers) { int c = msb(capturers); capturers &= ~BIT(c); add(l, c, p->ep_square, 0); }
}

/* Board.generate_pseudo_legal_moves() */
static void gen_pseudo(const oc_board *b, mlist *l) {
    const oc_pos *p = &b->cur;
    u64 ours = p->occ_co[p->turn];
    u64 non_pawns = ours & ~p->bb[OC_PAWN];
    while (non_pawns) {
        int f = msb(non_pawns); non_pawns &= ~BIT(f);
        u64 mv = attacks_mask(p, f) & ~ours;
        while (mv) { int t = msb(mv); mv &= ~BIT(t); add(l, f, t, 0); }
    }
    gen_castling(b, l);
    u64 pawns = p->bb[OC_PAWN] & ours;
    if (!pawns) return;
    u64 caps = pawns;
    while (caps) {
        int f = msb(caps); caps &= ~BIT(f);
        u64 tg = pawn_att(p->turn, f) & p->occ_co[!p->turn];
        while (tg) {
            int t = msb(tg); tg &= ~BIT(t);
            if (sq_rank(t) == 0 || sq_rank(t) == 7) { add(l, f, t, OC_QUEEN); add(l, f, t, OC_ROOK); add(l, f, t, OC_BISHOP); add(l, f, t, OC_KNIGHT); }
            else add(l, f, t, 0);
        }
    }
    u64 single, dbl;
    if (p->turn == OC_WHITE) { single = (pawns << 8) & ~p->occupied; dbl = (single << 8) & ~p->occupied & (RANK_3 | RANK_4); }
    else { single = (pawns >> 8) & ~p->occupied; dbl = (single >> 8) & ~p->occupied & (RANK_6 | RANK_5); }
    while (single) {
        int t = msb(single); single &= ~BIT(t);
        int f = t + (p->turn == OC_BLACK ? 8 : -8);
        if (sq_rank(t) == 0 || sq_rank(t) == 7) { add(l, f, t, OC_QUEEN); add(l, f, t, OC_ROOK); add(l, f, t, OC_BISHOP); add(l, f, t, OC_KNIGHT); }
        else add(l, f, t, 0);
    }
    while (dbl) {
        int t = msb(dbl); dbl &= ~BIT(t);
        int f = t + (p->turn == OC_BLACK ? 16 : -16);
        add(l, f, t, 0);
    }
    gen_pseudo_ep(p, l);
}

static int is_castling_move(const oc_pos *p, oc_move m) {   /* Board.is_castling */
    if (p->bb[OC_KING] & BIT(m.from)) {
        int diff = sq_file(m.from) - sq_file(m.to);
        return abs(diff) > 1 || ((p->bb[OC_ROOK] & p->occ_co[p->turn] & BIT(m.to)) != 0);
    }
    return 0;
}

/* legality by construction: play the move, then ask whether the mover's king is attacked.
   Castling legality is fully decided inside gen_castling (python-chess _is_safe returns
   True for castling moves). */
static int leaves_king_safe(const oc_board *b, oc_move m) {
    if (is_castling_move(&b->cur, m)) return 1;
    oc_pos q = b->cur;
    int us = q.turn;
    pos_push(&q, to_chess960(&q, m), b->chess960, 1 /* rights already irrelevant for safety */);
    u64 k = q.bb[OC_KING] & q.occ_co[us];
    if (!k) return 1;
    return !is_attacked_by(&q, !us, lsb(k));
}

int oc_legal_moves(const oc_board *b, oc_move *out) {
    mlist l; l.n = 0;
    gen_pseudo(b, &l);
    int n = 0;
    for (int i = 0; i < l.n; i++) if (leaves_king_safe(b, l.m[i])) out[n++] = l.m[i];
    return n;
}
int oc_is_legal(const oc_board *b, oc_move m) {
    oc_move mv[OC_MAX_MOVES];
    int n = oc_legal_moves(b, mv);
    for (int i = 0; i < n; i++) if (mv[i].from == m.from && mv[i].to == m.to && mv[i].promo == m.promo) return 1;
    return 0;
}
static int any_legal(const oc_board *b) { oc_move mv[OC_MAX_MOVES]; return oc_legal_moves(b, mv) > 0; }

int oc_is_check(const oc_board *b) {
    const oc_pos *p = &b->cur;
    u64 k = p->bb[OC_KING] & p->occ_co[p->turn];
    return k && is_attacked_by(p, !p->turn, lsb(k));
}

int oc_has_legal_en_passant(const oc_board *b) {
    if (b->cur.ep_square < 0) return 0;
    mlist l; l.n = 0;
    gen_pseudo_ep(&b->cur, &l);
    for (int i = 0; i < l.n; i++) if (leaves_king_safe(b, l.m[i])) return 1;
    return 0;
}

/* ------------------------------------------------------------- repetition */

typedef struct { u64 bb[7]; u64 w, bl; int turn; u64 cr; int ep; } tkey;
static tkey transposition_key(const oc_board *b) {
    tkey k; memset(&k, 0, sizeof k);
    for (int t = 1; t <= 6; t++) k.bb[t] = b->cur.bb[t];
    k.w = b->cur.occ_co[OC_WHITE]; k.bl = b->cur.occ_co[OC_BLACK];
    k.turn = b->cur.turn;
    k.cr = clean_castling_rights(&b->cur, b->chess960, b->n_stack);
    k.ep = oc_has_legal_en_passant(b) ? b->cur.ep_square : -1;
    return k;
}
static int tkey_eq(const tkey *a, const tkey *b) {
    for (int t = 1; t <= 6; t++) if (a->bb[t] != b->bb[t]) return 0;
    return a->w == b->w && a->bl == b->bl && a->turn == b->turn && a->cr == b->cr && a->ep == b->ep;
}

/* Board._reduces_castling_rights(move) on the current position */
static int reduces_castling_rights(const oc_board *b, oc_move m) {
    const oc_pos *p = &b->cur;
    u64 cr = clean_castling_rights(p, b->chess960, b->n_stack);
    u64 touched = BIT(m.from) ^ BIT(m.to);
    return (touched & cr) ||
           ((cr & RANK_1) && (touched & p->bb[OC_KING] & p->occ_co[OC_WHITE])) ||
           ((cr & RANK_8) && (touched & p->bb[OC_KING] & p->occ_co[OC_BLACK]));
}
/* Board.is_irreversible(move) on the current position */
static int is_irreversible(const oc_board *b, oc_move m) {
    return is_zeroing(&b->cur, m) || reduces_castling_rights(b, m) || oc_has_legal_en_passant(b);
}

/* Board.is_repetition(count).  The occupancy-only fast path of the library is a pure
   shortcut and is omitted; the walk-back below is the defining loop. */
int oc_is_repetition(const oc_board *b0, int count) {
    oc_board w = *b0;                         /* shallow view: we only move n_stack / cur */
    tkey key = transposition_key(&w);
    int result = 0;
    for (;;) {
        if (count <= 1) { result = 1; break; }
        if (w.n_stack < count - 1) break;
        oc_move mv = w.moves[w.n_stack - 1];
        w.n_stack--; w.cur = w.stack[w.n_stack];          /* pop */
        if (is_irreversible(&w, mv)) break;
        tkey k2 = transposition_key(&w);
        if (tkey_eq(&k2, &key)) count--;
    }
    return result;
}

/* --------------------------------------------------------------- outcome */

static int has_insufficient_material(const oc_pos *p, int color) {
    if (p->occ_co[color] & (p->bb[OC_PAWN] | p->bb[OC_ROOK] | p->bb[OC_QUEEN])) return 0;
    if (p->occ_co[color] & p->bb[OC_KNIGHT])
        return popcnt(p->occ_co[color]) <= 2 &&
               !(p->occ_co[!color] & ~p->bb[OC_KING] & ~p->bb[OC_QUEEN]);
    if (p->occ_co[color] & p->bb[OC_BISHOP]) {
        int same_color = !(p->bb[OC_BISHOP] & DARK_SQUARES) || !(p->bb[OC_BISHOP] & LIGHT_SQUARES);
        return same_color && !p->bb[OC_PAWN] && !p->bb[OC_KNIGHT];
    }
    return 1;
}
int oc_is_insufficient_material(const oc_board *b) {
    return has_insufficient_material(&b->cur, OC_WHITE) && has_insufficient_material(&b->cur, OC_BLACK);
}

int oc_outcome(const oc_board *b, int *winner) {
    if (winner) *winner = -1;
    int legal = any_legal(b);
    if (oc_is_check(b) && !legal) { if (winner) *winner = !b->cur.turn; return 1; }
    if (oc_is_insufficient_material(b)) return 2;
    if (!legal) return 3;
    if (b->cur.halfmove_clock >= 150 && legal) return 4;
    if (oc_is_repetition(b, 5)) return 5;
    return 0;
}

/* ---------------------------------------------------------- construction */

static void pos_clear(oc_pos *p) { memset(p, 0, sizeof *p); p->ep_square = -1; p->turn = OC_WHITE; p->fullmove_number = 1; }

static oc_board *board_alloc(void) {
    oc_board *b = (oc_board *)calloc(1, sizeof *b);
    pos_clear(&b->cur);
    return b;
}
static void backrank_setup(oc_pos *p, const int files_type[8]) {
    pos_clear(p);
    for (int f = 0; f < 8; f++) {
        set_piece_at(p, f, files_type[f], OC_WHITE);
        set_piece_at(p, 56 + f, files_type[f], OC_BLACK);
        set_piece_at(p, 8 + f, OC_PAWN, OC_WHITE);
        set_piece_at(p, 48 + f, OC_PAWN, OC_BLACK);
    }
}
oc_board *oc_board_new(void) {
    static const int std[8] = {OC_ROOK, OC_KNIGHT, OC_BISHOP, OC_QUEEN, OC_KING, OC_BISHOP, OC_KNIGHT, OC_ROOK};
    oc_board *b = board_alloc();
    backrank_setup(&b->cur, std);
    b->cur.castling_rights = BIT(0) | BIT(7) | BIT(56) | BIT(63);
    b->chess960 = 0;
    return b;
}
/* Board.set_chess960_pos(scharnagl) */
oc_board *oc_board_new_960(int scharnagl) {
    int files[8] = {0,0,0,0,0,0,0,0};
    int n = scharnagl, bw, bbq, q;
    bw = n % 4; n /= 4;
    bbq = n % 4; n /= 4;
    q = n % 6; n /= 6;
    int n1, n2 = 0;
    for (n1 = 0; n1 < 4; n1++) {
        n2 = n + (3 - n1) * (4 - n1) / 2 - 5;
        if (n1 < n2 && 1 <= n2 && n2 <= 4) break;
    }
    int bw_file = bw * 2 + 1, bb_file = bbq * 2;
    files[bw_file] = OC_BISHOP; files[bb_file] = OC_BISHOP;
    int lo = bw_file < bb_file ? bw_file : bb_file, hi = bw_file < bb_file ? bb_file : bw_file;
    int q_file = q;
    q_file += (lo <= q_file);
    q_file += (hi <= q_file);
    files[q_file] = OC_QUEEN;
    for (int i = 0; i < 8; i++) {
        if (!files[i]) {
            if (n1 == 0 || n2 == 0) files[i] = OC_KNIGHT;
            n1--; n2--;
        }
    }
    int i;
    for (i = 0; i < 8; i++) if (!files[i]) { files[i] = OC_ROOK; break; }
    for (i = 1; i < 8; i++) if (!files[i]) { files[i] = OC_KING; break; }
    for (i = 2; i < 8; i++) if (!files[i]) { files[i] = OC_ROOK; break; }
    oc_board *b = board_alloc();
    backrank_setup(&b->cur, files);
    b->cur.castling_rights = b->cur.bb[OC_ROOK];
    b->chess960 = 1;
    return b;
}

oc_board *oc_board_from_fen(const char *fen, int chess960) {
    oc_board *b = board_alloc();
    b->chess960 = chess960;
    oc_pos *p = &b->cur;
    int r = 7, f = 0;
    const char *c = fen;
    for (; *c && *c != ' '; c++) {
        if (*c == '/') { r--; f = 0; }
        else if (isdigit((unsigned char)*c)) f += *c - '0';
        else {
            int color = isupper((unsigned char)*c) ? OC_WHITE : OC_BLACK, t = 0;
            switch (tolower((unsigned char)*c)) { case 'p': t = OC_PAWN; break; case 'n': t = OC_KNIGHT; break; case 'b': t = OC_BISHOP; break;
                case 'r': t = OC_ROOK; break; case 'q': t = OC_QUEEN; break; case 'k': t = OC_KING; break; }
            if (t) set_piece_at(p, r * 8 + f, t, color);
            f++;
        }
    }
    while (*c == ' ') c++;
    p->turn = (*c == 'b') ? OC_BLACK : OC_WHITE;
    while (*c && *c != ' ') c++;
    while (*c == ' ') c++;
    /* castling field (KQkq, Shredder-FEN file letters, or '-'), as Board._set_castling_fen */
    p->castling_rights = 0;
    for (; *c && *c != ' '; c++) {
        if (*c == '-') continue;
        int color = isupper((unsigned char)*c) ? OC_WHITE : OC_BLACK;
        char flag = (char)tolower((unsigned char)*c);
        u64 backrank = color == OC_WHITE ? RANK_1 : RANK_8;
        u64 rooks = p->occ_co[color] & p->bb[OC_ROOK] & backrank;
        u64 king = p->occ_co[color] & p->bb[OC_KING] & backrank;
        if (flag == 'q') {
            if (king && rooks && lsb(rooks) < lsb(king)) p->castling_rights |= rooks & (~rooks + 1);
            else p->castling_rights |= FILE_A & backrank;
        } else if (flag == 'k') {
            int rook = rooks ? msb(rooks) : -1;
            if (king && rook >= 0 && msb(king) < rook) p->castling_rights |= BIT(rook);
            else p->castling_rights |= (FILE_A << 7) & backrank;
        } else if (flag >= 'a' && flag <= 'h') {
            p->castling_rights |= (FILE_A << (flag - 'a')) & backrank;
        }
    }
    while (*c == ' ') c++;
    if (*c && *c != '-') { p->ep_square = (c[0] - 'a') + 8 * (c[1] - '1'); }
    while (*c && *c != ' ') c++;
    while (*c == ' ') c++;
    if (*c) { p->halfmove_clock = atoi(c); while (*c && *c != ' ') c++; while (*c == ' ') c++; }
    if (*c) p->fullmove_number = atoi(c);
    return b;
}

oc_board *oc_board_copy(const oc_board *s) {
    oc_board *b = (oc_board *)malloc(sizeof *b);
    *b = *s;
    b->cap = s->n_stack + 8;
    b->stack = (oc_pos *)malloc(sizeof(oc_pos) * b->cap);
    b->moves = (oc_move *)malloc(sizeof(oc_move) * b->cap);
    if (s->n_stack) {
        memcpy(b->stack, s->stack, sizeof(oc_pos) * s->n_stack);
        memcpy(b->moves, s->moves, sizeof(oc_move) * s->n_stack);
    }
    return b;
}
void oc_board_free(oc_board *b) { if (!b) return; free(b->stack); free(b->moves); free(b); }

u64 oc_perft(oc_board *b, int depth) {
    oc_move mv[OC_MAX_MOVES];
    int n = oc_legal_moves(b, mv);
    if (depth <= 1) return depth == 1 ? (u64)n : 1;
    u64 t = 0;
    for (int i = 0; i < n; i++) { oc_push(b, mv[i]); t += oc_perft(b, depth - 1); oc_pop(b); }
    return t;
}

void oc_board_fen_pieces(const oc_board *b, char *out) {
    static const char sym[] = " pnbrqk";
    int k = 0;
    for (int r = 7; r >= 0; r--) {
        int empty = 0;
        for (int f = 0; f < 8; f++) {
            int color = 0, t = oc_piece_at(b, r * 8 + f, &color);
            if (!t) { empty++; continue; }
            if (empty) { out[k++] = (char)('0' + empty); empty = 0; }
            out[k++] = color == OC_WHITE ? (char)toupper(sym[t]) : sym[t];
        }
        if (empty) out[k++] = (char)('0' + empty);
        if (r) out[k++] = '/';
    }
    out[k] = 0;
}

int oc_board_turn(const oc_board *b) { return b->cur.turn; }
int oc_board_halfmove_clock(const oc_board *b) { return b->cur.halfmove_clock; }
int oc_board_ply(const oc_board *b) { return b->n_stack; }
int oc_board_ep_square(const oc_board *b) { return b->cur.ep_square; }
u64 oc_board_castling_rights(const oc_board *b) { return b->cur.castling_rights; }
int oc_board_is_chess960(const oc_board *b) { return b->chess960; }
/* raw bitboards: pawns,knights,bishops,rooks,queens,kings,white,black */
void oc_board_bitboards(const oc_board *b, u64 *out) {
    for (int t = 1; t <= 6; t++) out[t - 1] = b->cur.bb[t];
    out[6] = b->cur.occ_co[OC_WHITE]; out[7] = b->cur.occ_co[OC_BLACK];
}
