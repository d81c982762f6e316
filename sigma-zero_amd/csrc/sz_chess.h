// sz_chess.h — chess rules for the MI355X self-play engine, shared by device kernels and host mirror.
//
// Replaces what the reference obtains from python-chess 1.10.0 through
// /root/reference/chess_tensor.py (call sites listed in SURVEY.md §8(c)): legal move generation
// (chess_tensor.py:91,146,158), push (:95), is_repetition (:101-102), castling rights (:114-117),
// halfmove clock (:118), is_game_over / outcome (:161-162), Chess960 start positions (:69).
//
// Design (MI355X-first, not a translation of the library):
//   * a position is an 80-byte record (10 x u64) that lives in HBM; one wavefront owns one board
//     and ONE LANE OWNS ONE SQUARE: sz_piece_targets()/sz_king_targets() are pure per-square
//     functions, so the 64 lanes generate all moves of a position with no cross-lane traffic
//     except one ballot (king danger squares);
//   * sliders use hyperbola quintessence with v_bfrev (no lookup tables, no LDS, no divergence
//     beyond the piece-type switch);
//   * legality comes from check masks and per-piece pin lines, not from make/unmake.
// Every function is __host__ __device__: the host-side ChessTensor mirror (sz_host.cpp) runs the very
// same code, and tests compare both against the independent oracle (oracle/oc_chess.c).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define SZ_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define SZ_HD inline
#endif

typedef uint64_t u64;
typedef uint32_t u32;

enum { SZ_P = 0, SZ_N = 1, SZ_B = 2, SZ_R = 3, SZ_Q = 4, SZ_K = 5 };

#define SZ_NUM_PLANES 119
#define SZ_NUM_ACTIONS 4672
#define SZ_MAX_CHILDREN 218
#define SZ_MASK_WORDS 73

#define SZ_RANK1 0x00000000000000FFULL
#define SZ_RANK8 0xFF00000000000000ULL
#define SZ_FILEA 0x0101010101010101ULL
#define SZ_FILEH 0x8080808080808080ULL
#define SZ_DARK  0xAA55AA55AA55AA55ULL
#define SZ_LIGHT 0x55AA55AA55AA55AAULL

// ---------------------------------------------------------------------------------------------
// Position record: 80 bytes, 16-byte aligned.  Words 0..7 are everything the plane encoder needs.
// ---------------------------------------------------------------------------------------------
struct alignas(16) SzPos {
    u64 pc[6];     // pawns, knights, bishops, rooks, queens, kings (both colours)
    u64 white;     // white occupancy (black = all & ~white)
    u64 meta;      // packed scalars, see SZM_* below
    u64 key;       // transposition key (python-chess Board._transposition_key, hashed to 64 bit)
    u64 castling;  // rook squares that still carry castling rights (clean_castling_rights)
};

// meta layout
#define SZM_EP_SHIFT 0        // 8 bits: ep square + 1 (0 = None); set after ANY double push like python-chess
#define SZM_TURN_BIT 8        // 1 = white to move
#define SZM_HALF_SHIFT 9      // 8 bits halfmove clock; saturates at 255, which loses nothing: every clock >= 150 is the same 75-move terminal state
#define SZM_IRREV_BIT 17      // move that led here was irreversible (Board.is_irreversible)
#define SZM_EPLEGAL_BIT 18    // Board.has_legal_en_passant()
#define SZM_REPS_SHIFT 19     // 3 bits: earlier occurrences in the reversible window, saturating at 4
#define SZM_TERM_BIT 22       // outcome(claim_draw=False) is not None
#define SZM_LOSS_BIT 23       // ... and it is checkmate (value -1 for the side to move)
#define SZM_CHECK_BIT 24
#define SZM_PLY_SHIFT 32      // 16 bits: len(move_stack)
#define SZM_NLEGAL_SHIFT 48   // 8 bits: number of legal moves

SZ_HD int  szm_ep(u64 m)      { return (int)((m >> SZM_EP_SHIFT) & 0xFF) - 1; }
SZ_HD int  szm_turn(u64 m)    { return (int)((m >> SZM_TURN_BIT) & 1); }
SZ_HD int  szm_half(u64 m)    { return (int)((m >> SZM_HALF_SHIFT) & 0xFF); }
SZ_HD int  szm_irrev(u64 m)   { return (int)((m >> SZM_IRREV_BIT) & 1); }
SZ_HD int  szm_eplegal(u64 m) { return (int)((m >> SZM_EPLEGAL_BIT) & 1); }
SZ_HD int  szm_reps(u64 m)    { return (int)((m >> SZM_REPS_SHIFT) & 7); }
SZ_HD int  szm_term(u64 m)    { return (int)((m >> SZM_TERM_BIT) & 1); }
SZ_HD int  szm_loss(u64 m)    { return (int)((m >> SZM_LOSS_BIT) & 1); }
SZ_HD int  szm_check(u64 m)   { return (int)((m >> SZM_CHECK_BIT) & 1); }
SZ_HD int  szm_ply(u64 m)     { return (int)((m >> SZM_PLY_SHIFT) & 0xFFFF); }
SZ_HD int  szm_nlegal(u64 m)  { return (int)((m >> SZM_NLEGAL_SHIFT) & 0xFF); }

SZ_HD u64 sz_all(const SzPos& p) { return p.pc[0] | p.pc[1] | p.pc[2] | p.pc[3] | p.pc[4] | p.pc[5]; }

// ---------------------------------------------------------------------------------------------
// bit helpers
// ---------------------------------------------------------------------------------------------
SZ_HD int sz_lsb(u64 b) { return __builtin_ctzll(b); }
SZ_HD int sz_msb(u64 b) { return 63 - __builtin_clzll(b); }
SZ_HD int sz_pop(u64 b) { return __builtin_popcountll(b); }
SZ_HD u64 sz_brev(u64 b) { return __builtin_bitreverse64(b); }
SZ_HD u64 sz_bit(int s) { return 1ULL << s; }

// ---------------------------------------------------------------------------------------------
// attack sets
// ---------------------------------------------------------------------------------------------
SZ_HD u64 sz_knight_att(u64 b) {
    u64 l1 = (b >> 1) & 0x7f7f7f7f7f7f7f7fULL, l2 = (b >> 2) & 0x3f3f3f3f3f3f3f3fULL;
    u64 r1 = (b << 1) & 0xfefefefefefefefeULL, r2 = (b << 2) & 0xfcfcfcfcfcfcfcfcULL;
    u64 h1 = l1 | r1, h2 = l2 | r2;
    return (h1 << 16) | (h1 >> 16) | (h2 << 8) | (h2 >> 8);
}
SZ_HD u64 sz_king_att(u64 b) {
    u64 h = ((b >> 1) & 0x7f7f7f7f7f7f7f7fULL) | ((b << 1) & 0xfefefefefefefefeULL);
    u64 row = h | b;
    return h | (row << 8) | (row >> 8);
}
// squares attacked by pawns of one colour standing on b
SZ_HD u64 sz_pawn_att(u64 b, int white) {
    u64 l = b & ~SZ_FILEA, r = b & ~SZ_FILEH;
    return white ? ((l << 7) | (r << 9)) : ((l >> 9) | (r >> 7));
}
SZ_HD u64 sz_rank_mask(int sq) { return 0xFFULL << (sq & 56); }
SZ_HD u64 sz_file_mask(int sq) { return SZ_FILEA << (sq & 7); }
SZ_HD u64 sz_diag_mask(int sq) {            // a1-h8 direction
    int d = 8 * (sq & 7) - (sq & 56);
    int nort = -d & (d >> 31), sout = d & (-d >> 31);
    return (0x8040201008040201ULL >> sout) << nort;
}
SZ_HD u64 sz_anti_mask(int sq) {            // h1-a8 direction
    int d = 56 - 8 * (sq & 7) - (sq & 56);
    int nort = -d & (d >> 31), sout = d & (-d >> 31);
    return (0x0102040810204080ULL >> sout) << nort;
}
// hyperbola quintessence along one line (mask includes sq); both directions, first blockers included
SZ_HD u64 sz_line_att(u64 occ, u64 mask, int sq) {
    u64 r = sz_bit(sq), m = mask & ~r, o = occ & m;
    u64 fwd = o - 2 * r;
    u64 rev = sz_brev(sz_brev(o) - 2 * sz_brev(r));
    return (fwd ^ rev) & m;
}
SZ_HD u64 sz_rook_att(int sq, u64 occ) { return sz_line_att(occ, sz_rank_mask(sq), sq) | sz_line_att(occ, sz_file_mask(sq), sq); }
SZ_HD u64 sz_bishop_att(int sq, u64 occ) { return sz_line_att(occ, sz_diag_mask(sq), sq) | sz_line_att(occ, sz_anti_mask(sq), sq); }

// pieces of colour `by_white` that attack sq under occupancy occ (Board.attackers_mask)
SZ_HD u64 sz_attackers(const SzPos& p, int sq, u64 occ, int by_white) {
    u64 b = sz_bit(sq);
    u64 side = by_white ? p.white : (sz_all(p) & ~p.white);
    u64 a = (sz_knight_att(b) & p.pc[SZ_N]) | (sz_king_att(b) & p.pc[SZ_K]) |
            (sz_pawn_att(b, !by_white) & p.pc[SZ_P]) |
            (sz_rook_att(sq, occ) & (p.pc[SZ_R] | p.pc[SZ_Q])) |
            (sz_bishop_att(sq, occ) & (p.pc[SZ_B] | p.pc[SZ_Q]));
    return a & side;
}

// squares strictly between two squares on a common line (0 if not aligned)
SZ_HD u64 sz_between(int a, int b) {
    if (a == b) return 0;
    u64 ab = sz_bit(a) | sz_bit(b);
    u64 m = sz_rank_mask(a);
    if (!(m & sz_bit(b))) { m = sz_file_mask(a);
        if (!(m & sz_bit(b))) { m = sz_diag_mask(a);
            if (!(m & sz_bit(b))) { m = sz_anti_mask(a);
                if (!(m & sz_bit(b))) return 0; } } }
    return sz_line_att(ab, m, a) & sz_line_att(ab, m, b);
}

// ---------------------------------------------------------------------------------------------
// per-position uniform information (identical in every lane of the board's wavefront)
// ---------------------------------------------------------------------------------------------
struct SzInfo {
    u64 occ, us, them;
    u64 checkers;
    u64 check_mask;     // squares a non-king move must land on (all ones when not in check)
    u64 need;           // squares whose "attacked by them with our king lifted" status the king needs
    int ksq;
    int white;
};

SZ_HD SzInfo sz_info(const SzPos& p) {
    SzInfo I;
    I.occ = sz_all(p);
    I.white = szm_turn(p.meta);
    I.us = I.white ? p.white : (I.occ & ~p.white);
    I.them = I.occ & ~I.us;
    u64 k = p.pc[SZ_K] & I.us;
    I.ksq = k ? sz_lsb(k) : 0;
    I.checkers = k ? sz_attackers(p, I.ksq, I.occ, !I.white) : 0;
    if (!I.checkers) I.check_mask = ~0ULL;
    else if (I.checkers & (I.checkers - 1)) I.check_mask = 0;
    else I.check_mask = I.checkers | sz_between(I.ksq, sz_lsb(I.checkers));
    u64 backrank = I.white ? SZ_RANK1 : SZ_RANK8;
    I.need = (sz_king_att(k) & ~I.us) | ((p.castling & backrank) ? backrank : 0);
    return I;
}

// "is sq attacked by them once our king is lifted off the board" (one lane per needed square)
SZ_HD bool sz_danger_at(const SzPos& p, const SzInfo& I, int sq) {
    return sz_attackers(p, sq, I.occ & ~sz_bit(I.ksq), !I.white) != 0;
}

// ---------------------------------------------------------------------------------------------
// legal targets of the piece on one square (king excluded).  Promotions are implied by a pawn
// reaching the last rank; en passant lands on the ep square.
// ---------------------------------------------------------------------------------------------
SZ_HD u64 sz_piece_targets(const SzPos& p, const SzInfo& I, int sq) {
    u64 b = sz_bit(sq);
    if (!(I.us & b) || (p.pc[SZ_K] & b)) return 0;
    // pin line, if any: the line through the king on which we are the only blocker before an enemy slider
    u64 allowed = ~0ULL;
    u64 kb = sz_bit(I.ksq);
    {
        u64 m = sz_rank_mask(sq); int ortho = 1;
        if (!(m & kb)) { m = sz_file_mask(sq);
            if (!(m & kb)) { ortho = 0; m = sz_diag_mask(sq);
                if (!(m & kb)) { m = sz_anti_mask(sq);
                    if (!(m & kb)) m = 0; } } }
        if (m) {
            u64 att = sz_line_att(I.occ, m, sq);
            if (att & kb) {
                // mask arithmetic instead of `ortho ? rooks : bishops`: hipcc lowers the per-lane select of two
                // wave-uniform values to a scratch-memory table lookup
                const u64 om = (u64)0 - (u64)ortho;
                u64 snipers = att & I.them & (p.pc[SZ_Q] | (p.pc[SZ_R] & om) | (p.pc[SZ_B] & ~om));
                if (snipers) allowed = m;
            }
        }
    }
    u64 t;
    if (p.pc[SZ_P] & b) {
        u64 empty = ~I.occ, push, dbl;
        if (I.white) { push = (b << 8) & empty; dbl = ((push & 0x0000000000FF0000ULL) << 8) & empty; }
        else         { push = (b >> 8) & empty; dbl = ((push & 0x0000FF0000000000ULL) >> 8) & empty; }
        u64 att = sz_pawn_att(b, I.white);
        t = (push | dbl | (att & I.them)) & I.check_mask & allowed;
        int ep = szm_ep(p.meta);
        if (ep >= 0 && (att & sz_bit(ep)) && !(I.occ & sz_bit(ep)) && (b & (I.white ? 0x000000FF00000000ULL : 0x00000000FF000000ULL))) {
            // en passant: decide by looking at the board after the capture
            u64 cap = I.white ? (sz_bit(ep) >> 8) : (sz_bit(ep) << 8);
            u64 occ2 = (I.occ ^ b ^ cap) | sz_bit(ep);
            u64 them2 = I.them & ~cap;
            u64 bb = sz_bit(I.ksq);
            u64 a = (sz_knight_att(bb) & p.pc[SZ_N]) | (sz_pawn_att(bb, I.white) & p.pc[SZ_P]) | (sz_king_att(bb) & p.pc[SZ_K]) |
                    (sz_rook_att(I.ksq, occ2) & (p.pc[SZ_R] | p.pc[SZ_Q])) | (sz_bishop_att(I.ksq, occ2) & (p.pc[SZ_B] | p.pc[SZ_Q]));
            if (!(a & them2)) t |= sz_bit(ep);
        }
        return t;
    }
    if (p.pc[SZ_N] & b) t = sz_knight_att(b);
    else {
        t = 0;
        if ((p.pc[SZ_B] | p.pc[SZ_Q]) & b) t |= sz_bishop_att(sq, I.occ);
        if ((p.pc[SZ_R] | p.pc[SZ_Q]) & b) t |= sz_rook_att(sq, I.occ);
    }
    return t & ~I.us & I.check_mask & allowed;
}

// king targets incl. castling.  `danger` = squares of I.need that are attacked (sz_danger_at).
// Castling restates Board.generate_castling_moves: Chess960 boards spell it king-takes-rook,
// classical boards e1g1 / e1c1.
SZ_HD u64 sz_king_targets(const SzPos& p, const SzInfo& I, u64 danger, int chess960) {
    u64 kb = sz_bit(I.ksq);
    u64 t = sz_king_att(kb) & ~I.us & ~danger;
    u64 backrank = I.white ? SZ_RANK1 : SZ_RANK8;
    u64 cands = p.castling & backrank;
    if (cands && (kb & backrank)) {
        int base = I.white ? 0 : 56;
        while (cands) {
            int rsq = sz_lsb(cands); cands &= cands - 1;
            u64 rook = sz_bit(rsq);
            int a_side = rsq < I.ksq;
            int kto = base + (a_side ? 2 : 6), rto = base + (a_side ? 3 : 5);
            u64 king_to = sz_bit(kto), rook_to = sz_bit(rto);
            u64 king_path = sz_between(I.ksq, kto), rook_path = sz_between(rsq, rto);
            if ((I.occ ^ kb ^ rook) & (king_path | rook_path | king_to | rook_to)) continue;
            if ((king_path | kb) & danger) continue;
            // king's destination is tested with the rook already relocated (matters in Chess960 only)
            if (sz_attackers(p, kto, I.occ ^ kb ^ rook ^ rook_to, !I.white)) continue;
            t |= chess960 ? rook : ((I.ksq == base + 4) ? king_to : rook);
        }
    }
    return t;
}

// ---------------------------------------------------------------------------------------------
// action codec (chess_tensor.py:221-306 actionToTensor, :309-410 tensorToAction), table-free.
// view index v = row*8+col of the mover-at-the-bottom board; real square = v ^ 56 (white), v ^ 7 (black)
// ---------------------------------------------------------------------------------------------
SZ_HD int sz_view_flip(int white) { return white ? 56 : 7; }

// direction tables as arithmetic: plane d in 0..7 -> (dcol, drow) clockwise from "up" (0,-1)
SZ_HD int sz_sx8(u64 packed, int d) { return (int)(int8_t)((packed >> (8 * d)) & 0xFF); }
SZ_HD int sz_dir_dc(int d) { return sz_sx8(0xFFFFFF0001010100ULL, d); }   //  0, 1, 1, 1, 0,-1,-1,-1
SZ_HD int sz_dir_dr(int d) { return sz_sx8(0xFF0001010100FFFFULL, d); }   // -1,-1, 0, 1, 1, 1, 0,-1
SZ_HD int sz_kn_dc(int d)  { return sz_sx8(0xFFFEFEFF01020201ULL, d); }   //  1, 2, 2, 1,-1,-2,-2,-1
SZ_HD int sz_kn_dr(int d)  { return sz_sx8(0xFEFF01020201FFFEULL, d); }   // -2,-1, 1, 2, 2, 1,-1,-2

// target view coordinates of (plane, from view row/col); returns false when off the board
SZ_HD bool sz_plane_target(int plane, int row, int col, int& trow, int& tcol, int& promo /*0,SZ_N,SZ_B,SZ_R*/) {
    promo = 0;
    if (plane < 56) { int d = plane / 7, n = 1 + plane % 7; tcol = col + sz_dir_dc(d) * n; trow = row + sz_dir_dr(d) * n; }
    else if (plane < 64) { int d = plane - 56; tcol = col + sz_kn_dc(d); trow = row + sz_kn_dr(d); }
    else { int q = plane - 64; trow = row - 1; int m = q % 3; tcol = col + (m == 1 ? 1 : (m == 2 ? -1 : 0)); promo = SZ_N + q / 3; }
    return trow >= 0 && trow < 8 && tcol >= 0 && tcol < 8;
}

// does the piece on view square v (legal targets T in real squares) own a legal move in action plane `plane`?
// This is one bit of actionsToTensor's mask (chess_tensor.py:190-218): mask[plane*64 + v].
SZ_HD bool sz_lane_plane_bit(u64 T, bool is_pawn, int v, int plane, int white) {
    int trow, tcol, promo;
    if (!sz_plane_target(plane, v >> 3, v & 7, trow, tcol, promo)) return false;
    if (plane >= 64 && (!is_pawn || trow != 0)) return false;   // under-promotion planes: pawns stepping onto view row 0
    int tsq = (trow * 8 + tcol) ^ sz_view_flip(white);
    return (T >> tsq) & 1;
}

// All 73 action-plane bits of the piece on view square v at once (the bit-parallel form of sz_lane_plane_bit, which stays as the
// definition and is what tests compare against).  q[d] bit k-1 = plane d*7+k-1 (direction d, distance k); kn bit d = plane 56+d;
// up bit m = planes 64+m, 67+m, 70+m (knight / bishop / rook promotion share the geometry).
// The legal-target set is brought into view coordinates (bit p = view square p); the four orientations {T, bit-reversed, byte-swapped,
// rank-mirrored} are the view board, its 180-degree rotation, its file mirror and the mirror's rotation in an order that depends on
// the side to move, so every ray becomes a +1, +8 or +9 walk from the lane's own (possibly transformed) index: a shift by the index,
// a ray mask and a shift-fold that gathers every 8th / 9th bit into 7 consecutive ones.  ~130 vector instructions per lane replace
// 73 evaluations of the coordinate arithmetic (~2,500).
struct SzPlaneBits { uint32_t q[8]; uint32_t kn, up; };
SZ_HD uint32_t sz_fold9(u64 x) { x &= 0x0040201008040201ULL; x |= x >> 8; x |= x >> 16; x |= x >> 32; return (uint32_t)x & 0x7Fu; }   // bits 9j -> j
SZ_HD uint32_t sz_fold8(u64 x) { x &= 0x0001010101010101ULL; x |= x >> 7; x |= x >> 14; x |= x >> 28; return (uint32_t)x & 0x7Fu; }   // bits 8j -> j
SZ_HD uint32_t sz_low_mask(int n) { return (1u << n) - 1u; }                                          // n in 0..7
SZ_HD SzPlaneBits sz_lane_plane_bits(u64 T, bool is_pawn, int v, int white) {
    SzPlaneBits o;
    const u64 R = sz_brev(T), S = __builtin_bswap64(T), Mi = __builtin_bswap64(R);
    const u64 Tv = white ? S : Mi;         // view board: bit p = real square p ^ sz_view_flip(white)
    const u64 Tr = white ? Mi : S;         // rotated by 180 degrees (index 63 - p)
    const u64 Tm = white ? R : T;          // files mirrored (index p ^ 7)
    const u64 Tmr = white ? T : R;         // mirrored and rotated
    const int row = v >> 3, col = v & 7, vm = v ^ 7;
    const u64 A = Tv >> v, B = Tr >> (63 - v), C = Tm >> vm, D = Tmr >> (63 - vm);
    const int e = 7 - col, sth = 7 - row;
    o.q[2] = (uint32_t)(A >> 1) & sz_low_mask(e);                                   // (+1, 0)  right
    o.q[6] = (uint32_t)(B >> 1) & sz_low_mask(col);                                 // (-1, 0)  left
    o.q[4] = sz_fold8(A >> 8);                                                      // ( 0,+1)  down
    o.q[0] = sz_fold8(B >> 8);                                                      // ( 0,-1)  up
    o.q[3] = sz_fold9(A >> 9) & sz_low_mask(e < sth ? e : sth);                     // (+1,+1)
    o.q[7] = sz_fold9(B >> 9) & sz_low_mask(col < row ? col : row);                 // (-1,-1)
    o.q[5] = sz_fold9(C >> 9) & sz_low_mask(col < sth ? col : sth);                 // (-1,+1) = (+1,+1) on the mirrored board
    o.q[1] = sz_fold9(D >> 9) & sz_low_mask(e < row ? e : row);                     // (+1,-1) = (-1,-1) on the mirrored board
    // knight jumps d = 0..7: (dc,dr) = (1,-2),(2,-1),(2,1),(1,2),(-1,2),(-2,1),(-2,-1),(-1,-2); the constants hold the view squares
    // from which jump d stays on the board
    uint32_t kn = 0;
    kn |= (uint32_t)((Tv >> ((v - 15) & 63)) & (0x7f7f7f7f7f7f0000ULL >> v) & 1) << 0;
    kn |= (uint32_t)((Tv >> ((v - 6) & 63)) & (0x3f3f3f3f3f3f3f00ULL >> v) & 1) << 1;
    kn |= (uint32_t)((Tv >> ((v + 10) & 63)) & (0x003f3f3f3f3f3f3fULL >> v) & 1) << 2;
    kn |= (uint32_t)((Tv >> ((v + 17) & 63)) & (0x00007f7f7f7f7f7fULL >> v) & 1) << 3;
    kn |= (uint32_t)((Tv >> ((v + 15) & 63)) & (0x0000fefefefefefeULL >> v) & 1) << 4;
    kn |= (uint32_t)((Tv >> ((v + 6) & 63)) & (0x00fcfcfcfcfcfcfcULL >> v) & 1) << 5;
    kn |= (uint32_t)((Tv >> ((v - 10) & 63)) & (0xfcfcfcfcfcfcfc00ULL >> v) & 1) << 6;
    kn |= (uint32_t)((Tv >> ((v - 17) & 63)) & (0xfefefefefefe0000ULL >> v) & 1) << 7;
    o.kn = kn;
    // under-promotions: a pawn on view row 1 stepping onto row 0 straight / to the right / to the left
    uint32_t up = 0;
    if (is_pawn && row == 1) {
        const uint32_t t2 = (uint32_t)(Tv >> ((v - 8) & 63)) & 3u;                  // bit 0: v-8 (straight), bit 1: v-7 (right)
        const uint32_t tl = (uint32_t)(Tv >> ((v - 9) & 63)) & 1u;                  // v-9 (left)
        up = (t2 & 1u) | ((col < 7 ? t2 >> 1 : 0u) << 1) | ((col > 0 ? tl : 0u) << 2);
    }
    o.up = up;
    return o;
}

// move -> action index.  promo: 0 none / SZ_N / SZ_B / SZ_R / SZ_Q
SZ_HD int sz_action_index(int from, int to, int promo, int white) {
    int f = from ^ sz_view_flip(white), t = to ^ sz_view_flip(white);
    int row = f >> 3, col = f & 7, dr = (t >> 3) - row, dc = (t & 7) - col;
    int adr = dr < 0 ? -dr : dr, adc = dc < 0 ? -dc : dc;
    if (dc == 0 || dr == 0 || adr == adc) {
        if (promo == SZ_N || promo == SZ_B || promo == SZ_R)
            return (64 + 3 * (promo - SZ_N) + (dc > 0 ? 1 : (dc < 0 ? 2 : 0))) * 64 + f;
        int n = adr > adc ? adr : adc;
        int sc = (dc > 0) - (dc < 0), sr = (dr > 0) - (dr < 0);
        int d = 0;
        for (int k = 0; k < 8; k++) if (sz_dir_dc(k) == sc && sz_dir_dr(k) == sr) d = k;
        return (d * 7 + n - 1) * 64 + f;
    }
    int d = -1;
    for (int k = 0; k < 8; k++) if (sz_kn_dc(k) == dc && sz_kn_dr(k) == dr) d = k;
    return d < 0 ? -1 : (56 + d) * 64 + f;
}

// action index -> move on position p (queen promotion inferred: pawn reaching the last rank on a sliding plane)
SZ_HD bool sz_action_decode(const SzPos& p, int idx, int& from, int& to, int& promo) {
    int white = szm_turn(p.meta);
    int plane = idx >> 6, v = idx & 63, trow, tcol;
    if (!sz_plane_target(plane, v >> 3, v & 7, trow, tcol, promo)) return false;
    from = v ^ sz_view_flip(white);
    to = (trow * 8 + tcol) ^ sz_view_flip(white);
    if (!promo && (p.pc[SZ_P] & sz_bit(from)) && ((to >> 3) == (white ? 7 : 0))) promo = SZ_Q;
    return true;
}

// ---------------------------------------------------------------------------------------------
// make move (Board.push).  Input move in the board's external form.  Output has pieces, white,
// castling, ep/turn/halfmove/ply and the IRREV flag; key, ep-legal, reps, terminal are finished
// by sz_finish_node once the child's own move generation is known.
// ---------------------------------------------------------------------------------------------
SZ_HD SzPos sz_make_move(const SzPos& q, int from, int to, int promo, int chess960) {
    SzPos p = q;
    int white = szm_turn(q.meta);
    u64 fb = sz_bit(from), tb = sz_bit(to);
    u64 occ = sz_all(q), us = white ? q.white : (occ & ~q.white), them = occ & ~us;
    int pt = 0;
#pragma unroll
    for (int k = 0; k < 6; k++) pt = (q.pc[k] & fb) ? k : pt;
    int half = szm_half(q.meta) + 1;
    bool zeroing = (pt == SZ_P) || (them & tb);
    // castling: king takes own rook (Chess960 form) or the classical two-step from the e-file
    bool castle = false; int rook_from = to;
    if (pt == SZ_K) {
        if (us & q.pc[SZ_R] & tb) castle = true;
        else if (!chess960 && (from == (white ? 4 : 60)) && ((to == from + 2) || (to == from - 2)) ) {
            castle = true; rook_from = (to > from) ? from + 3 : from - 4;
        }
    }
    u64 backrank = white ? SZ_RANK1 : SZ_RANK8;
    u64 touched = fb | tb | (castle ? sz_bit(rook_from) : 0);
    bool reduces = (q.castling & touched) || (pt == SZ_K && (q.castling & backrank));
    u64 castling = q.castling & ~touched;
    if (pt == SZ_K) castling &= ~backrank;

    int ep_new = -1;
    if (castle) {
        int a_side = rook_from < from;
        int base = white ? 0 : 56;
        u64 rb = sz_bit(rook_from), kto = sz_bit(base + (a_side ? 2 : 6)), rto = sz_bit(base + (a_side ? 3 : 5));
        p.pc[SZ_K] = (q.pc[SZ_K] & ~fb) | kto;
        p.pc[SZ_R] = (q.pc[SZ_R] & ~rb) | rto;
        if (white) p.white = (q.white & ~fb & ~rb) | kto | rto;
        zeroing = false;
    } else {
        // remove any captured piece on the target; en passant removes the pawn behind it
        u64 clear = fb | tb;
        if (pt == SZ_P) {
            int diff = to - from;
            if (diff == 16 && (from >> 3) == 1) ep_new = from + 8;
            else if (diff == -16 && (from >> 3) == 6) ep_new = from - 8;
            else if (to == szm_ep(q.meta) && (diff == 7 || diff == 9 || diff == -7 || diff == -9) && !(occ & tb))
                clear |= white ? (tb >> 8) : (tb << 8);
        }
        const int np = promo ? promo : pt;
        // no runtime-indexed register arrays (they would live in scratch memory): select with compile-time indices
#pragma unroll
        for (int k = 0; k < 6; k++) p.pc[k] = (q.pc[k] & ~clear) | (k == np ? tb : 0ULL);
        p.white = (q.white & ~clear) | (white ? tb : 0ULL);
    }
    if (zeroing) half = 0;
    if (half > 255) half = 255;
    int ply = szm_ply(q.meta) + 1; if (ply > 65535) ply = 65535;
    bool irrev = zeroing || reduces || szm_eplegal(q.meta);
    p.castling = castling;
    p.key = 0;
    p.meta = ((u64)(ep_new + 1) << SZM_EP_SHIFT) | ((u64)(!white) << SZM_TURN_BIT) | ((u64)half << SZM_HALF_SHIFT) |
             ((u64)irrev << SZM_IRREV_BIT) | ((u64)ply << SZM_PLY_SHIFT);
    return p;
}

// 64-bit hash of python-chess's _transposition_key (pieces, colours, turn, clean castling rights,
// ep square only when an en-passant capture is legal)
SZ_HD u64 sz_mix(u64 h, u64 w) {
    h ^= w + 0x9E3779B97F4A7C15ULL + (h << 6) + (h >> 2);
    h *= 0xBF58476D1CE4E5B9ULL; h ^= h >> 31;
    return h;
}
SZ_HD u64 sz_hash_key(const SzPos& p, int ep_legal) {
    u64 h = 0x243F6A8885A308D3ULL;
    for (int k = 0; k < 6; k++) h = sz_mix(h, p.pc[k]);
    h = sz_mix(h, p.white);
    h = sz_mix(h, p.castling);
    h = sz_mix(h, (u64)szm_turn(p.meta) | ((u64)(ep_legal ? szm_ep(p.meta) + 1 : 0) << 8));
    h *= 0x94D049BB133111EBULL; h ^= h >> 29;
    return h;
}

// Board.is_insufficient_material()
SZ_HD bool sz_insufficient_side(const SzPos& p, u64 side, u64 other) {
    if (side & (p.pc[SZ_P] | p.pc[SZ_R] | p.pc[SZ_Q])) return false;
    if (side & p.pc[SZ_N]) return sz_pop(side) <= 2 && !(other & ~p.pc[SZ_K] & ~p.pc[SZ_Q]);
    if (side & p.pc[SZ_B]) {
        bool same = !(p.pc[SZ_B] & SZ_DARK) || !(p.pc[SZ_B] & SZ_LIGHT);
        return same && !p.pc[SZ_P] && !p.pc[SZ_N];
    }
    return true;
}
SZ_HD bool sz_insufficient(const SzPos& p) {
    u64 all = sz_all(p), blk = all & ~p.white;
    return sz_insufficient_side(p, p.white, blk) && sz_insufficient_side(p, blk, p.white);
}

// complete a freshly made position once its move generation is known.
//   n_legal: number of legal moves; ep_legal: an en-passant capture is among them; reps: earlier
//   occurrences inside the reversible window (0..4).  Sets key-independent flags + terminal status
//   following Board.outcome(claim_draw=False) order: checkmate, insufficient material, stalemate,
//   seventy-five moves, fivefold repetition.
SZ_HD u64 sz_finish_meta(const SzPos& p, u64 checkers, int n_legal, int ep_legal, int reps) {
    u64 m = p.meta & ~(((u64)1 << SZM_EPLEGAL_BIT) | ((u64)7 << SZM_REPS_SHIFT) | ((u64)1 << SZM_TERM_BIT) |
                       ((u64)1 << SZM_LOSS_BIT) | ((u64)1 << SZM_CHECK_BIT) | ((u64)0xFF << SZM_NLEGAL_SHIFT));
    if (reps > 4) reps = 4;
    bool mate = checkers && n_legal == 0;
    bool term = mate || sz_insufficient(p) || n_legal == 0 || (szm_half(p.meta) >= 150 && n_legal > 0) || reps >= 4;
    m |= ((u64)(ep_legal != 0) << SZM_EPLEGAL_BIT) | ((u64)reps << SZM_REPS_SHIFT) | ((u64)term << SZM_TERM_BIT) |
         ((u64)mate << SZM_LOSS_BIT) | ((u64)(checkers != 0) << SZM_CHECK_BIT) | ((u64)(n_legal & 0xFF) << SZM_NLEGAL_SHIFT);
    return m;
}

// castling planes of chess_tensor.py:114-117: bit0 WK, bit1 WQ, bit2 BK, bit3 BQ
SZ_HD int sz_castling_flags(const SzPos& p) {
    int f = 0;
    u64 blk = sz_all(p) & ~p.white;
    u64 wk = p.pc[SZ_K] & p.white & SZ_RANK1, bk = p.pc[SZ_K] & blk & SZ_RANK8;
    u64 wr = p.castling & SZ_RANK1, br = p.castling & SZ_RANK8;
    if (wk) { if (wr > wk) f |= 1; if (wr & (wk - 1)) f |= 2; }
    if (bk) { if (br > bk) f |= 4; if (br & (bk - 1)) f |= 8; }
    return f;
}

// ---------------------------------------------------------------------------------------------
// plane encoder primitive (chess_tensor.py:131-142 get_representation + plane map in SURVEY §8(a)).
// Returns the 8 cells of (plane c, view row r) as a byte, bit j = view column j.
//   hist[t] = position t plies before the leaf (t = 0 is the leaf), valid[t] tells whether it exists.
//   view_white: side to move AT THE LEAF.
// ---------------------------------------------------------------------------------------------
SZ_HD uint8_t sz_row_bits(u64 bb, int r, int view_white) {
    if (view_white) return (uint8_t)(bb >> (8 * (7 - r)));                // row 0 = rank 8, col 0 = file a
    uint8_t x = (uint8_t)(bb >> (8 * r));                                // row 0 = rank 1, col 0 = file h
    return (uint8_t)(sz_brev((u64)x) >> 56);
}

// plane bitboard for history slot words (pc[0..5], white, meta) of ONE position
SZ_HD u64 sz_hist_plane(const u64* w /* 8 words */, int k /* 0..13 */, int view_white) {
    u64 all = w[0] | w[1] | w[2] | w[3] | w[4] | w[5];
    u64 own = view_white ? w[6] : (all & ~w[6]);
    if (k < 6) return w[k] & own;
    if (k < 12) return w[k - 6] & (all & ~own);
    int reps = szm_reps(w[7]);
    return (k == 12 ? reps >= 1 : reps >= 2) ? ~0ULL : 0ULL;
}
// the seven L planes (112..118) of the leaf position as all-ones / all-zeros
SZ_HD u64 sz_aux_plane(const SzPos& leaf, int j /* 0..6 */) {
    int white = szm_turn(leaf.meta), ply = szm_ply(leaf.meta);
    int cf = sz_castling_flags(leaf);
    if (ply == 0) cf = 15;                                              // start_board(): castling planes are ones unconditionally
    int own_k = white ? (cf & 1) : ((cf >> 2) & 1), own_q = white ? ((cf >> 1) & 1) : ((cf >> 3) & 1);
    int opp_k = white ? ((cf >> 2) & 1) : (cf & 1), opp_q = white ? ((cf >> 3) & 1) : ((cf >> 1) & 1);
    int v;
    switch (j) {
        case 0: v = white; break;
        case 1: v = ply > 0; break;
        case 2: v = own_k; break;
        case 3: v = own_q; break;
        case 4: v = opp_k; break;
        case 5: v = opp_q; break;
        default: v = ply > 0 && szm_half(leaf.meta) > 0; break;        // start_board(): no_progress plane starts at 0
    }
    return v ? ~0ULL : 0ULL;
}

// ---------------------------------------------------------------------------------------------
// start positions
// ---------------------------------------------------------------------------------------------
SZ_HD SzPos sz_start_from_backrank(u64 kn, u64 bi, u64 ro, u64 qu, u64 ki /* file masks, bits 0..7 */, int all_rooks_castle) {
    SzPos p;
    const u64 both = 0x0100000000000001ULL;                 // file bit on rank 1 and rank 8
    p.pc[SZ_P] = 0x00FF00000000FF00ULL;
    p.pc[SZ_N] = kn * both; p.pc[SZ_B] = bi * both; p.pc[SZ_R] = ro * both; p.pc[SZ_Q] = qu * both; p.pc[SZ_K] = ki * both;
    p.white = 0x000000000000FFFFULL;
    p.castling = all_rooks_castle ? p.pc[SZ_R] : (sz_bit(0) | sz_bit(7) | sz_bit(56) | sz_bit(63));
    p.meta = ((u64)1 << SZM_TURN_BIT) | ((u64)1 << SZM_IRREV_BIT);
    p.key = 0;
    return p;
}
// Board.from_chess960_pos(n) (Scharnagl numbering); n < 0 -> chess.Board()
SZ_HD SzPos sz_startpos(int scharnagl) {
    if (scharnagl < 0) return sz_start_from_backrank(0x42, 0x24, 0x81, 0x08, 0x10, 0);   // RNBQKBNR
    // file occupancy is tracked as an 8-bit mask (no runtime-indexed arrays: they would live in scratch memory on the GPU)
    int n = scharnagl;
    const int bw = n % 4; n /= 4;
    const int bd = n % 4; n /= 4;
    const int q = n % 6; n /= 6;
    u64 bi = sz_bit(bw * 2 + 1) | sz_bit(bd * 2);
    u64 used = bi;
    // k-th free file (k counted from the a-file)
    auto kth_free = [&](int k) -> u64 {
        u64 freem = ~used & 0xFF;
        for (int i = 0; i < k; i++) freem &= freem - 1;
        return freem & (~freem + 1);
    };
    const u64 qu = kth_free(q); used |= qu;
    // knights: n in 0..9 enumerates the 10 pairs of the 5 free files in lexicographic order
    int a = 0, b2 = 1, cnt = n;
    for (a = 0; a < 4; a++) { int span = 4 - a; if (cnt < span) { b2 = a + 1 + cnt; break; } cnt -= span; }
    const u64 kn = kth_free(a) | kth_free(b2); used |= kn;
    const u64 r1 = kth_free(0), ki = kth_free(1), r2 = kth_free(2);
    return sz_start_from_backrank(kn, bi, r1 | r2, qu, ki, 1);
}
