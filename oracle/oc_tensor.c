/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see oc_tensor.h).
 * Literal restatement of /root/reference/chess_tensor.py: the two 119-plane history
 * stacks are kept as byte arrays and shifted on every move exactly like the torch.cat
 * sequence at chess_tensor.py:120-129.
 */
#include "oc_tensor.h"
#include <stdlib.h>
#include <string.h>

#define M_ 14
#define T_ 8
#define L_ 7

/* chess_tensor.py:38-63  __board_to_tensor: [channel][row][col], row,col = divmod(square, 8) */
static void board_to_tensor(const oc_board *b, uint8_t *out /* 12*64 */) {
    memset(out, 0, 12 * 64);
    for (int sq = 0; sq < 64; sq++) {
        int color = 0, t = oc_piece_at(b, sq, &color);
        if (!t) continue;
        int channel = t - 1;                       /* PAWN..KING -> 0..5 */
        if (color == OC_BLACK) channel += 6;
        out[channel * 64 + sq] = 1;               /* row*8+col == square */
    }
}
static void fill_plane(uint8_t *p, int v) { memset(p, v ? 1 : 0, 64); }

/* chess_tensor.py:65-86 start_board (tensor part) */
static void start_tensors(oc_ct *ct) {
    uint8_t cur[14 * 64];
    board_to_tensor(ct->board, cur);
    memset(cur + 12 * 64, 0, 2 * 64);
    memset(ct->representation, 0, sizeof ct->representation);
    memset(ct->black_representation, 0, sizeof ct->black_representation);
    memcpy(ct->representation, cur, 14 * 64);
    memcpy(ct->black_representation, cur + 6 * 64, 6 * 64);
    memcpy(ct->black_representation + 6 * 64, cur, 6 * 64);
    /* L planes: [colour, total_moves=0, castling=1,1,1,1, no_progress=0] */
    uint8_t *Lw = ct->representation + (M_ * T_) * 64, *Lb = ct->black_representation + (M_ * T_) * 64;
    fill_plane(Lw + 0 * 64, 1); fill_plane(Lb + 0 * 64, 0);
    for (int i = 2; i <= 5; i++) { fill_plane(Lw + i * 64, 1); fill_plane(Lb + i * 64, 1); }
}

oc_ct *oc_ct_from_board(oc_board *b) {
    oc_ct *ct = (oc_ct *)calloc(1, sizeof *ct);
    ct->board = b;
    start_tensors(ct);
    return ct;
}
oc_ct *oc_ct_new(int chess960, int scharnagl) {
    return oc_ct_from_board(chess960 ? oc_board_new_960(scharnagl) : oc_board_new());
}
oc_ct *oc_ct_copy(const oc_ct *s) {
    oc_ct *ct = (oc_ct *)malloc(sizeof *ct);
    memcpy(ct, s, sizeof *ct);
    ct->board = oc_board_copy(s->board);
    return ct;
}
void oc_ct_free(oc_ct *ct) { if (!ct) return; oc_board_free(ct->board); free(ct); }

/* chess_tensor.py:88-129 move_piece */
int oc_ct_move_piece(oc_ct *ct, oc_move m) {
    if (!oc_is_legal(ct->board, m)) return -1;             /* :91-92 ValueError("Invalid move") */
    oc_push(ct->board, m);                                 /* :95 */
    uint8_t cur[14 * 64];
    board_to_tensor(ct->board, cur);                       /* :98 */
    fill_plane(cur + 12 * 64, oc_is_repetition(ct->board, 2));   /* :101,104 */
    fill_plane(cur + 13 * 64, oc_is_repetition(ct->board, 3));   /* :102,105 */

    uint8_t Lw[7 * 64], Lb[7 * 64];
    int total_moves = ct->board->n_stack != 0;             /* :113 bool(len(move_stack)) */
    int wk = oc_has_kingside_castling_rights(ct->board, OC_WHITE);
    int wq = oc_has_queenside_castling_rights(ct->board, OC_WHITE);
    int bk = oc_has_kingside_castling_rights(ct->board, OC_BLACK);
    int bq = oc_has_queenside_castling_rights(ct->board, OC_BLACK);
    int no_progress = ct->board->cur.halfmove_clock != 0;  /* :118 bool(halfmove_clock) */
    const int lw[7] = {1, total_moves, wk, wq, bk, bq, no_progress};   /* :120 */
    const int lb[7] = {0, total_moves, bk, bq, wk, wq, no_progress};   /* :121 */
    for (int i = 0; i < 7; i++) { fill_plane(Lw + i * 64, lw[i]); fill_plane(Lb + i * 64, lb[i]); }

    /* :124-129  rep = cat([current, rep[:-M-L], L]) */
    const size_t keep = (size_t)(OC_PLANES - M_ - L_) * 64;          /* 98 planes */
    memmove(ct->representation + 14 * 64, ct->representation, keep);
    memcpy(ct->representation, cur, 14 * 64);
    memcpy(ct->representation + 14 * 64 + keep, Lw, 7 * 64);
    memmove(ct->black_representation + 14 * 64, ct->black_representation, keep);
    memcpy(ct->black_representation, cur + 6 * 64, 6 * 64);
    memcpy(ct->black_representation + 6 * 64, cur, 6 * 64);
    memcpy(ct->black_representation + 12 * 64, cur + 12 * 64, 2 * 64);
    memcpy(ct->black_representation + 14 * 64 + keep, Lb, 7 * 64);
    return 0;
}

/* chess_tensor.py:131-142 get_representation */
void oc_ct_get_representation(const oc_ct *ct, uint8_t *out) {
    if (ct->board->cur.turn == OC_WHITE) {                 /* torch.flip(representation, [1]) */
        for (int c = 0; c < OC_PLANES; c++)
            for (int r = 0; r < 8; r++)
                memcpy(out + c * 64 + r * 8, ct->representation + c * 64 + (7 - r) * 8, 8);
    } else {                                               /* torch.flip(black_representation, [2]) */
        for (int c = 0; c < OC_PLANES; c++)
            for (int r = 0; r < 8; r++)
                for (int k = 0; k < 8; k++)
                    out[c * 64 + r * 8 + k] = ct->black_representation[c * 64 + r * 8 + (7 - k)];
    }
}

/* chess_tensor.py:160-172 */
int oc_ct_get_value_and_terminated(const oc_ct *ct, int *value) {
    int winner, o = oc_outcome(ct->board, &winner);
    if (o) { *value = (winner < 0) ? 0 : -1; return 1; }
    *value = 0;
    return 0;
}

/* ------------------------------------------------------------------ codec */

static int sgn(int v) { return v > 0 ? 1 : (v < 0 ? -1 : 0); }
static const int DIRS[8][2] = {{0,-1},{1,-1},{1,0},{1,1},{0,1},{-1,1},{-1,0},{-1,-1}};      /* chess_tensor.py:225-234 / :329-338 */
static const int KNIGHTS[8][2] = {{1,-2},{2,-1},{2,1},{1,2},{-1,2},{-2,1},{-2,-1},{-1,-2}}; /* :236-245 / :340-349 */

/* chess_tensor.py:221-306 */
int oc_action_to_index(oc_move m, int color) {
    int row, col, toRow, toCol;
    if (color == OC_WHITE) { row = 7 - m.from / 8; col = m.from % 8; toRow = 7 - m.to / 8; toCol = m.to % 8; }
    else                   { row = m.from / 8; col = 7 - m.from % 8; toRow = m.to / 8; toCol = 7 - m.to % 8; }
    int dr = toRow - row, dc = toCol - col;
    if (toCol == col || toRow == row || abs(dr) == abs(dc)) {
        if (m.promo == OC_KNIGHT || m.promo == OC_BISHOP || m.promo == OC_ROOK) {   /* uci()[-1] in "nbr" */
            int i = 3 * (m.promo == OC_KNIGHT ? 0 : (m.promo == OC_BISHOP ? 1 : 2));
            if (toCol > col) i += 1; else if (toCol < col) i += 2;
            return (64 + i) * 64 + row * 8 + col;
        }
        int squares = abs(dr) > abs(dc) ? abs(dr) : abs(dc);
        int d = -1;
        for (int k = 0; k < 8; k++) if (DIRS[k][0] == sgn(dc) && DIRS[k][1] == sgn(dr)) d = k;
        return (d * 7 + (squares - 1)) * 64 + row * 8 + col;
    }
    int kn = -1;
    for (int k = 0; k < 8; k++) if (KNIGHTS[k][0] == dc && KNIGHTS[k][1] == dr) kn = k;
    if (kn < 0) return -1;                                  /* KeyError in the reference */
    return (56 + kn) * 64 + row * 8 + col;
}

/* chess_tensor.py:309-410, one index */
int oc_index_to_action(int idx, int color, const oc_move *qp, int n_qp, oc_move *out) {
    int plane = idx / 64, rem = idx % 64, row = rem / 8, col = rem % 8;
    int toRow, toCol, promo = 0, may_be_queen_promo = 0;
    if (plane < 56) {
        int d = plane / 7, squares = 1 + plane % 7;
        toCol = col + DIRS[d][0] * squares; toRow = row + DIRS[d][1] * squares;
        may_be_queen_promo = 1;
    } else if (plane < 64) {
        int d = plane - 56;
        toCol = col + KNIGHTS[d][0]; toRow = row + KNIGHTS[d][1];
    } else {
        int pl = plane - 64;
        toRow = row - 1;
        toCol = (pl % 3 == 1) ? col + 1 : ((pl % 3 == 2) ? col - 1 : col);
        promo = (pl / 3 == 0) ? OC_KNIGHT : ((pl / 3 == 1) ? OC_BISHOP : OC_ROOK);
    }
    if (toRow < 0 || toRow > 7 || toCol < 0 || toCol > 7) return -1;
    int from, to;
    if (color == OC_WHITE) { from = (7 - row) * 8 + col; to = (7 - toRow) * 8 + toCol; }
    else                   { from = row * 8 + (7 - col); to = toRow * 8 + (7 - toCol); }
    if (may_be_queen_promo)
        for (int i = 0; i < n_qp; i++) if (qp[i].from == from && qp[i].to == to) promo = OC_QUEEN;
    out->from = (int8_t)from; out->to = (int8_t)to; out->promo = (int8_t)promo;
    return 0;
}

/* actionsToTensor(list(board.legal_moves), color) reduced to its support, ascending */
int oc_legal_action_indices(const oc_board *b, int color, int *idx_out, oc_move *moves_sorted) {
    oc_move mv[OC_MAX_MOVES];
    int n = oc_legal_moves(b, mv);
    int idx[OC_MAX_MOVES];
    for (int i = 0; i < n; i++) idx[i] = oc_action_to_index(mv[i], color);
    for (int i = 1; i < n; i++) {                           /* insertion sort by index */
        int k = idx[i]; oc_move m = mv[i]; int j = i - 1;
        while (j >= 0 && idx[j] > k) { idx[j + 1] = idx[j]; mv[j + 1] = mv[j]; j--; }
        idx[j + 1] = k; mv[j + 1] = m;
    }
    for (int i = 0; i < n; i++) { idx_out[i] = idx[i]; if (moves_sorted) moves_sorted[i] = mv[i]; }
    return n;
}
