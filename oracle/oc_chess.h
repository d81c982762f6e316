/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 *
 * CPU restatement (plain C99) of the chess rules the reference obtains from the
 * third-party dependency python-chess, pinned by the reference at
 *   /root/reference/environment.yml:21   chess==1.10.0
 * The library source is NOT under /root/reference and is not installed in this
 * image, so this file restates python-chess 1.10.0's published algorithm
 * (chess/__init__.py: Board.push, generate_legal_moves, generate_castling_moves,
 * is_repetition, outcome, has_insufficient_material, set_chess960_pos ...).
 *
 * Parity status for THIS file: "parity unpinned" by the reference (it holds no
 * tests/golden vectors for chess rules); pinned instead by the
 * public perft known-answer tables in tests/test_oracle_chess.py.
 *
 * Call sites in the reference that this file serves:
 *   chess_tensor.py:53 piece_at, :69 Board.from_chess960_pos, :71 Board(),
 *   :91 `move in legal_moves`, :95 push, :101-102 is_repetition(2/3),
 *   :113 move_stack, :114-117 has_{king,queen}side_castling_rights,
 *   :118 halfmove_clock, :135 turn, :146/:158 legal_moves,
 *   :161-162 is_game_over()/outcome().winner, sim.py:46 is_game_over, sim.py:86 result().
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this code.
 */
#ifndef OC_CHESS_H
#define OC_CHESS_H
#include <stdint.h>

typedef uint64_t u64;

#define OC_WHITE 1
#define OC_BLACK 0
enum { OC_PAWN = 1, OC_KNIGHT, OC_BISHOP, OC_ROOK, OC_QUEEN, OC_KING };

typedef struct { int8_t from, to, promo; } oc_move;   /* promo: 0 or OC_KNIGHT..OC_QUEEN */

typedef struct {
    u64 bb[7];            /* by piece type 1..6 (index 0 unused) */
    u64 occ_co[2];        /* [OC_BLACK], [OC_WHITE] */
    u64 occupied;
    u64 castling_rights;  /* rook-square mask */
    int ep_square;        /* -1 = None */
    int turn;
    int halfmove_clock;
    int fullmove_number;
} oc_pos;

typedef struct {
    oc_pos cur;
    int chess960;
    int n_stack, cap;     /* len(move_stack) == len(_stack) == n_stack */
    oc_pos *stack;        /* board states before each pushed move */
    oc_move *moves;       /* move_stack (external move form) */
} oc_board;

#define OC_MAX_MOVES 256

/* construction / lifetime */
oc_board *oc_board_new(void);                         /* chess.Board() */
oc_board *oc_board_new_960(int scharnagl);            /* chess.Board.from_chess960_pos(n) */
oc_board *oc_board_from_fen(const char *fen, int chess960);
oc_board *oc_board_copy(const oc_board *b);           /* copy.deepcopy(board) */
void oc_board_free(oc_board *b);

/* rules */
int  oc_legal_moves(const oc_board *b, oc_move *out); /* list(board.legal_moves); returns count */
int  oc_is_legal(const oc_board *b, oc_move m);       /* `move in board.legal_moves` */
void oc_push(oc_board *b, oc_move m);                 /* board.push(move) (external move form) */
void oc_pop(oc_board *b);
int  oc_is_check(const oc_board *b);
int  oc_is_repetition(const oc_board *b, int count);
int  oc_has_legal_en_passant(const oc_board *b);
int  oc_has_kingside_castling_rights(const oc_board *b, int color);
int  oc_has_queenside_castling_rights(const oc_board *b, int color);
int  oc_is_insufficient_material(const oc_board *b);
/* outcome(claim_draw=False): 0 none, 1 checkmate, 2 insufficient material, 3 stalemate,
   4 seventyfive moves, 5 fivefold repetition.  *winner: 1 white, 0 black, -1 None */
int  oc_outcome(const oc_board *b, int *winner);
int  oc_piece_at(const oc_board *b, int sq, int *color); /* piece type or 0 */
u64  oc_perft(oc_board *b, int depth);
void oc_board_fen_pieces(const oc_board *b, char *out /* >= 72 bytes */);
int  oc_board_turn(const oc_board *b);
int  oc_board_halfmove_clock(const oc_board *b);
int  oc_board_ply(const oc_board *b);
int  oc_board_ep_square(const oc_board *b);
u64  oc_board_castling_rights(const oc_board *b);
int  oc_board_is_chess960(const oc_board *b);
void oc_board_bitboards(const oc_board *b, u64 *out8);

#endif
