/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 *
 * CPU restatement of the reference's search:
 *   Node      /root/reference/mctsnode.py:7-63   (select / get_ucb / expand / backpropagate)
 *   MCTS0     /root/reference/mcts.py:24-122     (search loop, masked renormalise, noise mix)
 * with the exact fp32/fp64 operation order of Node.get_ucb (mctsnode.py:33-37 under torch
 * type promotion) and a per-game pointer tree like the reference's.
 *
 * The search core is written against a small game v-table so the same arithmetic runs
 *  (1) on real chess (oc_tensor/oc_chess), the checker for the HIP engine, and
 *  (2) on table-driven toy games replayed from tests/golden/search_*.npz, which were
 *      produced by running the reference's own mcts.py/mctsnode.py (see
 *      tests/golden/gen_reference_fixtures.py).  (2) pins this file to the reference.
 *
 * One deliberate, documented divergence: the reference normalises the masked policy with
 * torch.sum over 4672 floats (mcts.py:79), whose internal reduction order is a torch
 * implementation detail.  The oracle uses a fixed order (64 strided partial sums, XOR
 * butterfly) that the HIP kernel reproduces bit for bit; versus the reference this is a
 * <=1-ulp difference in the normaliser (tolerance 1e-4 per north_star; fixtures with
 * dyadic policies are exact in any order).
 */
#ifndef OC_MCTS_H
#define OC_MCTS_H
#include "oc_tensor.h"

typedef struct {
    void *(*copy)(void *g);                       /* copy.deepcopy(game) */
    void  (*release)(void *g);
    int   (*turn)(void *g);                       /* game.board.turn */
    int   (*move_piece)(void *g, oc_move m);      /* game.move_piece(action) ; <0 = ValueError */
    int   (*value_and_terminated)(void *g, int *value);
    /* legal moves as ascending action indices for `color` + the decoded moves (what
       actionsToTensor -> mask -> tensorToAction yields for the non-zero entries) */
    int   (*legal_actions)(void *g, int color, int *idx, oc_move *moves);
    void  (*representation)(void *g, uint8_t *planes /* 119*64 */);
} oc_game_vt;

typedef struct oc_node {
    void *game;                                    /* None until first visited */
    struct oc_node *parent;
    oc_move action_taken;
    int action_index;
    float prior;                                   /* f32-valued python float */
    int color;
    struct oc_node **children; int n_children;
    long visit_count;
    double value_sum;
    double value;
    int owns_game;
} oc_node;

typedef struct {
    const oc_game_vt *vt;
    double C; int num_searches; int learning; float noise_value;
    oc_node *root;
    int sims_done;
    oc_node *pending;                              /* leaf waiting for (policy, value) */
    int pend_idx[OC_MAX_MOVES]; oc_move pend_moves[OC_MAX_MOVES]; int pend_k;
    /* trace of the last simulation's descent: child slot chosen at each level */
    int trace_depth; int trace[4096];
    long n_expansions, n_terminal_hits;
} oc_search;

/* Node.get_ucb for one child (vsum already rounded to f32), mctsnode.py:33-37 */
float oc_ucb(long vc, float vsum, float prior, long parent_visits, double C);
/* Node.select: index of argmax (first max wins) */
int   oc_select_child(const oc_node *n, double C);

oc_search *oc_search_begin(void *root_game, const oc_game_vt *vt, double C, int num_searches,
                           int learning, float noise_value);
/* run simulations until one needs a network evaluation: 1 = pending leaf, 0 = all done */
int  oc_search_advance(oc_search *s);
void oc_search_leaf_planes(oc_search *s, uint8_t *planes);
int  oc_search_leaf_actions(oc_search *s, int *idx);          /* legal indices of the pending leaf */
/* policy = model(..., inference=True)[0] (probabilities, 4672 f32), value = model(...)[1] */
void oc_search_feed(oc_search *s, const float *policy, float value);
/* mcts.py:113-122 readout: ascending-index children of the root */
int  oc_search_root_children(const oc_search *s, int *action_idx, long *visits, oc_move *moves);
void oc_search_root_stats(const oc_search *s, float *priors, double *value_sums);
void oc_search_free(oc_search *s);

/* masked renormalise + drop zeros + optional noise mix (mcts.py:77-99) in the oracle's fixed
   summation order.  idx ascending, k legal; writes priors for the kept entries, returns kept count */
int  oc_priors_from_policy(const float *policy, const int *idx, int k, int learning, float noise_value,
                           float *priors_out, int *kept_pos);

extern const oc_game_vt OC_CHESS_VT;               /* v-table over oc_ct */

/* ---- table-driven toy game (replay of reference-generated fixtures) ---- */
typedef struct {
    int n_states;
    const int *terminal;        /* [n] 0/1 */
    const int *term_value;      /* [n] 0 or -1 */
    const int *turn;            /* [n] */
    const int *move_off;        /* [n+1] */
    const int *move_from, *move_to, *move_promo;   /* [m] */
    const int *move_child;      /* [m] state id after the move (or -1) */
} oc_table_game;
void *oc_table_state_new(const oc_table_game *tg, int state);
int   oc_table_state_id(void *g);
extern const oc_game_vt OC_TABLE_VT;

/* sampler of sim.py:68  np.random.choice(keys, p=values) given the uniform it draws */
int oc_sample_move(const long *visits, int k, double u);

#endif
