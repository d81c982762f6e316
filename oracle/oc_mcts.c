/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see oc_mcts.h).
 * Restatement of /root/reference/mctsnode.py and /root/reference/mcts.py.
 * Build with -ffp-contract=off: every float operation below is one IEEE rounding,
 * in the order torch performs it in the reference.
 */
#include "oc_mcts.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------ Node.get_ucb
 * mctsnode.py:33-37 under torch type promotion (verified bitwise against the reference by
 * tests/golden/ucb_vectors.npz):
 *   vc    : int64 tensor        vsum, prior : float32 tensors
 *   q     = 1 - (vsum / (vc + 1e-6) + 1) / 2            all float32
 *   term  = C * (sqrt(N_parent) / (vc + 1)) * prior
 *         = ((reciprocal(f32(vc+1)) * f32(sqrt_f64(N))) * f32(C)) * prior
 */
float oc_ucb(long vc, float vsum, float prior, long parent_visits, double C) {
    volatile float t1 = (float)vc + (float)1e-6;
    volatile float a = vsum / t1;
    volatile float b = a + 1.0f;
    volatile float c = b / 2.0f;
    volatile float q = 1.0f - c;
    volatile float r = 1.0f / (float)(vc + 1);
    volatile float s = (float)sqrt((double)parent_visits);
    volatile float u = r * s;
    volatile float u2 = u * (float)C;
    volatile float u3 = u2 * prior;
    volatile float out = q + u3;
    return out;
}

/* mctsnode.py:23-31 select: torch.argmax -> first maximal element */
int oc_select_child(const oc_node *n, double C) {
    int best = 0; float bestv = 0;
    for (int i = 0; i < n->n_children; i++) {
        const oc_node *ch = n->children[i];
        float v = oc_ucb(ch->visit_count, (float)ch->value_sum, ch->prior, n->visit_count, C);
        if (i == 0 || v > bestv) { best = i; bestv = v; }
    }
    return best;
}

static oc_node *node_new(oc_node *parent, oc_move action, int action_index, float prior, int color) {
    oc_node *n = (oc_node *)calloc(1, sizeof *n);
    n->parent = parent; n->action_taken = action; n->action_index = action_index;
    n->prior = prior; n->color = color;
    return n;
}
static void node_free(oc_node *n, const oc_game_vt *vt) {
    for (int i = 0; i < n->n_children; i++) node_free(n->children[i], vt);
    free(n->children);
    if (n->game && n->owns_game) vt->release(n->game);
    free(n);
}

/* mctsnode.py:56-63 */
static void backpropagate(oc_node *n, double value) {
    while (n) {
        n->value_sum += value;
        n->visit_count += 1;
        value = -value;                               /* game.get_opponent_value */
        n = n->parent;
    }
}

/* mcts.py:77-99 in the oracle's fixed summation order (see header note) */
int oc_priors_from_policy(const float *policy, const int *idx, int k, int learning, float noise_value,
                          float *priors_out, int *kept_pos) {
    volatile float lane[64];
    for (int l = 0; l < 64; l++) lane[l] = 0.0f;
    for (int i = 0; i < k; i++) {                     /* ascending idx => ascending plane per lane */
        int l = idx[i] & 63;
        lane[l] = lane[l] + policy[idx[i]];
    }
    for (int off = 32; off >= 1; off >>= 1) {         /* XOR butterfly, all lanes end equal */
        float nxt[64];
        for (int l = 0; l < 64; l++) nxt[l] = lane[l] + lane[l ^ off];
        for (int l = 0; l < 64; l++) lane[l] = nxt[l];
    }
    float sum = lane[0];
    int kept = 0;
    for (int i = 0; i < k; i++) {
        volatile float p = policy[idx[i]] / sum;      /* policy /= torch.sum(policy) */
        if (p == 0.0f) continue;                      /* policy.nonzero(): NaN counts as non-zero */
        if (learning) {                               /* (1-eps)*probs + eps*noise, eps = 0.25 */
            volatile float x = 0.75f * p;
            volatile float y = 0.25f * noise_value;
            p = x + y;
        }
        priors_out[kept] = p;
        if (kept_pos) kept_pos[kept] = i;
        kept++;
    }
    return kept;
}

oc_search *oc_search_begin(void *root_game, const oc_game_vt *vt, double C, int num_searches,
                           int learning, float noise_value) {
    oc_search *s = (oc_search *)calloc(1, sizeof *s);
    s->vt = vt; s->C = C; s->num_searches = num_searches; s->learning = learning; s->noise_value = noise_value;
    oc_move none = {0, 0, 0};
    s->root = node_new(NULL, none, -1, 0.0f, vt->turn(root_game));   /* mcts.py:43 */
    s->root->game = root_game; s->root->owns_game = 0;               /* live game, not copied */
    s->root->visit_count = 1;                                        /* mcts.py:46 */
    return s;
}

int oc_search_advance(oc_search *s) {
    while (s->sims_done < s->num_searches) {                         /* mcts.py:49 */
        oc_node *node = s->root;
        s->trace_depth = 0;
        while (node->n_children) {                                   /* :54-55 */
            int c = oc_select_child(node, s->C);
            if (s->trace_depth < 4096) s->trace[s->trace_depth++] = c;
            node = node->children[c];
        }
        if (node->parent) {                                          /* :57-59 (redone on every visit) */
            if (node->game && node->owns_game) s->vt->release(node->game);
            node->game = s->vt->copy(node->parent->game);
            node->owns_game = 1;
            s->vt->move_piece(node->game, node->action_taken);
        }
        int value, term = s->vt->value_and_terminated(node->game, &value);   /* :64 */
        if (!term) {
            s->pend_k = s->vt->legal_actions(node->game, node->color, s->pend_idx, s->pend_moves);
            s->pending = node;
            return 1;
        }
        node->value = (double)value;                                 /* :106 */
        backpropagate(node, node->value);                            /* :109 */
        s->n_terminal_hits++;
        s->sims_done++;
    }
    return 0;
}

void oc_search_leaf_planes(oc_search *s, uint8_t *planes) { s->vt->representation(s->pending->game, planes); }
int oc_search_leaf_actions(oc_search *s, int *idx) { memcpy(idx, s->pend_idx, sizeof(int) * s->pend_k); return s->pend_k; }
void *oc_search_pending_game(oc_search *s) { return s->pending ? s->pending->game : NULL; }
int oc_search_trace(const oc_search *s, int *out) { memcpy(out, s->trace, sizeof(int) * s->trace_depth); return s->trace_depth; }

void oc_search_feed(oc_search *s, const float *policy, float value) {
    oc_node *node = s->pending;
    float priors[OC_MAX_MOVES]; int pos[OC_MAX_MOVES];
    int kept = oc_priors_from_policy(policy, s->pend_idx, s->pend_k, s->learning, s->noise_value, priors, pos);
    node->value = (double)value;                                     /* :85 value.item() */
    node->children = (oc_node **)malloc(sizeof(oc_node *) * (kept ? kept : 1));   /* :102 expand */
    node->n_children = kept;
    for (int i = 0; i < kept; i++)
        node->children[i] = node_new(node, s->pend_moves[pos[i]], s->pend_idx[pos[i]], priors[i], !node->color);
    backpropagate(node, node->value);                                /* :109 */
    s->pending = NULL;
    s->n_expansions++;
    s->sims_done++;
}

int oc_search_root_children(const oc_search *s, int *action_idx, long *visits, oc_move *moves) {
    for (int i = 0; i < s->root->n_children; i++) {
        if (action_idx) action_idx[i] = s->root->children[i]->action_index;
        if (visits) visits[i] = s->root->children[i]->visit_count;
        if (moves) moves[i] = s->root->children[i]->action_taken;
    }
    return s->root->n_children;
}
void oc_search_root_stats(const oc_search *s, float *priors, double *value_sums) {
    for (int i = 0; i < s->root->n_children; i++) {
        if (priors) priors[i] = s->root->children[i]->prior;
        if (value_sums) value_sums[i] = s->root->children[i]->value_sum;
    }
}
long oc_search_root_visits(const oc_search *s) { return s->root->visit_count; }
double oc_search_root_value_sum(const oc_search *s) { return s->root->value_sum; }
long oc_search_counters(const oc_search *s, int which) { return which ? s->n_terminal_hits : s->n_expansions; }

void oc_search_free(oc_search *s) { if (!s) return; node_free(s->root, s->vt); free(s); }

/* ------------------------------------------------------- chess v-table */
static void *cvt_copy(void *g) { return oc_ct_copy((oc_ct *)g); }
static void cvt_release(void *g) { oc_ct_free((oc_ct *)g); }
static int cvt_turn(void *g) { return ((oc_ct *)g)->board->cur.turn; }
static int cvt_move(void *g, oc_move m) { return oc_ct_move_piece((oc_ct *)g, m); }
static int cvt_term(void *g, int *v) { return oc_ct_get_value_and_terminated((oc_ct *)g, v); }
static int cvt_legal(void *g, int color, int *idx, oc_move *moves) { return oc_legal_action_indices(((oc_ct *)g)->board, color, idx, moves); }
static void cvt_rep(void *g, uint8_t *planes) { oc_ct_get_representation((oc_ct *)g, planes); }
const oc_game_vt OC_CHESS_VT = { cvt_copy, cvt_release, cvt_turn, cvt_move, cvt_term, cvt_legal, cvt_rep };

/* ------------------------------------------------------- table v-table */
typedef struct { const oc_table_game *tg; int state; } tstate;
void *oc_table_state_new(const oc_table_game *tg, int state) {
    tstate *t = (tstate *)malloc(sizeof *t); t->tg = tg; t->state = state; return t;
}
int oc_table_state_id(void *g) { return ((tstate *)g)->state; }
static void *tvt_copy(void *g) { tstate *t = (tstate *)malloc(sizeof *t); *t = *(tstate *)g; return t; }
static void tvt_release(void *g) { free(g); }
static int tvt_turn(void *g) { tstate *t = (tstate *)g; return t->tg->turn[t->state]; }
static int tvt_move(void *g, oc_move m) {
    tstate *t = (tstate *)g; const oc_table_game *tg = t->tg;
    for (int i = tg->move_off[t->state]; i < tg->move_off[t->state + 1]; i++)
        if (tg->move_from[i] == m.from && tg->move_to[i] == m.to && tg->move_promo[i] == m.promo) { t->state = tg->move_child[i]; return 0; }
    return -1;
}
static int tvt_term(void *g, int *v) { tstate *t = (tstate *)g; *v = t->tg->term_value[t->state]; return t->tg->terminal[t->state]; }
static int tvt_legal(void *g, int color, int *idx, oc_move *moves) {
    tstate *t = (tstate *)g; const oc_table_game *tg = t->tg;
    int n = 0;
    for (int i = tg->move_off[t->state]; i < tg->move_off[t->state + 1]; i++) {
        oc_move m = { (int8_t)tg->move_from[i], (int8_t)tg->move_to[i], (int8_t)tg->move_promo[i] };
        int k = oc_action_to_index(m, color), j = n - 1;
        while (j >= 0 && idx[j] > k) { idx[j + 1] = idx[j]; moves[j + 1] = moves[j]; j--; }
        idx[j + 1] = k; moves[j + 1] = m; n++;
    }
    return n;
}
static void tvt_rep(void *g, uint8_t *planes) { memset(planes, 0, OC_PLANES * 64); int id = ((tstate *)g)->state; memcpy(planes, &id, sizeof id); }
const oc_game_vt OC_TABLE_VT = { tvt_copy, tvt_release, tvt_turn, tvt_move, tvt_term, tvt_legal, tvt_rep };

/* ------------------------------------------------------------ sampler
 * sim.py:68  np.random.choice(keys, p=[visits/sum]) == searchsorted(cumsum(p)/cumsum(p)[-1], u, 'right')
 * with u = the single np.random.random_sample() the call consumes. */
int oc_sample_move(const long *visits, int k, double u) {
    long total = 0;
    for (int i = 0; i < k; i++) total += visits[i];
    double cdf[OC_MAX_MOVES], acc = 0.0;
    for (int i = 0; i < k; i++) { acc += (double)visits[i] / (double)total; cdf[i] = acc; }
    double last = cdf[k - 1];
    int idx = 0;
    for (int i = 0; i < k; i++) { if (cdf[i] / last <= u) idx = i + 1; }
    return idx;
}

/* DFS in child order, like tests/golden/gen_reference_fixtures.py:dump_tree */
static int dump_rec(const oc_node *n, int depth, int pos, int max, int *d, int *a, long *v, double *w, float *p) {
    for (int i = 0; i < n->n_children; i++) {
        const oc_node *c = n->children[i];
        if (pos < max) { d[pos] = depth; a[pos] = c->action_index; v[pos] = c->visit_count; w[pos] = c->value_sum; p[pos] = c->prior; }
        pos = dump_rec(c, depth + 1, pos + 1, max, d, a, v, w, p);
    }
    return pos;
}
int oc_search_dump_tree(const oc_search *s, int max, int *depth, int *action, long *visits, double *wsum, float *prior) {
    return dump_rec(s->root, 0, 0, max, depth, action, visits, wsum, prior);
}
