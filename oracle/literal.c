/* ORACLE No. 2 (test infrastructure only) — a LITERAL restatement of the reference's data structures.
 *
 * oracle/selfplay.c shares three storage shortcuts with the HIP engine (child.p never stored, the placeholder policy
 * implicit, (n, w) in the parent's action-indexed table, re-rooting by stable arena compaction), so a slip in one of
 * those invariants would sit on both sides of every "bit-identical" assertion.  This file restates the same reference
 * code WITHOUT any of them, shaped like the Rust:
 *
 *   Node { parent, action, children: Vec (insertion order), p, w, n: u64, state }        mcts/src/node.rs:10-21
 *   BoardState { env, status, policy: [f32; HW] (always explicit), z }                    alpha-zero/src/mcts_node.rs:7-12
 *   select_leaf / expand / propagate through parent pointers                              mcts/src/node.rs:39-99
 *   MCTS::transition: recursive free of the sibling subtrees, n = sum of children n      mcts/src/lib.rs:47-93
 *   generate_requests / scatter with the explicit `child.p = policy[action]` refresh loops
 *                                                                                         alpha-zero/src/parallel_mcts_executor.rs:44-265
 *   Agent { env, mcts }: compute_policy, sample_action, ensure_action_exists, play_action alpha-zero/src/agent.rs:16-232
 *   the trainer's four parallel vectors with swap_remove of finished games                src/trainer.rs:81-205
 *
 * It shares with selfplay.c only what the reference does not define or what its own tests pin: the rules (env.c, pinned by
 * the reference's 8 environment tests) and the build-defined RNG stream (rng.c; the reference is unseeded).  tests/
 * drives both oracles with the same (p, v) rows and requires identical canonical dumps (nodes ordered by creation stamp),
 * moves and replay tuples; tools/make_golden.py regenerates tests/golden/ through this file as well.
 */
#include "omok_oracle.h"
#include <stdlib.h>
#include <string.h>
#include <math.h>

typedef struct lnode lnode;
struct lnode {
    lnode* parent; /* Option<NodePtr<S>> */
    int action;    /* Option<usize>: -1 = None */
    lnode** children;
    int len, cap; /* RwLock<Vec<NodePtr<S>>>, Vec::with_capacity(32) */
    float p, w;
    uint64_t n;
    /* state: BoardState */
    orc_env env;
    int status;
    float policy[ORC_MAX_HW];
    float z;
    uint64_t stamp; /* creation stamp inside its tree: NOT a reference field, only orders the canonical dump */
};

typedef struct {
    orc_env env; /* Agent::env */
    lnode* root; /* Agent::mcts */
    uint64_t next_stamp;
    long live_nodes; /* allocator balance (leak check) */
} lagent;

typedef struct {
    orc_env env;
    float policy[ORC_MAX_HW];
    float z;
} ltransition;

struct lit_sp {
    int n, hw, games;
    uint64_t seed, episode, key;
    int64_t game_offset;
    int ply, error;
    /* src/trainer.rs:83-87 */
    lagent** agents_1;
    lagent** agents_2;
    int* turn_counts;
    int* transition_indices;
    int live;
    ltransition** transitions;
    int* n_transitions;
    int* cap_transitions;
    int* status; /* final GameStatus per game id */
    int* moves;  /* moves played per game id */
    /* requests of the current round */
    lnode** req_node;
    int* req_game;
    int n_req, cap_req;
    /* sample_action results of the current ply, per slot */
    int* s_action;
    float* s_policy;
    int* s_game; /* game id of row i of the mirror batch */
    int n_rows;
    double sims;
};
typedef struct lit_sp lit_sp;

/* ---- mcts crate ------------------------------------------------------------------------------ */
static lnode* node_new(lagent* ag, lnode* parent, int action, float p, const orc_env* env, int status, const float* policy, float z, int hw) {
    lnode* nd = (lnode*)calloc(1, sizeof(lnode)); /* node.rs:27-37 */
    nd->parent = parent;
    nd->action = action;
    nd->cap = 32;
    nd->children = (lnode**)malloc(sizeof(lnode*) * (size_t)nd->cap);
    nd->p = p;
    nd->w = 0.0f;
    nd->n = 0;
    nd->env = *env;
    nd->status = status;
    memcpy(nd->policy, policy, sizeof(float) * (size_t)hw);
    nd->z = z;
    nd->stamp = ag->next_stamp++;
    ag->live_nodes++;
    return nd;
}

static void dealloc_node(lagent* ag, lnode* nd) { /* mcts/src/lib.rs:81-93 */
    for (int i = 0; i < nd->len; ++i) dealloc_node(ag, nd->children[i]);
    free(nd->children);
    free(nd);
    ag->live_nodes--;
}

static float compute_ucb_1(uint64_t parent_n, const lnode* node, float c) { /* pme.rs:277-286 */
    const uint64_t n = node->n;
    const float q_s_a = node->w / ((float)n + ORC_EPS);
    const float p_s_a = node->p;
    const float bias = sqrtf((float)parent_n) / (float)(1 + n);
    return q_s_a + c * p_s_a * bias;
}

static int32_t total_cmp_key(float f) { /* f32::total_cmp */
    int32_t b;
    memcpy(&b, &f, 4);
    b ^= (int32_t)(((uint32_t)(b >> 31)) >> 1);
    return b;
}

static lnode* select_leaf(lnode* root) { /* node.rs:39-59 with the selector closure of pme.rs:81-90 */
    lnode* node = root;
    for (;;) {
        if (node->len != (int)node->env.legal) return node; /* children.len() != available_actions_len() */
        if (node->len == 0) return node;
        const uint64_t parent_n = node->n > 1 ? node->n : 1;
        int index = 0;
        float best = 0.0f;
        for (int i = 0; i < node->len; ++i) { /* enumerate().max_by(total_cmp): the LAST maximum */
            const float s = compute_ucb_1(parent_n, node->children[i], 1.0f);
            if (i == 0 || total_cmp_key(s) >= total_cmp_key(best)) { best = s; index = i; }
        }
        node = node->children[index];
    }
}

static lnode* expand(lagent* ag, lnode* self, int action, const orc_env* env, int status, const float* policy, float z, int hw) { /* node.rs:61-81 */
    for (int i = 0; i < self->len; ++i)
        if (self->children[i]->action == action) return NULL;
    lnode* child = node_new(ag, self, action, self->policy[action], env, status, policy, z, hw);
    if (self->len == self->cap) {
        self->cap *= 2;
        self->children = (lnode**)realloc(self->children, sizeof(lnode*) * (size_t)self->cap);
    }
    self->children[self->len++] = child;
    return child;
}

static void propagate(lnode* self, float w) { /* node.rs:83-99 */
    lnode* node = self;
    for (;;) {
        node->n += 1;
        node->w += w;
        w = -w;
        if (node->parent) node = node->parent;
        else break;
    }
}

static void transition(lagent* ag, int children_index) { /* mcts/src/lib.rs:47-78 */
    lnode* root = ag->root;
    for (int index = 0; index < root->len; ++index) {
        if (index == children_index) continue;
        dealloc_node(ag, root->children[index]);
    }
    lnode* new_root = root->children[children_index];
    new_root->parent = NULL;
    uint64_t new_n = 0;
    for (int i = 0; i < new_root->len; ++i) new_n += new_root->children[i]->n;
    new_root->n = new_n;
    free(root->children); /* allocator.deallocate(self.root): the old root only */
    free(root);
    ag->live_nodes--;
    ag->root = new_root;
}

/* ---- Agent (alpha-zero/src/agent.rs) ------------------------------------------------------------ */
static lagent* agent_new(int n, const float* root_policy) { /* agent.rs:16-35 */
    lagent* ag = (lagent*)calloc(1, sizeof(lagent));
    orc_env_init(&ag->env, n);
    ag->root = node_new(ag, NULL, -1, 1.0f, &ag->env, ORC_IN_PROGRESS, root_policy, 0.0f, n * n); /* MCTS::new: p = 1 */
    return ag;
}

static void agent_free(lagent* ag) {
    if (!ag) return;
    dealloc_node(ag, ag->root);
    free(ag);
}

static int compute_policy(const lagent* ag, float* policy, int hw) { /* agent.rs:43-77; 0 = None */
    const lnode* root = ag->root;
    float sum = 0.0f;
    for (int a = 0; a < hw; ++a) policy[a] = 0.0f;
    if (root->len == 0) return 0;
    for (int i = 0; i < root->len; ++i) {
        const lnode* child = root->children[i];
        const float n = (float)child->n;
        sum += n;
        policy[child->action] = n;
    }
    if (sum < ORC_EPS) return 0;
    const float sum_inv = 1.0f / sum;
    for (int a = 0; a < hw; ++a) policy[a] *= sum_inv;
    return 1;
}

static void ensure_action_exists(lagent* ag, int action, const float* p, int hw) { /* agent.rs:144-197 (p = evaluate_p result) */
    if (hw <= action) return;
    orc_env env = ag->env;
    orc_env_place_stone(&env, action);
    float policy[ORC_MAX_HW];
    memcpy(policy, p, sizeof(float) * (size_t)hw);
    policy[action] = 0.0f;
    for (int a = 0; a < hw; ++a)
        if (ag->root->env.board[a] != ORC_EMPTY) policy[a] = 0.0f; /* !root.state.is_available_action(a) */
    float sum = 0.0f;
    for (int a = 0; a < hw; ++a) sum += policy[a];
    if (ORC_EPS <= sum) {
        const float sum_inv = 1.0f / sum;
        for (int a = 0; a < hw; ++a) policy[a] *= sum_inv;
    }
    (void)expand(ag, ag->root, action, &env, ORC_IN_PROGRESS, policy, 0.0f, hw);
}

static int play_action(lagent* ag, int action) { /* agent.rs:206-232; -1 = None */
    if (ag->root->status != ORC_IN_PROGRESS) return -1;
    int children_index = -1;
    for (int i = 0; i < ag->root->len; ++i)
        if (ag->root->children[i]->action == action) { children_index = i; break; }
    if (children_index < 0) return -1;
    const int status = orc_env_place_stone(&ag->env, action);
    if (status < 0) return -1;
    transition(ag, children_index);
    return status;
}

/* ---- ParallelMCTSExecutor::execute, one round (pme.rs:44-192) for one agent ---------------------- */
/* Dirichlet noise on the root (pme.rs:48-76; the same statements open MCTSExecutor::run, mcts_executor.rs:38-68) */
static void root_noise(lit_sp* sp, lagent* ag, uint32_t tree_global, float epsilon, float alpha) {
    const int hw = sp->hw;
    float noise[ORC_MAX_HW];
    float total = 0.0f;
    for (int a = 0; a < hw; ++a) {
        noise[a] = orc_gamma(alpha, sp->key, (uint32_t)a, (uint32_t)sp->ply, tree_global);
        total += noise[a];
    }
    if (total > 0.0f) {
        const float inv = 1.0f / total;
        for (int a = 0; a < hw; ++a) noise[a] *= inv;
    } else {
        for (int a = 0; a < hw; ++a) noise[a] = 1.0f / (float)hw;
    }
    float* policy = ag->root->policy;
    for (int a = 0; a < hw; ++a) policy[a] = (1.0f - epsilon) * policy[a] + epsilon * noise[a];
    float sum = 0.0f;
    for (int a = 0; a < hw; ++a) sum += policy[a];
    const float sum_inv = 1.0f / sum;
    for (int a = 0; a < hw; ++a) policy[a] *= sum_inv;
    for (int i = 0; i < ag->root->len; ++i) { /* update children's prior probability */
        lnode* child = ag->root->children[i];
        child->p = policy[child->action];
    }
}

/* one iteration of the simulation loop (pme.rs:80-189 == mcts_executor.rs:83-192); returns the node to evaluate or NULL */
static lnode* one_simulation(lit_sp* sp, lagent* ag, uint32_t sim_index, uint32_t tree_global) {
    const int hw = sp->hw;
    sp->sims += 1;
    lnode* node = select_leaf(ag->root);
    if (node->status != ORC_IN_PROGRESS) { /* :92-97 */
        propagate(node, node->z);
        return NULL;
    }
    uint8_t bits[ORC_MAX_HW];
    memset(bits, 0, sizeof(bits));
    for (int i = 0; i < node->len; ++i) bits[node->children[i]->action] = 1;
    int available[ORC_MAX_HW], n_available = 0;
    for (int a = 0; a < hw; ++a)
        if (node->env.board[a] == ORC_EMPTY && !bits[a]) available[n_available++] = a;
    if (n_available == 0) return NULL; /* "There's no action for now." */
    uint32_t o[4];
    orc_philox(sp->key, sim_index, (uint32_t)sp->ply, tree_global, ORC_RNG_EXPAND, o);
    const int action = available[(uint32_t)(((uint64_t)o[0] * (uint64_t)n_available) >> 32)];
    orc_env env = node->env;
    const int status = orc_env_place_stone(&env, action);
    const int has_reward = status != ORC_IN_PROGRESS;
    const float terminal_reward = status == ORC_DRAW ? 0.0f : 1.0f;
    float policy[ORC_MAX_HW]; /* the uniform placeholder, :140-156 */
    for (int a = 0; a < hw; ++a) policy[a] = env.board[a] != ORC_EMPTY ? 0.0f : 1.0f;
    float sum = 0.0f;
    for (int a = 0; a < hw; ++a) sum += policy[a];
    if (ORC_EPS <= sum) {
        const float sum_inv = 1.0f / sum;
        for (int a = 0; a < hw; ++a) policy[a] *= sum_inv;
    }
    lnode* expanded_child = expand(ag, node, action, &env, status, policy, has_reward ? terminal_reward : 0.0f, hw);
    if (!expanded_child) return NULL;
    if (has_reward) {
        propagate(expanded_child, terminal_reward);
        return NULL;
    }
    return expanded_child;
}

static void generate_requests(lit_sp* sp, lagent* ag, int game, int side, int round, int batch_size, float epsilon, float alpha) {
    const uint32_t tree_global = (uint32_t)((sp->game_offset + game) * 2 + side);
    if (round == 0) root_noise(sp, ag, tree_global, epsilon, alpha); /* processed_count == 0 (:48-76) */
    for (int it = 0; it < batch_size; ++it) {
        lnode* child = one_simulation(sp, ag, (uint32_t)(round * batch_size + it), tree_global);
        if (!child) continue;
        if (sp->n_req < sp->cap_req) {
            sp->req_node[sp->n_req] = child;
            sp->req_game[sp->n_req] = game;
            sp->n_req++;
        } else {
            sp->error = 2;
        }
    }
}

/* ---- the driver object ------------------------------------------------------------------------------ */
lit_sp* lit_create(int n, int games, uint64_t seed, int64_t game_offset) {
    if (n < 5 || n > ORC_MAX_N || games < 1) return NULL;
    lit_sp* sp = (lit_sp*)calloc(1, sizeof(lit_sp));
    sp->n = n;
    sp->hw = n * n;
    sp->games = games;
    sp->seed = seed;
    sp->game_offset = game_offset;
    sp->agents_1 = (lagent**)calloc((size_t)games, sizeof(lagent*));
    sp->agents_2 = (lagent**)calloc((size_t)games, sizeof(lagent*));
    sp->turn_counts = (int*)calloc((size_t)games, sizeof(int));
    sp->transition_indices = (int*)calloc((size_t)games, sizeof(int));
    sp->transitions = (ltransition**)calloc((size_t)games, sizeof(ltransition*));
    sp->n_transitions = (int*)calloc((size_t)games, sizeof(int));
    sp->cap_transitions = (int*)calloc((size_t)games, sizeof(int));
    sp->status = (int*)calloc((size_t)games, sizeof(int));
    sp->moves = (int*)calloc((size_t)games, sizeof(int));
    sp->cap_req = games * 64;
    sp->req_node = (lnode**)calloc((size_t)sp->cap_req, sizeof(lnode*));
    sp->req_game = (int*)calloc((size_t)sp->cap_req, sizeof(int));
    sp->s_action = (int*)calloc((size_t)games, sizeof(int));
    sp->s_policy = (float*)calloc((size_t)games * (size_t)sp->hw, sizeof(float));
    sp->s_game = (int*)calloc((size_t)games, sizeof(int));
    return sp;
}

static void drop_agents(lit_sp* sp) {
    for (int i = 0; i < sp->live; ++i) {
        agent_free(sp->agents_1[i]);
        agent_free(sp->agents_2[i]);
        sp->agents_1[i] = sp->agents_2[i] = NULL;
    }
    sp->live = 0;
}

void lit_destroy(lit_sp* sp) {
    if (!sp) return;
    drop_agents(sp);
    for (int g = 0; g < sp->games; ++g) free(sp->transitions[g]);
    free(sp->agents_1); free(sp->agents_2); free(sp->turn_counts); free(sp->transition_indices);
    free(sp->transitions); free(sp->n_transitions); free(sp->cap_transitions); free(sp->status); free(sp->moves);
    free(sp->req_node); free(sp->req_game); free(sp->s_action); free(sp->s_policy); free(sp->s_game);
    free(sp);
}

void lit_set_episode(lit_sp* sp, uint64_t episode) { sp->episode = episode; }

void lit_reset(lit_sp* sp, const float* root_policy) { /* trainer.rs:79-93 */
    drop_agents(sp);
    sp->key = orc_stream_key(sp->seed, sp->episode);
    sp->episode += 1;
    sp->ply = 0;
    sp->error = 0;
    sp->n_req = 0;
    sp->n_rows = 0;
    sp->sims = 0;
    for (int g = 0; g < sp->games; ++g) {
        sp->agents_1[g] = agent_new(sp->n, root_policy);
        sp->agents_2[g] = agent_new(sp->n, root_policy);
        sp->turn_counts[g] = 0;
        sp->transition_indices[g] = g;
        sp->n_transitions[g] = 0;
        sp->status[g] = ORC_IN_PROGRESS;
        sp->moves[g] = 0;
    }
    sp->live = sp->games;
}

int lit_ply(const lit_sp* sp) { return sp->ply; }
int lit_error(const lit_sp* sp) { return sp->error; }
int lit_alive_count(const lit_sp* sp) { return sp->live; }
int lit_game_status(const lit_sp* sp, int game) { return sp->status[game]; }
int lit_game_plies(const lit_sp* sp, int game) { return sp->moves[game]; }
int lit_game_alive(const lit_sp* sp, int game) {
    for (int i = 0; i < sp->live; ++i)
        if (sp->transition_indices[i] == game) return 1;
    return 0;
}
long lit_live_nodes(const lit_sp* sp) { /* allocator balance over all live agents */
    long t = 0;
    for (int i = 0; i < sp->live; ++i) t += sp->agents_1[i]->live_nodes + sp->agents_2[i]->live_nodes;
    return t;
}

static lagent** side_agents(lit_sp* sp, int side) { return side == 0 ? sp->agents_1 : sp->agents_2; }

/* One round over the agents of the side to move, in SLOT order (the order of the reference's vectors after its
 * swap_removes).  Request r belongs to game *req_game[r]; within a game requests are in simulation order. */
int lit_round_generate(lit_sp* sp, int round, int batch_size, float epsilon, float alpha, float* inputs, int32_t* req_games, int max_req) {
    const int side = sp->ply & 1; /* agents_1[0].env.turn: Black moves on even plies (trainer.rs:96-97) */
    lagent** agents = side_agents(sp, side);
    sp->n_req = 0;
    for (int i = 0; i < sp->live; ++i) generate_requests(sp, agents[i], sp->transition_indices[i], side, round, batch_size, epsilon, alpha);
    if (sp->n_req > max_req) { sp->error = 3; return -1; }
    for (int r = 0; r < sp->n_req; ++r) {
        if (inputs) orc_encode_nn_input(&sp->req_node[r]->env, ORC_MODE_PLAYER, inputs + (size_t)r * 3 * (size_t)sp->hw);
        if (req_games) req_games[r] = sp->req_game[r];
    }
    return sp->n_req;
}

void lit_round_scatter(lit_sp* sp, const float* p, const float* v) { /* pme.rs:222-265, rows in THIS object's request order */
    const int hw = sp->hw;
    for (int r = 0; r < sp->n_req; ++r) {
        lnode* node = sp->req_node[r];
        const float* raw_policy = p + (size_t)r * (size_t)hw;
        const float value = -v[r];
        float policy[ORC_MAX_HW];
        memcpy(policy, raw_policy, sizeof(float) * (size_t)hw);
        for (int a = 0; a < hw; ++a)
            if (node->env.board[a] != ORC_EMPTY) policy[a] = 0.0f;
        float sum = 0.0f;
        for (int a = 0; a < hw; ++a) sum += policy[a];
        if (ORC_EPS <= sum) {
            const float sum_inv = 1.0f / sum;
            for (int a = 0; a < hw; ++a) policy[a] *= sum_inv;
        }
        memcpy(node->policy, policy, sizeof(float) * (size_t)hw);
        for (int i = 0; i < node->len; ++i) {
            lnode* child = node->children[i];
            child->p = policy[child->action];
        }
        propagate(node, value);
    }
    sp->n_req = 0;
}

/* ---- MCTSExecutor::run (alpha-zero/src/mcts_executor.rs:29-255) under a GIVEN interleaving ----------------------------------------
 * The reference runs its ceil(count / batch_size) rounds as rayon tasks on one tree; which task's simulation touches the tree next is up
 * to the thread pool.  These functions run game 0's side-to-move agent with the tasks' simulations interleaved in a caller-given order
 * (the lock order an engine recorded): `order[i]` = index w of the task (round group * waves + w) whose next simulation runs i-th.  Every
 * simulation is the loop body of :83-192 executed alone, i.e. the schedule in which the pool never overlaps two simulations.  The
 * requests of the group's tasks are evaluated as one batch, rows in task order then simulation order; the scatter (:206-250) writes all
 * policies, then runs the propagate calls in `order` of the second list (a request's policy is only read by later descents, so policy
 * writes and backups commute while no descent runs). */
void lit_shared_noise(lit_sp* sp, float epsilon, float alpha) { /* mcts_executor.rs:38-68 */
    if (sp->live < 1) return;
    const int side = sp->ply & 1;
    lagent* ag = side_agents(sp, side)[0];
    root_noise(sp, ag, (uint32_t)((sp->game_offset + sp->transition_indices[0]) * 2 + side), epsilon, alpha);
}

int lit_shared_group_generate(lit_sp* sp, int group, int waves, int rounds_total, int batch_size, const uint8_t* order, int n_order,
                              float* inputs, int max_req) {
    if (sp->live < 1 || waves < 1 || waves > 64) return -1;
    if (waves * batch_size > sp->cap_req) { /* (the request list was sized for one round per game) */
        sp->cap_req = waves * batch_size;
        sp->req_node = (lnode**)realloc(sp->req_node, (size_t)sp->cap_req * sizeof(lnode*));
        sp->req_game = (int*)realloc(sp->req_game, (size_t)sp->cap_req * sizeof(int));
    }
    const int side = sp->ply & 1, game = sp->transition_indices[0];
    lagent* ag = side_agents(sp, side)[0];
    const uint32_t tree_global = (uint32_t)((sp->game_offset + game) * 2 + side);
    int done[64];
    memset(done, 0, sizeof(done));
    lnode** per = (lnode**)calloc((size_t)waves * (size_t)batch_size, sizeof(lnode*));
    int n_per[64];
    memset(n_per, 0, sizeof(n_per));
    for (int i = 0; i < n_order; ++i) {
        const int w = order[i];
        const int round = group * waves + w;
        if (w >= waves || round >= rounds_total || done[w] >= batch_size) { sp->error = 4; free(per); return -1; }
        lnode* child = one_simulation(sp, ag, (uint32_t)(round * batch_size + done[w]), tree_global);
        done[w] += 1;
        if (child) per[(size_t)w * (size_t)batch_size + (size_t)n_per[w]++] = child;
    }
    for (int w = 0; w < waves; ++w) /* every live task runs all its simulations (:83) */
        if (group * waves + w < rounds_total && done[w] != batch_size) { sp->error = 4; free(per); return -1; }
    sp->n_req = 0;
    for (int w = 0; w < waves; ++w)
        for (int r = 0; r < n_per[w]; ++r) {
            if (sp->n_req >= sp->cap_req || sp->n_req >= max_req) { sp->error = 3; free(per); return -1; }
            sp->req_node[sp->n_req] = per[(size_t)w * (size_t)batch_size + (size_t)r];
            sp->req_game[sp->n_req] = w; /* (the task the request belongs to) */
            if (inputs) orc_encode_nn_input(&sp->req_node[sp->n_req]->env, ORC_MODE_PLAYER, inputs + (size_t)sp->n_req * 3 * (size_t)sp->hw);
            sp->n_req++;
        }
    free(per);
    return sp->n_req;
}

void lit_shared_group_scatter(lit_sp* sp, const float* p, const float* v, const uint8_t* order, int n_order) {
    const int hw = sp->hw;
    if (n_order != sp->n_req) { sp->error = 4; return; }
    for (int r = 0; r < sp->n_req; ++r) { /* :212-243 */
        lnode* node = sp->req_node[r];
        float policy[ORC_MAX_HW];
        memcpy(policy, p + (size_t)r * (size_t)hw, sizeof(float) * (size_t)hw);
        for (int a = 0; a < hw; ++a)
            if (node->env.board[a] != ORC_EMPTY) policy[a] = 0.0f;
        float sum = 0.0f;
        for (int a = 0; a < hw; ++a) sum += policy[a];
        if (ORC_EPS <= sum) {
            const float sum_inv = 1.0f / sum;
            for (int a = 0; a < hw; ++a) policy[a] *= sum_inv;
        }
        memcpy(node->policy, policy, sizeof(float) * (size_t)hw);
        for (int i = 0; i < node->len; ++i) node->children[i]->p = policy[node->children[i]->action];
    }
    int next[64], first[64];
    memset(next, 0, sizeof(next));
    for (int w = 0; w < 64; ++w) first[w] = -1;
    for (int r = 0; r < sp->n_req; ++r)
        if (first[sp->req_game[r]] < 0) first[sp->req_game[r]] = r;
    for (int i = 0; i < n_order; ++i) { /* :246-247 in the recorded order */
        const int w = order[i];
        if (w >= 64 || first[w] < 0) { sp->error = 4; return; }
        const int r = first[w] + next[w]++;
        if (r >= sp->n_req || sp->req_game[r] != w) { sp->error = 4; return; }
        propagate(sp->req_node[r], -v[r]);
    }
    sp->n_req = 0;
}

/* sample_action for every slot (trainer.rs:138-146 mode rule); actions are reported by GAME ID (-1: finished) */
void lit_sample(lit_sp* sp, float temperature, int threshold, int32_t* actions) {
    const int side = sp->ply & 1, hw = sp->hw;
    lagent** agents = side_agents(sp, side);
    for (int g = 0; g < sp->games; ++g) actions[g] = -1;
    sp->n_rows = sp->live;
    for (int i = 0; i < sp->live; ++i) {
        const int game = sp->transition_indices[i];
        float* policy = sp->s_policy + (size_t)i * (size_t)hw;
        sp->s_game[i] = game;
        sp->s_action[i] = -1;
        if (!compute_policy(agents[i], policy, hw)) { sp->error = 4; continue; }
        int action = 0;
        if (sp->turn_counts[i] < threshold) { /* Boltzmann (agent.rs:106-133), categorical draw by the build's stream */
            float heated_policy[ORC_MAX_HW];
            float sum = 0.0f;
            const float temperature_inv = 1.0f / temperature;
            for (int a = 0; a < hw; ++a) {
                heated_policy[a] = 0.0f;
                const float prob = policy[a];
                if (prob < ORC_EPS) continue;
                const float heated = orc_det_expf(policy[a] * temperature_inv);
                sum += heated;
                heated_policy[a] = heated;
            }
            const float sum_inv = 1.0f / sum;
            for (int a = 0; a < hw; ++a) heated_policy[a] *= sum_inv;
            float total = 0.0f; /* WeightedIndex::new: total weight */
            for (int a = 0; a < hw; ++a) total += heated_policy[a];
            uint32_t o[4];
            orc_philox(sp->key, 0, (uint32_t)sp->ply, (uint32_t)((sp->game_offset + game) * 2 + side), ORC_RNG_SAMPLE, o);
            const float u = (float)(o[0] >> 8) * 5.9604644775390625e-8f;
            const float target = u * total;
            float cum = 0.0f;
            int chosen = -1, last_nz = 0;
            for (int a = 0; a < hw; ++a) {
                if (!(heated_policy[a] > 0.0f)) continue;
                last_nz = a;
                cum += heated_policy[a];
                if (chosen < 0 && cum > target) chosen = a;
            }
            action = chosen < 0 ? last_nz : chosen;
        } else { /* Best (agent.rs:98-105): last maximum */
            for (int a = 1; a < hw; ++a)
                if (total_cmp_key(policy[a]) >= total_cmp_key(policy[action])) action = a;
        }
        sp->s_action[i] = action;
        actions[game] = action;
    }
}

/* NN inputs of ensure_action_exists (agent.rs:153-158) for the moves chosen by lit_sample / given to lit_set_actions, in
 * slot order; row_games[i] = game id of row i */
int lit_mirror_generate(lit_sp* sp, float* inputs, int32_t* row_games, int max_req) {
    const int side = sp->ply & 1;
    lagent** agents = side_agents(sp, side);
    if (sp->n_rows > max_req) { sp->error = 3; return -1; }
    for (int i = 0; i < sp->n_rows; ++i) {
        orc_env env = agents[i]->env;
        if (sp->s_action[i] >= 0) orc_env_place_stone(&env, sp->s_action[i]);
        if (inputs) orc_encode_nn_input(&env, ORC_MODE_OPPONENT, inputs + (size_t)i * 3 * (size_t)sp->hw);
        if (row_games) row_games[i] = sp->s_game[i];
    }
    return sp->n_rows;
}

/* externally chosen moves (gui/src/agent.rs, benchmark/src/agent.rs style): replaces lit_sample; no transition is recorded */
void lit_set_actions(lit_sp* sp, const int32_t* actions) {
    sp->n_rows = sp->live;
    for (int i = 0; i < sp->live; ++i) {
        sp->s_game[i] = sp->transition_indices[i];
        sp->s_action[i] = actions[sp->transition_indices[i]];
    }
}

static void push_transition(lit_sp* sp, int game, const orc_env* env, const float* policy, float z) {
    if (sp->n_transitions[game] == sp->cap_transitions[game]) {
        sp->cap_transitions[game] = sp->cap_transitions[game] ? 2 * sp->cap_transitions[game] : 64;
        sp->transitions[game] = (ltransition*)realloc(sp->transitions[game], sizeof(ltransition) * (size_t)sp->cap_transitions[game]);
    }
    ltransition* t = &sp->transitions[game][sp->n_transitions[game]++];
    t->env = *env;
    memcpy(t->policy, policy, sizeof(float) * (size_t)sp->hw);
    t->z = z;
}

/* the per-game loop of trainer.rs:131-204; p row i = evaluate_p of row i of lit_mirror_generate.
 * external != 0: the moves came from lit_set_actions: ensure_action_exists + play_action on BOTH agents, no transition. */
void lit_advance(lit_sp* sp, const float* p, int external) {
    const int side = sp->ply & 1, hw = sp->hw;
    lagent** agents = side_agents(sp, side);
    lagent** opposite_agents = side_agents(sp, 1 - side);
    int* row_of_game = (int*)malloc(sizeof(int) * (size_t)sp->games);
    for (int i = 0; i < sp->n_rows; ++i) row_of_game[sp->s_game[i]] = i;
    int index = 0;
    while (index < sp->live) {
        lagent* agent = agents[index];
        lagent* opposite_agent = opposite_agents[index];
        const int game = sp->transition_indices[index];
        const int row = row_of_game[game];
        const int action = sp->s_action[row];
        const float* policy = sp->s_policy + (size_t)row * (size_t)hw;
        const float* p_row = p + (size_t)row * (size_t)hw;
        if (action < 0) { sp->error = 5; index += 1; continue; }
        if (!external) sp->turn_counts[index] += 1;
        const orc_env env_before_action = agent->env;
        if (external) ensure_action_exists(agent, action, p_row, hw);
        const int status = play_action(agent, action);
        if (status < 0) { sp->error = 5; index += 1; continue; }
        const float z = (status == ORC_BLACK_WIN || status == ORC_WHITE_WIN) ? 1.0f : 0.0f;
        const int is_terminal = status != ORC_IN_PROGRESS;
        ensure_action_exists(opposite_agent, action, p_row, hw);
        if (play_action(opposite_agent, action) < 0) sp->error = 6;
        if (!external) push_transition(sp, game, &env_before_action, policy, z);
        sp->moves[game] += 1;
        sp->status[game] = status;
        if (is_terminal) { /* swap_remove on the four vectors (trainer.rs:187-190) */
            agent_free(agent);
            agent_free(opposite_agent);
            const int last = sp->live - 1;
            agents[index] = agents[last];
            opposite_agents[index] = opposite_agents[last];
            agents[last] = opposite_agents[last] = NULL;
            sp->turn_counts[index] = sp->turn_counts[last];
            sp->transition_indices[index] = sp->transition_indices[last];
            sp->live = last;
            continue;
        }
        index += 1;
    }
    free(row_of_game);
    sp->n_rows = 0;
    sp->ply += 1;
}

/* Agent::compute_policy of the side-to-move agent of one game; returns 0 for None */
int lit_compute_policy(const lit_sp* sp, int game, float* policy) {
    const int side = sp->ply & 1;
    for (int i = 0; i < sp->live; ++i)
        if (sp->transition_indices[i] == game) return compute_policy((side == 0 ? sp->agents_1 : sp->agents_2)[i], policy, sp->hw);
    return 0;
}

/* ---- canonical dump: nodes in creation order (the stamp), the format of orc_sp_tree_dump -------- */
static int count_nodes(const lnode* nd) {
    int c = 1;
    for (int i = 0; i < nd->len; ++i) c += count_nodes(nd->children[i]);
    return c;
}
static void collect(const lnode* nd, const lnode** out, int* k) {
    out[(*k)++] = nd;
    for (int i = 0; i < nd->len; ++i) collect(nd->children[i], out, k);
}
static int by_stamp(const void* a, const void* b) {
    const lnode* x = *(const lnode* const*)a;
    const lnode* y = *(const lnode* const*)b;
    return x->stamp < y->stamp ? -1 : (x->stamp > y->stamp ? 1 : 0);
}
static const lagent* find_agent(const lit_sp* sp, int game, int side) {
    for (int i = 0; i < sp->live; ++i)
        if (sp->transition_indices[i] == game) return (side == 0 ? sp->agents_1 : sp->agents_2)[i];
    return NULL;
}

/* ints [n][8] = parent, action, status, turn, legal, nch, n, insertion rank among its siblings (no has_policy bit: a
 * literal node always carries an explicit policy); floats [n][1 + HW] = w, policy.  Finished games have no agents: 0. */
int lit_tree_dump(const lit_sp* sp, int game, int side, int32_t* ints, float* floats, int cap_nodes) {
    const lagent* ag = find_agent(sp, game, side);
    if (!ag) return 0;
    const int hw = sp->hw;
    const int nn = count_nodes(ag->root);
    if (nn > cap_nodes) return -nn;
    const lnode** all = (const lnode**)malloc(sizeof(lnode*) * (size_t)nn);
    int k = 0;
    collect(ag->root, all, &k);
    qsort(all, (size_t)nn, sizeof(lnode*), by_stamp);
    for (int i = 0; i < nn; ++i) {
        const lnode* nd = all[i];
        int32_t* o = ints + (size_t)i * 8;
        float* f = floats + (size_t)i * (size_t)(1 + hw);
        int parent = -1, rank = -1;
        if (nd->parent) {
            const lnode* key = nd->parent;
            const lnode** hit = (const lnode**)bsearch(&key, all, (size_t)nn, sizeof(lnode*), by_stamp);
            parent = (int)(hit - all);
            for (int c = 0; c < nd->parent->len; ++c)
                if (nd->parent->children[c] == nd) rank = c;
        }
        o[0] = parent;
        o[1] = nd->action;
        o[2] = nd->status;
        o[3] = nd->env.turn;
        o[4] = nd->env.legal;
        o[5] = nd->len;
        o[6] = (int32_t)nd->n;
        o[7] = rank & 0xffff;
        f[0] = nd->w;
        memcpy(f + 1, nd->policy, sizeof(float) * (size_t)hw);
    }
    free(all);
    return nn;
}

/* child.p of every non-root node in dump order (the field the other oracle and the engine never store), for the test
 * of the invariant child.p == parent.policy[child.action] */
int lit_tree_priors(const lit_sp* sp, int game, int side, float* p_out, float* parent_policy_at_action, int cap_nodes) {
    const lagent* ag = find_agent(sp, game, side);
    if (!ag) return 0;
    const int nn = count_nodes(ag->root);
    if (nn > cap_nodes) return -nn;
    const lnode** all = (const lnode**)malloc(sizeof(lnode*) * (size_t)nn);
    int k = 0;
    collect(ag->root, all, &k);
    qsort(all, (size_t)nn, sizeof(lnode*), by_stamp);
    for (int i = 0; i < nn; ++i) {
        p_out[i] = all[i]->p;
        parent_policy_at_action[i] = all[i]->parent ? all[i]->parent->policy[all[i]->action] : all[i]->p; /* a root's p is never read */
    }
    free(all);
    return nn;
}

int lit_replay(const lit_sp* sp, int game, uint8_t* boards, uint8_t* turns, float* pi, float* z, int cap_plies) {
    const int hw = sp->hw;
    const int n = sp->n_transitions[game] < cap_plies ? sp->n_transitions[game] : cap_plies;
    for (int i = 0; i < n; ++i) {
        const ltransition* t = &sp->transitions[game][i];
        memcpy(boards + (size_t)i * (size_t)hw, t->env.board, (size_t)hw);
        turns[i] = t->env.turn;
        memcpy(pi + (size_t)i * (size_t)hw, t->policy, sizeof(float) * (size_t)hw);
        z[i] = t->z;
    }
    return sp->n_transitions[game];
}
