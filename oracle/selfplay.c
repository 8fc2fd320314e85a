/* ORACLE (test infrastructure only) — tree search and self-play loop.
 *
 * Restates, for G games x two trees:
 *   mcts/src/node.rs:39-99, mcts/src/lib.rs:43-93                  (select/expand/propagate/transition)
 *   alpha-zero/src/parallel_mcts_executor.rs:44-192,222-265,277-286 (one round, scatter, PUCT)
 *   alpha-zero/src/agent.rs:43-232                                  (policy, sampling, ensure, play)
 *   src/trainer.rs:95-205                                           (ply loop)
 *
 * Storage model (shared with the HIP engine so canonical dumps compare element for element):
 *   - a tree is an arena of nodes in creation order (parent index < child index); node 0 = root;
 *   - a node's (n, w) live in its PARENT's child table, indexed by action; the root's own
 *     (n, w) are root_n/root_w;  child.p is never stored: the reference keeps
 *     child.p == parent.policy[child.action] at all times (node.rs:76, pme.rs:71-75,256-261),
 *     so PUCT reads the parent's policy row;
 *   - a node that has not been NN-evaluated yet carries the uniform placeholder of
 *     pme.rs:140-156 implicitly (has_policy = 0): value 1/legal on empty cells, 0 elsewhere;
 *   - transition() (mcts/src/lib.rs:47-78) keeps the chosen subtree by STABLE compaction
 *     (surviving nodes/tables keep their relative order).
 */
#define _POSIX_C_SOURCE 200809L
#include "omok_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define NONE16 0xFFFFu
#define NONE8 0xFFu

typedef struct {
    uint16_t parent;
    uint8_t action;
    uint8_t status;
    uint8_t turn;
    uint8_t has_policy;
    uint16_t legal;
    uint16_t nch;
    uint16_t table;
    uint8_t board[ORC_MAX_HW];
    float policy[ORC_MAX_HW];
} node_t;

typedef struct {
    uint32_t cn[ORC_MAX_HW];
    float cw[ORC_MAX_HW];
    uint16_t cidx[ORC_MAX_HW];
    uint8_t corder[ORC_MAX_HW];
    uint16_t owner;
} table_t;

typedef struct {
    node_t* nodes;
    table_t* tables;
    int n_nodes, n_tables;
    uint32_t root_n;
    float root_w;
} tree_t;

typedef struct {
    uint8_t* boards; /* [cap][HW] */
    uint8_t* turns;
    float* pi; /* [cap][HW] */
    float* z;
    int plies;
} replay_t;

struct orc_sp {
    int n, hw, games, cap_nodes, cap_tables;
    uint64_t seed, episode, key; /* key = orc_stream_key(seed, episode of the current reset) */
    int64_t game_offset;
    int ply;
    int error;
    int external; /* the pending moves came from orc_sp_set_actions */
    int threads;  /* orc_sp_set_threads: > 1 = the loops over games of round_generate / round_scatter run under OpenMP (bench.py's CPU baseline: the reference's
                     rayon par_iter over agents, pme.rs:200-205); every game then pushes its requests into its own segment [64 g, 64 g + 64) and the segments are
                     packed in game order afterwards, so the request list -- and everything else -- is the serial loop's (tests/test_oracle_selfplay.py) */
    int32_t* mt_cnt; /* [games] requests in the game's segment (threads > 1 only) */
    tree_t* trees[2];
    orc_env* envs;
    uint8_t* alive;
    uint8_t* status;
    int32_t* plies;
    int32_t* last_action;
    replay_t* replay;
    /* requests of the current round */
    int n_req, cap_req;
    int32_t* req_game;
    int32_t* req_node;
    /* scratch */
    uint8_t* alive_mark;
    uint16_t* node_map;
    uint16_t* table_map;
    double stat_sims;
};

static int32_t total_key(float f) {
    int32_t b;
    memcpy(&b, &f, 4);
    b ^= (int32_t)(((uint32_t)(b >> 31)) >> 1); /* f32::total_cmp */
    return b;
}

static float eff_policy(const node_t* nd, int a) {
    if (nd->has_policy) return nd->policy[a];
    if (nd->board[a] != ORC_EMPTY || nd->legal == 0) return 0.0f;
    return 1.0f / (float)nd->legal; /* pme.rs:140-156: 1 * sum.recip() */
}

static void tree_init(tree_t* t, const orc_env* env, const float* root_policy, int hw) {
    node_t* r = &t->nodes[0];
    memset(r, 0, sizeof(*r));
    r->parent = NONE16;
    r->action = NONE8;
    r->status = ORC_IN_PROGRESS;
    r->turn = env->turn;
    r->has_policy = 1;
    r->legal = env->legal;
    r->nch = 0;
    r->table = NONE16;
    memcpy(r->board, env->board, (size_t)hw);
    memcpy(r->policy, root_policy, sizeof(float) * (size_t)hw);
    t->n_nodes = 1;
    t->n_tables = 0;
    t->root_n = 0;    /* mcts/src/lib.rs:25-32: p=1, w=0, n=0 */
    t->root_w = 0.0f;
}

/* mcts/src/node.rs:83-99 */
static void backup(tree_t* t, int x, float v) {
    for (;;) {
        if (x == 0) {
            t->root_n += 1;
            t->root_w += v;
            return;
        }
        const node_t* nd = &t->nodes[x];
        table_t* tb = &t->tables[t->nodes[nd->parent].table];
        tb->cn[nd->action] += 1;
        tb->cw[nd->action] += v;
        v = -v;
        x = nd->parent;
    }
}

/* returns new node index or -1 on arena overflow */
static int add_child(orc_sp* sp, tree_t* t, int parent, int action, const orc_env* env, int status) {
    node_t* pn = &t->nodes[parent];
    if (t->n_nodes >= sp->cap_nodes) { sp->error = 1; return -1; }
    if (pn->table == NONE16) {
        if (t->n_tables >= sp->cap_tables) { sp->error = 1; return -1; }
        table_t* tb = &t->tables[t->n_tables];
        memset(tb->corder, NONE8, sizeof(tb->corder));
        tb->owner = (uint16_t)parent;
        pn->table = (uint16_t)t->n_tables++;
    }
    table_t* tb = &t->tables[pn->table];
    const int idx = t->n_nodes++;
    tb->corder[action] = (uint8_t)pn->nch;
    tb->cidx[action] = (uint16_t)idx;
    tb->cn[action] = 0;
    tb->cw[action] = 0.0f;
    pn->nch += 1;
    node_t* c = &t->nodes[idx];
    c->parent = (uint16_t)parent;
    c->action = (uint8_t)action;
    c->status = (uint8_t)status;
    c->turn = env->turn;
    c->has_policy = 0;
    c->legal = env->legal;
    c->nch = 0;
    c->table = NONE16;
    memcpy(c->board, env->board, (size_t)sp->hw);
    return idx;
}

/* pme.rs:48-76 */
static void apply_noise(orc_sp* sp, tree_t* t, float epsilon, float alpha, uint32_t tree_global) {
    const int hw = sp->hw;
    node_t* r = &t->nodes[0];
    float noise[ORC_MAX_HW];
    if (!r->has_policy) { /* defensive: materialise the placeholder */
        for (int a = 0; a < hw; ++a) r->policy[a] = eff_policy(r, a);
        r->has_policy = 1;
    }
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
    for (int a = 0; a < hw; ++a) r->policy[a] = (1.0f - epsilon) * r->policy[a] + epsilon * noise[a];
    float sum = 0.0f;
    for (int a = 0; a < hw; ++a) sum += r->policy[a];
    const float sum_inv = 1.0f / sum;
    for (int a = 0; a < hw; ++a) r->policy[a] *= sum_inv;
    /* children p refresh (:71-75) is implicit: PUCT reads this row */
}

/* one simulation, pme.rs:80-189 */
static void run_sim(orc_sp* sp, tree_t* t, int game, uint32_t sim_index, uint32_t tree_global) {
    const int hw = sp->hw;
    int node = 0;
    uint32_t node_n = t->root_n;
    for (;;) { /* node.rs:43-58 */
        const node_t* nd = &t->nodes[node];
        if (nd->nch != nd->legal) break;
        if (nd->nch == 0) break;
        const table_t* tb = &t->tables[nd->table];
        uint8_t by_rank[ORC_MAX_HW];
        for (int a = 0; a < hw; ++a)
            if (tb->corder[a] != NONE8) by_rank[tb->corder[a]] = (uint8_t)a;
        const uint32_t parent_n = node_n > 1 ? node_n : 1; /* pme.rs:82 */
        const float sq = sqrtf((float)parent_n);
        int best = by_rank[0];
        int32_t best_key = 0;
        for (int r = 0; r < nd->nch; ++r) {
            const int a = by_rank[r];
            const uint32_t n = tb->cn[a];
            const float q = tb->cw[a] / ((float)n + ORC_EPS); /* pme.rs:282 */
            const float p = eff_policy(nd, a);
            const float bias = sq / (float)(1u + n);
            const float score = q + (1.0f * p) * bias; /* C_PUCT = 1.0, pme.rs:18,285 */
            const int32_t key = total_key(score);
            if (r == 0 || key >= best_key) { best_key = key; best = a; } /* max_by: last max */
        }
        node_n = tb->cn[best];
        node = tb->cidx[best];
    }
    node_t* leaf = &t->nodes[node];
    if (leaf->status != ORC_IN_PROGRESS) { /* pme.rs:92-97 */
        backup(t, node, leaf->status >= ORC_BLACK_WIN ? 1.0f : 0.0f);
        return;
    }
    /* pme.rs:101-125 */
    int avail[ORC_MAX_HW], n_avail = 0;
    const table_t* ltb = leaf->table == NONE16 ? NULL : &t->tables[leaf->table];
    for (int a = 0; a < hw; ++a)
        if (leaf->board[a] == ORC_EMPTY && !(ltb && ltb->corder[a] != NONE8)) avail[n_avail++] = a;
    if (n_avail == 0) return;
    uint32_t o[4];
    orc_philox(sp->key, sim_index, (uint32_t)sp->ply, tree_global, ORC_RNG_EXPAND, o);
    const int action = avail[(uint32_t)(((uint64_t)o[0] * (uint64_t)n_avail) >> 32)];
    /* pme.rs:128-135 */
    orc_env env;
    env.n = sp->n;
    env.turn = leaf->turn;
    env.legal = leaf->legal;
    memcpy(env.board, leaf->board, (size_t)hw);
    const int status = orc_env_place_stone(&env, action);
    const int child = add_child(sp, t, node, action, &env, status);
    if (child < 0) return;
    if (status != ORC_IN_PROGRESS) { /* pme.rs:177-181 */
        backup(t, child, status == ORC_DRAW ? 0.0f : 1.0f);
    } else if (sp->threads > 1) { /* this game's own segment of the request list */
        if (sp->mt_cnt[game] < 64) {
            sp->req_game[64 * game + sp->mt_cnt[game]] = game;
            sp->req_node[64 * game + sp->mt_cnt[game]] = child;
            sp->mt_cnt[game]++;
        } else {
            sp->error = 2;
        }
    } else if (sp->n_req < sp->cap_req) {
        sp->req_game[sp->n_req] = game;
        sp->req_node[sp->n_req] = child;
        sp->n_req++;
    } else {
        sp->error = 2;
    }
}

/* mcts/src/lib.rs:47-78 with stable compaction */
static int tree_transition(orc_sp* sp, tree_t* t, int action) {
    const int hw = sp->hw;
    node_t* root = &t->nodes[0];
    if (root->table == NONE16) return -1;
    table_t* rtb = &t->tables[root->table];
    if (rtb->corder[action] == NONE8) return -1;
    const int c = rtb->cidx[action];
    const float new_w = rtb->cw[action];
    uint32_t new_n = 0; /* lib.rs:65-71 */
    if (t->nodes[c].table != NONE16) {
        const table_t* ctb = &t->tables[t->nodes[c].table];
        for (int a = 0; a < hw; ++a)
            if (ctb->corder[a] != NONE8) new_n += ctb->cn[a];
    }
    uint8_t* alive = sp->alive_mark;
    for (int i = 0; i < t->n_nodes; ++i) alive[i] = 0;
    alive[c] = 1;
    for (int i = c + 1; i < t->n_nodes; ++i) alive[i] = alive[t->nodes[i].parent];
    int nn = 0;
    for (int i = 0; i < t->n_nodes; ++i) sp->node_map[i] = alive[i] ? (uint16_t)nn++ : NONE16;
    int nt = 0;
    for (int k = 0; k < t->n_tables; ++k)
        sp->table_map[k] = alive[t->tables[k].owner] ? (uint16_t)nt++ : NONE16;
    for (int i = 0; i < t->n_nodes; ++i) {
        if (!alive[i]) continue;
        const int d = sp->node_map[i];
        if (d != i) t->nodes[d] = t->nodes[i];
        node_t* nd = &t->nodes[d];
        nd->parent = (i == c) ? NONE16 : sp->node_map[nd->parent];
        if (nd->table != NONE16) nd->table = sp->table_map[nd->table];
    }
    for (int k = 0; k < t->n_tables; ++k) {
        if (sp->table_map[k] == NONE16) continue;
        const int d = sp->table_map[k];
        if (d != k) t->tables[d] = t->tables[k];
        table_t* tb = &t->tables[d];
        tb->owner = sp->node_map[tb->owner];
        for (int a = 0; a < hw; ++a)
            if (tb->corder[a] != NONE8) tb->cidx[a] = sp->node_map[tb->cidx[a]];
    }
    t->n_nodes = nn;
    t->n_tables = nt;
    t->root_n = new_n;
    t->root_w = new_w;
    return 0;
}

/* agent.rs:144-197 */
static void ensure_action_exists(orc_sp* sp, tree_t* t, const orc_env* agent_env, int action, const float* p_raw) {
    const int hw = sp->hw;
    if (action >= hw) return;
    orc_env env = *agent_env;
    orc_env_place_stone(&env, action);
    float policy[ORC_MAX_HW];
    const node_t* root = &t->nodes[0];
    for (int a = 0; a < hw; ++a) policy[a] = p_raw[a];
    policy[action] = 0.0f;
    for (int a = 0; a < hw; ++a)
        if (root->board[a] != ORC_EMPTY) policy[a] = 0.0f;
    float sum = 0.0f;
    for (int a = 0; a < hw; ++a) sum += policy[a];
    if (ORC_EPS <= sum) {
        const float inv = 1.0f / sum;
        for (int a = 0; a < hw; ++a) policy[a] *= inv;
    }
    if (root->table != NONE16 && t->tables[root->table].corder[action] != NONE8) return; /* node.rs:69-71 */
    const int idx = add_child(sp, t, 0, action, &env, ORC_IN_PROGRESS);
    if (idx < 0) return;
    node_t* c = &t->nodes[idx];
    c->has_policy = 1;
    memcpy(c->policy, policy, sizeof(float) * (size_t)hw);
}

/* ------------------------------------------------------------------------------------------ */

orc_sp* orc_sp_create(int n, int games, int cap_nodes, int cap_tables, uint64_t seed, int64_t game_offset) {
    if (n < 5 || n > ORC_MAX_N || cap_nodes > 65535 || cap_tables > 65535) return NULL;
    orc_sp* sp = (orc_sp*)calloc(1, sizeof(orc_sp));
    sp->n = n;
    sp->hw = n * n;
    sp->games = games;
    sp->cap_nodes = cap_nodes;
    sp->cap_tables = cap_tables;
    sp->seed = seed;
    sp->game_offset = game_offset;
    for (int s = 0; s < 2; ++s) {
        sp->trees[s] = (tree_t*)calloc((size_t)games, sizeof(tree_t));
        for (int g = 0; g < games; ++g) {
            sp->trees[s][g].nodes = (node_t*)malloc(sizeof(node_t) * (size_t)cap_nodes);
            sp->trees[s][g].tables = (table_t*)malloc(sizeof(table_t) * (size_t)cap_tables);
        }
    }
    sp->envs = (orc_env*)calloc((size_t)games, sizeof(orc_env));
    sp->alive = (uint8_t*)calloc((size_t)games, 1);
    sp->status = (uint8_t*)calloc((size_t)games, 1);
    sp->plies = (int32_t*)calloc((size_t)games, sizeof(int32_t));
    sp->last_action = (int32_t*)calloc((size_t)games, sizeof(int32_t));
    sp->replay = (replay_t*)calloc((size_t)games, sizeof(replay_t));
    for (int g = 0; g < games; ++g) {
        replay_t* r = &sp->replay[g];
        r->boards = (uint8_t*)malloc((size_t)sp->hw * (size_t)sp->hw);
        r->turns = (uint8_t*)malloc((size_t)sp->hw);
        r->pi = (float*)malloc(sizeof(float) * (size_t)sp->hw * (size_t)sp->hw);
        r->z = (float*)malloc(sizeof(float) * (size_t)sp->hw);
    }
    sp->cap_req = games * 64;
    sp->req_game = (int32_t*)malloc(sizeof(int32_t) * (size_t)sp->cap_req);
    sp->req_node = (int32_t*)malloc(sizeof(int32_t) * (size_t)sp->cap_req);
    sp->alive_mark = (uint8_t*)malloc((size_t)cap_nodes);
    sp->node_map = (uint16_t*)malloc(sizeof(uint16_t) * (size_t)cap_nodes);
    sp->table_map = (uint16_t*)malloc(sizeof(uint16_t) * (size_t)cap_tables);
    sp->threads = 1;
    sp->mt_cnt = (int32_t*)calloc((size_t)games, sizeof(int32_t));
    return sp;
}

void orc_sp_destroy(orc_sp* sp) {
    if (!sp) return;
    for (int s = 0; s < 2; ++s) {
        for (int g = 0; g < sp->games; ++g) {
            free(sp->trees[s][g].nodes);
            free(sp->trees[s][g].tables);
        }
        free(sp->trees[s]);
    }
    for (int g = 0; g < sp->games; ++g) {
        free(sp->replay[g].boards); free(sp->replay[g].turns); free(sp->replay[g].pi); free(sp->replay[g].z);
    }
    free(sp->envs); free(sp->alive); free(sp->status); free(sp->plies); free(sp->last_action);
    free(sp->replay); free(sp->req_game); free(sp->req_node); free(sp->alive_mark);
    free(sp->node_map); free(sp->table_map); free(sp->mt_cnt);
    free(sp);
}

void orc_sp_set_episode(orc_sp* sp, uint64_t episode) { sp->episode = episode; }
void orc_sp_set_threads(orc_sp* sp, int threads) { sp->threads = threads > 1 ? threads : 1; }

void orc_sp_reset(orc_sp* sp, const float* root_policy) {
    sp->key = orc_stream_key(sp->seed, sp->episode);
    sp->episode += 1;
    sp->external = 0;
    sp->ply = 0;
    sp->error = 0;
    sp->n_req = 0;
    sp->stat_sims = 0;
    for (int g = 0; g < sp->games; ++g) {
        orc_env_init(&sp->envs[g], sp->n);
        sp->alive[g] = 1;
        sp->status[g] = ORC_IN_PROGRESS;
        sp->plies[g] = 0;
        sp->last_action[g] = -1;
        sp->replay[g].plies = 0;
        tree_init(&sp->trees[0][g], &sp->envs[g], root_policy, sp->hw);
        tree_init(&sp->trees[1][g], &sp->envs[g], root_policy, sp->hw);
    }
}

int orc_sp_ply(const orc_sp* sp) { return sp->ply; }
int orc_sp_error(const orc_sp* sp) { return sp->error; }
int orc_sp_game_alive(const orc_sp* sp, int g) { return sp->alive[g]; }
int orc_sp_game_status(const orc_sp* sp, int g) { return sp->status[g]; }
int orc_sp_game_plies(const orc_sp* sp, int g) { return sp->plies[g]; }
int orc_sp_alive_count(const orc_sp* sp) {
    int c = 0;
    for (int g = 0; g < sp->games; ++g) c += sp->alive[g];
    return c;
}

int orc_sp_round_generate(orc_sp* sp, int round, int batch_size, float epsilon, float alpha,
                          float* inputs, int max_req) {
    const int side = sp->ply & 1;
    sp->n_req = 0;
    if (sp->threads > 1 && batch_size <= 64) { /* games are independent (pme.rs:200-205): one game per task, requests packed in game order afterwards */
        int64_t sims = 0;
#pragma omp parallel for schedule(dynamic, 4) num_threads(sp->threads) reduction(+ : sims)
        for (int g = 0; g < sp->games; ++g) {
            sp->mt_cnt[g] = 0;
            if (!sp->alive[g]) continue;
            tree_t* t = &sp->trees[side][g];
            const uint32_t tree_global = (uint32_t)((sp->game_offset + g) * 2 + side);
            if (round == 0) apply_noise(sp, t, epsilon, alpha, tree_global);
            for (int i = 0; i < batch_size; ++i) {
                run_sim(sp, t, g, (uint32_t)(round * batch_size + i), tree_global);
                sims += 1;
            }
        }
        sp->stat_sims += sims;
        for (int g = 0; g < sp->games; ++g) /* (ascending, destination <= source: in place) */
            for (int i = 0; i < sp->mt_cnt[g]; ++i) {
                sp->req_game[sp->n_req] = sp->req_game[64 * g + i];
                sp->req_node[sp->n_req] = sp->req_node[64 * g + i];
                sp->n_req++;
            }
    } else
    for (int g = 0; g < sp->games; ++g) {
        if (!sp->alive[g]) continue;
        tree_t* t = &sp->trees[side][g];
        const uint32_t tree_global = (uint32_t)((sp->game_offset + g) * 2 + side);
        if (round == 0) apply_noise(sp, t, epsilon, alpha, tree_global);
        for (int i = 0; i < batch_size; ++i) {
            run_sim(sp, t, g, (uint32_t)(round * batch_size + i), tree_global);
            sp->stat_sims += 1;
        }
    }
    if (inputs) {
        if (sp->n_req > max_req) { sp->error = 3; return -1; }
#pragma omp parallel for schedule(static) num_threads(sp->threads) if (sp->threads > 1)
        for (int r = 0; r < sp->n_req; ++r) {
            orc_env env;
            env.n = sp->n;
            const node_t* nd = &sp->trees[side][sp->req_game[r]].nodes[sp->req_node[r]];
            env.turn = nd->turn;
            env.legal = nd->legal;
            memcpy(env.board, nd->board, (size_t)sp->hw);
            orc_encode_nn_input(&env, ORC_MODE_PLAYER, inputs + (size_t)r * 3 * (size_t)sp->hw);
        }
    }
    return sp->n_req;
}

void orc_sp_request_info(const orc_sp* sp, int r, int* game, int* node) {
    *game = sp->req_game[r];
    *node = sp->req_node[r];
}

/* pme.rs:222-265 */
void orc_sp_round_scatter(orc_sp* sp, const float* p, const float* v) {
    const int side = sp->ply & 1, hw = sp->hw;
    /* a game's requests are adjacent and must be applied in order (one tree); different games are independent: with threads > 1 a task = one game's stretch */
    int n_seg = 0;
    int32_t* seg = sp->mt_cnt; /* (reused: start of every stretch, n_seg <= games) */
    if (sp->threads > 1)
        for (int r = 0; r < sp->n_req; ++r)
            if (r == 0 || sp->req_game[r] != sp->req_game[r - 1]) seg[n_seg++] = r;
#pragma omp parallel for schedule(dynamic, 4) num_threads(sp->threads) if (sp->threads > 1)
    for (int sgi = 0; sgi < (sp->threads > 1 ? n_seg : 1); ++sgi) {
      const int r0 = sp->threads > 1 ? seg[sgi] : 0, r1 = sp->threads > 1 ? (sgi + 1 < n_seg ? seg[sgi + 1] : sp->n_req) : sp->n_req;
      for (int r = r0; r < r1; ++r) {
        tree_t* t = &sp->trees[side][sp->req_game[r]];
        node_t* nd = &t->nodes[sp->req_node[r]];
        const float* raw = p + (size_t)r * (size_t)hw;
        const float value = -v[r];
        for (int a = 0; a < hw; ++a) nd->policy[a] = nd->board[a] == ORC_EMPTY ? raw[a] : 0.0f;
        float sum = 0.0f;
        for (int a = 0; a < hw; ++a) sum += nd->policy[a];
        if (ORC_EPS <= sum) {
            const float inv = 1.0f / sum;
            for (int a = 0; a < hw; ++a) nd->policy[a] *= inv;
        }
        nd->has_policy = 1;
        backup(t, sp->req_node[r], value);
      }
    }
    sp->n_req = 0;
}

/* agent.rs:43-137 + trainer.rs:138-173 (record) */
void orc_sp_sample(orc_sp* sp, float temperature, int threshold, int32_t* actions) {
    const int side = sp->ply & 1, hw = sp->hw;
    sp->external = 0;
    for (int g = 0; g < sp->games; ++g) {
        actions[g] = -1;
        sp->last_action[g] = -1;
        if (!sp->alive[g]) continue;
        tree_t* t = &sp->trees[side][g];
        const node_t* root = &t->nodes[0];
        float policy[ORC_MAX_HW];
        for (int a = 0; a < hw; ++a) policy[a] = 0.0f;
        float sum = 0.0f;
        if (root->table == NONE16 || root->nch == 0) { sp->error = 4; continue; }
        const table_t* tb = &t->tables[root->table];
        for (int a = 0; a < hw; ++a)
            if (tb->corder[a] != NONE8) { policy[a] = (float)tb->cn[a]; sum += policy[a]; }
        if (sum < ORC_EPS) { sp->error = 4; continue; }
        const float sum_inv = 1.0f / sum;
        for (int a = 0; a < hw; ++a) policy[a] *= sum_inv;
        int action = 0;
        if (sp->replay[g].plies < threshold) { /* turn_counts[index] (trainer.rs:139): moves SAMPLED so far; Boltzmann, agent.rs:106-133 */
            float heated[ORC_MAX_HW];
            float hsum = 0.0f;
            const float tinv = 1.0f / temperature;
            for (int a = 0; a < hw; ++a) {
                heated[a] = 0.0f;
                if (policy[a] < ORC_EPS) continue;
                heated[a] = orc_det_expf(policy[a] * tinv);
                hsum += heated[a];
            }
            const float hinv = 1.0f / hsum;
            for (int a = 0; a < hw; ++a) heated[a] *= hinv;
            float total = 0.0f;
            for (int a = 0; a < hw; ++a) total += heated[a];
            uint32_t o[4];
            orc_philox(sp->key, 0, (uint32_t)sp->ply, (uint32_t)((sp->game_offset + g) * 2 + side), ORC_RNG_SAMPLE, o);
            const float u = (float)(o[0] >> 8) * 5.9604644775390625e-8f; /* 2^-24 */
            const float target = u * total;
            float cum = 0.0f;
            int chosen = -1, last_nz = 0;
            for (int a = 0; a < hw; ++a) {
                if (!(heated[a] > 0.0f)) continue;
                last_nz = a;
                cum += heated[a];
                if (chosen < 0 && cum > target) chosen = a;
            }
            action = chosen < 0 ? last_nz : chosen;
        } else { /* Best: last max by total_cmp, agent.rs:98-105 */
            int32_t best_key = total_key(policy[0]);
            for (int a = 1; a < hw; ++a) {
                const int32_t k = total_key(policy[a]);
                if (k >= best_key) { best_key = k; action = a; }
            }
        }
        actions[g] = action;
        sp->last_action[g] = action;
        replay_t* rp = &sp->replay[g];
        if (rp->plies < hw) {
            memcpy(rp->boards + (size_t)rp->plies * (size_t)hw, sp->envs[g].board, (size_t)hw);
            rp->turns[rp->plies] = sp->envs[g].turn;
            memcpy(rp->pi + (size_t)rp->plies * (size_t)hw, policy, sizeof(float) * (size_t)hw);
            rp->z[rp->plies] = 0.0f;
        }
    }
}

void orc_sp_set_actions(orc_sp* sp, const int32_t* actions) {
    sp->external = 1;
    for (int g = 0; g < sp->games; ++g) sp->last_action[g] = sp->alive[g] ? actions[g] : -1;
}

int orc_sp_compute_policy(const orc_sp* sp, int game, float* policy) { /* agent.rs:43-77 */
    const tree_t* t = &sp->trees[sp->ply & 1][game];
    const node_t* root = &t->nodes[0];
    const int hw = sp->hw;
    for (int a = 0; a < hw; ++a) policy[a] = 0.0f;
    if (!sp->alive[game] || root->table == NONE16 || root->nch == 0) return 0;
    const table_t* tb = &t->tables[root->table];
    float sum = 0.0f;
    for (int a = 0; a < hw; ++a)
        if (tb->corder[a] != NONE8) { policy[a] = (float)tb->cn[a]; sum += policy[a]; }
    if (sum < ORC_EPS) return 0;
    const float sum_inv = 1.0f / sum;
    for (int a = 0; a < hw; ++a) policy[a] *= sum_inv;
    return 1;
}

int orc_sp_mirror_generate(orc_sp* sp, float* inputs, int max_req) {
    int cnt = 0;
    for (int g = 0; g < sp->games; ++g) {
        if (!sp->alive[g] || sp->last_action[g] < 0) continue;
        if (cnt >= max_req) { sp->error = 3; return -1; }
        orc_env env = sp->envs[g];
        orc_env_place_stone(&env, sp->last_action[g]);
        orc_encode_nn_input(&env, ORC_MODE_OPPONENT, inputs + (size_t)cnt * 3 * (size_t)sp->hw);
        ++cnt;
    }
    return cnt;
}

void orc_sp_advance(orc_sp* sp, const float* p) {
    const int side = sp->ply & 1, hw = sp->hw;
    int cnt = 0;
    for (int g = 0; g < sp->games; ++g) {
        if (!sp->alive[g] || sp->last_action[g] < 0) continue;
        const int action = sp->last_action[g];
        tree_t* own = &sp->trees[side][g];
        tree_t* opp = &sp->trees[1 - side][g];
        const orc_env before = sp->envs[g];
        if (sp->external) ensure_action_exists(sp, own, &before, action, p + (size_t)cnt * (size_t)hw); /* an external move need not be in the tree */
        /* agent.play_action (agent.rs:206-232) */
        if (own->nodes[0].status != ORC_IN_PROGRESS) { sp->error = 5; }
        const int status = orc_env_place_stone(&sp->envs[g], action);
        if (status < 0 || tree_transition(sp, own, action) != 0) sp->error = 5;
        /* opposite.ensure_action_exists + play_action (trainer.rs:163-167) */
        ensure_action_exists(sp, opp, &before, action, p + (size_t)cnt * (size_t)hw);
        if (tree_transition(sp, opp, action) != 0) sp->error = 6;
        ++cnt;
        replay_t* rp = &sp->replay[g];
        if (!sp->external && rp->plies < hw) { /* transitions.push (trainer.rs:169-173): sampled moves only */
            rp->z[rp->plies] = (status == ORC_BLACK_WIN || status == ORC_WHITE_WIN) ? 1.0f : 0.0f;
            rp->plies++;
        }
        sp->plies[g] += 1;
        sp->status[g] = (uint8_t)(status < 0 ? 0 : status);
        if (status != ORC_IN_PROGRESS) sp->alive[g] = 0;
        sp->last_action[g] = -1;
    }
    sp->external = 0;
    sp->ply += 1;
}

int orc_sp_tree_dump(const orc_sp* sp, int game, int side, int32_t* ints, float* floats, int cap_nodes) {
    const tree_t* t = &sp->trees[side][game];
    const int hw = sp->hw;
    if (t->n_nodes > cap_nodes) return -t->n_nodes;
    for (int i = 0; i < t->n_nodes; ++i) {
        const node_t* nd = &t->nodes[i];
        int32_t* o = ints + (size_t)i * 8;
        float* f = floats + (size_t)i * (size_t)(1 + hw);
        uint32_t n = t->root_n;
        float w = t->root_w;
        int order = -1;
        if (i != 0) {
            const table_t* tb = &t->tables[t->nodes[nd->parent].table];
            n = tb->cn[nd->action];
            w = tb->cw[nd->action];
            order = tb->corder[nd->action];
        }
        o[0] = nd->parent == NONE16 ? -1 : nd->parent;
        o[1] = nd->action == NONE8 ? -1 : nd->action;
        o[2] = nd->status;
        o[3] = nd->turn;
        o[4] = nd->legal;
        o[5] = nd->nch;
        o[6] = (int32_t)n;
        o[7] = (order & 0xffff) | ((int32_t)nd->has_policy << 16);
        f[0] = w;
        for (int a = 0; a < hw; ++a) f[1 + a] = eff_policy(nd, a);
    }
    return t->n_nodes;
}

void orc_sp_tree_root(const orc_sp* sp, int game, int side, uint32_t* root_n, float* root_w, int* n_nodes, int* n_tables) {
    const tree_t* t = &sp->trees[side][game];
    *root_n = t->root_n; *root_w = t->root_w; *n_nodes = t->n_nodes; *n_tables = t->n_tables;
}

int orc_sp_replay(const orc_sp* sp, int game, uint8_t* boards, uint8_t* turns, float* pi, float* z, int cap_plies) {
    const replay_t* rp = &sp->replay[game];
    const int hw = sp->hw;
    const int n = rp->plies < cap_plies ? rp->plies : cap_plies;
    memcpy(boards, rp->boards, (size_t)n * (size_t)hw);
    memcpy(turns, rp->turns, (size_t)n);
    memcpy(pi, rp->pi, sizeof(float) * (size_t)n * (size_t)hw);
    memcpy(z, rp->z, sizeof(float) * (size_t)n);
    return rp->plies;
}

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* trainer.rs:95-205 with the oracle net as evaluator */
int orc_selfplay_run(orc_sp* sp, const orc_net* net, int count, int batch_size, float epsilon,
                     float alpha, float temperature, int threshold, int max_plies, int threads,
                     double* stats) {
    const int hw = sp->hw;
    const size_t max_req = (size_t)sp->games * (size_t)batch_size;
    float* inputs = (float*)malloc(sizeof(float) * 3 * (size_t)hw * (max_req > (size_t)sp->games ? max_req : (size_t)sp->games));
    float* p = (float*)malloc(sizeof(float) * (size_t)hw * (max_req > (size_t)sp->games ? max_req : (size_t)sp->games));
    float* v = (float*)malloc(sizeof(float) * (max_req > (size_t)sp->games ? max_req : (size_t)sp->games));
    int32_t* actions = (int32_t*)malloc(sizeof(int32_t) * (size_t)sp->games);
    double evals = 0, t_net = 0, plies_games = 0;
    const double t0 = now_s();
    const int games0 = orc_sp_alive_count(sp);
    int plies_done = 0;
    while (orc_sp_alive_count(sp) > 0 && (max_plies <= 0 || plies_done < max_plies)) {
        int processed = 0, round = 0;
        while (processed < count) { /* pme.rs:39-42,207 */
            const int b = orc_sp_round_generate(sp, round, batch_size, epsilon, alpha, inputs, (int)max_req);
            processed += batch_size;
            round += 1;
            if (b <= 0) continue;
            const double tn = now_s();
            orc_net_forward(net, inputs, b, p, v, threads);
            t_net += now_s() - tn;
            evals += b;
            orc_sp_round_scatter(sp, p, v);
        }
        plies_games += orc_sp_alive_count(sp);
        orc_sp_sample(sp, temperature, threshold, actions);
        const int m = orc_sp_mirror_generate(sp, inputs, sp->games);
        const double tn = now_s();
        orc_net_forward(net, inputs, m, p, v, threads);
        t_net += now_s() - tn;
        evals += m;
        orc_sp_advance(sp, p);
        plies_done += 1;
    }
    if (stats) {
        stats[0] = sp->stat_sims;
        stats[1] = evals;
        stats[2] = plies_games;
        stats[3] = (double)(games0 - orc_sp_alive_count(sp));
        stats[4] = t_net;
        stats[5] = now_s() - t0;
    }
    free(inputs); free(p); free(v); free(actions);
    return sp->error;
}
