/* ORACLE (test infrastructure only) — board rules, restating environment/src/lib.rs:62-194,
 * NN-input encoding restating alpha-zero/src/encoder.rs:10-46, symmetry helpers restating
 * src/utils.rs:1-64.  See omok_oracle.h for the pinning statement. */
#include "omok_oracle.h"
#include <string.h>

/* environment/src/lib.rs:73-79 */
void orc_env_init(orc_env* e, int n) {
    e->n = n;
    e->turn = ORC_TURN_BLACK;
    e->legal = (uint16_t)(n * n);
    memset(e->board, ORC_EMPTY, sizeof(e->board));
}

/* environment/src/lib.rs:168-193: walk k = 1..5 along (dx,dy); stop at the edge or at the
 * first cell that is not `stone`. */
static int count_serial(const orc_env* e, int stone, int index, int dx, int dy) {
    const int n = e->n;
    const int x0 = index % n, y0 = index / n;
    int count = 0;
    for (int k = 1; k <= 5; ++k) {
        const int x = x0 + dx * k, y = y0 + dy * k;
        if (x < 0 || n <= x || y < 0 || n <= y) break;
        if (e->board[y * n + x] != stone) break;
        ++count;
    }
    return count;
}

/* environment/src/lib.rs:104-166 */
int orc_env_place_stone(orc_env* e, int index) {
    if (e->board[index] != ORC_EMPTY) return -1; /* None (:105-107) */
    e->legal -= 1;
    const int stone = e->turn == ORC_TURN_BLACK ? ORC_BLACK : ORC_WHITE;
    e->board[index] = (uint8_t)stone;
    const int h = 1 + count_serial(e, stone, index, -1, 0) + count_serial(e, stone, index, 1, 0);
    const int v = 1 + count_serial(e, stone, index, 0, -1) + count_serial(e, stone, index, 0, 1);
    const int d1 = 1 + count_serial(e, stone, index, -1, -1) + count_serial(e, stone, index, 1, 1);
    const int d2 = 1 + count_serial(e, stone, index, -1, 1) + count_serial(e, stone, index, 1, -1);
    const int turn = e->turn;
    e->turn = (uint8_t)(1 - e->turn); /* :148 */
    if (h == 5 || v == 5 || d1 == 5 || d2 == 5) /* exactly five, :151-154 */
        return turn == ORC_TURN_BLACK ? ORC_BLACK_WIN : ORC_WHITE_WIN;
    if (e->legal == 0) return ORC_DRAW;
    return ORC_IN_PROGRESS;
}

/* environment/src/lib.rs:81-102 */
void orc_env_encode_board(const orc_env* e, int turn, float* dst) {
    const int hw = e->n * e->n;
    for (int i = 0; i < 2 * hw; ++i) dst[i] = 0.0f;
    const int black_offset = turn == ORC_TURN_BLACK ? 0 : 1;
    const int white_offset = 1 - black_offset;
    for (int i = 0; i < hw; ++i) {
        if (e->board[i] == ORC_EMPTY) continue;
        dst[i * 2 + (e->board[i] == ORC_BLACK ? black_offset : white_offset)] = 1.0f;
    }
}

/* alpha-zero/src/encoder.rs:22-43 (one sample) */
void orc_encode_nn_input(const orc_env* e, int mode, float* dst) {
    const int hw = e->n * e->n;
    const int persp = mode == ORC_MODE_PLAYER ? e->turn : 1 - e->turn;
    orc_env_encode_board(e, persp, dst);
    const float value = e->turn == ORC_TURN_BLACK ? 1.0f : 0.0f;
    for (int i = 2 * hw; i < 3 * hw; ++i) dst[i] = value;
}

/* src/utils.rs:1-64 */
void orc_rotate_90(const float* src, float* dst, int size) {
    for (int i = 0; i < size; ++i)
        for (int j = 0; j < size; ++j) dst[i * size + j] = src[(size - j - 1) * size + i];
}
void orc_rotate_180(const float* src, float* dst, int size) {
    for (int i = 0; i < size; ++i)
        for (int j = 0; j < size; ++j) dst[i * size + j] = src[(size - i - 1) * size + (size - j - 1)];
}
void orc_rotate_270(const float* src, float* dst, int size) {
    for (int i = 0; i < size; ++i)
        for (int j = 0; j < size; ++j) dst[i * size + j] = src[j * size + (size - i - 1)];
}
void orc_flip_horizontal(const float* src, float* dst, int size) {
    for (int i = 0; i < size; ++i)
        for (int j = 0; j < size; ++j) dst[i * size + j] = src[i * size + (size - j - 1)];
}
void orc_flip_vertical(const float* src, float* dst, int size) {
    for (int i = 0; i < size; ++i)
        for (int j = 0; j < size; ++j) dst[i * size + j] = src[(size - i - 1) * size + j];
}

/* Replay post-processing of Trainer::train, src/trainer.rs:207-324, for ONE game of `len` transitions.
 * in : boards [len][hw] Stone bytes, turns [len], pi [len][hw], z [len] (z as recorded at play time, :156-173)
 * out: 6*len records: first the transitions with z back-filled (:209-214: z = last.z; walking backwards
 *      transition.z = z; z = -z), then per transition (:222-318) rotate_90, rotate_180, rotate_270, flip_horizontal,
 *      flip_vertical of env.board and policy (env.turn cloned, z copied); replay_memory.extend(transitions) then
 *      .extend(augmented) (:320-321). */
static void orc_xform_u8(int k, const uint8_t* src, uint8_t* dst, int n) {
    float a[ORC_MAX_HW], b[ORC_MAX_HW];
    for (int i = 0; i < n * n; ++i) a[i] = (float)src[i];
    switch (k) {
        case 0: orc_rotate_90(a, b, n); break;
        case 1: orc_rotate_180(a, b, n); break;
        case 2: orc_rotate_270(a, b, n); break;
        case 3: orc_flip_horizontal(a, b, n); break;
        default: orc_flip_vertical(a, b, n); break;
    }
    for (int i = 0; i < n * n; ++i) dst[i] = (uint8_t)b[i];
}
static void orc_xform_f32(int k, const float* src, float* dst, int n) {
    switch (k) {
        case 0: orc_rotate_90(src, dst, n); break;
        case 1: orc_rotate_180(src, dst, n); break;
        case 2: orc_rotate_270(src, dst, n); break;
        case 3: orc_flip_horizontal(src, dst, n); break;
        default: orc_flip_vertical(src, dst, n); break;
    }
}
void orc_replay_postprocess(int n, int len, const uint8_t* boards, const uint8_t* turns, const float* pi, const float* z_in,
                            uint8_t* boards_out, uint8_t* turns_out, float* pi_out, float* z_out) {
    const int hw = n * n;
    if (len <= 0) return;
    float z = z_in[len - 1];
    for (int t = len - 1; t >= 0; --t) { /* :211-214 */
        z_out[t] = z;
        z = -z;
    }
    for (int t = 0; t < len; ++t) { /* :320 */
        for (int i = 0; i < hw; ++i) { boards_out[(size_t)t * hw + i] = boards[(size_t)t * hw + i]; pi_out[(size_t)t * hw + i] = pi[(size_t)t * hw + i]; }
        turns_out[t] = turns[t];
    }
    for (int t = 0; t < len; ++t)
        for (int k = 0; k < 5; ++k) { /* :222-318, :321 */
            const size_t o = (size_t)len + 5 * (size_t)t + k;
            orc_xform_u8(k, boards + (size_t)t * hw, boards_out + o * hw, n);
            orc_xform_f32(k, pi + (size_t)t * hw, pi_out + o * hw, n);
            turns_out[o] = turns[t];
            z_out[o] = z_out[t];
        }
}
