/* TEST INFRASTRUCTURE — NOT PRODUCT CODE.
 *
 * Sanitizer harness for the oracle: replays a dump of the golden traces (inputs + every expected
 * output, written by tests/test_oracle_sanitizers.py) through oracle/qttt_oracle.c, compiled
 * together with it under -fsanitize=address,undefined.  Exits 0 iff every output matches and the
 * sanitizers stayed silent (they abort the process otherwise: -fno-sanitize-recover).
 *
 * File: u32 E, u32 T, then per (t, e) records in t-major order:
 *   actions u8[2], bit u8, board i8[9], n_moves u8, moves u8[9][2], n_q u8, qmask u16[4],
 *   reward_bits u32, terminated u8, p1 i8, p2 i8, q_p1 u8[5][2], l1 u8, q_p2 u8[4][2], l2 u8, turn u8
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "qttt_oracle.h"

#pragma pack(push, 1)
typedef struct {
    uint8_t actions[2], bit;
    int8_t board[9];
    uint8_t n_moves, moves[9][2], n_q;
    uint16_t qmask[4];
    uint32_t reward_bits;
    uint8_t terminated;
    int8_t p1, p2;
    uint8_t q_p1[5][2], l1, q_p2[4][2], l2, turn;
} rec_t;
#pragma pack(pop)

static int fail(const char *what, unsigned e, unsigned t) {
    fprintf(stderr, "san_replay: %s differs at episode %u step %u\n", what, e, t);
    return 1;
}

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    uint32_t E, T;
    if (fread(&E, 4, 1, f) != 1 || fread(&T, 4, 1, f) != 1) return 2;
    qo_board *b = (qo_board *)malloc(sizeof(qo_board) * E);
    rec_t *rec = (rec_t *)malloc(sizeof(rec_t) * E);
    uint8_t *acts = (uint8_t *)malloc(2u * E), *bits = (uint8_t *)malloc(E), *term = (uint8_t *)malloc(E);
    float *reward = (float *)malloc(sizeof(float) * E);
    qo_reset_batch(b, E);
    for (uint32_t t = 0; t < T; ++t) {
        if (fread(rec, sizeof(rec_t), E, f) != E) return 2;
        for (uint32_t e = 0; e < E; ++e) {
            acts[2 * e] = rec[e].actions[0];
            acts[2 * e + 1] = rec[e].actions[1];
            bits[e] = rec[e].bit;
        }
        qo_step_batch(b, E, acts, bits, 0, t, 0, 0, reward, term);          /* env.py:34-53 */
        for (uint32_t e = 0; e < E; ++e) {
            const rec_t *r = &rec[e];
            uint32_t rb;
            memcpy(&rb, &reward[e], 4);
            if (rb != r->reward_bits) return fail("reward bits", e, t);
            if (term[e] != r->terminated) return fail("terminated", e, t);
            if (memcmp(b[e].board, r->board, 9)) return fail("board", e, t);
            if (b[e].n_moves != r->n_moves) return fail("n_moves", e, t);
            for (int i = 0; i < b[e].n_moves; ++i)
                if ((uint8_t)b[e].moves[i][0] != r->moves[i][0] || (uint8_t)b[e].moves[i][1] != r->moves[i][1])
                    return fail("moves", e, t);
            if (b[e].n_q != r->n_q) return fail("n_q", e, t);
            for (int i = 0; i < b[e].n_q; ++i)
                if (b[e].q[i] != r->qmask[i]) return fail("qmask", e, t);
            int p1, p2;
            qo_check_win(&b[e], &p1, &p2);                                     /* board.py:71-115 */
            if (p1 != r->p1 || p2 != r->p2) return fail("check_win", e, t);
            int8_t cl[9];
            uint8_t q1[5][2], q2[4][2];
            int l1, l2, turn;
            memset(q1, 255, sizeof q1);
            memset(q2, 255, sizeof q2);
            qo_observe(&b[e], cl, q1, &l1, q2, &l2, &turn);                    /* env.py:68-85 */
            if (memcmp(cl, r->board, 9) || l1 != r->l1 || l2 != r->l2 || turn != r->turn ||
                memcmp(q1, r->q_p1, (size_t)2 * l1) || memcmp(q2, r->q_p2, (size_t)2 * l2))
                return fail("observation", e, t);
            /* the §8(f) helpers on the same states: no expected values here, the sanitizers watch */
            int w, tl;
            qo_update_winner(&b[e], &w, &tl);
            double vec[180];
            qo_to_vector(&b[e], vec);
            (void)qo_legal_mask(&b[e]);
            (void)qo_pyhash(&b[e]);
            qo_board kid[2];
            int kw[2], kt[2];
            uint64_t kl[2];
            (void)qo_expand(&b[e], (int)((e + t) % 37u), kid, kw, kt, kl);     /* 36 = out of range on purpose */
            uint8_t sa[2];
            qo_sample_action(&b[e], 7, e, t, sa);
        }
    }
    fclose(f);
    free(b); free(rec); free(acts); free(bits); free(term); free(reward);
    printf("san_replay ok: %u episodes x %u steps\n", E, T);
    return 0;
}
