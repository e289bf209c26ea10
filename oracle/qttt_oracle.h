/* TEST INFRASTRUCTURE — NOT PRODUCT CODE.
 *
 * CPU restatement (plain C) of the reference's Env.step() hot path, one function per
 * reference function, each citing the /root/reference file:line it follows.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the product
 * (qtttgym_amd + libqttt_hip.so) never does.
 *
 * Parity pin: checked bit-for-bit against tests/golden/step_traces.npz, which was produced by
 * running the unmodified reference here (tests/golden/make_golden.py), and fuzzed live against
 * the imported reference in the build container (tests/test_oracle_vs_reference.py).
 */
#ifndef QTTT_ORACLE_H
#define QTTT_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Mirror of qtttgym.Board's attributes (board.py:2-7). Sets are 9-bit masks over squares. */
typedef struct {
    int32_t n_moves;          /* len(self.moves)                         */
    int8_t  moves[9][2];      /* self.moves[i][0:2]; moves[i][2] == i    */
    int8_t  board[9];         /* self.board: -1 empty else round 0..8    */
    int32_t n_q;              /* len(self.qstructs)                      */
    uint16_t q[5];            /* self.qstructs in list order             */
} qo_board;                   /* sizeof == 48 */

enum { QO_OK = 0, QO_ERR_SAME_SQUARE = 1, QO_ERR_CLASSICAL = 2, QO_ERR_INDEX = 3 };

void qo_init(qo_board *b);                                         /* board.py:2-7   */
int  qo_make_move(qo_board *b, int a, int c, int bit, int *consumed); /* board.py:9-25  */
void qo_check_win(const qo_board *b, int *p1_round, int *p2_round);   /* board.py:71-115 */
/* env.py:34-53.  reward is the reference's f64 (-0.0 / -1.0). Returns make_move's status. */
int  qo_step(qo_board *b, int a, int c, int bit, double *reward, int *terminated, int *consumed);
/* env.py:68-85 */
void qo_observe(const qo_board *b, int8_t classical[9], uint8_t q_p1[5][2], int *q_p1_len,
                uint8_t q_p2[4][2], int *q_p2_len, int *turn);

/* ---- batch forms (parity at N = 4096.. and the bench's cpu_baseline leg) ---- */
void qo_reset_batch(qo_board *b, int64_t n);
/* bits may be NULL: then the collapse bit is qo_collapse_bit(seed, board_offset+i, step_idx).
 * auto_reset != 0: a board that was terminated (env.py:51) before this step is re-initialised
 * first (the build's throughput mode; the reference has no auto-reset). */
void qo_step_batch(qo_board *b, int64_t n, const uint8_t *actions, const uint8_t *bits,
                   uint64_t seed, uint32_t step_idx, int64_t board_offset, int auto_reset,
                   float *reward, uint8_t *terminated);
void qo_replay_batch(qo_board *b, int64_t n, const uint8_t *actions, int64_t stride, int32_t n_steps,
                     uint64_t seed, uint32_t step_idx0, int64_t board_offset, int auto_reset,
                     float *reward, uint8_t *terminated);
int  qo_terminated(const qo_board *b);                             /* env.py:48,51 */

/* ---- the build's synthetic-input spec, shared with the HIP policy kernel ---- */
uint64_t qo_hash(uint64_t seed, uint64_t board_id, uint32_t step_idx);
int      qo_collapse_bit(uint64_t seed, uint64_t board_id, uint32_t step_idx);
/* uniform over legal unordered pairs (mcts.py:20-27 rule), lexicographic ind2move order
 * (mcts.py:339-343).  No legal pair -> (0,0), a noop. */
void qo_sample_action(const qo_board *b, uint64_t seed, uint64_t board_id, uint32_t step_idx,
                      uint8_t out[2]);
void qo_sample_actions_batch(const qo_board *b, int64_t n, uint64_t seed, uint32_t step_idx,
                             int64_t board_offset, int auto_reset, uint8_t *actions);

/* ---- MCTS expand (mcts.py:233-267, :52-65, :20-27) for the §8(f) row ---- */
/* winner: 1 = True (p1), 0 = False (p2), -1 = None.  Returns number of children (1 or 2);
 * child[0] is the bit-0 branch, child[1] the bit-1 branch (only when a collapse happened).
 * Returns 0 if the move raised inside make_move (children untouched). */
int  qo_expand(const qo_board *parent, int action36, qo_board child[2], int winner[2],
               int terminal[2], uint64_t legal_mask[2]);
void qo_ind2move(int action36, int *lo, int *hi);                  /* mcts.py:339-343 */
void qo_update_winner(const qo_board *b, int *winner, int *terminal); /* mcts.py:52-65 */
uint64_t qo_legal_mask(const qo_board *b);                         /* mcts.py:20-27  */
/* mcts.py:67-85 to_vector: out[18][10] row-major, f64 like numpy's default. */
void qo_to_vector(const qo_board *b, double out[180]);
/* mcts.py:93-94 under CPython >= 3.8 */
int64_t qo_pyhash(const qo_board *b);

#ifdef __cplusplus
}
#endif
#endif
