/* qttt.h — C ABI of libqttt_hip.so, the MI355X (gfx950) vectorised Quantum Tic-Tac-Toe
 * environment.  This is the drop-in boundary for the reference's Env.step() hot path
 * (Oxel40/qtttgym @ v1): each entry point cites the reference interface it replaces.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is DEVICE memory owned by the caller
 *     (contiguous; torch-ROCm tensors in the host package), nothing is allocated or freed here (the records of
 *     qttt_board_op* only need to be device-ACCESSIBLE: pinned host memory works, and qttt_board_op_host requires it);
 *   - `stream` is a hipStream_t passed as void*; work is enqueued on it and ordered by it;
 *     no call waits for the device (qttt_board_op_sync and qttt_board_op_host, which say so, are the two exceptions);
 *   - return value: 0 = ok, >0 = hipError_t of the launch, <0 = argument error
 *     (QTTT_ERR_NULL / QTTT_ERR_SIZE / QTTT_ERR_ACTION = a pointer not aligned as the entry needs);
 *   - the library keeps no per-call and no per-environment state: every entry point may be called from
 *     any number of host threads at once (on different buffers).  The one process-wide datum is the
 *     DEFAULT launch shape (qttt_set_tuning, one relaxed atomic word); it never changes results, and a
 *     call that carries QTTT_FLAG_SHAPE(...) in its flags does not read it;
 *   - illegal *actions* are data, not errors: they are noops exactly as env.py:36-43.
 *
 * State: an opaque device buffer of qttt_state_bytes(n) bytes for n boards (16 B/board,
 * structure-of-arrays: u64 plane P[s], u64 plane Q[s], plane stride s = n rounded up to a
 * multiple of 64).  16-byte aligned at least.
 */
#ifndef QTTT_H
#define QTTT_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QTTT_ABI_VERSION 5

#define QTTT_ERR_NULL   (-1)
#define QTTT_ERR_SIZE   (-2)
#define QTTT_ERR_ACTION (-3)

/* qttt_step / qttt_sample_actions flags */
#define QTTT_FLAG_AUTO_RESET 1u   /* a board whose previous step returned terminated is
                                     re-initialised before the action is applied (the build's
                                     throughput mode; the reference has no auto-reset) */

/* Launch shape carried by the call itself (qttt_step / _observe / _random / _many, qttt_env.flags): boards
 * per lane 1 | 2 | 4 (0 = the library's choice) and workgroup size 256 | 512 | 1024 (0 = the library's
 * choice).  Results never depend on it.  Overrides the process-wide default of qttt_set_tuning. */
#define QTTT_FLAG_SHAPE(boards_per_lane, workgroup_size)                                          \
    ((((uint32_t)(boards_per_lane) & 7u) << 8) |                                                  \
     ((workgroup_size) == 256 ? 1u << 12 : (workgroup_size) == 512 ? 2u << 12 : (workgroup_size) == 1024 ? 3u << 12 : 0u))

#define QTTT_FLAG_FUSED 2u        /* qttt_step_many only: run the n_steps steps in ONE launch per 64 steps with
                                     the boards held in registers (same results; for replay / evaluation
                                     where all actions are known up front) */

int     qttt_abi_version(void);
/* bytes of device memory needed for n boards */
int64_t qttt_state_bytes(int64_t n);

/* Env.reset / Env.__init__ (env.py:55-57, 16-32) -> Board.__init__ (board.py:2-7), n boards */
int qttt_reset(void *state, int64_t n, void *stream);

/* Env.reset INCLUDING the observation it returns (env.py:55-57 -> env.py:68-85 of the empty board), one launch:
 * qttt_reset followed by qttt_observe (classical all -1, q_p1 / q_p2 all 255 with length 0, turn 0).  Buffers as
 * qttt_observe; no alignment is required of them. */
int qttt_reset_observe(void *state, int8_t *classical, uint8_t *q_p1, uint8_t *q_p1_len, uint8_t *q_p2,
                       uint8_t *q_p2_len, uint8_t *turn, int64_t n, void *stream);

/* Env.step (env.py:34-53) for n boards = Board.make_move (board.py:9-25) +
 * Board.update_qstructs (board.py:27-69) + QEvalClassic.eval (qeval.py:5-51) +
 * Board.check_win (board.py:71-115) + reward/terminated (env.py:49,51), one fused kernel.
 *   actions    u8[n,2]   action[0], action[1] per board (env.py:37-38); anything that makes
 *                        make_move raise (a==b, classical square, >8) is a noop
 *   bits       u8[n] or NULL. Stand-in for random.choice at qeval.py:35: 0 -> the closing move
 *                        collapses onto min(a,b), 1 -> onto max(a,b); read only on a collapse.
 *                        NULL -> bit = counter hash of (seed, board_offset+i, step_idx)
 *   reward     f32[n]    -0.0f or -1.0f (env.py:49; sign bit is part of the contract)
 *   terminated u8[n]     env.py:51
 */
int qttt_step(void *state, const uint8_t *actions, const uint8_t *bits, uint64_t seed,
              uint32_t step_idx, int64_t board_offset, uint32_t flags, float *reward,
              uint8_t *terminated, int64_t n, void *stream);

/* Env.step INCLUDING the observation it returns (env.py:46,53), one fused kernel: qttt_step followed
 * by qttt_observe, with the observation written from the registers the step holds (no second read
 * of the state).  Arguments as qttt_step, then as qttt_observe.  q_p1 must be 2-byte and q_p2
 * 8-byte aligned (QTTT_ERR_ACTION otherwise). */
int qttt_step_observe(void *state, const uint8_t *actions, const uint8_t *bits, uint64_t seed,
                      uint32_t step_idx, int64_t board_offset, uint32_t flags, float *reward,
                      uint8_t *terminated, int8_t *classical, uint8_t *q_p1, uint8_t *q_p1_len,
                      uint8_t *q_p2, uint8_t *q_p2_len, uint8_t *turn, int64_t n, void *stream);

/* n_steps consecutive qttt_step launches enqueued back to back from C, so a replay / rollout
 * loop is not paced by the host interpreter.  Step t (0-based) reads actions + t*2n and
 * bits + t*n (when bits != NULL), uses step_idx0 + t, and writes reward + t*out_stride and
 * terminated + t*out_stride (out_stride 0: every step overwrites the same n outputs; with
 * QTTT_FLAG_FUSED only the last step's outputs are then written). */
int qttt_step_many(void *state, const uint8_t *actions, const uint8_t *bits, uint64_t seed,
                   uint32_t step_idx0, int64_t board_offset, uint32_t flags, float *reward,
                   uint8_t *terminated, int64_t out_stride, int64_t n, int32_t n_steps,
                   void *stream);

/* n_steps consecutive qttt_step_random steps (policy -> collapse bit -> step, ply t keyed by the counter
 * hash of (seed, board_offset + i, step_idx0 + t)) in ONE launch per 64 plies (the plies' launch keys travel as a
 * kernel argument; a longer run is ceil(n_steps / 64) launches, same results) with the boards in registers: the
 * policy -> step loop of MCTS._simulate (mcts.py:185-198) as the env's random-policy throughput mode
 * (with QTTT_FLAG_AUTO_RESET a finished board restarts on its next ply).  Bit-identical to n_steps calls
 * of qttt_step_random.  Step t writes actions_out + 2*t*out_stride, reward + t*out_stride and
 * terminated + t*out_stride (out_stride in boards; 0: only the last step's outputs are written, to the
 * first n elements).  actions_out is nullable; reward and terminated are nullable together.
 *   returns f32[n], nullable: ACCUMULATED, returns[i] += the sum of board i's n_steps rewards (each -1.0 or -0.0,
 *   env.py:49) — with auto-reset minus the number of its plies that ended with a completed line: the per-board
 *   episode returns a multi-GPU caller gathers without keeping any per-ply output. */
int qttt_step_random_many(void *state, uint64_t seed, uint32_t step_idx0, int64_t board_offset,
                          uint32_t flags, uint8_t *actions_out, float *reward, uint8_t *terminated,
                          int64_t out_stride, float *returns, int64_t n, int32_t n_steps, void *stream);

/* Env._observation (env.py:68-85) for n boards.
 *   classical i8[n,9]   Board.board (-1 empty else round)
 *   q_p1 u8[n,5,2], q_p1_len u8[n]   un-collapsed even-round moves (lo,hi) in move order, 255 pad
 *   q_p2 u8[n,4,2], q_p2_len u8[n]   un-collapsed odd-round moves
 *   turn u8[n]          len(moves) % 2
 * q_p1 must be 2-byte and q_p2 8-byte aligned (QTTT_ERR_ACTION otherwise).
 */
int qttt_observe(const void *state, int8_t *classical, uint8_t *q_p1, uint8_t *q_p1_len,
                 uint8_t *q_p2, uint8_t *q_p2_len, uint8_t *turn, int64_t n, void *stream);

/* Board.check_win (board.py:71-115): p1_round i8[n], p2_round i8[n] (-1 = no line) */
int qttt_check_win(const void *state, int8_t *p1_round, int8_t *p2_round, int64_t n,
                   void *stream);

/* The Board attributes L3 callers read and assign (board.py:4-6; mcts.py:11-17,241):
 *   moves u8[n,9,2] (255 pad), n_moves u8[n], board i8[n,9], qmask u16[n,4] (qstructs in list
 *   order as 9-bit square masks, 0 pad), n_q u8[n]
 * qttt_export: every output is nullable (only what is asked for is computed and written; n_moves alone
 * is Env.turn, env.py:65-66). */
int qttt_export(const void *state, uint8_t *moves, uint8_t *n_moves, int8_t *board,
                uint16_t *qmask, uint8_t *n_q, int64_t n, void *stream);
int qttt_import(void *state, const uint8_t *moves, const uint8_t *n_moves, const int8_t *board,
                const uint16_t *qmask, const uint8_t *n_q, int64_t n, void *stream);

/* Board.make_move (board.py:9-25), Board.update_qstructs (board.py:27-69) and Board.check_win
 * (board.py:71-115) for callers that hold the reference's Python attributes (the `Board` duck type L3
 * code subclasses, mcts.py:9-17): n records of QTTT_BOARD_RECORD_BYTES in, n out, ONE launch
 * (import, the same step function as qttt_step, export, check_win).  The records only need to be
 * device-accessible: pinned host memory works, and is what the single-board façade uses.
 *   byte  0..17  moves u8[9][2] (255 pad)     18 n_moves     19..27 board i8[9]     28 n_q
 *         29     op (QTTT_OP_*)               30..37 qmask u16[4] little endian
 *         38,39  the move's two squares       40 collapse bit (stand-in for qeval.py:35)
 *   out only:    41 1 = make_move would raise (state unchanged)   44..47 reward f32 (env.py:49)
 *         48     terminated (env.py:51)       49,50 check_win p1_round, p2_round (i8)
 *         63     completion stamp (qttt_board_op_host only)
 * QTTT_OP_MAKE_MOVE: validate, append, entangle / collapse, autofill.  QTTT_OP_UPDATE_QSTRUCTS: the
 * same without the autofill — moves must NOT yet contain the move (the caller appended it: drop it
 * from the record).  QTTT_OP_CHECK_WIN: no move, outputs only. */
#define QTTT_BOARD_RECORD_BYTES 64
#define QTTT_OP_MAKE_MOVE 0
#define QTTT_OP_UPDATE_QSTRUCTS 1
#define QTTT_OP_CHECK_WIN 2
int qttt_board_op(const void *records_in, void *records_out, int64_t n, void *stream);
/* The same followed by hipStreamSynchronize(stream): the one entry point that waits (for a caller
 * whose records live in pinned host memory and are read back right after — the Board façade). */
int qttt_board_op_sync(const void *records_in, void *records_out, int64_t n, void *stream);
/* The same for records that live in HOST-accessible pinned memory (what the Board façade owns): completion is detected
 * by polling a stamp the kernel writes into byte 63 of every out record once the record is visible system-wide —
 * 9.7 us per single-record call instead of 14.7 - 16.0 through hipStreamSynchronize.  For n <= 256 records byte 63 of
 * every out record is cleared by the call and is 1 on return; a poll that has not ended after ~2 ms (e.g. a long kernel
 * queued ahead on the stream) falls back to synchronising the stream — the records are complete and stamped on return
 * either way.  For n > 256 the call does not poll at all: it synchronises the stream, and byte 63 is neither cleared nor
 * stamped (whatever the caller left there stays).  records_out MUST be dereferenceable by the host.
 * n == 1 goes through a BOUNDED MAILBOX instead of a launch: one resident wave on a private non-blocking stream serves a
 * pinned request slot — the record travels as FOUR 16-byte pieces of 12 data bytes + the request number each, the numbers
 * written last, and the wave takes a request only when all four pieces carry the wanted number (the only assumption: one
 * aligned 16-byte device read of pinned host memory is served from one moment of that memory; x86 keeps the host's
 * stores in program order); the answer is polled.  Per call, same box (profiles/r06/facade_latency.json): see DESIGN.md §4;
 * the interpreter running the reference's own Env.step (env.py:34-53) is FASTER than either path on the same host.
 * The wave leaves BY ITSELF after QTTT_BOARD_MAILBOX_US microseconds without a request (default 20, capped at 200; 0
 * disables the mailbox), QTTT_BOARD_MAILBOX_MAX_US after it started whatever the traffic (default 1000, capped at 10000:
 * a thread that keeps calling cannot keep it resident), or when qttt_board_mailbox_retire() asks it to, and is launched
 * again by the next call; a DEVICE-wide synchronise issued while it is resident waits for it to leave (<= the idle
 * window after the last call; <= the residency bound when another thread keeps calling), stream-level synchronisation
 * does not.  `stream` is not used on this path (host records have no device-side producer to be ordered after); results
 * are identical on both paths. */
int qttt_board_op_host(const void *records_in, void *records_out, int64_t n, void *stream);
/* Asks the mailbox wave of qttt_board_op_host to leave NOW (the host writes a "leave" request into the slot; the wave
 * exits at its next poll, a few microseconds; a no-op when none is resident — one relaxed atomic load).  wait != 0: returns
 * once the wave has said it left (bounded: 2 ms), so that a device-wide synchronise issued next has nothing of this
 * library to wait for.  The next single-record call simply launches the wave again.  The step entries (qttt_step,
 * _observe, _random, _many, _random_many, qttt_env_step) call it themselves (wait = 0) for batches of >= 512 K boards:
 * a launch that fills the chip would otherwise run a second partial round beside the resident wave (+1.4 us at 1 M
 * boards).  For a caller that mixes a real Board with batched search, as strat_eval.py:34-63 does. */
int qttt_board_mailbox_retire(int wait);

/* Synthetic policy for measurement: uniform over legal unordered pairs
 * (GameState.actions rule, mcts.py:20-27) in ind2move order (mcts.py:339-343), index and
 * collapse bit from the counter hash of (seed, board_offset+i, step_idx).  actions u8[n,2]. */
int qttt_sample_actions(const void *state, uint64_t seed, uint32_t step_idx,
                        int64_t board_offset, uint32_t flags, uint8_t *actions, int64_t n,
                        void *stream);

/* ---- next rows: the callers either side of the path -------------------------- */

/* What an MCTS node needs about a state (GameState, mcts.py:9-94):
 *   winner i8[n]   1 = True (p1), 0 = False (p2), -1 = None   (update_winner, mcts.py:52-65)
 *   terminal u8[n] line or len(moves) == 9                     (mcts.py:65)
 *   legal u64[n]   bit a = action a in GameState.actions       (mcts.py:20-27, ind2move order)
 *   key i64[n]     GameState.__hash__ (mcts.py:93-94) = CPython (>= 3.8) hash(tuple(board) +
 *                  tuple(moves)), bit-exact, so keys match a host-side transposition table built by
 *                  reference code (~3/4 of the kernel's work when asked for)
 *   state_key u64[n]  the native position key = qttt_state_key() of the board's two packed words: the
 *                  reference only ever uses __hash__ / __eq__ as dict keys (mcts.py:93-97,160-164,210-221),
 *                  i.e. needs "equal keys <=> equal (board, moves)", which this key gives at a twentieth
 *                  of the arithmetic — for transposition tables that live on the device
 * Every output is nullable; only what is asked for is computed. */
int qttt_node_info(const void *state, int8_t *winner, uint8_t *terminal, uint64_t *legal,
                   int64_t *key, uint64_t *state_key, int64_t n, void *stream);

/* The native position key of one board from its packed words P[i], Q[i] (host-callable, no device work).
 * Equal for two boards iff their (board, moves) are equal (up to 64-bit collisions): the packed state is a
 * canonical form of (board, moves) whether it was reached by stepping or written by qttt_import; the cached
 * qstructs (list order is not part of a position) and the done bit are left out of the mix. */
uint64_t qttt_state_key(uint64_t plane_p_word, uint64_t plane_q_word);

/* MCTS._step (mcts.py:233-267) for n (state, action) pairs, both collapse branches at once.
 *   action36 u8[n]    index into ind2move order (mcts.py:339-343); > 35 is illegal
 *   child0, child1    packed states, qttt_state_bytes(n) each (child 1 meaningful when n_children == 2)
 *   n_children u8[n]  0 = make_move raises (children = copies of the parent), 1 = no collapse,
 *                     2 = collapse: child 0 = closing move on min(a,b) (bit 0), child 1 = on max(a,b)
 *   winner i8[n,2], terminal u8[n,2], legal u64[n,2], key i64[n,2], state_key u64[n,2]: as qttt_node_info,
 *                     per child (-1 / 0 / 0 / 0 / 0 for a child that does not exist); each nullable
 * The reference returns the two children in random order; compare as a set.  winner / terminal must be
 * 2-byte and legal / key / state_key 16-byte aligned (one vector store per pair of children;
 * QTTT_ERR_ACTION otherwise). */
int qttt_expand(const void *state, const uint8_t *action36, void *child0, void *child1,
                uint8_t *n_children, int8_t *winner, uint8_t *terminal, uint64_t *legal,
                int64_t *key, uint64_t *state_key, int64_t n, void *stream);

/* MCTS._simulate (mcts.py:185-198) with the uniform priors of mcts.py:287-292: uniform-legal random
 * playout to the end with the board in registers; ply p draws action and collapse branch from the
 * counter hash of (seed, board_offset+i, step_idx0+p), i.e. exactly what qttt_sample_actions +
 * qttt_step would do launch by launch.
 *   result i8[n]   MCTS._reward (mcts.py:200-209): +1 winner True, -1 winner False, 0 None
 *   plies u8[n]    moves played; final_state (nullable): packed end states */
int qttt_rollout(const void *state, uint64_t seed, uint32_t step_idx0, int64_t board_offset,
                 int8_t *result, uint8_t *plies, void *final_state, int64_t n, void *stream);

/* MCTS._rollout's simulation loop (mcts.py:170-176: num_simulations playouts of one leaf) for n boards in ONE launch:
 * result[i, s] and plies[i, s] (nullable) are what qttt_rollout(step_idx0 + s * QTTT_SIM_STRIDE) gives for board i,
 * s = 0 .. n_sims - 1; one lane per (board, simulation), so 65 536 leaves x 10 simulations fill the chip.
 *   result i8[n, n_sims], plies u8[n, n_sims] */
#define QTTT_SIM_STRIDE 16u      /* a playout has at most 9 plies: simulations use disjoint step indices */
int qttt_rollout_many(const void *state, uint64_t seed, uint32_t step_idx0, int64_t board_offset,
                      int32_t n_sims, int8_t *result, uint8_t *plies, int64_t n, void *stream);

/* One MCTS._rollout below the selected node in ONE launch (mcts.py:166-176: select -> _expand_child -> num_simulations
 * x _simulate from the leaf -> the value handed to _backpropogate; :210-221,233-267): qttt_expand of the n
 * (state, action) pairs — same outputs, child0 / child1 nullable here too — and n_sims playouts from EACH child:
 * exactly qttt_rollout_many(child0, seed, step_idx0, board_offset, n_sims) and
 * qttt_rollout_many(child1, seed, step_idx0 + n_sims * QTTT_SIM_STRIDE, board_offset, n_sims).
 *   value_sum i32[n,2]        sum over the child's simulations of `r if leaf.turn else -r` (mcts.py:174); divide by
 *                             n_sims for the value of mcts.py:176.  leaf.turn is True after an even number of real
 *                             moves (MCTS.reset's len(moves) % 2 == 0, flipped once per _step: mcts.py:140,243).
 *                             0 for a child that does not exist
 *   result i8[n,2,n_sims]     nullable: every simulation's MCTS._reward (0 for a child that does not exist)
 * 1 <= n_sims <= QTTT_EXPAND_ROLLOUT_MAX_SIMS (a workgroup owns whole pairs; the reference's default is 10). */
#define QTTT_EXPAND_ROLLOUT_MAX_SIMS 128
int qttt_expand_rollout(const void *state, const uint8_t *action36, void *child0, void *child1,
                        uint8_t *n_children, int8_t *winner, uint8_t *terminal, uint64_t *legal,
                        int64_t *key, uint64_t *state_key, uint64_t seed, uint32_t step_idx0,
                        int64_t board_offset, int32_t n_sims, int32_t *value_sum, int8_t *result,
                        int64_t n, void *stream);

/* GameState.to_vector (mcts.py:67-85) -> vec f32[n,18,10]; action_mask (mcts.py:87-91) ->
 * mask u8[n,36] (nullable).  The reference builds float64; values are 0, 1 and 1/3 rounded to f32. */
int qttt_encode(const void *state, float *vec, uint8_t *mask, int64_t n, void *stream);

/* The buffers of one batch of boards as one caller-owned plain record, so that a per-step host loop
 * passes 6 arguments instead of 11 - 17 (a Python / ctypes caller is host-bound below ~500 K boards:
 * DESIGN.md §6).  The library keeps nothing: the record is read during the call only. */
typedef struct qttt_env {
    void    *state;                 /* qttt_state_bytes(n) bytes */
    int64_t  n;
    int64_t  board_offset;
    uint64_t seed;
    uint32_t flags;                 /* QTTT_FLAG_AUTO_RESET | QTTT_FLAG_SHAPE(...) */
    uint32_t reserved;              /* 0 */
    float   *reward;                /* f32[n] */
    uint8_t *terminated;            /* u8[n] */
    int8_t  *classical;             /* the qttt_observe outputs; only read by QTTT_ENV_STEP_OBSERVE */
    uint8_t *q_p1, *q_p1_len, *q_p2, *q_p2_len, *turn;
    const uint32_t *step_counter;   /* nullable DEVICE u32: the step index a launch uses is step_idx + *step_counter,
                                       read when the kernel runs — so qttt_env_step calls captured in a hipGraph can be
                                       replayed (a graph node cannot carry a host-side step counter): capture node t with
                                       step_idx = t and end the graph with qttt_counter_add(step_counter, T).  Such launches
                                       run in one shape (one board per lane, 256-thread workgroups: graphs are for small,
                                       launch-bound batches); calls with explicit bits need no step index and are unaffected */
} qttt_env;
#define QTTT_ENV_STEP         0     /* = qttt_step(actions, bits) */
#define QTTT_ENV_STEP_OBSERVE 1     /* = qttt_step_observe(actions, bits) */
#define QTTT_ENV_STEP_RANDOM  2     /* = qttt_step_random(actions_out = actions, nullable); bits ignored */
#define QTTT_ENV_SAMPLE       3     /* = qttt_sample_actions(actions): the policy alone, the state is not touched */
int qttt_env_step(const qttt_env *env, uint8_t *actions, const uint8_t *bits, uint32_t step_idx,
                  int mode, void *stream);
/* *counter += by, on the stream (one lane): advances a qttt_env.step_counter; capturable like every other entry */
int qttt_counter_add(uint32_t *counter, uint32_t by, void *stream);

/* Process-wide DEFAULT launch shape of qttt_step / qttt_step_observe / qttt_step_random (results never
 * depend on it): boards per lane (1, 2 or 4) and workgroup size (256, 512 or 1024); 0 = chosen by the
 * library from the batch size.  With only boards_per_lane given the workgroup size stays
 * the library's choice for the batch (512 for four boards per lane, the only size that exists).  Also
 * settable through QTTT_STEP_BPL / QTTT_STEP_BLOCK before the first call.  One atomic word: safe to call
 * while other threads launch.  Prefer QTTT_FLAG_SHAPE on the call. */
int qttt_set_tuning(int boards_per_lane, int workgroup_size);
/* The shape a batch of n boards is launched with by a call with these flags (`observe` != 0: by
 * qttt_step_observe, whose tiles take at most two boards per lane).  A caller whose pointers are not
 * aligned for boards_per_lane elements gets fewer boards per lane. */
int qttt_step_launch_shape(int64_t n, uint32_t flags, int observe, int *boards_per_lane, int *workgroup_size);

/* One step of every board under that policy with policy and step in ONE kernel: exactly
 * qttt_sample_actions followed by qttt_step(bits = NULL) with the same seed / step_idx /
 * board_offset / flags.  actions_out u8[n,2] receives the actions played (nullable). */
int qttt_step_random(void *state, uint64_t seed, uint32_t step_idx, int64_t board_offset,
                     uint32_t flags, uint8_t *actions_out, float *reward, uint8_t *terminated,
                     int64_t n, void *stream);

/* The counter hash itself (host-callable, no device work), so callers can reproduce bits. */
uint64_t qttt_hash(uint64_t seed, uint64_t board_id, uint32_t step_idx);

#ifdef __cplusplus
}
#endif
#endif
