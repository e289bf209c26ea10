// qttt_kernels.hip — gfx950 (MI355X / CDNA4) kernels + the C ABI of include/qttt.h.
//
// Mapping: ONE LANE PER BOARD (64 boards per wavefront, BPL consecutive boards per lane),
// everything in VGPRs, structure-of-arrays state so that every load/store of a wave is one
// contiguous 16-byte-per-lane segment.  No MFMA; LDS holds one lookup table (and, in the kernels
// that return the observation, the output tiles).  DESIGN.md §2 explains why the wave-per-board
// mapping was rejected after measurement and why the kernel is written for minimum VALU
// *instruction count* (measured issue cost ~4 cycles per wave-instruction for this instruction mix,
// tools/valu_rates.cpp).
//
// Formulation — deliberately NOT the reference's algorithm:
//   * the un-collapsed moves of a board form a forest on the 9 squares (a move that closes a
//     cycle collapses its whole component at once, board.py:42-56).  The forest is kept ROOTED:
//     nibble sq[v] of a non-classical square v is the round of the move joining v to its parent
//     (root / isolated = none).  For a classical square, sq[v] is the round that landed there
//     (= Board.board[v]).
//   * QEvalClassic.eval (qeval.py:5-51: leaf-peel + forced walk round the cycle) is equivalent
//     to: re-root the tree at the square t the closing move lands on (bit picks lo/hi), then
//     every other square of the component receives its parent edge.  So a collapse is one path
//     reversal + `classical |= component`; no per-edge work.
//   * Every move ever played is therefore HELD by exactly one square c (sq[c] = its round): the
//     child end of an un-collapsed move, the landing square of a collapsed one.  The move itself
//     is then (c, c ^ x) with x = lo ^ hi, so the state stores only the 4-bit x of each move — the
//     re-rooting walk needs nothing else ("other end of edge e" = v ^ x_e) and the cold kernels
//     rebuild Board.moves from the holders.
//   * Board.qstructs (board.py:6) is cached as 4 slots x 9-bit square masks, in the reference's
//     list order, so "same component?" is two shifts and an AND.
//
// Packed state, 16 B/board = 39 algorithmic bytes per step, planes
// P[s] u64 | Q[s] u64 (s = n rounded up to 64).  The all-zero state is the empty board.
//   P bits [2,38)  nine nibbles, square v at bits [4v+2, 4v+6), COMPLEMENT-coded: 0 = root /
//                  isolated / empty, round e is stored as 15-e.  The 2-bit offset makes
//                  `(P >> 4v) & 0x3C` the code times four, the unit every shift amount below wants.
//   P1 = P >> 32:  [0,6) nibbles | [6,8) 0 | [8,12) n = moves PLAYED | [12,16) comps bits 32..35 |
//                  [16,20) x of the last move | [20,22) 0 | [22,31) classical mask | 31 done
//   Q0:            x = lo^hi of the moves of rounds 0..7: round e in the nibble at bit
//                  (4(7-e)+2) mod 32, so that rotating Q0 right by four times the CODE of e
//                  (4(15-e) = 4(7-e) mod 32) lands 4x on bits 2..5.  The move of round 8 can only
//                  be the last one of a game: its x is the `last x` field of P1 (it is also XORed
//                  onto round 0's nibble, where it is harmless: the game is over; the cold
//                  kernels undo it).
//   Q1:            comps bits 0..31 (comps = 4 x 9-bit masks, slot k at bit 9k, list order, compact)
//   The autofill of board.py:22-25 is IMPLICIT: a board with exactly 8 classical squares stands
//   for the reference state in which the 9th square holds round 8 and moves ends with (idx,idx,8)
//   (the autofill round is always 8); the cold kernels materialise it.
//
// Files: qttt_state.h (layout, loads/stores, shared tables) -> qttt_step_core.h (the step) ->
// qttt_observation.h -> qttt_step_kernels.h; qttt_board_forms.h (unpacked views, winner, legal mask,
// tuple hash) -> qttt_aux_kernels.h, qttt_mcts_kernels.h; this file: launch logic + the C ABI.
#include <atomic>
#include <chrono>
#include <cstring>
#include <mutex>
#include "qttt_step_kernels.h"
#include "qttt_aux_kernels.h"
#include "qttt_mcts_kernels.h"

namespace {

// Boards per lane and workgroup size of the step kernel, by batch size.  Measured on MI355X with
// tools/stepbench (interleaved A/B, profiles/r02/stepbench_block_sweep.txt), us per launch:
//   boards      (1,256) (1,1024) (2,256) (2,512) (2,1024)
//   131 072      3.11    3.41     3.36    3.46    4.25
//   262 144      3.63    3.66     3.90    3.86    4.62
//   393 216      4.38    4.74     4.61    4.92    4.83
//   524 288      5.16    4.93     5.30    5.23    5.08
//   786 432      7.29    6.75     6.59    6.27    6.95
//   1 048 576    8.84    8.72     7.68    7.37    7.25
//   1 572 864   12.05   12.75    11.56   11.78   11.39
//   2 097 152   14.38   16.10    13.53   13.64   14.08
//   4 194 304   28.40   30.52    27.51   27.78   29.12
//   16 777 216  103.2   109.0    105.0   107.4   108.5
// Below ~450 K boards the launch is latency-bound and one board per lane in small workgroups puts the
// most waves in flight; 1024-thread workgroups win where they fill the chip exactly once (512 K lanes =
// 2 workgroups on each of the 256 CUs); past that, small workgroups backfill best.
inline void auto_tuning(int64_t n, int &bpl, int &blk) {
    if (n <= 448 * 1024) { bpl = 1; blk = 256; }
    else if (n <= 512 * 1024) { bpl = 1; blk = 1024; }
    else if (n < 896 * 1024) { bpl = 2; blk = 512; }
    else if (n <= 1536 * 1024) { bpl = 2; blk = 1024; }
    else { bpl = 2; blk = 256; }
}
// Process-wide DEFAULT launch shape (bench / profiling): boards per lane 1|2|4 and workgroup size
// 256|512|1024, 0 = by batch size.  Initialised from QTTT_STEP_BPL / QTTT_STEP_BLOCK, changeable through
// qttt_set_tuning(); one relaxed atomic word (bpl | block << 8), so concurrent callers never race on it.
// A call that carries QTTT_FLAG_SHAPE(...) in its flags does not look at it at all.
inline std::atomic<int> &tuning_word() {
    static std::atomic<int> v([] {
        int bpl = 0, blk = 0;
        if (const char *e = getenv("QTTT_STEP_BPL")) { int q = atoi(e); if (q == 1 || q == 2 || q == 4) bpl = q; }
        if (const char *e = getenv("QTTT_STEP_BLOCK")) { int q = atoi(e); if (q == 256 || q == 512 || q == 1024) blk = q; }
        return bpl | (blk << 8);
    }());
    return v;
}
// the shape one call is launched with: the call's own QTTT_FLAG_SHAPE bits, else the process default,
// else the table; `observe`: the observation tiles are sized for <= 2 boards per lane
inline void resolve_shape(int64_t n, uint32_t flags, bool observe, int &bpl, int &blk) {
    int f_bpl = (int)((flags >> 8) & 7u), f_blk = 0;
    switch ((flags >> 12) & 3u) { case 1: f_blk = 256; break; case 2: f_blk = 512; break; case 3: f_blk = 1024; break; default: break; }
    if (f_bpl != 1 && f_bpl != 2 && f_bpl != 4) f_bpl = 0;
    if (!f_bpl && !f_blk) {
        const int w = tuning_word().load(std::memory_order_relaxed);
        f_bpl = w & 0xFF;
        f_blk = w >> 8;
    }
    auto_tuning(n, bpl, blk);
    if (f_bpl) bpl = f_bpl;
    if (f_blk) blk = f_blk;
    if (observe && bpl > 2) bpl = 2;
    if (bpl == 4) blk = QTTT_BLOCK;                      // four boards per lane exist with 512 threads only
}

inline int grid_for(int64_t n) { return (int)((n + QTTT_BLOCK - 1) / QTTT_BLOCK); }
inline int cold_grid_for(int64_t n) { return (int)((n + QTTT_COLD_BLOCK - 1) / QTTT_COLD_BLOCK); }
inline int blocks_for(int64_t n_groups, int block) { return (int)((n_groups + block - 1) / block); }

inline int launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

// A hint: the single-record mailbox wave (board_mailbox, below) MAY be resident.  It holds one wave slot of one CU, so a
// launch that fills the chip exactly runs a second partial round beside it (+1.4 us at 1 M boards,
// profiles/r05/keepwarm_probe.txt): such launches ask it to leave first (it is gone within one poll).  One relaxed load
// per launch when no wave is resident.  (What is NOT done from here: querying the mailbox's stream so that the runtime
// retires the finished kernel.  A finished mailbox kernel nobody has queried leaves the launches of other streams
// 0.05 - 0.4 us longer for a while — tools/probes/mailbox_rest_delta_probe.py — and one hipStreamQuery after the wave has
// said it left removes most of that, which qttt_board_mailbox_retire(1) does; but the query returns "not ready" for a few
// microseconds after the wave's last store, and repeating it from the launch path cost the launches 0.4 - 1.0 us each:
// measured, profiles/r06/mailbox_rest_delta_probe_query_from_the_launch_path.txt, not adopted.)
std::atomic<bool> g_mailbox_resident{false};
constexpr int64_t CHIP_FILLING_BOARDS = 512 * 1024;
void mailbox_housekeeping(hipStream_t user_stream);     // (defined beside BoardMailbox)
inline void retire_mailbox_for(int64_t n, void *stream) {
    if (n >= CHIP_FILLING_BOARDS && g_mailbox_resident.load(std::memory_order_relaxed)) mailbox_housekeeping((hipStream_t)stream);
}

}  // namespace

// ====================================================================== C ABI
extern "C" {

int qttt_abi_version(void) { return QTTT_ABI_VERSION; }

#ifdef QTTT_DEBUG_STAMPS
int qttt_debug_set_stamps(void *buf) {
    u64 *p = (u64 *)buf;
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_debug_stamps), &p, sizeof(p));
}
#endif

int qttt_set_tuning(int boards_per_lane, int workgroup_size) {
    if (!(boards_per_lane == 0 || boards_per_lane == 1 || boards_per_lane == 2 || boards_per_lane == 4)) return QTTT_ERR_SIZE;
    if (!(workgroup_size == 0 || workgroup_size == 256 || workgroup_size == 512 || workgroup_size == 1024)) return QTTT_ERR_SIZE;
    tuning_word().store(boards_per_lane | (workgroup_size << 8), std::memory_order_relaxed);
    return 0;
}

int qttt_step_launch_shape(int64_t n, uint32_t flags, int observe, int *boards_per_lane, int *workgroup_size) {
    if (n < 0) return QTTT_ERR_SIZE;
    if (!boards_per_lane || !workgroup_size) return QTTT_ERR_NULL;
    resolve_shape(n, flags, observe != 0, *boards_per_lane, *workgroup_size);
    return 0;
}

int64_t qttt_state_bytes(int64_t n) { return n < 0 ? (int64_t)QTTT_ERR_SIZE : plane_stride(n) * QTTT_STATE_BYTES; }

uint64_t qttt_hash(uint64_t seed, uint64_t board_id, uint32_t step_idx) {
    const u64 key = launch_key(seed, step_idx);
    const u32 h1 = lowbias32(fold_id(board_id) ^ (u32)key);
    const u32 h2 = lowbias32(h1 ^ (u32)(key >> 32));
    return ((u64)h2 << 32) | h1;
}

}  // extern "C"
namespace {
// the empty board is the all-zero state: 16 bytes of zeros per lane, with the step kernel's own
// non-temporal stores
__global__ __launch_bounds__(256) void reset_kernel(u32x4 *state, int64_t n16) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n16) __builtin_nontemporal_store(u32x4{0u, 0u, 0u, 0u}, &state[i]);
}
}  // namespace
extern "C" {

// Env.reset INCLUDING the observation it returns (env.py:55-57,68-85): the empty board's observation is constant
// (classical -1, no quantum states: 255 pad and length 0, turn 0), so state and observation are seven byte fills in
// one launch.  The seven buffers' 16-byte pieces are numbered through (first[k] = pieces in front of buffer k): every
// thread of the grid stores one piece, non-temporally; the unaligned head / tail of a caller's odd pointer is written
// bytewise by the first workgroup.
namespace {
struct FillSegs {
    uint8_t *p[7];
    int64_t bytes[7];
    int64_t first[8];                                        // prefix sums of the 16-byte piece counts
    u32 word[7];                                             // the fill byte, four times
};
__device__ __forceinline__ int64_t fill_head(const uint8_t *p, int64_t nb) {
    const int64_t h = (int64_t)((16u - (u32)(reinterpret_cast<uintptr_t>(p) & 15u)) & 15u);
    return h < nb ? h : nb;
}
__global__ __launch_bounds__(256) void reset_observe_kernel(FillSegs f) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g < f.first[7]) {
        int k = 0;
#pragma unroll
        for (int j = 1; j < 7; ++j) k += g >= f.first[j] ? 1 : 0;
        const u32 w = f.word[k];
        uint8_t *p = f.p[k];
        __builtin_nontemporal_store(u32x4{w, w, w, w}, reinterpret_cast<u32x4 *>(p + fill_head(p, f.bytes[k])) + (g - f.first[k]));
    }
    if (blockIdx.x == 0 && threadIdx.x < 7 * 32) {           // 16 head + 16 tail bytes per buffer
        const int k = threadIdx.x >> 5;
        const int64_t j = threadIdx.x & 15;
        uint8_t *p = f.p[k];
        const int64_t nb = f.bytes[k], head = fill_head(p, nb), nvec = f.first[k + 1] - f.first[k];
        if ((threadIdx.x & 31) < 16) { if (j < head) p[j] = (uint8_t)f.word[k]; }
        else { const int64_t off = head + (nvec << 4) + j; if (off < nb) p[off] = (uint8_t)f.word[k]; }
    }
}
}  // namespace

int qttt_reset_observe(void *state, int8_t *classical, uint8_t *q_p1, uint8_t *q_p1_len, uint8_t *q_p2,
                       uint8_t *q_p2_len, uint8_t *turn, int64_t n, void *stream) {
    if (n < 0) return QTTT_ERR_SIZE;
    if (n == 0) return 0;
    if (!state || !classical || !q_p1 || !q_p1_len || !q_p2 || !q_p2_len || !turn) return QTTT_ERR_NULL;
    FillSegs f;
    const int64_t sb = plane_stride(n) * QTTT_STATE_BYTES;
    uint8_t *ptrs[7] = {static_cast<uint8_t *>(state), reinterpret_cast<uint8_t *>(classical), q_p1, q_p1_len, q_p2, q_p2_len, turn};
    const int64_t bytes[7] = {sb, 9 * n, 10 * n, n, 8 * n, n, n};
    const u32 words[7] = {0u, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0xFFFFFFFFu, 0u, 0u};
    f.first[0] = 0;
    for (int k = 0; k < 7; ++k) {
        f.p[k] = ptrs[k]; f.bytes[k] = bytes[k]; f.word[k] = words[k];
        int64_t head = (int64_t)((16u - (u32)(reinterpret_cast<uintptr_t>(ptrs[k]) & 15u)) & 15u);
        if (head > bytes[k]) head = bytes[k];
        f.first[k + 1] = f.first[k] + ((bytes[k] - head) >> 4);
    }
    const unsigned gx = (unsigned)((f.first[7] + 255) / 256);
    hipLaunchKernelGGL(reset_observe_kernel, dim3(gx ? gx : 1u), dim3(256), 0, (hipStream_t)stream, f);
    return launch_status();
}

int qttt_reset(void *state, int64_t n, void *stream) {
    if (n < 0) return QTTT_ERR_SIZE;
    if (n == 0) return 0;
    if (!state) return QTTT_ERR_NULL;
#ifdef QTTT_RESET_MEMSET                      // (A/B builds only: hipMemsetAsync, as up to round 4)
    hipError_t e = hipMemsetAsync(state, 0, (size_t)(plane_stride(n) * QTTT_STATE_BYTES), (hipStream_t)stream);
    return e == hipSuccess ? 0 : (int)e;
#else
    const int64_t n16 = plane_stride(n) * QTTT_STATE_BYTES / 16;
    hipLaunchKernelGGL(reset_kernel, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       static_cast<u32x4 *>(state), n16);
    return launch_status();
#endif
}

static int launch_step(void *state, uint8_t *actions, const uint8_t *bits, uint64_t seed,
                       uint32_t step_idx, int64_t board_offset, uint32_t flags, float *reward,
                       uint8_t *terminated, int64_t n, void *stream, bool sample, const ObsOut *obs,
                       const uint32_t *step_ctr = nullptr) {
    if (n < 0 || board_offset < 0) return QTTT_ERR_SIZE;
    if (n == 0) return 0;
    if (!state || !reward || !terminated || (!sample && !actions)) return QTTT_ERR_NULL;
    if ((uintptr_t)actions & 1u) return QTTT_ERR_ACTION;   // actions are accessed as u16 pairs
    retire_mailbox_for(n, stream);
    Planes p = planes(state, n);
    // with a device-side step counter the kernel makes the key itself: it gets the offset and the id fold
    const u64 key = step_ctr ? ((u64)step_idx << 32) : launch_key(seed, step_idx);
    const u32 key_lo = (u32)key, key_hi = (u32)(key >> 32);
    hipStream_t s = (hipStream_t)stream;
    uint16_t *a16 = reinterpret_cast<uint16_t *>(actions);
    u32 *rb = reinterpret_cast<u32 *>(reward);
    const bool ar = (flags & QTTT_FLAG_AUTO_RESET) != 0;
    int bpl_max, blk_sel;
    resolve_shape(n, flags, obs != nullptr, bpl_max, blk_sel);
    // widest boards-per-lane the caller's pointers are aligned for (the planes always are)
    auto aligned = [&](int k) {
        return ((uintptr_t)actions % (2u * k)) == 0 && ((uintptr_t)reward % (4u * k)) == 0 &&
               ((uintptr_t)terminated % (unsigned)k) == 0 && (!bits || ((uintptr_t)bits % (unsigned)k) == 0);
    };
    while (bpl_max > 1 && !aligned(bpl_max)) bpl_max >>= 1;
    const ObsOut oo = obs ? *obs : ObsOut{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
#define QTTT_LAUNCH_B(BLK, BPL, HB, AR, SM, OB, I0, NG, KF, IDB)                                        \
    hipLaunchKernelGGL((step_kernel<BLK, BPL, HB, AR, SM, OB>), dim3(blocks_for(NG, BLK)), dim3(BLK), 0, s, \
                       p.P, p.Q, a16, bits, (u32)(KF), key_hi, (u32)(IDB), rb, terminated, oo,         \
                       (int64_t)(I0), (u32)((NG) - (int64_t)(blocks_for(NG, BLK) - 1) * (BLK)), StepKeySource<false>{})
    // workgroup size as chosen above; four boards per lane exists with 512 threads only
#define QTTT_LAUNCH(BPL, HB, AR, SM, OB, I0, NG, KF, IDB)                                              \
    do {                                                                                              \
        if ((BPL) == 4 || blk_sel == QTTT_BLOCK) QTTT_LAUNCH_B(QTTT_BLOCK, BPL, HB, AR, SM, OB, I0, NG, KF, IDB); \
        else if (blk_sel == 1024) QTTT_LAUNCH_B(1024, ((BPL) == 4 ? 2 : (BPL)), HB, AR, SM, OB, I0, NG, KF, IDB); \
        else QTTT_LAUNCH_B(256, ((BPL) == 4 ? 2 : (BPL)), HB, AR, SM, OB, I0, NG, KF, IDB);             \
    } while (0)
#define QTTT_DISPATCH(BPL, I0, NG, KF, IDB)                                                           \
    do {                                                                                              \
        if (obs) {                                                                                    \
            if (bits) { if (ar) QTTT_LAUNCH(BPL, true, true, false, true, I0, NG, KF, IDB); else QTTT_LAUNCH(BPL, true, false, false, true, I0, NG, KF, IDB); } \
            else { if (ar) QTTT_LAUNCH(BPL, false, true, false, true, I0, NG, KF, IDB); else QTTT_LAUNCH(BPL, false, false, false, true, I0, NG, KF, IDB); } \
        }                                                                                             \
        else if (sample) { if (ar) QTTT_LAUNCH(BPL, false, true, true, false, I0, NG, KF, IDB); else QTTT_LAUNCH(BPL, false, false, true, false, I0, NG, KF, IDB); } \
        else if (bits) { if (ar) QTTT_LAUNCH(BPL, true, true, false, false, I0, NG, KF, IDB); else QTTT_LAUNCH(BPL, true, false, false, false, I0, NG, KF, IDB); } \
        else { if (ar) QTTT_LAUNCH(BPL, false, true, false, false, I0, NG, KF, IDB); else QTTT_LAUNCH(BPL, false, false, false, false, I0, NG, KF, IDB); } \
    } while (0)
    // Device-side step counter (graph capture: small, launch-bound batches): one launch shape — one board per
    // lane, 256-thread workgroups — and the kernels that make the launch key themselves.
    const StepKeySource<true> sk = {step_ctr, (u64)seed};
#define QTTT_LAUNCH_DEV(AR, SM, OB, I0, NG, KF, IDB)                                                    \
    hipLaunchKernelGGL((step_kernel<256, 1, false, AR, SM, OB, true>), dim3(blocks_for(NG, 256)), dim3(256), 0, s, \
                       p.P, p.Q, a16, bits, (u32)(KF), key_hi, (u32)(IDB), rb, terminated, oo,         \
                       (int64_t)(I0), (u32)((NG) - (int64_t)(blocks_for(NG, 256) - 1) * 256), sk)
#define QTTT_DISPATCH_DEV(I0, NG, KF, IDB)                                                            \
    do {                                                                                              \
        if (obs) { if (ar) QTTT_LAUNCH_DEV(true, false, true, I0, NG, KF, IDB); else QTTT_LAUNCH_DEV(false, false, true, I0, NG, KF, IDB); } \
        else if (sample) { if (ar) QTTT_LAUNCH_DEV(true, true, false, I0, NG, KF, IDB); else QTTT_LAUNCH_DEV(false, true, false, I0, NG, KF, IDB); } \
        else { if (ar) QTTT_LAUNCH_DEV(true, false, false, I0, NG, KF, IDB); else QTTT_LAUNCH_DEV(false, false, false, I0, NG, KF, IDB); } \
    } while (0)
    // The hash folds the global board id as lo32 ^ hi32*C (fold_id).  hi32 is uniform over a
    // range of boards unless the range crosses a multiple of 2^32; the batch is cut there (at most
    // once), so the kernel only ever adds a lane index to a 32-bit base.
    int64_t seg_begin = 0;
    while (seg_begin < n) {
        const u64 first = (u64)board_offset + (u64)seg_begin;
        const u64 to_boundary = (((first >> 32) + 1u) << 32) - first;
        const int64_t seg_n = (int64_t)((u64)(n - seg_begin) < to_boundary ? (u64)(n - seg_begin) : to_boundary);
        const u32 key_fold = key_lo ^ ((u32)(first >> 32) * 0x9E3779B9u);
        const u32 id_base = (u32)first;
        if (step_ctr && !bits) {                                 // (explicit bits need no key: the ordinary kernels do)
            QTTT_DISPATCH_DEV(seg_begin, seg_n, key_fold, id_base);
            seg_begin += seg_n;
            continue;
        }
        int bpl = bpl_max;
        while (bpl > 1 && (seg_begin % bpl) != 0) bpl >>= 1;     // vector accesses need an aligned start
        const int64_t n_groups = seg_n / bpl, n_main = n_groups * bpl;
        if (n_groups > 0) {
            if (bpl == 4) QTTT_DISPATCH(4, seg_begin, n_groups, key_fold, id_base);
            else if (bpl == 2) QTTT_DISPATCH(2, seg_begin, n_groups, key_fold, id_base);
            else QTTT_DISPATCH(1, seg_begin, n_groups, key_fold, id_base);
        }
        if (n_main < seg_n)                                      // ragged tail, one board per lane
            QTTT_DISPATCH(1, seg_begin + n_main, seg_n - n_main, key_fold, id_base + (u32)n_main);
        seg_begin += seg_n;
    }
#undef QTTT_DISPATCH_DEV
#undef QTTT_LAUNCH_DEV
#undef QTTT_DISPATCH
#undef QTTT_LAUNCH
#undef QTTT_LAUNCH_B
    return launch_status();
}

int qttt_step(void *state, const uint8_t *actions, const uint8_t *bits, uint64_t seed,
              uint32_t step_idx, int64_t board_offset, uint32_t flags, float *reward,
              uint8_t *terminated, int64_t n, void *stream) {
    return launch_step(state, const_cast<uint8_t *>(actions), bits, seed, step_idx, board_offset, flags,
                       reward, terminated, n, stream, false, nullptr);
}

int qttt_step_observe(void *state, const uint8_t *actions, const uint8_t *bits, uint64_t seed,
                      uint32_t step_idx, int64_t board_offset, uint32_t flags, float *reward,
                      uint8_t *terminated, int8_t *classical, uint8_t *q_p1, uint8_t *q_p1_len,
                      uint8_t *q_p2, uint8_t *q_p2_len, uint8_t *turn, int64_t n, void *stream) {
    if (n > 0 && (!classical || !q_p1 || !q_p1_len || !q_p2 || !q_p2_len || !turn)) return QTTT_ERR_NULL;
    if (((uintptr_t)q_p1 & 1u) || ((uintptr_t)q_p2 & 7u)) return QTTT_ERR_ACTION;   // 2- / 8-byte LDS row stores
    const ObsOut o = {classical, q_p1, q_p1_len, q_p2, q_p2_len, turn};
    return launch_step(state, const_cast<uint8_t *>(actions), bits, seed, step_idx, board_offset, flags,
                       reward, terminated, n, stream, false, &o);
}

int qttt_step_random(void *state, uint64_t seed, uint32_t step_idx, int64_t board_offset,
                     uint32_t flags, uint8_t *actions_out, float *reward, uint8_t *terminated,
                     int64_t n, void *stream) {
    return launch_step(state, actions_out, nullptr, seed, step_idx, board_offset, flags, reward,
                       terminated, n, stream, true, nullptr);
}

static int launch_sample(const void *state, uint64_t seed, uint32_t step_idx, int64_t board_offset,
                         uint32_t flags, uint8_t *actions, int64_t n, void *stream, const uint32_t *step_ctr);

int qttt_env_step(const qttt_env *e, uint8_t *actions, const uint8_t *bits, uint32_t step_idx, int mode,
                  void *stream) {
    if (!e) return QTTT_ERR_NULL;
    switch (mode) {
    case QTTT_ENV_STEP:
        return launch_step(e->state, actions, bits, e->seed, step_idx, e->board_offset, e->flags, e->reward,
                           e->terminated, e->n, stream, false, nullptr, e->step_counter);
    case QTTT_ENV_STEP_OBSERVE: {
        if (e->n > 0 && (!e->classical || !e->q_p1 || !e->q_p1_len || !e->q_p2 || !e->q_p2_len || !e->turn)) return QTTT_ERR_NULL;
        if (((uintptr_t)e->q_p1 & 1u) || ((uintptr_t)e->q_p2 & 7u)) return QTTT_ERR_ACTION;
        const ObsOut o = {e->classical, e->q_p1, e->q_p1_len, e->q_p2, e->q_p2_len, e->turn};
        return launch_step(e->state, actions, bits, e->seed, step_idx, e->board_offset, e->flags, e->reward,
                           e->terminated, e->n, stream, false, &o, e->step_counter);
    }
    case QTTT_ENV_STEP_RANDOM:
        return launch_step(e->state, actions, nullptr, e->seed, step_idx, e->board_offset, e->flags, e->reward,
                           e->terminated, e->n, stream, true, nullptr, e->step_counter);
    case QTTT_ENV_SAMPLE:
        return launch_sample(e->state, e->seed, step_idx, e->board_offset, e->flags, actions, e->n, stream, e->step_counter);
    default:
        return QTTT_ERR_SIZE;
    }
}

int qttt_counter_add(uint32_t *counter, uint32_t by, void *stream) {
    if (!counter) return QTTT_ERR_NULL;
    hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, counter, by);
    return launch_status();
}

int qttt_step_many(void *state, const uint8_t *actions, const uint8_t *bits, uint64_t seed,
                   uint32_t step_idx0, int64_t board_offset, uint32_t flags, float *reward,
                   uint8_t *terminated, int64_t out_stride, int64_t n, int32_t n_steps,
                   void *stream) {
    if (n_steps < 0 || out_stride < 0) return QTTT_ERR_SIZE;
    const u64 first = (u64)(board_offset < 0 ? 0 : board_offset);
    const bool one_hi = n > 0 && (first >> 32) == ((first + (u64)n - 1u) >> 32);
    if ((flags & QTTT_FLAG_FUSED) && n > 0 && n_steps > 0 && one_hi) {
        if (board_offset < 0) return QTTT_ERR_SIZE;
        if (!state || !actions || !reward || !terminated) return QTTT_ERR_NULL;
        if ((uintptr_t)actions & 1u) return QTTT_ERR_ACTION;
        retire_mailbox_for(n, stream);
        Planes p = planes(state, n);
        const bool ar = (flags & QTTT_FLAG_AUTO_RESET) != 0;
        const u32 hi_fold = (u32)(first >> 32) * 0x9E3779B9u;
        dim3 g(grid_for(n)), b(QTTT_BLOCK);
        hipStream_t s = (hipStream_t)stream;
        const uint16_t *a16 = reinterpret_cast<const uint16_t *>(actions);
        u32 *rb = reinterpret_cast<u32 *>(reward);
        // at most FUSED_MAX_PLIES plies per launch (their keys travel as a kernel argument): a longer run is that many
        // launches.  With out_stride == 0 every launch writes its last ply's outputs to the same place; the run's last wins.
#define QTTT_FUSED(HB, AR)                                                                                    \
    hipLaunchKernelGGL((step_fused_kernel<HB, AR>), g, b, 0, s, p.P, p.Q, a16 + (int64_t)done * n,            \
                       bits ? bits + (int64_t)done * n : nullptr, keys, hi_fold, (u32)first,                   \
                       rb + (int64_t)done * out_stride, terminated + (int64_t)done * out_stride, out_stride, n, plies)
        for (int64_t done = 0; done < n_steps; done += FUSED_MAX_PLIES) {
            const int32_t plies = (int32_t)(n_steps - done < FUSED_MAX_PLIES ? n_steps - done : FUSED_MAX_PLIES);
            FusedKeys keys;
            for (int32_t t = 0; t < FUSED_MAX_PLIES; ++t) keys.k[t] = launch_key(seed, step_idx0 + (u32)done + (u32)(t < plies ? t : 0));
            if (bits) { if (ar) QTTT_FUSED(true, true); else QTTT_FUSED(true, false); }
            else      { if (ar) QTTT_FUSED(false, true); else QTTT_FUSED(false, false); }
            const int rc = launch_status();
            if (rc) return rc;
        }
#undef QTTT_FUSED
        return 0;
    }
    for (int32_t t = 0; t < n_steps; ++t) {
        int rc = qttt_step(state, actions + (int64_t)t * 2 * n, bits ? bits + (int64_t)t * n : nullptr,
                           seed, step_idx0 + (uint32_t)t, board_offset, flags,
                           reward + (int64_t)t * out_stride, terminated + (int64_t)t * out_stride,
                           n, stream);
        if (rc != 0) return rc;
    }
    return 0;
}

int qttt_step_random_many(void *state, uint64_t seed, uint32_t step_idx0, int64_t board_offset, uint32_t flags,
                          uint8_t *actions_out, float *reward, uint8_t *terminated, int64_t out_stride,
                          float *returns, int64_t n, int32_t n_steps, void *stream) {
    if (n < 0 || board_offset < 0 || n_steps < 0 || out_stride < 0) return QTTT_ERR_SIZE;
    if (n == 0 || n_steps == 0) return 0;
    if (!state || (reward == nullptr) != (terminated == nullptr)) return QTTT_ERR_NULL;
    if (((uintptr_t)actions_out & 1u) || ((uintptr_t)reward & 3u) || ((uintptr_t)returns & 3u)) return QTTT_ERR_ACTION;
    retire_mailbox_for(n, stream);
    Planes p = planes(state, n);
    hipStream_t s = (hipStream_t)stream;
    uint16_t *a16 = reinterpret_cast<uint16_t *>(actions_out);
    u32 *rb = reinterpret_cast<u32 *>(reward);
    const bool ar = (flags & QTTT_FLAG_AUTO_RESET) != 0;
    // 256-thread workgroups: the finest spread of a small batch over the 256 CUs (4 096 boards = 16 CUs
    // with 512 threads, 16 with 256 — but 262 144 boards = 1 024 workgroups, four per CU, instead of two)
    // at most FUSED_MAX_PLIES plies per launch (their keys travel as a kernel argument); a longer run is that many launches,
    // the boards going through HBM in between (32 bytes per board and 64 plies).  With out_stride == 0 only the LAST ply's
    // outputs are kept, so the earlier launches of such a run write none.
#define QTTT_RFK(AR, RT, KP) hipLaunchKernelGGL((step_random_fused_kernel<256, AR, RT, KP>), dim3(blocks_for(n, 256)), dim3(256), 0, s, \
                                               p.P, p.Q, keys, (u64)board_offset, a_c, r_c, t_c, out_stride, n, plies, returns)
    // the instantiation without the per-ply "what is kept" tests, where it pays: one or two waves per SIMD are bound by a wave's
    // own in-order stream (65 536 boards 0.59 -> 0.56 us per ply, 4 096: 0.58 -> 0.55), from four waves up the test-free loop
    // is no faster and at 1 M boards 2 % slower (profiles/r05/fused_keep_instantiation_ab.txt, same box, alternating)
    const bool keep_all = out_stride != 0 && a16 && rb && n < 262144;
#define QTTT_RF(AR, RT) do { if (keep_all) QTTT_RFK(AR, RT, true); else QTTT_RFK(AR, RT, false); } while (0)
    for (int64_t done = 0; done < n_steps; done += FUSED_MAX_PLIES) {
        const int32_t plies = (int32_t)(n_steps - done < FUSED_MAX_PLIES ? n_steps - done : FUSED_MAX_PLIES);
        const bool last = done + plies == n_steps;
        FusedKeys keys;
        for (int32_t t = 0; t < FUSED_MAX_PLIES; ++t) keys.k[t] = launch_key(seed, step_idx0 + (u32)done + (u32)(t < plies ? t : 0));
        const bool writes = out_stride != 0 || last;
        uint16_t *a_c = (a16 && writes) ? a16 + (int64_t)done * out_stride : nullptr;
        u32 *r_c = (rb && writes) ? rb + (int64_t)done * out_stride : nullptr;
        uint8_t *t_c = (terminated && writes) ? terminated + (int64_t)done * out_stride : nullptr;
        if (returns) { if (ar) QTTT_RF(true, true); else QTTT_RF(false, true); }
        else         { if (ar) QTTT_RF(true, false); else QTTT_RF(false, false); }
        const int rc = launch_status();
        if (rc) return rc;
    }
#undef QTTT_RF
#undef QTTT_RFK
    return 0;
}

int qttt_observe(const void *state, int8_t *classical, uint8_t *q_p1, uint8_t *q_p1_len,
                 uint8_t *q_p2, uint8_t *q_p2_len, uint8_t *turn, int64_t n, void *stream) {
    if (n < 0) return QTTT_ERR_SIZE;
    if (n == 0) return 0;
    if (!state || !classical || !q_p1 || !q_p1_len || !q_p2 || !q_p2_len || !turn) return QTTT_ERR_NULL;
    if (((uintptr_t)q_p1 & 1u) || ((uintptr_t)q_p2 & 7u)) return QTTT_ERR_ACTION;   // 2- / 8-byte LDS row stores
    Planes p = planes(const_cast<void *>(state), n);
    const ObsOut o = {classical, q_p1, q_p1_len, q_p2, q_p2_len, turn};
    // 256-thread workgroups: best or tied at every batch size for this write-heavy kernel (tools/rowbench, us per
    // launch, 256 / 512 / 1024 threads: 65 536 boards 3.7 / 4.1 / 5.2, 1 M: 8.7 / 8.7 / 8.6)
    const int blk_default = tuning_word().load(std::memory_order_relaxed) >> 8;
    const int blk = blk_default ? blk_default : 256;
#define QTTT_OBSERVE(BLK) hipLaunchKernelGGL((observe_kernel<BLK>), dim3((unsigned)((n + 2 * (BLK) - 1) / (2 * (BLK)))), \
                                             dim3(BLK), 0, (hipStream_t)stream, p.P, p.Q, o, n)
    if (blk == 1024) QTTT_OBSERVE(1024);
    else if (blk == 256) QTTT_OBSERVE(256);
    else QTTT_OBSERVE(QTTT_BLOCK);
#undef QTTT_OBSERVE
    return launch_status();
}

int qttt_check_win(const void *state, int8_t *p1_round, int8_t *p2_round, int64_t n, void *stream) {
    if (n < 0) return QTTT_ERR_SIZE;
    if (n == 0) return 0;
    if (!state || !p1_round || !p2_round) return QTTT_ERR_NULL;
    Planes p = planes(const_cast<void *>(state), n);
    hipLaunchKernelGGL(check_win_kernel, dim3(cold_grid_for((n + 1) / 2)), dim3(QTTT_COLD_BLOCK), 0, (hipStream_t)stream,
                       p.P, p.Q, p1_round, p2_round, n);
    return launch_status();
}

int qttt_export(const void *state, uint8_t *moves, uint8_t *n_moves, int8_t *board,
                uint16_t *qmask, uint8_t *n_q, int64_t n, void *stream) {
    if (n < 0) return QTTT_ERR_SIZE;
    if (n == 0) return 0;
    if (!state) return QTTT_ERR_NULL;
    if (!moves && !n_moves && !board && !qmask && !n_q) return 0;            // nothing asked for
    if ((uintptr_t)qmask & 1u) return QTTT_ERR_ACTION;
    Planes p = planes(const_cast<void *>(state), n);
    const ExpOut o = {moves, n_moves, board, qmask, n_q};
    // tools/rowbench (profiles/r03/rowbench_*.txt), us per launch, boards per lane x workgroup size:
    //   1 M boards: 1 x 256 / 512 / 1024 = 13.7 / 14.1 / 12.7, 2 x 256 / 512 / 1024 = 9.2 / 9.4 / 9.4 (one occupancy round)
    //   64 K boards: 1 x 256 = 3.3, 2 x 256 = 3.8 (latency-bound: more waves in flight win)
    if (n >= 384 * 1024)
        hipLaunchKernelGGL((export_kernel<QTTT_COLD_BLOCK, 2>), dim3(cold_grid_for((n + 1) / 2)), dim3(QTTT_COLD_BLOCK), 0,
                           (hipStream_t)stream, p.P, p.Q, o, n);
    else
        hipLaunchKernelGGL((export_kernel<QTTT_COLD_BLOCK, 1>), dim3(cold_grid_for(n)), dim3(QTTT_COLD_BLOCK), 0,
                           (hipStream_t)stream, p.P, p.Q, o, n);
    return launch_status();
}

int qttt_import(void *state, const uint8_t *moves, const uint8_t *n_moves, const int8_t *board,
                const uint16_t *qmask, const uint8_t *n_q, int64_t n, void *stream) {
    if (n < 0) return QTTT_ERR_SIZE;
    if (n == 0) return 0;
    if (!state || !moves || !n_moves || !board || !qmask || !n_q) return QTTT_ERR_NULL;
    Planes p = planes(state, n);
    const ExpOut in = {const_cast<uint8_t *>(moves), const_cast<uint8_t *>(n_moves), const_cast<int8_t *>(board),
                       const_cast<uint16_t *>(qmask), const_cast<uint8_t *>(n_q)};
    hipLaunchKernelGGL((import_kernel<QTTT_COLD_BLOCK>), dim3(cold_grid_for(n)), dim3(QTTT_COLD_BLOCK), 0, (hipStream_t)stream,
                       p.P, p.Q, in, n);
    return launch_status();
}

static int launch_board_op(const void *records_in, void *records_out, int64_t n, void *stream, u32 stamp) {
    if (n < 0) return QTTT_ERR_SIZE;
    if (n == 0) return 0;
    if (!records_in || !records_out) return QTTT_ERR_NULL;
    hipLaunchKernelGGL(board_op_kernel, dim3(cold_grid_for(n)), dim3(QTTT_COLD_BLOCK), 0, (hipStream_t)stream,
                       (const uint8_t *)records_in, (uint8_t *)records_out, n, stamp);
    return launch_status();
}

int qttt_board_op(const void *records_in, void *records_out, int64_t n, void *stream) {
    return launch_board_op(records_in, records_out, n, stream, 0u);
}

int qttt_board_op_sync(const void *records_in, void *records_out, int64_t n, void *stream) {
    const int rc = qttt_board_op(records_in, records_out, n, stream);
    if (rc) return rc;
    const hipError_t e = hipStreamSynchronize((hipStream_t)stream);
    return e == hipSuccess ? 0 : (int)e;
}

// Records in HOST-accessible pinned memory: the host clears the stamp byte of every out record, launches, and polls the
// stamps — the kernel writes a record's stamp after the record itself is visible system-wide.  tools/sync_latency, one
// record: launch + hipStreamSynchronize 14.7 - 16.0 us per call, launch + poll 9.7.  A poll that has not ended after
// ~2 ms (or a batch too large to poll) falls back to synchronising the stream, so the call always returns.
// ---- the bounded mailbox for SINGLE records (board_mailbox_kernel, qttt_aux_kernels.h) ----
// One resident wave on a private non-blocking stream serves a pinned request slot; a call is: copy the record into the
// slot (four 16-byte pieces of 12 data bytes + the request number each; the numbers are written last), poll the
// answer's number.  The wave leaves by itself after QTTT_BOARD_MAILBOX_US microseconds without a request (default 20,
// at most 200; 0 = no mailbox: every call is a launch, as before round 5), QTTT_BOARD_MAILBOX_MAX_US after it started
// whatever the traffic (default 1000, at most 10000), or when qttt_board_mailbox_retire() asks it to, and says so in
// `exited`; the next call then launches it again.  A request that meets a wave which has just left is answered by the
// relaunch (the host watches `exited` while it polls), and a call that gets no answer within 20 ms turns the mailbox
// off for the rest of the process and goes through the launch path — the call always returns.
// What a resident wave costs others: a DEVICE-wide synchronise (hipDeviceSynchronize, torch.cuda.synchronize()) issued
// within the idle window after a Board call waits for the wave to leave (<= the window; <= the residency bound when
// another thread keeps calling); stream-level synchronisation and the legacy default stream do not (the stream is
// non-blocking).  A step launch that fills the chip (>= 512 K boards) retires it first (launch_step), so that the wave's
// CU slot does not cost that launch a second partial round.  The `stream` argument of the call is not used on this path:
// host records have no device-side producer to be ordered after.
}  // extern "C"
namespace {
struct BoardMailbox {
    std::mutex mu;
    bool tried = false, on = false, alive = false, leaving = false;
    int device = -1;
    uint8_t *slot_in = nullptr, *slot_out = nullptr;     // 64 bytes each, pinned, system-coherent
    u32 *exited = nullptr;
    hipStream_t stream = nullptr;
    u32 ring = 0, generation = 0;
    u64 idle_ticks = 0, resident_ticks = 0;

    bool start() {                                       // once per process
        tried = true;
        long us = 20, max_us = 1000;
        if (const char *e = getenv("QTTT_BOARD_MAILBOX_US")) us = atol(e);
        if (const char *e = getenv("QTTT_BOARD_MAILBOX_MAX_US")) max_us = atol(e);
        if (us <= 0) return false;
        if (us > 200) us = 200;
        if (max_us < us) max_us = us;
        if (max_us > 10000) max_us = 10000;
        idle_ticks = (u64)us * 100u;                     // s_memrealtime: 100 MHz
        resident_ticks = (u64)max_us * 100u;
        uint8_t *mem = nullptr;
        if (hipGetDevice(&device) != hipSuccess) return false;
        if (hipHostMalloc(reinterpret_cast<void **>(&mem), 256, hipHostMallocCoherent) != hipSuccess) { (void)hipGetLastError(); return false; }
        memset(mem, 0, 256);
        slot_in = mem; slot_out = mem + 64; exited = reinterpret_cast<u32 *>(mem + 128);
        if (hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); (void)hipHostFree(mem); return false; }
        on = true;
        return true;
    }
    bool launch() {
        ++generation;
        hipLaunchKernelGGL(board_mailbox_kernel, dim3(1), dim3(64), 0, stream, reinterpret_cast<const mbox_u32x4 *>(slot_in),
                           reinterpret_cast<mbox_u32x4 *>(slot_out), exited, generation, ring, 1u << 20, idle_ticks, 1u << 20,
                           resident_ticks);
        alive = hipGetLastError() == hipSuccess;
        leaving = false;
        g_mailbox_resident.store(alive, std::memory_order_relaxed);
        return alive;
    }
    void write_numbers(u32 v) {
        volatile u32 *w = reinterpret_cast<volatile u32 *>(slot_in);
        w[3] = v; w[7] = v; w[11] = v; w[15] = v;
    }
    bool has_left() { return *static_cast<volatile u32 *>(exited) == generation; }
    // the wave was asked to leave: wait until it has said so (its next poll: a few us; bounded), then give the runtime ONE
    // chance to retire the finished kernel here rather than beside the caller's next launches (see g_mailbox_resident)
    void await_exit() {
        const auto give_up = std::chrono::steady_clock::now() + std::chrono::milliseconds(2);
        bool gone = has_left();
        for (unsigned spin = 1; !gone; ++spin, gone = has_left())
            if ((spin & 1023u) == 0u && std::chrono::steady_clock::now() > give_up) break;   // (it then leaves by its idle exit)
        alive = leaving = false;
        g_mailbox_resident.store(false, std::memory_order_relaxed);
        if (gone) { (void)hipStreamQuery(stream); (void)hipGetLastError(); }
    }
    // 0 = a resident wave was asked to leave (or none was resident); it is gone within a poll (a few us)
    int retire(bool wait) {
        std::lock_guard<std::mutex> g(mu);
        if (!on || !alive) { g_mailbox_resident.store(false, std::memory_order_relaxed); return 0; }
        if (has_left()) { alive = leaving = false; g_mailbox_resident.store(false, std::memory_order_relaxed); return 0; }
        if (!leaving) {
            write_numbers(MBOX_LEAVE);
            leaving = true;
        }
        if (wait) await_exit();
        return 0;
    }
    // From the step entries, in front of a launch that fills the chip (never blocks, never calls into the runtime): ask a
    // resident wave to leave.
    void housekeeping() {
        std::unique_lock<std::mutex> g(mu, std::try_to_lock);
        if (!g.owns_lock()) return;                              // a Board call is in flight on another thread: its business
        if (!on || !alive) { g_mailbox_resident.store(false, std::memory_order_relaxed); return; }
        if (has_left()) { alive = leaving = false; g_mailbox_resident.store(false, std::memory_order_relaxed); return; }
        if (!leaving) {
            static const bool keep = [] { const char *e = getenv("QTTT_BOARD_MAILBOX_KEEP"); return e && atoi(e) != 0; }();
            if (keep) return;                                    // (QTTT_BOARD_MAILBOX_KEEP=1: A/B diagnostics of this very rule)
            write_numbers(MBOX_LEAVE);
            leaving = true;
        }
        g_mailbox_resident.store(false, std::memory_order_relaxed);   // asked once: the later launches have nothing to do here
    }
    // 0 = answered (out filled), 1 = not served: use the launch path
    int call(const void *rec_in, void *rec_out) {
        std::lock_guard<std::mutex> g(mu);
        if (!tried) start();
        int dev = -1;
        if (!on || hipGetDevice(&dev) != hipSuccess || dev != device) return 1;
        if (leaving) await_exit();
        ring = mbox_next(ring);
        volatile u32 *answer = reinterpret_cast<volatile u32 *>(slot_out) + 15;
        volatile u32 *gone = exited;
        if (alive && *gone == generation) alive = false;
        const uint8_t *src = static_cast<const uint8_t *>(rec_in);
        for (int k = 0; k < 4; ++k) memcpy(slot_in + 16 * k, src + 12 * k, 12);   // record bytes 0..47 (41 are read)
        std::atomic_thread_fence(std::memory_order_release);
        write_numbers(ring);
        if (!alive && !launch()) { on = false; return 1; }
        const auto give_up = std::chrono::steady_clock::now() + std::chrono::milliseconds(20);
        for (unsigned spin = 1; *answer != ring; ++spin) {
            if (*gone == generation && *answer != ring) {          // the wave left before it saw this request
                if (!launch()) { on = false; return 1; }
            }
            if ((spin & 4095u) == 0u && std::chrono::steady_clock::now() > give_up) {
                on = false;                                        // something is wrong with this path on this host: stop using it
                return 1;
            }
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        memcpy(rec_out, slot_out, 60);
        static_cast<uint8_t *>(rec_out)[60] = static_cast<uint8_t *>(rec_out)[61] = static_cast<uint8_t *>(rec_out)[62] = 0;
        static_cast<uint8_t *>(rec_out)[63] = 1;                   // the completion stamp of the contract
        return 0;
    }
};
BoardMailbox &board_mailbox() {
    static BoardMailbox m;
    return m;
}
void mailbox_housekeeping(hipStream_t) { board_mailbox().housekeeping(); }
}  // namespace
extern "C" {

int qttt_board_mailbox_retire(int wait) {
    if (!g_mailbox_resident.load(std::memory_order_relaxed) && !wait) return 0;
    return board_mailbox().retire(wait != 0);
}

int qttt_board_op_host(const void *records_in, void *records_out, int64_t n, void *stream) {
    if (n < 0) return QTTT_ERR_SIZE;
    if (n == 0) return 0;
    if (!records_in || !records_out) return QTTT_ERR_NULL;
    if (n == 1 && board_mailbox().call(records_in, records_out) == 0) return 0;
    constexpr int64_t POLL_MAX_RECORDS = 256;
    volatile uint8_t *out = static_cast<volatile uint8_t *>(records_out);
    const bool poll = n <= POLL_MAX_RECORDS;
    if (poll)
        for (int64_t i = 0; i < n; ++i) out[i * QTTT_BOARD_RECORD_BYTES + QTTT_BOARD_RECORD_BYTES - 1] = 0;
    std::atomic_thread_fence(std::memory_order_release);
    const int rc = launch_board_op(records_in, records_out, n, stream, poll ? 1u : 0u);
    if (rc) return rc;
    if (poll) {
        const auto give_up = std::chrono::steady_clock::now() + std::chrono::milliseconds(2);
        bool done = false;
        for (unsigned spin = 1; !done; ++spin) {
            done = true;
            for (int64_t i = n - 1; i >= 0 && done; --i) done = out[i * QTTT_BOARD_RECORD_BYTES + QTTT_BOARD_RECORD_BYTES - 1] != 0;
            if (!done && (spin & 1023u) == 0u && std::chrono::steady_clock::now() > give_up) break;
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        if (done) return 0;
    }
    const hipError_t e = hipStreamSynchronize((hipStream_t)stream);
    return e == hipSuccess ? 0 : (int)e;
}

static int launch_sample(const void *state, uint64_t seed, uint32_t step_idx, int64_t board_offset,
                         uint32_t flags, uint8_t *actions, int64_t n, void *stream, const uint32_t *step_ctr) {
    if (n < 0 || board_offset < 0) return QTTT_ERR_SIZE;
    if (n == 0) return 0;
    if (!state || !actions) return QTTT_ERR_NULL;
    if ((uintptr_t)actions & 1u) return QTTT_ERR_ACTION;   // written as u16 pairs
    Planes p = planes(const_cast<void *>(state), n);
    const u64 key = step_ctr ? ((u64)step_idx << 32) : launch_key(seed, step_idx);
    hipLaunchKernelGGL(sample_actions_kernel, dim3(grid_for((n + 1) / 2)), dim3(QTTT_BLOCK), 0,
                       (hipStream_t)stream, p.P, (u32)key, (u32)(key >> 32), (u64)board_offset,
                       (u32)((flags & QTTT_FLAG_AUTO_RESET) != 0), reinterpret_cast<uint16_t *>(actions), n,
                       step_ctr, (u64)seed);
    return launch_status();
}

int qttt_sample_actions(const void *state, uint64_t seed, uint32_t step_idx, int64_t board_offset,
                        uint32_t flags, uint8_t *actions, int64_t n, void *stream) {
    return launch_sample(state, seed, step_idx, board_offset, flags, actions, n, stream, nullptr);
}

int qttt_node_info(const void *state, int8_t *winner, uint8_t *terminal, uint64_t *legal,
                   int64_t *key, uint64_t *state_key, int64_t n, void *stream) {
    if (n < 0) return QTTT_ERR_SIZE;
    if (n == 0) return 0;
    if (!state) return QTTT_ERR_NULL;
    if (!winner && !terminal && !legal && !key && !state_key) return 0;      // nothing asked for
    Planes p = planes(const_cast<void *>(state), n);
    // workgroup size by batch (tools/rowbench, us per launch, 256 / 512 / 1024 threads: 1 M boards 11.7 / 11.2 / 10.5 —
    // the 12 KB of tables are filled once per workgroup; 64 K boards 4.6 / 4.7 / 5.8 — latency-bound).  Measured and not
    // adopted: a 1 000-entry table of the accumulator after the first three board elements (three multiply steps
    // less per board): 10.2 us with 1024 threads, but every smaller shape and expand lose as much to the 8 KB fill.
#define QTTT_NI(BLK, PK) hipLaunchKernelGGL((node_info_kernel<BLK, PK>), dim3(blocks_for((n + 1) / 2, BLK)), dim3(BLK), 0, \
                                            (hipStream_t)stream, p.P, p.Q, winner, terminal, (u64 *)legal, key, (u64 *)state_key, n)
    if (n >= 384 * 1024) { if (key) QTTT_NI(1024, true); else QTTT_NI(1024, false); }
    else                 { if (key) QTTT_NI(256, true);  else QTTT_NI(256, false); }
#undef QTTT_NI
    return launch_status();
}

uint64_t qttt_state_key(uint64_t plane_p_word, uint64_t plane_q_word) { return state_key(plane_p_word, (u32)plane_q_word); }

// the per-child rows [n,2] are written as one vector per pair
static int expand_rows_misaligned(const int8_t *winner, const uint8_t *terminal, const uint64_t *legal, const int64_t *key,
                                  const uint64_t *state_key) {
    return ((uintptr_t)winner & 1u) || ((uintptr_t)terminal & 1u) || ((uintptr_t)legal & 15u) || ((uintptr_t)key & 15u) ||
           ((uintptr_t)state_key & 15u);
}

int qttt_expand(const void *state, const uint8_t *action36, void *child0, void *child1,
                uint8_t *n_children, int8_t *winner, uint8_t *terminal, uint64_t *legal,
                int64_t *key, uint64_t *state_key, int64_t n, void *stream) {
    if (n < 0) return QTTT_ERR_SIZE;
    if (n == 0) return 0;
    if (!state || !action36 || !child0 || !child1) return QTTT_ERR_NULL;
    if (expand_rows_misaligned(winner, terminal, legal, key, state_key)) return QTTT_ERR_ACTION;
    Planes p = planes(const_cast<void *>(state), n), c0 = planes(child0, n), c1 = planes(child1, n);
    const ExpandOut o = {n_children, winner, terminal, (u64 *)legal, key, (u64 *)state_key};
    // workgroup size by batch (tools/rowbench, us per launch, 256 / 512 / 1024 threads: 1 M pairs with native keys 18.3 /
    // 18.2 / 17.5, with the CPython keys 28.3 / 26.8 / 25.0; 64 K pairs 4.5 / 4.4 / 4.7 and 5.8 / 6.1 / 7.7)
#define QTTT_EX(BLK, PK) hipLaunchKernelGGL((expand_kernel<BLK, PK>), dim3(blocks_for(n, BLK)), dim3(BLK), 0, (hipStream_t)stream, \
                                            p.P, p.Q, action36, c0.P, c0.Q, c1.P, c1.Q, o, n)
    if (n >= 384 * 1024) { if (key) QTTT_EX(1024, true); else QTTT_EX(1024, false); }
    else                 { if (key) QTTT_EX(256, true);  else QTTT_EX(256, false); }
#undef QTTT_EX
    return launch_status();
}

int qttt_expand_rollout(const void *state, const uint8_t *action36, void *child0, void *child1,
                        uint8_t *n_children, int8_t *winner, uint8_t *terminal, uint64_t *legal,
                        int64_t *key, uint64_t *state_key, uint64_t seed, uint32_t step_idx0,
                        int64_t board_offset, int32_t n_sims, int32_t *value_sum, int8_t *result,
                        int64_t n, void *stream) {
    constexpr int BLK = 256;
    if (n < 0 || board_offset < 0 || n_sims < 1 || n_sims > QTTT_EXPAND_ROLLOUT_MAX_SIMS) return QTTT_ERR_SIZE;
    if (n == 0) return 0;
    if (!state || !action36 || !value_sum) return QTTT_ERR_NULL;
    if (expand_rows_misaligned(winner, terminal, legal, key, state_key) || ((uintptr_t)value_sum & 3u)) return QTTT_ERR_ACTION;
    Planes p = planes(const_cast<void *>(state), n);
    Planes c0 = {nullptr, nullptr}, c1 = {nullptr, nullptr};
    if (child0) c0 = planes(child0, n);
    if (child1) c1 = planes(child1, n);
    const ExpandOut o = {n_children, winner, terminal, (u64 *)legal, key, (u64 *)state_key};
    // Two mappings with the same results (tools/rowbench): a lane per (pair, simulation, child) for the latency-bound
    // case — few playouts in all — and the job-list kernel, whose workgroups expand P pairs once and deal the playouts of
    // the children that exist to their lanes, wherever the playouts are the work.  P: as many pairs as lanes, but at
    // least ~1 000 workgroups so that a small batch still covers the chip.
    const int64_t playouts = n * (int64_t)n_sims;
    if (playouts >= 262144) {
        // P pairs per workgroup: a power of two (the workgroups' rows of every output then start on whole cache lines),
        // at most one pair per lane, and few enough that ~1 000 workgroups exist.  tools/rowbench, us per launch, 10
        // playouts per child: 65 536 pairs P = 32 / 48 / 58 / 64 / 128 / 256 -> 25.9 / 26.2 / 26.7 / 24.9 / 26.9 / 37.7;
        // 1 M pairs 64 / 128 / 251 / 256 -> 194 / 180 / 180 / 175 (one playout per child: 128 / 193 / 256 -> 46.7 / 39.9 /
        // 37.3).  Filling the workgroup's last round of lanes (P = 58: 708 jobs = 2.8 rounds instead of 3.05) does not
        // pay: the chip is bound by the total of wave-rounds, not by a workgroup's own span.
        int64_t P = 8;
        while (P * 2 <= XR_MAX_PAIRS && P * 2 * 1024 <= n) P *= 2;
        const u32 ppb = (u32)P;
        const unsigned grid = (unsigned)((n + ppb - 1) / ppb);
#define QTTT_XJ(PK) hipLaunchKernelGGL((expand_rollout_jobs_kernel<BLK, PK>), dim3(grid), dim3(BLK), 0, (hipStream_t)stream,  \
                                       p.P, p.Q, action36, c0.P, c0.Q, c1.P, c1.Q, o, (u64)seed, step_idx0,                 \
                                       (u64)board_offset, (u32)n_sims, ppb, value_sum, result, n)
        if (key) QTTT_XJ(true); else QTTT_XJ(false);
#undef QTTT_XJ
        return launch_status();
    }
    const u32 ppb = (u32)(BLK / (2 * n_sims));                    // whole pairs per workgroup
    const unsigned grid = (unsigned)((n + ppb - 1) / ppb);
#define QTTT_XR(PK) hipLaunchKernelGGL((expand_rollout_kernel<BLK, PK>), dim3(grid), dim3(BLK), 0, (hipStream_t)stream,     \
                                       p.P, p.Q, action36, c0.P, c0.Q, c1.P, c1.Q, o, (u64)seed, step_idx0,               \
                                       (u64)board_offset, (u32)n_sims, ppb, value_sum, result, n)
    if (key) QTTT_XR(true); else QTTT_XR(false);
#undef QTTT_XR
    return launch_status();
}

int qttt_rollout(const void *state, uint64_t seed, uint32_t step_idx0, int64_t board_offset,
                 int8_t *result, uint8_t *plies, void *final_state, int64_t n, void *stream) {
    if (n < 0 || board_offset < 0) return QTTT_ERR_SIZE;
    if (n == 0) return 0;
    if (!state || !result || !plies) return QTTT_ERR_NULL;
    Planes p = planes(const_cast<void *>(state), n);
    Planes f = {nullptr, nullptr};
    if (final_state) f = planes(final_state, n);
    hipLaunchKernelGGL(rollout_kernel, dim3(grid_for(n)), dim3(QTTT_BLOCK), 0, (hipStream_t)stream,
                       p.P, p.Q, (u64)seed, step_idx0, (u64)board_offset, result, plies, f.P, f.Q, n);
    return launch_status();
}

int qttt_rollout_many(const void *state, uint64_t seed, uint32_t step_idx0, int64_t board_offset,
                      int32_t n_sims, int8_t *result, uint8_t *plies, int64_t n, void *stream) {
    if (n < 0 || board_offset < 0 || n_sims < 0) return QTTT_ERR_SIZE;
    if (n == 0 || n_sims == 0) return 0;
    if (!state || !result) return QTTT_ERR_NULL;
    Planes p = planes(const_cast<void *>(state), n);
    const int64_t lanes = n * (int64_t)n_sims;
    hipLaunchKernelGGL(rollout_many_kernel, dim3(grid_for(lanes)), dim3(QTTT_BLOCK), 0, (hipStream_t)stream,
                       p.P, p.Q, (u64)seed, step_idx0, (u64)board_offset, (u32)n_sims, result, plies, lanes);
    return launch_status();
}

int qttt_encode(const void *state, float *vec, uint8_t *mask, int64_t n, void *stream) {
    if (n < 0) return QTTT_ERR_SIZE;
    if (n == 0) return 0;
    if (!state || !vec) return QTTT_ERR_NULL;
    Planes p = planes(const_cast<void *>(state), n);
    if (((uintptr_t)vec & 15u) || ((uintptr_t)mask & 3u)) return QTTT_ERR_ACTION;   // vector stores
    hipLaunchKernelGGL(encode_kernel, dim3((unsigned)((n + QTTT_ENC_BOARDS - 1) / QTTT_ENC_BOARDS)),
                       dim3(QTTT_ENC_BLOCK), 0, (hipStream_t)stream, p.P, p.Q, vec, mask, n);
    return launch_status();
}

}  // extern "C"
