// qttt_step_core.h — one Env.step (env.py:34-53) on a board held in four registers: the hot path.
#ifndef QTTT_STEP_CORE_H
#define QTTT_STEP_CORE_H
#include "qttt_state.h"

namespace {

// ====================================================================== the hot path
// One Env.step (env.py:34-53) on the board held in (P0,P1,Q0,Q1), in four pieces so that qttt_expand can share
// everything that does not depend on the collapse bit between its two children (all forceinline: step_core below
// compiles to the code it was as one function):
//   step_prep     validity (board.py:10-18) + the components of lo / hi, cycle test (board.py:28-42)
//   step_reroot   the path reversal: re-root x's tree at x, x receives the move as its parent edge
//   step_fields   append, n += 1, qstructs insert / union / pop in list order, classical |= component (board.py:19,42-69)
//   step_line     "does any line exist" + done bit (board.py:71-115 reduced to what env.py:49,51 need)
//@isa other
struct StepPrep {
    u32 lo, hi, x16, pm, n6, mlo, mhi;
    u64 comps;
    bool legal, has_lo, cyc;
};

// three-input bitwise function as ONE v_bitop3_b32 (the "fast" VALU class of this part, tools/valu_rates.cpp);
// F is the function written on the three truth-table columns: BITOP3(a, b, c, (A & B) | (C & ~B))
#define BITOP3(a, b, c, F) __builtin_amdgcn_bitop3_b32((a), (b), (c), (u32)([] { constexpr u32 A = 0xF0u, B = 0xCCu, C = 0xAAu; \
                                                                                 (void)A; (void)B; (void)C; return (F) & 0xFFu; }()))
// (a << SH) | c as ONE v_lshl_or_b32 (left to itself the compiler rewrites "mask, shift, or" chains into longer ones)
template <u32 SH>
__device__ __forceinline__ u32 lshl_or(u32 a, u32 c) {
    u32 r;
    asm("v_lshl_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "n"(SH), "v"(c));
    return r;
}

//@isa decode
template <bool TRUSTED>
__device__ __forceinline__ StepPrep step_prep(u32 P1, u32 Q1, u32 act) {
    StepPrep s;
    const u32 a = act & 0xFFu, b = act >> 8;            // action[0], action[1] (env.py:37-38)
    s.lo = TRUSTED ? a : min(a, b);                     // board.py:16-18
    s.hi = TRUSTED ? b : max(a, b);
    s.x16 = (a ^ b) << 16;
    // the two squares as a mask at the classical mask's place in P1 (only looked at when hi < 9)
    const u32 pmS = ((1u << P1_CL_SHIFT) << (s.lo & 31u)) | ((1u << P1_CL_SHIFT) << (s.hi & 31u));
    // board.py:10-15 (+ IndexError for >8, swallowed at env.py:41): reject before mutating
    s.legal = TRUSTED || (s.hi < 9u && s.lo != s.hi && (P1 & pmS) == 0u);
    s.pm = pmS >> P1_CL_SHIFT;
//@isa comps
    s.n6 = P1 >> (P1_N_SHIFT - 2u);                      // bits 2..5 = 4 * moves played (bits 6,7 of P1 are 0)
    s.comps = (u64)Q1 | ((u64)((P1 >> P1_CHI_SHIFT) & 0xFu) << 32);
    // a garbage square (>= 9, only when !legal: nothing below is looked at then) must not make the shift itself
    // undefined: the count is reduced as v_lshrrev_b64 reduces it, which costs no instruction
    s.mlo = (u32)(s.comps >> (s.lo & 63u)) & SLOT_LSB;   // slot holding lo (board.py:28-33)
    s.mhi = (u32)(s.comps >> (s.hi & 63u)) & SLOT_LSB;   // slot holding hi (board.py:35-40)
    asm("" : "+v"(s.mlo), "+v"(s.mhi));                  // (masked once: keeps the cycle test a plain two-operand AND)
    s.has_lo = s.mlo != 0u;
    s.cyc = (s.mlo & s.mhi) != 0u;                       // board.py:42: same component -> cycle
    return s;
}

//@isa childend
// x: the square that becomes the child end of the new edge (its tree is re-rooted at it).  On a cycle it is the
// square the closing move lands on (qeval.py:35: bit 0 -> lo, 1 -> hi); otherwise either end will do: lo when lo
// is in no component (an isolated square: no walk at all), else hi (isolated: no walk; in a component: a walk up
// hi's tree).  ONE select on two scalar conditions.  qttt_import replays this rule (import_board).
// (Scalars by value, not the StepPrep: a select between two members of a struct whose address is taken is turned into
// an indexed load from a stack copy of the struct, which the AMDGPU back end then "promotes" to 48 bytes of LDS per
// thread — measured in expand_kernel: 17.8 -> 30.3 us per 1 M pairs.)
__device__ __forceinline__ u32 step_child_end4(u32 lo, u32 hi, bool has_lo, bool cyc, u32 bit) {
    return ((!has_lo || (cyc && bit == 0u)) ? lo : hi) * 4u;
}

//@isa walk
// re-root x's tree at x: reverse the parent edges along the path x -> old root.  All quantities are "times four":
// v4 = 4v is the shift that brings square v's nibble to bits 2..5, ec4 = 4 * code of the edge found there, and
// rotating Q0 right by ec4 brings 4 * (lo^hi) of that edge to bits 2..5: the other end of the edge is one rotate
// and one xor-and away.  x itself receives this move as its parent edge (code of round n), every later node on
// the path receives the edge its child used to have.  n6: bits 2..5 = 4 * round of this move (StepPrep::n6).
__device__ __forceinline__ u64 step_reroot(u64 P, u32 Q0, u32 x4, u32 n6) {
    u32 v4 = x4, prev4 = ~n6;                    // code of round n = 15 - n: only bits 2..5 are looked at
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const u32 t = (u32)(P >> v4);
        const u32 ec4 = t & 0x3Cu;
        P ^= (u64)((t ^ prev4) & 0x3Cu) << v4;   // sq[v] = prev
        if (ec4 == 0u) break;                    // v was the root
        v4 ^= rotr32(Q0, ec4) & 0x3Cu;
        prev4 = ec4;
    }
    return P;
}

//@isa append
// everything of the move that does not depend on the collapse bit; P1 holds the fields (its nibble bits pass through)
// n6 is StepPrep::n6 of the state BEFORE the move.
__device__ __forceinline__ void step_fields(const StepPrep &s, u32 &P1, u32 &Q0, u32 &Q1) {
    // board.py:19: append.  Only x = lo^hi is kept (see the header): round n <= 7 goes to its
    // nibble of Q0 (x*4 rotated right by 4n+4, i.e. x<<16 rotated by 4n+18; a rotate only looks at the low five
    // bits of the count), every move to the `last x` field; n += 1.
    Q0 ^= rotr32(s.x16, s.n6 + 18u);
//@isa qstructs
    // ---- board.py:42-69 on the cached qstructs, all cases in one straight line ----
    // ffbl_raw(0) = -1, a 64-bit shift by -1 (= 63) gives 0: c1 = component of hi, 0 if none
    const u32 c1 = (u32)(s.comps >> (ffbl_raw(s.mhi) & 63u)) & 0x1FFu;
    // the slot the move goes to (board.py:58-69): lo's, else hi's, else the first empty one.  Slots are compact,
    // so every empty slot's bit lies above every occupied slot's and ONE "lowest set bit" picks the right one out
    // of: the slot of lo / hi (tsel) | the LSBs of ALL empty slots.  y = non-empty flags of slots 0..2 at bits
    // 8,17,26 (carry trick); shifted right by 8 they sit on the slots' LSBs, and bit 27 of the shifted word is 0
    // (slot 3 is the fallback: it is empty whenever the others are all taken and a new component appears).
    const u32 tsel = s.has_lo ? s.mlo : s.mhi;
    const u32 c32 = (u32)s.comps;
    const u32 y8 = (((c32 & 0x03FDFEFFu) + 0x03FDFEFFu) | c32) >> 8;
    const u32 sT = ffbl_raw(BITOP3(tsel, y8, SLOT_LSB, A | (~B & C)));
    // the move's squares join the slot; so does hi's component (a no-op unless this is a
    // union, board.py:58-61: on a cycle or when only hi is in a slot it is that slot already)
    const u64 ins = (u64)(s.pm | c1) << sT;
    const u32 q1 = c32 | (u32)ins;
    const u32 chi2 = (u32)(s.comps >> 32) | (u32)(ins >> 32);
    // pop hi's slot on a cycle (board.py:56) or a union (board.py:61) <=> both are in a slot
    const u32 mpop = s.has_lo ? s.mhi : 0u;
    const u32 low = mpop - 1u;                                      // all ones = keep everything
    Q1 = BITOP3(q1, low, __builtin_amdgcn_alignbit(chi2, q1, 9u), (A & B) | (C & ~B));
//@isa fields
    // P1: chi and `last x` replaced (after a pop at most three slots are left: bits 27..35 are empty); on a cycle
    // (board.py:44-56 + qeval.py:5-51) every square of the component goes classical — it already holds its parent
    // edge's round, x the closing move's; n += 1
    u32 p = P1 & ~((0xFu << P1_CHI_SHIFT) | (0xFu << P1_LX_SHIFT));
    p = lshl_or<P1_CHI_SHIFT>(mpop ? 0u : chi2, p);
    p = lshl_or<P1_CL_SHIFT>(s.cyc ? c1 : 0u, p);
    P1 = (p | s.x16) + (1u << P1_N_SHIFT);
}

//@isa line
// board.py:71-115 reduced to "does any line exist" (all that env.py:49,51 need): parity of the
// round on each classical square -> X / O masks -> table lookup.  Codes are complemented, so
// a set low bit means an EVEN round (X).  All masks here are "times four" (bit v+2 = square v).
// Eight classical squares = the autofill of board.py:22-25 is due: the ninth square counts as
// X (round 8) — with 8 or 9 classical squares X is simply "everything that is not O".
// Returns 0x7F iff a completed line exists (else 0).  P1's done bit is only ever SET: a line stays a line and
// classical squares stay classical, so "done" (env.py:51) is monotone along any sequence of moves; a state whose
// done bit is set without cause (only VecEnv.from_state with a hand-made tensor can bring one) keeps it.
__device__ __forceinline__ u32 step_line(u32 P0, u32 &P1, const uint8_t *lut) {
    const u32 par4 = P0 & 0x44444444u;
    const u32 even4 = lshl_or<8>(P1 & 4u, __builtin_amdgcn_udot8(par4, 0x00008421u, 0u, false) |
                                          (__builtin_amdgcn_udot8(par4, 0x84210000u, 0u, false) << 4));
    u32 cl4 = (P1 >> (P1_CL_SHIFT - 2u)) & 0x7FCu;                      // bits 20,21 of P1 are 0
    asm("" : "+v"(cl4));    // masked ONCE: otherwise the mask is folded into three v_bitop3 with an SGPR operand (slow class)
    const u32 pc = (u32)__builtin_popcount(cl4);
    u32 O4 = cl4 & ~even4;
    asm("" : "+v"(O4));                                             // (keeps the xor below a literal-operand VOP2)
    const u32 X4 = pc >= 8u ? (O4 ^ 0x7FCu) : (cl4 & even4);
    // a dword per mask: the byte offset is the mask "times four"
    const u32 win = *reinterpret_cast<const u32 *>(lut + X4) | *reinterpret_cast<const u32 *>(lut + O4);
    // env.py:51: a line, or len(moves) > 8  <=>  at least 8 classical squares.  win is 0 or 0x7F,
    // pc <= 9: bit 3 of (win | pc) is the answer
    P1 = lshl_or<28>(BITOP3(win, pc, 8u, (A | B) & C), P1);
    return win;
}

// The same for callers that also want to know WHO holds a line (GameState.update_winner, mcts.py:52-65): returns
// bit 0 = X (player 1) has a line, bit 1 = O (player 2) has one; P1's done bit is updated as by step_line.
__device__ __forceinline__ u32 step_line_xo(u32 P0, u32 &P1, const uint8_t *lut) {
    const u32 par4 = P0 & 0x44444444u;
    const u32 even4 = lshl_or<8>(P1 & 4u, __builtin_amdgcn_udot8(par4, 0x00008421u, 0u, false) |
                                          (__builtin_amdgcn_udot8(par4, 0x84210000u, 0u, false) << 4));
    u32 cl4 = (P1 >> (P1_CL_SHIFT - 2u)) & 0x7FCu;
    asm("" : "+v"(cl4));
    const u32 pc = (u32)__builtin_popcount(cl4);
    u32 O4 = cl4 & ~even4;
    asm("" : "+v"(O4));                                             // (keeps the xor below a literal-operand VOP2)
    const u32 X4 = pc >= 8u ? (O4 ^ 0x7FCu) : (cl4 & even4);
    const u32 wx = *reinterpret_cast<const u32 *>(lut + X4), wo = *reinterpret_cast<const u32 *>(lut + O4);
    P1 = lshl_or<28>(BITOP3(wx | wo, pc, 8u, (A | B) & C), P1);      // only ever set (see step_line)
    return (wx & 1u) | ((wo & 1u) << 1);
}

//@isa reset
// Auto-reset: a finished board (done bit) restarts empty = all zero.  The mask is made with a shift the compiler
// cannot see through: left to itself it turns the four ANDs into a compare and four v_cndmask (five slow-class
// instructions instead of five fast ones).
__device__ __forceinline__ void step_auto_reset(u32 &P0, u32 &P1, u32 &Q0, u32 &Q1) {
    u32 gone = (u32)((int)P1 >> 31);                    // all ones iff done
    asm("" : "+v"(gone));
    P0 &= ~gone;
    P1 &= ~gone;
    Q0 &= ~gone;
    Q1 &= ~gone;
}

// `lut` is the workgroup's LDS line table (fill_line_lut: one entry per dword, 0x7F = the mask holds a line).
// Returns 0x7F iff a completed line exists afterwards (else 0); P1's done bit is updated.
// TRUSTED: the caller guarantees a legal action with action[0] < action[1] (the in-kernel policy of the
// rollout does): no sorting, no validation.
template <bool AUTO_RESET, bool TRUSTED = false>
__device__ __forceinline__ u32 step_core(u32 &P0, u32 &P1, u32 &Q0, u32 &Q1, u32 act, u32 bit,
                                         const uint8_t *lut) {
    if (AUTO_RESET) {                                   // finished boards restart: empty = all zero
        step_auto_reset(P0, P1, Q0, Q1);
    }
//@isa glue
    const StepPrep s = step_prep<TRUSTED>(P1, Q1, act);
    if (s.legal) {
        const u64 P = step_reroot((u64)P0 | ((u64)P1 << 32), Q0, step_child_end4(s.lo, s.hi, s.has_lo, s.cyc, bit), s.n6);
        P0 = (u32)P;
        P1 = (u32)(P >> 32);
        step_fields(s, P1, Q0, Q1);
    }
    return step_line(P0, P1, lut);
}

//@isa other
// Both values of the collapse bit at once (MCTS._step, mcts.py:233-267): child a = the closing move on lo (bit 0),
// child b = on hi (bit 1).  Everything but the path reversal and the line test is shared; without a cycle the
// children are the same board.  Returns n_children: 0 = make_move raises, 1 = no collapse, 2 = collapse.
// xo_a / xo_b: who holds a line in each child (step_line_xo).
__device__ __forceinline__ u32 step_core_both(u32 P0, u32 P1, u32 &Q0, u32 &Q1, u32 act, const uint8_t *lut,
                                              u32 &P0a, u32 &P1a, u32 &P0b, u32 &P1b, u32 &xo_a, u32 &xo_b) {
    const StepPrep s = step_prep<false>(P1, Q1, act);
    P0a = P0b = P0;
    P1a = P1b = P1;
    if (s.legal) {
        const u64 P = (u64)P0 | ((u64)P1 << 32);
        const u64 Pa = step_reroot(P, Q0, step_child_end4(s.lo, s.hi, s.has_lo, s.cyc, 0u), s.n6);
        const u64 Pb = s.cyc ? step_reroot(P, Q0, step_child_end4(s.lo, s.hi, s.has_lo, s.cyc, 1u), s.n6) : Pa;
        u32 F = P1;                                      // the fields are the same for both children
        step_fields(s, F, Q0, Q1);
        F &= ~0x3Fu;
        P0a = (u32)Pa;
        P1a = F | ((u32)(Pa >> 32) & 0x3Fu);
        P0b = (u32)Pb;
        P1b = F | ((u32)(Pb >> 32) & 0x3Fu);
    }
    xo_a = step_line_xo(P0a, P1a, lut);
    xo_b = step_line_xo(P0b, P1b, lut);
    return s.legal ? (s.cyc ? 2u : 1u) : 0u;
}

}  // namespace

#endif  // QTTT_STEP_CORE_H
