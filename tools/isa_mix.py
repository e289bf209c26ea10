#!/usr/bin/env python3
"""Issue-time estimate of a kernel's inner loop from its ISA and the measured issue cost of the two VALU classes on
this part (profiles/r02/valu_rates.txt: 1.03 ns per wave-instruction per SIMD for plain logic / add / sub / right
shifts / moves / v_bitop3 on VGPRs; 1.75 ns for everything else, incl. any VALU instruction with an SGPR or literal
operand... here: classified by opcode and operands).  Static counts; the re-rooting walk's unrolled copies are
weighted so that the total matches the measured SQ_INSTS_VALU per ply.

    python tools/isa_mix.py step_random_fused_kernelILi256ELb1E 147.3
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAST = {"v_xor_b32", "v_and_b32", "v_or_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_lshrrev_b32", "v_ashrrev_i32",
        "v_not_b32", "v_mov_b32", "v_bitop3_b32", "v_add_co_u32", "v_xnor_b32"}
T_FAST, T_SLOW = 1.03, 1.75


def main():
    pat, measured = sys.argv[1], float(sys.argv[2])
    asm = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                          "-I" + os.path.join(ROOT, "include"), "-o", "-",
                          os.path.join(ROOT, "qtttgym_amd", "csrc", "qttt_kernels.hip")],
                         stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True).stdout
    lines = asm.splitlines()
    start = next(i for i, l in enumerate(lines) if re.match(r"^_ZN.*%s.*:\s*(;.*)?$" % re.escape(pat), l))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    body = lines[start:end]
    # the ply loop = the LAST loop header of the kernel (the table fills come first) up to its back edge
    heads = [i for i, l in enumerate(body) if "Loop Header" in l]
    h = heads[-1]
    label = body[h].split(":")[0].strip()
    prev = max(i for i in range(h) if re.match(r"^\.LBB\d+_\d+:", body[i]) and i < h)      # the loop's latch block precedes its header
    tail = next(i for i in range(h, len(body)) if re.search(r"s_cbranch\w+ %s\b|s_branch %s\b" % (re.escape(body[prev].split(":")[0]), re.escape(body[prev].split(":")[0])), body[i]))
    loop = body[prev:tail + 1]
    fast = slow = 0
    walk_fast = walk_slow = 0
    copies, in_walk = 0, False                 # the walk's further nodes: from the first s_and_saveexec to the first exec restore
    done_walk = False
    for l in loop:
        if "s_and_saveexec" in l and not done_walk:
            copies += 1
            in_walk = True
        if in_walk and re.match(r"^\s+s_or_b64 exec, exec", l):
            in_walk, done_walk = False, True
        m = re.match(r"^\s+(v_\w+?)(_e32|_e64|_sdwa|_dpp)?\s+(.*)$", l)
        if not m:
            continue
        op, enc, args = m.group(1), m.group(2) or "", m.group(3)
        rest = args.split(",", 1)[1] if "," in args else ""
        # an SGPR (or vcc) source puts a plain-logic instruction in the slow class (valu_rates: k_and_s, k_bitop3_s, k_mov_s
        # x1.7); a 32-bit LITERAL does not (k_and_lit, k_bitop3_c x0.92 - 0.95) — round 3 counted literals as slow
        scalar_operand = bool(re.search(r"\bs\d|s\[\d|\bvcc\b", rest.split(" bitop3")[0]))
        is_fast = op in FAST and enc != "_sdwa" and not scalar_operand
        if in_walk:
            walk_fast += is_fast
            walk_slow += not is_fast
        else:
            fast += is_fast
            slow += not is_fast
    copies = max(copies, 1)
    wf, ws = walk_fast / copies, walk_slow / copies
    static = fast + slow
    iters = (measured - static) / (wf + ws)
    t = (fast + iters * wf) * T_FAST + (slow + iters * ws) * T_SLOW
    print("ply loop %s: %d VALU outside the walk's further nodes (%d fast class, %d slow), %.1f per further node (%.1f fast, %.1f slow)"
          % (label, static, fast, slow, wf + ws, wf, ws))
    print("measured %.1f VALU per ply -> %.2f further nodes executed per wave and ply" % (measured, iters))
    print("issue time of that mix: %.1f ns per wave and ply (fast class %.2f ns, slow class %.2f ns per instruction per SIMD)" % (t, T_FAST, T_SLOW))
    print("share of the slow class: %.0f %%" % (100.0 * (slow + iters * ws) / measured))


if __name__ == "__main__":
    main()
