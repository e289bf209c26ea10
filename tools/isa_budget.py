#!/usr/bin/env python3
"""Per-phase VALU budget of a kernel, from the ISA the compiler really emits.

The library is compiled once more with `-gline-tables-only -S` (line tables do not change code generation: the
instruction stream is checked against the plain build's, instruction for instruction).  Every instruction then carries
the source line it came from, with its inlining chain; the sources carry `//@isa <phase>` tags (a tag holds from its
line to the next tag of the same file), so each VALU instruction of the kernel falls into one phase of the step:

    lane      per lane, shared by its boards: ids, offsets, loads, unpack / pack / stores
    table     this thread's entry of the 3-in-a-row LDS table (computed while the loads are in flight)
    hash      the counter hash: collapse bit (and the policy's word) — in front of the barrier, off the critical path
    policy    the in-kernel uniform-legal policy (SAMPLE / fused kernels only)
    reset     auto-reset: a finished board restarts empty
    decode    action decode, sort, validity (board.py:10-18; env.py:41)
    comps     components of lo / hi from the cached qstructs, cycle test (board.py:28-42)
    childend  which end is re-rooted (qeval.py:35 on a cycle)
    walk0     first node of the path reversal (always executed)
    walk+     one FURTHER node of the path reversal (executed while any lane of the wave still walks)
    append    moves.append: x into Q0 (board.py:19)
    qstructs  insert / union / pop in list order (board.py:42-69)
    fields    P1: chi, last x, classical |= component, n += 1
    line      X / O masks, two table reads, done bit (board.py:71-115, env.py:49,51)
    pack      reward / terminated words, register moves for the stores

and into one of the two issue classes measured on this part (tools/valu_rates.cpp, profiles/r02/valu_rates.txt): FAST
(1.03 ns per wave-instruction per SIMD: plain logic, add / sub, right shifts, moves, v_bitop3 on VGPRs / inline
constants / literals) or SLOW (1.75 ns: left shifts, bfe, compares, selects, multiplies, 64-bit shifts, ffbl / bcnt,
SDWA, three-operand or / and-or / lshl-or, anything with an SGPR source).

    python tools/isa_budget.py 'step_kernelILi1024ELi2ELb0ELb1ELb0ELb0ELb0E' 2 [measured SQ_INSTS_VALU per board] [-v]
    python tools/isa_budget.py 'step_random_fused_kernelILi256ELb1ELb0E' 1 138.0 --loop        (the ply loop only)

Arguments: a substring of the kernel's mangled name, boards per lane of that instantiation (counts are divided by
it), and optionally the measured VALU instructions per board-step (profiles/rNN/pmc_sq_summary.csv): the number of
further walk nodes executed per wave is then solved from it, otherwise 3.4 (the 64-lane maximum of the walk's depth
under the uniform-legal policy, tools/walk_depth_sim.py).  -v lists every instruction under its phase.
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "qtttgym_amd", "csrc")
FAST = {"v_xor_b32", "v_and_b32", "v_or_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_lshrrev_b32", "v_ashrrev_i32",
        "v_not_b32", "v_mov_b32", "v_bitop3_b32", "v_add_co_u32", "v_bitop3_b16"}
T_FAST, T_SLOW = 1.03, 1.75
GENERIC = {"other", "glue", "lane"}
ORDER = ["lane", "table", "hash", "policy", "reset", "decode", "comps", "childend", "walk0", "walk+", "append", "qstructs",
         "fields", "line", "pack", "glue", "other"]
FLAGS = ["-O3", "-std=c++17", "--cuda-device-only", "-mllvm", "-amdgpu-kernarg-preload-count=4", "-I" + os.path.join(ROOT, "include")]


def tags_of(path):
    out = []
    for i, l in enumerate(open(path), 1):
        m = re.match(r"\s*//@isa (\w+)", l)
        if m:
            out.append((i, m.group(1)))
    return out


def phase_at(tags, path, line):
    cur = "other"
    for ln, t in tags.get(os.path.basename(path), []):
        if ln <= line:
            cur = t
    return cur


def compile_s(extra):
    return subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-S", "-o", "-", os.path.join(CSRC, "qttt_kernels.hip")] + FLAGS + extra,
                          stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, check=True).stdout.splitlines()


def kernel_body(lines, pat):
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*%s\w*:" % re.escape(pat), l))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    return lines[start:end + 1]


def is_instr(l):
    return re.match(r"^\s+[a-z]\w+", l) and not re.match(r"^\s+\.", l)


def classify(op, enc, args):
    rest = args.split(",", 1)[1] if "," in args else ""
    scalar = bool(re.search(r"\bs\d|s\[\d|\bvcc\b", rest.split(" bitop3")[0]))
    return op in FAST and enc != "_sdwa" and not scalar


def main():
    argv = [a for a in sys.argv[1:] if a not in ("-v", "--loop")]
    verbose = "-v" in sys.argv
    loop_only = "--loop" in sys.argv        # count the kernel's LAST depth-1 loop only (the ply loop of the fused kernels)
    pat, bpl = argv[0], int(argv[1])
    measured = float(argv[2]) if len(argv) > 2 else None
    tags = {f: tags_of(os.path.join(CSRC, f)) for f in os.listdir(CSRC) if f.endswith((".h", ".hip"))}
    plain = [l.split(";")[0].rstrip() for l in kernel_body(compile_s([]), pat) if is_instr(l)]
    dbg = kernel_body(compile_s(["-gline-tables-only"]), pat)
    if [l.split(";")[0].rstrip() for l in dbg if is_instr(l)] != plain:
        sys.exit("the line-table build's instruction stream differs from the plain build's")
    if loop_only:
        labels = [(i, l) for i, l in enumerate(dbg) if re.match(r"^\.LBB\d+_\d+:", l)]
        head = [l.split(":")[0] for i, l in labels if "Loop Header: Depth=1" in l][-1]                  # ".LBB89_15"
        tag = "Header=" + head[2:]
        inside = [k for k, (i, l) in enumerate(labels) if l.startswith(head + ":") or tag in l]
        first, last = labels[inside[0]][0], (labels[inside[-1] + 1][0] if inside[-1] + 1 < len(labels) else len(dbg))
        # the location in force at the loop's first instruction is the last .loc in front of it
        loc = next((l for l in reversed(dbg[:first]) if re.match(r"^\s+\.loc\s", l)), None)
        dbg = ([loc] if loc else []) + dbg[first:last]
    per = {}            # phase -> [fast, slow, instrs]
    salu = lds = vmem = 0
    cur = "other"
    in_walk, node, nodes_seen = False, 0, []
    for l in dbg:
        m = re.match(r"^\s+\.loc\s+\d+\s+\d+\s+\d+.*?;\s*(.*)$", l)
        if m:
            frames = re.findall(r"([^\s:\[\]@]+):(\d+):\d+", m.group(1))    # inner -> outer
            ph = [phase_at(tags, f, int(n)) for f, n in frames]
            cur = next((p for p in ph if p not in GENERIC), ph[0] if ph else "other")
            continue
        if not is_instr(l):
            continue
        txt = l.split(";")[0].strip()
        op = txt.split()[0]
        if op.startswith("s_and_saveexec") and in_walk:
            node += 1
        if op.startswith("s_"):
            salu += 1
            continue
        if op.startswith("ds_"):
            lds += 1
            continue
        if op.startswith(("global_", "buffer_", "flat_")):
            vmem += 1
            continue
        m = re.match(r"^(v_\w+?)(_e32|_e64|_sdwa|_dpp)?\s+(.*)$", txt)
        if not m:
            continue
        ph = cur
        if ph == "walk":
            if not in_walk:
                in_walk, node = True, 0
            ph = "walk0" if node == 0 else "walk+"
            if node:
                nodes_seen.append(node)
        elif ph in ("append", "qstructs", "fields", "line"):
            in_walk = False
        fast = classify(m.group(1), m.group(2) or "", m.group(3))
        e = per.setdefault(ph, [0, 0, []])
        e[0 if fast else 1] += 1
        e[2].append(("F " if fast else "S ") + txt)
    n_further = len(set(nodes_seen)) * bpl if nodes_seen else 0          # unrolled further nodes over all boards of the lane
    further = per.get("walk+", [0, 0, []])
    per_node = (further[0] + further[1]) / n_further if n_further else 0.0
    per_node_f = further[0] / n_further if n_further else 0.0
    per_node_s = further[1] / n_further if n_further else 0.0
    static = sum(v[0] + v[1] for k, v in per.items() if k != "walk+") / bpl
    iters = (measured - static) / per_node if (measured and per_node) else 3.4
    print("kernel *%s*: %d VALU instructions in the binary (%d boards per lane), %d SALU, %d LDS, %d VMEM" %
          (pat, sum(v[0] + v[1] for v in per.values()), bpl, salu, lds, vmem))
    print("per board-step; further walk nodes executed per wave: %.2f%s" % (iters, " (solved from the measured %.1f)" % measured if measured else " (assumed)"))
    print("%-9s %7s %6s %6s %8s" % ("phase", "VALU", "fast", "slow", "issue ns"))
    tot = tf = ts = 0.0
    for ph in ORDER + sorted(k for k in per if k not in ORDER):
        if ph not in per:
            continue
        f, s = per[ph][0], per[ph][1]
        if ph == "walk+":
            f, s = per_node_f * iters, per_node_s * iters
            label = "%-9s" % ("walk+ x%.1f" % iters)
        else:
            f, s = f / bpl, s / bpl
            label = "%-9s" % ph
        print("%s %7.1f %6.1f %6.1f %8.1f" % (label, f + s, f, s, f * T_FAST + s * T_SLOW))
        tot += f + s; tf += f; ts += s
    print("%-9s %7.1f %6.1f %6.1f %8.1f" % ("total", tot, tf, ts, tf * T_FAST + ts * T_SLOW))
    print("(one further node = %.1f VALU: %.1f fast + %.1f slow)" % (per_node, per_node_f, per_node_s))
    if measured and "table" in per and not loop_only:
        print("(the counts are static: a phase only SOME waves of a workgroup execute — the line table's entries, threads < 512 —\n"
              " is counted in full, so the number of further nodes solved from the measured total is a lower bound; the fused kernel's\n"
              " ply loop, where every wave runs everything, solves to the walk's true 64-lane maximum)")
    if verbose:
        for ph in ORDER + sorted(k for k in per if k not in ORDER):
            if ph in per:
                print("\n[%s]" % ph)
                seen = per[ph][2]
                if ph == "walk+" and n_further:
                    seen = seen[:int(round(per_node))]                  # one node: the others are copies
                for t in seen:
                    print("   " + t)


if __name__ == "__main__":
    main()
