#!/usr/bin/env python3
"""Instruction mix per basic block of one kernel in a gfx950 assembly dump.
usage: tools/asm_mix.py <file.s> <kernel-substring> [min_instrs]"""
import re, sys, collections
path, key = sys.argv[1], sys.argv[2]
mn = int(sys.argv[3]) if len(sys.argv) > 3 else 40
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.rstrip().endswith(":") or (l.startswith("_Z") and key in l and "; @" in l))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".Lfunc_end"))
def cat(op):
    if op.startswith("v_"):
        if "f64" in op: return "v_f64"
        if re.search(r"_f32|_f16", op) and not op.startswith("v_cvt") and not op.startswith("v_cmp"): return "v_f32"
        if op.startswith("v_cvt"): return "v_cvt"
        if op.startswith("v_cmp"): return "v_cmp"
        if op.startswith("v_cndmask"): return "v_sel"
        if op.startswith("v_mov") or op.startswith("v_accvgpr"): return "v_mov"
        if op.startswith(("v_readlane", "v_readfirstlane", "v_writelane", "v_permlane")): return "v_lane"
        return "v_int"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "scratch" if op.startswith("scratch_") else "vmem"
    if op.startswith("s_waitcnt"): return "s_wait"
    if op.startswith(("s_cbranch", "s_branch")): return "s_br"
    if op.startswith("s_load") or op.startswith("s_buffer"): return "smem"
    if op.startswith("s_"): return "salu"
    return "other"
blocks, cur, name = [], collections.Counter(), "entry"
for l in lines[start + 1:end]:
    s = l.strip()
    if not s or s.startswith(";") or s.startswith("."):
        m = re.match(r"^(\.LBB\d+_\d+):", s)
        if m:
            blocks.append((name, cur)); cur = collections.Counter(); name = m.group(1)
        continue
    op = s.split()[0]
    cur[cat(op)] += 1
blocks.append((name, cur))
tot = collections.Counter()
for n, c in blocks:
    tot.update(c)
    if sum(c.values()) >= mn:
        print(f"{n:12s} {sum(c.values()):5d} ", " ".join(f"{k}={v}" for k, v in sorted(c.items(), key=lambda x: -x[1])))
print("TOTAL", sum(tot.values()), " ".join(f"{k}={v}" for k, v in sorted(tot.items(), key=lambda x: -x[1])))
