#!/usr/bin/env python3
"""Static look at a gather kernel's ISA for DESIGN 3.12 (container, no GPU).

usage: tools/isa_war_pairs.py <kernel.s> [window]

Counts, in the assembly of one kernel (hipcc --cuda-device-only -S output, cut to the kernel), the places where a vector-memory
load WRITES a VGPR that a VALU instruction issued at most `window` (default 4) instructions earlier READ as a source, with no
s_waitcnt / s_nop / branch between the two -- a write-after-read pair the hardware has to keep in order by itself (gfx9 needs
no software wait state for it: a VALU instruction reads its operands when it issues, the load's data returns later).  Also
prints how many accumulation instructions of each kind the kernel has and how many registers its loads target."""
import re
import sys

VREG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def regs(tok):
    out = set()
    for m in VREG.finditer(tok):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def parse(path):
    ins = []
    for line in open(path):
        line = line.split(";")[0].strip()
        if not line or line.endswith(":") or line.startswith("."):
            continue
        parts = line.split(None, 1)
        op = parts[0]
        ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        ins.append((op, ops))
    return ins


def main():
    path = sys.argv[1]
    window = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    ins = parse(path)
    kinds = {}
    for op, _ in ins:
        if op in ("v_pk_fma_f32", "v_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "buffer_load_dwordx4", "global_load_dwordx4"):
            kinds[op] = kinds.get(op, 0) + 1
    pairs = []
    for i, (op, ops) in enumerate(ins):
        if not (op.startswith("buffer_load") or op.startswith("global_load")) or not ops:
            continue
        dst = regs(ops[0])
        for back in range(1, window + 1):
            j = i - back
            if j < 0:
                break
            pop, pops = ins[j]
            if pop.startswith(("s_waitcnt", "s_nop", "s_cbranch", "s_branch", "s_barrier")):
                break
            if pop.startswith("v_") and len(pops) > 1:
                src = set().union(*[regs(o) for o in pops[1:]]) if pops[1:] else set()
                hit = dst & src
                if hit:
                    pairs.append((back, pop, " ".join(pops), op, ops[0]))
                    break
    print(f"{path}: {len(ins)} instructions; " + ", ".join(f"{k} x{v}" for k, v in sorted(kinds.items())))
    print(f"  load-overwrites-recent-VALU-source pairs (window {window}, nothing but plain instructions between): {len(pairs)}")
    by = {}
    for back, pop, pops, op, d in pairs:
        by[(pop, back)] = by.get((pop, back), 0) + 1
    for (pop, back), n in sorted(by.items()):
        print(f"    {pop:16s} read ... {back} instruction(s) later a load overwrites the register: {n}")
    for p in pairs[:4]:
        print(f"    e.g.  {p[1]} {p[2]}   ->   {p[3]} {p[4]}  ({p[0]} later)")


if __name__ == "__main__":
    main()
