#!/usr/bin/env python3
"""Static instruction mix of one kernel of the gfx950 assembly (hipcc --cuda-device-only -S of jda_kernels.hip):
python tools/isa_mix.py k.s '<2,1,1,0>' [--spills]   -> counts per class, SGPR-spill traffic (v_readlane / v_writelane) per basic block."""
import collections
import re
import sys

MANGLED = {"<2,1,1,0>": "_Z27jda_decode_tiles_persistentILi2ELi1ELi1ELi0EEvPK12jda_dev_descPK9jda_stripjj",
           "<1,1,1,0>": "_Z27jda_decode_tiles_persistentILi1ELi1ELi1ELi0EEvPK12jda_dev_descPK9jda_stripjj",
           "<0,1,3,0>": "_Z27jda_decode_tiles_persistentILi0ELi1ELi3ELi0EEvPK12jda_dev_descPK9jda_stripjj"}


def kernel_lines(path, name):
    sym = MANGLED.get(name, name)
    out, on = [], False
    for line in open(path):
        if line.startswith(sym + ":"):
            on = True
            continue
        if on:
            if line.startswith("\t.section") or line.startswith(".Lfunc_end"):
                break
            out.append(line.rstrip("\n"))
    return out


def main():
    path, name = sys.argv[1], sys.argv[2]
    lines = kernel_lines(path, name)
    cls = collections.Counter()
    ops = collections.Counter()
    block, per_block = "entry", collections.OrderedDict()
    for ln in lines:
        s = ln.strip()
        if not s or s.startswith(";") or s.startswith("."):
            if re.match(r"^\.LBB\d+_\d+:", s):
                block = s.split(":")[0]
            continue
        op = s.split()[0]
        ops[op] += 1
        k = ("valu" if op.startswith("v_") else "salu" if op.startswith("s_") and not op.startswith(("s_load", "s_waitcnt", "s_cbranch", "s_branch", "s_barrier", "s_nop", "s_setprio", "s_endpgm")) else
             "lds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "smem" if op.startswith("s_load") else
             "branch" if op.startswith(("s_cbranch", "s_branch")) else "wait" if op.startswith("s_waitcnt") else "other")
        cls[k] += 1
        if op in ("v_readlane_b32", "v_writelane_b32"):
            per_block.setdefault(block, collections.Counter())[op] += 1
    print("kernel %s: %d instructions" % (name, sum(cls.values())))
    for k, v in cls.most_common():
        print("  %-8s %5d" % (k, v))
    for op in ("v_mov_b32_e32", "v_readlane_b32", "v_writelane_b32", "v_readfirstlane_b32", "s_and_saveexec_b64", "s_cbranch_execz", "s_waitcnt", "v_cndmask_b32_e32"):
        print("  %-22s %5d" % (op, ops[op]))
    if "--spills" in sys.argv:
        for b, c in per_block.items():
            print("  %-12s readlane %3d writelane %3d" % (b, c["v_readlane_b32"], c["v_writelane_b32"]))
    if "--top" in sys.argv:
        for op, n in ops.most_common(40):
            print("  %-28s %5d" % (op, n))


if __name__ == "__main__":
    main()
