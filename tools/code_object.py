#!/usr/bin/env python3
"""The kernels of the product library's gfx950 code object: python tools/code_object.py [lib.so] -> one line per kernel
(symbol, demangled name, VGPRs, SGPRs, SGPR spills, VGPR spills, LDS, scratch) from the code object's metadata note.
Needs /opt/rocm/lib/llvm/bin (clang-offload-bundler, llvm-objcopy, llvm-readelf); also used by tests/test_gpu_zz_kernel_coverage.py
for the list of kernel symbols."""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def kernels(lib):
    """[{symbol, name, vgpr, sgpr, sgpr_spill, vgpr_spill, lds, scratch}] of lib's gfx950 code object"""
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "k.co")
        subprocess.run([LLVM + "/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
        subprocess.run([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + fat,
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], check=True)
        notes = subprocess.run([LLVM + "/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
    out, cur = [], None
    keys = {".name": "symbol", ".vgpr_count": "vgpr", ".sgpr_count": "sgpr", ".sgpr_spill_count": "sgpr_spill",
            ".vgpr_spill_count": "vgpr_spill", ".group_segment_fixed_size": "lds", ".private_segment_fixed_size": "scratch"}
    for line in notes.splitlines():
        m = re.match(r"\s*(?:- )?(\.[a-z_]+):\s*(.*)$", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip().strip("'\"")
        if line.lstrip().startswith("- .agpr_count") or (line.lstrip().startswith("- ") and k in (".args", ".agpr_count")):
            pass
        if k == ".symbol" and v.endswith(".kd"):
            cur = {"kd": v}
            out.append(cur)
        elif cur is None and k == ".name":
            pass
        if k in keys:
            # metadata entries of one kernel are contiguous; .symbol closes the group in llvm's output order, so collect and attach
            pending.setdefault(k, v) if False else None
    # simpler and order-independent: split the note's kernel list on '- .agpr_count' / '- .args' group starts
    groups = re.split(r"\n\s*- (?=\.(?:agpr_count|args):)", notes)
    out = []
    for g in groups[1:]:
        e = {}
        for k, name in keys.items():
            m = re.search(r"^\s*" + re.escape(k) + r":\s*(.*)$", g, re.M)
            if m:
                v = m.group(1).strip().strip("'\"")
                e[name] = int(v) if v.isdigit() else v
        if "symbol" in e:
            out.append(e)
    names = subprocess.run(["c++filt"] + [e["symbol"] for e in out], check=True, capture_output=True, text=True).stdout.splitlines()
    for e, n in zip(out, names):
        e["name"] = re.sub(r"\(.*$", "", n).replace("void ", "")
    return sorted(out, key=lambda e: e["name"])


def main():
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "jpegdec_amd", "libjpegdec_amd.so")
    ks = kernels(lib)
    print("%d kernels in %s" % (len(ks), os.path.relpath(lib, here)))
    print("%-58s %5s %5s %7s %7s %7s %7s" % ("kernel", "vgpr", "sgpr", "s-spill", "v-spill", "lds", "scratch"))
    for e in ks:
        print("%-58s %5s %5s %7s %7s %7s %7s" % (e["name"], e.get("vgpr"), e.get("sgpr"), e.get("sgpr_spill"), e.get("vgpr_spill"), e.get("lds"), e.get("scratch")))


if __name__ == "__main__":
    main()
