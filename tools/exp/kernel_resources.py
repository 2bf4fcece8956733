#!/usr/bin/env python3
"""Registers, scratch, LDS and the resulting waves per SIMD of every gfx950 kernel in a csrc file (working tree, product
flags), from the code-object metadata in the assembly.  CPU only.
usage: python tools/exp/kernel_resources.py [file.hip] [substring filter]"""
import os, re, subprocess, sys, tempfile

name = sys.argv[1] if len(sys.argv) > 1 else "gki_finder.hip"
flt = sys.argv[2] if len(sys.argv) > 2 else ""
root = subprocess.run(["git", "rev-parse", "--show-toplevel"], capture_output=True, text=True, check=True).stdout.strip()
csrc = os.path.join(root, "graph_kmer_index_amd", "csrc")
out = os.path.join(tempfile.mkdtemp(prefix="gki_res_"), "dev.s")
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-function", "--cuda-device-only", "-S",
                os.path.join(csrc, name), "-o", out] + sys.argv[3:], check=True, stderr=subprocess.DEVNULL)
txt = open(out).read()
# amdhsa.kernels metadata (YAML-ish): one block per kernel
rows = []
for blk in re.split(r"\n  - \.agpr_count:", txt)[1:]:
    blk = ".agpr_count:" + blk
    get = lambda key: (re.search(r"\.%s:\s*(\S+)" % key, blk) or [None, "?"])[1]
    sym = get("name")
    rows.append((sym, int(get("vgpr_count")), int(get("agpr_count")), int(get("sgpr_count")), int(get("private_segment_fixed_size")),
                 int(get("group_segment_fixed_size")), int(get("max_flat_workgroup_size"))))
dem = subprocess.run(["c++filt"] + [r[0] for r in rows], capture_output=True, text=True).stdout.strip().split("\n")
print("%-58s %5s %5s %8s %7s  %s" % ("kernel", "VGPR", "SGPR", "scratch", "LDS", "waves/SIMD (regs | LDS)"))
for r, d in sorted(zip(rows, dem), key=lambda x: x[1]):
    short = re.sub(r"\(anonymous namespace\)::|void ", "", d)
    short = re.sub(r"\((DevGraph|unsigned|int|long|const|NodeWalk|uint).*", "", short)
    if flt not in short:
        continue
    sym, vg, ag, sg, scr, lds, wg = r
    tot = vg + ag                                   # gfx90a+: unified 512-entry file per SIMD lane, granule 8
    by_regs = min(8, 512 // max(8, (tot + 7) // 8 * 8))
    waves_per_block = wg // 64
    by_lds = (163840 // lds) * waves_per_block // 4 if lds else 8
    print("%-58s %5d %5d %7dB %6dB  %d | %d" % (short[:58], tot, sg, scr, lds, by_regs, min(8, by_lds)))
