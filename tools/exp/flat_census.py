#!/usr/bin/env python3
"""FLAT memory instructions per gfx950 kernel of the library (CPU only).  A flat load/store means the compiler did not
know the pointer's address space (a struct of pointers escaped to memory, an LDS-or-global select was merged, a generic
pointer crossed an out-of-line call); on gfx950 it counts against the LDS wait counter as well as the memory one.
usage: python tools/exp/flat_census.py [file.hip ...]   (default: every .hip under graph_kmer_index_amd/csrc)"""
import glob, os, re, subprocess, sys, tempfile

root = subprocess.run(["git", "rev-parse", "--show-toplevel"], capture_output=True, text=True, check=True).stdout.strip()
csrc = os.path.join(root, "graph_kmer_index_amd", "csrc")
files = sys.argv[1:] or sorted(os.path.basename(f) for f in glob.glob(os.path.join(csrc, "*.hip")))
tmp = tempfile.mkdtemp(prefix="gki_flat_")
total = 0
for name in files:
    out = os.path.join(tmp, name + ".s")
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-function", "--cuda-device-only", "-S",
                    os.path.join(csrc, name), "-o", out], check=True, stderr=subprocess.DEVNULL)
    txt = open(out).read()
    rows = []
    for m in re.finditer(r"\n(_Z\w+):[^\n]*\n(.*?)\n\.Lfunc_end\d+:", txt, re.S):
        body = m.group(2)
        c = lambda p: len(re.findall(r"\n\s*" + p, body))
        rows.append((m.group(1), c("flat_load"), c("flat_store"), c("flat_atomic"), c("global_load"), c("ds_read"), c("scratch_load")))
    dem = subprocess.run(["c++filt"] + [r[0] for r in rows], capture_output=True, text=True).stdout.strip().split("\n") if rows else []
    print("== %s: %d device functions" % (name, len(rows)))
    for r, d in zip(rows, dem):
        if r[1] + r[2] + r[3] == 0:
            continue
        total += r[1] + r[2] + r[3]
        short = re.sub(r"\(anonymous namespace\)::|void ", "", d)
        short = re.sub(r"\((DevGraph|unsigned|int|long|const|NodeWalk|uint|ProbeDev|IndexDev|WalkSrc|HIP_vector).*", "", short)[:64]
        print("   %-64s flat_load %2d  flat_store %2d  flat_atomic %2d   (global_load %2d, ds_read %2d, scratch_load %2d)" % ((short,) + r[1:]))
print("flat instructions in total:", total)
