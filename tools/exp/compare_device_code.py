#!/usr/bin/env python3
"""Which gfx950 kernels of a .hip file differ between two source states?  Compiles the file device-only to assembly at a
git revision and in the working tree (product flags), strips debug directives and comments, and compares per function.
usage: python tools/exp/compare_device_code.py <git rev> [file under graph_kmer_index_amd/csrc, default gki_finder.hip]"""
import os, re, subprocess, sys, tempfile

rev = sys.argv[1]
name = sys.argv[2] if len(sys.argv) > 2 else "gki_finder.hip"
root = subprocess.run(["git", "rev-parse", "--show-toplevel"], capture_output=True, text=True, check=True).stdout.strip()
csrc = os.path.join(root, "graph_kmer_index_amd", "csrc")
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-function", "--cuda-device-only", "-S",
         "-I", csrc, "-I", os.path.join(root, "include")]


def asm_of(src_text, tag):
    d = tempfile.mkdtemp(prefix="gki_cmp_")
    src = os.path.join(d, name)
    open(src, "w").write(src_text)
    out = os.path.join(d, tag + ".s")
    subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + [src, "-o", out], check=True, stderr=subprocess.DEVNULL)
    return open(out).read()


def funcs(txt):
    res = {}
    for m in re.finditer(r"\n(_Z\w+):[^\n]*\n(.*?)\n\.Lfunc_end\d+:", txt, re.S):
        lines = []
        for l in m.group(2).split("\n"):
            if re.match(r"\s*\.(loc|file|cfi)", l) or re.match(r"\s*;", l):
                continue
            l = l.split(";")[0].rstrip()
            l = re.sub(r"\.LBB\d+_", ".LBB_", l)       # block labels carry the function's index in the file
            l = re.sub(r"\.L(tmp|func_begin|func_end|JTI)\d+(_\d+)?", r".L\1", l)
            if l:
                lines.append(l)
        res[m.group(1)] = lines
    return res


old = subprocess.run(["git", "show", "%s:graph_kmer_index_amd/csrc/%s" % (rev, name)], capture_output=True, text=True, check=True).stdout
new = open(os.path.join(csrc, name)).read()
A, B = funcs(asm_of(old, "old")), funcs(asm_of(new, "new"))
assert A and B, "no functions parsed"
names = sorted(set(A) | set(B))
dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.strip().split("\n")
n_same = 0
for n, d in zip(names, dem):
    short = re.sub(r"\(anonymous namespace\)::|void ", "", d)
    short = re.sub(r"\)\(.*|\(DevGraph.*|\((unsigned|int|long|const|DevGraph).*", "", short)[:70]
    if A.get(n) == B.get(n):
        n_same += 1
    else:
        print("DIFFERS  %-70s %5d -> %5d lines" % (short, len(A.get(n, [])), len(B.get(n, []))))
print("%d functions, %d identical to %s" % (len(names), n_same, rev))
