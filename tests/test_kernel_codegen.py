"""Properties of the compiled gfx950 kernels that the measured performance rests on (CPU only: hipcc cross-compiles).

DESIGN.md 4.2: (1) no FLAT memory instruction in any kernel -- a struct of pointers that escapes to an out-of-line call
turns every load through it into a FLAT load, which waits on the LDS counter as well as the memory counter (cost the
general count kernel 22 %); only the out-of-line history_ok, whose pointers cross a call boundary, has them.
(2) the walk kernels are latency-bound, their speed is the number of resident waves: the register and LDS footprints that
give the measured occupancies are pinned here."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "graph_kmer_index_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"

pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC) or shutil.which("c++filt") is None, reason="needs hipcc and c++filt")


@pytest.fixture(scope="module")
def finder_asm(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("codegen") / "gki_finder.s")
    subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-function", "--cuda-device-only", "-S",
                    os.path.join(CSRC, "gki_finder.hip"), "-o", out], check=True, stderr=subprocess.DEVNULL)
    return open(out).read()


@pytest.fixture(scope="module")
def forward_asm(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("codegen_fw") / "gki_forward.s")
    subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-function", "--cuda-device-only", "-S",
                    os.path.join(CSRC, "gki_forward.hip"), "-o", out], check=True, stderr=subprocess.DEVNULL)
    return open(out).read()


def _functions(txt):
    found = {m.group(1): m.group(2) for m in re.finditer(r"\n(_Z\w+):[^\n]*\n(.*?)\n\.Lfunc_end\d+:", txt, re.S)}
    names = list(found)
    dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True, check=True).stdout.strip().split("\n")
    return {re.sub(r"\(anonymous namespace\)::|void ", "", d): found[n] for n, d in zip(names, dem)}


def _resources(txt):
    res = {}
    for blk in re.split(r"\n  - \.agpr_count:", txt)[1:]:
        blk = ".agpr_count:" + blk
        get = lambda key: re.search(r"\.%s:\s*(\S+)" % key, blk).group(1)
        res[get("name")] = dict(vgpr=int(get("vgpr_count")) + int(get("agpr_count")), lds=int(get("group_segment_fixed_size")),
                                scratch=int(get("private_segment_fixed_size")))
    names = list(res)
    dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True, check=True).stdout.strip().split("\n")
    return {re.sub(r"\(anonymous namespace\)::|void ", "", d).split("(")[0]: res[n] for n, d in zip(names, dem)}


def test_no_flat_memory_instructions_outside_history_ok(finder_asm):
    funcs = _functions(finder_asm)
    assert len(funcs) >= 35
    offenders = {}
    for name, body in funcs.items():
        n = len(re.findall(r"\n\s*flat_(load|store|atomic)", body))
        short = name.split("(")[0]
        # (the slow path for deep windows -- last template argument true -- selects between a register level and an
        # arena level through one generic pointer; it is not where the time goes)
        deep = (short.startswith("k_emit_boundary_one<") and short.count(",") == 4 and short.endswith(", true>")) or \
               (short.startswith("k_count_boundary<") and short.count(",") == 2 and short.endswith(", true>"))
        if n and not name.startswith("history_ok") and not deep:
            offenders[short] = n
    assert not offenders, "FLAT memory instructions (address space lost): %s" % offenders


def test_occupancy_footprints_of_the_walk_kernels(finder_asm):
    r = _resources(finder_asm)
    waves_by_regs = lambda v: min(8, 512 // ((v + 7) // 8 * 8))
    blocks_by_lds = lambda b: 163840 // b
    # count pass: 8 waves per SIMD in every variant (the general one is held there by __launch_bounds__)
    # (last template argument: false = the product kernels; true = the slow path for windows deeper than their stacks)
    for v in ("k_count_boundary<false, false, false>", "k_count_boundary<true, false, false>", "k_count_boundary<true, true, false>",
              "k_count_boundary<false, true, false>"):
        assert waves_by_regs(r[v]["vgpr"]) == 8, (v, r[v])
        assert blocks_by_lds(r[v]["lds"]) >= 8, (v, r[v])
    # emit pass: workgroups of ONE wave since round 3 (the hardware's scheduling unit; the waves never talk to each other), so
    # the LDS figure is per wave and 16 workgroups per CU are what 4 workgroups of 4 waves were
    blocks_by_lds = lambda b: 163840 // b // 4
    # emit pass, flat layouts (FMT 0 / 2): 4 workgroups per CU in one-node mode, and since round 3 in all-nodes mode too
    # (per-lane facts shuffled instead of staged, 32-bit record slots, node lists of five, no flag words: 40.9 KB); the
    # all-nodes variant of runs with lossy restart points keeps the flag words and stays at 3; registers never the limit
    for lossy in ("false", "true"):
        for fmt in ("0", "2"):
            one = r["k_emit_boundary_one<%s, %s, false, false, false>" % (lossy, fmt)]
            allm = r["k_emit_boundary_one<%s, %s, true, false, false>" % (lossy, fmt)]
            assert blocks_by_lds(one["lds"]) == 4 and waves_by_regs(one["vgpr"]) >= 4, (lossy, fmt, one)
            want = 4 if lossy == "false" else 3
            assert blocks_by_lds(allm["lds"]) == want and waves_by_regs(allm["vgpr"]) >= want, (lossy, fmt, allm)
    # the slow path keeps its stacks out of scratch: a global-memory arena sized for the run
    for name, res in r.items():
        if name.endswith(", true>") and (name.startswith("k_emit_boundary_one<") or name.startswith("k_count_boundary<")) and name.count(",") == (4 if "emit" in name else 2):
            assert res["scratch"] <= 256, (name, res)
    for lossy in ("false", "true"):            # general variants, with and without the lossy-restart logic
        gen = r["k_emit_boundary_one<%s, 2, false, true, false>" % lossy]
        assert blocks_by_lds(gen["lds"]) == 4 and waves_by_regs(gen["vgpr"]) >= 4, (lossy, gen)


def test_forward_search_kernel_runs_at_full_occupancy(forward_asm):
    # the early-stop search reads everything from global memory, one lane per start position: left alone the compiler
    # builds it with 178 VGPRs (2 waves per SIMD) and it is 1.6-2.3x slower (csrc/gki_forward.hip, tools/bench_forward.py)
    r = _resources(forward_asm)
    # (second argument true: the slow path for deep windows; third true: the count pass that writes the emit pass's script)
    for v in ("k_forward<false, false, false>", "k_forward<true, false, false>", "k_forward<false, false, true>", "k_forward_expand"):
        assert r[v]["vgpr"] <= 64, (v, r[v])
    for name, body in _functions(forward_asm).items():
        if name.split("(")[0].startswith("k_forward<") and name.split("(")[0].split(",")[1].strip() == "true":
            continue                  # the slow path
        assert not re.findall(r"\n\s*flat_(load|store|atomic)", body), name
        # the first levels of the walk are registers: what is left in scratch are the levels behind them, addressed by a
        # level index in a VGPR.  (Held in one aggregate with that stack they were 25 scratch stores, most of them at fixed
        # offsets, and the search 1.3x slower: profiles/r04_forward_node_records_ab.txt.)
        if name.split("(")[0].startswith("k_forward<"):
            assert len(re.findall(r"\n\s*scratch_store", body)) <= 12, (name, len(re.findall(r"\n\s*scratch_store", body)))
            assert len(re.findall(r"\n\s*scratch_store_\w+ off,", body)) <= 2, name


@pytest.fixture(scope="module")
def index_rows_asm(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("codegen_ix") / "gki_index_rows.s")
    subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-function", "--cuda-device-only", "-S",
                    os.path.join(CSRC, "gki_index_rows.hip"), "-o", out], check=True, stderr=subprocess.DEVNULL)
    return open(out).read()


def test_footprints_of_the_index_build_kernels(index_rows_asm):
    """the finish runs four workgroups of 512 threads per CU (<= 40 KB of LDS, <= 64 VGPRs), the bucket-range
    partition two workgroups of 256 threads on 2048-row tiles (<= 80 KB), the build's passes one workgroup on a 4096-row
    tile staged in four parts (<= 80 KB: two workgroups per CU), the 4096-row finish of the grouped build sixteen waves
    (<= 128 VGPRs)."""
    r = _resources(index_rows_asm)
    # (the second pass holds a slot per payload word at the 128-VGPR cap of two workgroups per CU: three dwords spill)
    assert all(v["scratch"] <= (16 if k.startswith("k_partition_rows_staged<512, 8, false") else 0) for k, v in r.items()), \
        {k: v["scratch"] for k, v in r.items() if v["scratch"]}
    # the build's passes: two workgroups of 512 threads per CU (<= 80 KB of LDS, <= 128 VGPRs)
    for name in ("k_partition_rows_staged<512, 8, true, 4>", "k_partition_rows_staged<512, 8, false, 4>"):
        assert r[name]["lds"] <= 81920 and r[name]["vgpr"] <= 128, (name, r[name])
    for name in ("k_group_finish<true, 1024, 512>", "k_group_finish<false, 1024, 512>"):
        assert r[name]["lds"] <= 40960 and r[name]["vgpr"] <= 64, (name, r[name])
    for name in ("k_group_finish<true, 4096, 1024>", "k_group_finish<false, 4096, 1024>"):
        assert r[name]["lds"] <= 163840 and r[name]["vgpr"] <= 128, (name, r[name])
    assert r["k_partition_rows<256, 8, true, true>"]["lds"] <= 81920 and r["k_partition_rows<256, 8, true, true>"]["vgpr"] <= 128
    for name in ("k_partition_rows<512, 8, true, false>", "k_partition_rows<512, 8, true, true>"):      # (rows out beyond 2^32 records; > 256 parts into columns)
        assert r[name]["lds"] <= 163840 and r[name]["vgpr"] <= 256, (name, r[name])
    for name in ("k_kmer_digit_hist<512, 8>", "k_kmer_digit_hist<256, 8>", "k_digit_hist<512, 8>"):
        assert r[name]["vgpr"] <= 64 and r[name]["lds"] <= 16384, (name, r[name])      # 8 waves per SIMD; the LDS leaves the thread limit (4 x 512) in charge
