#!/usr/bin/env python3
"""Random 8-byte-load rate (gki_measure_random_loads) as a function of the table size.  The index build's row gather reads
32-byte rows at random from a 9.9 GB array and reaches 63 % of the rate measured on a 2 GB table -- is that reference rate
reachable at all on a table of that size?   usage: python tools/exp/random_rate_vs_table.py"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from graph_kmer_index_amd import _lib
lib = _lib.load()
for gb in (0.25, 1, 2, 4, 10, 20, 40):
    rates = []
    for _ in range(3):
        r = C.c_double(0.0)
        _lib.check(lib.gki_measure_random_loads(int(gb * (1 << 30)), 1 << 31, C.byref(r)))
        rates.append(r.value)
    print("table %5.2f GiB: %.3g random loads/s  (runs: %s)" % (gb, max(rates), ", ".join("%.3g" % x for x in rates)), flush=True)
