#!/bin/bash
# usage: tools/exp/check_then_ab.sh "<pytest -k expression>"  -> (1) the named GPU tests against the BASE build
# (tools/exp/_ab/libgki_base.so: is the test itself right?), (2) the same tests and then the whole GPU suite against the
# new build, (3) only if all of that is green, the same-box A/B of tools/exp/ab_all_nodes.sh.
set -u
R="$(pwd)"; L="$R/graph_kmer_index_amd/libgki_hip.so"; K="$1"
cp "$L" /tmp/gki_new_keep.so
cp "$R/tools/exp/_ab/libgki_base.so" "$L"
timeout -k 10 200 python -m pytest tests -x -q -m gpu -k "$K" > gpurun_out/cta_base.log 2>&1; rb=$?
echo "targeted tests, BASE build: rc=$rb  $(tail -1 gpurun_out/cta_base.log)"
cp /tmp/gki_new_keep.so "$L"
timeout -k 10 200 python -m pytest tests -x -q -m gpu -k "$K" > gpurun_out/cta_new.log 2>&1; rn=$?
echo "targeted tests, NEW build:  rc=$rn  $(tail -1 gpurun_out/cta_new.log)"
[ $rn -eq 0 ] || { tail -30 gpurun_out/cta_new.log; exit 1; }
timeout -k 10 600 python -m pytest tests -x -q -m gpu > gpurun_out/cta_full.log 2>&1; rf=$?
echo "whole GPU suite, NEW build: rc=$rf  $(tail -1 gpurun_out/cta_full.log)"
[ $rf -eq 0 ] || { tail -30 gpurun_out/cta_full.log; exit 1; }
tools/exp/ab_all_nodes.sh
