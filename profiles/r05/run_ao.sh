#!/bin/bash
# round 5, call AO: the long-line filter's restart table -- long-line parity, then the sweep (no reference) with it and (SEEQ_NO_WINDOW=1) with the absorbing table
set -o pipefail
out=$PWD/gpurun_out/r05_ao; mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_randomized.py -x -q -m gpu -k "long or sweep or stress or filter or fasta" > $out/pytest.log 2>&1; echo "pytest exit $?"; tail -4 $out/pytest.log | cut -c1-300
SEEQ_EXPLAIN=1 timeout -k 10 300 python profiles/chrom_sweep.py --no-ref > $out/sweep_restart.jsonl 2>$out/sweep_restart.txt; echo "sweep exit $?"; grep -v "^seeq plan\|amdgpu" $out/sweep_restart.txt
grep "^seeq plan" $out/sweep_restart.txt | grep "filter on long" | sed 's/.*m=\([0-9]*\) tau=\([0-9]*\).*-> \(k_stream[^|]*\).*p_acc \([0-9.e-]*\)).*/m=\1 tau=\2 \3 p_acc=\4/' | sort -u
SEEQ_NO_WINDOW=1 timeout -k 10 600 python profiles/chrom_sweep.py --no-ref > $out/sweep_absorb.jsonl 2>$out/sweep_absorb.txt; echo "sweep (absorbing) exit $?"; grep "k_stream  True" $out/sweep_absorb.txt
