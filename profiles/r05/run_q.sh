#!/bin/bash
# round 5, call Q: the Myers mode steps only the warm-up words its pattern needs (m + tau - 1 bytes): long-line parity + the sweep (no reference)
set -o pipefail
out=$PWD/gpurun_out/r05_q; mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_randomized.py -x -q -m gpu -k "long or sweep or myers or chrom or stress or forced" > $out/pytest.log 2>&1 || { tail -60 $out/pytest.log; exit 1; }
tail -5 $out/pytest.log
timeout -k 10 900 python profiles/chrom_sweep.py --no-ref > $out/sweep_noref.jsonl 2>$out/sweep_noref.txt || { tail -5 $out/sweep_noref.txt; exit 1; }
cat $out/sweep_noref.txt
