#!/bin/bash
# round 5, call AQ: the restart table under the long-line campaigns -- the stress test x 3 runs (fresh seeds, three variants each), the extended fuzz on long lines (four seed sets,
# two with foreign bytes), the kinds-of-lines fuzz (four seeds), then the sweep against the reference binary
out=$PWD/gpurun_out/r05_aq; mkdir -p $out
export TMPDIR=/tmp
rc=0
for i in 1 2 3; do timeout -k 10 600 python -m pytest tests/test_gpu_randomized.py -q -m gpu -k "stress or fuzz_long" -s 2>&1 | grep -a "SEEQ_FUZZ_SEED\|passed\|failed\|Error" | tr '\n' ' ' | cut -c1-600; echo; done
pids=""
FUZZ_SEED=800 timeout -k 10 900 python3 profiles/extended_fuzz.py long > $out/ext_a.log 2>&1 & pids="$pids $!"
FUZZ_SEED=812 timeout -k 10 900 python3 profiles/extended_fuzz.py long > $out/ext_b.log 2>&1 & pids="$pids $!"
FUZZ_SEED=824 FUZZ_FOREIGN=0.3 timeout -k 10 900 python3 profiles/extended_fuzz.py long > $out/ext_c.log 2>&1 & pids="$pids $!"
FUZZ_SEED=836 FUZZ_FOREIGN=0.1 timeout -k 10 900 python3 profiles/extended_fuzz.py long > $out/ext_d.log 2>&1 & pids="$pids $!"
for p in $pids; do wait $p || rc=1; done
for f in a b c d; do tail -1 $out/ext_$f.log | cut -c1-300; done
pids=""
for i in 1 2 3 4; do IGNORE_FUZZ_SEGMENTS=$(( i % 2 )) timeout -k 10 900 python3 profiles/ignore_fuzz.py $(( ( $(date +%s%N) / 1000 + i * 7919 ) % 1000000007 )) 40 > $out/fuzz_$i.log 2>&1 & pids="$pids $!"; done
for p in $pids; do wait $p || rc=1; done
for i in 1 2 3 4; do head -1 $out/fuzz_$i.log; tail -1 $out/fuzz_$i.log | cut -c1-300; done
echo "rc so far $rc"
timeout -k 10 900 python3 profiles/chrom_sweep.py > $out/chrom_sweep.jsonl 2> $out/chrom_sweep.txt || rc=1; grep -v amdgpu $out/chrom_sweep.txt
exit $rc
