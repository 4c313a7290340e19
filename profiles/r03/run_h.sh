#!/bin/bash
# Round 3, GPU call h: which phase of k_pair holds the time?  Timing-only variants (SEEQ_PAIR_EXP): 1 next tile requested
# after the walk, 2 no LDS gathers, 3 no bookkeeping, 4 no per-word checks; and 1 workgroup per CU.
set -u
O=gpurun_out/r03j; mkdir -p $O
export TMPDIR=/tmp
B="--steps 10 --warmup 3 --no-cpu-baseline --no-e2e --no-per-call --check-lines 0 --workload all"
run() { python3 -c "
import json,subprocess,os,sys
d=json.loads(subprocess.run([sys.executable,'bench.py']+'$B'.split(),capture_output=True,text=True).stdout)
print('$1', d['roofline']['kernel'], 'step', round(d['ms_per_step'],3), 'launch', round(d['roofline']['avg_launch_ms'],4), d['results']['matching_lines'])" | tee -a $O/exp.txt; }
for rep in 1 2; do
run exp0
SEEQ_PAIR_EXP=1 run exp1_prefetch
SEEQ_PAIR_EXP=5 run exp5_dbuf


SEEQ_DFA_WGS=1 run exp0_1wg
SEEQ_FUSED_KERNEL=stream run stream
done
