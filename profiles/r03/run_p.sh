#!/bin/bash
# Round 3, GPU call p: the post-pass of segment k on a second stream under k_pair of segment k+1 (A/B: SEEQ_OVERLAP=0).
set -u
O=gpurun_out/r03p; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "(batch_scan_vs_oracle and pair) or (edge_buffers and pair) or (stream_fuzz_patterns and (pair or auto)) or chunk_and_tile or shard or smoke" > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log
tail -5 $O/pytest.log
for rep in 1 2; do
for mode in best count all; do
python profiles/time_scan.py overlap_$mode 100000000 10 $mode | tee -a $O/ab.txt
SEEQ_OVERLAP=0 python profiles/time_scan.py serial_$mode 100000000 10 $mode | tee -a $O/ab.txt
done
done
SEEQ_TS_PATTERN='GATG[TA]AGCNCGATTAGC[CG]TGAAAATGNGAGTAC[GAT]GCGCGA' SEEQ_TS_TAU=5 SEEQ_TS_LEN=250 python profiles/time_scan.py cfg5_all_overlap 100000000 5 all | tee -a $O/ab.txt
SEEQ_OVERLAP=0 SEEQ_TS_PATTERN='GATG[TA]AGCNCGATTAGC[CG]TGAAAATGNGAGTAC[GAT]GCGCGA' SEEQ_TS_TAU=5 SEEQ_TS_LEN=250 python profiles/time_scan.py cfg5_all_serial 100000000 5 all | tee -a $O/ab.txt
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-e2e --no-per-call > $O/bench_best.json 2> $O/bench_best.err; python3 -c "
import json; d=json.load(open('$O/bench_best.json')); print('bench best', d['roofline']['kernel'], round(d['ms_per_step'],3), d['device_ms_per_step'], d['roofline']['avg_launch_ms'], d['results']['oracle_check'])"
