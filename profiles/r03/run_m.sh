#!/bin/bash
# Round 3, GPU call m: k_verify (lane queue) + k_emit_copy behind k_pair: parity tests, then A/B against k_exact1 COUNT / EMIT.
set -u
O=gpurun_out/r03m; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "(batch_scan_vs_oracle and pair) or (edge_buffers and pair) or (stream_fuzz_patterns and (pair or auto)) or chunk_and_tile or shard or smoke or filematch or cli_golden" > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log
tail -5 $O/pytest.log
for rep in 1 2; do
for mode in best count; do
python profiles/time_scan.py verify_$mode 100000000 10 $mode | tee -a $O/ab.txt
SEEQ_NO_VERIFY=1 python profiles/time_scan.py exact1_$mode 100000000 10 $mode | tee -a $O/ab.txt
done
SEEQ_TS_PATTERN='GATG[TA]AGCNCGATTAGC[CG]TGAAAATGNGAGTAC[GAT]GCGCGA' SEEQ_TS_TAU=5 SEEQ_TS_LEN=250 python profiles/time_scan.py cfg5_best 100000000 5 best | tee -a $O/ab.txt
SEEQ_NO_VERIFY=1 SEEQ_TS_PATTERN='GATG[TA]AGCNCGATTAGC[CG]TGAAAATGNGAGTAC[GAT]GCGCGA' SEEQ_TS_TAU=5 SEEQ_TS_LEN=250 python profiles/time_scan.py cfg5_best_exact1 100000000 5 best | tee -a $O/ab.txt
done
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-e2e --no-per-call > $O/bench_best.json 2> $O/bench_best.err; python3 -c "
import json; d=json.load(open('$O/bench_best.json')); print('bench best', d['roofline']['kernel'], round(d['ms_per_step'],3), d['device_ms_per_step'], d['results']['oracle_check'])"
