#!/bin/bash
# Round 3, GPU call y: behind k_pair, a line with one candidate is scanned over that candidate's window only (A/B: SEEQ_NO_WINDOW=1).
set -u
O=gpurun_out/r03ad; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "pair or fuzz or chunk or edge or shard or config1" > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log
tail -6 $O/pytest.log
for rep in 1 2; do
for mode in best all count; do
python profiles/time_scan.py window_$mode 100000000 10 $mode | tee -a $O/ab.txt
SEEQ_NO_WINDOW=1 python profiles/time_scan.py whole_$mode 100000000 10 $mode | tee -a $O/ab.txt
done
done
SEEQ_TS_PATTERN='GATG[TA]AGCNCGATTAGC[CG]TGAAAATGNGAGTAC[GAT]GCGCGA' SEEQ_TS_TAU=5 SEEQ_TS_LEN=250 python profiles/time_scan.py cfg5_all_window 100000000 5 all | tee -a $O/ab.txt
SEEQ_NO_WINDOW=1 SEEQ_TS_PATTERN='GATG[TA]AGCNCGATTAGC[CG]TGAAAATGNGAGTAC[GAT]GCGCGA' SEEQ_TS_TAU=5 SEEQ_TS_LEN=250 python profiles/time_scan.py cfg5_all_whole 100000000 5 all | tee -a $O/ab.txt
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-e2e --no-per-call > $O/bench_best.json 2> $O/bench_best.err; python3 -c "
import json; d=json.load(open('$O/bench_best.json')); print('bench best', d['roofline']['kernel'], round(d['ms_per_step'],3), d['device_ms_per_step'], d['roofline']['avg_launch_ms'], d['results']['oracle_check'])"
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-e2e --no-per-call --workload cfg5 > $O/bench_cfg5.json 2> $O/bench_cfg5.err; python3 -c "
import json; d=json.load(open('$O/bench_cfg5.json')); print('bench cfg5', d['roofline']['kernel'], round(d['ms_per_step'],3), d['device_ms_per_step'], d['roofline']['avg_launch_ms'], d['results']['oracle_check'])"
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-e2e --no-per-call --workload all > $O/bench_all.json 2> $O/bench_all.err; python3 -c "
import json; d=json.load(open('$O/bench_all.json')); print('bench all', d['roofline']['kernel'], round(d['ms_per_step'],3), d['device_ms_per_step'], d['roofline']['avg_launch_ms'], d['results']['oracle_check'])"
