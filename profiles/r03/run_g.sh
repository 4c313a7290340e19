#!/bin/bash
# Round 3, GPU call g: k_pair requests the next tile (8 loads back to back + halo word) right after the walk.
set -u
O=gpurun_out/r03g; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 420 python -m pytest tests -m gpu -x -q -k "(batch_scan_vs_oracle and pair) or (edge_buffers and pair) or (stream_fuzz_patterns and pair) or chunk_and_tile" > $O/pytest_pair.log 2>&1; echo "pytest exit $?" >> $O/pytest_pair.log
tail -5 $O/pytest_pair.log
B="--steps 10 --warmup 3 --no-cpu-baseline --no-e2e --no-per-call"
summ() { python3 -c "
import json,sys; d=json.load(open('$1')); print('$2', d['roofline']['kernel'], round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['device_ms_per_step'].items()}, 'launch', round(d['roofline']['avg_launch_ms'],4), 'frac', round(d['roofline']['frac'],3), d['results']['matching_lines'], d['results']['oracle_check'] and d['results']['oracle_check']['result'])"; }
timeout -k 10 300 python bench.py $B > $O/bench_best_pair.json 2> $O/bench_best_pair.err && summ $O/bench_best_pair.json best_pair || tail -5 $O/bench_best_pair.err

SEEQ_FUSED_KERNEL=stream timeout -k 10 300 python bench.py $B --check-lines 0 > $O/bench_best_stream.json 2> $O/bench_best_stream.err && summ $O/bench_best_stream.json best_stream
timeout -k 10 300 python bench.py $B --workload cfg5 --check-lines 0 > $O/bench_cfg5_pair.json 2> $O/bench_cfg5_pair.err && summ $O/bench_cfg5_pair.json cfg5_pair || tail -5 $O/bench_cfg5_pair.err

timeout -k 10 300 python bench.py $B --workload count --check-lines 0 > $O/bench_count_pair.json 2> $O/bench_count_pair.err && summ $O/bench_count_pair.json count_pair
timeout -k 10 300 python bench.py $B --workload all --check-lines 0 > $O/bench_all_pair.json 2> $O/bench_all_pair.err && summ $O/bench_all_pair.json all_pair
