#!/bin/bash
set -u
O=gpurun_out/r03az; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_packed.py -m gpu -x -q > $O/t.log 2>&1; rc=$?; echo "tests exit $rc"; tail -5 $O/t.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 600 python bench.py --steps 10 --warmup 3 --no-per-call --no-cli --no-cpu-baseline > $O/bench_best.json 2> $O/bench_best.err; echo "bench exit $?"; tail -3 $O/bench_best.err
python3 -c "
import json; d=json.load(open('$O/bench_best.json'))
print(round(d['value']/1e9,2),'G lines/s', round(d['ms_per_step'],3),'ms', d['device_ms_per_step'], d['roofline']['avg_launch_ms'])
print('packed', json.dumps(d.get('packed_scan'), indent=1)); print('e2e packed', d.get('end_to_end_pinned_host_packed'))"
