#!/bin/bash
set -u
O=gpurun_out/r03bh; mkdir -p $O
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 || exit 1
for i in 1 2; do
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-e2e --no-per-call --no-cli --no-multi > $O/b_$i.json 2> $O/b_$i.err || { tail -5 $O/b_$i.err; exit 1; }
python3 -c "
import json; d=json.load(open('$O/b_$i.json')); print('best', round(d['value']/1e9,2), round(d['ms_per_step'],3), d['device_ms_per_step'], round(d['roofline']['avg_launch_ms'],4), d['results']['oracle_check']['result'], 'packed', round(d['packed_scan']['ms_per_step'],3), d['packed_scan']['identical_to_ascii_run'])"
done
SEEQ_NO_FUSE=1 timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-e2e --no-per-call --no-cli --no-multi > $O/b_nofuse.json 2> $O/b_nofuse.err
python3 -c "
import json; d=json.load(open('$O/b_nofuse.json')); print('nofuse', round(d['value']/1e9,2), round(d['ms_per_step'],3), d['device_ms_per_step'], round(d['roofline']['avg_launch_ms'],4), 'packed', round(d['packed_scan']['ms_per_step'],3))"
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "tests exit $?"; tail -5 $O/gpu_tests.log
