#!/bin/bash
# Round 3, GPU call x: the bench line with its new fields (regions, CLI wall clock, pinned CPU baseline), the chrom workload, PMC traffic.
set -u
O=gpurun_out/r03x; mkdir -p $O
nproc; free -g | head -2; df -h /dev/shm | tail -1
bash profiles/pmc_traffic.sh r03x_pmc > $O/pmc.log 2>&1; tail -22 $O/pmc.log
cp gpurun_out/r03x_pmc/pmc_scan_kernels.json profiles/pmc_scan_kernels.json
timeout -k 10 500 python bench.py --steps 20 --warmup 5 > $O/bench_best.json 2> $O/bench_best.err; echo "bench exit $?"; tail -3 $O/bench_best.err
python3 -c "
import json; d=json.load(open('$O/bench_best.json'))
print(round(d['value']/1e9,2),'G lines/s', round(d['ms_per_step'],3),'ms', d['device_ms_per_step'])
print('roofline', d['roofline']['kernel'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['traffic_source'])
print('cpu', d['cpu_baseline']); print('regions', json.dumps(d['regions'], indent=1)); print('cli', d.get('cli_wall_clock')); print('e2e', d.get('end_to_end_pinned_host')); print('check', d['results']['oracle_check'])"
timeout -k 10 300 python bench.py --workload chrom --steps 10 --warmup 2 > $O/bench_chrom.json 2> $O/bench_chrom.err; echo "chrom exit $?"; cat $O/bench_chrom.json | cut -c1-1500; tail -3 $O/bench_chrom.err
