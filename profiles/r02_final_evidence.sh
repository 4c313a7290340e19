#!/bin/bash
# Round 2, final evidence of the committed build: the four bench lines, rocprofv3 kernel stats + PMC passes (default
# workload, cfg5), FASTQ-shaped runs, CLI wall clock.  Outputs are copied into profiles/ by hand afterwards.
set -u
O=gpurun_out/r02final; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 300 python bench.py > $O/bench_best.json 2> $O/bench_best.err; echo "best exit $?"
timeout -k 10 200 python bench.py --workload count --no-cpu-baseline --no-e2e --no-per-call > $O/bench_count.json 2> $O/bench_count.err; echo "count exit $?"
timeout -k 10 200 python bench.py --workload all --no-cpu-baseline --no-e2e --no-per-call > $O/bench_all.json 2> $O/bench_all.err; echo "all exit $?"
timeout -k 10 400 python bench.py --workload cfg5 --steps 20 --warmup 3 --no-e2e --no-per-call > $O/bench_cfg5.json 2> $O/bench_cfg5.err; echo "cfg5 exit $?"
for f in $O/bench_*.json; do python3 - "$f" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[1], round(d["value"]/1e9,3), "G lines/s", round(d["ms_per_step"],3), "ms", d["device_ms_per_step"], d["roofline"]["kernel"], round(d["roofline"]["avg_launch_ms"],4), round(d["roofline"]["frac"],3), d["results"].get("oracle_check",{}).get("result"), (d.get("cpu_baseline") or {}).get("value"))
PY
done
for x in fail convert ignore; do timeout 300 python profiles/fastq_shape_bench.py 5000000 best fastq $x > $O/fastq_$x.json 2> $O/fastq_$x.err; cat $O/fastq_$x.json; done
TEXT_BYTES_TOTAL=60400000000 timeout -k 10 400 bash profiles/gpu_profile.sh r02_final_best --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-per-call --check-lines 0 > $O/profile_best.log 2>&1; echo "profile best exit $?"
TEXT_BYTES_TOTAL=100400000000 timeout -k 10 400 bash profiles/gpu_profile.sh r02_final_cfg5 --workload cfg5 --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-per-call --check-lines 0 > $O/profile_cfg5.log 2>&1; echo "profile cfg5 exit $?"
find gpurun_out/prof_r02_final_best gpurun_out/prof_r02_final_cfg5 -name "*.csv" -size +2M -delete
head -24 gpurun_out/prof_r02_final_best/summary.txt
