#!/bin/bash
# Round 4, evidence of the committed build: the four bench lines, rocprofv3 kernel stats + PMC passes of the headline workload and
# of cfg5, the PMC traffic file bench.py ties to the build, the published sweep against the reference binary, the box probe.
# Usage: bash profiles/r04_final_evidence.sh [bench|profile|pmc|sweep] (a gpurun call is at most 20 minutes: one part per call).
# Outputs land under gpurun_out/r04final/ and gpurun_out/prof_r04_final_*; the summaries are copied into profiles/ afterwards.
set -u
PART=${1:-bench}
O=gpurun_out/r04final; mkdir -p $O
export TMPDIR=/tmp
if [ $PART = bench ]; then
bash profiles/r04/box_probe.sh
timeout -k 10 400 python bench.py > $O/bench_best.json 2> $O/bench_best.err; echo "best exit $?"
timeout -k 10 200 python bench.py --workload count --no-cpu-baseline --no-e2e --no-per-call --no-cli --no-packed --no-multi --no-fastq > $O/bench_count.json 2> $O/bench_count.err; echo "count exit $?"
timeout -k 10 200 python bench.py --workload all --no-cpu-baseline --no-e2e --no-per-call --no-cli --no-packed --no-multi --no-fastq > $O/bench_all.json 2> $O/bench_all.err; echo "all exit $?"
timeout -k 10 400 python bench.py --workload cfg5 --steps 20 --warmup 3 --no-e2e --no-per-call --no-cli --no-packed --no-multi --no-fastq > $O/bench_cfg5.json 2> $O/bench_cfg5.err; echo "cfg5 exit $?"
for f in $O/bench_*.json; do python3 - "$f" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
chk=d["results"].get("oracle_check") or {}
print(sys.argv[1], round(d["value"]/1e9,3), "G lines/s", round(d["ms_per_step"],3), "ms", d["device_ms_per_step"], d["roofline"]["kernel"], round(d["roofline"]["avg_launch_ms"],4), round(d["roofline"]["frac"],3), round(d["roofline"]["whole_step_frac"],3), chk.get("result"), chk.get("reference_lines_checked"), (d.get("cpu_baseline") or {}).get("value"), d["per_step"]["scan_kernel_core_clock_mhz"])
PY
done
timeout -k 10 600 python3 profiles/multi_bench.py 10000000 > $O/multi.jsonl 2> $O/multi.err; echo "multi exit $?"
fi
if [ $PART = profile ]; then
# (profiled with four placement candidates instead of the default eight: bench.py scans each 2 x 2 times before the warm-up, 16 + 1 + 3 = 20 scans of the text per run)
TIMED_SCAN_DISPATCHES=12 TEXT_BYTES_TOTAL=302000000000 timeout -k 10 500 bash profiles/gpu_profile.sh r04_final_best --placement-candidates 4 --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-per-call --no-cli --no-packed --no-multi --no-fastq --check-lines 0 > $O/profile_best.log 2>&1; echo "profile best exit $?"
TIMED_SCAN_DISPATCHES=21 TEXT_BYTES_TOTAL=502000000000 timeout -k 10 500 bash profiles/gpu_profile.sh r04_final_cfg5 --placement-candidates 4 --workload cfg5 --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-per-call --no-cli --no-packed --no-multi --no-fastq --check-lines 0 > $O/profile_cfg5.log 2>&1; echo "profile cfg5 exit $?"
find gpurun_out/prof_r04_final_best gpurun_out/prof_r04_final_cfg5 -name "*.csv" -size +2M -delete
head -40 gpurun_out/prof_r04_final_best/summary.txt
fi
if [ $PART = pmc ]; then
timeout -k 10 500 bash profiles/pmc_traffic.sh r04final_pmc > $O/pmc_traffic.log 2>&1; echo "pmc traffic exit $?"; tail -30 $O/pmc_traffic.log
fi
if [ $PART = sweep ]; then
timeout -k 10 1100 python3 profiles/chrom_sweep.py > $O/chrom_sweep.jsonl 2> $O/chrom_sweep.txt; echo "sweep exit $?"; tail -30 $O/chrom_sweep.txt
fi
