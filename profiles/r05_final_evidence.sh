#!/bin/bash
# Round 5, evidence of the committed build: the default bench line (headline + FASTQ shape + configs[4] section) and the other workloads' lines,
# rocprofv3 kernel stats + PMC passes of the headline workload and of cfg5, the PMC traffic file bench.py ties to the build, the published sweep
# against the reference binary.  Usage: bash profiles/r05_final_evidence.sh [suite|bench|profile|fastq|generic|pmc|sweep] (a gpurun call is at most 20 minutes).
# Outputs land under gpurun_out/r05final/ and gpurun_out/prof_r05_final_*; the summaries are copied into profiles/ afterwards.
set -u
PART=${1:-bench}
O=gpurun_out/r05final; mkdir -p $O
export TMPDIR=/tmp
if [ $PART = suite ]; then
s=$(date +%s)
timeout -k 10 1100 python -m pytest tests/ -x -q -m gpu --durations=8 > $O/pytest_gpu.log 2>&1; echo "suite exit $? in $(( $(date +%s) - s )) s"; tail -14 $O/pytest_gpu.log
fi
if [ $PART = bench ]; then
s=$(date +%s)
timeout -k 10 600 python bench.py > $O/bench_best.json 2> $O/bench_best.err; echo "best (default line) exit $? in $(( $(date +%s) - s )) s, $(wc -c < $O/bench_best.json) bytes"
timeout -k 10 300 python bench.py --workload count --sections none > $O/bench_count.json 2> $O/bench_count.err; echo "count exit $?"
timeout -k 10 300 python bench.py --workload all --sections none > $O/bench_all.json 2> $O/bench_all.err; echo "all exit $?"
timeout -k 10 400 python bench.py --workload cfg5 --steps 20 --warmup 3 --sections none > $O/bench_cfg5.json 2> $O/bench_cfg5.err; echo "cfg5 exit $?"
for f in $O/bench_*.json; do python3 - "$f" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
chk=d["results"].get("oracle_check") or {}
fa=d.get("first_allocation") or {}
print(sys.argv[1], round(d["value"]/1e9,3), "G lines/s", round(d["ms_per_step"],3), "ms", d["device_ms_per_step"], d["roofline"]["kernel"][:6], round(d["roofline"]["avg_launch_ms"],4), round(d["roofline"]["frac"],3), round(d["roofline"]["whole_step_frac"],3), chk.get("result"), chk.get("reference_lines_checked"), "first_allocation", round(fa.get("value",0)/1e9,3), fa.get("scan_launch_ms"), "probe", d["placement"]["probe_forward_ms"])
if "fastq_shape" in d and "modes" in d["fastq_shape"]:
    print("   fastq", {m: (round(v["gb_per_s"]), v["kernel"], round(v["ms_per_step"],3), v.get("identical_to_reference_count")) for m, v in d["fastq_shape"]["modes"].items()})
if "cfg5" in d and "value" in d["cfg5"]:
    c=d["cfg5"]; print("   cfg5 section", round(c["value"]/1e9,3), round(c["ms_per_step"],3), c["device_ms_per_step"], round(c["whole_step_frac"],3), (c["results"]["oracle_check"] or {}).get("reference_lines_checked"))
if "regions" in d: print("   regions", {k: (round(v.get("gpu_over_cpu",0),1), round(v.get("over_whole_socket_estimate",0),1)) for k, v in d["regions"].items() if k != "cpu"})
PY
done
fi
if [ $PART = profile ]; then
TIMED_SCAN_DISPATCHES=12 TEXT_BYTES_TOTAL=181200000000 timeout -k 10 500 bash profiles/gpu_profile.sh r05_final_best --placement-candidates 4 --first-steps 0 --steps 3 --warmup 1 --sections none --check-lines 0 > $O/profile_best.log 2>&1; echo "profile best exit $?"
TIMED_SCAN_DISPATCHES=21 TEXT_BYTES_TOTAL=301200000000 timeout -k 10 500 bash profiles/gpu_profile.sh r05_final_cfg5 --placement-candidates 4 --first-steps 0 --workload cfg5 --steps 3 --warmup 1 --sections none --check-lines 0 > $O/profile_cfg5.log 2>&1; echo "profile cfg5 exit $?"
find gpurun_out/prof_r05_final_best gpurun_out/prof_r05_final_cfg5 -name "*.csv" -size +2M -delete
head -50 gpurun_out/prof_r05_final_best/summary.txt
fi
if [ $PART = pmc ]; then
timeout -k 10 500 bash profiles/pmc_traffic.sh r05final_pmc > $O/pmc_traffic.log 2>&1; echo "pmc traffic exit $?"; tail -30 $O/pmc_traffic.log
fi
if [ $PART = sweep ]; then
timeout -k 10 1100 python3 profiles/chrom_sweep.py > $O/chrom_sweep.jsonl 2> $O/chrom_sweep.txt; echo "sweep exit $?"; tail -30 $O/chrom_sweep.txt
fi
if [ $PART = fastq ]; then
# rocprofv3 kernel stats of the FASTQ shape (25 M records = 100 M lines, 7.9 GB) under the three non-DNA modes
REPO=$PWD; cd /tmp
for m in fail convert ignore; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$O/fastq_$m -- python3 $REPO/profiles/fastq_shape_bench.py 25000000 best fastq $m > $REPO/$O/fastq_$m.log 2>&1; echo "fastq $m exit $?"
done
cd $REPO
python3 profiles/fastq_kernel_stats.py $O
find $O -name "*.csv" -size +1M -delete
fi
if [ $PART = generic ]; then
# the generic path (patterns of 63 .. 512 positions: newline index + k_forward<W> + k_exact<W>), whose speed no other evidence states
rp() { python3 -c "import random; random.seed($1); print(''.join(random.choice('ACGT') for _ in range($2)))"; }
{
SEEQ_TS_PATTERN=$(rp 5 100) SEEQ_TS_TAU=5 SEEQ_TS_LEN=150 timeout -k 10 200 python3 profiles/time_scan.py "100-mer,d=5,150bp" 20000000 5 best
SEEQ_TS_PATTERN=$(rp 6 300) SEEQ_TS_TAU=10 SEEQ_TS_LEN=500 timeout -k 10 200 python3 profiles/time_scan.py "300-mer,d=10,500bp" 5000000 5 best
SEEQ_TS_PATTERN=$(rp 7 512) SEEQ_TS_TAU=20 SEEQ_TS_LEN=1000 timeout -k 10 200 python3 profiles/time_scan.py "512-mer,d=20,1000bp" 2000000 5 best
} > $O/generic_path.txt 2>&1; echo "generic exit $?"; cat $O/generic_path.txt
fi
