#!/bin/bash
# Round 3, GPU call e: kernel trace + PMC of the k_pair build (headline and configs[4]); repeated A/B of the scan kernels.
set -u
O=$PWD/gpurun_out/r03e; mkdir -p $O
export TMPDIR=/tmp
REPO=$PWD
B="--steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-per-call --check-lines 0"
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_best -- python3 $REPO/bench.py $B > $O/trace_best.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_cfg5 -- python3 $REPO/bench.py $B --workload cfg5 > $O/trace_cfg5.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS --output-format csv -d $O/pmc1 -- python3 $REPO/bench.py $B > $O/pmc1.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE SQ_INSTS_SMEM --output-format csv -d $O/pmc2 -- python3 $REPO/bench.py $B > $O/pmc2.log 2>&1
cd $REPO
python3 - <<'PY'
import csv, glob, os
from collections import defaultdict
O = "gpurun_out/r03e"
for k in ("best", "cfg5"):
    for f in glob.glob("%s/trace_%s/**/*kernel_stats.csv" % (O, k), recursive=True):
        for r in list(csv.DictReader(open(f)))[:22]:
            print(k, "%-56s calls %5s avg_us %10.2f pct %s" % (r["Name"][:56], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
agg = defaultdict(lambda: defaultdict(list))
for sub in ("pmc1", "pmc2"):
    for f in glob.glob("%s/%s/**/*counter_collection.csv" % (O, sub), recursive=True):
        for r in csv.DictReader(open(f)):
            kn = r["Kernel_Name"]
            if "k_pair<" in kn or "k_exact1<" in kn:
                agg[kn[:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for kn in agg:
    print(kn)
    for c in sorted(agg[kn]):
        v = agg[kn][c]
        print("     %-26s n=%-4d mean=%.5g" % (c, len(v), sum(v) / len(v)))
PY
find $O -name "*.csv" -size +4M -delete
B="--steps 10 --warmup 3 --no-cpu-baseline --no-e2e --no-per-call --check-lines 0"
for rep in 1 2 3; do
for k in pair stream; do
  python3 -c "
import json,subprocess,os,sys
env=dict(os.environ)
if '$k'=='stream': env['SEEQ_FUSED_KERNEL']='stream'
d=json.loads(subprocess.run([sys.executable,'bench.py']+'$B'.split(),env=env,capture_output=True,text=True).stdout)
print('rep$rep', d['roofline']['kernel'], 'step', round(d['ms_per_step'],3), 'launch', round(d['roofline']['avg_launch_ms'],4), {k: round(v,3) for k,v in d['device_ms_per_step'].items()})"
done; done
