#!/bin/bash
# round 5, call H: after the prune / the split of run_segments / the fault-path fixes -- the whole GPU suite with its slowest tests listed,
# then PMC passes of k_stream's Myers mode (what holds it at half its VALU bound?), one-word (m=20,k=4) and two-word (m=42,k=9) columns
set -o pipefail
out=$PWD/gpurun_out/r05_h; mkdir -p $out
s=$(date +%s)
timeout -k 10 1100 python -m pytest tests/ -x -q -m gpu --durations=30 > $out/pytest.log 2>&1 || { tail -60 $out/pytest.log; exit 1; }
tail -45 $out/pytest.log; echo "suite: $(( $(date +%s) - s )) s"
ls -la seeq_amd/lib/libseeq_amd.so
REPO=$PWD
export TMPDIR=/tmp
cd /tmp
for cell in 20:4 42:9; do
  tag=$(echo $cell | tr ':' '_')
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/myers_${tag}_trace -- python3 $REPO/profiles/chrom_sweep.py --cells $cell --no-ref > $out/myers_${tag}_trace.log 2>&1 || { tail -5 $out/myers_${tag}_trace.log; exit 1; }
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE SQ_WAVES" "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_IFETCH_LEVEL"; do
    i=$((i+1))
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/myers_${tag}_p$i -- python3 $REPO/profiles/chrom_sweep.py --cells $cell --no-ref > $out/myers_${tag}_p$i.log 2>&1 || echo "pass $i of $cell failed"
  done
done
cd $REPO
python3 - <<'PY'
import csv, glob, os
from collections import defaultdict
out = "gpurun_out/r05_h"
for tag in ("20_4", "42_9"):
    print("== k_stream Myers mode, cell", tag)
    for f in glob.glob("%s/myers_%s_trace/**/*kernel_stats.csv" % (out, tag), recursive=True):
        for r in list(csv.DictReader(open(f)))[:6]:
            print("   %-70s calls %s avg_us %.1f" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
    vals = defaultdict(list)
    for f in glob.glob("%s/myers_%s_p*/**/*counter_collection.csv" % (out, tag), recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_stream<" in r["Kernel_Name"]:
                vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in sorted(vals):
        v = vals[k]
        print("   %-28s mean %.4g  (n=%d)" % (k, sum(v) / len(v), len(v)))
PY
find $out -name "*.csv" -size +1M -delete
