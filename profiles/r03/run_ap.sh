#!/bin/bash
set -u
O=gpurun_out/r03bb; mkdir -p $O
export TMPDIR=/tmp; REPO=$PWD; cd /tmp
cat > /tmp/mb.py <<'PY'
import os, sys, time
sys.path.insert(0, os.environ["REPO"]); sys.path.insert(0, os.path.join(os.environ["REPO"], "profiles"))
import numpy as np, torch
from seeq_amd import device as dev
import multi_bench as mb
torch.cuda.set_device(0)
rng = np.random.default_rng(3)
barcodes = ["".join("ACGT"[i] for i in rng.integers(0, 4, size=10)) for _ in range(16)]
text = mb.make_reads(10_000_000, 150, barcodes, 0.9, 17)
pats = [dev.Pattern(b, 1) for b in barcodes]
sc = dev.Scanner(torch.cuda.current_stream().cuda_stream)
want = dev.WANT_RECORDS if os.environ.get("WANT") == "rec" else dev.WANT_COUNTLINES
for it in range(3):
    t0 = time.perf_counter(); got = sc.scan_tensor_multi(pats, text, dev.SQ_BEST, want); print(time.perf_counter() - t0, sc.last_multi_one_pass())
PY
for w in cnt rec; do
WANT=$w REPO=$REPO timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$O/trace_$w -- python3 /tmp/mb.py > $REPO/$O/trace_$w.log 2>&1 || break
done
cd $REPO
python3 - <<'PY'
import csv, glob
for w in ("cnt", "rec"):
  print("==", w)
  for f in glob.glob("gpurun_out/r03bb/trace_%s/**/*kernel_stats.csv" % w, recursive=True):
    for r in list(csv.DictReader(open(f)))[:16]:
        print("%-60s calls %5s avg_us %10.2f total_ms %8.2f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
tail -3 $O/trace_cnt.log
