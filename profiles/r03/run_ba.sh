#!/bin/bash
set -u
O=gpurun_out/r03ba; mkdir -p $O
export TMPDIR=/tmp; REPO=$PWD; cd /tmp
cat > /tmp/pk.py <<'PY'
import sys, os, time
sys.path.insert(0, os.environ["REPO"])
import torch
from seeq_amd import device as dev
n, L = 100_000_000, 150
pat = dev.Pattern("GATGTAGCGCGATTAGCCTG", 3)
torch.cuda.set_device(0); stream = torch.cuda.current_stream().cuda_stream
text = torch.empty(n * (L + 1), dtype=torch.uint8, device="cuda:0")
dev.synth_reads(text.data_ptr(), 0, n, L, "GATGTAGCGCGATTAGCCTG", 3, stream=stream)
pb = torch.empty(n * 38, dtype=torch.uint8, device="cuda:0"); pn = torch.empty(n * 19, dtype=torch.uint8, device="cuda:0")
dev.pack_reads_device(text.data_ptr(), n, L, pb.data_ptr(), pn.data_ptr(), stream=stream); torch.cuda.synchronize()
sc = dev.Scanner(stream); sc.reserve(0, 0, n // 6, n // 8)
for _ in range(3):
    sc.run_packed(pat, pb.data_ptr(), pn.data_ptr(), n, L, options=dev.SQ_BEST, want=dev.WANT_RECORDS); print(sc.fetch())
PY
REPO=$REPO timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$O/trace -- python3 /tmp/pk.py > $REPO/$O/trace.log 2>&1
cd $REPO
python3 - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/r03ba/trace/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:16]:
        print("%-56s calls %5s avg_us %10.2f total_ms %8.2f" % (r["Name"][:56], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
find $O -name "*.csv" -size +2M -delete
