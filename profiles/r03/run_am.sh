#!/bin/bash
set -u
O=gpurun_out/r03am; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_packed.py -m gpu -x -q 2>&1 | tail -3
timeout -k 10 600 python bench.py --steps 10 --warmup 3 --no-per-call --no-cli --no-cpu-baseline > $O/bench_best.json 2> $O/bench_best.err; echo "bench exit $?"; tail -3 $O/bench_best.err
python3 -c "
import json; d=json.load(open('$O/bench_best.json'))
print(round(d['value']/1e9,2),'G lines/s', round(d['ms_per_step'],3),'ms', d['device_ms_per_step'], d['roofline']['avg_launch_ms'])
print('packed', json.dumps(d.get('packed_scan'), indent=1)); print('e2e packed', d.get('end_to_end_pinned_host_packed'))"
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
for f in glob.glob("gpurun_out/r03am/trace/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:14]:
        print("%-56s calls %5s avg_us %10.2f total_ms %8.2f" % (r["Name"][:56], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
