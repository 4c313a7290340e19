#!/bin/bash
# round 5, call P: A/B of the fast alphabet check's column index (old: (w >> 1) & 7, 6.5 VALU per word; new: w & 7, 5.5) -- two libraries, alternating
# processes, per process the scan-kernel time on the fastest of eight candidate buffers and on the plain first allocation
out=$PWD/gpurun_out/r05_p; mkdir -p $out
export TMPDIR=/tmp
cp seeq_amd/lib/libseeq_amd.so /tmp/lib_new.so
for it in 1 2 3; do
  for v in old new; do
    if [ $v = old ]; then cp profiles/r05/ab_libs/libseeq_amd_oldchk.so seeq_amd/lib/libseeq_amd.so; else cp /tmp/lib_new.so seeq_amd/lib/libseeq_amd.so; fi
    timeout -k 10 200 python bench.py --sections none --check-lines 0 --placement-candidates 8 --first-steps 10 --steps 10 --warmup 2 > $out/b_${v}_$it.json 2> $out/b_${v}_$it.err || { echo "bench failed"; exit 1; }
    python3 - $out/b_${v}_$it.json $v $it <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[2], sys.argv[3], "chosen launch", round(d["roofline"]["avg_launch_ms"],4), "step", round(d["ms_per_step"],3), "first alloc launch", d["first_allocation"]["scan_launch_ms"], "probes", d["placement"]["probe_forward_ms"])
PY
  done
done
cp /tmp/lib_new.so seeq_amd/lib/libseeq_amd.so
