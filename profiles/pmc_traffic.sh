#!/bin/bash
# HBM traffic of the scan kernels (k_pair, k_stream) from the PMC counters, as MI355X_MICROARCH.md prescribes: FETCH_SIZE and
# WRITE_SIZE in separate rocprofv3 --pmc passes (KiB units; gfx950 counts half of a wide streaming read: FETCH_SIZE x 2).
# Writes gpurun_out/<tag>/pmc_scan_kernels.json with the content hash of the kernel sources the figures belong to (bench.py
# only reports `traffic` when the hash matches the build it runs).  Usage (GPU box): bash profiles/pmc_traffic.sh <tag>
set -u
TAG=${1:-pmc}
O=$PWD/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
REPO=$PWD
B="100000000 3 best"
cd /tmp
for k in pair stream; do
  [ $k = stream ] && export SEEQ_FUSED_KERNEL=stream
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/${k}_$c -- python3 $REPO/profiles/time_scan.py pmc $B > $O/${k}_$c.log 2>&1
  done
done
cd $REPO
python3 - "$O" <<'PY'
import csv, glob, json, os, sys
sys.path.insert(0, os.getcwd())
import bench
O = sys.argv[1]
out = {}
text_per_launch = 100_000_000 * 151 / 4.0
for k, name in (("pair", "k_pair"), ("stream", "k_stream")):
    vals = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        v = []
        for f in glob.glob("%s/%s_%s/**/*counter_collection.csv" % (O, k, c), recursive=True):
            for r in csv.DictReader(open(f)):
                if (name + "<") in r["Kernel_Name"] and r["Counter_Name"] == c:
                    v.append(float(r["Counter_Value"]))
        vals[c] = v
    if not vals["FETCH_SIZE"]:
        continue
    f = sum(vals["FETCH_SIZE"]) / len(vals["FETCH_SIZE"])
    w = sum(vals["WRITE_SIZE"]) / max(1, len(vals["WRITE_SIZE"]))
    hbm = (2.0 * f + w) * 1024.0
    out[name] = {"kernel": name, "dispatches": len(vals["FETCH_SIZE"]), "fetch_size_kib_mean": f, "write_size_kib_mean": w,
                 "hbm_bytes_per_launch_mean": hbm, "correction": "2 x FETCH_SIZE (gfx950 wide-read under-count) + WRITE_SIZE, KiB -> bytes",
                 "hbm_bytes_per_text_byte": hbm / text_per_launch, "source_hash": bench.source_hash(),
                 "source": "profiles/pmc_traffic.sh: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes), 100 M x 150 bp reads, --best"}
json.dump(out, open(os.path.join(O, "pmc_scan_kernels.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
PY
find $O -name "*.csv" -size +2M -delete
