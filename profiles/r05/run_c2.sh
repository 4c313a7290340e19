#!/bin/bash
# round 5, call C2: the VMM placement probe again, one process at a time, stopping at the first one that fails or hangs (call C: the
# first process answered -- plain 0.80, one VMM allocation 0.91, 1 GiB chunks 1.05 ms per launch -- the second did not return in 120 s)
out=gpurun_out/r05_c2; mkdir -p $out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -I include profiles/r05/vmm_probe.hip -L seeq_amd/lib -lseeq_amd -Wl,-rpath,$PWD/seeq_amd/lib -o /tmp/vmm_probe || exit 1
for i in 1 2 3 4 5 6 7 8 9; do
  s=$(date +%s.%N)
  timeout -k 5 90 /tmp/vmm_probe 1024 >> $out/vmm_probe.jsonl 2>>$out/vmm_probe.err; rc=$?
  echo "run $i rc $rc seconds $(echo "$(date +%s.%N) - $s" | bc)" >> $out/runs.txt
  if [ $rc -ne 0 ]; then break; fi
done
cat $out/runs.txt
python - <<'PY'
import json
for l in open("gpurun_out/r05_c2/vmm_probe.jsonl"):
    d = json.loads(l)
    print(d["chunk_mib"], "plain %.3f" % d["plain_hipMalloc"]["forward_ms"], "one %.3f" % d["vmm_one_allocation"]["forward_ms"], "chunks %.3f" % d["vmm_chunks"]["forward_ms"])
PY
