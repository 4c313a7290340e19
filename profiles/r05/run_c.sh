#!/bin/bash
# round 5, call C: the bounded VMM placement experiment (profiles/r05/vmm_probe.hip): ten processes, then other chunk sizes
set -o pipefail
out=gpurun_out/r05_c; mkdir -p $out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -I include profiles/r05/vmm_probe.hip -L seeq_amd/lib -lseeq_amd -Wl,-rpath,$PWD/seeq_amd/lib -o /tmp/vmm_probe || exit 1
for i in 1 2 3 4 5 6 7 8 9 10; do timeout -k 10 120 /tmp/vmm_probe 1024 >> $out/vmm_probe.jsonl 2>>$out/vmm_probe.err || { tail -3 $out/vmm_probe.err; exit 1; }; done
for c in 2 64 4096; do for i in 1 2 3; do timeout -k 10 120 /tmp/vmm_probe $c >> $out/vmm_probe_chunks.jsonl 2>>$out/vmm_probe.err || { tail -3 $out/vmm_probe.err; exit 1; }; done; done
python - <<'PY'
import json
for f in ("vmm_probe.jsonl", "vmm_probe_chunks.jsonl"):
    for l in open("gpurun_out/r05_c/" + f):
        d = json.loads(l)
        print(d["chunk_mib"], "plain %.3f" % d["plain_hipMalloc"]["forward_ms"], "one %.3f" % d["vmm_one_allocation"]["forward_ms"], "chunks %.3f" % d["vmm_chunks"]["forward_ms"], d["granularity_recommended"])
PY
