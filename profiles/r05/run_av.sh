#!/bin/bash
# round 5, call AV: the workspace arrays allocated as power-of-two blocks (experiment library) -- do the slow (text, context) pairings go away?
out=$PWD/gpurun_out/r05_av; mkdir -p $out
export TMPDIR=/tmp
cp seeq_amd/lib/libseeq_amd.so /tmp/lib_new.so
cp profiles/r05/ab_libs/libseeq_amd_ws_pow2.so seeq_amd/lib/libseeq_amd.so
echo "== workspace in power-of-two blocks"; timeout -k 10 300 python3 profiles/r05/workspace_probe.py 2>&1 | grep -v amdgpu
timeout -k 10 200 python bench.py --sections none --check-lines 0 --first-steps 10 --steps 10 --warmup 2 > $out/b_pow2.json 2> $out/b_pow2.err; python3 -c "
import json; d=json.load(open('$out/b_pow2.json')); print('value %.2f  chosen launch %.4f  first allocation %.2f launch %s  probes %s' % (d['value']/1e9, d['roofline']['avg_launch_ms'], d['first_allocation']['value']/1e9, d['first_allocation']['scan_launch_ms'], d['placement']['probe_forward_ms']))"
cp /tmp/lib_new.so seeq_amd/lib/libseeq_amd.so
echo "== as committed"; timeout -k 10 300 python3 profiles/r05/workspace_probe.py 2>&1 | grep -v amdgpu
