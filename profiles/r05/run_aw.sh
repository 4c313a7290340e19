#!/bin/bash
# round 5, call AW: k_pair without its two 4-byte stores per tile (experiment library, results void): does the (text, context) pairing go away?
export TMPDIR=/tmp
cp seeq_amd/lib/libseeq_amd.so /tmp/lib_new.so
cp profiles/r05/ab_libs/libseeq_amd_notilestores.so seeq_amd/lib/libseeq_amd.so
echo "== without the per-tile stores"; timeout -k 10 300 python3 profiles/r05/workspace_probe.py 2>&1 | grep -v amdgpu | tail -4
cp /tmp/lib_new.so seeq_amd/lib/libseeq_amd.so
echo "== as committed"; timeout -k 10 300 python3 profiles/r05/workspace_probe.py 2>&1 | grep -v amdgpu | tail -4
