#!/bin/bash
# Round 2: the GPU suite under forced variants (small segments everywhere, 64-byte chunks, one walk per lane, no filters,
# per-line kernels only): every parity test again on the paths the defaults do not take.
set -u
O=gpurun_out/r02sweep; mkdir -p $O
K="not cli_golden and not two_real_segments and not bench_launches and not last_kernel"
run() { local tag=$1; shift; env "$@" timeout -k 10 400 python -m pytest tests -m gpu -q -k "$K" -p no:cacheprovider > $O/$tag.log 2>&1; echo "$tag: exit $? $(tail -1 $O/$tag.log)"; }
run seg64k SEEQ_SEGMENT_BYTES=65536
run seg1m SEEQ_SEGMENT_BYTES=1048576
run ch64 SEEQ_STREAM_CH=64
run ilp1 SEEQ_STREAM_ILP=1
run nofilter SEEQ_NO_FILTER=1
run nosub SEEQ_STREAM_SUB=0
run direct SEEQ_FUSED_KERNEL=direct
run generic SEEQ_PATH=generic
