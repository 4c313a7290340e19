#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats + PMC passes of bench.py.
# Usage: profiles/gpu_profile.sh <tag> [bench args...]   -> gpurun_out/prof_<tag>/
set -u
TAG=${1:-run}; shift || true
ARGS=${@:---reads 20000000 --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --check-lines 0}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
# pass 1: per-kernel time
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$REPO/bench.py" $ARGS > "$OUT/trace.log" 2>&1
# pass 2..: counters (each in its own run, kernel-trace only)
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --output-format csv -d "$OUT/pmc_sq" -- python3 "$REPO/bench.py" $ARGS > "$OUT/pmc_sq.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 "$REPO/bench.py" $ARGS > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/pmc_write" -- python3 "$REPO/bench.py" $ARGS > "$OUT/pmc_write.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_sq2" -- python3 "$REPO/bench.py" $ARGS > "$OUT/pmc_sq2.log" 2>&1
cd "$REPO"
python3 profiles/summarize_prof.py "$OUT" ${TEXT_BYTES_TOTAL:-} > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
# keep only small files
find "$OUT" -name "*.csv" -size +8M -delete
