#!/bin/bash
# k_emit_all (SQ_ALL records behind k_verify): parity tests, then A/B against k_exact1's EMIT pass on the --all and cfg5 workloads.
set -o pipefail
O=gpurun_out/r04s; mkdir -p $O
python -m pytest tests/test_gpu_randomized.py tests/test_gpu_packed.py tests/test_gpu_parity.py -m gpu -x -q > $O/pytest.log 2>&1; rc=$?
tail -4 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
B="--no-cpu-baseline --no-e2e --no-per-call --no-cli --no-packed --no-multi --no-fastq"
for v in new old; do
  SEEQ_EMIT_ALL=$v timeout -k 10 200 python bench.py --workload all $B > $O/all_$v.json 2> $O/all_$v.err || exit 1
  SEEQ_EMIT_ALL=$v timeout -k 10 300 python bench.py --workload cfg5 --steps 20 --warmup 3 $B > $O/cfg5_$v.json 2> $O/cfg5_$v.err || exit 1
done
python3 - <<'PY'
import json
for w in ("all","cfg5"):
    for v in ("new","old"):
        d=json.load(open(f"gpurun_out/r04s/{w}_{v}.json"))
        chk=d["results"].get("oracle_check") or {}
        print(w, v, round(d["value"]/1e9,3), "G lines/s", round(d["ms_per_step"],3), "ms", d["device_ms_per_step"], chk.get("result"), chk.get("reference_lines_checked"))
PY
