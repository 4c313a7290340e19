set -u
O=gpurun_out/r02e; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log; [ $rc = 0 ] || exit 1
timeout -k 10 200 python bench.py --no-cpu-baseline --no-e2e --no-per-call > $O/bench_best.json 2> $O/bench_best.err; echo "best exit $?"
timeout -k 10 200 python bench.py --workload all --no-cpu-baseline --no-e2e --no-per-call > $O/bench_all.json 2> $O/bench_all.err; echo "all exit $?"
timeout -k 10 400 python bench.py --workload cfg5 --steps 20 --warmup 3 --no-cpu-baseline --no-e2e --no-per-call > $O/bench_cfg5.json 2> $O/bench_cfg5.err; echo "cfg5 exit $?"
for f in $O/bench_*.json; do python3 - "$f" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[1], round(d["value"]/1e9,3), "G lines/s", round(d["ms_per_step"],3), "ms", d["device_ms_per_step"], d["results"].get("oracle_check",{}).get("result"))
PY
done
