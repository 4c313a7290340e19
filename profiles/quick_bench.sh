#!/bin/bash
# Quick GPU-side A/B: prints one line per bench invocation.  Usage: bash profiles/quick_bench.sh "<label>" <bench args...>
label=$1; shift
python bench.py --no-cpu-baseline --no-e2e "$@" 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
print('$label', '%.2f Glines/s %.0f GB/s %.2f ms/step' % (d['value']/1e9, d['gb_per_s'], d['ms_per_step']), {k: round(v,3) for k,v in d['device_ms_per_step'].items()}, 'frac=%.3f' % d['roofline']['frac'], d['results']['matching_lines'], (d['results'].get('oracle_check') or {}).get('result'), d['results'].get('oracle_lines_checked'))"
