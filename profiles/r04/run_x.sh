#!/bin/bash
# k_pair's launch time over a long run: does it change with time (a DPM state of memory / fabric ramping up)?
O=gpurun_out/r04x; mkdir -p $O
B="--no-per-call --no-packed --no-cli --no-multi --no-fastq --no-cpu-baseline --no-e2e --check sample --check-lines 0"
( for i in $(seq 1 40); do /opt/rocm/bin/rocm-smi --showclocks 2>/dev/null | grep -i "clk" | awk -F: '{print $(NF-1) $NF}' | tr '\n' ' '; echo; sleep 0.5; done ) > $O/clocks.txt 2>/dev/null &
python bench.py $B --steps 2000 --warmup 0 > $O/long.json 2> $O/long.err
wait
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r04x/long.json'))
a=d['per_step']['scan_launch_ms_all']
print('launches', len(a), 'step ms', d['ms_per_step'])
full=[x for i,x in enumerate(a) if i%4!=3]
for lo in (0,3,6,12,24,48,96,192,384,768,1500,3000,4500,5900):
    seg=full[lo:lo+6]
    if seg: print('full-segment launches %5d..: ' % lo, ' '.join('%.3f'%x for x in seg))
PY
head -3 $O/clocks.txt; sed -n 10,12p $O/clocks.txt; tail -2 $O/clocks.txt
python bench.py $B --steps 16 --warmup 3 2>/dev/null | python3 -c "
import json,sys
d=json.load(sys.stdin); print('short run:', d['ms_per_step'], d['per_step']['scan_launch_ms_all'][:8])"
