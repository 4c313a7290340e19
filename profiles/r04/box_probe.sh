#!/bin/bash
# One line per GPU box: what a plain read sweep reaches on it (profiles/microbench/hbm_read) beside the scan kernel's launch time
# on the same box, same call -- the scan kernel comes in two speeds box by box (0.72-0.78 / 0.82-0.91 ms per 3.75 GiB launch).
mkdir -p gpurun_out
SW=$(./profiles/microbench/hbm_read 2>/dev/null | grep "128-B chunk per lane" | tail -3 | awk '{print $(NF-1)}' | tr '\n' ' ')
CO=$(./profiles/microbench/hbm_read 2>/dev/null | grep "coalesced" | tail -5 | awk '{print $(NF-1)}' | sort -n | tail -1)
( for i in 1 2 3 4 5 6 7 8 9 10 11 12; do sleep 1; /opt/rocm/bin/rocm-smi --showpower 2>/dev/null | grep -i "Power (W)" | awk -F: '{print $NF}'; done ) > /tmp/box_power.txt 2>/dev/null &
KP=$(python bench.py --no-per-call --no-packed --no-cli --no-multi --no-fastq --no-cpu-baseline --no-e2e --check sample --check-lines 0 --steps 1500 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); p=d['per_step']
print('launch_ms', p['scan_launch_ms_full_segments'], 'clock_mhz', p['scan_kernel_core_clock_mhz'], 'step_ms', d['ms_per_step'], 'post', round(d['device_ms_per_step']['compaction_exact_records'],3))")
SMI=$(/opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Power (W)" | awk -F: '{print $NF}' | tr '\n' ' ')
wait; PW=$(sort -n /tmp/box_power.txt | tail -1)
echo "box $(hostname) $(date +%H:%M:%S) | sweep 128B/lane TB/s: $SW | best coalesced TB/s: $CO | k_pair: $KP | socket W under k_pair (max of 1 s samples): $PW | idle smi: $SMI" | tee -a gpurun_out/r04_boxes.txt
