#!/bin/bash
set -u
O=gpurun_out/r03ae; mkdir -p $O
python profiles/r03/offset_sweep.py | tee $O/offsets.txt
python profiles/r03/offset_sweep.py | tee -a $O/offsets.txt
