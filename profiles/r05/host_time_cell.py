#!/usr/bin/env python3
"""Where does the host spend a scan of a sweep cell?  run / fetch timed apart, several scans.  Usage: host_time_cell.py m k"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "profiles"))
import torch
import chrom_sweep
from seeq_amd import device as dev
m, k = int(sys.argv[1]), int(sys.argv[2])
text = chrom_sweep.make_text(24, 128 << 20, torch.device("cuda:0"))
P = dev.Pattern(chrom_sweep.FULL[:m], k)
sc = dev.Scanner(); sc.set_profiling(True)
for it in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sc.run(P, text.data_ptr(), text.numel(), dev.SQ_ALL, dev.WANT_RECORDS)
    t1 = time.perf_counter()
    cnt = sc.fetch()
    t2 = time.perf_counter()
    print(m, k, it, "run %.3f ms  fetch %.3f ms" % (1e3 * (t1 - t0), 1e3 * (t2 - t1)), sc.last_times_ms(), cnt["nrecords"], flush=True)
