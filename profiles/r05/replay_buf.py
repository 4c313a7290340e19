#!/usr/bin/env python3
"""Replays one saved fuzz buffer under a set of environments: line counts and records against the oracle."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle.pyoracle import Oracle, SQ_ALL, SQ_BEST, SQ_FIRST
from seeq_amd import device as dev
buf = open(sys.argv[1], "rb").read(); pattern = sys.argv[2]; tau = int(sys.argv[3])
o = Oracle()
for env in ({}, {"SEEQ_FUSED_KERNEL": "direct"}, {"SEEQ_SEGMENT_BYTES": "65536"}, {"SEEQ_SEGMENT_BYTES": "65536", "SEEQ_FUSED_KERNEL": "direct"},
            {"SEEQ_SEGMENT_BYTES": "32768", "SEEQ_FUSED_KERNEL": "direct"}, {"SEEQ_SEGMENT_BYTES": "65536", "SEEQ_FUSED_KERNEL": "stream"}, {"SEEQ_SEGMENT_BYTES": "65536", "SEEQ_PATH": "generic"}):
    for k in ("SEEQ_FUSED_KERNEL", "SEEQ_SEGMENT_BYTES", "SEEQ_PATH"):
        os.environ.pop(k, None)
    os.environ.update(env)
    p = dev.Pattern(pattern, tau); sc = dev.Scanner()
    for nd in (dev.SQ_IGNORE, 0, dev.SQ_CONVERT):
        for rep in range(2):
            exp = o.buffer_scan(pattern, tau, buf, SQ_BEST | nd)
            got = sc.scan_host(p, buf, SQ_BEST | nd, dev.WANT_RECORDS)
            same = np.array_equal(got["records"].astype(np.uint64), exp["records"])
            print(env, "nd", nd, "rep", rep, sc.last_kernel(), "lines", got["nlines"], exp["nlines"], "matching", got["nmatchlines"], exp["nmatchlines"], "records identical", same, flush=True)
    sc.close(); p.close()
