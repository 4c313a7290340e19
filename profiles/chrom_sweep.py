#!/usr/bin/env python3
"""The reference's published parameter sweep (doc/response.tex:181-232: prefixes of 20 / 27 / 34 / 42 positions of
GATGTAGCGCGATTAGCCTGAAAATGCGAGTACGGCGCGAAT, k = 3 .. 15 errors, searched in a genome with one chromosome per line) on its
own input SHAPE: 24 lines x 128 MiB of uniform random DNA (3.2 GB) with planted approximate copies of the 42-mer.
Per cell (m, k), `--all` with positions (the CLI's -a -f): which scan kernel served it, device-resident milliseconds,
records -- compared, all of them, with the output of the reference binary (oracle/_ref/seeq_ref -a -f) on the same
bytes -- and the reference's seconds on one core.  One JSON line per cell on stdout, a table on stderr.

Usage: python profiles/chrom_sweep.py [--lines N] [--mib M] [--cells m:k,m:k,...] [--no-ref] [--jobs J]"""
import argparse
import json
import os
import subprocess
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

FULL = "GATGTAGCGCGATTAGCCTGAAAATGCGAGTACGGCGCGAAT"
CELLS = [(20, k) for k in (3, 4, 5)] + [(27, k) for k in range(3, 9)] + [(34, k) for k in range(5, 11)] + [(42, k) for k in range(8, 16)]


def make_text(nlines, L, device):
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(2025)
    lut = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device=device)
    text = torch.empty(nlines * L, dtype=torch.uint8, device=device)
    for i in range(nlines):                                # line by line: bounded temporaries
        text[i * L:(i + 1) * L] = lut[torch.randint(0, 4, (L,), device=device, generator=g, dtype=torch.uint8).long()]
        text[(i + 1) * L - 1] = 10
    rng = np.random.default_rng(7)
    pat = np.frombuffer(FULL.encode(), dtype=np.uint8)
    for _ in range(64):                                    # planted copies of the 42-mer with 0 .. 12 substitutions
        p = int(rng.integers(100, nlines * L - 100))
        c = pat.copy()
        for _e in range(int(rng.integers(0, 13))):
            c[int(rng.integers(0, len(c)))] = b"ACGT"[int(rng.integers(0, 4))]
        if (p % L) < L - 60:
            text[p:p + len(c)] = torch.from_numpy(c).to(device)
    torch.cuda.synchronize()
    return text


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lines", type=int, default=24)
    ap.add_argument("--mib", type=int, default=128)
    ap.add_argument("--cells", default="")
    ap.add_argument("--no-ref", action="store_true")
    ap.add_argument("--jobs", type=int, default=max(1, min(16, (os.cpu_count() or 2) - 1)))
    args = ap.parse_args()
    import torch
    from seeq_amd import device as dev
    from oracle.pyoracle import REF_BIN
    cells = [tuple(int(x) for x in c.split(":")) for c in args.cells.split(",") if c] or CELLS
    L = args.mib << 20
    device = torch.device("cuda:0")
    text = make_text(args.lines, L, device)
    nbytes = int(text.numel())
    ref_ok = (not args.no_ref) and os.path.exists(REF_BIN)
    path = "/dev/shm/seeq_chrom_%d.txt" % os.getpid()
    pool, futs = None, {}
    if ref_ok:
        text.cpu().numpy().tofile(path)

        def run_ref(m, k):
            t0 = time.perf_counter()
            r = subprocess.run([REF_BIN, "-d", str(k), "-a", "-f", FULL[:m], path], capture_output=True, text=True)
            return time.perf_counter() - t0, r.stdout

        pool = ThreadPoolExecutor(args.jobs)
        futs = {c: pool.submit(run_ref, *c) for c in cells}
    rows = []
    try:
        for m, k in cells:
            P = dev.Pattern(FULL[:m], k)
            sc = dev.Scanner()
            sc.set_profiling(True)
            best = None
            for _ in range(4):                             # (the first run sizes the workspace and builds the automata; best of the rest: the clock ramps up over the first scans of a process)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                cnt = sc.scan_tensor(P, text, dev.SQ_ALL, dev.WANT_RECORDS)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            rec = sc.records(cnt["nrecords"])
            row = {"m": m, "k": k, "kernel": sc.last_kernel(), "filter": sc.last_filter(), "gpu_ms": best * 1e3, "gb_per_s": nbytes / best / 1e9,
                   "records": int(cnt["nrecords"]), "matching_lines": int(cnt["nmatchlines"]), "times_ms": sc.last_times_ms()}
            if ref_ok:
                secs, out = futs[(m, k)].result()
                exp = []
                for ln in out.splitlines():                # "line:start-end:dist", end inclusive (seeq.c:150-160)
                    a, b, c = ln.split(":")
                    s_, e_ = b.split("-")
                    exp.append((int(a), int(s_), int(e_) + 1, int(c)))
                got = [tuple(int(x) for x in r) for r in rec.tolist()]
                row.update(reference_seconds_one_core=secs, reference_records=len(exp), records_identical=got == exp)
            rows.append(row)
            print(json.dumps(row), flush=True)
            sc.close()
            P.close()
    finally:
        if pool:
            pool.shutdown(wait=True)
        if ref_ok and os.path.exists(path):
            os.remove(path)
    sys.stderr.write("%3s %3s  %-9s %-6s %10s %9s %8s %10s %s\n" % ("m", "k", "kernel", "filter", "GPU ms", "GB/s", "records", "ref s", "identical"))
    for r in rows:
        sys.stderr.write("%3d %3d  %-9s %-6s %10.2f %9.1f %8d %10s %s\n" % (r["m"], r["k"], r["kernel"], r["filter"], r["gpu_ms"], r["gb_per_s"], r["records"],
                         ("%.1f" % r["reference_seconds_one_core"]) if "reference_seconds_one_core" in r else "-", r.get("records_identical", "-")))


if __name__ == "__main__":
    main()
