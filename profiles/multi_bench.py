"""Sixteen barcodes over 10 M reads: one walk for all of them (seeq_multi.h) against a scan per pattern (SEEQ_MULTI=sequential).
Usage (GPU box): python3 profiles/multi_bench.py [reads] -> one JSON line per barcode set."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from seeq_amd import device as dev

def make_reads(n, L, barcodes, planted, seed):
    """n reads of L random bases + newline; a fraction `planted` starts with one of the barcodes (a third of those with one substitution)."""
    rng = np.random.default_rng(seed)
    out = torch.empty(n * (L + 1), dtype=torch.uint8, device="cuda:0")
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    bl = max(len(b) for b in barcodes)
    bc = np.zeros((len(barcodes), bl), dtype=np.uint8)
    blen = np.array([len(b) for b in barcodes])
    for k, b in enumerate(barcodes):
        bc[k, :len(b)] = np.frombuffer(b.encode(), dtype=np.uint8)
    step = 1_000_000
    for f in range(0, n, step):
        c = min(step, n - f)
        a = lut[rng.integers(0, 4, size=(c, L + 1), dtype=np.uint8)]
        a[:, L] = 10
        pl = np.nonzero(rng.random(c) < planted)[0]
        k = rng.integers(0, len(barcodes), size=pl.size)
        for j in range(bl):
            m = j < blen[k]
            a[pl[m], j] = bc[k[m], j]
        sub = pl[rng.random(pl.size) < 0.33]
        a[sub, rng.integers(0, 8, size=sub.size)] = lut[rng.integers(0, 4, size=sub.size)]
        out[f * (L + 1):(f + c) * (L + 1)] = torch.from_numpy(a.reshape(-1)).cuda()
    return out

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    L = 150
    torch.cuda.set_device(0)
    rng = np.random.default_rng(3)
    for name, blen, tau, planted in (("16 x 8 bp, d 1", 8, 1, 0.9), ("16 x 10 bp, d 1", 10, 1, 0.9), ("16 x 12 bp, d 1", 12, 1, 0.9),
                                     ("16 x 10 bp, d 1, no barcode planted", 10, 1, 0.0), ("16 x 8 bp, d 0", 8, 0, 0.9)):
        barcodes = ["".join("ACGT"[i] for i in rng.integers(0, 4, size=blen)) for _ in range(16)]
        text = make_reads(n, L, barcodes, planted, 17)
        pats = [dev.Pattern(b, tau) for b in barcodes]
        sc = dev.Scanner(torch.cuda.current_stream().cuda_stream)
        res = {}
        for mode in ("one_pass", "sequential"):
            if mode == "sequential": os.environ["SEEQ_MULTI"] = "sequential"
            else: os.environ.pop("SEEQ_MULTI", None)
            for want_name, opt, want in (("best_records", dev.SQ_BEST, dev.WANT_RECORDS), ("count_lines", 0, dev.WANT_COUNTLINES)):
                best = None
                for it in range(4):
                    torch.cuda.synchronize(); t0 = time.perf_counter()
                    got = sc.scan_tensor_multi(pats, text, opt, want, copy=False)
                    dt = time.perf_counter() - t0
                    if it and (best is None or dt < best): best = dt
                res[(mode, want_name)] = (best, sc.last_multi_one_pass(), [g["nmatchlines"] for g in got], sum(g["nrecords"] for g in got),
                                          [g["records"].copy() for g in got] if want == dev.WANT_RECORDS else None)
        os.environ.pop("SEEQ_MULTI", None)
        row = {"set": name, "reads": n, "read_len": L}
        for want_name in ("best_records", "count_lines"):
            o, s = res[("one_pass", want_name)], res[("sequential", want_name)]
            assert o[1] and not s[1], (o[1], s[1])
            assert o[2] == s[2] and o[3] == s[3], want_name
            if o[4] is not None:
                assert all(np.array_equal(x, y) for x, y in zip(o[4], s[4]))
            row[want_name] = {"one_pass_ms": round(o[0] * 1e3, 3), "sequential_ms": round(s[0] * 1e3, 3), "speedup": round(s[0] / o[0], 2),
                              "matching_line_pattern_pairs": int(sum(o[2])), "records": int(o[3]),
                              "one_pass_lines_per_s": round(n / o[0]), "identical_results": True}
        print(json.dumps(row), flush=True)
        sc.close()
        for p in pats: p.close()
        del text

if __name__ == "__main__":
    main()
