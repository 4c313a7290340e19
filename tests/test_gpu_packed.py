"""GPU (-m gpu): packed read batches (include/seeq_amd.h: seeqdev_packed_t; seeq_amd/csrc/seeq_packed.h) -- 2 bits per base,
one read per lane, no warm-up -- against the oracle over the same reads as ASCII text, one read per line: every match
option, both counts, read lengths from 1 to 256 (odd lengths, lengths that are no multiple of four), N through the mask,
patterns served by a prefix automaton and by a partition filter, workspace growth (every read a candidate)."""
import random
import sys

import numpy as np
import pytest

from conftest import GOLDEN
from oracle.pyoracle import SQ_ALL, SQ_BEST, SQ_FIRST

pytestmark = pytest.mark.gpu


def _packed_scan(dev, torch, pat, text, read_len, opt, want, with_n=True):
    bases, nmask, n = dev.pack_reads(text, read_len, with_n)
    db = torch.from_numpy(bases.copy()).cuda()
    dn = torch.from_numpy(nmask.copy()).cuda() if nmask is not None else None
    sc = dev.Scanner()
    sc.run_packed(pat, db.data_ptr(), dn.data_ptr() if dn is not None else None, n, read_len, options=opt, want=want)
    cnt = sc.fetch()
    res = dict(cnt)
    if want == dev.WANT_RECORDS:
        res["records"] = sc.records(cnt["nrecords"])
    res["kernel"] = sc.last_kernel()
    sc.close()
    return res


def test_packed_vs_oracle(gpu, capi, oracle):
    import torch
    from seeq_amd import device as dev
    sys.path.insert(0, GOLDEN)
    from make_golden import mutate, plain
    rng = random.Random(31)
    cases = [("GATGTAGCGCGATTAGCCTG", 3, 150), ("GATGTAGCGCGATTAGCCTG", 3, 149), ("GATTAGC", 1, 37), ("CACAGAT", 3, 50), ("ACGT", 1, 7),
             ("AC", 0, 1), ("ACNNGT[AC]TTG", 2, 100), ("GATG[TA]AGCNCGATTAGC[CG]TGAAAATGNGAGTAC[GAT]GCGCGA", 5, 250), ("GATGTAGCGCGATTAGCCTGAAAA", 3, 256),
             ("AAAAAAAAAAAAAAAAAAAA", 3, 151), ("GATGTAGCGCGATTAG", 4, 63)]
    for pattern, tau, L in cases:
        core = plain(pattern).replace("N", "A")
        lines = []
        for i in range(3000):
            t = "".join(rng.choice("ACGT") for _ in range(L))
            if i % 3 == 0 and L >= len(core):
                cp = mutate(rng, core, rng.randint(0, tau + 2))
                q = rng.choice([0, max(0, L - len(cp)), rng.randrange(max(1, L - len(cp) + 1))])
                t = (t[:q] + cp + t[q + len(cp):])[:L]
            if i % 11 == 0:
                q = rng.randrange(L)
                t = t[:q] + "N" + t[q + 1:]
            if i % 37 == 0:
                t = t.lower()
            lines.append(t)
        text = ("\n".join(lines) + ("\n" if rng.random() < 0.5 else "")).encode()
        pat = dev.Pattern(pattern, tau)
        for mo in (SQ_FIRST, SQ_BEST, SQ_ALL):
            exp = oracle.buffer_scan(pattern, tau, text, mo)
            got = _packed_scan(dev, torch, pat, text, L, mo, dev.WANT_RECORDS)
            assert got["kernel"] == "k_packed"
            assert got["nlines"] == exp["nlines"] == len(lines) and got["nmatchlines"] == exp["nmatchlines"], (pattern, tau, L, mo)
            assert np.array_equal(got["records"].astype(np.uint64), exp["records"]), (pattern, tau, L, mo)
        expa = oracle.buffer_scan(pattern, tau, text, SQ_ALL)
        c1 = _packed_scan(dev, torch, pat, text, L, 0, dev.WANT_COUNTLINES)
        c2 = _packed_scan(dev, torch, pat, text, L, 0, dev.WANT_COUNTMATCH)
        assert c1["nmatchlines"] == expa["nmatchlines"] and c1["nlines"] == expa["nlines"], (pattern, tau, L)
        assert c2["nhits"] == len(expa["records"]) and c2["nmatchlines"] == expa["nmatchlines"], (pattern, tau, L)
        pat.close()


def test_packed_every_read_a_candidate_and_no_nmask(gpu, capi, oracle):
    """Workspace growth (the optimistic hit-list capacity is one read in eight: here every read holds the pattern) and a
    batch without an N mask."""
    import torch
    from seeq_amd import device as dev
    rng = random.Random(5)
    pattern, tau, L = "GATTAGCCTG", 1, 60
    lines = []
    for _ in range(40000):
        q = rng.randrange(L - len(pattern) + 1)
        t = "".join(rng.choice("ACGT") for _ in range(L))
        lines.append(t[:q] + pattern + t[q + len(pattern):])
    text = ("\n".join(lines) + "\n").encode()
    pat = dev.Pattern(pattern, tau)
    exp = oracle.buffer_scan(pattern, tau, text, SQ_BEST)
    got = _packed_scan(dev, torch, pat, text, L, SQ_BEST, dev.WANT_RECORDS, with_n=False)
    assert got["nmatchlines"] == exp["nmatchlines"] == len(lines)
    assert np.array_equal(got["records"].astype(np.uint64), exp["records"])
    with pytest.raises(dev.SeeqDeviceError):
        dev.pack_reads(b"ACGN\n", 4, with_nmask=False)        # an N without a mask to put it in
    with pytest.raises(dev.SeeqDeviceError):
        dev.pack_reads(b"ACGT\nACG\n", 4)                     # a line of another length
    pat.close()
