#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the reference itself.

Run in the build container only (needs oracle/_ref/, built from /root/reference
by `make -C oracle ref`).  Outputs are DATA: inputs plus the outputs the
reference produced for them.  Nothing of the reference's source is stored.

  ref_string_cases.json : seeqStringMatch(pattern, tau, text, options) -> sq->match[]
  ref_cli_cases.json    : reference CLI stdout for small input files
  reads_small.txt       : the shape-R / FASTQ-ish inputs those CLI cases use
"""
import hashlib
import json
import os
import random
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle.pyoracle import (Reference, Oracle, REF_BIN, SQ_FIRST, SQ_BEST, SQ_ALL, SQ_COUNT,  # noqa: E402
                             SQ_FAIL, SQ_CONVERT, SQ_IGNORE, SQ_STREAM)


def rand_pattern(rng):
    m = rng.choice([1, 2, 3, 4, 6, 8, 12, 16, 20, 20, 27, 31, 32, 33, 40, 48, 63, 64, 65, 70])
    out = []
    for _ in range(m):
        x = rng.random()
        if x < 0.05:
            out.append('N')
        elif x < 0.12:
            out.append('[' + ''.join(rng.choice('ACGTacgtUu') for _ in range(rng.randint(1, 3))) + ']')
        else:
            out.append(rng.choice('ACGT'))
    return ''.join(out)


def plain(p):
    res, i = [], 0
    while i < len(p):
        if p[i] == '[':
            j = p.index(']', i)
            res.append(p[i + 1].upper().replace('U', 'T'))
            i = j + 1
        else:
            res.append('A' if p[i] in 'Nn' else p[i].upper())
            i += 1
    return ''.join(res)


def mutate(rng, s, e):
    s = list(s)
    for _ in range(e):
        if not s:
            break
        t, p = rng.randint(0, 2), rng.randrange(len(s))
        if t == 0:
            s[p] = rng.choice('ACGT')
        elif t == 1:
            s.insert(p, rng.choice('ACGT'))
        else:
            del s[p]
    return ''.join(s)


def rand_text(rng, pat, tau, L):
    text = ''.join(rng.choice('ACGT') for _ in range(L))
    for _ in range(rng.randint(0, 3)):
        cp = mutate(rng, plain(pat), rng.randint(0, tau + 2))
        if text:
            p = rng.randrange(len(text) + 1)
            text = text[:p] + cp + text[p + len(cp):]
    tl = list(text)
    for i in range(len(tl)):
        x = rng.random()
        if x < 0.01:
            tl[i] = 'N'
        elif x < 0.015:
            tl[i] = rng.choice('RYKM-*xz@+!')
        elif x < 0.018:
            tl[i] = '\n'
        elif x < 0.03:
            tl[i] = tl[i].lower()
    return ''.join(tl)


def string_cases(ref, rng, n):
    cases = []
    for _ in range(n):
        pat = rand_pattern(rng)
        m = len(plain(pat))
        tau = rng.randint(0, min(m - 1, rng.choice([0, 1, 2, 3, 3, 5, 8])))
        text = rand_text(rng, pat, tau, rng.choice([0, 1, 5, 20, 60, 150, 250]))
        opt = rng.choice([SQ_FIRST, SQ_BEST, SQ_ALL, SQ_COUNT]) | rng.choice([SQ_FAIL, SQ_CONVERT, SQ_IGNORE]) \
            | rng.choice([0, 0, 0, SQ_STREAM])
        hits = ref.string_match(pat, tau, text, opt)
        cases.append(dict(pattern=pat, tau=tau, text=text, options=opt, hits=[list(h) for h in hits]))
    return cases


def cli(args, path):
    return subprocess.run([REF_BIN] + args + [path], capture_output=True, text=True).stdout


def cli_case(name, args, path):
    """Small outputs are stored whole; large ones as sha256 + size + head."""
    out = cli(args, path)
    if len(out) <= 20000:
        return dict(file=name, args=args, stdout=out)
    return dict(file=name, args=args, sha256=hashlib.sha256(out.encode()).hexdigest(),
                nbytes=len(out), head=out[:2000])


def main():
    ref = Reference()
    orc = Oracle()
    rng = random.Random(20251003)
    cases = string_cases(ref, rng, 1500)
    # A few deterministic edge cases around the zero-run quirk and overlaps.
    for pat, tau, text in [("AA", 0, "AA"), ("AA", 0, "AAA"), ("AA", 0, "AAAA"), ("AA", 0, "AAAAA"),
                           ("AA", 0, "AAACAA"), ("GAAG", 0, "GAAGAAG"), ("GAAG", 1, "GAAGAAG"),
                           ("GAAG", 1, "GAAGACG"), ("A", 0, ""), ("A", 0, "A"), ("ACGT", 3, "T"),
                           ("GATC", 1, "TGACTGATGACGTAGTCTACGATCGATCAGTCA")]:
        for mo in (SQ_FIRST, SQ_BEST, SQ_ALL):
            hits = ref.string_match(pat, tau, text, mo)
            cases.append(dict(pattern=pat, tau=tau, text=text, options=mo, hits=[list(h) for h in hits]))
    with open(os.path.join(HERE, "ref_string_cases.json"), "w") as f:
        json.dump(cases, f, separators=(",", ":"))

    # File-level: a small shape-R file (own generator) + FASTQ-ish + FASTA-ish files.
    pat20 = "GATGTAGCGCGATTAGCCTG"
    pat40 = "GATG[TA]AGCNCGATTAGC[CG]TGAAAATGNGAGTAC[GAT]GCGCGA"
    files = {}
    reads = orc.synth_reads(0, 3000, 150, pat20, 3).tobytes().decode()
    files["reads_small.txt"] = reads
    lines = reads.split("\n")[:600]
    fq = []
    for i, ln in enumerate(lines):
        q = ''.join(rng.choice("!\"#$%&'()*+,-./0123456789:;<=>?@ABCDEFGHIJ") for _ in ln)
        fq += ["@r%d" % i, ln, "+", q]
    files["fastq_small.txt"] = "\n".join(fq) + "\n"
    fa = []
    for i, ln in enumerate(lines[:300]):
        fa += [">seq%d some description" % i, ln[:80], ln[80:]]
    files["fasta_small.txt"] = "\n".join(fa)  # no trailing newline on purpose
    reads250 = orc.synth_reads(0, 800, 250, plain(pat40), 5).tobytes().decode()
    files["reads250_small.txt"] = reads250
    for name, content in files.items():
        with open(os.path.join(HERE, name), "w") as f:
            f.write(content)
    cli_cases = []
    for name, pat, d in [("reads_small.txt", pat20, 3), ("fastq_small.txt", pat20, 3),
                         ("fasta_small.txt", pat20, 3), ("reads250_small.txt", pat40, 5),
                         ("reads_small.txt", "GATTAGC", 1)]:
        path = os.path.join(HERE, name)
        for x in ("0", "1", "2"):
            for extra in ([], ["-b"], ["-a"]):
                args = ["-f", "-d", str(d), "-x", x] + extra + [pat]
                cli_cases.append(cli_case(name, args, path))
            for extra in (["-c"], ["-i", "-l"], ["-l", "-p", "-k", "-n"], ["-m"], ["-r"], ["-e"]):
                args = ["-d", str(d), "-x", x] + extra + [pat]
                cli_cases.append(cli_case(name, args, path))
    with open(os.path.join(HERE, "ref_cli_cases.json"), "w") as f:
        json.dump(cli_cases, f, separators=(",", ":"))
    print("string cases:", len(cases), "cli cases:", len(cli_cases))


if __name__ == "__main__":
    main()
