#!/usr/bin/env python3
"""Round 5: a differential fuzz of the CLI against the REFERENCE BINARY (oracle/_ref/seeq_ref, compiled from the reference's own sources by
oracle/Makefile): files made of line kinds (reads, quality-like lines, headers, '+', empty lines, short lines, lines of a few KB, CR LF ends,
a NUL now and then, copies of the pattern with foreign bytes inside), random patterns (classes, N), distances, match modes (-b -a -i), non-DNA
modes (-x 0 1 2) and format options (-c -m -n -l -p -k -f -e -r in any combination, the ones the reference rejects included): stdout and the exit
status of seeq_amd/bin/seeq must be the reference's, byte for byte -- half of the runs with the ingest pipeline cut into small chunks (700 bytes
to 300 KB, 1-3 lanes, a device listed twice).  Bytes >= 0x80 are left out (the reference indexes a table with a negative char
there: a stated difference).  Usage: python profiles/cli_diff_fuzz.py [seed] [files] [invocations per file]"""
import os
import random
import subprocess
import sys
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OURS = os.path.join(ROOT, "seeq_amd", "bin", "seeq")
REF = os.path.join(ROOT, "oracle", "_ref", "seeq_ref")
seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else (int(time.time() * 1000) ^ os.getpid()) % 1_000_000_007
nfiles = int(sys.argv[2]) if len(sys.argv) > 2 else 6
ninv = int(sys.argv[3]) if len(sys.argv) > 3 else 24
print("CLI_DIFF_FUZZ_SEED=%d files=%d invocations=%d" % (seed0, nfiles, ninv), flush=True)
assert os.path.exists(OURS) and os.path.exists(REF), "build first: make -C seeq_amd/csrc && make -C oracle"
QUAL = "".join(chr(c) for c in range(33, 75))
SKIP = "!#$%&*+-./0123456789:;<=>?@BDEFHIJ"


def mutate(rng, s, k):
    s = list(s)
    for _ in range(k):
        r = rng.random()
        if r < 0.5 and s:
            s[rng.randrange(len(s))] = rng.choice("ACGT")
        elif r < 0.75 and len(s) > 1:
            del s[rng.randrange(len(s))]
        else:
            s.insert(rng.randrange(len(s) + 1), rng.choice("ACGT"))
    return "".join(s)


tot = 0
tmp = tempfile.mkdtemp(prefix="clifuzz")
for fno in range(nfiles):
    rng = random.Random(seed0 * 1000 + fno)
    m = rng.choice([6, 10, 16, 20, 20, 25, 30])
    core = "".join(rng.choice("ACGT") for _ in range(m))
    pattern = core
    if rng.random() < 0.3:
        pl = list(core)
        for _ in range(rng.randint(1, 2)):
            pl[rng.randrange(m)] = rng.choice(["N", "[AC]", "[GT]", "[ACG]", "n", "[ac]"])
        pattern = "".join(pl)
    if rng.random() < 0.2:
        pattern = pattern.lower()
    crlf = rng.random() < 0.15
    dna_share = rng.choice([0.0, 0.1, 0.3, 0.6])
    lines = []
    for i in range(rng.choice([300, 1500, 5000])):
        kind = rng.random()
        if kind < 0.35:
            n = rng.choice([50, 100, 150, 150, 151, 250]); t = [rng.choice("ACGT") for _ in range(n)]
            if rng.random() < 0.05:
                t = [c.lower() if rng.random() < 0.6 else ("U" if c == "T" else c) for c in t]
        elif kind < 0.60:
            n = rng.choice([50, 100, 150, 151]); t = [rng.choice("ACGTN") if rng.random() < dna_share else rng.choice(QUAL) for _ in range(n)]
        elif kind < 0.70:
            t = list("@r%09d %s" % (i, "".join(rng.choice("acgtnACGTlength=xyz0123 ") for _ in range(rng.randint(0, 40)))))
        elif kind < 0.78:
            t = list("+")
        elif kind < 0.80:
            t = []
        elif kind < 0.88:
            n = rng.choice([300, 600, 1200, 4000]); sh = rng.choice([0.0, 0.02, 0.3])
            t = [rng.choice("ACGT") if rng.random() >= sh else rng.choice(SKIP) for _ in range(n)]
        else:
            t = [rng.choice("ACGTN" + SKIP) for _ in range(rng.randint(1, 30))]
        n = len(t)
        for _ in range(rng.choice([0, 0, 1, 1, 2])):
            if n < m + 8:
                break
            c = list(mutate(rng, core, rng.randint(0, 3)))
            r = rng.random()
            if r < 0.35:
                for _ in range(rng.randint(1, 3)):
                    c.insert(rng.randrange(1, len(c)), rng.choice(SKIP))
            p = rng.randrange(0, max(1, n - len(c)))
            t[p:p + len(c)] = c
            t = t[:n]
        if rng.random() < 0.003 and n:
            t[rng.randrange(n)] = "\0"
        if crlf:
            t.append("\r")
        lines.append("".join(t))
    data = ("\n".join(lines) + ("\n" if fno % 3 else "")).encode("latin-1")
    path = os.path.join(tmp, "f%d.txt" % fno)
    open(path, "wb").write(data)
    jobs = []
    for _ in range(ninv):
        args = ["-d", str(rng.randint(0, min(4, m // 5 + 1)))]
        args += rng.choice([[], [], ["-b"], ["-a"], ["-i"], ["-b", "-i"]])
        if rng.random() < 0.6:
            args += ["-x", str(rng.randint(0, 2))]
        fmt = [f for f in ("-c", "-m", "-n", "-l", "-p", "-k", "-f", "-e", "-r") if rng.random() < 0.22]
        rng.shuffle(fmt)
        # the ingest pipeline of OUR binary (seeq_file.c): now and then small chunks (lines longer than a chunk: it has to grow), 1-3 lanes, one device listed twice
        env = {}
        if rng.random() < 0.5:
            env["SEEQ_CHUNK_BYTES"] = str(rng.choice([700, 3000, 20000, 65536, 300001]))
            env["SEEQ_LANES"] = str(rng.randint(1, 3))
            if rng.random() < 0.3:
                env["SEEQ_DEVICES"] = "0,0"
        jobs.append((args + fmt, env))

    def run(job):
        args, env = job
        a = subprocess.run([OURS] + args + [pattern, path], capture_output=True, timeout=120, env=dict(os.environ, **env))
        b = subprocess.run([REF] + args + [pattern, path], capture_output=True, timeout=120)
        return args + ["   env:"] + ["%s=%s" % kv for kv in sorted(env.items())], a, b
    with ThreadPoolExecutor(max_workers=int(os.environ.get("CLI_FUZZ_PAR", "4"))) as pool:      # (CLI processes on the card at a time)
        for args, a, b in pool.map(run, jobs):
            if a.returncode != b.returncode or a.stdout != b.stdout:
                k = 0
                while k < min(len(a.stdout), len(b.stdout)) and a.stdout[k] == b.stdout[k]:
                    k += 1
                print("DIFF seed", seed0, "file", fno, "pattern", pattern, "args", " ".join(args), "rc", a.returncode, b.returncode, "stdout bytes", len(a.stdout), len(b.stdout),
                      "first difference at", k, "\n   ours:", a.stdout[max(0, k - 60):k + 120], "\n   ref: ", b.stdout[max(0, k - 60):k + 120],
                      "\n   stderr ours:", a.stderr[-300:], "\n   stderr ref: ", b.stderr[-300:], flush=True)
                open(os.path.join(ROOT, "gpurun_out", "clifuzz_fail_%d_%d.txt" % (seed0, fno)), "wb").write(data) if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else None
                sys.exit(1)
            tot += 1
print("cli diff fuzz OK:", tot, "invocations identical to the reference binary")
