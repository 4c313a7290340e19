"""seeqStringMatch: microseconds per call against the string length (is the per-call cost the launch or the kernel?)."""
import sys, time, random
sys.path.insert(0, ".")
from seeq_amd import _capi
L = _capi.lib()
sq = L.seeqNew(b"GATGTAGCGCGATTAGCCTG", 3, 0)
rng = random.Random(1)
for n in (1, 20, 75, 150, 300, 600, 2000, 8000, 16000, 32000):
    strs = ["".join(rng.choice("ACGT") for _ in range(n)).encode() for _ in range(64)]
    for s in strs: L.seeqStringMatch(s, sq, 1)
    t0 = time.perf_counter(); k = 0
    for _ in range(30):
        for s in strs:
            L.seeqStringMatch(s, sq, 1); k += 1
    dt = time.perf_counter() - t0
    print("len %5d: %.1f us per call" % (n, 1e6 * dt / k))
