import sys, time
sys.path.insert(0, ".")
import seeq_amd as seeq
from oracle.pyoracle import Oracle
orc = Oracle()
pat="GATGTAGCGCGATTAGCCTG"
data = orc.synth_reads(0, 20000, 150, pat, 3).tobytes()
texts=[s.decode() for s in data.split(b"\n")[:20000]]
m = seeq.compile(pat, 3)
m.matchBestBatch(texts[:100])
for i in range(4):
    t0=time.perf_counter(); r=m.matchBestBatch(texts); dt=time.perf_counter()-t0
    print("call", i, round(dt*1e3,2), "ms", round(len(texts)/dt/1e6,3), "M strings/s", sum(1 for x in r if x))
import cProfile, pstats
cProfile.run("m.matchBestBatch(texts)", "/tmp/prof.out")
pstats.Stats("/tmp/prof.out").sort_stats("cumtime").print_stats(12)
