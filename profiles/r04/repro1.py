import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from seeq_amd import device as dev
n, L = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000, 24
mode = sys.argv[2] if len(sys.argv) > 2 else "rec"
pattern, tau = "GATTAGCC", 1
stream = torch.cuda.current_stream().cuda_stream
text = torch.empty(n * (L + 1), dtype=torch.uint8, device="cuda:0")
dev.synth_reads(text.data_ptr(), 0, n, L, pattern, tau, stream=stream)
torch.cuda.synchronize()
pat = dev.Pattern(pattern, tau)
sc = dev.Scanner(stream)
if mode == "lines":
    sc.run(pat, text.data_ptr(), text.numel(), 0, dev.WANT_COUNTLINES)
    print("lines", sc.fetch(), sc.last_kernel(), flush=True)
elif mode == "match":
    sc.run(pat, text.data_ptr(), text.numel(), 0, dev.WANT_COUNTMATCH)
    print("match", sc.fetch(), sc.last_kernel(), flush=True)
else:
    sel = sys.argv[3].split(",") if len(sys.argv) > 3 else ["best", "first", "all"]
    for opt, name in ((dev.SQ_BEST, "best"), (0, "first"), (2, "all")):
        if name not in sel: continue
        print("run", name, flush=True)
        sc.run(pat, text.data_ptr(), text.numel(), opt, dev.WANT_RECORDS)
        a = sc.fetch()
        print(name, a, sc.last_kernel(), flush=True)
