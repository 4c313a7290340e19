"""--best / first-hit / count on the dense cells of the sweep shape (24 lines x 128 MiB): how long one lane per line takes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from seeq_amd import device as dev
import chrom_sweep as cs
torch.cuda.set_device(0)
text = cs.make_text(24, 128 << 20, "cuda:0")
sc = dev.Scanner(torch.cuda.current_stream().cuda_stream)
for m, k in ((20, 5), (42, 14)):
    pat = dev.Pattern(cs.FULL[:m], k)
    for name, opt, want in (("all", dev.SQ_ALL, dev.WANT_RECORDS), ("best", dev.SQ_BEST, dev.WANT_RECORDS), ("first", 0, dev.WANT_RECORDS), ("count lines", 0, dev.WANT_COUNTLINES)):
        best = None
        for it in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            sc.run(pat, text.data_ptr(), text.numel(), opt, want); c = sc.fetch()
            dt = time.perf_counter() - t0
            if it: best = dt if best is None else min(best, dt)
        print("m=%d k=%d %-12s %9.2f ms  records %d matching lines %d kernel %s" % (m, k, name, best * 1e3, c["nrecords"], c["nmatchlines"], sc.last_kernel()), flush=True)
    pat.close()
