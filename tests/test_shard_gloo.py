"""The N>1 path: line sharding + count all-reduce + line-number base, world_size 2 over gloo.

CPU (`-m "not gpu"`): each rank counts its shard with the oracle -- this checks the sharding arithmetic, the
collectives and the merge.  GPU (`-m gpu`): the SAME worker with the HIP path as the ranks' scanner (two processes on
the one GPU of the test box, gloo for the 24-byte count reduce; on a multi-GPU node bench.py runs it over RCCL), plus
single-process tests that scan the shards of one buffer cut mid-file on two scan contexts and compare the merged
records and line numbers with the oracle on the whole buffer."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from conftest import GOLDEN, ROOT

PAT, TAU, N, LEN = "GATGTAGCGCGATTAGCCTG", 3, 6001, 150


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _scan_shard(kind, buf, opt, fasta=False):
    """The ranks' scanner: 'oracle' (CPU checker) or 'hip' (the product: device C-ABI on cuda:0)."""
    if kind == "oracle":
        from oracle.pyoracle import Oracle
        res = Oracle().buffer_scan(PAT, TAU, buf, opt, fasta=fasta)
        return dict(nlines=res["nlines"], nmatchlines=res["nmatchlines"], nhits=len(res["records"]), records=res["records"])
    from seeq_amd import device as dev
    pat = dev.Pattern(PAT, TAU)
    sc = dev.Scanner()
    res = sc.scan_host(pat, bytes(buf), opt | (dev.SEEQDEV_FASTA if fasta else 0), dev.WANT_RECORDS)
    assert sc.last_kernel() in ("k_pair", "k_stream", "k_direct")
    sc.close()
    pat.close()
    return res


def _worker(rank, world, port, out, kind):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from oracle.pyoracle import Oracle
    from seeq_amd import shard
    dist.init_process_group("gloo", rank=rank, world_size=world)
    first, count = shard.shard_range(N, rank, world)
    buf = Oracle().synth_reads(first, count, LEN, PAT, TAU)
    res = _scan_shard(kind, buf, 1)
    local = dict(nlines=res["nlines"], nmatchlines=res["nmatchlines"], nhits=res["nhits"])
    total = shard.reduce_counts(local)
    base = shard.line_base(local["nlines"])
    out.put((rank, first, count, local, total, base, np.asarray(res["records"], dtype=np.uint64)))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_ranges():
    sys.path.insert(0, ROOT)
    from seeq_amd import shard
    for n in (0, 1, 7, 100, 6001):
        for w in (1, 2, 3, 8):
            r = [shard.shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and sum(c for _, c in r) == n
            assert all(r[k][0] + r[k][1] == r[k + 1][0] for k in range(w - 1))
            assert max(c for _, c in r) - min(c for _, c in r) <= 1


def _two_ranks(oracle, kind):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, kind)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=300) for _ in range(world)), key=lambda g: g[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    whole = oracle.buffer_scan(PAT, TAU, oracle.synth_reads(0, N, LEN, PAT, TAU), 1)
    merged = []
    for rank, first, count, local, total, base, rec in got:
        assert total == dict(nlines=N, nmatchlines=whole["nmatchlines"], nhits=len(whole["records"]))
        assert base == first                      # one line per read: line base == first read index
        r = rec.copy()
        r[:, 0] += base                           # rank-local line numbers -> global (records stay rank-local otherwise)
        merged.append(r)
    assert sum(g[3]["nmatchlines"] for g in got) == whole["nmatchlines"]
    assert np.array_equal(np.concatenate(merged), whole["records"])


def test_two_rank_count_reduce(oracle):
    _two_ranks(oracle, "oracle")


@pytest.mark.gpu
def test_two_rank_count_reduce_hip(gpu, oracle):
    """world_size 2 with the HIP path as every rank's scanner (both ranks on this box's one GPU)."""
    _two_ranks(oracle, "hip")


def _cut_cases():
    fa = open(os.path.join(GOLDEN, "fasta_small.txt"), "rb").read()
    fq = open(os.path.join(GOLDEN, "fastq_small.txt"), "rb").read()
    rd = open(os.path.join(GOLDEN, "reads_small.txt"), "rb").read()
    return [("reads", rd, False), ("fastq", fq, False), ("fasta", fa, True), ("reads-no-trailing-newline", rd[:-1], False)]


def test_cut_at_newlines_and_merge(oracle):
    """Byte-range sharding of real files: cuts land right after a newline; the merged per-shard oracle results equal
    the oracle on the whole buffer -- counts, records and line numbers, FASTA headers discounted across the cuts."""
    from seeq_amd import shard
    for name, buf, fasta in _cut_cases():
        whole = oracle.buffer_scan(PAT, TAU, buf, 2, fasta=fasta)
        for world in (1, 2, 3, 8):
            ranges = shard.cut_at_newlines(buf, world)
            assert ranges[0][0] == 0 and ranges[-1][1] == len(buf)
            assert all(ranges[k][1] == ranges[k + 1][0] for k in range(world - 1))
            assert all(hi == lo or hi == len(buf) or buf[hi - 1:hi] == b"\n" for lo, hi in ranges)
            parts = []
            for lo, hi in ranges:
                r = oracle.buffer_scan(PAT, TAU, buf[lo:hi], 2, fasta=fasta)
                parts.append(dict(nlines=r["nlines"], nmatchlines=r["nmatchlines"], nhits=len(r["records"]), records=r["records"]))
            m = shard.merge_results(parts)
            assert m["nlines"] == whole["nlines"] and m["nmatchlines"] == whole["nmatchlines"], (name, world)
            assert np.array_equal(m["records"], whole["records"]), (name, world)
    # a FASTA buffer cut so that a header is the first line of the second shard, and one where it is the last of the first
    fa = b">h1\nACGT\n>h2 " + PAT.encode() + b"\n" + PAT.encode() + b"\nACGT\n"
    for cutpos in (fa.index(b">h2"), fa.index(PAT.encode() + b"\nACGT")):
        parts = []
        for piece in (fa[:cutpos], fa[cutpos:]):
            r = oracle.buffer_scan(PAT, TAU, piece, 2, fasta=True)
            parts.append(dict(nlines=r["nlines"], nmatchlines=r["nmatchlines"], nhits=len(r["records"]), records=r["records"]))
        whole = oracle.buffer_scan(PAT, TAU, fa, 2, fasta=True)
        assert np.array_equal(shard.merge_results(parts)["records"], whole["records"])


@pytest.mark.gpu
def test_two_contexts_cut_mid_file_hip(gpu, capi, oracle):
    """The HIP path on the shards of one buffer cut mid-file, each shard on its own scan context (what a rank per GPU
    does), merged on the host: records, line numbers and counts equal the oracle on the whole buffer -- reads, FASTQ,
    FASTA (a header right at the cut), no trailing newline, FIRST / BEST / ALL."""
    from seeq_amd import device as dev
    from seeq_amd import shard
    pat = dev.Pattern(PAT, TAU)
    ctx = [dev.Scanner(), dev.Scanner(), dev.Scanner()]
    fa_cut = b">h1\nACGT\n>h2 " + PAT.encode() + b"\n" + PAT.encode() + b"\nACGT\n" + b">h3\n" + PAT.encode()[:18] + b"\n"
    for name, buf, fasta in _cut_cases() + [("fasta-header-at-cut", fa_cut, True)]:
        for opt in (0, 1, 2):
            whole = oracle.buffer_scan(PAT, TAU, buf, opt, fasta=fasta)
            for world in (2, 3):
                ranges = shard.cut_at_newlines(buf, world)
                if name == "fasta-header-at-cut" and world == 2:
                    c = buf.index(b">h2")
                    ranges = [(0, c), (c, len(buf))]
                parts = [ctx[k].scan_host(pat, buf[lo:hi], opt | (dev.SEEQDEV_FASTA if fasta else 0), dev.WANT_RECORDS)
                         for k, (lo, hi) in enumerate(ranges)]
                m = shard.merge_results(parts)
                assert m["nlines"] == whole["nlines"] and m["nmatchlines"] == whole["nmatchlines"], (name, opt, world)
                assert np.array_equal(m["records"], whole["records"]), (name, opt, world)
    for c in ctx:
        c.close()
    pat.close()


@pytest.mark.gpu
def test_bench_launches_its_own_ranks(gpu):
    """`bench.py --gpus 2` without a launcher starts two ranks itself (here both on GPU 0, gloo collectives:
    SEEQ_BENCH_SHARE_GPU=1) and prints one line with n_gpus 2 and both ranks' lines in the total."""
    import json
    import subprocess
    env = dict(os.environ, SEEQ_BENCH_SHARE_GPU="1")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--reads", "300000", "--steps", "2",
                        "--warmup", "1", "--no-cpu-baseline", "--no-e2e", "--no-per-call", "--check-lines", "100000"],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["results"]["lines"] == 600000
    assert line["results"]["oracle_lines_checked"] >= 100000
    # both ranks in the line: disjoint contiguous read ranges, their own clocks, their counts adding up to the total
    rows = line["ranks"]
    assert [r["rank"] for r in rows] == [0, 1]
    assert rows[0]["first_read"] == 0 and rows[1]["first_read"] == rows[0]["reads"] == 300000
    assert all(r["lines"] == 300000 and r["ms_per_step"] > 0 and r["forward_scan_ms"] > 0 for r in rows)
    assert sum(r["matching_lines"] for r in rows) == line["results"]["matching_lines"]
    assert rows[0]["matching_lines"] != rows[1]["matching_lines"] or rows[0]["matching_lines"] > 0     # (different reads: the generator is indexed by the read number)
    assert line["ms_per_step"] >= max(r["ms_per_step"] for r in rows) * 0.999                          # the step is the slowest rank's
    # under N ranks rank 0 runs none of the single-GPU sections: the run ends soon after the timed steps, every rank says when it leaves
    assert line["seconds"]["sections_after"] < 60 and "cfg5" not in line and "fastq_shape" not in line and "packed_scan" not in line
    assert line["slowest_rank"]["rank"] in (0, 1) and "placement_chosen" in line["slowest_rank"]
    assert r.stderr.count("leaving the process group") == 2
    assert list(line)[-2:] == ["results", "roofline"]


@pytest.mark.gpu
def test_bench_default_line_sections_at_small_sizes(gpu):
    """The sections of the default `bench.py` line, shrunk: the headline (text from seeqdevTextAllocFor, `first_allocation` beside it), the
    FASTQ shape (k_pair under SQ_FAIL / SQ_CONVERT since round 5, counts = the reference binary's, prefix records = the oracle's) and
    BASELINE configs[4] as a section with its own roofline and full check; `results` and `roofline` close the line."""
    import json
    import subprocess
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--reads", "1500000", "--steps", "3", "--warmup", "1", "--first-steps", "2",
                        "--placement-candidates", "3", "--fastq-records", "400000", "--cfg5-reads", "700000", "--cpu-sample", "200000",
                        "--no-e2e", "--no-per-call", "--no-packed", "--no-multi", "--no-cli", "--check-lines", "200000"],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert list(line)[-4:] == ["placement", "first_allocation", "results", "roofline"]
    assert line["results"]["lines"] == 1500000 and line["results"]["oracle_check"]["result"] == "bit-exact"
    assert line["placement"]["probed"] == 3 and line["first_allocation"]["steps"] == 2
    fq = line["fastq_shape"]
    assert fq["lines"] == 1600000 and all(fq["modes"][m]["oracle_prefix_records_identical"] for m in ("fail", "convert", "ignore")), fq
    assert fq["modes"]["fail"]["kernel"] == "k_pair" and fq["modes"]["convert"]["kernel"] == "k_pair", fq
    c5 = line["cfg5"]
    assert c5["results"]["lines"] == 700000 and c5["results"]["oracle_check"]["result"] == "bit-exact" and 0 < c5["whole_step_frac"] < 1, c5
    assert "device_resident" in line["regions"] and line["cpu_baseline"]["value"] > 0
    from oracle.pyoracle import REF_BIN
    if os.path.exists(REF_BIN):
        assert c5["results"]["oracle_check"]["reference_lines_checked"] == 700000
        assert all(fq["modes"][m]["identical_to_reference_count"] for m in ("fail", "convert", "ignore")), fq


@pytest.mark.gpu
def test_bench_rccl_process_group_over_one_rank(gpu):
    """The first RCCL call of this repository must not be the driver's 8-GPU run: `bench.py --force-dist` under the driver's own
    launcher (`python -m torch.distributed.run --nproc-per-node 1`) initialises the `nccl` backend (= RCCL), runs the join
    check, the per-step count all-reduce and the line-numbering all-gather (seeq.c:377) on the GPU, and says so in its line."""
    import json
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--reads", "2000000",
                        "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-e2e", "--no-per-call", "--no-packed", "--no-cli",
                        "--no-multi", "--no-cfg5", "--no-fastq", "--check-lines", "100000"], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    d = line["dist"]
    assert d["backend"] == "nccl" and d["world"] == 1 and d["all_reduce_of_ones"] == 1 and d["all_gather_ok"] and d["line_base_of_rank0"] == 0
    assert line["n_gpus"] == 1 and line["results"]["lines"] == 2000000 and line["results"]["oracle_lines_checked"] >= 100000


@pytest.mark.gpu
def test_bench_dry_run_reports_the_memory_a_rank_of_eight_needs(gpu):
    """`bench.py --gpus 8 --dry-run` on ONE GPU: allocates what a rank of the 8-GPU line holds (100 M reads of text, the scan
    workspace, the record buffers), reports it against the card's memory, starts no ranks and scans nothing."""
    import json
    import subprocess
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dry-run"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    h = line["hbm_per_rank"]
    assert line["dry_run"] and h["ranks"] == 8 and h["fits"]
    assert h["text_bytes"] == 100_000_000 * 151 and h["text_workspace_records_bytes"] >= h["text_bytes"]
    assert h["text_workspace_records_bytes"] < h["total_bytes"] // 4           # 288 GB per GPU: a rank's share is a small part of it


@pytest.mark.gpu
@pytest.mark.parametrize("workload,reads", [("best", 30_000_000), ("cfg5", 18_000_000)])
def test_two_real_segments_full_size_parity(gpu, workload, reads):
    """More than one REAL segment (3.75 GiB each: 30 M x 151 B = 4.5 GB, 18 M x 251 B = 4.5 GB) through `bench.py`'s
    full-size check: a 1 M-line prefix, every 97th block of 64 Ki lines and both sides of the segment seam, records bit for
    bit against the oracle plus per-range line / matching-line counts.  (The small-segment tests use SEEQ_SEGMENT_BYTES;
    this one is the seam at its real size, for the complete automaton and for the partition filter + two-word exact pass.)"""
    import json
    import subprocess
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload, "--reads", str(reads), "--steps", "1",
                        "--warmup", "0", "--first-steps", "2", "--no-cpu-baseline", "--no-e2e", "--no-per-call", "--no-cfg5", "--no-fastq", "--no-multi", "--no-packed"],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    chk = line["results"]["oracle_check"]
    assert line["results"]["lines"] == reads and chk["result"] == "bit-exact"
    assert chk["oracle_lines_checked"] >= 1_000_000 and chk["segment_seams_checked"] >= 1
    assert line["roofline"]["launches_per_step"] >= 2
    # the text lies in a buffer from the PRODUCT's allocator (seeqdevTextAllocFor: several candidates probed with the run's own scan context, the fastest kept, DESIGN.md
    # section 5), and the same text in a plain first allocation was timed beside it
    pl = line["placement"]
    assert pl["api"] == "seeqdevTextAllocFor" and pl["probed"] >= 2 and pl["selected"] and pl["allocated_bytes"] >= reads * (151 if workload == "best" else 251)
    assert pl["probe_forward_ms"][pl["chosen"]] == min(pl["probe_forward_ms"])
    fa = line["first_allocation"]
    assert fa["steps"] == 2 and fa["value"] > 0 and fa["scan_launch_ms"] > 0
    assert list(line)[-2:] == ["results", "roofline"]                            # the driver's record keeps the END of the line
    # ... and EVERY line against the reference binary itself (bench.py --check full, the default when oracle/_ref travelled)
    from oracle.pyoracle import REF_BIN
    if os.path.exists(REF_BIN):
        assert chk["reference_lines_checked"] == reads and chk["reference_rows_compared"] == line["results"]["hits"] and chk["result"] == "bit-exact"
