"""CPU: the product's HOST code -- libseeq_api.c, seeq_file.c (ingest pipeline: reader thread, lanes, SEEQ_DEVICES, replay),
seeq_main.c (CLI, output formatter) -- built with -fsanitize=address,undefined and with -fsanitize=thread on top of a test
double of the device C-ABI (tests/fake_seeqdev.c: include/seeq_amd.h served by the oracle on worker threads), and run
through the reference-generated CLI fixtures and through call sequences of its own.  No GPU is involved and nothing here
is a product path: the product library has no CPU matcher (test_capi_host.py::test_no_gpu_fails_loudly).

Also: the reference's own callers (src/seeq-main.c, src/seeq.c, src/seeqmodule.c) compile and link against include/ +
libseeq_amd.so unchanged -- the drop-in claim of INTEGRATION.md -- whenever /root/reference is present."""
import hashlib
import os
import select
import subprocess
import sys
import sysconfig
import time

import pytest

from conftest import GOLDEN, ROOT

CSRC = os.path.join(ROOT, "seeq_amd", "csrc")
BUILD = os.path.join(ROOT, "tests", "build")
HOST_SRC = [os.path.join(CSRC, "seeq_file.c"), os.path.join(CSRC, "libseeq_api.c"),
            os.path.join(ROOT, "tests", "fake_seeqdev.c"), os.path.join(ROOT, "oracle", "seeq_oracle.c")]
INC = ["-I" + os.path.join(ROOT, "include"), "-I" + CSRC]
SAN = {"asan": ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"], "tsan": ["-fsanitize=thread"]}
SAN_ENV = {"ASAN_OPTIONS": "detect_leaks=0:abort_on_error=0", "UBSAN_OPTIONS": "print_stacktrace=1",
           "TSAN_OPTIONS": "halt_on_error=1:second_deadlock_stack=1"}


def _build(kind, main, out):
    os.makedirs(BUILD, exist_ok=True)
    exe = os.path.join(BUILD, out + "_" + kind)
    srcs = [main] + HOST_SRC
    deps = srcs + [os.path.join(ROOT, "include", h) for h in ("libseeq.h", "seeq.h", "seeq_amd.h")]
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(d) for d in deps):
        cmd = ["gcc", "-std=gnu99", "-g", "-O1", "-fno-omit-frame-pointer", "-pthread"] + SAN[kind] + INC + srcs + ["-o", exe]
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-3000:]
    return exe


@pytest.fixture(scope="module", params=["asan", "tsan"])
def host_cli(request):
    exe = _build(request.param, os.path.join(CSRC, "seeq_main.c"), "seeq_host")
    # TSan needs a kernel / personality it can map its shadow into: probe once
    r = subprocess.run([exe, "--version"], capture_output=True, text=True, env=dict(os.environ, **SAN_ENV))
    if request.param == "tsan" and r.returncode != 0 and "ThreadSanitizer" in r.stderr and "unexpected memory mapping" in r.stderr:
        pytest.skip("ThreadSanitizer cannot map its shadow memory in this container")
    return exe, request.param


def _run(exe, args, env_extra=None, **kw):
    env = dict(os.environ, **SAN_ENV)
    env.update(env_extra or {})
    return subprocess.run([exe] + args, capture_output=True, env=env, **kw)


def test_cli_fixtures_under_sanitizers(host_cli, cli_cases):
    """The 135 reference-generated CLI invocations through the host code, with tiny chunks (every line straddles a chunk
    boundary somewhere), several lanes and several (fake) devices: byte-identical stdout, no sanitizer report."""
    exe, kind = host_cli
    variants = [{"SEEQ_CHUNK_BYTES": "64", "SEEQ_LANES": "1"},
                {"SEEQ_CHUNK_BYTES": "1000", "SEEQ_LANES": "2", "FAKE_SEEQ_DEVICES": "3", "SEEQ_DEVICES": "0-2"},
                {"SEEQ_CHUNK_BYTES": "70000", "SEEQ_LANES": "3", "FAKE_SEEQ_DEVICES": "2", "SEEQ_DEVICES": "all"}]
    cases = cli_cases if kind == "asan" else cli_cases[::4]          # (TSan runs are slow: every fourth case)
    for c in cases:
        for env in variants if kind == "asan" else variants[1:2]:
            r = _run(exe, c["args"] + [os.path.join(GOLDEN, c["file"])], env)
            out = r.stdout.decode("latin-1")
            err = r.stderr.decode("latin-1")
            assert "Sanitizer" not in err and "runtime error" not in err, (c["file"], c["args"], env, err[-3000:])
            if "stdout" in c:
                assert out == c["stdout"], (c["file"], c["args"], env, err[-500:])
            else:
                assert len(out) == c["nbytes"] and hashlib.sha256(out.encode()).hexdigest() == c["sha256"], (c["file"], c["args"], env)


def test_pipe_streaming_under_sanitizers(host_cli):
    """`producer | seeq`: lines are matched and printed as they arrive (reference: getline loop, seeq.c:361), an unfinished
    line waits for its newline, and closing the pipe ends the run."""
    exe, kind = host_cli
    env = dict(os.environ, **SAN_ENV)
    p = subprocess.Popen([exe, "-d", "1", "-l", "CACAGAT"], stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         bufsize=0, env=env)
    try:
        def expect(text, within=30.0):
            got = b""
            end = time.time() + within
            while len(got) < len(text) and time.time() < end:
                if select.select([p.stdout], [], [], 0.25)[0]:
                    piece = os.read(p.stdout.fileno(), 4096)
                    if not piece:
                        break
                    got += piece
            assert got == text, (got, text)
        p.stdin.write(b"TTTTCACAGATTTT\n")
        expect(b"1 TTTTCACAGATTTT\n")
        p.stdin.write(b"GGGG\nAAAA\n")
        time.sleep(0.2)
        p.stdin.write(b"CACAGTT")                          # an unfinished line ...
        time.sleep(0.2)
        p.stdin.write(b"\nxCACAGATx\n")                    # ... finished here; the next one has a non-DNA byte in front
        expect(b"4 CACAGTT\n")
        p.stdin.close()
        assert p.stdout.read() == b""
        assert p.wait(timeout=30) == 0
        err = p.stderr.read().decode("latin-1")
        assert "Sanitizer" not in err and "runtime error" not in err, err[-3000:]
    finally:
        if p.poll() is None:
            p.kill()


@pytest.mark.parametrize("kind", ["asan", "tsan"])
def test_file_api_sequences_under_sanitizers(kind):
    """tests/host_driver.c: pattern / option switch in the middle of a file, the engine freed before the file, all file
    options, a file closed while scans are in flight -- against the oracle, under the sanitizers."""
    exe = _build(kind, os.path.join(ROOT, "tests", "host_driver.c"), "host_driver")
    # (one lane and tiny chunks: every read-ahead scan is in flight when the options change -- the configuration in which
    #  seeqFileMatch once slept for good; four fake devices listed out of order: one pattern copy per device, all freed
    #  when the engine goes before the file)
    for env in ({"SEEQ_CHUNK_BYTES": "300"}, {"SEEQ_CHUNK_BYTES": "300", "SEEQ_LANES": "1"},
                {"SEEQ_CHUNK_BYTES": "5000", "SEEQ_LANES": "3", "FAKE_SEEQ_DEVICES": "2", "SEEQ_DEVICES": "0,1"},
                {"SEEQ_CHUNK_BYTES": "2000", "SEEQ_LANES": "2", "FAKE_SEEQ_DEVICES": "4", "SEEQ_DEVICES": "3,0,2"}, {}):
        r = _run(exe, [os.path.join(GOLDEN, "reads_small.txt")], env, text=True, timeout=600)
        if kind == "tsan" and "unexpected memory mapping" in r.stderr:
            pytest.skip("ThreadSanitizer cannot map its shadow memory in this container")
        assert r.returncode == 0 and r.stdout.strip() == "OK", (env, r.stdout[-500:], r.stderr[-3000:])
        assert "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]


FAULTS = ["begin@1", "begin@2", "begin@5", "fetch@1", "fetch@3", "fetch@6", "copy@1", "copy@4", "offsets@2", "new@1", "new@2", "new@3",
          "pattern@1", "pattern@2", "hostalloc@1", "hostalloc@2", "hostalloc@4", "setdevice@1", "setdevice@3", "fetch@2,begin@9", "fetch@4-"]
FAULT_ENVS = [{"SEEQ_CHUNK_BYTES": "2000", "SEEQ_LANES": "1"},
              {"SEEQ_CHUNK_BYTES": "3000", "SEEQ_LANES": "3"},
              {"SEEQ_CHUNK_BYTES": "5000", "SEEQ_LANES": "3", "FAKE_SEEQ_DEVICES": "3", "SEEQ_DEVICES": "2,0,1"},
              {"SEEQ_CHUNK_BYTES": "100000", "SEEQ_LANES": "2"}]          # (one chunk holds the whole file: the failure is the first and the last chunk's)


@pytest.mark.parametrize("kind", ["asan", "tsan"])
def test_device_failures_under_the_file_api(kind):
    """Round 5: the device boundary FAILING under seeqFileMatch / seeqStringMatch / seeqNew (FAKE_SEEQDEV_FAIL: the N-th call of
    seeqdevScanNew / PatternNew / HostAlloc / SetDevice / ScanHostBegin / ScanFetch / CopyRecords / CopyOffsets / StringMatch returns
    -1 or NULL with errno set, as the HIP library does).  tests/host_fault_driver.c checks the reference's contract (libseeq.c:75-135,
    505; test/faultymalloc.c:20-56 is how the reference tests its own failures): -1 / NULL, seeqerr = 0, errno kept, every line
    delivered before the failure is the oracle's, nothing after it, no hang, seeqClose / seeqFree in either order -- on the first
    chunk, in the middle of the file and on its last chunks, one and several lanes and devices; under ASan with its leak check on
    (a failed call must not leak the chunk, the lane or the scan it was working on), under TSan for the reader thread."""
    exe = _build(kind, os.path.join(ROOT, "tests", "host_fault_driver.c"), "host_fault_driver")
    path = os.path.join(GOLDEN, "reads_small.txt")
    probe = _run(exe, [path, "file", "close-first"], {"SEEQ_CHUNK_BYTES": "3000"}, text=True, timeout=300)
    if kind == "tsan" and "unexpected memory mapping" in probe.stderr:
        pytest.skip("ThreadSanitizer cannot map its shadow memory in this container")
    assert probe.returncode == 0 and probe.stdout.startswith("OK file") and "rv=0" in probe.stdout, (probe.stdout, probe.stderr[-2000:])
    leak = {"ASAN_OPTIONS": "detect_leaks=1:abort_on_error=0"} if kind == "asan" else {}
    # (LeakSanitizer needs ptrace rights some containers do not give: probe once, fall back to the leak check off)
    if leak:
        pr = _run(exe, [path, "string", "close-first"], leak, text=True, timeout=300)
        if "LeakSanitizer has encountered a fatal error" in pr.stderr or "LeakSanitizer does not work" in pr.stderr:
            leak = {}
    failures_seen = 0
    faults = FAULTS if kind == "asan" else FAULTS[::2]
    for fault in faults:
        for ei, env in enumerate(FAULT_ENVS if kind == "asan" else FAULT_ENVS[1:3]):
            for order in (("close-first", "free-first") if ei == 1 else ("close-first",)):
                e = dict(env, FAKE_SEEQDEV_FAIL=fault, **leak)
                r = _run(exe, [path, "file", order], e, text=True, timeout=300)       # (a hang is the timeout's failure)
                assert r.returncode == 0 and r.stdout.startswith("OK"), (fault, env, order, r.stdout[-500:], r.stderr[-3000:])
                assert "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, (fault, env, order, r.stderr[-3000:])
                failures_seen += "rv=-1" in r.stdout or "seeqNew=NULL" in r.stdout
    assert failures_seen >= len(faults)                      # (the injected failures really reached the API: most runs end in -1)
    for fault, mode in (("string@1", "string"), ("string@2-4", "string"), ("new@1", "string"), ("new@1", "new"), ("pattern@1", "new"), ("pattern@2", "new"), ("string@1", "new")):
        if True:
            r = _run(exe, [path, mode, "close-first"], dict(FAKE_SEEQDEV_FAIL=fault, **leak), text=True, timeout=300)
            assert r.returncode == 0 and r.stdout.startswith("OK"), (fault, mode, r.stdout[-500:], r.stderr[-3000:])
            assert "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, (fault, mode, r.stderr[-3000:])


def test_cli_with_a_failing_device_boundary(host_cli, cli_cases):
    """The CLI over the same failures: the reference's seeq() reports a failed seeqFileMatch on stderr -- "error in 'seeqFileMatch()':
    <strerror(errno)>" (seeq.c:105-107,176-178) -- and returns; a failed seeqNew is EXIT_FAILURE (seeq.c:77-81).  What reached stdout
    before the failure is a prefix of the reference's output made of whole lines, nothing is printed after the report, no hang."""
    exe, kind = host_cli
    picked = [c for c in cli_cases if "stdout" in c and c["file"] in ("reads_small.txt", "fastq_small.txt") and len(c["stdout"]) > 2000][:6 if kind == "asan" else 2]
    assert picked
    for c in picked:
        for fault in ("begin@1", "fetch@2", "fetch@5", "copy@3", "offsets@1", "begin@4", "hostalloc@2"):
            for env in ({"SEEQ_CHUNK_BYTES": "3000", "SEEQ_LANES": "1"}, {"SEEQ_CHUNK_BYTES": "2000", "SEEQ_LANES": "3", "FAKE_SEEQ_DEVICES": "2", "SEEQ_DEVICES": "all"}):
                r = _run(exe, c["args"] + [os.path.join(GOLDEN, c["file"])], dict(env, FAKE_SEEQDEV_FAIL=fault), timeout=300)
                out, err = r.stdout.decode("latin-1"), r.stderr.decode("latin-1")
                assert "Sanitizer" not in err and "runtime error" not in err, (c["args"], fault, env, err[-3000:])
                assert c["stdout"].startswith(out) and (out == "" or out.endswith("\n")), (c["args"], fault, env, out[-300:])
                if out != c["stdout"]:                       # the failure reached the run (a count-only case fetches no records: copy@N never fires)
                    assert "error in 'seeqFileMatch()': " in err and r.returncode == 0, (c["args"], fault, env, r.returncode, err[-500:])
                else:
                    assert r.returncode == 0
    for fault in ("pattern@1",):                             # (the scan contexts are made by the file layer: new@N fails seeqFileMatch, above)
        c = picked[0]
        r = _run(exe, c["args"] + [os.path.join(GOLDEN, c["file"])], {"FAKE_SEEQDEV_FAIL": fault}, timeout=300)
        err = r.stderr.decode("latin-1")
        assert "Sanitizer" not in err and "runtime error" not in err, (fault, err[-3000:])
        assert r.returncode == 1 and "error in 'seeqNew()'" in err and r.stdout == b"", (fault, r.returncode, err[-500:])


REF_SRC = "/root/reference/src"


@pytest.mark.skipif(not os.path.isdir(REF_SRC), reason="/root/reference is not present (GPU box)")
def test_reference_callers_link_against_the_drop_in(capi):
    """The reference's OWN callers, unmodified, against include/ + libseeq_amd.so: its CLI front end (seeq-main.c) on this
    library's seeq(); its file layer (seeq.c) on this library's libseeq.h entry points; its CPython module (seeqmodule.c).
    Compiling and linking is the test (running them needs a GPU; the CLI / Python parity tests cover behaviour)."""
    os.makedirs(BUILD, exist_ok=True)
    libdir = os.path.dirname(capi.LIB_PATH)
    inc = ["-I" + os.path.join(ROOT, "include")]
    link = ["-L" + libdir, "-lseeq_amd", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"]
    base = ["gcc", "-std=gnu99", "-O2", "-w"]
    # 1. reference main() + this library's seeq()
    exe1 = os.path.join(BUILD, "ref_main_on_drop_in")
    r = subprocess.run(base + inc + [os.path.join(REF_SRC, "seeq-main.c")] + link + ["-o", exe1], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    # 2. reference main() + reference seeq.c (its own seeqFileMatch loop) + this library's libseeq.h entry points
    exe2 = os.path.join(BUILD, "ref_main_and_file_layer_on_drop_in")
    r = subprocess.run(base + inc + [os.path.join(REF_SRC, "seeq-main.c"), os.path.join(REF_SRC, "seeq.c")] + link + ["-o", exe2],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    # 3. the reference's CPython extension
    pyinc = sysconfig.get_paths()["include"]
    if os.path.exists(os.path.join(pyinc, "Python.h")):
        so = os.path.join(BUILD, "ref_seeqmodule_on_drop_in.so")
        r = subprocess.run(base + ["-shared", "-fPIC", "-I" + pyinc] + inc + [os.path.join(REF_SRC, "seeqmodule.c")] + link + ["-o", so],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-3000:]
        nm = subprocess.run(["nm", "-D", "--undefined-only", so], capture_output=True, text=True).stdout
        undefined = {ln.split()[-1] for ln in nm.splitlines() if ln.split()[-1].startswith("seeq")}
        assert undefined and undefined <= set(capi.EXPORTS), undefined      # every seeq* it needs is one this library exports
    # the executables resolve their seeq* symbols in the drop-in (no reference libseeq linked)
    for exe in (exe1, exe2):
        nm = subprocess.run(["nm", "-D", "--undefined-only", exe], capture_output=True, text=True).stdout
        undefined = {ln.split()[-1] for ln in nm.splitlines() if ln.split()[-1].startswith("seeq")}
        assert undefined and undefined <= set(capi.EXPORTS), (exe, undefined)
