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
