#!/usr/bin/env python3
"""Golden vectors for the match-stack utilities of libseeq.h (stackNew / stackAddMatch / recursive_merge, reference
libseeq.c:355-424), generated from the reference itself (oracle/_ref/libseeq_ref.so; build container only).

A case: tau, and per distance 0..tau a stack of matches in text order.  Every stack starts with a sentinel [0, 1) and the
merge is asked for [1, 1000): the sentinel is never popped, so the reference's loop that drops "upper overlaps" never
steps below the bottom of a stack (it reads match[-1] when it does).  Output: sq->match after the merge (in the order the
reference leaves it) and what is left on every stack.  -> ref_stack_cases.json (data only)."""
import ctypes as C
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


class match_t(C.Structure):
    _fields_ = [("start", C.c_size_t), ("end", C.c_size_t), ("dist", C.c_size_t)]


class seeq_t(C.Structure):
    _fields_ = [("hits", C.c_size_t), ("stacksize", C.c_size_t), ("match", C.POINTER(match_t)),
                ("bufsz", C.c_size_t), ("string", C.c_void_p), ("tau", C.c_int), ("wlen", C.c_int),
                ("keys", C.c_void_p), ("rkeys", C.c_void_p), ("dfa", C.c_void_p), ("rdfa", C.c_void_p)]


class mstack_head(C.Structure):
    _fields_ = [("size", C.c_size_t), ("pos", C.c_size_t)]


def bind(L):
    L.stackNew.restype = C.c_void_p
    L.stackNew.argtypes = [C.c_size_t]
    L.stackAddMatch.restype = C.c_int
    L.stackAddMatch.argtypes = [C.POINTER(C.c_void_p), match_t]
    L.recursive_merge.restype = C.c_int
    L.recursive_merge.argtypes = [C.c_size_t, C.c_size_t, C.c_int, C.POINTER(seeq_t), C.POINTER(C.c_void_p)]
    return L


def run_case(L, libc, case):
    """-> (merged [[start, end, dist], ...], per stack what is left [[start, end, dist], ...])"""
    tau = case["tau"]
    stacks = (C.c_void_p * (tau + 1))()
    for d, ivs in enumerate(case["stacks"]):
        st = C.c_void_p(L.stackNew(1))                    # grows by doubling: stackAddMatch is exercised too
        for s, e in ivs:
            assert L.stackAddMatch(C.byref(st), match_t(s, e, d)) == 0
        stacks[d] = st
    sq = seeq_t()
    sq.tau = tau
    sq.stacksize = 4
    libc.malloc.restype = C.c_void_p
    sq.match = C.cast(libc.malloc(4 * C.sizeof(match_t)), C.POINTER(match_t))
    assert L.recursive_merge(1, 1000, 0, C.byref(sq), stacks) == 0
    merged = [[sq.match[i].start, sq.match[i].end, sq.match[i].dist] for i in range(sq.hits)]
    left = []
    for d in range(tau + 1):
        head = C.cast(stacks[d], C.POINTER(mstack_head)).contents
        arr = C.cast(stacks[d] + C.sizeof(mstack_head), C.POINTER(match_t))
        left.append([[arr[i].start, arr[i].end, arr[i].dist] for i in range(head.pos)])
        libc.free(C.c_void_p(stacks[d]))
    libc.free(C.cast(sq.match, C.c_void_p))
    return merged, left


def rand_case(rng):
    tau = rng.randint(0, 4)
    stacks = []
    for _ in range(tau + 1):
        ivs, p = [[0, 1]], 2
        while True:
            p += rng.randint(0, 40)
            ln = rng.randint(1, 25)
            if p + ln >= 990:
                break
            ivs.append([p, p + ln])
            p += ln
            if rng.random() < 0.1:
                break
        stacks.append(ivs)
    return {"tau": tau, "stacks": stacks}


def main():
    from oracle.pyoracle import REF_LIB
    L = bind(C.CDLL(REF_LIB))
    libc = C.CDLL(None)
    rng = random.Random(20261004)
    cases = []
    for _ in range(300):
        c = rand_case(rng)
        c["merged"], c["left"] = run_case(L, libc, c)
        cases.append(c)
    with open(os.path.join(HERE, "ref_stack_cases.json"), "w") as f:
        json.dump(cases, f, separators=(",", ":"))
    print("%d stack cases, %d merged matches" % (len(cases), sum(len(c["merged"]) for c in cases)))


if __name__ == "__main__":
    main()
