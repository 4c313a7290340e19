"""Known answers of the reference's own test-suite (test/testset.c), restated as data.

Each block cites the lines of /root/reference/test/testset.c it restates.
`testdata.txt` (tests/golden/testdata.txt) is the 3-line data file those tests use.
"""

TESTDATA_LINES = ["GTATGTACCACAGATGTCGATCGAC", "TCTATCATCCGTACTCTGATCTCAT", "RCACAGATCACAGATCACAGRATCAC"]

# testset.c:719-757  pattern -> key bytes, or seeqerr
PARSE_OK = [
    ("AaAaAaAa", [1] * 8), ("CcCcCcCc", [2] * 8), ("GgGgGgGg", [4] * 8), ("TtTtTtTt", [8] * 8),
    ("NnNnNnNn", [31] * 8), ("Nn[]Nn[]NnN[]n", [31] * 8), ("[GATC][gatc][GaTc][gAtC]", [15] * 4),
    ("ACTGA", [1, 2, 8, 4, 1]), ("ACG[AT]", [1, 2, 4, 9]),          # testset.c:768-801
]
PARSE_ERR = [("[GATCgatc", 5), ("A]", 3), ("[ATG[C]]", 2), ("Z", 4)]

# testset.c:803-826  (pattern, tau) -> seeqerr of seeqNew
SEEQNEW_ERR = [("ACG[AT]", -1, 1), ("ACG[AT]", 4, 9), ("ACT[A[AG]", 1, 2), ("ACT[AG]T]A", 1, 3),
               ("ACHT[AG]", 1, 4), ("ACT[AG]A[TG", 1, 5)]

# testset.c:546-622 and :628-685: per-character (capped distance, min_to_match)
TRACE = [
    ("CATG", 1, "ATCCTCATGA", [2, 2, 2, 2, 2, 2, 2, 1, 0, 1], [2, 1, 2, 2, 1, 2, 1, 0, 0, 0]),
    ("AAAA", 1, "ATTAAAT", [2, 2, 2, 2, 2, 1, 1], [2, 2, 3, 2, 1, 0, 0]),
]

# testset.c:939-1030: seeqStringMatch -> sq->match[] (index 0 first)
STRING_MATCH = [
    ("GATC", 1, "TGACTGATGACGTAGTCTACGATCGATCAGTCA", "FIRST", [(1, 4, 1)]),
    ("GATC", 1, "TGACTGATGACGTAGTCTACGATCGATCAGTCA", "BEST", [(20, 24, 0)]),
    ("GATC", 1, "TGACTGATGACGTAGTCTACGATCGATCAGTCA", "ALL",
     [(29, 32, 1), (24, 28, 0), (20, 24, 0), (14, 17, 1), (8, 11, 1), (5, 9, 1), (1, 4, 1)]),
    ("GAAG", 0, "GAAGAAG", "ALL", [(3, 7, 0), (0, 4, 0)]),
    ("GAAG", 1, "GAAGAAG", "ALL", [(3, 7, 0), (0, 4, 0)]),
    ("GAAG", 1, "GAAGACG", "ALL", [(3, 7, 1), (0, 4, 0)]),
]

# testset.c:833-929: sequences of seeqFileMatch calls on testdata.txt.
# (pattern, tau, [(match_opt, file_opt, retval, hits, line, string, [(start,end,dist) as popped by seeqMatchIter])])
FILE_MATCH = [
    ("ATCG", 1, [("FIRST", "MATCH", 1, 1, 1, TESTDATA_LINES[0], [(2, 5, 1)]),
                 ("FIRST", "MATCH", 1, 1, 2, TESTDATA_LINES[1], [(3, 7, 1)]),
                 ("FIRST", "MATCH", 0, None, None, None, None)]),
    ("TGTC", 1, [("BEST", "MATCH", 1, 1, 1, TESTDATA_LINES[0], [(14, 18, 0)]),
                 ("BEST", "MATCH", 1, 1, 2, TESTDATA_LINES[1], [(2, 6, 1)])]),
    ("CACAGAT", 1, [("FIRST", "NOMATCH", 1, 0, 2, TESTDATA_LINES[1], []),
                    ("FIRST", "NOMATCH", 1, 0, 3, TESTDATA_LINES[2], [])]),
    ("CACAGAT", 1, [("BEST", "ANY", 1, 1, 1, TESTDATA_LINES[0], [(8, 15, 0)]),
                    ("BEST", "ANY", 1, 0, None, TESTDATA_LINES[1], []),
                    ("FIRST", "MATCH", 0, None, None, None, None)]),
]
FILE_COUNTS = [("ATC", 0, "COUNTLINES", 2), ("ATC", 0, "COUNTMATCH", 4)]      # testset.c:915-929

# testset.c:1078-1208: CLI stdout on testdata.txt (flags as the reference CLI would set args)
CLI = [
    (["CACAGAT"], "GTATGTACCACAGATGTCGATCGAC\n"),
    (["-l", "-p", "-k", "CACAGAT"], "1 8-14 0 GTATGTACCACAGATGTCGATCGAC\n"),
    (["-f", "-d", "3", "CACAGAT"], "1:8-14:0\n2:8-11:3\n"),
    (["-c", "CACAGAT"], "1\n"),
    (["-i", "CACAGAT"], "TCTATCATCCGTACTCTGATCTCAT\nRCACAGATCACAGATCACAGRATCAC\n"),
    (["-i", "-l", "CACAGAT"], "2 TCTATCATCCGTACTCTGATCTCAT\n3 RCACAGATCACAGATCACAGRATCAC\n"),
    (["-m", "-d", "3", "CACAGAT"], "CACAGAT\nCCGT\n"),
    (["-m", "-d", "3", "-x", "1", "CACAGAT"], "CACAGAT\nCCGT\nCACAGAT\n"),
    (["-b", "-d", "1", "-m", "CTCAT"], "CTCAT\n"),
    (["-d", "1", "-m", "CTCAT"], "CTAT\n"),
    (["-r", "-d", "3", "CACAGAT"], "GTATGTAC\nTCTATCAT\n"),
    (["-e", "-d", "3", "CACAGAT"], "GTCGATCGAC\nACTCTGATCTCAT\n"),
    (["-l", "-m", "-x", "2", "CACAGAT"], "1 CACAGAT\n3 CACAGAT\n"),
    (["-a", "-l", "-x", "1", "CACAGAT"], "1 CACAGAT\n3 CACAGAT\n3 CACAGAT\n"),
    (["-a", "-l", "-x", "2", "CACAGAT"], "1 CACAGAT\n3 CACAGAT\n3 CACAGAT\n3 CACAGRAT\n"),
    (["-c", "-d", "2", "GTATGTACCACA"], "1\n"),            # BASELINE.json configs[0]
]

# test/python_lib_test.py:15-35 and SURVEY section 8c (captured from the reference module)
PY_PATTERN, PY_TAU = "CGCTAATTAATGGAAT", 3
PY_MATCH, PY_NOMATCH = "GGGGCGCTAATAATGGAATGGGG", "ATGCTGATGCTGGGGG"
PY_EXPECT = dict(prefix_true="GGGGCGCTAATAATGGAAT", prefix_false="GGGG", suffix_true="CGCTAATAATGGAATGGGG",
                 suffix_false="GGGG", matchlist=[(4, 19, 1)], tokenize=("GGGG", "CGCTAATAATGGAAT", "GGGG"),
                 split=("GGGG", "GGGG"))
