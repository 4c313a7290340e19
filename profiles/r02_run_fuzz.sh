#!/bin/bash
# Round 2: extended parity fuzz against the oracle on the final build (random patterns x FAIL/CONVERT/IGNORE x FASTA x long lines).
set -u
O=gpurun_out/r02fuzz; mkdir -p $O
FUZZ_SEED=100 timeout -k 10 280 python profiles/extended_fuzz.py > $O/fuzz_100.txt 2>&1; echo "exit $?" >> $O/fuzz_100.txt; tail -3 $O/fuzz_100.txt
FUZZ_SEED=700 FUZZ_FOREIGN=0.05 timeout -k 10 280 python profiles/extended_fuzz.py > $O/fuzz_700_foreign5.txt 2>&1; echo "exit $?" >> $O/fuzz_700_foreign5.txt; tail -3 $O/fuzz_700_foreign5.txt
FUZZ_SEED=900 FUZZ_FOREIGN=0.5 timeout -k 10 280 python profiles/extended_fuzz.py > $O/fuzz_900_foreign50.txt 2>&1; echo "exit $?" >> $O/fuzz_900_foreign50.txt; tail -3 $O/fuzz_900_foreign50.txt
FUZZ_SEED=300 timeout -k 10 280 python profiles/extended_fuzz.py long > $O/fuzz_300_long.txt 2>&1; echo "exit $?" >> $O/fuzz_300_long.txt; tail -3 $O/fuzz_300_long.txt
