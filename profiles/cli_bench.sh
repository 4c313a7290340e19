#!/bin/bash
# End-to-end CLI wall clock on the GPU box: seeq-mi355x CLI vs the reference CLI (oracle/_ref/seeq_ref)
# on a page-cache-warm shape-R file.  Usage: bash profiles/cli_bench.sh [reads]
N=${1:-10000000}
F=/dev/shm/seeq_cli_$N.txt
python - <<PY
import sys
sys.path.insert(0, ".")
from oracle.pyoracle import Oracle
o = Oracle()
with open("$F", "wb") as f:
    step = 1000000
    for first in range(0, $N, step):
        o.synth_reads(first, min(step, $N - first), 150, "GATGTAGCGCGATTAGCCTG", 3).tofile(f)
PY
ls -la $F
P=GATGTAGCGCGATTAGCCTG
t() { local s=$(date +%s%N); "$@" > /tmp/cli_out.$$ ; local e=$(date +%s%N); echo "$(( (e - s) / 1000000 )) ms  $(md5sum < /tmp/cli_out.$$ | cut -c1-12)  $*"; }
cat $F > /dev/null
for args in "-c -d 3" "-d 3 -b -f" "-d 3 -a -f" "-d 3 -i -l"; do
  t seeq_amd/bin/seeq $args $P $F
  t seeq_amd/bin/seeq $args $P $F
  if [ -x oracle/_ref/seeq_ref ]; then t oracle/_ref/seeq_ref $args $P $F; fi
done
rm -f $F /tmp/cli_out.$$
