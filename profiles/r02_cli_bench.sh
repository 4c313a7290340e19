#!/bin/bash
# Round 2: CLI wall clock with the pipelined ingest, on the GPU box, page-cache-warm shape-R files (10 M and 40 M lines;
# /dev/shm holds them), -z phase split, a "hello" run for the fixed process + HIP start-up cost, reference CLI beside it.
O=${1:-gpurun_out/r02_cli}; mkdir -p $O
P=GATGTAGCGCGATTAGCCTG
gen() { python - <<PY
import sys
sys.path.insert(0, ".")
from oracle.pyoracle import Oracle
o = Oracle()
with open("$2", "wb") as f:
    step = 1000000
    for first in range(0, $1, step):
        o.synth_reads(first, min(step, $1 - first), 150, "GATGTAGCGCGATTAGCCTG", 3).tofile(f)
PY
}
t() { local s=$(date +%s%N); "$@" > /tmp/cli_out.$$ 2> /tmp/cli_err.$$; local e=$(date +%s%N); echo "$(( (e - s) / 1000000 )) ms  $(md5sum < /tmp/cli_out.$$ | cut -c1-12)  $*"; grep -E "ingest:|caller waited" /tmp/cli_err.$$ | sed 's/^/      /'; }
printf 'ACGT\n' > /dev/shm/seeq_tiny.txt
echo "== start-up: a 5-byte file (process start, HIP init, first launch)"
t seeq_amd/bin/seeq -c ACGT /dev/shm/seeq_tiny.txt
t seeq_amd/bin/seeq -c ACGT /dev/shm/seeq_tiny.txt
for N in 10000000 40000000; do
  F=/dev/shm/seeq_cli_$N.txt
  gen $N $F; ls -la $F; cat $F > /dev/null
  echo "== $N lines"
  for args in "-c -d 3" "-d 3 -b -f" "-d 3 -a -f" "-d 3 -i -l"; do
    t seeq_amd/bin/seeq -z $args $P $F
    t seeq_amd/bin/seeq $args $P $F
    if [ $N = 10000000 ] && [ -x oracle/_ref/seeq_ref ]; then t oracle/_ref/seeq_ref $args $P $F; fi
  done
  echo "-- one lane (no overlap of H2D and kernels), 16 MiB chunks, 256 MiB chunks"
  SEEQ_LANES=1 t seeq_amd/bin/seeq -z -c -d 3 $P $F
  SEEQ_CHUNK_BYTES=16777216 t seeq_amd/bin/seeq -z -c -d 3 $P $F
  SEEQ_CHUNK_BYTES=268435456 t seeq_amd/bin/seeq -z -c -d 3 $P $F
  rm -f $F
done
rm -f /tmp/cli_out.$$ /tmp/cli_err.$$ /dev/shm/seeq_tiny.txt
