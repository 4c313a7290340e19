#!/bin/bash
# CLI on the reference's benchmark shape: a FASTA-like file with one chromosome per line (8 x 128 MiB, headers optional),
# seeq-mi355x vs the reference binary, stdout md5 compared.  Usage: bash profiles/cli_chromosome.sh
F=/dev/shm/seeq_chrom.txt
python - <<PY
import numpy as np
rng = np.random.default_rng(5)
L, n = 128 << 20, 8
pat = np.frombuffer(b"GATGTAGCGCGATTAGCCTG", dtype=np.uint8)
with open("$F", "wb") as f:
    for i in range(n):
        a = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, L - 1, dtype=np.uint8)]
        for _ in range(6):
            p = int(rng.integers(100, L - 100)); c = pat.copy()
            for _e in range(int(rng.integers(0, 4))): c[int(rng.integers(0, 20))] = b"ACGT"[int(rng.integers(0, 4))]
            a[p:p + 20] = c
        f.write((">chr%02d\n" % i).encode()); a.tofile(f); f.write(b"\n")
PY
ls -la $F
P=GATGTAGCGCGATTAGCCTG
t() { local s=$(date +%s%N); "$@" > /tmp/cli_out.$$ ; local e=$(date +%s%N); echo "$(( (e - s) / 1000000 )) ms  $(md5sum < /tmp/cli_out.$$ | cut -c1-12)  $(wc -l < /tmp/cli_out.$$) lines out  $*"; }
cat $F > /dev/null
for args in "-c -d 3" "-d 3 -a -f" "-d 3 -b -p -k -m"; do
  t seeq_amd/bin/seeq $args $P $F
  if [ -x oracle/_ref/seeq_ref ]; then t oracle/_ref/seeq_ref $args $P $F; fi
done
rm -f $F /tmp/cli_out.$$
