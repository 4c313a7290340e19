python - <<PY
import sys
sys.path.insert(0, ".")
from oracle.pyoracle import Oracle
o = Oracle()
with open("/dev/shm/r.txt", "wb") as f:
    for first in range(0, 10000000, 1000000):
        o.synth_reads(first, 1000000, 150, "GATGTAGCGCGATTAGCCTG", 3).tofile(f)
open("/dev/shm/tiny.txt","w").write("ACGT\n")
PY
t() { local s=$(date +%s%N); "$@" > /dev/null; local e=$(date +%s%N); echo "$(( (e - s) / 1000000 )) ms  $*"; }
P=GATGTAGCGCGATTAGCCTG
t seeq_amd/bin/seeq -c -d 3 $P /dev/shm/tiny.txt
t seeq_amd/bin/seeq -c -d 3 $P /dev/shm/tiny.txt
t cat /dev/shm/r.txt
t dd if=/dev/shm/r.txt of=/dev/null bs=64M
t seeq_amd/bin/seeq -c -d 3 $P /dev/shm/r.txt
SEEQ_CHUNK_BYTES=268435456 t seeq_amd/bin/seeq -c -d 3 $P /dev/shm/r.txt
nproc
rm -f /dev/shm/r.txt /dev/shm/tiny.txt
