#!/bin/bash
# round 5, call W: after the k_direct region clamp -- the new graded test, the saved buffer under the forced paths, eight fresh seeds of the fuzz (four processes at a time)
out=$PWD/gpurun_out/r05_w; mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_randomized.py -x -q -m gpu -k "kinds" > $out/pytest.log 2>&1; echo "pytest $?"; tail -3 $out/pytest.log
timeout -k 10 300 python3 profiles/r05/replay_buf.py profiles/r05/fuzzbuf_790497392_19.bin AACAAAAAAAAA 1 > $out/replay.log 2>&1; echo "replay exit $?"; grep -c "identical True" $out/replay.log; grep -c "identical False" $out/replay.log
rc=0
for grp in "1 2 3 4" "5 6 7 8"; do
  pids=""
  for i in $grp; do timeout -k 10 900 python3 profiles/ignore_fuzz.py $(( ( $(date +%s%N) / 1000 + i * 7919 ) % 1000000007 )) 40 > $out/fuzz_$i.log 2>&1 & pids="$pids $!"; done
  for p in $pids; do wait $p || rc=1; done
done
for i in 1 2 3 4 5 6 7 8; do echo "--- $i"; head -1 $out/fuzz_$i.log; tail -2 $out/fuzz_$i.log | cut -c1-1200; done
exit $rc
