#!/bin/bash
export TMPDIR=/tmp
REPO=$PWD
O=$REPO/gpurun_out/r04p; mkdir -p $O
cd /tmp
(rocprofv3-avail list 2>/dev/null || rocprofv3 --list-avail 2>/dev/null || rocprofv3 -L 2>/dev/null) > $O/avail.txt 2>&1
grep -i -o "TCP_UTCL1[A-Z0-9_]*\|UTCL2[A-Z0-9_]*\|TCP_TA_TCP_STATE_READ[A-Z0-9_]*\|TCP_PENDING_STALL[A-Z0-9_]*\|TCC_EA0_RDREQ[A-Z0-9_]*\|TCC_EA_RDREQ[A-Z0-9_]*\|TCC_TAG_STALL[A-Z0-9_]*\|TCP_TCC_READ_REQ_LATENCY[A-Z0-9_]*\|TCC_EA0_RD_UNCACHED[A-Z0-9_]*" $O/avail.txt | sort -u | head -60
wc -l $O/avail.txt
