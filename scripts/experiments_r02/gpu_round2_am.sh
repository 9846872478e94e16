#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2am; mkdir -p $O
SDX_RT_SEG=0 timeout 300 python scripts/rt_err_probe.py > $O/err.txt 2>&1
SDX_RT_SEG=1 timeout 300 python scripts/rt_err_probe.py >> $O/err.txt 2>&1
