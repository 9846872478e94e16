#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2j; mkdir -p $O
timeout 900 python scripts/strong_scaling_probe.py S-c3 1 8 --balanced > $O/strong_c3.txt 2>&1
SDX_NO_CULL=1 timeout 600 python scripts/strong_scaling_probe.py S-c3 8 --balanced > $O/strong_c3_nocull.txt 2>&1
