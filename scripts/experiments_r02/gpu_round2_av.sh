#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2av; mkdir -p $O
STARDIS_AMD_LIB=$GRAFT_REPO_ROOT/_ab/TM/stardis_amd/lib/libstardis_hip.so timeout 300 python scripts/profile_step.py S-c2 4 > $O/timing.txt 2>&1
