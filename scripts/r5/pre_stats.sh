#!/bin/bash
# analysis build of the library (phase time stamps of the pre-pass blocks) and a few steps with it: bash scripts/r5/pre_stats.sh [TAG] [WORLD RANK]
cd $GRAFT_REPO_ROOT
mkdir -p stardis_amd/lib_prestats
[ -f stardis_amd/lib_prestats/libstardis_hip.so ] || (cd stardis_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -DSDX_PRE_STATS -shared -o ../lib_prestats/libstardis_hip.so stardis_hip.hip 2>/dev/null)
STARDIS_AMD_LIB=$PWD/stardis_amd/lib_prestats/libstardis_hip.so python3 scripts/r5/pre_stats.py "$@"
