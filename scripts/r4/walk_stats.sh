#!/bin/bash
# analysis build of the library (wide-role wave statistics) and one step with it: bash scripts/r4/walk_stats.sh [TAG] [WORLD RANK]
cd $GRAFT_REPO_ROOT
mkdir -p stardis_amd/lib_stats
[ -f stardis_amd/lib_stats/libstardis_hip.so ] || (cd stardis_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -DSDX_WALK_STATS $SDX_STATS_DEFS -shared -o ../lib_stats/libstardis_hip.so stardis_hip.hip)
STARDIS_AMD_LIB=$PWD/stardis_amd/lib_stats/libstardis_hip.so python3 scripts/r4/walk_stats.py "$@"
