#!/bin/bash
# A/B of library builds on the far-field probe: scripts/r5/far_ab.sh OUT "TAGS" LIBDIR ...
out=$1; tags=$2; shift 2
for lib in "$@"; do
  echo "=== $lib" >> "$out"
  STARDIS_AMD_LIB=$PWD/stardis_amd/$lib/libstardis_hip.so SDX_EXPERIMENT=1 $EXTRA python scripts/r5/far_probe.py $tags 2>&1 | grep -v "far_field=0" >> "$out"
done
