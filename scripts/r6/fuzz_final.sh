#!/bin/bash
# Round 6, final commit: every seeded fuzzer on fresh seed ranges (records: profiles/r06_fuzz_final_*.txt).  bash scripts/r6/fuzz_final.sh OUTDIR
O=${1:-gpurun_out/fuzz_final}; mkdir -p $O
timeout 600 python scripts/fuzz_random_cases.py 7000 7060 > $O/random_cases.txt 2>&1
timeout 900 python scripts/fuzz_long_lists.py 1400 1430 > $O/long_lists.txt 2>&1
timeout 900 python scripts/fuzz_long_lists.py 1430 1450 --mixed > $O/long_lists_mixed.txt 2>&1
timeout 600 python scripts/fuzz_long_lists.py 1450 1462 --raw > $O/long_lists_raw.txt 2>&1
timeout 600 python scripts/fuzz_fused_dropin.py 800 860 > $O/dropin.txt 2>&1
timeout 600 python scripts/fuzz_continuum.py 800 880 > $O/continuum.txt 2>&1
timeout 600 python scripts/fuzz_linelist.py 800 840 > $O/linelist.txt 2>&1
timeout 600 python scripts/fuzz_raytrace.py 800 900 > $O/raytrace.txt 2>&1
timeout 600 python scripts/fuzz_postprocess.py 800 860 > $O/postprocess.txt 2>&1
SDX_EXPERIMENT=1 SDX_GROUP_LOOPBACK=1 timeout 600 python scripts/fuzz_group_loopback.py 800 830 > $O/group.txt 2>&1
for f in $O/*.txt; do echo "$(basename $f): $(grep -c -i 'ok' $f) ok lines, $(grep -c -i 'fail\|mismatch\|error\|Traceback' $f) fail lines; last: $(tail -1 $f | cut -c1-160)"; done
