for s in 12 31 44 75 76 77 84 120 138; do timeout 100 python scripts/fuzz_continuum.py $s $((s+1)) 2>&1 | grep FAILED; done
