#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2ae; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
timeout 900 python bench.py > $O/bench_full.json 2> $O/bench_full.err
for CW in "14 0.6" "18 0.6" "14 1.2"; do
  set -- $CW
  echo "== core_weight $1 scan_weight $2" >> $O/strong.txt
  SDX_CORE_WEIGHT=$1 SDX_SCAN_WEIGHT=$2 timeout 900 python scripts/strong_scaling_probe.py S-c3 1 2 4 8 --balanced >> $O/strong.txt 2>&1
done
