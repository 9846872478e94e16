# round 3, experiment M: where an eighth of S-c3 loses against the whole grid — line subsets per tile (SDX_WIDE_BLOCKS) and
# the two roles of k_line_all timed apart (SDX_SPLIT_LAUNCHES), whole grid and the eight shards
mkdir -p gpurun_out
for f in 1 2 4; do
echo "== roles apart, SDX_NARROW_F=$f"
SDX_NARROW_F=$f SDX_SPLIT_LAUNCHES=1 python scripts/strong_scaling_probe.py S-c3 1 8 --balanced --verbose 2>&1 | grep -v "^    rank [1-6]"
done
for p in 2; do
echo "== SDX_RT_P=$p"
SDX_RT_P=$p python scripts/strong_scaling_probe.py S-c3 1 8 --balanced --verbose 2>&1 | grep -v "^    rank [1-6]"
done
