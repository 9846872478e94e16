#!/bin/bash
# round 3, experiment C: continuum plane rides with the classification launch of a culled shard
python -m pytest tests/test_gpu_engine.py tests/test_gpu_group.py tests/test_gpu_multi.py -x -q 2>&1 | grep -E "passed|failed|error" | tail -3
echo "== ride (default)"; python scripts/strong_scaling_probe.py S-c3 1 8 --balanced
echo "== no ride"; SDX_NO_CONT_RIDE=1 python scripts/strong_scaling_probe.py S-c3 8 --balanced
python bench.py --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/r3_bench_c.json 2> gpurun_out/r3_bench_c.err; python - <<'PY'
import json
d=json.load(open("gpurun_out/r3_bench_c.json"))
print("S-c2 ms", d["ms_per_step"], "cold", d["ms_per_step_cold"], d["roofline"]["avg_kernel_ms"])
print("dropin", json.dumps(d["dropin"], indent=1))
for k,v in d["secondary"].items(): print(k, v.get("ms_per_step", v.get("ms_per_step_with_postprocessing")), {a:b for a,b in v.items() if a in ("avg_kernel_ms","mixed_vs_fp64","hbm_bytes_saved")})
PY
