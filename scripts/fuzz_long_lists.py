"""GPU experiment: the seeded random long-list configurations of tests/test_gpu_long_random.py beyond the four the suite runs.
python scripts/fuzz_long_lists.py FIRST LAST [--mixed] [--ticket] [--raw]   (--ticket: the counter-driven pre-pass of very long lists forced on
these; --raw: option narrow_records = 0, the narrow role reads the caller's tables whatever the density of the list)"""
import os, sys, traceback
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tests"))
import test_gpu_long_random as T
from stardis_amd._lib import default_context

ctx = default_context()
if "--ticket" in sys.argv:
    ctx.set_option("prepass_ticket_min_blocks", 0)
if "--raw" in sys.argv:
    ctx.set_option("narrow_records", 0)
bad = 0
args = [a for a in sys.argv[1:] if not a.startswith("--")]
for seed in range(int(args[0]), int(args[1])):
    try:
        if "--mixed" in sys.argv:
            print(f"seed {seed}: ok mixed vs fp64 (opacity, flux) = {T.check_long_case_mixed(ctx, seed)}", flush=True)
        else:
            print(f"seed {seed}: ok (n_depth, n_nu, n_lines, n_theta) = {T.check_long_case(ctx, seed)}", flush=True)
    except Exception:
        bad += 1
        print(f"seed {seed}: FAILED", flush=True)
        traceback.print_exc()
print("failures:", bad)
