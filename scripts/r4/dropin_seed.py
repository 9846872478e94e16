import sys, tempfile
sys.path.insert(0,"/root/repo"); sys.path.insert(0,"/root/repo/tests")
import test_gpu_dropin_random as TR
for seed in map(int, sys.argv[1:]):
    with tempfile.TemporaryDirectory() as tmp:
        print(seed, TR.random_dropin_case(seed, tmp))
