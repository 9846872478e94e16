"""GPU experiment: the post-processing kernels (direct convolution with scipy's reflect boundary and summation order) on random
sizes against scipy itself, bit for bit: gaussian_filter1d and the rotation kernel of broadening.py:824-877 on spectra of 1 .. 20000
points, kernels shorter and LONGER than the data (reflection more than once), arbitrary non-symmetric weights of odd length (the
reference's kernels are odd; an even length is refused with a ValueError).
python scripts/fuzz_postprocess.py FIRST LAST"""
import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from scipy.ndimage import convolve1d, gaussian_filter1d
from stardis_amd import postprocess as pp

bad = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    try:
        rng = np.random.default_rng(52000 + seed)
        n = int(rng.choice([1, 2, 3, 7, 64, 65, 1000, 4097, 20000]))
        flux = rng.uniform(0.0, 2.0, n) * 10.0 ** rng.uniform(-8, 2)
        sigma = float(10.0 ** rng.uniform(-1.0, 2.2))
        assert np.array_equal(pp.gaussian_filter1d(flux, sigma), gaussian_filter1d(flux, sigma)), ("gaussian", n, sigma)
        m = int(rng.choice([1, 3, 5, 9, 65, 301, 2 * n + 3]))
        k = rng.uniform(-0.2, 1.0, m)
        assert np.array_equal(pp.convolve1d_reflect(flux, k), convolve1d(flux, k)), ("weights", n, m)
        vpp, v = float(10.0 ** rng.uniform(-1.5, 1.0)), float(10.0 ** rng.uniform(-1.0, 2.7))
        ld = float(rng.uniform(0.0, 1.0))
        lam = np.linspace(5000.0, 5100.0, n)
        _, out = pp.rotation_broadening(vpp, lam, flux, v, ld)
        prof = pp.rotation_profile(vpp, v, ld)
        assert np.array_equal(out, convolve1d(flux, prof)), ("rotation", n, vpp, v, ld)
        print(f"seed {seed}: ok  n {n} sigma {sigma:.2f} weights {m} rotation kernel {prof.size}", flush=True)
    except Exception:
        bad += 1
        print(f"seed {seed}: FAILED", flush=True)
        traceback.print_exc()
print("failures:", bad)
