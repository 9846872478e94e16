"""GPU check + timing of the saturation workload of SURVEY §8d: S-big = 3000-10000 A at R = 1e6 (1 203 973 frequencies), 1e6 lines,
gamma (N_l, 1).  Evaluation count against the host window rule, every line of the list on 201 columns against the oracle, the
flux of those columns from the oracle's own total, step time as a graph replay.
python scripts/s_big_check.py [TAG]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from stardis_amd import constants as K
from stardis_amd import parallel, synth
from stardis_amd.engine import SpectralSynthesizer

tag = sys.argv[1] if len(sys.argv) > 1 else "S-big"
t0 = time.perf_counter()
w = synth.make_workload(tag)
atm, nus, lines, cont = w["atm"], w["nus"], w["lines"], w["cont"]
print(f"{tag}: N_nu {nus.size} N_l {lines['line_nus'].size} (host set-up {time.perf_counter() - t0:.1f} s)", flush=True)
syn = SpectralSynthesizer(nus, atm["temperatures"], atm["dist"], w["thetas"], w["weights"], lines, cont)
syn.step()
ev = syn.evaluations()
ev_host = parallel.window_evaluations(nus, lines["line_nus"], lines["doppler_widths"], lines["gammas"], lines["alphas"])
print(f"evaluations {ev:.6e} host window rule {ev_host:.6e} equal {ev == ev_host}", flush=True)
F, total, line = syn.F_nu(), syn.total_alphas(), syn.alpha_line()
rel = lambda a, b: float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)))
cols = np.arange(300, nus.size, nus.size // 200)
cutoff = (cont["ionization_energy"] - cont["level_excitation"]) / K.H_CGS
c_ref = oracle.alpha_file_1d(K.nu_to_angstrom(nus[cols]), cont["hminus_bf_wavelength"], cont["hminus_bf_cross_section"], cont["n_hminus"])
c_ref = c_ref + oracle.alpha_bf(nus[cols], [0, len(cutoff)], [0], cutoff, cont["level_density"])
c_ref = c_ref + oracle.alpha_ff(nus[cols], atm["temperatures"], [1], cont["n_e"] * cont["n_h2"])
c_ref = c_ref + oracle.alpha_electron(cols.size, cont["n_e"])
t0 = time.perf_counter()
line_ref, ev_cols = oracle.calc_alan_entries_columns(cols, 56, nus, lines["line_nus"], lines["doppler_widths"], lines["gammas"], lines["alphas"], return_evals=True)
F_cpu, _ = oracle.raytrace(nus[cols], atm["temperatures"], atm["dist"], w["thetas"], w["weights"], c_ref + line_ref)
print(f"{cols.size} columns, {ev_cols:.3e} oracle evaluations ({time.perf_counter() - t0:.1f} s): line opacity rel err {rel(line[:, cols], line_ref):.2e}, "
      f"total {rel(total[:, cols], c_ref + line_ref):.2e}, flux {rel(F[1:, cols], F_cpu[1:]):.2e}; zero pattern equal {np.array_equal(line[:, cols] == 0, line_ref == 0)}", flush=True)
del F, total, line
syn.keep_line = False
syn.capture()
for _ in range(3): syn.step()
syn.synchronize()
t0 = time.perf_counter()
n = 10
for _ in range(n): syn.step()
syn.synchronize()
ms = (time.perf_counter() - t0) / n * 1e3
print(f"step {ms:.2f} ms = {nus.size * 56 / ms * 1e3:.3e} spectral points/s, {ev / ms * 1e3:.3e} evaluations/s, algorithmic bytes {syn.algorithmic_bytes() / 1e9:.2f} GB", flush=True)
