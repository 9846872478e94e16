"""The drop-in boundary used from plain C: examples/c_abi_demo.c (gcc, no Python, no torch, no HIP headers) is compiled
against include/stardis_hip.h, linked with libstardis_hip.so, run on the GPU, and its printed results are checked against
the CPU oracle on the same inputs."""
import os
import shutil
import subprocess

import numpy as np
import pytest

import oracle
from conftest import ROOT, rel_err

pytestmark = pytest.mark.gpu

ND, NNU, NL, NTH = 5, 400, 7, 3
MASK = (1 << 64) - 1


class XorShift:
    def __init__(self):
        self.s = 0x9E3779B97F4A7C15

    def uniform(self):
        s = self.s
        s ^= s >> 12
        s = (s ^ (s << 25)) & MASK
        s ^= s >> 27
        self.s = s
        return ((s * 0x2545F4914F6CDD1D & MASK) >> 11) / 9007199254740992.0


def inputs():
    nus = 4.57e14 - 1.0e9 * np.arange(NNU)
    line_nus = np.array([nus[-1] + (nus[0] - nus[-1]) * (l + 0.5) / NL for l in range(NL)])
    rng = XorShift()
    dw, gam, al = np.empty(NL * ND), np.empty(NL * ND), np.empty(NL * ND)
    for k in range(NL * ND):
        dw[k] = 2.0e9 * (1.0 + rng.uniform())
        gam[k] = 1.0e8 * (1.0 + 9.0 * rng.uniform())
        al[k] = 10.0 ** (-3.0 + 4.0 * rng.uniform())
    temps = 9000.0 - 1000.0 * np.arange(ND)
    thetas = np.array([0.2, 0.8, 1.3])
    dist = 1.0e7 * (1 + np.arange(ND - 1))
    wts = 0.3 + 0.1 * np.arange(NTH)
    return nus, line_nus, dw.reshape(NL, ND), gam.reshape(NL, ND), al.reshape(NL, ND), temps, thetas, dist, wts


def test_plain_c_program_through_the_abi(tmp_path):
    if shutil.which("gcc") is None:
        pytest.skip("no gcc on this box")
    exe = tmp_path / "c_abi_demo"
    libdir = os.path.join(ROOT, "stardis_amd", "lib")
    subprocess.run(["gcc", "-O2", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "c_abi_demo.c"),
                    "-L", libdir, "-lstardis_hip", f"-Wl,-rpath,{libdir}", "-lm", "-o", str(exe)], check=True)
    run = subprocess.run([str(exe)], check=True, capture_output=True, text=True, timeout=120)
    lines = run.stdout.strip().split("\n")
    first = next(k for k, ln in enumerate(lines) if ln.startswith("version stardis_hip"))  # (RCCL prints a banner of its own first)
    lines = lines[first:]
    evals = int(lines[1].split()[1])
    assert int(lines[2].split()[1]) == -1  # an ascending grid is refused as a bad argument
    # sdx_synthesize_sharded_f64 on a group of every visible GPU: RCCL ran (version reported), each rank contributed its padded
    # shard, and F_nu / the gathered emergent flux are those of sdx_synthesize_f64 bit for bit
    sh = lines[3].split()
    assert sh[0] == "sharded" and int(sh[2]) >= 1 and int(sh[4]) == 8 * -(-NNU // int(sh[2])) and int(sh[6]) > 0 and int(sh[8]) == 1, lines[3]
    got = np.array([[float(x) for x in row.split()] for row in lines[4:]])
    assert got.shape == (ND * NNU, 3)
    nus, line_nus, dw, gam, al, temps, thetas, dist, wts = inputs()
    line, ref_evals = oracle.calc_alan_entries(ND, nus, line_nus, dw, gam, al, return_evals=True)
    assert evals == ref_evals
    total = line + 1.0e-9
    assert rel_err(got[:, 0].reshape(ND, NNU), total) < 1e-12
    F_ref, _ = oracle.raytrace(nus, temps, dist, thetas, wts, total)
    F = got[:, 1].reshape(ND, NNU)
    assert np.all(F[0] == 0) and rel_err(F[1:], F_ref[1:]) < 1e-10
    # the fused synthesis: line opacity + Thomson scattering, then the formal solution
    total_fused = oracle.alpha_electron(NNU, 1.0e17 / (1 + np.arange(ND))) + line
    F_fused, _ = oracle.raytrace(nus, temps, dist, thetas, wts, total_fused)
    F1 = got[:, 2].reshape(ND, NNU)
    assert np.all(F1[0] == 0) and rel_err(F1[1:], F_fused[1:]) < 1e-10
