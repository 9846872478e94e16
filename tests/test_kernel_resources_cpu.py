"""Register budgets of the hot kernels, checked at build time (hipcc cross-compiles without a GPU).  Round 5 measured what a few
spilled vector registers cost on this chip — a pre-pass variant with 6 spilled VGPRs ran 45 - 57 us where the spill-free one runs
36 - 47, the line kernel with 24 spilled in its wide walk 2.77 ms instead of 2.03 — and that a 1024-thread pre-pass block needs
<= 64 VGPRs for two blocks to share a CU.  `make -C stardis_amd/csrc resources` prints the compiler's figures; this test holds
them."""
import os
import re
import shutil
import subprocess

import pytest

from conftest import ROOT

HIPCC = "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def resources():
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    proc = subprocess.run(["make", "-C", os.path.join(ROOT, "stardis_amd", "csrc"), "resources"], capture_output=True, text=True, timeout=900)
    text = proc.stdout + proc.stderr
    assert "Function Name" in text, text[-2000:]
    cur, table = None, {}
    for line in text.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = m.group(1)
            table[cur] = {}
        for key, tag in (("VGPRs:", "vgpr"), ("VGPRs Spill:", "spill"), ("Occupancy [waves/SIMD]:", "occ"), ("LDS Size [bytes/block]:", "lds")):
            if cur and key in line:
                table[cur][tag] = int(line.split(key)[1].split()[0])
    filt = shutil.which("c++filt")
    names = {}
    for mangled in table:
        out = subprocess.run([filt, mangled], capture_output=True, text=True).stdout if filt else mangled
        names[out.split("(")[0].replace("void sdx::", "").strip()] = table[mangled]
    return names


def test_hot_kernels_do_not_spill_vector_registers(resources):
    hot = [k for k in resources if k.startswith(("k_line_all<", "k_line_far<", "k_raytrace<1>", "k_raytrace_seg<8", "k_line_prepass<", "k_prepass_continuum<"))]
    assert len(hot) >= 10, sorted(resources)
    for k in hot:
        # the GENERATING pre-pass variants (<true, ...>: line parameters from per-line scalars, f1) evaluate every pow, log and tgamma of the
        # block's depth points and lines once per block (gen_depth / gen_line, two waves) and park up to 21 registers around that section;
        # measured in round 6 at 1e6 lines: 1.59 ms with these spills against 1.84 for the spill-free round-5 kernel (two library exp
        # calls per item) — and 2.44 ms with the section moved out of line to get rid of them (its calls copy LineParams to scratch)
        cold_calls = k.startswith(("k_line_prepass<true", "k_prepass_continuum<true"))
        assert resources[k]["spill"] <= (24 if cold_calls else 0), (k, resources[k])


def test_prepass_blocks_fit_two_per_cu(resources):
    """1024-thread blocks: 8 waves per SIMD (<= 64 VGPRs) and <= 80 KB of LDS, or only ONE block fits a CU"""
    pre = [k for k in resources if k.startswith(("k_line_prepass", "k_prepass_continuum"))]
    assert len(pre) >= 12, sorted(resources)
    for k in pre:
        assert resources[k]["occ"] == 8 and resources[k]["vgpr"] <= 64 and resources[k]["lds"] <= 80 * 1024, (k, resources[k])


def test_line_kernels_keep_their_occupancy(resources):
    assert resources["k_line_all<4, false, false>"]["occ"] >= 7 and resources["k_line_all<4, true, false>"]["occ"] >= 7
    # (the kernels of the far field queue their hits: twelve more registers, six waves — measured against five and seven)
    assert resources["k_line_all<4, false, true>"]["occ"] >= 6 and resources["k_line_all<4, true, true>"]["occ"] >= 6
    assert resources["k_line_far<4, 2>"]["occ"] >= 6 and resources["k_line_far<4, 1>"]["occ"] >= 7
    assert resources["k_raytrace<1>"]["occ"] >= 7 and resources["k_raytrace_seg<8, 7>"]["occ"] >= 6
    assert resources["k_line_all_mixed<4, false, false>"]["occ"] >= 6


def test_analysis_builds_compile():
    """the two analysis builds (per-wave statistics of the line kernel, phase time stamps of the pre-pass) still compile"""
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    src = os.path.join(ROOT, "stardis_amd", "csrc", "stardis_hip.hip")
    for macro in ("SDX_PRE_STATS", "SDX_WALK_STATS"):
        proc = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O1", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", f"-D{macro}",
                               "-c", "-o", os.devnull, src], capture_output=True, text=True, timeout=900)
        assert proc.returncode == 0, (macro, proc.stderr[-2000:])
