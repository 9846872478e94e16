"""tests/golden/make_golden.py is a reproducible recipe: run with --verify it regenerates every fixture from the reference
(un-jitted, from /root/reference) into a temporary directory and finds each array identical to the committed one.  Runs only
where the reference and its interpreter exist (the authoring container); skipped on the GPU box."""
import os
import subprocess

import pytest

from conftest import ROOT

PY39 = "/opt/conda/bin/python3.9"


@pytest.mark.skipif(not (os.path.isdir("/root/reference/stardis") and os.path.exists(PY39)), reason="needs /root/reference and its interpreter")
def test_every_fixture_regenerates_bit_identically():
    proc = subprocess.run([PY39, os.path.join(ROOT, "tests", "golden", "make_golden.py"), "--verify"], capture_output=True, text=True, timeout=1200)
    assert proc.returncode == 0, proc.stdout[-3000:] + proc.stderr[-2000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("VERIFY g")]
    assert len(lines) >= 15 and all(ln.endswith("identical") for ln in lines), proc.stdout[-3000:]
    assert "VERIFY: all identical" in proc.stdout
