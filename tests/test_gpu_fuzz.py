"""A bounded, fixed-seed slice of the randomised parity sweeps (tests/fuzz_parity.py: the bug hunt that found the
NaN-depth point-face and the collinear-face bounding-box bugs) inside the driver's ``-m gpu`` run: eight sweeps of 40
cases each -- random / tiny / degenerate / collinear / off-screen faces with textures and all three image gradients
against the C oracle, indexed meshes through the vertex-colour path and the compat entry points, warp / occlusion /
pair loss, MANO LBS, get_opticalflow fused vs op-by-op vs oracle, head post-processing, trunk glue, frames -> batch,
the fused pair node (what training runs) vs the oracle and the composed path."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_fixed_seed_slice_of_the_parity_sweeps(cuda):
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz_parity.py"), "40", "77000"], capture_output=True,
                         text=True, timeout=1500, cwd=ROOT)
    assert res.returncode == 0, (res.stdout[-2000:], res.stderr[-2000:])
    counts = [int(m) for m in re.findall(r"(\d+) with mismatches", res.stdout)]
    assert len(counts) == 8, res.stdout[-3000:]
    assert all(c == 0 for c in counts), res.stdout[-3000:]
