"""conv_s1d.hip's forward kernel beyond its shipped range (2-d planes, 3 x 3 taps, nine exact limb products; FSC_S1D_GENERAL=1):
kept correct although the planner does not route to it by default.  Reference conv: networks/classifiers.py:37-69."""
import json, os, subprocess, sys
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_general_route_is_fp32_accurate():
    env = dict(os.environ, FSC_S1D_GENERAL="1")
    r = subprocess.run([sys.executable, os.path.join(HERE, "s1d_general_worker.py")], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    rows = json.loads(r.stdout.strip().splitlines()[-1])
    assert len(rows) == 6
    for row in rows:
        assert row["fwd"].startswith("conv_s1d_fwd_kernel") and row["dgrad"].startswith("conv_s1d_fwd_kernel"), row
        nine = row["arith"] == "bf16x9"
        assert row["fwd"].endswith(",9>") == nine, row
        # nine products of exact limbs: fp32 accumulation error only (K <= 6831 terms); bf16: against fp64 on the ROUNDED operands
        tol = 2e-6 if nine else 1e-5
        assert row["fwd_err"] < tol and row["dgrad_err"] < tol, row


def test_default_process_does_not_take_the_general_route():
    from freesound_classification_amd import functional as F
    assert os.environ.get("FSC_S1D_GENERAL", "0") != "1"
    assert F.plan_name(F._desc(128, 759, 759, 2, 6, 3, 3, 9), 0).startswith("conv_fwd_x3_kernel")
    assert F.plan_name(F._desc(16, 100, 70, 5, 9, 3, 3, 1), 0).startswith("conv_fwd_x3_kernel")
    assert F.plan_name(F._desc(128, 476, 476, 1, 3, 1, 3, 1), 0).startswith("conv_s1d_fwd_kernel")
