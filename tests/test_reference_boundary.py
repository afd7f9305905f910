"""The plugin contract against the reference's REAL base classes and registries (HAVE_REFERENCE = True), in a fresh
interpreter so the by-path mocks of oracle.ref_import cannot interfere: see tests/boundary_checks.py for the checks
(subclassing ModelWrapper / Offline*, _MODEL_MAPPING equality, download(), checkpoint lookup, register() + get_*, the reference's
own CommonDetector.detect / CommonOCR / CommonInpainter callers, exception propagation, and the reference's REAL _infer of the ctd
detector, the default detector and the LaMa inpainter beside the plugins' with the network stubbed identically).  Only where
/root/reference exists."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.isdir("/root/reference/manga_translator"), reason="the reference checkout is only present in the build container")
def test_plugins_against_the_real_reference_boundary(tmp_path):
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "boundary_checks.py")], capture_output=True, text=True, timeout=600,
                         cwd=str(tmp_path), env=env)
    assert out.returncode == 0, out.stdout[-3000:] + "\n" + out.stderr[-3000:]
    assert "ALL 16 BOUNDARY CHECKS PASSED" in out.stdout
    for name in ("ComicTextDetector._infer", "DefaultDetector._infer", "LamaMPEInpainter._infer"):   # the end-to-end stubs ran
        assert f"ok the reference's REAL {name}" in out.stdout, name
