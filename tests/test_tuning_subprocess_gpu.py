"""The fault-injection tests need the -DSDT_TUNING library (a muted stream-K partner, a muted chain member, the LDS polluter: hooks that must not
exist in the product library, tests/test_cabi.py::test_product_library_has_no_tuning_switches).  The default GPU suite runs on the PRODUCT
library, where those tests are skipped -- so that "a lost partner is loud"  and "no kernel reads LDS it does not own"
 are nevertheless checked by every `pytest -m gpu`, this test runs them in a child process on the tuning library
(__graft_entry__.build() builds both libraries)."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TUNING = os.path.join(REPO, "speechdrivestemplates_amd", "lib", "libsdt_hip_tuning.so")


def test_fault_injection_tests_pass_on_the_tuning_library():
    from speechdrivestemplates_amd import _lib
    if _lib.has_tuning():
        pytest.skip("this process already runs on the tuning library: the tuning tests run in it")
    if not os.path.exists(TUNING):
        import __graft_entry__ as entry
        if not os.path.exists(entry.HIPCC):
            pytest.skip("tuning library not built and no hipcc here")
        entry.build_tuning()
    env = dict(os.environ, SDT_HIP_LIB=TUNING)
    out = subprocess.run([sys.executable, "-m", "pytest", "tests", "-q", "-m", "gpu and tuning", "-p", "no:cacheprovider"],
                         cwd=REPO, env=env, capture_output=True, text=True, timeout=1800)
    tail = out.stdout[-4000:] + "\n" + out.stderr[-1500:]
    print(tail)
    assert out.returncode == 0, tail
    summary = [ln for ln in out.stdout.splitlines() if re.search(r"\d+ passed", ln)]
    assert summary, tail
    m = re.search(r"(\d+) passed", summary[-1])
    # stream-K lost partner (every persistent forward kernel family), chain lost member, the LDS-residue configurations
    assert int(m.group(1)) >= 9 and "skipped" not in summary[-1] and "failed" not in summary[-1], summary[-1]
