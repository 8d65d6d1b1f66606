"""The drop-in boundary is a C ABI: a plain C program (gcc, -std=c99) includes include/sdt_hip.h, links libsdt_hip.so and calls
entry points.  No GPU needed -- only the argument-validation paths run."""
import os
import shutil
import subprocess

import pytest

from conftest import REPO


@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not available")
def test_plain_c_program_links_and_calls_the_library(tmp_path):
    from speechdrivestemplates_amd import _lib
    _lib.load()  # builds the library if this is a fresh checkout
    libdir = os.path.dirname(_lib.LIB_PATH)
    exe = str(tmp_path / "cabi_smoke")
    rocm_lib = "/opt/rocm/lib"
    cmd = ["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(REPO, "include"), os.path.join(REPO, "tests", "cabi", "cabi_smoke.c"),
           "-o", exe, "-L", libdir, "-lsdt_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath," + rocm_lib, "-L", rocm_lib,
           "-Wl,--allow-shlib-undefined"]
    subprocess.check_call(cmd)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)
    assert "C ABI OK" in out.stdout and "last error:" in out.stdout


def test_product_library_has_no_tuning_switches():
    """The ablation / A-B instantiations of the conv kernels (some compute wrong results by design) are compiled only with
    -DSDT_TUNING into a separate library; the shipped one must not even contain the names of the environment switches."""
    from speechdrivestemplates_amd import _lib
    _lib.load()
    blob = open(_lib.LIB_PATH, "rb").read()
    for name in (b"SDT_CONV_PRIO", b"SDT_CONV_TILE", b"SDT_DW_TILE", b"conv_taps_dma_kernel"):
        assert name not in blob, name
