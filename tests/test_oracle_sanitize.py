"""CPU suite: the oracle itself under AddressSanitizer + UndefinedBehaviorSanitizer.

Every parity test trusts oracle/hf_oracle.c; oracle/selftest.c drives all of its entry points
over seeded streams and tables (round trips, Kraft sums, container packing) in a build with
-fsanitize=address,undefined, where any out-of-bounds access or undefined operation aborts."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_selftest_under_sanitizers():
    if shutil.which("gcc") is None and shutil.which("cc") is None:
        pytest.skip("no C compiler")
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "sanitize"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "oracle selftest ok" in r.stdout


def test_host_side_of_the_c_abi_under_sanitizers():
    # hf_api.hip's host code (argument checks, host arithmetic, context handling) compiled with
    # -Xarch_host -fsanitize=address,undefined and driven by huffman_amd/csrc/abi_selftest.c;
    # device code is untouched and no GPU is needed
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "huffman_amd", "csrc"), "sanitize-host"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "abi selftest ok" in r.stdout
