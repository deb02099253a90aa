"""The Maze geometry of the HIP kernels (recovery_rl_amd/csrc/maze_device.hpp: contact predicate + the bracketed search
that replaces the 64-sub-step scan) compiled for the HOST and run against the oracle's sequential scan on millions of
random and adversarial (position, action) pairs -- bit-for-bit, no GPU needed.  The GPU tests repeat the comparison
through the kernels themselves."""
import os
import shutil
import subprocess

import pytest

from oracle import c_oracle as co

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.skipif(not (os.path.exists(HIPCC) or shutil.which("hipcc")), reason="hipcc not installed")
def test_kernel_geometry_equals_the_oracle_scan(tmp_path):
    co.build()
    exe = str(tmp_path / "maze_geometry_host")
    oracle_dir = os.path.join(ROOT, "oracle")
    subprocess.check_call([HIPCC if os.path.exists(HIPCC) else "hipcc", "-O2", "-std=c++17", "-ffp-contract=off",
                           "--offload-arch=gfx950", "-I", os.path.join(ROOT, "recovery_rl_amd", "csrc"),
                           "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "host", "maze_geometry_host.hip"),
                           "-o", exe, "-L", oracle_dir, "-lrrl_oracle", "-Wl,-rpath," + oracle_dir])
    out = subprocess.run([exe, "3000000", "11"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-2000:]
    assert out.stdout.startswith("mismatches 0 of 3000000"), out.stdout
    hits = int(out.stdout.split("ran into something: ")[1].split(",")[0])
    assert hits > 100000                                 # the collision branch is what is being compared
