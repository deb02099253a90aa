"""GPU half of the whole-line loads + ds_bpermute restage (csrc/mlp_kernels.hip: -DRRL_COALESCE_W2=2 / 3 for W2 in the fused
forward, -DRRL_COALESCE_DIRECT=1 for the k-contiguous operand of the backward's GEMM tiles): the experimental libraries of `_lib.VARIANTS` against the default library on the same seeded work, each in its own process
(tests/w2_permute_probe.py under RRL_HIP_LIB).  The restage moves the same values into the same registers, so EVERYTHING must
be equal bit for bit: activations and outputs of the forward at every shape, and after 700 graph replays of the headline
iteration every network parameter, env position, replay cursor and counter.

Status: lane arithmetic and compiled code are checked on the CPU (tests/test_w2_permute_cpu.py); the torch-free harness
profiles/w2perm_check.cpp showed every side library equal to the default one call by call on the MI355X; this file ran there at
the end of round 3 (profiles/round3_w2perm/pytest_iteration_level.txt: reproducible, bit-identical, 0.1978 -> 0.2033 ms per
iteration -- the restage is NOT faster, DESIGN 11, and stays opt-in).  The timings of each run are in the warnings summary.
"""
import json
import os
import subprocess
import sys
import warnings

import pytest
import torch

from recovery_rl_amd import _lib

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
_default = {}


def run_probe(lib, prefix):
    env = dict(os.environ)
    env.pop("RRL_HIP_LIB", None)
    if lib:
        env["RRL_HIP_LIB"] = lib
    r = subprocess.run([sys.executable, os.path.join(HERE, "w2_permute_probe.py"), prefix], env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    with open(prefix + ".json") as f:
        info = json.load(f)
    return torch.load(prefix + ".pt"), info


def default_run(tmp_path_factory):
    if "run" not in _default:
        _default["run"] = run_probe("", str(tmp_path_factory.mktemp("w2perm") / "default"))
    return _default["run"]


def test_default_library_probe_is_reproducible(tmp_path_factory, tmp_path):
    """the comparison below means something only if two runs of ONE library agree bit for bit"""
    first, _ = default_run(tmp_path_factory)
    again, _ = run_probe("", str(tmp_path / "again"))
    assert first.keys() == again.keys()
    for k in first:
        assert torch.equal(first[k], again[k]), k


# the library with every flag on (a superset of the two others, which the harness has covered call by call): keeps the
# driver-run suite short -- three probe processes of ~20 s
@pytest.mark.parametrize("name", ["w2perm_bwd"])
def test_variant_equals_the_default_library_bit_for_bit(name, tmp_path_factory, tmp_path):
    path = _lib.variant_path(name)
    assert os.path.exists(path), "variant library not built: python -c 'import __graft_entry__ as g; g.build()'"
    base, base_info = default_run(tmp_path_factory)
    got, info = run_probe(path, str(tmp_path / name))
    assert os.path.samefile(info["library"], path) and not os.path.samefile(base_info["library"], path)
    report = {k: (round(base_info[k], 4), round(info[k], 4)) for k in base_info if k.endswith(("_us", "ms_per_iteration"))}
    different = [k for k in base if not torch.equal(base[k], got[k])]
    warnings.warn("w2 restage by ds_bpermute, library %s: (default, variant) %s; differing results: %s"
                  % (name, json.dumps(report), different or "none"))
    assert base.keys() == got.keys() and not different, different
