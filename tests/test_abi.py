"""CPU suite: the C-ABI library builds for gfx950, loads, and exports every symbol that
include/rrl_hip.h declares.  No compute calls (no GPU here)."""
import ctypes
import os
import re

from recovery_rl_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "rrl_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rrl_[a-z0-9_]+)\s*\(", src)))


def test_library_builds_and_exports_every_declared_symbol():
    _lib.build()
    assert os.path.exists(_lib.SO_PATH)
    lib = ctypes.CDLL(_lib.SO_PATH)
    names = _declared()
    assert len(names) >= 10
    for name in names:
        assert hasattr(lib, name), name
    assert sorted(_lib.EXPORTS) == names


def test_loader_declares_signatures_and_abi_version():
    lib = _lib.load()
    assert lib.rrl_abi_version() >= 1
    assert lib.rrl_nav_offline_rollouts(0, 20000) == 2000
    assert lib.rrl_nav_offline_rollouts(1, 20000) == 666 + 4 * 500
    assert lib.rrl_nav_offline_rollouts(7, 10) < 0            # unknown env kind -> error code


def test_argument_validation_without_gpu():
    lib = _lib.load()
    # null pointers / bad kinds are rejected before any launch
    assert lib.rrl_nav_step(9, 4, None, None, None, 0, 0, None, 0, None, None, None, None, None, None,
                            None, None, 100, 0, None) == -1
    assert lib.rrl_nav_step(0, 4, None, None, None, 0, 0, None, 0, None, None, None, None, None, None,
                            None, None, 100, 0, None) == -1
    assert lib.rrl_counter_add(None, 1, None) == -1


def test_product_has_no_cpu_fallback():
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from recovery_rl_amd.env import make_vec_env
    with pytest.raises(_lib.RRLError):
        make_vec_env("navigation1", 4, device="cuda")
    with pytest.raises(_lib.RRLError):
        make_vec_env("navigation1", 4, device="cpu")


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "recovery_rl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in txt.lower() or f in (), (dirpath, f)


def test_header_is_plain_c(tmp_path):
    """include/rrl_hip.h is the C ABI a binding includes: it must compile as C99 on its own."""
    import subprocess
    src = tmp_path / "t.c"
    src.write_text('#include "rrl_hip.h"\nint main(void) { return rrl_abi_version() < 0; }\n')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-I", os.path.join(root, "include"),
                           str(src)])


def test_pos_cnt_length_macro_matches_the_python_allocation(tmp_path):
    """RRL_POS_CNT_LEN(cap) of the header (what a binding allocates for rrl_replay_t.pos_cnt: chunk counts, super-chunk
    counts, chunk masks) equals the length replay_memory.py allocates, for aligned and ragged capacities."""
    import subprocess
    caps = [1, 63, 64, 65, 1000, 1023, 1024, 1025, 4096, 5000, 70000, 1000000, 1 << 21]
    src = tmp_path / "len.c"
    src.write_text('#include <stdio.h>\n#include "rrl_hip.h"\nint main(void) {\n' +
                   "".join('  printf("%%lld\\n", (long long)RRL_POS_CNT_LEN(%dLL));\n' % c for c in caps) +
                   "  return 0;\n}\n")
    exe = tmp_path / "len"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o",
                           str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    want = []
    for cap in caps:
        n_chunks = (cap + 63) // 64
        want.append((((n_chunks + 3) // 4) * 4 + (cap + 1023) // 1024 + 1) // 2 * 2 + 2 * n_chunks)
    assert got == want
    src_py = open(os.path.join(ROOT, "recovery_rl_amd", "replay_memory.py")).read()
    assert "(((n_chunks + 3) // 4) * 4 + (cap + 1023) // 1024 + 1) // 2 * 2 + 2 * n_chunks" in src_py
