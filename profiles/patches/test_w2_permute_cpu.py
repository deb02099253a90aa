"""The whole-line W2 loads + ds_bpermute restage of the fused forward (csrc/mlp_kernels.hip, -DRRL_COALESCE_W2=2 / 3; the
experimental libraries of `_lib.VARIANTS`) checked WITHOUT a GPU:

* the lane arithmetic of the restage, emulated lane by lane: what arrives by whole-line loads ends up in MFMA fragment order,
  every source lane is asked exactly once per pass (the property a ds_bpermute needs);
* the compiled variants: the permutes are there, the loads are as many as before, the kernels keep the register count that
  puts four waves on a SIMD and no LDS beyond the default's, nothing spills;
* the DEFAULT library has none of it (the variants are opt-in until they are measured on the MI355X).

The GPU half (bit-equality with the default library + timing) is tests/test_w2_permute_gpu.py.
"""
import glob
import os
import re
import shutil
import subprocess

import numpy as np
import pytest

from recovery_rl_amd import _lib

LLVM = "/opt/rocm/lib/llvm/bin"


def restage_panel(reg0, reg1):
    """One panel of the restage exactly as the kernel spells it.  reg0 / reg1: [64 lanes, 4] = the two registers a lane holds
    after the whole-line loads.  Returns (frag0, frag1, source lanes of pass A, of pass B)."""
    lane = np.arange(64)
    i, q = lane & 15, lane >> 4
    low_src = (lane & 4) == 0
    low_row = (i & 8) == 0
    addr_a = 4 * (8 * (i & 7) + np.where(low_row, 0, 4) + q)
    addr_b = 4 * (8 * (i & 7) + np.where(low_row, 4, 0) + q)
    send_a = np.where(low_src[:, None], reg0, reg1)
    send_b = np.where(low_src[:, None], reg1, reg0)
    got_a = send_a[addr_a // 4]                      # ds_bpermute_b32: lane l receives the data of lane addr[l] / 4
    got_b = send_b[addr_b // 4]
    frag0 = np.where(low_row[:, None], got_a, got_b)
    frag1 = np.where(low_row[:, None], got_b, got_a)
    return frag0, frag1, addr_a // 4, addr_b // 4


def test_restage_lane_arithmetic_gives_fragment_order():
    H = 256
    W = np.arange(16 * H, dtype=np.int64).reshape(16, H)      # my 16 rows of W2, every element its own id
    lane = np.arange(64)
    i, q = lane & 15, lane >> 4
    for p in range(H // 32):
        # whole-line loads: register c of lane l = row 8 c + (l >> 3), floats 32 p + 4 (l & 7) .. + 3
        regs = [np.stack([W[8 * c + (lane >> 3), 32 * p + 4 * (lane & 7) + t] for t in range(4)], 1) for c in (0, 1)]
        f0, f1, src_a, src_b = restage_panel(*regs)
        for jj, frag in ((0, f0), (1, f1)):
            j = 2 * p + jj
            want = np.stack([W[i, 16 * j + 4 * q + t] for t in range(4)], 1)     # wv[j] of lane (i, q): the MFMA B operand
            assert np.array_equal(frag, want), (p, jj)
        # a permute moves one register per source lane: each pass must ask every lane exactly once
        assert sorted(src_a) == list(range(64)) and sorted(src_b) == list(range(64))


def _code_objects(so, tmp):
    if not os.path.exists(os.path.join(LLVM, "llvm-objdump")):
        pytest.skip("llvm tools not in this image")
    so = shutil.copy(so, tmp)
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", so], cwd=tmp, check=True, capture_output=True)
    return sorted(glob.glob(os.path.join(tmp, "*gfx950")))


def kernel_table(so, tmp):
    """{kernel symbol: {"ins": Counter-like dict of mnemonics, "vgpr": n, "lds": bytes, "scratch": bytes}}"""
    out = {}
    for co in _code_objects(so, tmp):
        text = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", co], check=True, capture_output=True, text=True).stdout
        name = None
        for line in text.splitlines():
            m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
            if m:
                name = m.group(1)
                out[name] = {"ins": {}}
            elif name and line.startswith("\t"):
                op = line.split()[0]
                out[name]["ins"][op] = out[name]["ins"].get(op, 0) + 1
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], check=True, capture_output=True,
                               text=True).stdout
        entry = None
        for line in notes.splitlines():
            m = re.match(r"^\s*(- )?\.(\w+):\s*(\S.*)?$", line)
            if not m:
                continue
            if m.group(1) and m.group(2) == "agpr_count":      # first key of a kernel's entry (keys are sorted)
                entry = {}
            if entry is None:
                continue
            entry[m.group(2)] = m.group(3)
            if m.group(2) == "vgpr_count":                      # last key: the entry is complete
                k = out.get(entry.get("name"))
                if k is not None:
                    k["vgpr"] = int(entry["vgpr_count"])
                    k["lds"] = int(entry["group_segment_fixed_size"])
                    k["scratch"] = int(entry["private_segment_fixed_size"])
                entry = None
    return out


def _need(path):
    if not os.path.exists(path):
        pytest.skip("%s not built (python -c 'import __graft_entry__ as g; g.build()')" % os.path.basename(path))
    return path


@pytest.fixture(scope="module")
def default_table(tmp_path_factory):
    return kernel_table(_need(os.path.join(_lib.CSRC, "librrl_hip.so")), str(tmp_path_factory.mktemp("isa_default")))


def _fwd(table, multi_row):
    """the split-forward kernels with several row tiles per workgroup (acting, 4096 rows) or with one (B = 256 updates)"""
    hits = {}
    for k, v in table.items():
        if "mlp3_fwd_split" not in k:
            continue
        single = "ILi1E" in k
        if "mixed" in k:
            single = False
        if single != multi_row:
            hits[k] = v
    assert hits
    return hits


def test_default_library_has_no_restage_and_four_waves_per_simd(default_table):
    for k, v in {**_fwd(default_table, True), **_fwd(default_table, False)}.items():
        assert v["ins"].get("ds_bpermute_b32", 0) <= 8, k          # the policy head's few lane exchanges
        assert v["scratch"] == 0, k
        if "ILi4E" not in k:                                       # the 4-row-tile packed kernel is an opt-in shape (123 VGPRs)
            assert v["vgpr"] <= 128, (k, v["vgpr"])                # 512 / 128 = four waves per SIMD = four workgroups per CU


@pytest.mark.parametrize("name", sorted(_lib.VARIANTS))
def test_variant_library_restages_by_permute_at_the_default_footprint(name, default_table, tmp_path):
    table = kernel_table(_need(_lib.variant_path(name)), str(tmp_path))
    all_forwards = name in ("w2perm_all", "w2perm_bwd")
    checked = 0
    for multi_row in (True, False):
        restaged = multi_row or all_forwards
        for k, v in _fwd(table, multi_row).items():
            d = default_table[k]
            members = 2 if "mixed" in k else 1                     # the mixed kernel holds a 1-tile and a 2-tile body
            n_perm = v["ins"].get("ds_bpermute_b32", 0) - d["ins"].get("ds_bpermute_b32", 0)
            if restaged:
                want = 64 * (members if all_forwards else 1)
                assert n_perm == want, (k, n_perm)                 # 8 panels x 8 permutes per restaged body
            else:
                assert n_perm == 0, k
            # the same number of 16-byte loads, only their addresses differ; nothing more in LDS, nothing in scratch
            assert v["ins"]["global_load_dwordx4"] == d["ins"]["global_load_dwordx4"], k
            assert v["ins"]["v_mfma_f32_16x16x4_f32"] == d["ins"]["v_mfma_f32_16x16x4_f32"], k
            assert v["lds"] == d["lds"] and v["scratch"] == 0, k
            if "ILi4E" not in k:
                assert v["vgpr"] <= 128, (k, v["vgpr"])
            checked += 1
    assert checked >= 8
    # the GEMM tiles of the backward pass (16 x 16 tile form, block form, the fused hidden + head form): restaged in the
    # `_bwd` library only -- same loads, same MFMAs, no more registers than a handful, same LDS
    gemm_restaged = 0
    for k, v in table.items():
        if "mlp3_fwd_split" in k:
            continue
        d = default_table[k]
        if name == "w2perm_bwd" and v["ins"] != d["ins"]:
            assert any(piece in k for piece in ("gemm16", "gemm_block", "hidden_head", "gemm_kernel")), k
            assert v["ins"].get("ds_bpermute_b32", 0) > d["ins"].get("ds_bpermute_b32", 0), k
            assert v["ins"].get("global_load_dwordx4", 0) == d["ins"].get("global_load_dwordx4", 0), k
            assert v["ins"]["v_mfma_f32_16x16x4_f32"] == d["ins"]["v_mfma_f32_16x16x4_f32"], k
            assert v["lds"] == d["lds"] and v["scratch"] == d["scratch"] and v["vgpr"] <= d["vgpr"] + 4, (k, v["vgpr"], d["vgpr"])
            gemm_restaged += 1
        else:
            assert v["ins"] == d["ins"], k         # every other kernel is the default's, instruction for instruction
    if name == "w2perm_bwd":
        hot = [k for k in table if "gemm16_group_kernel" in k or "gemm16_pack_kernel" in k or "gemm_block_pack_kernel" in k]
        assert gemm_restaged >= 3 and hot
        for k in hot:
            assert table[k]["ins"] != default_table[k]["ins"], k
