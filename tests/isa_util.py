"""Helpers for the tests that look at the COMPILED gfx950 code inside librrl_hip.so (llvm-objdump / llvm-readelf ship with ROCm;
no GPU needed)."""
import glob
import os
import re
import shutil
import subprocess

import pytest

LLVM = "/opt/rocm/lib/llvm/bin"


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
