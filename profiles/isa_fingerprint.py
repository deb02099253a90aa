"""Per-kernel fingerprint of the compiled gfx950 code of an object file / shared library: sha1 over the disassembled
instructions (mnemonic + operands, addresses stripped).  Used to show that a source clean-up left the default build's
kernels instruction-identical:

    python profiles/isa_fingerprint.py recovery_rl_amd/csrc/_build/mlp_kernels.o > before.json
    ... edit ...
    python profiles/isa_fingerprint.py recovery_rl_amd/csrc/_build/mlp_kernels.o [more.o] before.json   # the differences
"""
import glob
import hashlib
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def fingerprint(path):
    d = tempfile.mkdtemp()
    so = shutil.copy(path, d)
    subprocess.run([OBJDUMP, "--offloading", so], cwd=d, check=True, capture_output=True)
    out = {}
    for co in sorted(glob.glob(os.path.join(d, "*gfx950*"))):
        text = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", co], check=True, capture_output=True, text=True).stdout
        name, h, hm, hv, n = None, None, None, None, 0
        for line in text.splitlines():
            m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
            if m:
                if name:
                    out[name] = [h.hexdigest()[:16], n, hm.hexdigest()[:16], hv.hexdigest()[:16]]
                name, h, hm, hv, n = m.group(1), hashlib.sha1(), hashlib.sha1(), hashlib.sha1(), 0
            elif name and line.startswith("\t"):
                ins = line.split("//")[0].strip()
                h.update(ins.encode() + b"\n")
                hm.update(ins.split()[0].encode() + b"\n")      # mnemonics only: blind to kernarg / struct offsets
                if not ins.startswith("s_"):                      # vector / LDS / memory / MFMA stream, operands included
                    hv.update(re.sub(r"\bs\d+\b|s\[\d+:\d+\]", "s", ins).encode() + b"\n")
                n += 1
        if name:
            out[name] = [h.hexdigest()[:16], n, hm.hexdigest()[:16], hv.hexdigest()[:16]]
    shutil.rmtree(d)
    return out


if __name__ == "__main__":
    objs = [a for a in sys.argv[1:] if not a.endswith(".json")]
    ref = [a for a in sys.argv[1:] if a.endswith(".json")]
    now = {}
    for o in objs:
        now.update(fingerprint(o))
    if not ref:
        print(json.dumps(now, indent=0, sort_keys=True))
    else:
        before = json.load(open(ref[0]))
        gone = sorted(set(before) - set(now))
        new = sorted(set(now) - set(before))
        vec = sorted(k for k in now if k in before and before[k][2] != now[k][2] and before[k][3] == now[k][3])
        changed = sorted(k for k in now if k in before and before[k][2] != now[k][2] and before[k][3] != now[k][3])
        offsets = sorted(k for k in now if k in before and before[k][2] == now[k][2] and before[k] != now[k])
        same = len(set(now) & set(before)) - len(changed) - len(offsets) - len(vec)
        print("kernels: %d before, %d now; %d identical, %d with the same instruction sequence but other operands "
              "(argument-block offsets)" % (len(before), len(now), same, len(offsets)))
        for k in offsets:
            print("  operands", k, before[k][:2], "->", now[k][:2])
        for k in vec:
            print("  scalar preamble differs, vector / LDS / memory / MFMA instruction stream identical:", k, before[k][1], "->", now[k][1])
        for k in gone:
            print("  gone   ", k, before[k])
        for k in new:
            print("  new    ", k, now[k])
        for k in changed:
            print("  CHANGED", k, before[k], "->", now[k])
        sys.exit(1 if changed else 0)
