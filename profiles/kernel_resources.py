"""Resources of every kernel of the built library, read from its gfx950 code objects (no GPU needed): VGPRs (the unified count of the code
object's metadata, accumulation registers included -> waves per SIMD: 512 / VGPRs in granules of 8, at most 8), static LDS bytes (dynamic LDS is granted at launch: see the `split_lds_floats`
/ `BlkLds` constants of csrc/mlp_kernels.hip), scratch bytes per lane, instruction count and the counts of the instruction
kinds the timings rest on.      python profiles/kernel_resources.py [library.so] > profiles/roundN_kernel_resources.txt"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from recovery_rl_amd import _lib  # noqa: E402
from test_w2_permute_cpu import LLVM, kernel_table  # noqa: E402


def demangle(names):
    import shutil
    filt = os.path.join(LLVM, "llvm-cxxfilt")
    if not os.path.exists(filt):
        filt = shutil.which("c++filt")
    if not filt:
        return {n: n for n in names}
    out = subprocess.run([filt], input="\n".join(names), capture_output=True, text=True, check=True).stdout.splitlines()
    return dict(zip(names, out))


def waves_per_simd(vgpr):
    return min(8, 512 // (8 * ((vgpr + 7) // 8))) if vgpr else 8


def main():
    so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(_lib.CSRC, "librrl_hip.so")
    with tempfile.TemporaryDirectory() as tmp:
        table = kernel_table(so, tmp)
    table = {k: v for k, v in table.items() if "vgpr" in v}          # kernels (device functions have no metadata entry)
    names = demangle(sorted(table))
    print("# %s: %d kernels" % (os.path.basename(so), len(table)))
    print("# %-5s %-6s %-7s %-8s %-6s %-5s %-6s %-6s %-6s %-5s  kernel" %
          ("VGPR", "waves", "LDS", "scratch", "insns", "MFMA", "gload", "gstore", "ds", "flat"))
    for k in sorted(table, key=lambda n: names[n]):
        v, ins = table[k], table[k]["ins"]
        count = lambda *prefix: sum(n for op, n in ins.items() if op.startswith(prefix))
        short = names[k].replace("(anonymous namespace)::", "")
        short = short.split("(")[0] if len(short) > 110 else short
        print("  %-5d %-6d %-7d %-8d %-6d %-5d %-6d %-6d %-6d %-5d  %s" %
              (v["vgpr"], waves_per_simd(v["vgpr"]), v["lds"], v["scratch"], sum(ins.values()), count("v_mfma"),
               count("global_load"), count("global_store", "global_atomic"), count("ds_"), count("flat_"), short[:150]))


if __name__ == "__main__":
    main()
