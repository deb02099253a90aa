"""Properties of the COMPILED gfx950 code that the timings rest on, checked on the code objects inside librrl_hip.so
(no GPU needed: the library is cross-compiled here; `llvm-objdump` ships with ROCm).

* The packed kernels (`*_pack_kernel`, csrc/pack.hpp) copy their argument blocks out of device memory.  A pointer loaded
  from memory is a generic pointer to the compiler, and accesses through it are FLAT instructions: they count against the
  vector-memory AND the LDS counter, so every LDS wait drains the global loads in flight.  `rrl_pack::to_global` passes the
  copied pointers through the global address space; this test keeps it that way (a new pointer field that is not passed
  through it shows up here as flat loads).
* No kernel of the hot path keeps data in scratch memory beyond a few spilled scalars.
"""
import glob
import os
import re
import shutil
import subprocess

import pytest

from recovery_rl_amd import _lib

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


@pytest.fixture(scope="module")
def kernels(tmp_path_factory):
    """{demangled-ish kernel symbol: list of instruction mnemonics} for every kernel of the built library"""
    if not os.path.exists(OBJDUMP):
        pytest.skip("llvm-objdump not in this image")
    if not os.path.exists(_lib.SO_PATH):
        _lib.build()
    d = tmp_path_factory.mktemp("isa")
    so = shutil.copy(_lib.SO_PATH, d)                       # the bundles are extracted next to the input file
    subprocess.run([OBJDUMP, "--offloading", so], cwd=d, check=True, capture_output=True)
    out = {}
    for co in sorted(glob.glob(os.path.join(d, "*gfx950"))):
        text = subprocess.run([OBJDUMP, "-d", co], check=True, capture_output=True, text=True).stdout
        name = None
        for line in text.splitlines():
            m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
            if m:
                name = m.group(1)
                out[name] = []
            elif name and line.startswith("\t"):
                out[name].append(line.split()[0])
    assert out, "no gfx950 code object in the library"
    return out


def test_packed_kernels_use_global_not_flat_memory_instructions(kernels):
    packed = {k: v for k, v in kernels.items() if "pack_kernel" in k}
    # every packed launch of the iteration: draws, forwards, head / hidden backward (tile and block form), Adam, env step
    for piece in ("sample_pack_kernel", "mlp3_fwd_split_pack_kernel", "head_bwd_pack_kernel", "gemm16_pack_kernel",
                  "gemm_block_pack_kernel", "adam_pack_kernel", "step_push_pack_kernel"):
        assert any(piece in k for k in packed), piece
    for name, ins in packed.items():
        flat_loads = sum(i.startswith("flat_load") for i in ins)
        global_loads = sum(i.startswith("global_load") for i in ins)
        assert flat_loads == 0 and global_loads > 0, (name, flat_loads, global_loads)
        # (a handful of flat STORES / atomics through generic LDS-or-global helpers exist in the solo twins as well)
        assert sum(i.startswith(("flat_store", "flat_atomic")) for i in ins) <= 4, name


def test_solo_kernels_of_the_iteration_have_no_flat_loads_either(kernels):
    for piece in ("gemm16_group_kernel", "head_bwd_group_kernel", "backward_pair_kernel", "mlp3_fwd_split_group_kernel",
                  "mlp3_fwd_split_flat_group_kernel", "adam_multi_kernel", "sample_group_kernel", "step_push_kernel"):
        hits = [k for k in kernels if piece in k]
        assert hits, piece
        for k in hits:
            assert sum(i.startswith("flat_load") for i in kernels[k]) <= 3, k


def test_hot_kernels_do_not_live_in_scratch(kernels):
    for name, ins in kernels.items():
        n = sum(i.startswith("scratch_") for i in ins)
        assert n <= 16, (name, n)          # the f16x3 planner spills five scalars around its main loop; nothing else does


# ---- register / LDS budgets the occupancy figures of DESIGN.md rest on (profiles/round3_kernel_resources.txt) -------------
# (unified VGPR count of the code object's metadata, accumulation registers included; 512 per SIMD lane in granules of 8)
BUDGETS = (
    # kernel-name piece, max VGPRs, what the budget buys
    ("mlp3_fwd_split_group_kernelILi1E", 128, "B = 256 forwards: four waves per SIMD"),
    ("mlp3_fwd_split_group_kernelILi2E", 128, "4096-row acting forwards: four 4-wave workgroups per CU"),
    ("mlp3_fwd_split_pack_kernelILi2E", 128, "packed forwards"),
    ("plan_cost_kernelILb0E", 128, "planner f32: four waves per SIMD"),
    ("plan_cost_kernelILb1E", 128, "planner f16x3"),
    ("plan_first_step_kernel", 128, "planner, first step once per distinct row: the rollout kernel's phases, same occupancy"),
    ("mlp3_fwd_split_flat_group_kernel", 128, "acting forward riding with 256-row update forwards (both tile forms in one kernel)"),
    ("ens_big_fwd_bwd_kernel", 128, "large-batch ensemble step"),
    ("step_push_kernelIN12_GLOBAL__N_16NavEnvILi0EEELi0E", 96, "fused env step + pushes + episode table, latency variant (speculated reset)"),
    ("step_push_kernelIN12_GLOBAL__N_16NavEnvILi0EEELi2E", 64, "the same, bandwidth variant: 1024-thread workgroups, eight waves per SIMD"),
    ("backward_pair_kernel", 256, "paired head + hidden backward: two 4-wave workgroups (8 tiles) per CU"),
    ("nav_step_kernel", 64, "env step: eight waves per SIMD (bandwidth regime)"),
    ("sample_group_kernel", 64, "replay draws"),
)


def test_hot_kernels_keep_their_register_budgets(tmp_path):
    from isa_util import kernel_table
    if not os.path.exists(_lib.SO_PATH):
        _lib.build()
    table = {k: v for k, v in kernel_table(_lib.SO_PATH, str(tmp_path)).items() if "vgpr" in v}
    assert len(table) > 100
    for piece, limit, why in BUDGETS:
        hits = {k: v for k, v in table.items() if piece in k}
        assert hits, piece
        for k, v in hits.items():
            assert v["vgpr"] <= limit, (k, v["vgpr"], limit, why)
            assert v["scratch"] <= 32, (k, v["scratch"])


def test_step_push_has_no_loop_and_few_sgpr_spills(tmp_path):
    """The fused env-step kernel covers its envs in ONE pass (step_push.hpp: grid_cover).  Around a grid-stride loop the
    compiler kept every loop-invariant address of its ~35 arrays in scalar registers: 347 SGPRs spilled into vector lanes,
    1 039 v_readlane / v_writelane among 3 899 instructions (profiles/round4_step_push_isa.txt)."""
    from isa_util import kernel_table
    if not os.path.exists(_lib.SO_PATH):
        _lib.build()
    table = kernel_table(_lib.SO_PATH, str(tmp_path))
    hits = {k: v for k, v in table.items() if "step_push_kernelIN12_GLOBAL__N_16NavEnv" in k}
    assert len(hits) == 6, sorted(hits)          # two env kinds x three launch shapes
    for k, v in hits.items():
        lanes = v["ins"].get("v_readlane_b32", 0) + v["ins"].get("v_writelane_b32", 0)
        assert lanes <= 500, (k, lanes)
        assert sum(v["ins"].values()) <= 3300, (k, sum(v["ins"].values()))


def test_solo_group_kernels_fetch_their_argument_block_in_one_batch(kernels):
    """A solo group kernel takes its member from blockIdx.y and pins the member's scalars (mlp_common.hpp: arrive_together):
    ONE batch of scalar loads and one wait before the first vector-memory request -- the flat grid's first[] walk and the
    compiler's one-wait-per-first-use were two to three dependent round trips of ~0.5 us each (profiles/round4_levels.txt)."""
    limits = {"gemm16_group_kernel": 1, "head_bwd_group_kernel": 1, "mlp3_fwd_split_group_kernelILi1E": 1,
              "mlp3_fwd_split_group_kernelILi2E": 1, "adam_multi_kernel": 1,
              # the paired launch: the head's block, then (tile workgroups) the job's block; the policy-head kinds (round 5) read
              # a few scalars of the loss description more
              "backward_pair_kernel": 4}
    for piece, most in limits.items():
        hits = [k for k in kernels if piece in k]
        assert hits, piece
        for k in hits:
            ins = kernels[k]
            first = next(i for i, op in enumerate(ins) if op.startswith(("global_load", "buffer_load")))
            waits = sum(op == "s_waitcnt" for op in ins[:first])           # no vector request is out yet: scalar waits only
            loads = sum(op.startswith("s_load") for op in ins[:first])
            assert waits <= most and loads >= 5, (k, waits, loads)


def test_early_tickets_are_not_waited_for_on_the_spot(kernels):
    """The cursor ticket of step_push_kernel, its episode-table reservation and Adam's step ticket are returning atomics whose
    value is needed late.  LLVM's atomic optimiser rewrites such an atomic as "first lane adds, v_readfirstlane the result" with
    an s_waitcnt right behind the atomic -- the round trip (~0.7 us) is then waited for on the spot.  _lib.SOURCE_FLAGS switches
    the optimiser off for those sources; this test sees it if the flag is lost.  (The flag exists in newer LLVM only: where
    hipcc does not know it, the build drops it -- _lib.flags_supported -- and there is nothing to assert.)"""
    from recovery_rl_amd import _lib
    if not _lib.flags_supported(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), _lib.SOURCE_FLAGS["update_kernels.hip"]):
        pytest.skip("this hipcc does not know -amdgpu-atomic-optimizer-strategy")
    for piece in ("step_push_kernelIN12_GLOBAL__N_16NavEnvILi0EEELi0E", "adam_multi_kernel"):
        hits = [k for k in kernels if piece in k]
        assert hits, piece
        for k in hits:
            ins = kernels[k]
            for i, op in enumerate(ins):
                if op.startswith("global_atomic_add"):
                    tail = ins[i + 1:i + 5]
                    assert not ("s_waitcnt" in tail and "v_readfirstlane_b32" in tail), (k, i, tail)
