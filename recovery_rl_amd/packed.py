"""S independent learners on ONE MI355X, sharing every launch of the lock-step iteration (`--seeds_per_gpu S`).

The reference's unit of parallelism is the seed loop (scripts/navigation1.sh:4-8: ten runs, seeds 1..10, one after the
other).  One launch of the lock-step iteration occupies <= 64 of the 256 CUs and mostly waits on memory round trips, and
S graphs replayed on S streams overlap to 1.6x at most (the command processor, not the CUs, is the limit:
profiles/seed_pack_probe.py).  So the seeds are packed INSIDE the launches: every seed is an ordinary `VectorLoop` (own envs,
replay rings, networks, optimiser state, Philox keys and device-side ticks); one iteration of each is recorded on a launch
tape (fast_update.set_tape) -- the argument blocks of the steady-state iteration never change -- and launch k of all tapes
goes out as ONE rrl_*_packed call in which seed s runs its stand-alone code on its own workgroups.  The packed iteration
is captured in one hipGraph.  Every seed's state after K packed iterations equals its solo run's bit for bit
(tests/test_packed_gpu.py).
"""
import ctypes as C
import os

import torch

from . import _lib
from . import fast_update
from .fast_update import FLAT_NETS


class PackedLoop:
    MAX_SEEDS = 16
    # seeds up to which the library issues a head + hidden backward as ONE launch (csrc pack_pair_block_max_seeds)
    PAIR_MAX_SEEDS = int(os.environ.get("RRL_PACK_PAIR_BLOCK_MAX_SEEDS", "8"))

    FRAG_MAX_SEEDS = 8

    def __init__(self, loops, online_qrisk=True):
        """loops: VectorLoops in steady state (past start_steps, batch available), each on the fused grouped path."""
        if not 1 <= len(loops) <= self.MAX_SEEDS:
            raise ValueError("1 .. %d seeds per GPU" % self.MAX_SEEDS)
        self.loops = list(loops)
        self.S = len(loops)
        self.online_qrisk = online_qrisk
        self.lib = _lib.load()
        self.graph = None
        self.tapes = None
        self.stages = None
        # the fragment-order copy of W2 (FlatNet.w2p) pays up to 8 seeds per GPU: -6.5 / -17 / -10 us per packed iteration at
        # 1 / 4 / 8 seeds; with more, the optimiser launches are bound by L2 / Infinity-Cache bandwidth and the two extra
        # streams they write cost what the forwards gain (16 seeds x 16 updates: 6.58-6.75 against 6.55-6.59 ms)
        # (switched off on the nets while this packed loop lives -- the recorded descriptors then carry no copy -- and handed
        # back by close(), so that a loop used solo afterwards has its copy again)
        self._w2p_taken = []
        if self.S > self.FRAG_MAX_SEEDS:
            for loop in self.loops:
                fast = getattr(loop.agent, "fast", None)
                for name in FLAT_NETS:
                    net = getattr(fast, name, None)
                    if net is not None and net.w2p is not None:
                        self._w2p_taken.append((net, net.w2p))
                        net.w2p = None

    # -- recording -------------------------------------------------------------------------------------------------------
    def record(self):
        """One REAL iteration of every seed (launched as usual, so every seed advances by one iteration), teed onto a tape."""
        tapes = []
        for loop in self.loops:
            tape = []
            fast_update.set_tape(tape)
            try:
                loop.vector_step(True, False, self.online_qrisk)
            finally:
                fast_update.set_tape(None)
            bad = [op for op in tape if op[0] == "unsupported"]
            if bad:
                raise _lib.RRLError("seed packing needs the grouped fused path (hidden_size 256-style stacks, first layer "
                                    "fused into the hidden-layer backward): %s was launched" % bad[0][1])
            tapes.append(tape)
        kinds = [tuple(op[0] for op in t) for t in tapes]
        if any(k != kinds[0] for k in kinds):
            raise _lib.RRLError("the seeds' iterations differ in structure: %r" % (kinds,))
        self.tapes = tapes
        self.stages = self._build_stages()
        return len(kinds[0])

    def _build_stages(self):
        S, p = self.S, C.POINTER
        stages = []
        for j, kind in enumerate(op[0] for op in self.tapes[0]):
            ops = [t[j] for t in self.tapes]
            if kind == "sample":
                args = (_lib.rrl_sample_args_t * S)(*[op[1] for op in ops])
                stages.append((self.lib.rrl_sample_multi_packed, (S, args), ops))
            elif kind in ("forward", "head_bwd", "hidden_bwd"):
                typ = {"forward": _lib.rrl_stack_t, "head_bwd": _lib.rrl_head_bwd_t, "hidden_bwd": _lib.rrl_hidden_bwd_t}[kind]
                fn = {"forward": self.lib.rrl_mlp3_forward_multi_packed,
                      "head_bwd": self.lib.rrl_mlp_head_backward_multi_packed,
                      "hidden_bwd": self.lib.rrl_mlp_hidden_backward_multi_packed}[kind]
                n = (C.c_int * S)(*[op[2] for op in ops])
                members = (p(typ) * S)(*[C.cast(op[1], p(typ)) for op in ops])
                stages.append((fn, (S, n, members), ops))
            elif kind == "pair_bwd":
                # head + hidden backward of every seed's stacks: rrl_mlp_backward_pair_multi for S seeds (one launch for the
                # critic-loss kinds, the two packed launches otherwise)
                n = (C.c_int * S)(*[op[3] for op in ops])
                heads = (p(_lib.rrl_head_bwd_t) * S)(*[C.cast(op[1], p(_lib.rrl_head_bwd_t)) for op in ops])
                hidden = (p(_lib.rrl_hidden_bwd_t) * S)(*[C.cast(op[2], p(_lib.rrl_hidden_bwd_t)) for op in ops])
                stages.append((self.lib.rrl_mlp_backward_pair_multi_packed, (S, n, heads, hidden), ops))
            elif kind == "adam":
                b1, b2, eps = ops[0][4], ops[0][5], ops[0][6]
                assert all(op[4:7] == (b1, b2, eps) for op in ops)
                n = (C.c_int * S)(*[op[2] for op in ops])
                segs = (p(_lib.rrl_adam_seg_t) * S)(*[C.cast(op[1], p(_lib.rrl_adam_seg_t)) for op in ops])
                lr = (C.c_float * S)(*[op[3] for op in ops])
                stages.append((self.lib.rrl_adam_step_multi_packed, (S, n, segs, lr, b1, b2, eps), ops))
            elif kind == "step":
                env_name, env_kind = ops[0][1], ops[0][2]
                assert all(op[1] == env_name and op[2] == env_kind for op in ops), "one env per packed run"
                args = (_lib.rrl_step_push_t * S)(*[op[3] for op in ops])
                if env_name == "maze":
                    stages.append((self.lib.rrl_maze_step_push_packed, (S, args), ops))
                else:
                    stages.append((self.lib.rrl_nav_step_push_packed, (S, env_kind, args), ops))
            elif kind == "call":
                # launches that stay per seed (the episode table's append): issued one after the other inside the same graph
                calls = [(op[1], op[2]) for op in ops]
                stages.append((self._call_each, (calls,), ops))
            else:
                raise _lib.RRLError("launch kind %r cannot be packed" % kind)
        return stages

    @property
    def launches(self):
        """Kernel launches of one packed iteration: a head + hidden backward stage is ONE launch when its stacks share a loss
        class and there are at most PAIR_MAX_SEEDS seeds (rrl_mlp_backward_pair_multi_packed: tile form up to two seeds,
        block form beyond), two otherwise; per-seed calls count once per seed."""
        total = 0
        for fn, args, ops in self.stages:
            if ops[0][0] == "pair_bwd":
                kinds = {op[1][k].loss.kind for op in ops for k in range(op[3])}
                outputs = {1 if kind <= _lib.LOSS_QRISK_POLICY else kind for kind in kinds}     # one kernel per output count
                total += 1 if min(kinds) >= 0 and len(outputs) == 1 and self.S <= self.PAIR_MAX_SEEDS else 2
            elif ops[0][0] == "call":
                total += len(ops)
            else:
                total += 1
        return total

    @staticmethod
    def _call_each(calls, stream):
        for fn, args in calls:
            rc = fn(*args, stream)
            if rc:
                return rc
        return 0

    # -- running ---------------------------------------------------------------------------------------------------------
    def launch(self):
        """One packed iteration: every recorded launch once, for all seeds."""
        st = _lib.current_stream()
        for fn, args, _ in self.stages:
            _lib.check(fn(*args, st), getattr(fn, "__name__", "packed stage"))

    def _advance_host_mirrors(self, iterations=1):
        for loop in self.loops:
            cfg = loop.cfg
            loop.total_numsteps += loop.n * iterations
            loop.host_updates[0] += cfg.updates_per_step * iterations
            loop.updates += cfg.updates_per_step * iterations
            if self.online_qrisk:
                loop.host_updates[1] += cfg.updates_per_step * iterations
                loop.agent.safety_critic.updates += cfg.updates_per_step * iterations
            from .experiment import uses_constraint_buffer
            for mem, rows in ((loop.memory, loop.n), (loop.recovery_memory, loop.n if uses_constraint_buffer(cfg) else 0)):
                mem._len = min(mem._len + rows * iterations, mem.capacity)

    def step(self):
        """Eager packed iteration."""
        self.launch()
        self._advance_host_mirrors()

    def capture(self, warmup=2, settle=2, around=None, iters=None):
        """Record (one real iteration per seed), run `warmup` eager packed iterations (they populate the library's
        argument-block cache, so the captured launches copy nothing), capture the packed iteration in ONE hipGraph.
        `around` = (before, after): called around every one of these real iterations (the per-step info stream).
        Returns the number of iterations every seed has advanced."""
        def real(fn):
            if around is not None:
                around[0]()
            fn()
            if around is not None:
                around[1]()

        def settle_once():
            for loop in self.loops:
                loop.vector_step(True, False, self.online_qrisk)
        for _ in range(settle):              # the first iterations size the noise buffer and build the acting workspace
            real(settle_once)
        real(self.record)
        dev = self.loops[0].device
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(max(warmup, 1)):
                real(self.step)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self.launch()
        self.graph = g
        # ... and several iterations as one graph (VectorLoop.capture: the queue idles ~2.7 us between two graphs)
        self.graph_many = None
        self.graph_many_iters = max(1, int(iters if iters is not None else getattr(self.loops[0].cfg, "graph_iterations", 4)))
        if self.graph_many_iters > 1:
            self.graph_many = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph_many):
                for _ in range(self.graph_many_iters):
                    self.launch()
        return settle + 1 + max(warmup, 1)

    def replay(self):
        self.graph.replay()
        self._advance_host_mirrors()

    def advance(self, iterations):
        """`iterations` packed iterations: the many-iteration graph while that many remain, single ones for the rest."""
        k = self.graph_many_iters if self.graph_many is not None else 0
        while k > 1 and iterations >= k:
            self.graph_many.replay()
            self._advance_host_mirrors(k)
            iterations -= k
        for _ in range(iterations):
            self.replay()

    def close(self):
        """Drop the captured graph and free the library's cached argument blocks of the packed launches (device memory):
        call when this was the last packed loop of the process (the cache is shared by all of them)."""
        self.graph = self.graph_many = None
        torch.cuda.synchronize(self.loops[0].device)
        for net, w2p in self._w2p_taken:       # stale by now: every eager forward re-makes it (FlatNet.w2_packed), and a
            net.w2p = w2p                      # graph captured from here on starts from an eager iteration's copy
        self._w2p_taken = []
        return self.lib.rrl_pack_clear()

    def read_stats(self):
        return [loop.read_stats() for loop in self.loops]
