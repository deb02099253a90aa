"""Device-resident replay buffers: host-side mirror of recovery_rl/replay_memory.py over
rrl_replay_push / rrl_replay_sample_gather / rrl_creplay_sample_gather.

Same names and call shapes as the reference (`push`, `sample`, `__len__`), batched:
`push` takes N rows per call (one row per env) and `sample` returns five CUDA tensors
`(state[B,2], action[B,2], reward[B], next_state[B,2], mask[B])` instead of numpy arrays.
Rows are float32, 32 B each, structure-of-arrays in HBM (1e6 rows = 32 MB).
"""
import ctypes as C

import torch

from . import _lib


class ReplayMemory:
    """Ring buffer for the SAC task policy (replay_memory.py:11-33)."""

    _WITH_POS_COUNTS = False
    _SEED_SALT = 0  # the two buffers share one seed in the reference (replay_memory.py:16,41)

    def __init__(self, capacity, seed, device="cuda", obs_dim=2, act_dim=2):
        if obs_dim != 2 or act_dim != 2:
            raise ValueError("the HIP replay rows are laid out for 2-D states and actions")
        self.device = _lib.require_gpu(device)
        self.lib = _lib.load()
        self.capacity = int(capacity)
        self.seed = (int(seed) ^ self._SEED_SALT) & 0xFFFFFFFFFFFFFFFF
        cap, dev = self.capacity, self.device
        self.s = torch.zeros(cap, 2, dtype=torch.float32, device=dev)
        self.a = torch.zeros(cap, 2, dtype=torch.float32, device=dev)
        self.r = torch.zeros(cap, dtype=torch.float32, device=dev)
        self.s2 = torch.zeros(cap, 2, dtype=torch.float32, device=dev)
        self.m = torch.zeros(cap, dtype=torch.float32, device=dev)
        self.state = torch.zeros(4, dtype=torch.int64, device=dev)   # position, size, ticket, error
        self.tick = torch.zeros(2, dtype=torch.int64, device=dev)    # sampling RNG tick
        # per-64-slot positive counts, then (from the next multiple of 4) per-1024-slot counts, then (from the next
        # multiple of 2) one 64-bit mask per chunk: RRL_POS_CNT_LEN(cap)
        n_chunks = (cap + 63) // 64
        n_cnt = (((n_chunks + 3) // 4) * 4 + (cap + 1023) // 1024 + 1) // 2 * 2 + 2 * n_chunks
        self.pos_cnt = torch.zeros(n_cnt, dtype=torch.int32, device=dev) if self._WITH_POS_COUNTS else None
        self._desc = _lib.rrl_replay_t(self.s.data_ptr(), self.a.data_ptr(), self.r.data_ptr(),
                                       self.s2.data_ptr(), self.m.data_ptr(), cap,
                                       self.state.data_ptr(),
                                       self.pos_cnt.data_ptr() if self.pos_cnt is not None else None, 0, 0)
        self.pinned = 0
        self._len = 0          # host mirror of `size`; exact unless masked pushes were used
        self._len_exact = True
        self._scratch = None
        self._out = {}

    def pin(self, rows=None):
        """Never overwrite rows [0, rows) (default: everything stored so far): past the last slot the ring continues at
        slot `rows`.  The lock-step loop pins the offline constraint demonstrations (experiment.py:280-286): thousands of
        envs fill the ring in capacity / num_envs iterations, while the one-env reference never wraps within a run and so
        never loses them.  Must be called while the ring has not wrapped."""
        rows = int(self.state[1].item()) if rows is None else int(rows)
        if not 0 <= rows < self.capacity:
            raise ValueError("cannot pin %d of %d rows" % (rows, self.capacity))
        self.pinned = rows
        self._desc.pinned = rows

    def rebuild_pos_cnt(self):
        """Recompute the three regions of `pos_cnt` (per-chunk counts, per-super-chunk counts, per-chunk slot masks;
        RRL_POS_CNT_LEN) from the filled rows' r != 0 -- for checkpoints written with another table layout."""
        if self.pos_cnt is None:
            return
        cap, dev = self.capacity, self.device
        size = int(self.state[1].item())
        n_chunks, n_super = (cap + 63) // 64, (cap + 1023) // 1024
        pos = torch.zeros(n_super * 1024, dtype=torch.int64, device=dev)
        pos[:size] = (self.r[:size] != 0).to(torch.int64)
        per_chunk = pos.view(-1, 64)
        self.pos_cnt.zero_()
        self.pos_cnt[:n_chunks] = per_chunk.sum(1)[:n_chunks].to(torch.int32)
        sb = (n_chunks + 3) // 4 * 4
        self.pos_cnt[sb:sb + n_super] = pos.view(n_super, 1024).sum(1).to(torch.int32)
        mb = (sb + n_super + 1) // 2 * 2
        masks = (per_chunk << torch.arange(64, device=dev)).sum(1)[:n_chunks]     # bit 63 wraps into the sign: intended
        self.pos_cnt[mb:mb + 2 * n_chunks] = masks.contiguous().view(torch.int32)

    # -- push ---------------------------------------------------------------------------------
    def push(self, state, action, reward, next_state, done, valid=None):
        """Append N rows (row i = env i). `done` is the reference's 5th tuple field: the
        bootstrap mask float(not done) (experiment.py:434,439)."""
        n = int(reward.shape[0])
        for x in (state, action, reward, next_state, done):
            assert x.dtype == torch.float32 and x.is_contiguous() and x.device == self.s.device
        scratch = None
        if valid is not None:
            assert valid.dtype == torch.uint8 and valid.is_contiguous()
            need = (n + 1023) // 1024 + 1
            if self._scratch is None or self._scratch.numel() < need:
                self._scratch = torch.zeros(need, dtype=torch.int32, device=self.device)
            scratch = self._scratch
            self._len_exact = False
        rc = self.lib.rrl_replay_push(C.byref(self._desc), n, _lib.ptr(state), _lib.ptr(action),
                                      _lib.ptr(reward), _lib.ptr(next_state), _lib.ptr(done),
                                      _lib.ptr(valid), _lib.ptr(scratch), _lib.current_stream())
        _lib.check(rc, "rrl_replay_push")
        if valid is None:
            self._len = min(self._len + n, self.capacity)

    def __len__(self):
        if not self._len_exact:
            self._len = int(self.state[1].item())
            self._len_exact = True
        return self._len

    @property
    def position(self):
        return int(self.state[0].item())

    def check_error(self):
        """Raise if a sampler flagged an error on the device (one sync)."""
        code = int(self.state[3].item())
        if code == 1:
            raise ValueError("Sample larger than population or is negative")
        if code:
            raise _lib.RRLError("replay sampler error flag %d" % code)

    # -- sample -------------------------------------------------------------------------------
    def _batch(self, B):
        if B not in self._out:
            dev = self.device
            self._out[B] = (torch.empty(B, 2, dtype=torch.float32, device=dev),
                            torch.empty(B, 2, dtype=torch.float32, device=dev),
                            torch.empty(B, dtype=torch.float32, device=dev),
                            torch.empty(B, 2, dtype=torch.float32, device=dev),
                            torch.empty(B, dtype=torch.float32, device=dev),
                            torch.empty(B, dtype=torch.int64, device=dev))
        return self._out[B]

    def sample(self, batch_size, out=None, rows=None):
        """B distinct uniform rows (random.sample semantics, replay_memory.py:27-30).
        Returns persistent batch tensors (overwritten by the next sample of the same size).
        `rows` = (xu, x2u, xpu) [B,4] buffers that additionally receive (s,a), (s',-,-), (s,-,-)."""
        B = int(batch_size)
        if self._len_exact and B > self._len:
            raise ValueError("Sample larger than population or is negative")
        s, a, r, s2, m, idx = out if out is not None else self._batch(B)
        xu, x2u, xpu = rows if rows is not None else (None, None, None)
        rc = self.lib.rrl_replay_sample_gather(C.byref(self._desc), B, self.seed, 0,
                                               _lib.ptr(self.tick), 1, _lib.ptr(s), _lib.ptr(a),
                                               _lib.ptr(r), _lib.ptr(s2), _lib.ptr(m), _lib.ptr(idx),
                                               _lib.ptr(xu), _lib.ptr(x2u), _lib.ptr(xpu),
                                               _lib.current_stream())
        _lib.check(rc, "rrl_replay_sample_gather")
        return s, a, r, s2, m


    def draw_desc(self, batch_size, pos_fraction=None, out=None, rows=None, demo_share=None):
        """The arguments of sample() as an rrl_draw_t for rrl_sample_multi (several draws in one launch) and the batch
        tensors it fills.  Same checks, same tick, same rows as sample()."""
        B = int(batch_size)
        s, a, r, s2, m, idx = out if out is not None else self._batch(B)
        xu, x2u, xpu = rows if rows is not None else (None, None, None)
        if pos_fraction is None:
            if self._len_exact and B > self._len:
                raise ValueError("Sample larger than population or is negative")
            if demo_share:
                stratified, n_pos = _lib.DRAW_DEMO_SHARE, int(B * demo_share)
                n_neg = B - n_pos
            else:
                stratified, n_pos, n_neg = _lib.DRAW_UNIFORM, 0, B
        else:
            stratified, n_pos = _lib.DRAW_STRATIFIED, int(B * pos_fraction)
            n_neg = B - n_pos
        p = _lib.ptr
        d = _lib.rrl_draw_t(C.pointer(self._desc), stratified, n_pos, n_neg, self.seed, 0, p(self.tick), 1, p(s), p(a),
                            p(r), p(s2), p(m), p(idx), p(xu), p(x2u), p(xpu))
        return d, (s, a, r, s2, m)


class ConstraintReplayMemory(ReplayMemory):
    """Ring buffer for the safety critic (replay_memory.py:36-75): `reward` holds the constraint
    indicator and `sample(..., pos_fraction)` stratifies on it."""

    _WITH_POS_COUNTS = True
    _SEED_SALT = 0x9E3779B97F4A7C15  # decorrelate its index stream from the task buffer's

    @property
    def clamp_stratified(self):
        """False (default): a stratified draw that needs more rows of a class than the ring holds is an error, as
        random.sample's ValueError in the reference (replay_memory.py:61-66).  True (the lock-step loop): the short
        class gives every row it has and the other class fills the batch (RRL_REPLAY_CLAMP_STRATIFIED)."""
        return bool(self._desc.flags & _lib.REPLAY_CLAMP_STRATIFIED)

    @clamp_stratified.setter
    def clamp_stratified(self, on):
        self._desc.flags = (self._desc.flags | _lib.REPLAY_CLAMP_STRATIFIED) if on else \
            (self._desc.flags & ~_lib.REPLAY_CLAMP_STRATIFIED)

    def sample(self, batch_size, pos_fraction=None, out=None, rows=None, demo_share=None):
        """`demo_share` (lock-step loop only, ignored with pos_fraction): int(B * demo_share) rows from the pinned
        demonstrations, the rest from the online rows (rrl_replay_sample_gather_split)."""
        if pos_fraction is None and demo_share:
            return self._sample_split(batch_size, demo_share, out, rows)
        if pos_fraction is None:
            return super().sample(batch_size, out=out, rows=rows)
        B = int(batch_size)
        n_pos = int(B * pos_fraction)          # replay_memory.py:56-57
        n_neg = B - n_pos
        s, a, r, s2, m, idx = out if out is not None else self._batch(B)
        xu, x2u, xpu = rows if rows is not None else (None, None, None)
        rc = self.lib.rrl_creplay_sample_gather(C.byref(self._desc), n_pos, n_neg, self.seed, 0,
                                                _lib.ptr(self.tick), 1, _lib.ptr(s), _lib.ptr(a),
                                                _lib.ptr(r), _lib.ptr(s2), _lib.ptr(m),
                                                _lib.ptr(idx), _lib.ptr(xu), _lib.ptr(x2u), _lib.ptr(xpu),
                                                _lib.current_stream())
        _lib.check(rc, "rrl_creplay_sample_gather")
        return s, a, r, s2, m

    def _sample_split(self, batch_size, demo_share, out, rows):
        B = int(batch_size)
        if self._len_exact and B > self._len:
            raise ValueError("Sample larger than population or is negative")
        n_demo = int(B * demo_share)
        s, a, r, s2, m, idx = out if out is not None else self._batch(B)
        xu, x2u, xpu = rows if rows is not None else (None, None, None)
        rc = self.lib.rrl_replay_sample_gather_split(C.byref(self._desc), n_demo, B - n_demo, self.seed, 0,
                                                     _lib.ptr(self.tick), 1, _lib.ptr(s), _lib.ptr(a), _lib.ptr(r),
                                                     _lib.ptr(s2), _lib.ptr(m), _lib.ptr(idx), _lib.ptr(xu),
                                                     _lib.ptr(x2u), _lib.ptr(xpu), _lib.current_stream())
        _lib.check(rc, "rrl_replay_sample_gather_split")
        return s, a, r, s2, m
