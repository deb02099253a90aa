"""Compact per-env state of the lock-step loop: ONE u16 status word per env (step count in bits 0-11, done / constraint /
success / ep_done of the last step in bits 12-15) instead of the i32 step count and the four u8 flag arrays of the
reference-shaped API.  The fused step + push kernel (rrl_*_step_push_x) reads and writes the word; everything else (env.step,
resets, evaluation, checkpoints, tests) uses the arrays.  Exactly one of the two representations is live at a time; the
conversions are three element-wise launches and only happen when the loop switches between the fused compact step and
the array API."""
import torch


class StatusWordMixin:
    """Needs self.t (i32[n]), self._flags (u8[4,n] = done, constraint, success, ep_done), self.device, self.num_envs."""

    def _init_status(self):
        self.status = torch.zeros(self.num_envs, dtype=torch.int16, device=self.device)
        self._status_live = False

    def use_status(self):
        """Make the status word the live representation (encode the arrays if they were live)."""
        if not self._status_live:
            f = self._flags.to(torch.int32)
            word = (self.t & 0xFFF) | (f[0] << 12) | (f[1] << 13) | (f[2] << 14) | (f[3] << 15)
            self.status.copy_(word.to(torch.int16))          # bit 15 wraps into the sign: the same 16 bits
            self._status_live = True
        return self.status

    def refresh_arrays(self):
        """Decode the live status word into t / flags without changing which representation is live."""
        if self._status_live:
            w = self.status.to(torch.int32) & 0xFFFF
            self.t.copy_(w & 0xFFF)
            for k in range(4):
                self._flags[k].copy_(((w >> (12 + k)) & 1).to(torch.uint8))

    def use_arrays(self):
        """Make t / flags the live representation (decode the status word if it was live)."""
        self.refresh_arrays()
        self._status_live = False
