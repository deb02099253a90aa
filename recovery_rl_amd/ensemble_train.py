"""Fused optimiser step of the PETS ensemble: `rrl_ens_train_grad` (gather + forward + loss + backward in one
launch) followed by `rrl_adam_step_multi` over the parameters -- the counterpart of one iteration of the batch
loop of MPC.train (recovery_rl/MPC.py:266-292) with torch.optim.Adam(lr=1e-3) (config/navigation1.py:109).  The
PyTorch step in MPC._train_step stays as the general path (other widths / batch sizes) and the cross-check."""
import ctypes as C

import torch

from . import _lib

# weight-decay terms of the loss (config/navigation1.py:52-59: (0.00025 |W0|^2 + 0.0005 |W1|^2 + 0.0005 |W2|^2 +
# 0.00075 |W3|^2) / 2), applied by the Adam kernel as g += weight_decay * W
DECAY = {"lin0_w": 0.00025, "lin1_w": 0.0005, "lin2_w": 0.0005, "lin3_w": 0.00075}
PARAMS = ("lin0_w", "lin0_b", "lin1_w", "lin1_b", "lin2_w", "lin2_b", "lin3_w", "lin3_b", "max_logvar", "min_logvar")


class FusedEnsembleTrainer:
    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        self.model, self.lr, self.betas, self.eps = model, lr, betas, eps
        self.lib = _lib.load()
        dev = model.lin0_w.device
        self.device = dev
        self.E = int(model.num_nets)
        self.params = [getattr(model, n) for n in PARAMS]
        for p in self.params:
            assert p.dtype == torch.float32 and p.is_contiguous()
        z = lambda p: torch.zeros_like(p.data)
        self.grads = [z(p) for p in self.params]
        # a member's 32 rows are split over two workgroups: rows 16..31 write a second partial gradient (weights and
        # biases only), Adam adds the two
        self.grads2 = [z(p) for p in self.params[:8]]
        self.m, self.v = [z(p) for p in self.params], [z(p) for p in self.params]
        self.steps = [torch.zeros(2, dtype=torch.int64, device=dev) for _ in self.params]    # {t, ticket}
        self.part = torch.zeros(2 * self.E, 4, device=dev)
        self.loss_part = torch.zeros(2 * self.E, device=dev)
        self.scratch = torch.empty(int(self.lib.rrl_ens_scratch_floats(self.E)), device=dev)
        self.loss = torch.zeros(self.E, device=dev)
        self._segs = (_lib.rrl_adam_seg_t * len(self.params))()
        for k, p in enumerate(self.params):
            self._segs[k] = _lib.rrl_adam_seg_t(p.numel(), p.data_ptr(), self.grads[k].data_ptr(), self.m[k].data_ptr(),
                                                self.v[k].data_ptr(), self.steps[k].data_ptr(), None, 0.0,
                                                DECAY.get(PARAMS[k], 0.0),
                                                self.grads2[k].data_ptr() if k < 8 else None)

        # large-batch path (rrl_ens_train_grad_big): gradients land in self.grads only
        self._segs_big = (_lib.rrl_adam_seg_t * len(self.params))()
        for k, p in enumerate(self.params):
            self._segs_big[k] = _lib.rrl_adam_seg_t(p.numel(), p.data_ptr(), self.grads[k].data_ptr(),
                                                    self.m[k].data_ptr(), self.v[k].data_ptr(), self.steps[k].data_ptr(),
                                                    None, 0.0, DECAY.get(PARAMS[k], 0.0), None)
        self._scratch_big = None

    SMALL_BATCH = 32

    @staticmethod
    def supported(model, batch_size):
        """The kernels cover the reference's ensemble shape (4-200-200-200-4): batch <= 32 on the one-launch kernel,
        anything larger on the large-batch kernels."""
        lib = _lib.load()
        if not model.lin0_w.is_cuda:
            return False
        shape = (int(model.in_features), int(model.lin1_w.shape[1]), int(model.out_features))
        if batch_size <= FusedEnsembleTrainer.SMALL_BATCH:
            return bool(lib.rrl_ens_train_supported(shape[0], shape[1], shape[2], int(batch_size)))
        return bool(lib.rrl_ens_train_big_supported(*shape))

    def _desc(self):
        m = self.model
        ptrs = [p.data_ptr() for p in self.params]
        mu = m.inputs_mu.data.reshape(-1).contiguous()
        sigma = m.inputs_sigma.data.reshape(-1).contiguous()
        self._keep = (mu, sigma)
        return _lib.rrl_ens_t(self.E, int(m.in_features), int(m.lin1_w.shape[1]), int(m.out_features), *ptrs,
                              mu.data_ptr(), sigma.data_ptr(), *[g.data_ptr() for g in self.grads],
                              self.part.data_ptr(), *[g.data_ptr() for g in self.grads2], self.loss_part.data_ptr())

    def begin(self, train_in, train_targ):
        """Bind the dataset and the current input statistics (they change with every MPC.train call)."""
        assert train_in.is_contiguous() and train_targ.is_contiguous()
        self._data = (train_in, train_targ)
        self._d = self._desc()

    def gradients(self, idx):
        """idx: int64 [E, 1..32] (rows may be strided views of a wider table).  Fills self.grads + self.grads2 (the two
        row halves; without the weight-decay terms, which Adam adds) and self.loss."""
        assert idx.dtype == torch.int64 and idx.stride(1) == 1 and idx.shape[0] == self.E
        rc = self.lib.rrl_ens_train_grad(C.byref(self._d), int(idx.shape[1]), _lib.ptr(self._data[0]),
                                         _lib.ptr(self._data[1]), _lib.ptr(idx), idx.stride(0), _lib.ptr(self.scratch),
                                         _lib.ptr(self.loss), _lib.current_stream())
        _lib.check(rc, "rrl_ens_train_grad")

    def step(self, idx):
        self.gradients(idx)
        rc = self.lib.rrl_adam_step_multi(len(self.params), self._segs, self.lr, self.betas[0], self.betas[1],
                                          self.eps, _lib.current_stream())
        _lib.check(rc, "rrl_adam_step_multi")

    def _big_scratch(self, batch):
        need = int(self.lib.rrl_ens_big_scratch_floats(self.E, int(batch)))
        if self._scratch_big is None or self._scratch_big.numel() < need:
            self._scratch_big = None        # release first: 3.1 GB at batch 131 072
            self._scratch_big = torch.empty(need, device=self.device)
        return self._scratch_big

    def gradients_big(self, idx):
        """idx: int64 [E, batch], any batch >= 1: the large-batch kernels (rrl_ens_train_grad_big).  Fills self.grads
        (without the weight-decay terms) and self.loss."""
        assert idx.dtype == torch.int64 and idx.stride(1) == 1 and idx.shape[0] == self.E
        rc = self.lib.rrl_ens_train_grad_big(C.byref(self._d), int(idx.shape[1]), _lib.ptr(self._data[0]),
                                             _lib.ptr(self._data[1]), _lib.ptr(idx), idx.stride(0),
                                             _lib.ptr(self._big_scratch(idx.shape[1])), _lib.ptr(self.loss),
                                             _lib.current_stream())
        _lib.check(rc, "rrl_ens_train_grad_big")

    def step_big(self, idx):
        self.gradients_big(idx)
        rc = self.lib.rrl_adam_step_multi(len(self.params), self._segs_big, self.lr, self.betas[0], self.betas[1],
                                          self.eps, _lib.current_stream())
        _lib.check(rc, "rrl_adam_step_multi")

    def epoch(self, idxs, batch_size):
        """All ceil(n / batch_size) steps over the columns of idxs [E, n], launched from one C loop."""
        assert idxs.dtype == torch.int64 and idxs.stride(1) == 1 and idxs.shape[0] == self.E
        if batch_size > self.SMALL_BATCH:
            rc = self.lib.rrl_ens_train_epoch_big(C.byref(self._d), len(self.params), self._segs_big, self.lr,
                                                  self.betas[0], self.betas[1], self.eps, _lib.ptr(self._data[0]),
                                                  _lib.ptr(self._data[1]), _lib.ptr(idxs), idxs.stride(0),
                                                  int(idxs.shape[1]), int(batch_size),
                                                  _lib.ptr(self._big_scratch(min(batch_size, idxs.shape[1]))),
                                                  _lib.ptr(self.loss), _lib.current_stream())
            _lib.check(rc, "rrl_ens_train_epoch_big")
            return
        rc = self.lib.rrl_ens_train_epoch(C.byref(self._d), len(self.params), self._segs, self.lr, self.betas[0],
                                          self.betas[1], self.eps, _lib.ptr(self._data[0]), _lib.ptr(self._data[1]),
                                          _lib.ptr(idxs), idxs.stride(0), int(idxs.shape[1]), int(batch_size),
                                          _lib.ptr(self.scratch), _lib.ptr(self.loss), _lib.current_stream())
        _lib.check(rc, "rrl_ens_train_epoch")

    # -- checkpoint ------------------------------------------------------------------------------
    def state_dict(self):
        return {"m": [t.cpu() for t in self.m], "v": [t.cpu() for t in self.v], "steps": [t.cpu() for t in self.steps]}

    def load_state_dict(self, sd):
        for dst, src in zip(self.m + self.v + self.steps, sd["m"] + sd["v"] + sd["steps"]):
            dst.copy_(src)
