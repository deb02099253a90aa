"""Hand-written forward + backward of the SAC / Q_risk updates on the fused HIP kernels
(csrc/mlp_kernels.hip, csrc/update_kernels.hip) -- the `fast path` of `SAC.update_parameters`
(recovery_rl/sac.py:170-277) and `QRiskWrapper.update_parameters` (recovery_rl/qrisk.py:86-163).

Same mathematics as the autograd path in sac.py / qrisk.py (which stays as the general path for
the baseline flags: DGD / RSPO / RCPO / SQRL / automatic entropy tuning / Deterministic policy);
`tests/test_fast_update_gpu.py` checks the two paths against each other and against the
reference KATs.  ~37 launches per update instead of ~150, no vendor GEMM.

Parameters of every network live in ONE flat f32 buffer (the nn.Module parameters are views into
it), twin heads stacked on a leading dimension so both heads run in one batched launch; Adam and
the Polyak target update are one kernel over the flat buffer.
"""
import math

import ctypes as C

import os

import torch

from . import _lib
from .fused import NN, NT, TN, gemm, mlp3_forward, mlp3_supported


# -- launch tape (seed packing, recovery_rl_amd/packed.py) -----------------------------------------------------------
# While a tape is set, every launch of the grouped path is ALSO appended to it: (kind, ctypes payload...).  The argument
# blocks of the steady-state iteration never change, so one recorded iteration of every seed is the launch list of the
# packed iteration: launch k of all seeds goes out as one rrl_*_packed call.
_TAPE = None


def set_tape(tape):
    global _TAPE
    _TAPE = tape


def record(kind, *payload):
    if _TAPE is not None:
        _TAPE.append((kind,) + payload)


# the FlatNets of a FastUpdater (attribute names): what checkpoints save and what seed packing switches
FLAT_NETS = ("critic", "critic_target", "policy", "qrisk", "qrisk_target", "recpolicy")


class FlatNet:
    """Flat parameter / gradient / Adam-state storage for one network, plus the layer views."""

    def __init__(self, named_shapes, device):
        self.device = device
        self.shapes = dict(named_shapes)
        total = sum(int(torch.Size(s).numel()) for _, s in named_shapes)
        self.flat = torch.zeros(total, dtype=torch.float32, device=device)
        self.grad = torch.zeros_like(self.flat)
        self.m = torch.zeros_like(self.flat)
        self.v = torch.zeros_like(self.flat)
        self.step = torch.zeros(2, dtype=torch.int64, device=device)   # {t, ticket}
        self.p, self.g, self.offset = {}, {}, {}
        off = 0
        for name, shape in named_shapes:
            n = int(torch.Size(shape).numel())
            self.p[name] = self.flat[off:off + n].view(shape)
            self.g[name] = self.grad[off:off + n].view(shape)
            self.offset[name] = off
            off += n
        # W2 a second time in the forward kernels' MFMA fragment order (rrl_w2_pack / rrl_stack_t.W2p; hidden width 256):
        # written by the fused optimiser launch together with the parameters (adam_multi), re-made from the row-major
        # values by every EAGER forward (w2_packed()) -- torch code may have written the parameters through the modules'
        # views since, and a permutation of 64 K floats costs less than finding out
        shape = self.shapes.get("W2")
        self.w2p = None
        if shape is not None and len(shape) == 3 and shape[1] == shape[2] == 256 and self.offset["W2"] % 4 == 0 \
                and torch.device(device).type == "cuda" and os.environ.get("RRL_W2_FRAG", "1") != "0":   # (0: A/B runs of profiles/)
            self.w2p = torch.empty(int(torch.Size(shape).numel()), dtype=torch.float32, device=device)

    def w2_packed(self):
        """The fragment-order copy of W2, current: re-made here unless a hipGraph is being captured (a captured iteration is
        replayed with nothing but the library's kernels between its launches, and those keep the copy in step)."""
        if self.w2p is None:
            return None
        if not torch.cuda.is_current_stream_capturing():
            W2 = self.p["W2"]
            _lib.check(_lib.load().rrl_w2_pack(W2.shape[0], W2.shape[1], W2.data_ptr(), self.w2p.data_ptr(),
                                               _lib.current_stream()), "rrl_w2_pack")
        return self.w2p

    def rebind_grad(self, storage):
        """Move the gradient buffer into `storage` (a slice of a bucket shared with other networks, so that one
        collective reduces them together)."""
        assert storage.numel() == self.flat.numel() and storage.is_contiguous()
        self.grad = storage
        off = 0
        for name, shape in self.shapes.items():
            n = int(torch.Size(shape).numel())
            self.g[name] = storage[off:off + n].view(shape)
            off += n

    def adopt(self, name, params):
        """Copy the current values of `params` (list of nn.Parameter, stacked along dim 0 when more
        than one) into the flat buffer and re-point them at it."""
        dst = self.p[name]
        with torch.no_grad():
            if len(params) == 1:
                dst.copy_(params[0].data.reshape(dst.shape))
                params[0].data = dst.view(params[0].shape)
            else:
                for i, prm in enumerate(params):
                    dst[i].copy_(prm.data.reshape(dst[i].shape))
                    prm.data = dst[i].view(prm.shape)

    def adam(self, lr, target=None, tau=0.0, betas=(0.9, 0.999), eps=1e-8, part=None):
        """`part` = (first_part tensor [T, stride], n_first): the gradients of the leading n_first parameters
        (W1, b1) arrive as T row-tile partials (Stack.backward with fuse_first)."""
        packed_copy = self.w2p is not None or (target is not None and target.w2p is not None)
        if part is not None or packed_copy:
            # with a fragment-order W2 copy the step MUST be the launch that keeps the copy current (own parameters and
            # Polyak target): a captured iteration re-makes nothing between its launches (w2_packed), so rrl_adam_step
            # would leave every replayed forward on the W2 of capture time (batch sizes without fuse_first: part = None)
            if part is None:
                record("unsupported", "rrl_adam_step_multi without first-layer partials")
            return adam_multi(lr, [(self, target, tau, part)], betas, eps)
        record("unsupported", "rrl_adam_step")
        lib = _lib.load()
        rc = lib.rrl_adam_step(self.flat.numel(), self.flat.data_ptr(), self.grad.data_ptr(),
                               self.m.data_ptr(), self.v.data_ptr(), self.step.data_ptr(), lr, betas[0],
                               betas[1], eps, None if target is None else target.flat.data_ptr(), tau,
                               _lib.current_stream())
        _lib.check(rc, "rrl_adam_step")


def adam_multi(lr, nets, betas=(0.9, 0.999), eps=1e-8):
    """One rrl_adam_step_multi launch over several FlatNets: nets = [(net, target or None, tau[, part]), ...];
    part = (first_part [T, stride], n_first) or None, see FlatNet.adam."""
    lib = _lib.load()
    segs = (_lib.rrl_adam_seg_t * len(nets))()
    for k, item in enumerate(nets):
        net, target, tau = item[:3]
        part = item[3] if len(item) > 3 else None
        gp, n_part, stride, n_first = (None, 0, 0, 0) if part is None else \
            (part[0].data_ptr(), part[0].shape[0], part[0].stride(0), part[1])
        pack = net.w2p is not None and (target is None or target.w2p is not None)
        if not pack and (net.w2p is not None or (target is not None and target.w2p is not None)):
            raise _lib.RRLError("a network and its Polyak target must both keep the fragment-order W2 copy, or neither")
        segs[k] = _lib.rrl_adam_seg_t(net.flat.numel(), net.flat.data_ptr(), net.grad.data_ptr(), net.m.data_ptr(),
                                      net.v.data_ptr(), net.step.data_ptr(),
                                      None if target is None else target.flat.data_ptr(), tau, 0.0, None,
                                      gp, n_part, stride, n_first,
                                      net.w2p.data_ptr() if pack else None,
                                      target.w2p.data_ptr() if pack and target is not None else None,
                                      net.offset["W2"] if pack else 0, net.p["W2"].shape[0] if pack else 0)
    record("adam", segs, len(nets), float(lr), float(betas[0]), float(betas[1]), float(eps))
    _lib.check(lib.rrl_adam_step_multi(len(nets), segs, lr, betas[0], betas[1], eps, _lib.current_stream()),
               "rrl_adam_step_multi")


def flatten_twin_q(net, device):
    """QNetwork / QNetworkConstraint -> FlatNet with heads stacked: W1 [2,H,din] ... b3 [2,1]."""
    H, din = net.linear1.weight.shape
    shapes = [("W1", (2, H, din)), ("b1", (2, H)), ("W2", (2, H, H)), ("b2", (2, H)),
              ("W3", (2, 1, H)), ("b3", (2, 1))]
    has_bn = hasattr(net, "bn1")
    if has_bn:
        shapes += [("bn_w", (din,)), ("bn_b", (din,))]
    f = FlatNet(shapes, device)
    f.adopt("W1", [net.linear1.weight, net.linear4.weight])
    f.adopt("b1", [net.linear1.bias, net.linear4.bias])
    f.adopt("W2", [net.linear2.weight, net.linear5.weight])
    f.adopt("b2", [net.linear2.bias, net.linear5.bias])
    f.adopt("W3", [net.linear3.weight, net.linear6.weight])
    f.adopt("b3", [net.linear3.bias, net.linear6.bias])
    if has_bn:
        f.adopt("bn_w", [net.bn1.weight])
        f.adopt("bn_b", [net.bn1.bias])
    f.G, f.H, f.din, f.dout = 2, H, din, 1
    return f


def flatten_policy(net, device):
    """GaussianPolicy (head = [mean; log_std], 4 rows) or StochasticPolicy (head = mean, 2 rows +
    log_std[2]) -> FlatNet with a leading head dimension of 1."""
    H, din = net.linear1.weight.shape
    gaussian = hasattr(net, "mean_linear")
    dout = 4 if gaussian else 2
    shapes = [("W1", (1, H, din)), ("b1", (1, H)), ("W2", (1, H, H)), ("b2", (1, H)),
              ("W3", (1, dout, H)), ("b3", (1, dout))]
    if not gaussian:
        shapes.append(("log_std", (2,)))
    f = FlatNet(shapes, device)
    f.adopt("W1", [net.linear1.weight])
    f.adopt("b1", [net.linear1.bias])
    f.adopt("W2", [net.linear2.weight])
    f.adopt("b2", [net.linear2.bias])
    with torch.no_grad():
        if gaussian:
            W3, b3 = f.p["W3"][0], f.p["b3"][0]
            W3[0:2].copy_(net.mean_linear.weight.data)
            W3[2:4].copy_(net.log_std_linear.weight.data)
            b3[0:2].copy_(net.mean_linear.bias.data)
            b3[2:4].copy_(net.log_std_linear.bias.data)
            net.mean_linear.weight.data, net.log_std_linear.weight.data = W3[0:2], W3[2:4]
            net.mean_linear.bias.data, net.log_std_linear.bias.data = b3[0:2], b3[2:4]
        else:
            f.adopt("W3", [net.mean.weight])
            f.adopt("b3", [net.mean.bias])
            f.adopt("log_std", [net.log_std])
    f.G, f.H, f.din, f.dout = 1, H, din, dout
    return f


class Stack:
    """Workspace + forward / backward of one 2-hidden-layer MLP stack at batch size B."""

    def __init__(self, net, B):
        self.net, self.B = net, B
        dev, G, H = net.device, net.G, net.H
        z = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        self.h1, self.h2, self.out = z(G, B, H), z(G, B, H), z(G, B, net.dout)
        self.dh1, self.dh2, self.dx = z(G, B, H), z(G, B, H), z(G, B, net.din)
        # partial last-layer sums of the small-batch forward: rrl_mlp3_is_split = number of parts (0: not split)
        self.nsplit = int(_lib.load().rrl_mlp3_is_split(B, H)) if mlp3_supported(H, net.din, net.dout) else 0
        self.split = self.nsplit > 0
        self.scratch = z(max(self.nsplit, 1), G, B, net.dout)
        self.finalize = False                       # True: always hand back the summed output tensor
        self.pair_hidden = True                     # dW2 and dh1 of the backward in one launch
        self._folded = False
        self._init_first(dev, G, B, H, net.din)

    def _init_first(self, dev, G, B, H, din):
        """First layer of the backward inside the hidden-layer launch (rrl_first_layer_t): the 16 x 16 tiles of dh1 emit
        row-tile partials of (dW1, db1) -- laid out like the head [W1 | b1] of the flat gradient buffer, summed by
        Adam -- and column-tile partials of dx, summed by the policy-head backward.  One launch per stack backward less;
        dh1 never goes to memory."""
        self.fuse_first = B % 128 == 0 and H % 128 == 0 and B // 16 <= 64 and H // 16 <= 16
        # dx partials folded by the producer (rrl_first_layer_t.dx_fold): sums over four consecutive column tiles, H / 64
        # instead of H / 16 partials for the policy-head backward to add up -- in the paired launches (B x 4 outputs of a
        # policy head fit their dOut tile) and the block form of the packed ones; fold_dx = False keeps the tile partials
        # (the one-tile-per-workgroup launches; the consumer then sums them in the same grouped order: the same bits)
        self.fold_dx = self.fuse_first and B <= 256
        self.n_first = G * H * (din + 1)
        self.first_part = self.dx_part = None
        if self.fuse_first:
            self.first_part = torch.zeros(B // 16, self.n_first, dtype=torch.float32, device=dev)
            self.dx_part = torch.zeros(H // 16, G, B, din, dtype=torch.float32, device=dev)

    @property
    def grad_part(self):
        """What FlatNet.adam / adam_multi need to read this stack's (dW1, db1): (partials, count) or None."""
        return (self.first_part, self.n_first) if self.fuse_first else None

    def dx_parts(self):
        """(tensor [G, B, din] view of partial 0, number of partials, partial stride, group) of dL/dx after
        backward(input_grad): group = 4 -> tile partials, summed in groups of four first (rrl_loss_t.da_group)."""
        if self.fuse_first:
            n = self.dx_part.shape[0]
            return (self.dx_part[0], n // 4, self.dx_part.stride(0), 1) if self._folded else \
                (self.dx_part[0], n, self.dx_part.stride(0), 4 if n % 4 == 0 else 1)
        return self.dx, 1, 0, 1

    def forward(self, x, params=None, save=True):
        """x [B, din] shared by all heads.  `params` lets a target network reuse this workspace;
        `save` keeps the hidden activations for backward()."""
        P = (params or self.net).p
        G = self.net.G
        self.x = x
        self.parts = (self.out, 1, 0)       # (tensor, n_part, part_stride): how consumers read the output
        if mlp3_supported(self.net.H, self.net.din, self.net.dout):     # one launch for the whole stack
            record("unsupported", "rrl_mlp3_forward")
            mlp3_forward(x, P["W1"], P["b1"], P["W2"], P["b2"], P["W3"], P["b3"], out=self.out,
                         h1=self.h1 if save else None, h2=self.h2 if save else None, scratch=self.scratch,
                         finalize=self.finalize)
            if self.split and not self.finalize:   # partial last-layer sums: the consumer kernels add them up
                self.parts = (self.scratch, self.nsplit, self.scratch.stride(0))
            return self.parts
        xg = x.unsqueeze(0).expand(G, -1, -1)
        gemm(NT, xg, P["W1"], out=self.h1, bias=P["b1"], relu=True)
        gemm(NT, self.h1, P["W2"], out=self.h2, bias=P["b2"], relu=True)
        gemm(NT, self.h2, P["W3"], out=self.out, bias=P["b3"])
        return self.parts

    # -- descriptors of the same launches for the grouped entry points (rrl_*_multi) ----------------------------
    def forward_desc(self, x, params=None, save=True, in_head=None):
        """rrl_stack_t of forward(x, params, save); the caller launches it with forward_multi().  `in_head`
        (rrl_policy_head_t): columns 2..3 of x are computed by the stack kernel itself from that policy head."""
        assert mlp3_supported(self.net.H, self.net.din, self.net.dout) and not self.finalize
        P = (params or self.net).p
        net = self.net
        self.x = x
        assert x.stride(1) == 1
        self.parts = (self.scratch, self.nsplit, self.scratch.stride(0)) if self.split else (self.out, 1, 0)
        p = _lib.ptr
        if in_head is not None:
            assert self.split and net.din == 4, "the input head lives in the column-split kernels"
        w2p = (params or self.net).w2_packed() if self.split else None
        return _lib.rrl_stack_t(net.G, x.shape[0], net.H, net.din, net.dout, x.stride(0), p(x), p(P["W1"]), p(P["b1"]),
                                p(P["W2"]), p(P["b2"]), p(P["W3"]), p(P["b3"]), p(self.h1) if save else None,
                                p(self.h2) if save else None, p(self.out), p(self.scratch) if self.split else None,
                                in_head if in_head is not None else _lib.rrl_policy_head_t(), int(in_head is not None),
                                p(w2p) if w2p is not None else None)

    def backward_descs(self, dout, weight_grads=True, input_grad=False):
        """(rrl_head_bwd_t, rrl_hidden_bwd_t, rrl_input_bwd_t) of backward(dout, weight_grads, input_grad)."""
        P, Gr, net = self.net.p, self.net.g, self.net
        G, B, H = net.G, self.B, net.H
        p = _lib.ptr
        wg = weight_grads
        if isinstance(dout, _lib.rrl_loss_t):
            loss = dout
        else:
            assert dout.is_contiguous()
            loss = _lib.rrl_loss_t(-1, 1, 0, p(dout), None, None, None, None, None, None, 0.0, 0, 0, 0, None, None, 0, 0, 0)
        head = _lib.rrl_head_bwd_t(loss, G, B, H, net.dout, p(self.h2), p(P["W3"]), p(Gr["W3"]) if wg else None,
                                   p(Gr["b3"]) if wg else None, p(self.dh2))
        if self.fuse_first:
            # folded dx partials only where a folding launch is taken: a loss description (the paired launches), not a dOut tensor
            self._folded = bool(self.fold_dx and input_grad and loss.kind >= 0)
            first = _lib.rrl_first_layer_t(p(self.x), p(P["W1"]), self.x.stride(0), net.din,
                                           p(self.first_part) if wg else None, self.first_part.stride(0),
                                           p(self.dx_part) if input_grad else None, int(self._folded))
            hidden = _lib.rrl_hidden_bwd_t(G, B, H, p(self.dh2), p(self.h1), p(P["W2"]), p(Gr["W2"]) if wg else None,
                                           p(Gr["b2"]) if wg else None, None, first)
            return head, hidden, None
        hidden = _lib.rrl_hidden_bwd_t(G, B, H, p(self.dh2), p(self.h1), p(P["W2"]), p(Gr["W2"]) if wg else None,
                                       p(Gr["b2"]) if wg else None, p(self.dh1), _lib.rrl_first_layer_t())
        inp = _lib.rrl_input_bwd_t(G, B, H, net.din, self.x.stride(0), p(self.dh1), p(self.x), p(P["W1"]),
                                   p(Gr["W1"]) if wg else None, p(Gr["b1"]) if wg else None,
                                   p(self.dx) if input_grad else None)
        return head, hidden, inp

    def backward(self, dout, weight_grads=True, input_grad=False):
        """dout: [G, B, dout] tensor, or an rrl_loss_t describing how the kernel computes it itself
        (rrl_mlp_head_backward_loss).  Writes parameter gradients into net.g (weight_grads) and/or returns
        dL/dx per head [G, B, din] (input_grad)."""
        if self.fuse_first:             # head backward, then hidden + first layer in one launch (rrl_first_layer_t)
            backward_multi([self.backward_descs(dout, weight_grads, input_grad)])
            return self.dx_part if input_grad else None
        P, Gr = self.net.p, self.net.g
        net, lib, st = self.net, _lib.load(), _lib.current_stream()
        G, B, H = net.G, self.B, net.H
        assert self.x.stride(1) == 1
        gw3 = Gr["W3"].data_ptr() if weight_grads else None
        gb3 = Gr["b3"].data_ptr() if weight_grads else None
        # last layer (1..4 outputs): dW3, db3 and the masked dh2 in one streaming kernel
        if isinstance(dout, _lib.rrl_loss_t):
            _lib.check(lib.rrl_mlp_head_backward_loss(C.byref(dout), G, B, H, net.dout, self.h2.data_ptr(),
                                                      P["W3"].data_ptr(), gw3, gb3, self.dh2.data_ptr(), st),
                       "rrl_mlp_head_backward_loss")
        else:
            assert dout.is_contiguous()
            _lib.check(lib.rrl_mlp_head_backward(G, B, H, net.dout, dout.data_ptr(), self.h2.data_ptr(),
                                                 P["W3"].data_ptr(), gw3, gb3, self.dh2.data_ptr(), st),
                       "rrl_mlp_head_backward")
        # hidden layer: the two H x H GEMMs on the MFMA kernel -- one launch when both are needed
        if weight_grads and self.pair_hidden:
            _lib.check(lib.rrl_mlp_hidden_backward(G, B, H, self.dh2.data_ptr(), self.h1.data_ptr(),
                                                   P["W2"].data_ptr(), Gr["W2"].data_ptr(), Gr["b2"].data_ptr(),
                                                   self.dh1.data_ptr(), st), "rrl_mlp_hidden_backward")
        else:
            if weight_grads:
                gemm(TN, self.dh2, self.h1, out=Gr["W2"], colsum=Gr["b2"])
            gemm(NN, self.dh2, P["W2"], out=self.dh1, mask=self.h1)
        # first layer (2..4 inputs): dW1, db1 and/or dx in one streaming kernel
        _lib.check(lib.rrl_mlp_input_backward(G, B, H, net.din, self.dh1.data_ptr(), self.x.data_ptr(),
                                              self.x.stride(0), P["W1"].data_ptr(),
                                              Gr["W1"].data_ptr() if weight_grads else None,
                                              Gr["b1"].data_ptr() if weight_grads else None,
                                              self.dx.data_ptr() if input_grad else None, st),
                   "rrl_mlp_input_backward")
        return self.dx if input_grad else None


def forward_multi(descs):
    """Independent stack forwards in ONE launch (rrl_mlp3_forward_multi)."""
    arr = (_lib.rrl_stack_t * len(descs))(*descs)
    record("forward", arr, len(descs))
    _lib.check(_lib.load().rrl_mlp3_forward_multi(len(descs), arr, _lib.current_stream()), "rrl_mlp3_forward_multi")


def backward_multi(triples):
    """Independent stack backwards, stage by stage: three launches for all of them (head, hidden, input)."""
    lib, st, n = _lib.load(), _lib.current_stream(), len(triples)
    head_list = [t[0] for t in triples]
    heads = (_lib.rrl_head_bwd_t * len(head_list))(*head_list)
    hidden = (_lib.rrl_hidden_bwd_t * n)(*[t[1] for t in triples])
    rest = [t[2] for t in triples if t[2] is not None]          # stacks whose first layer is not fused into `hidden`
    inputs = (_lib.rrl_input_bwd_t * len(rest))(*rest) if rest else None
    record("pair_bwd", heads, hidden, n)
    if inputs is not None:
        record("unsupported", "rrl_mlp_input_backward_multi")
    # head + hidden backward: one launch for the critic-loss kinds (rrl_mlp_backward_pair_multi), else the two launches
    _lib.check(lib.rrl_mlp_backward_pair_multi(n, heads, hidden, st), "rrl_mlp_backward_pair_multi")
    if inputs is not None:
        _lib.check(lib.rrl_mlp_input_backward_multi(len(inputs), inputs, st), "rrl_mlp_input_backward_multi")


def heads_multi(heads):
    arr = (_lib.rrl_policy_head_t * len(heads))(*heads)
    record("unsupported", "rrl_policy_heads_fwd_multi")
    _lib.check(_lib.load().rrl_policy_heads_fwd_multi(len(heads), arr, _lib.current_stream()),
               "rrl_policy_heads_fwd_multi")


class StackRows(Stack):
    """Rows [lo, hi) of a single-head Stack whose forward ran on a taller batch (several inputs of the SAME
    network stacked along the batch: one launch instead of one per input).  Shares the parent's activations and
    outputs, owns the backward workspace of its rows."""

    def __init__(self, parent, lo, hi):
        assert parent.net.G == 1, "row slices of a multi-head stack are not contiguous"
        self.parent, self.lo, self.hi = parent, lo, hi
        self.net, self.B = parent.net, hi - lo
        dev, H = parent.net.device, parent.net.H
        z = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        self.h1, self.h2 = parent.h1[:, lo:hi], parent.h2[:, lo:hi]
        self.dh1, self.dh2, self.dx = z(1, self.B, H), z(1, self.B, H), z(1, self.B, parent.net.din)
        self.pair_hidden = True
        self._folded = False
        self._init_first(dev, 1, self.B, H, parent.net.din)

    def forward(self, *a, **k):
        raise RuntimeError("run the parent's forward, then use .after_forward()")

    def after_forward(self):
        """(tensor, n_part, part_stride) of this slice's outputs after the parent's forward."""
        t, n_part, ps = self.parent.parts
        self.x = self.parent.x[self.lo:self.hi]
        self.parts = (t[0, 0, self.lo:self.hi] if n_part > 1 else t[0, self.lo:self.hi], n_part, ps)
        return self.parts


class FastUpdater:
    """Fused-kernel implementation of one SAC step and one Q_risk (+ model-free recovery) step for the
    default Recovery-RL configuration.  Owns the flat parameter storage of the agent's networks."""

    def __init__(self, agent, batch_size):
        self.agent = agent
        self.qr = agent.safety_critic
        self.B = B = batch_size
        dev = self.dev = agent.device
        self.lib = _lib.load()
        self.critic = flatten_twin_q(agent.critic, dev)
        self.critic_target = flatten_twin_q(agent.critic_target, dev)
        self.policy = flatten_policy(agent.policy, dev)
        self.qrisk = flatten_twin_q(self.qr.safety_critic, dev)
        self.qrisk_target = flatten_twin_q(self.qr.safety_critic_target, dev)
        self.recpolicy = flatten_policy(self.qr.policy, dev)
        z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)
        # stacks (workspaces) -- a network evaluated twice with saved activations needs two
        # SAC evaluates the policy on s' and on s with the same weights: one forward over 2B stacked rows
        self.pol_a = Stack(self.policy, B)
        self.pol_ab = Stack(self.policy, 2 * B)
        self.pol_next, self.pol_b = StackRows(self.pol_ab, 0, B), StackRows(self.pol_ab, B, 2 * B)
        self.cri_a, self.cri_b = Stack(self.critic, B), Stack(self.critic, B)
        self.qr_a, self.qr_b = Stack(self.qrisk, B), Stack(self.qrisk, B)
        self.rec_a = Stack(self.recpolicy, B)
        # grouped launches evaluate the target networks in the same launch as the online ones: own workspaces
        self.cri_t, self.qr_t = Stack(self.critic, B), Stack(self.qrisk, B)
        # kernels that do not depend on each other share launches (rrl_*_multi): the grouped entry points exist for the
        # one-launch stack forward only (H % 16 == 0, H <= 256, <= 4 inputs / outputs); other widths (--hidden_size 512)
        # take the separate calls, whose Stack.forward falls back to the per-layer GEMM kernel
        self.grouped = all(mlp3_supported(net.H, net.din, net.dout)
                           for net in (self.critic, self.policy, self.qrisk, self.recpolicy))
        self.fuse_heads = True     # policy heads evaluated by the consuming critic stack (rrl_stack_t.in_head)
        self.xu_q, self.x2u_q, self.xpu_q = z(B, 4), z(B, 4), z(B, 4)   # the Q_risk batch's rows (drawn up front)
        self.xu = z(B, 4)                                           # [s | a]
        self.x_pol = z(2 * B, 4)                                    # [s' | a'] stacked on [s | pi]
        self.x2u, self.xpu = self.x_pol[:B], self.x_pol[B:]
        self.logp2, self.logp = z(B), z(B)
        self.dq, self.dhead, self.draw = z(2, B, 1), z(1, B, 4), z(1, B, 2)
        self.dact = z(B, 2)
        self.losses = z(8)   # q1, q2, policy, (pad) | qr1, qr2, recpolicy, (pad)
        self.sync_world, self._avg, self.sac_bucket = 1, None, None
        self.fuse_loss = True      # loss gradients computed inside the head-backward kernels (no grad launches)
        self._noise = None
        self._noise_buf, self._actor_noise, self._actor_noise_fresh = None, None, False
        self.actor_rows = 0
        self.noise_seed = (int(getattr(agent, "seed", 0)) ^ 0x6E6F697365) & 0xFFFFFFFFFFFFFFFF
        self.noise_tick = torch.zeros(2, dtype=torch.int64, device=dev)
        self.alpha = torch.full((1,), float(agent.alpha), dtype=torch.float32, device=dev)
        self.scale = agent.policy.action_scale.to(dev).float().contiguous()
        self.bias = agent.policy.action_bias.to(dev).float().contiguous()
        self.rscale = self.qr.policy.action_scale.to(dev).float().contiguous()
        self.rbias = self.qr.policy.action_bias.to(dev).float().contiguous()

    # -- helpers ---------------------------------------------------------------------------------
    def stacks(self):
        return [self.pol_a, self.pol_ab, self.pol_next, self.pol_b, self.cri_a, self.cri_b, self.cri_t, self.qr_a,
                self.qr_b, self.qr_t, self.rec_a]

    def set_fuse_first(self, on):
        """First layer of every stack backward inside the hidden-layer launch (partial sums read by Adam and by the
        policy-head backward) or as its own launch writing the flat gradient buffer (needed when the gradient buffers
        are all-reduced, and by the stand-alone loss-gradient kernels of fuse_loss = False)."""
        for st in self.stacks():
            st.fuse_first = bool(on) and st.first_part is not None

    def gather_first_grads(self):
        """Write the summed (dW1, db1) partials of the last backward into the flat gradient buffers (inspection and
        tests; the optimiser reads the partials directly)."""
        for net, st in ((self.critic, self.cri_a), (self.policy, self.pol_b), (self.qrisk, self.qr_a),
                        (self.recpolicy, self.rec_a)):
            if st.fuse_first:
                net.grad[:st.n_first] = st.first_part.sum(0)

    # -- env-shard data parallelism (one learner, envs and replay split over ranks) ---------------------------
    def enable_grad_sync(self, world):
        """Every rank holds the same weights and averages gradients before each optimiser step: three
        all-reduces per iteration ([critic | policy] in one bucket, Q_risk, recovery policy -- the last two
        cannot share one because the recovery policy's gradient is taken at the UPDATED Q_risk, qrisk.py:150).
        Parameters, targets and Adam state start from rank 0's."""
        import torch.distributed as dist
        n1, n2 = self.critic.flat.numel(), self.policy.flat.numel()
        self.sac_bucket = torch.zeros(n1 + n2, dtype=torch.float32, device=self.dev)
        self.critic.rebind_grad(self.sac_bucket[:n1])
        self.policy.rebind_grad(self.sac_bucket[n1:])
        for net in (self.critic, self.critic_target, self.policy, self.qrisk, self.qrisk_target, self.recpolicy):
            dist.broadcast(net.flat, 0)
        self.sync_world = world
        self.set_fuse_first(False)              # the all-reduce works on the flat gradient buffers
        self._avg = dist.ReduceOp.AVG if dist.get_backend() == "nccl" else None

    def _sync(self, grad):
        if self.sync_world <= 1:
            return
        import torch.distributed as dist
        if self._avg is not None:
            dist.all_reduce(grad, op=self._avg)               # RCCL ring over xGMI; <= 0.8 MB: latency-bound
        else:                                                 # gloo (CPU-side reduction, tests): no AVG op
            dist.all_reduce(grad, op=dist.ReduceOp.SUM)
            grad.mul_(1.0 / self.sync_world)

    def _loss(self, kind, out, n_part, part_stride, out_t=None, v0=None, v1=None, v2=None, v3=None, alpha=None,
              f0=0.0, d_action=None, loss=None):
        """rrl_loss_t for Stack.backward: the head-backward kernel evaluates the loss gradient itself.
        d_action = the critic's input gradient dx [2, B, 4] whose action columns feed a policy head."""
        p = _lib.ptr
        ld = n_heads = hs = parts = ps = group = 0
        da = None
        if d_action is not None:
            if isinstance(d_action, tuple):           # Stack.dx_parts(): column-tile partials of the critic's dx
                d_action, parts, ps, group = d_action
            da, ld, n_heads, hs = d_action[0, :, 2:4].data_ptr(), d_action.stride(1), 2, d_action.stride(0)
        return _lib.rrl_loss_t(kind, n_part, part_stride, p(out), p(out_t), p(v0), p(v1), p(v2), p(v3), p(alpha),
                               float(f0), ld, n_heads, hs, da, p(loss), parts, ps, group)

    def _check(self, rc, what):
        _lib.check(rc, what)

    def _load_batch(self, batch, rows_loaded=False):
        s, a, r, s2, m = batch
        if not rows_loaded:       # the sample-gather kernel normally writes these rows itself
            self.xu[:, 0:2] = s
            self.xu[:, 2:4] = a
            self.x2u[:, 0:2] = s2
            self.xpu[:, 0:2] = s
        return s, a, r.reshape(-1), s2, m.reshape(-1)

    @property
    def rows(self):
        return (self.xu, self.x2u, self.xpu)

    def _fill_noise(self):
        """ONE rrl_normal_fill launch per lock-step iteration: the 4 [B,2] draws of the two updates followed
        by the 2 [N,2] draws of the acting pass (Philox stream RRL_STREAM_NOISE, device-side tick)."""
        n_act = self.actor_rows
        need = 4 * self.B * 2 + 2 * n_act * 2
        if self._noise_buf is None or self._noise_buf.numel() != need:
            self._noise_buf = torch.zeros(need, dtype=torch.float32, device=self.dev)
        record("unsupported", "rrl_normal_fill")
        self._check(self.lib.rrl_normal_fill(need // 2, self.noise_seed, 0, _lib.ptr(self.noise_tick), 1,
                                             _lib.ptr(self._noise_buf), _lib.current_stream()), "rrl_normal_fill")
        self._noise = self._noise_buf[:4 * self.B * 2].view(4, self.B, 2)
        self._actor_noise = self._noise_buf[4 * self.B * 2:].view(2, n_act, 2) if n_act else None
        self._actor_noise_fresh = n_act > 0

    def noise(self, which):
        """Policy noise for the two updates of one iteration (which = 0: the SAC update draws fresh noise for
        the whole iteration, 1: the Q_risk update uses the second half)."""
        if which == 0 or self._noise is None:
            self._fill_noise()
        n = self._noise
        return (n[0], n[1]) if which == 0 else (n[2], n[3])

    def actor_noise(self, n):
        """[2, n, 2] draws for FastActor: the tail of this iteration's fill, or its own fill when the updates did
        not run (or ran for a different n) since the last acting pass."""
        if self.actor_rows != n:
            self.actor_rows = n
            self._actor_noise_fresh = False
        if not self._actor_noise_fresh:
            self._fill_noise()
        self._actor_noise_fresh = False
        return self._actor_noise

    def _gauss_fwd(self, head, eps, action_view, logp):
        t, n_part, ps = head
        self._check(self.lib.rrl_gauss_head_fwd(self.B, t.data_ptr(), n_part, ps, eps.data_ptr(),
                                                self.scale.data_ptr(), self.bias.data_ptr(),
                                                action_view.data_ptr(), action_view.stride(0), logp.data_ptr(),
                                                None, None, None, _lib.current_stream()), "rrl_gauss_head_fwd")

    # -- grouped path: the same kernels, independent ones sharing a launch ------------------------------------------
    @property
    def rows_q(self):
        return (self.xu_q, self.x2u_q, self.xpu_q)

    def _gauss_desc(self, head, eps, action_view, logp, n=None, obs_in=None, obs_out=None):
        t, n_part, ps = head
        p = _lib.ptr
        return _lib.rrl_policy_head_t(_lib.HEAD_GAUSS, n or self.B, p(t), n_part, ps, p(eps), p(self.scale),
                                      p(self.bias), p(action_view), action_view.stride(0), p(logp), None, p(obs_in),
                                      p(obs_out), None, 0.0)

    def _stoch_desc(self, head, eps, action_view, n=None):
        t, n_part, ps = head
        p = _lib.ptr
        return _lib.rrl_policy_head_t(_lib.HEAD_STOCH, n or self.B, p(t), n_part, ps, p(eps), p(self.rscale),
                                      p(self.rbias), p(action_view), action_view.stride(0), None, None, None, None,
                                      p(self.recpolicy.p["log_std"]), float(self.qr.policy.min_log_std))

    def update_pair(self, memory, recovery_memory, rider=None):
        """One SAC update and (recovery_memory not None) one Q_risk + recovery-policy update of a lock-step iteration
        (experiment.py:397-416): both replay draws and the iteration's policy noise in ONE launch, then the two
        updates on the grouped kernels.  Same draws, same arithmetic, same parameters as the separate calls.
        `rider` = (FastActor, obs): the acting pass that follows this update takes two of its three forwards along in the
        Q_risk update's launches (FastActor.ride_*; the LAST update pair of an iteration only)."""
        B, qr = self.B, self.qr
        d1, batch = memory.draw_desc(B, rows=self.rows)
        d2 = batch_q = None
        if recovery_memory is not None:
            d2, batch_q = recovery_memory.draw_desc(B, pos_fraction=qr.pos_fraction, rows=self.rows_q,
                                                    demo_share=qr.demo_share)
        n_act = self.actor_rows
        need = 4 * B * 2 + 2 * n_act * 2
        if self._noise_buf is None or self._noise_buf.numel() != need:
            self._noise_buf = torch.zeros(need, dtype=torch.float32, device=self.dev)
        record("sample", _lib.rrl_sample_args_t(C.pointer(d1), C.pointer(d2) if d2 is not None else None, need // 2,
                                                self.noise_seed, 0, _lib.ptr(self.noise_tick), 1,
                                                _lib.ptr(self._noise_buf)), d1, d2)
        self._check(self.lib.rrl_sample_multi(C.byref(d1), C.byref(d2) if d2 is not None else None, need // 2,
                                              self.noise_seed, 0, _lib.ptr(self.noise_tick), 1,
                                              _lib.ptr(self._noise_buf), _lib.current_stream()), "rrl_sample_multi")
        self._noise = self._noise_buf[:4 * B * 2].view(4, B, 2)
        self._actor_noise = self._noise_buf[4 * B * 2:].view(2, n_act, 2) if n_act else None
        self._actor_noise_fresh = n_act > 0
        n = self._noise
        self.sac_update_grouped(batch, n[0], n[1])
        if recovery_memory is not None:
            self.qrisk_update_grouped(batch_q, n[2], n[3], rider=rider)
        return self.losses

    def sac_update_grouped(self, batch, eps_next, eps_pi):
        """sac_update with 11 launches instead of 17 (rows already written by the draw)."""
        ag, B = self.agent, self.B
        s, a, r, s2, m = batch
        r, m = r.reshape(-1), m.reshape(-1)
        if self.pol_ab.split:      # one-member group: the stand-alone launch's kernel body, and a launch the tape can pack
            forward_multi([self.pol_ab.forward_desc(self.x_pol[:, 0:2])])
        else:
            self.pol_ab.forward(self.x_pol[:, 0:2])
        head2, head = self.pol_next.after_forward(), self.pol_b.after_forward()
        hd2 = self._gauss_desc(head2, eps_next, self.x2u[:, 2:4], self.logp2)
        hd1 = self._gauss_desc(head, eps_pi, self.xpu[:, 2:4], self.logp)
        fuse = self.fuse_heads and self.cri_t.split
        if not fuse:
            heads_multi([hd2, hd1])
            hd2 = hd1 = None
        # critic_target(s', a'), critic(s, a), critic(s, pi): three independent forwards (sac.py:192-218); a' and pi are
        # evaluated by the stacks that consume them
        forward_multi([self.cri_t.forward_desc(self.x2u, params=self.critic_target, save=False, in_head=hd2),
                       self.cri_a.forward_desc(self.xu), self.cri_b.forward_desc(self.xpu, in_head=hd1)])
        qt, n_part, ps = self.cri_t.parts
        q, qp = self.cri_a.parts[0], self.cri_b.parts[0]
        # the critic's backward for its own loss (weight gradients) and for the policy loss (input gradient)
        backward_multi([
            self.cri_a.backward_descs(self._loss(_lib.LOSS_SAC_CRITIC, q, n_part, ps, out_t=qt, v0=self.logp2, v1=r,
                                                 v2=m, alpha=self.alpha, f0=ag.gamma, loss=self.losses)),
            self.cri_b.backward_descs(self._loss(_lib.LOSS_SAC_POLICY, qp, n_part, ps, v0=self.logp, alpha=self.alpha,
                                                 loss=self.losses[2:]), weight_grads=False, input_grad=True)])
        ht, hn, hs = head
        self.pol_b.backward(self._loss(_lib.LOSS_GAUSS_HEAD, ht, hn, hs, v0=eps_pi, v1=self.scale,
                                       f0=float(ag.alpha) / B, d_action=self.cri_b.dx_parts()))
        if self.sync_world > 1:
            self._sync(self.sac_bucket)
        adam_multi(ag.lr, [(self.critic, self.critic_target, ag.tau, self.cri_a.grad_part),
                           (self.policy, None, 0.0, self.pol_b.grad_part)])
        return self.losses

    def can_carry_actor(self):
        """The acting pass's task-policy and Q_risk forwards can ride in this update's forward launches (qrisk_update_grouped):
        model-free recovery, policy heads evaluated by the consuming stacks, every stack on the column-split kernels."""
        return bool(self.grouped and self.qr.MF_recovery and self.fuse_heads and self.qr_t.split and self.qr_b.split
                    and self.pol_a.split and self.rec_a.split and self.sync_world == 1)

    def qrisk_update_grouped(self, batch, eps_next, eps_pi, rider=None):
        """qrisk_update with 15 launches instead of 19: the task policy on s' and the recovery policy on s in one
        forward launch (the recovery policy does not depend on the critic step in between), their heads in one, the
        target and online critics in one.
        rider = (FastActor, obs) (can_carry_actor()): the acting pass that follows needs the task policy on the N observations
        -- final since the SAC step -- and Q_risk(obs, a_task) -- final since this update's critic step: the first rides in
        this update's first forward launch, the second in its forward at the updated critic.  The acting pass is left with
        the recovery policy's forward (final only after this update's last step): 17 -> 16 launches per iteration, and the
        two 256-row launches that waited alone on the chip run under the 4096-row ones.  Same kernels, same inputs: same bits."""
        qr, B = self.qr, self.B
        s, a, c, s2, m = batch
        c, m = c.reshape(-1), m.reshape(-1)
        xu, x2u, xpu = self.rows_q
        mf = bool(qr.MF_recovery)
        fwd = [self.pol_a.forward_desc(x2u[:, 0:2], save=False)]          # a' from the TASK policy (qrisk.py:119-120)
        if mf:
            fwd.append(self.rec_a.forward_desc(xpu[:, 0:2]))
        fuse = self.fuse_heads and self.qr_t.split
        if rider is not None:
            assert self.can_carry_actor()
            fwd.insert(0, rider[0].ride_policy(rider[1]))       # the large member first (mlp_fwd_kernels.hip: measured forms)
        forward_multi(fwd)
        hd_next = self._gauss_desc(self.pol_a.parts, eps_next, x2u[:, 2:4], self.logp2)
        hd_rec = self._stoch_desc(self.rec_a.parts, eps_pi, xpu[:, 2:4]) if mf else None
        if not fuse:
            heads_multi([hd_next] + ([hd_rec] if mf else []))
            hd_next = hd_rec = None
        forward_multi([self.qr_t.forward_desc(x2u, params=self.qrisk_target, save=False, in_head=hd_next),
                       self.qr_a.forward_desc(xu)])
        zt, n_part, ps = self.qr_t.parts
        z = self.qr_a.parts[0]
        self.qr_a.backward(self._loss(_lib.LOSS_QRISK_CRITIC, z, n_part, ps, out_t=zt, v0=c, v1=m,
                                      f0=qr.gamma_safe, loss=self.losses[4:]))
        self._sync(self.qrisk.grad)
        self.qrisk.adam(qr.lr, target=self.qrisk_target, tau=qr.tau, part=self.qr_a.grad_part)
        if mf:                                                             # qrisk.py:150-158, at the UPDATED critic
            raw, rn, rs = self.rec_a.parts
            ls = self.recpolicy.p["log_std"]
            if hd_rec is not None:         # the recovery action is evaluated by the critic stack that consumes it
                forward_multi(([rider[0].ride_qrisk()] if rider else []) + [self.qr_b.forward_desc(xpu, in_head=hd_rec)])
                zp, n_part, ps = self.qr_b.parts
            else:
                zp, n_part, ps = self.qr_b.forward(xpu)
            self.qr_b.backward(self._loss(_lib.LOSS_QRISK_POLICY, zp, n_part, ps, loss=self.losses[6:]),
                               weight_grads=False, input_grad=True)
            self.rec_a.backward(self._loss(_lib.LOSS_STOCH_HEAD, raw, rn, rs, v0=eps_pi, v1=ls, v2=self.rscale,
                                           f0=qr.policy.min_log_std, d_action=self.qr_b.dx_parts(),
                                           loss=self.recpolicy.g["log_std"]))
            self._sync(self.recpolicy.grad)
            self.recpolicy.adam(qr.lr, part=self.rec_a.grad_part)
        return self.losses

    # -- SAC -------------------------------------------------------------------------------------
    def sac_update(self, batch, eps_next, eps_pi, rows_loaded=False):
        ag, B, lib, st = self.agent, self.B, self.lib, _lib.current_stream()
        s, a, r, s2, m = self._load_batch(batch, rows_loaded)
        # pi(s') and pi(s) share the weights (both gradients are taken before either step): ONE policy forward
        self.pol_ab.forward(self.x_pol[:, 0:2])
        # target: a' ~ pi(s'), min Q_target(s', a') - alpha log pi  (sac.py:192-201)
        head2 = self.pol_next.after_forward()
        self._gauss_fwd(head2, eps_next, self.x2u[:, 2:4], self.logp2)
        qt, n_part, ps = self.cri_b.forward(self.x2u, params=self.critic_target, save=False)
        q, _, _ = self.cri_a.forward(self.xu)
        if self.fuse_loss:                                             # critic gradients (sac.py:233-235)
            self.cri_a.backward(self._loss(_lib.LOSS_SAC_CRITIC, q, n_part, ps, out_t=qt, v0=self.logp2, v1=r, v2=m,
                                           alpha=self.alpha, f0=ag.gamma, loss=self.losses))
        else:
            self._check(lib.rrl_sac_critic_grad(B, q.data_ptr(), qt.data_ptr(), n_part, ps, self.logp2.data_ptr(),
                                                r.data_ptr(), m.data_ptr(), ag.gamma, self.alpha.data_ptr(), None,
                                                self.dq.data_ptr(), self.losses.data_ptr(), st),
                        "rrl_sac_critic_grad")
            self.cri_a.backward(self.dq)
        # policy loss at the PRE-update critic (both gradients before either step)
        head = self.pol_b.after_forward()
        self._gauss_fwd(head, eps_pi, self.xpu[:, 2:4], self.logp)
        qp, n_part, ps = self.cri_b.forward(self.xpu)
        ht, hn, hs = head
        if self.fuse_loss:
            self.cri_b.backward(self._loss(_lib.LOSS_SAC_POLICY, qp, n_part, ps, v0=self.logp, alpha=self.alpha,
                                           loss=self.losses[2:]), weight_grads=False, input_grad=True)
            # d pi = action columns of dx [2,B,4], summed over the two critic heads inside the policy's head backward
            self.pol_b.backward(self._loss(_lib.LOSS_GAUSS_HEAD, ht, hn, hs, v0=eps_pi, v1=self.scale,
                                           f0=float(ag.alpha) / B, d_action=self.cri_b.dx_parts()))
        else:
            assert not self.cri_b.fuse_first, "fuse_loss = False needs set_fuse_first(False)"
            self._check(lib.rrl_sac_policy_grad(B, qp.data_ptr(), n_part, ps, self.logp.data_ptr(),
                                                self.alpha.data_ptr(), self.dq.data_ptr(),
                                                self.losses[2:].data_ptr(), st), "rrl_sac_policy_grad")
            dx = self.cri_b.backward(self.dq, weight_grads=False, input_grad=True)      # [2,B,4]
            self._check(lib.rrl_gauss_head_bwd(B, ht.data_ptr(), hn, hs, eps_pi.data_ptr(), self.scale.data_ptr(),
                                               dx[0, :, 2:4].data_ptr(), dx.stride(1), 2, dx.stride(0),
                                               float(ag.alpha) / B, self.dhead.data_ptr(), st),
                        "rrl_gauss_head_bwd")
            self.pol_b.backward(self.dhead)
        if self.sync_world > 1:
            self._sync(self.sac_bucket)
        # both optimiser steps + the soft target update (:273-274) in one launch
        adam_multi(ag.lr, [(self.critic, self.critic_target, ag.tau, self.cri_a.grad_part),
                           (self.policy, None, 0.0, self.pol_b.grad_part)])
        return self.losses

    # -- Q_risk ------------------------------------------------------------------------------------
    def qrisk_update(self, batch, eps_next, eps_pi, rows_loaded=False):
        qr, B, lib, st = self.qr, self.B, self.lib, _lib.current_stream()
        s, a, c, s2, m = self._load_batch(batch, rows_loaded)
        head2 = self.pol_a.forward(s2, save=False)                     # a' from the TASK policy (qrisk.py:119-120)
        self._gauss_fwd(head2, eps_next, self.x2u[:, 2:4], self.logp2)
        zt, n_part, ps = self.qr_b.forward(self.x2u, params=self.qrisk_target, save=False)
        z, _, _ = self.qr_a.forward(self.xu)
        if self.fuse_loss:
            self.qr_a.backward(self._loss(_lib.LOSS_QRISK_CRITIC, z, n_part, ps, out_t=zt, v0=c, v1=m,
                                          f0=qr.gamma_safe, loss=self.losses[4:]))
        else:
            self._check(lib.rrl_qrisk_critic_grad(B, z.data_ptr(), zt.data_ptr(), n_part, ps, c.data_ptr(),
                                                  m.data_ptr(), qr.gamma_safe, self.dq.data_ptr(),
                                                  self.losses[4:].data_ptr(), st), "rrl_qrisk_critic_grad")
            self.qr_a.backward(self.dq)
        self._sync(self.qrisk.grad)
        self.qrisk.adam(qr.lr, target=self.qrisk_target, tau=qr.tau, part=self.qr_a.grad_part)
        if qr.MF_recovery:                                              # qrisk.py:150-158, at the UPDATED critic
            raw, rn, rs = self.rec_a.forward(s)
            ls = self.recpolicy.p["log_std"]
            self._check(lib.rrl_stoch_head_fwd(B, raw.data_ptr(), rn, rs, eps_pi.data_ptr(), ls.data_ptr(),
                                               qr.policy.min_log_std, self.rscale.data_ptr(), self.rbias.data_ptr(),
                                               self.xpu[:, 2:4].data_ptr(), 4, None, st), "rrl_stoch_head_fwd")
            zp, n_part, ps = self.qr_b.forward(self.xpu)
            if self.fuse_loss:
                self.qr_b.backward(self._loss(_lib.LOSS_QRISK_POLICY, zp, n_part, ps, loss=self.losses[6:]),
                                   weight_grads=False, input_grad=True)
                self.rec_a.backward(self._loss(_lib.LOSS_STOCH_HEAD, raw, rn, rs, v0=eps_pi, v1=ls, v2=self.rscale,
                                               f0=qr.policy.min_log_std, d_action=self.qr_b.dx_parts(),
                                               loss=self.recpolicy.g["log_std"]))
            else:
                self._check(lib.rrl_qrisk_policy_grad(B, zp.data_ptr(), n_part, ps, self.dq.data_ptr(),
                                                      self.losses[6:].data_ptr(), st), "rrl_qrisk_policy_grad")
                dx = self.qr_b.backward(self.dq, weight_grads=False, input_grad=True)
                self._check(lib.rrl_stoch_head_bwd(B, raw.data_ptr(), rn, rs, eps_pi.data_ptr(), ls.data_ptr(),
                                                   qr.policy.min_log_std, self.rscale.data_ptr(),
                                                   dx[0, :, 2:4].data_ptr(), dx.stride(1), 2, dx.stride(0),
                                                   self.draw.data_ptr(), self.recpolicy.g["log_std"].data_ptr(),
                                                   st), "rrl_stoch_head_bwd")
                self.rec_a.backward(self.draw)
            self._sync(self.recpolicy.grad)
            self.recpolicy.adam(qr.lr, part=self.rec_a.grad_part)
        return self.losses


class FastActor:
    """Batched get_action (experiment.py:546-577) for N envs on the fused kernels: task policy sample,
    Q_risk of (s, a_task), model-free recovery action and the recovery gate -- 8 launches instead of
    ~60 PyTorch ones."""

    def __init__(self, fast, n):
        self.f, self.n = fast, n
        dev = fast.dev
        z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)
        self.pol, self.qr, self.rec = Stack(fast.policy, n), Stack(fast.qrisk, n), Stack(fast.recpolicy, n)
        self.xa = z(n, 4)                       # [s | a_task]
        self.task_action, self.rec_action, self.real_action = z(n, 2), z(n, 2), z(n, 2)
        self.recovery = torch.zeros(n, dtype=torch.uint8, device=dev)
        self._ride = None

    # -- two of the three forwards of act(defer_select=True) as riders of the Q_risk update's launches -----------------------
    def ride_policy(self, obs):
        """rrl_stack_t of the task policy's forward on the acting observations, for a launch the caller issues (any time after
        the SAC step of this iteration).  Starts a ride: ride_qrisk() and act() complete it."""
        assert obs.shape[0] == self.n and self.f.can_carry_actor() and self.qr.split
        self._ride = {"obs": obs, "noise": self.f.actor_noise(self.n), "qrisk": False}
        return self.pol.forward_desc(obs, save=False)

    def ride_qrisk(self):
        """rrl_stack_t of Q_risk(obs, a_task) with the task head evaluated by the stack (it stores the action in xa for the
        step kernel), for a launch the caller issues after the policy rider's launch and the safety critic's step."""
        f, r = self.f, self._ride
        task_head = f._gauss_desc(self.pol.parts, r["noise"][0], self.xa[:, 2:4], None, n=self.n, obs_in=r["obs"],
                                  obs_out=self.xa)
        self.qr.finalize = False
        r["qrisk"] = True
        return self.qr.forward_desc(self.xa, save=False, in_head=task_head)

    def _finish_ride(self, obs, eps_safe):
        """What is left of act(defer_select=True) after both riders: the recovery policy's forward (its step is the last of
        the iteration's updates); its head and the gate run in the env-step kernel."""
        f, r = self.f, self._ride
        self._ride = None
        assert r["qrisk"] and obs is r["obs"], "the acting pass of a ride must follow its two riders, on the same observations"
        forward_multi([self.rec.forward_desc(obs, save=False)])
        rec_head = f._stoch_desc(self.rec.parts, r["noise"][1], self.rec_action, n=self.n)
        zq, zn, zs = self.qr.parts
        self.pending_select = (zq, zn, zs, float(eps_safe), None, rec_head)
        return self.xa[:, 2:4], self.real_action, self.recovery

    def act(self, obs, eps_safe, use_recovery, mf_recovery, noise=None, defer_select=False):
        """-> (task action [n,2], executed action [n,2], recovery u8[n] or None); persistent buffers.
        defer_select: the recovery gate is left to the env-step kernel (rrl_*_step_push_select); `pending_select` then
        holds its inputs, the task action is the strided view xa[:, 2:4] and the other two are filled by that kernel."""
        f, n, lib, st = self.f, self.n, self.f.lib, _lib.current_stream()
        self.pending_select = None
        if self._ride is not None:
            assert noise is None and defer_select and use_recovery and mf_recovery
            return self._finish_ride(obs, eps_safe)
        if noise is None:
            noise = f.actor_noise(n)
        if f.grouped and use_recovery and mf_recovery:
            # task policy and recovery policy on the same observations: one forward launch, one head launch
            forward_multi([self.pol.forward_desc(obs, save=False), self.rec.forward_desc(obs, save=False)])
            task_head = f._gauss_desc(self.pol.parts, noise[0], self.xa[:, 2:4], None, n=n, obs_in=obs, obs_out=self.xa)
            rec_head = f._stoch_desc(self.rec.parts, noise[1], self.rec_action, n=n)
            if defer_select and f.fuse_heads and self.qr.split:
                # no head launch: the task action is evaluated by the Q_risk stack that consumes it (and stored in xa
                # for the step kernel), the recovery action by the step kernel itself
                self.qr.finalize = False
                forward_multi([self.qr.forward_desc(self.xa, save=False, in_head=task_head)])
                zq, zn, zs = self.qr.parts
                self.pending_select = (zq, zn, zs, float(eps_safe), None, rec_head)
                return self.xa[:, 2:4], self.real_action, self.recovery
            heads_multi([task_head, rec_head])
            if defer_select:
                self.qr.finalize = False            # the step kernel adds the partial last-layer sums itself
                zq, zn, zs = self.qr.forward(self.xa, save=False)
                self.pending_select = (zq, zn, zs, float(eps_safe), self.rec_action, None)
                return self.xa[:, 2:4], self.real_action, self.recovery
            self.qr.finalize = True
            zq, _, _ = self.qr.forward(self.xa, save=False)
            _lib.check(lib.rrl_recovery_select(n, zq.data_ptr(), eps_safe, self.xa[:, 2:4].data_ptr(), 4,
                                               self.rec_action.data_ptr(), self.real_action.data_ptr(),
                                               self.recovery.data_ptr(), self.task_action.data_ptr(), st),
                       "rrl_recovery_select")
            return self.task_action, self.real_action, self.recovery
        head, hn, hs = self.pol.forward(obs, save=False)
        if not use_recovery:
            _lib.check(lib.rrl_gauss_head_fwd(n, head.data_ptr(), hn, hs, noise[0].data_ptr(), f.scale.data_ptr(),
                                              f.bias.data_ptr(), self.task_action.data_ptr(), 2, None, None, None,
                                              None, st), "rrl_gauss_head_fwd")
            return self.task_action, self.task_action, None
        # the head kernel also copies obs into columns 0..1 of xa: [s | a_task] is assembled without a copy launch
        _lib.check(lib.rrl_gauss_head_fwd(n, head.data_ptr(), hn, hs, noise[0].data_ptr(), f.scale.data_ptr(),
                                          f.bias.data_ptr(), self.xa[:, 2:4].data_ptr(), 4, None, None,
                                          obs.data_ptr(), self.xa.data_ptr(), st), "rrl_gauss_head_fwd")
        self.qr.finalize = True                  # recovery_select reads a plain [2,n] tensor
        zq, _, _ = self.qr.forward(self.xa, save=False)
        assert mf_recovery, "FastActor covers the model-free recovery policy"
        raw, rn, rs = self.rec.forward(obs, save=False)
        _lib.check(lib.rrl_stoch_head_fwd(n, raw.data_ptr(), rn, rs, noise[1].data_ptr(),
                                          f.recpolicy.p["log_std"].data_ptr(), f.qr.policy.min_log_std,
                                          f.rscale.data_ptr(), f.rbias.data_ptr(), self.rec_action.data_ptr(), 2,
                                          None, st), "rrl_stoch_head_fwd")
        _lib.check(lib.rrl_recovery_select(n, zq.data_ptr(), eps_safe, self.xa[:, 2:4].data_ptr(), 4,
                                           self.rec_action.data_ptr(), self.real_action.data_ptr(),
                                           self.recovery.data_ptr(), self.task_action.data_ptr(), st),
                   "rrl_recovery_select")
        return self.task_action, self.real_action, self.recovery


    def _act_gate(self, obs, eps_safe, noise=None):
        """Task action + recovery gate for a controller that acts elsewhere (model-based recovery: MPC.act on the gated rows):
        -> (task action [n,2], recovery u8[n]); persistent buffers."""
        f, n, lib, st = self.f, self.n, self.f.lib, _lib.current_stream()
        if noise is None:
            noise = f.actor_noise(n)
        self.pending_select = None
        head, hn, hs = self.pol.forward(obs, save=False)
        # the head kernel also copies obs into columns 0..1 of xa: [s | a_task] is assembled without a copy launch
        _lib.check(lib.rrl_gauss_head_fwd(n, head.data_ptr(), hn, hs, noise[0].data_ptr(), f.scale.data_ptr(),
                                          f.bias.data_ptr(), self.xa[:, 2:4].data_ptr(), 4, None, None,
                                          obs.data_ptr(), self.xa.data_ptr(), st), "rrl_gauss_head_fwd")
        self.qr.finalize = True                  # recovery_select reads a plain [2,n] tensor
        zq, _, _ = self.qr.forward(self.xa, save=False)
        # recovery[i] = max(sigmoid(z1), sigmoid(z2)) > eps_safe (experiment.py:566-571); the kernel's action selection runs
        # on a dummy recovery action: the planner's action is merged in by the caller
        _lib.check(lib.rrl_recovery_select(n, zq.data_ptr(), eps_safe, self.xa[:, 2:4].data_ptr(), 4,
                                           self.rec_action.data_ptr(), self.real_action.data_ptr(),
                                           self.recovery.data_ptr(), self.task_action.data_ptr(), st),
                   "rrl_recovery_select")
        return self.task_action, self.recovery


FastActor.act_gate = FastActor._act_gate


def fast_path_supported(cfg):
    """The fused path covers the Recovery-RL configurations (task SAC + Q_risk, model-free or
    model-based recovery, reward penalty); the comparison algorithms use the autograd path."""
    return (cfg.policy == "Gaussian" and not cfg.automatic_entropy_tuning and not cfg.DGD_constraints
            and not cfg.RCPO and not cfg.update_nu and not cfg.use_constraint_sampling
            and cfg.target_update_interval == 1 and not getattr(cfg, "cnn", False))
