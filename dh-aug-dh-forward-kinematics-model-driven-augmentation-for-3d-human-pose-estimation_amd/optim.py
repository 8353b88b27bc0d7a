"""Fused Adam on flat fp32 buffers + the data-parallel gradient exchange.

One optimizer owns ONE contiguous parameter buffer and ONE contiguous gradient buffer (the module's parameters
and .grad tensors are views into them), so that an optimizer step is
    [one RCCL all-reduce of the flat gradient bucket]  ->  one dhaug_adam_step launch.
With world_size > 1 (torch.distributed initialised, backend "nccl" = RCCL on ROCm, or "gloo" in the CPU tests)
the all-reduce sums the replicas' gradients and the kernel scales by 1/world_size.  No activation, parameter or
optimizer state crosses the fabric."""
import os

import torch
import torch.distributed as dist

from . import autograd_ops as A
from . import ops


# the optimizer step as two streaming launches (count + Adam + the weights' nt copies | nn = nt^T); DHAUG_NO_FUSED_ADAM=1: the four
# launches they replace (kept for the A/B and the equality test)
FUSED_STEP = os.environ.get("DHAUG_NO_FUSED_ADAM") is None


class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam semantics (lr 1e-4, betas (0.5, 0.9) at R/models_Fk_GAN/model_fk_gan_train.py:112-118)."""

    def __init__(self, params, lr=1e-4, betas=(0.5, 0.9), eps=1e-8, process_group=None, data_parallel=None):
        params = [p for p in params]
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self._params = params
        n = sum(p.numel() for p in params)
        dev = params[0].device
        self.flat_param = torch.empty(n, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        self._views = []
        off = 0
        for p in params:
            k = p.numel()
            self.flat_param[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.flat_param[off:off + k].view(p.shape)
            gv = self.flat_grad[off:off + k].view(p.shape)
            p.grad = gv
            p._dhaug_grad_slot = gv             # autograd_ops.LinearFn.backward accumulates into it directly
            self._views.append(gv)
            off += k
        self.step_count = 0
        # the same count on the device: the Adam kernel reads it there, so a captured hipGraph of a training step replays
        # with the right bias corrections (the host count is bookkeeping: state_dict, tests)
        self.step_dev = torch.zeros(1, dtype=torch.int32, device=dev) if dev.type == "cuda" else None
        self.process_group = process_group
        self.data_parallel = data_parallel      # None: follow torch.distributed state
        # overlap: step() only STARTS the gradient exchange (asynchronous all-reduce); the Adam launch waits for it in
        # flush(), which zero_grad() calls -- i.e. right before this network is used again.  The epoch loops switch it on
        # for multi-rank runs and interleave the steps of different networks, so that one network's all-reduce travels
        # while the next network's step computes (25-100 MB buckets in the video configuration).
        self.overlap = False
        self._pending = None
        self._packs = None                      # (bf16 arena, device descriptors, [(param, nt view, nn view)])

    def state_dict(self):
        """torch.optim.Optimizer.state_dict() plus the flat moments and the step count (checkpoint / resume)"""
        d = super().state_dict()
        count = int(self.step_dev.item()) if self.step_dev is not None else self.step_count     # (graph replays advance only the device count)
        d["dhaug_flat"] = dict(exp_avg=self.exp_avg.clone(), exp_avg_sq=self.exp_avg_sq.clone(), step_count=count)
        return d

    def load_state_dict(self, state_dict):
        flat = state_dict.get("dhaug_flat")
        super().load_state_dict({k: v for k, v in state_dict.items() if k != "dhaug_flat"})
        if flat is not None:
            self.exp_avg.copy_(flat["exp_avg"])
            self.exp_avg_sq.copy_(flat["exp_avg_sq"])
            self.step_count = int(flat["step_count"])
            if self.step_dev is not None:
                self.step_dev.fill_(self.step_count)
        self._check_views()

    def _check_views(self):
        """parameters must still be views of the flat buffer (a module.to() / load that re-allocates them breaks that)"""
        lo = self.flat_param.data_ptr()
        hi = lo + self.flat_param.numel() * 4
        for p in self._params:
            if not (lo <= p.data_ptr() < hi):
                raise RuntimeError("FusedAdam: a parameter no longer lives in the optimizer's flat buffer (module moved or "
                                   "re-allocated after the optimizer was built); rebuild the optimizer")

    def zero_grad(self, set_to_none=False):
        self.flush()                            # never zero a bucket that is still being reduced
        self.flat_grad.zero_()
        for p, gv in zip(self._params, self._views):
            p.grad = gv

    def _gather_grads(self):
        for p, gv in zip(self._params, self._views):
            if p.grad is None:
                gv.zero_()
            elif p.grad.data_ptr() != gv.data_ptr():     # a module.zero_grad(set_to_none=True) replaced the view
                gv.copy_(p.grad)
            p.grad = gv

    def world_size(self):
        use = self.data_parallel
        if use is None:
            use = dist.is_available() and dist.is_initialized()
        return dist.get_world_size(self.process_group) if use else 1

    @torch.no_grad()
    def exchange(self):
        """the one exchange step of the data-parallel path: sum the flat gradient bucket over the replicas.
        Returns the world size (the Adam kernel applies the 1/world scale)."""
        self._gather_grads()
        ws = self.world_size()
        if ws > 1:
            dist.all_reduce(self.flat_grad, op=dist.ReduceOp.SUM, group=self.process_group)
        return ws

    @torch.no_grad()
    def step(self, closure=None):
        self._check_views()
        self._gather_grads()
        ws = self.world_size()
        self.step_count += 1
        if not self.flat_param.is_cuda:
            raise RuntimeError("FusedAdam needs GPU parameters (no CPU fallback exists)")
        if ws > 1:
            from . import graphs
            rec = graphs.RECORDER
            if rec is not None:
                # a segmented hipGraph capture (graphs.SegmentedCall): the collective is an event BETWEEN two graphs
                if not hasattr(rec, "cut"):
                    raise RuntimeError("FusedAdam.step with world_size > 1 inside a %s capture: only graphs.SegmentedCall can "
                                       "cut a capture at a collective" % type(rec).__name__)
                rec.cut(("allreduce", self))
                if self.overlap:
                    self._pending = (None, ws)
                    return None
                rec.cut(("wait", self))
            else:
                work = dist.all_reduce(self.flat_grad, op=dist.ReduceOp.SUM, group=self.process_group, async_op=True)
                if self.overlap:
                    self._pending = (work, ws)
                    return None
                work.wait()
        self._apply(ws)
        return None

    def _apply(self, ws):
        g = self.param_groups[0]
        # the packed copies of THIS network's weights are stale (other networks keep theirs); the bf16 operand copies the
        # training path reads are rebuilt right here
        for p in self._params:
            p._dhaug_epoch = getattr(p, "_dhaug_epoch", 0) + 1
        if FUSED_STEP:
            # count + Adam + the weights' nt copies in one streaming launch, nn = nt^T in a second (dhaug_adam_repack_step)
            from . import _lib
            arena, dev, views, adam_dev, ndesc, nitems = self._ensure_packs()
            b = tuple(g["betas"])
            _lib.call("dhaug_adam_repack_step", self.flat_param.data_ptr(), self.flat_grad.data_ptr(), self.exp_avg.data_ptr(),
                      self.exp_avg_sq.data_ptr(), g["lr"], b[0], b[1], g["eps"], self.step_dev.data_ptr(), 1.0 / ws,
                      adam_dev.data_ptr(), ndesc, nitems, dev.data_ptr(), len(views), ops._stream())
            for p, nt, nn in views:
                A.install_packed(p, nt, nn)
            return
        ops.adam_step_dev(self.flat_param, self.flat_grad, self.exp_avg, self.exp_avg_sq, self.step_dev, g["lr"],
                          tuple(g["betas"]), g["eps"], 1.0 / ws)
        self._repack()

    def _ensure_packs(self):
        """bf16 arena + device descriptors of this network's weights (allocated once, outside any graph capture)"""
        import ctypes  # noqa: F401
        from . import _lib
        if self._packs is None:
            ws2 = [p for p in self._params if p.dim() == 2]
            c16 = A.ceil16
            total = sum(p.shape[0] * c16(p.shape[1]) + p.shape[1] * c16(p.shape[0]) for p in ws2)
            arena = torch.empty(total + 8 * len(ws2) * 2, dtype=torch.bfloat16, device=self.flat_param.device)
            descs = (_lib.RepackDesc * max(1, len(ws2)))()
            views, off = [], 0
            for i, p in enumerate(ws2):
                N, K = p.shape
                Kp, Np = c16(K), c16(N)
                nt = arena[off:off + N * Kp].view(N, Kp); off += (N * Kp + 7) // 8 * 8
                nn = arena[off:off + K * Np].view(K, Np); off += (K * Np + 7) // 8 * 8
                descs[i].W, descs[i].nt, descs[i].nn = p.data_ptr(), nt.data_ptr(), nn.data_ptr()
                descs[i].N, descs[i].K, descs[i].Kp, descs[i].Np = N, K, Kp, Np
                views.append((p, nt, nn))
            dev = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).to(self.flat_param.device)
            # the same weights, and everything between them, as the work list of the one-launch step (ascending item0)
            by_ptr = {p.data_ptr(): (nt, nn) for p, nt, nn in views}
            ad, item0, base = [], 0, self.flat_param.data_ptr()
            for p in self._params:
                e = _lib.AdamDesc()
                e.off, e.len, e.item0 = (p.data_ptr() - base) // 4, p.numel(), item0
                if p.data_ptr() in by_ptr and p.dim() == 2:
                    nt, nn = by_ptr[p.data_ptr()]
                    e.N, e.K, e.Kp = p.shape[0], p.shape[1], c16(p.shape[1])
                    e.nt = nt.data_ptr()
                    item0 += (e.N * e.Kp + 4095) // 4096
                else:
                    e.N = e.K = e.Kp = 0
                    e.nt = None
                    item0 += (p.numel() + 4095) // 4096
                ad.append(e)
            arr = (_lib.AdamDesc * max(1, len(ad)))(*ad)
            adam_dev = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.flat_param.device)
            self._packs = (arena, dev, views, adam_dev, len(ad), item0)
        return self._packs

    def _repack(self):
        from . import _lib
        arena, dev, views = self._ensure_packs()[:3]
        if views:
            _lib.call("dhaug_repack_weights", dev.data_ptr(), len(views), ops._stream())
            for p, nt, nn in views:
                A.install_packed(p, nt, nn)

    @torch.no_grad()
    def flush(self):
        """finish a step whose exchange was started with overlap=True: wait for the all-reduce (on the stream, not the
        host, with RCCL) and launch Adam.  A no-op otherwise."""
        if self._pending is not None:
            work, ws = self._pending
            self._pending = None
            if work is None:                                   # started inside a segmented capture: the wait is an event too
                from . import graphs
                rec = graphs.RECORDER
                if rec is None or not hasattr(rec, "cut"):
                    raise RuntimeError("FusedAdam.flush: an all-reduce started inside a segmented capture is pending, but the "
                                       "capture is gone (an exception between step() and flush()?)")
                rec.cut(("wait", self))
            else:
                work.wait()
            self._apply(ws)
