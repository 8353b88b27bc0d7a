"""hipGraph capture of the training iteration (torch.cuda.CUDAGraph drives hipStreamBeginCapture / hipGraphLaunch).

One GAN iteration is a few hundred kernel launches of a few microseconds to tens of microseconds each; issued from Python
its wall time follows the host (ctypes call + allocator per launch) as soon as the kernels are short -- the video
configuration (B = 512 clips) is ~4 000 launches of 5-10 us.  Captured once and replayed, the iteration costs one
hipGraphLaunch on the host.  What makes the step replayable:
  * every kernel of libdhaug.so is enqueued on the caller's stream with no host synchronisation and no allocation;
  * the Adam step count lives on the device (dhaug_adam_step_dev) -- the bias corrections are not baked into arguments;
  * random draws come from torch's graph-safe device generator (noise, GP coefficients, bone-length jitter);
  * the weights' bf16 re-packing happens inside the captured region: before their first use (no packed copy made outside
    the capture is ever read by a captured kernel) and after every optimizer step;
  * inputs are copied into static buffers before each replay; the camera is a launch argument, so a graph is keyed by it.
Multi-rank runs (SegmentedCall): the collective stays eager, everything between two collective events is a graph -- the
iteration is captured as SEGMENTS cut where FusedAdam starts an all-reduce or waits for one:
    graph | all_reduce(d3 bucket, async) | graph | all_reduce(d2 bucket, async) | wait(d3) | graph (Adam d3, ...) | ...
so an N-rank iteration costs a handful of hipGraphLaunch calls + its collectives on the host, not ~300 (single-frame) or
~1 400 (video) kernel launches."""
import itertools

import torch

from . import autograd_ops as A

_capture_ids = itertools.count(1)


_STREAMS = {}


def capture_streams():
    """(warm-up stream, capture stream) of the current device -- made once, shared by every graph (captures are built one
    at a time), together with everything the captured code forks to from the capture stream: no stream is created inside a
    capture, and torch's stream pool (32 handles, handed out round robin) is not walked through by building graphs."""
    from . import critic_step as CS, gen_step
    from .models_Fk_GAN import model_fk_gan_train as train
    dev = torch.cuda.current_device()
    if dev not in _STREAMS:
        warm, cap = torch.cuda.Stream(), torch.cuda.Stream()
        CS.tn_side_stream(cap)
        train._side_streams(4)
        gen_step.side_streams(cap, 8)
        _STREAMS[dev] = (warm, cap)
    return _STREAMS[dev]


class _CaptureRoot:
    """while a capture runs: critic_step.CAPTURE_ROOT names the stream it was begun on.  hipStreamEndCapture of this HIP
    release survives ONE fork level; it belongs to code running directly on that stream (critic_step.can_split,
    gen_step._parallel, the concurrent critics of run_critic_steps)."""

    def __init__(self, stream):
        self.h = stream.cuda_stream

    def __enter__(self):
        from . import critic_step as CS
        self.old, CS.CAPTURE_ROOT = CS.CAPTURE_ROOT, self.h

    def __exit__(self, *exc):
        from . import critic_step as CS
        CS.CAPTURE_ROOT = self.old


class GraphedCall:
    """capture fn(*static_inputs) once (after warm-up calls that populate caches / one-time kernel configuration) and replay it"""

    def __init__(self, fn, example_inputs, warmup=2, prologue=None, state=()):
        """state: tensors the warm-up calls modify in place (weights, Adam moments, step counts): they are put back before
        the capture, so building a graph advances nothing -- the first replay is the first iteration."""
        self.static_in = [t.clone() if torch.is_tensor(t) else t for t in example_inputs]
        saved = [t.clone() for t in state]
        side, self.cap = capture_streams()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                fn(*self.static_in)
            for t, s in zip(state, saved):
                t.copy_(s)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        if saved:
            A.bump_weight_epoch()                                  # packed copies made from the warm-up's weights are stale
        self.graph = torch.cuda.CUDAGraph()
        # every cached bf16 / fragment copy of a weight is re-made INSIDE the capture (see autograd_ops.CAPTURE_ID): the
        # graph then owns the memory its kernels read and refreshes it on every replay
        A.CAPTURE_ID = next(_capture_ids)
        try:
            with _CaptureRoot(self.cap), torch.cuda.graph(self.graph, stream=self.cap):
                if prologue is not None:
                    prologue()
                self.out = fn(*self.static_in)
        finally:
            A.CAPTURE_ID = 0

    def __call__(self, *inputs):
        for dst, src in zip(self.static_in, inputs):
            if torch.is_tensor(dst):
                dst.copy_(src, non_blocking=True)
        self.graph.replay()
        return self.out


RECORDER = None          # the SegmentedCall / ForkedCall being captured, if any: FusedAdam cuts the capture at its collective
                         # events, run_critic_steps forks it where independent chains begin
import os as _os
FORKED = _os.environ.get("DHAUG_NO_FORKED_GRAPHS") is None


class SegmentedCall:
    """fn(*static_inputs) as a sequence of captured graphs with the data-parallel collectives between them.

    While fn is captured, optim.FusedAdam calls cut(("allreduce", opt)) where it would start the all-reduce of its gradient
    bucket and cut(("wait", opt)) where it would wait for it: the running capture ends, the event is recorded, a new capture
    begins (same memory pool: tensors alive across a cut stay where they are).  Replay = graphs and events in order; the
    all-reduce is issued on the replaying stream's timeline exactly as the eager path issues it (async_op; wait() makes the
    stream wait, not the host, with RCCL)."""

    def __init__(self, fn, example_inputs, warmup=2, prologue=None, state=()):
        import torch.distributed as dist
        self.dist = dist
        self.static_in = [t.clone() if torch.is_tensor(t) else t for t in example_inputs]
        saved = [t.clone() for t in state]
        side, cap = capture_streams()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):                                # (eager, collectives included: every rank runs the same)
                fn(*self.static_in)
            for t, s in zip(state, saved):
                t.copy_(s)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        if saved:
            A.bump_weight_epoch()
        global RECORDER
        self.items, self.pool, self.cap = [], torch.cuda.graph_pool_handle(), cap
        self._g = None
        A.CAPTURE_ID = next(_capture_ids)
        RECORDER = self
        try:
            self.cap.wait_stream(torch.cuda.current_stream())
            with _CaptureRoot(self.cap), torch.cuda.stream(self.cap):
                self._begin()
                if prologue is not None:
                    prologue()
                self.out = fn(*self.static_in)
                self._end()
            torch.cuda.current_stream().wait_stream(self.cap)
        finally:
            RECORDER = None
            A.CAPTURE_ID = 0
            if self._g is not None:                                # an exception inside the capture: end it, keep the error
                try:
                    self._g.capture_end()
                except Exception:
                    pass
                self._g = None
        # every all-reduce that was started is waited for inside the same iteration (an optimizer left with a pending
        # exchange -- overlap=True and no flush() -- would replay a collective nobody waits for)
        open_ = []
        for kind, obj in self.items:
            if kind == "allreduce":
                assert id(obj) not in open_, "two all-reduces of one optimizer without a wait between them"
                open_.append(id(obj))
            elif kind == "wait":
                assert id(obj) in open_, "a wait without its all-reduce in the captured iteration"
                open_.remove(id(obj))
        assert not open_, "captured iteration ends with %d all-reduce(s) nobody waits for (FusedAdam.flush() missing)" % len(open_)
        self.work = {}

    def _begin(self):
        self._g = torch.cuda.CUDAGraph()
        self._g.capture_begin(pool=self.pool)

    def _end(self):
        self._g.capture_end()
        self.items.append(("graph", self._g))
        self._g = None

    def cut(self, event):
        """called by FusedAdam during the capture: event = ("allreduce" | "wait", optimizer)"""
        self._end()
        self.items.append(event)
        self._begin()

    def __call__(self, *inputs):
        for dst, src in zip(self.static_in, inputs):
            if torch.is_tensor(dst):
                dst.copy_(src, non_blocking=True)
        for kind, obj in self.items:
            if kind == "graph":
                obj.replay()
            elif kind == "allreduce":
                self.work[id(obj)] = self.dist.all_reduce(obj.flat_grad, op=self.dist.ReduceOp.SUM, group=obj.process_group, async_op=True)
            else:
                self.work.pop(id(obj)).wait()
        return self.out


class ForkedCall:
    """fn(*static_inputs) as captured graphs in which INDEPENDENT CHAINS are graphs of their own, replayed side by side.

    The branches of ONE hipGraph do overlap on this HIP release (sweep 4's side branch is worth 0.38 ms of a 7.4 ms single-frame
    iteration), but four long chains of short kernels -- the video iteration's four critics, ~1 400 nodes -- replay faster as four
    graphs launched on four streams than as four branches of one graph: 18.9 -> 18.05 ms per iteration (measured).  So while fn
    is captured, run_critic_steps hands the chains of independent work to fork(): the running capture ends, every chain is
    captured on its own stream into its own graph (and its own memory pool: the chains' graphs are in flight together), a
    new capture begins behind them.  Replay: graph | event | the chains' graphs on their streams | join | graph ..."""

    def __init__(self, fn, example_inputs, warmup=2, prologue=None, state=()):
        self.static_in = [t.clone() if torch.is_tensor(t) else t for t in example_inputs]
        saved = [t.clone() for t in state]
        side, cap = capture_streams()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                fn(*self.static_in)
            for t, s in zip(state, saved):
                t.copy_(s)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        if saved:
            A.bump_weight_epoch()
        global RECORDER
        self.items, self.pool, self.cap = [], torch.cuda.graph_pool_handle(), cap
        self._g = None
        A.CAPTURE_ID = next(_capture_ids)
        RECORDER = self
        try:
            self.cap.wait_stream(torch.cuda.current_stream())
            with _CaptureRoot(self.cap), torch.cuda.stream(self.cap):
                self._begin()
                if prologue is not None:
                    prologue()
                self.out = fn(*self.static_in)
                self._end()
            torch.cuda.current_stream().wait_stream(self.cap)
        finally:
            RECORDER = None
            A.CAPTURE_ID = 0

    def _begin(self):
        self._g = torch.cuda.CUDAGraph()
        self._g.capture_begin(pool=self.pool)

    def _end(self):
        self._g.capture_end()
        self.items.append(("graph", self._g))
        self._g = None

    def fork(self, chains):
        """chains: [(stream, fn)] -- independent of each other, dependent only on what was captured so far.  Called on the
        capture stream; returns [fn()]."""
        from . import critic_step as CS
        self._end()
        subs, outs = [], []
        root, CS.CAPTURE_ROOT = CS.CAPTURE_ROOT, None          # (a chain's stream is not the stream a fork may start from)
        try:
            for st, fn in chains:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.stream(st):
                    g.capture_begin(pool=torch.cuda.graph_pool_handle())
                    outs.append(fn())
                    g.capture_end()
                subs.append((st, g))
        finally:
            CS.CAPTURE_ROOT = root
        self.items.append(("fork", subs))
        self._begin()
        return outs

    def __call__(self, *inputs):
        for dst, src in zip(self.static_in, inputs):
            if torch.is_tensor(dst):
                dst.copy_(src, non_blocking=True)
        cur = torch.cuda.current_stream()
        for kind, obj in self.items:
            if kind == "graph":
                obj.replay()
            else:
                for st, g in obj:
                    st.wait_stream(cur)
                    with torch.cuda.stream(st):
                        g.replay()
                for st, _ in obj:
                    cur.wait_stream(st)
        return self.out


class GraphedGanIteration:
    """gan_iteration / video_gan_iteration behind hipGraphs: one graph per (camera, G-step or not).  Call like the eager
    function; the returned tensors live in the graph's static memory (copy what must outlive the next call)."""

    def __init__(self, iteration_fn, args, poseFk_dict, train_subjects, summary=None):
        self.fn, self.args, self.d, self.subj, self.summary = iteration_fn, args, poseFk_dict, train_subjects, summary
        self.graphs = {}

    def __call__(self, inputs_3d, cam_param, inputs_2d, do_g_step, camera, draws=None):
        """draws: None (the iteration draws on the device: graph-safe generator), or a ConstDraws whose DEVICE tensors are
        baked into the graph (it must outlive the graph and keep its values)."""
        return self.prepare(inputs_3d, cam_param, inputs_2d, do_g_step, camera, draws)(inputs_3d, cam_param, inputs_2d)

    def prepare(self, inputs_3d, cam_param, inputs_2d, do_g_step, camera, draws=None, warmup=2):
        """the captured form of this call (captured now if it is not yet), WITHOUT replaying it.  warmup = 0: no eager
        iteration in front of the capture -- with several ranks the capture then issues NO collective (FusedAdam only records
        where its all-reduces go), so a rank on which the capture fails leaves no peer waiting inside one (bench.py's
        calibration: every rank captures, all agree on the outcome, only then anything with a collective in it is replayed)."""
        key = (bool(do_g_step), tuple(camera[0]), tuple(camera[1]), tuple(camera[2]),
               tuple(inputs_3d.shape), tuple(cam_param.shape), tuple(inputs_2d.shape), id(draws))
        g = self.graphs.get(key)
        if g is None:
            def run(x3, cp, x2):
                return self.fn(self.args, self.d, x3, cp, x2, self.subj, self.summary, None, do_g_step=do_g_step, camera=camera,
                               draws=draws)
            # first thing in the graph: every network's bf16 operand copies, two launches per network (the lazy per-layer
            # packing would put ~50 small kernels there)
            opts = [v for k, v in self.d.items() if k.startswith("optimizer")]
            for o in opts:
                o._ensure_packs()                              # (allocation + descriptor upload must not happen in a capture)
            state = [t for o in opts for t in (o.flat_param, o.exp_avg, o.exp_avg_sq, o.step_dev)]
            counts = [o.step_count for o in opts]
            multi = any(o.world_size() > 1 for o in opts)          # collectives between graph segments
            g = self.graphs[key] = (SegmentedCall if multi else (ForkedCall if FORKED else GraphedCall))(
                run, (inputs_3d, cam_param, inputs_2d), warmup=warmup, prologue=lambda: [o._repack() for o in opts], state=state)
            for o, c in zip(opts, counts):                     # (host-side bookkeeping of the warm-up calls)
                o.step_count = c
        return g
