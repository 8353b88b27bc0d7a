"""Fused one-launch forward of the generator trunk and the critics (dhaug_mlp_forward).

A network is compiled into a short program of units (include/dhaug.h, struct dhaug_mlp_unit) over three LDS
activation buffers; its weights are re-packed into MFMA fragment order (dhaug_pack_wfrag) whenever a parameter
changes (the optimizer step invalidates the cache).  Used for the no-grad passes of the hot path: sampling fakes
(model_fk_gan_train.py:305-308 `.data`), the flipped critic evaluations of the G step (:463-468) and inference.
Same arithmetic as the layer-by-layer bf16 path (bf16 operands, fp32 accumulate, bf16 activations)."""
import ctypes
import os

import torch

from . import _lib, ops
from .autograd_ops import ACT_LRELU, ACT_NONE, ACT_RELU

LOAD_F32, LOAD_BF16, STORE_BF16, GEMM, LOAD_KCS = 0, 1, 2, 3, 5
F_OUT_F32, F_DOT_OUT, F_T16 = 4, 16, 32
_vp = ctypes.c_void_p
# arithmetic of a fused program: "bf16" (dhaug_mlp_forward: one bf16 MFMA pass, bf16 activations in LDS) or "f16x3"
# (dhaug_mlp_forward_x3: operands as fp16 hi + lo pairs, three MFMA terms, fp32-grade -- the mode that meets the path's
# 1e-4 logit tolerance against the fp32 reference)
MODES = ("bf16", "f16x3")
X3_WORKSPACE_BYTES = 2 * 256 * 4 * 64 * 128 * 4               # DHAUG_MLP_X3_WORKSPACE_BYTES of include/dhaug.h
SIGN_BITS = os.environ.get("DHAUG_NO_SIGN_BITS") is None      # forward-with-save also emits (y > 0) bit arrays of its run layers


def supported(*dims):
    """hidden widths the fused kernel has layer shapes for (1, 2 or 4 chunks of 64 columns); other widths run layer by
    layer"""
    return all(d in (64, 128, 256) for d in dims)


class _Layer:
    """packed fragments + padded bias of one nn.Linear (optionally split over two input column ranges)"""

    def __init__(self, lin, splits=None, mode="bf16", t16=False):
        W, b = lin.weight.detach(), lin.bias.detach()
        self.t16 = bool(t16) and mode == "f16x3"             # fragments in the order of v_mfma_f32_16x16x32_f16 (DHAUG_MLP_F_T16)
        N, K = W.shape
        self.N = N
        splits = splits or [(0, K)]
        self.w, self.ksteps = [], []
        for k0, k in splits:
            ks = (k + 15) // 16
            kpad = (k + 63) // 64 * 4                         # k-steps padded to whole 64-wide chunks
            if mode == "f16x3":                               # hi and lo fp16 fragments per (slice, k-step)
                blob = torch.empty(2 * 8 * kpad * 512, dtype=torch.float16, device=W.device)
                _lib.call("dhaug_pack_wfrag_f16x2_t16" if self.t16 else "dhaug_pack_wfrag_f16x2", _vp(W.data_ptr()), K,
                          _vp(blob.data_ptr()), N, k, k0, ops._stream())
            else:
                blob = torch.empty(8 * kpad * 512, dtype=torch.bfloat16, device=W.device)      # always 8 slices
                _lib.call("dhaug_pack_wfrag", _vp(W.data_ptr()), K, _vp(blob.data_ptr()), N, k, k0, ops._stream())
            self.w.append(blob)
            self.ksteps.append(ks)
        self.bias = torch.zeros(256, dtype=torch.float32, device=W.device)
        self.bias[:N] = b
        self.zero = torch.zeros(256, dtype=torch.float32, device=W.device) if len(splits) > 1 else None
        if N == 1 and mode == "bf16":                         # logit layer folded into its producer (DHAUG_MLP_F_DOT_OUT)
            self.dot = torch.zeros(260, dtype=torch.float32, device=W.device)
            self.dot[:K] = W[0].to(torch.bfloat16).float()
            self.dot[256] = b[0]
        self.lin, self.splits = lin, splits

    def descs(self):
        """dhaug_wfrag_desc entries that re-pack this layer in place from its (updated) parameters"""
        W, b = self.lin.weight, self.lin.bias
        out = []
        for i, ((k0, k), blob, ks) in enumerate(zip(self.splits, self.w, self.ksteps)):
            d = _lib.WfragDesc()
            d.W, d.ldw, d.dst, d.N, d.K, d.k0, d.ksteps = W.data_ptr(), W.shape[1], blob.data_ptr(), self.N, k, k0, (k + 63) // 64 * 4
            d.bias = b.data_ptr()
            d.bias_dst = self.bias.data_ptr() if i == 0 else None
            d.dot_dst = self.dot.data_ptr() if (i == 0 and hasattr(self, "dot")) else None
            out.append(d)
        return out


def _unit(kind, flags=0, src=-1, dst=-1, res=-1, src2=-1, ksteps2=0, ksteps=0, n=0, act=0, slope=0.0, cols=0, ld=0,
          g=None, w=None, w2=None, bias=None, save=None, bits=False, save_rows=0, unmasked=False):
    u = _lib.MlpUnit()
    if save is not None:                                      # forward-with-save: the layer's image also goes to `save`
        assert save.dtype == torch.bfloat16 and save.stride(1) == 1
        u.save, u.save_ld = save.data_ptr(), save.stride(0)
        # rows [0, save_rows) only (0: all, < 0: none) -- honoured only where the sign bits are written too: whoever asks for
        # fewer rows reads the other rows' masks from the bits
        # (or where the layer has no activation -- `unmasked`: no mask is ever read from its image)
        u.save_rows = int(save_rows) if ((bits and SIGN_BITS) or unmasked) else 0
        if bits and SIGN_BITS:
            # the layer also leaves (y > 0) as one bit per element (struct dhaug_mlp_unit.bits): the backward / tangent sweeps
            # read that instead of the bf16 image (critic_step.py); the array rides on the saved tensor
            save._dhaug_bits = new_bits(save.shape[0], save.device)
            u.bits = save._dhaug_bits.data_ptr()
    u.kind, u.flags, u.src, u.dst, u.res, u.src2, u.ksteps2 = kind, flags, src, dst, res, src2, ksteps2
    u.ksteps, u.n, u.act, u.slope, u.cols, u.ld = ksteps, n, act, float(slope), cols, ld
    u.g = None if g is None else g.data_ptr()
    u.w = None if w is None else w.data_ptr()
    u.w2 = None if w2 is None else w2.data_ptr()
    u.bias = None if bias is None else bias.data_ptr()
    return u


def new_bits(M, dev):
    """sign-bit array of an (M, 256) activation: one dword per lane and 32-row tile, padded to whole 128-row tiles"""
    return torch.empty(((M + 127) // 128 * 4 * 4 * 64,), dtype=torch.int32, device=dev)


def decode_bits(bits, M):
    """(M, 256) bool from a sign-bit array (layout: struct dhaug_mlp_unit.bits in include/dhaug.h) -- tests / debugging"""
    w = bits.reshape(-1, 4, 64).cpu().numpy().astype("uint32")           # [T][wave][lane]
    import numpy as np
    T = w.shape[0]
    out = np.zeros((T * 32, 256), dtype=bool)
    lane = np.arange(64)
    r31, h = lane & 31, lane >> 5
    for wave in range(4):
        for j in range(32):
            t, g, e = j >> 4, (j >> 2) & 3, j & 3
            pos = (j >> 1) + (16 if (j & 1) else 0)
            feat = 32 * (wave + 4 * t) + 8 * g + 4 * h + e                  # per lane
            val = (w[:, wave, :] >> pos) & 1                                # [T][lane]
            for tt in range(T):
                out[32 * tt + r31, feat] = val[tt].astype(bool)
    return torch.from_numpy(out[:M])


def encode_bits(mask):
    """sign-bit array (layout of struct dhaug_mlp_unit.bits) of an (M, 256) bool tensor -- the inverse of decode_bits, for tests"""
    import numpy as np
    m = mask.cpu().numpy().astype(bool)
    M = m.shape[0]
    T = (M + 127) // 128 * 4
    pad = np.zeros((T * 32, 256), dtype=bool)
    pad[:M] = m
    w = np.zeros((T, 4, 64), dtype=np.uint32)
    lane = np.arange(64)
    r31, h = lane & 31, lane >> 5
    for wave in range(4):
        for j in range(32):
            t, g, e = j >> 4, (j >> 2) & 3, j & 3
            pos = (j >> 1) + (16 if (j & 1) else 0)
            feat = 32 * (wave + 4 * t) + 8 * g + 4 * h + e
            for tt in range(T):
                w[tt, wave, :] |= pad[32 * tt + r31, feat].astype(np.uint32) << np.uint32(pos)
    return torch.from_numpy(w.reshape(-1).view(np.int32).copy()).to(mask.device)


def _gemm(layer, src, dst, act, slope=0.0, res=-1, out=None, src2=-1, save=None, bits=False, save_rows=0):
    kw = dict(src=src, ksteps=layer.ksteps[0], n=layer.N, act=act, slope=slope, w=layer.w[0], bias=layer.bias, res=res, save=save,
              bits=bits, save_rows=save_rows, unmasked=(act == ACT_NONE and save is not None))
    if len(layer.w) == 2:
        kw.update(src2=src2, ksteps2=layer.ksteps[1], w2=layer.w[1])
    t16 = F_T16 if getattr(layer, "t16", False) else 0
    if out is not None:
        return _unit(GEMM, flags=F_OUT_F32 | t16, g=out, ld=out.stride(0), dst=dst, **kw)
    return _unit(GEMM, flags=t16, dst=dst, **kw)


class FusedNet:
    """compiled program + packed weights of one module; rebuilt when any parameter changes"""

    def __init__(self, module, build, mode="bf16"):
        assert mode in MODES
        self.module, self.build, self.key, self.layers, self.params, self.mode = module, build, None, None, None, mode
        self.ptrs, self.descs_dev, self.ndescs = None, None, 0

    def _fresh(self):
        from . import autograd_ops as A
        if self.params is None:                                # (walking the module tree costs more than the launch)
            self.params = list(self.module.parameters())
        key = (A.WEIGHT_EPOCH, A.CAPTURE_ID) + tuple((p.data_ptr(), p._version, getattr(p, "_dhaug_epoch", 0)) for p in self.params)
        if key != self.key:
            ptrs = tuple(p.data_ptr() for p in self.params)
            if self.layers is not None and self.mode == "bf16" and ptrs == self.ptrs:
                # the same tensors with new values (an optimizer step): every blob, bias and logit vector is re-packed in
                # place by ONE launch (inside a hipGraph capture too: the graph replays the re-pack, and owns nothing new)
                _lib.call("dhaug_pack_wfrag_batch", self.descs_dev.data_ptr(), self.ndescs, ops._stream())
            else:
                self.layers = {name: _Layer(lin, splits, self.mode, self.build.get("t16", False))
                               for name, lin, splits in self.build["layers"](self.module)}
                if self.mode == "bf16":
                    ds = [d for L in self.layers.values() for d in L.descs()]
                    arr = (_lib.WfragDesc * len(ds))(*ds)
                    dev = next(iter(self.layers.values())).bias.device
                    self.descs_dev = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
                    self.ndescs = len(ds)
                self.ptrs = ptrs
            self.key = key
        return self.layers

    def invalidate(self):
        self.key = None
        self.params = None
        self.layers = None

    def run(self, inputs, M):
        L = self._fresh()
        units, keep = self.build["program"](self.module, L, inputs, M)
        launch(units, M, self.mode)
        return keep


def launch(units, M, mode):
    if mode == "f16x3":
        # residuals and parked partial sums wait in a global workspace (DHAUG_MLP_X3_WORKSPACE_BYTES, include/dhaug.h).  One per
        # launch from the caching allocator, i.e. stream-ordered: launches on different streams never share one, and a
        # hipGraph capture gets its own from the graph's pool.
        dev = torch.device("cuda", torch.cuda.current_device())
        ws = torch.empty(X3_WORKSPACE_BYTES, dtype=torch.uint8, device=dev)
        for u in units:
            if u.kind == GEMM and not (u.flags & F_OUT_F32):
                u.g = ws.data_ptr()
    arr = (_lib.MlpUnit * len(units))(*units)
    _lib.call("dhaug_mlp_forward" if mode == "bf16" else "dhaug_mlp_forward_x3", arr, len(units), M, ops._stream())


def _net(module, build, mode, tag=""):
    """the module's compiled program for `mode` (one per arithmetic [and purpose: tag], cached on the module)"""
    d = module.__dict__.setdefault("_fused", {})
    if mode + tag not in d:
        d[mode + tag] = FusedNet(module, build, mode)
    return d[mode + tag]


def _res_blocks(L, units, names, a=0, b=1, act=ACT_RELU):
    """three myResNet blocks: x in buffer a -> result in buffer a (h in b, in-place residual)"""
    for n in names:
        units.append(_gemm(L[n + ".fc1"], a, b, act))
        units.append(_gemm(L[n + ".fc2"], b, a, act, res=a))


# ---- generator trunk: z (B,128) fp32 -> head (B,35*R) fp32 ------------------------------------------------------
def _gen_layers(G):
    out = [("preprocess.0", G.preprocess[0], None)]
    for b in ("block1", "block2", "block3"):
        blk = getattr(G, b)
        out += [(b + ".fc1", blk.fc1, None), (b + ".fc2", blk.fc2, None)]
    out.append(("deconv_out", G.deconv_out, None))
    return out


def _gen_program(G, L, inputs, M):
    z = inputs["z"]
    head = torch.empty((M, G.deconv_out.weight.shape[0]), dtype=torch.float32, device=z.device)
    u = [_unit(LOAD_F32, dst=1, cols=z.shape[1], ld=z.stride(0), g=z), _gemm(L["preprocess.0"], 1, 0, ACT_RELU)]
    _res_blocks(L, u, ("block1", "block2", "block3"))
    u.append(_gemm(L["deconv_out"], 0, 1, ACT_NONE, out=head))
    return u, head


# (f16x3: which matrix instruction a network's parity program runs on.  The critics take v_mfma_f32_16x16x32_f16 -- the chip
# holds a higher clock on it: 283 us against 301 for the 3D critic.  The generator trunk keeps 32 x 32 x 16: its head feeds a
# 10 tanh root, and on the first golden set the pose is 8.9e-6 m from the reference's in that order of summation and 1.05e-5 in
# the other (which is the one closer to the exact result: 8.5e-6 against 9.2e-6 m) -- the bound is 1e-5.)
GEN = dict(layers=_gen_layers, program=_gen_program, t16=False)


# ---- 2D critic: x (B,32) fp32 -> logit (B,1) ----------------------------------------------------------------------
def _d2_layers(D):
    return [(n, getattr(D, n), None) for n in ("pose_layer_1", "pose_layer_2", "pose_layer_3", "pose_layer_4",
                                                "layer_last", "layer_pred")]


def _d2_program(D, L, inputs, M):
    x = inputs["x"]
    out = torch.empty((M, 1), dtype=torch.float32, device=x.device)
    s = D.slope
    u = [_unit(LOAD_BF16 if x.dtype == torch.bfloat16 else LOAD_F32, dst=1, cols=32, ld=x.stride(0), g=x),
         _gemm(L["pose_layer_1"], 1, 0, ACT_LRELU, s),                 # d1 -> 0
         _gemm(L["pose_layer_2"], 0, 1, ACT_LRELU, s),                 # d2 -> 1
         _gemm(L["pose_layer_3"], 1, 0, ACT_LRELU, s, res=0),          # d3 = lrelu(L3 d2 + d1) -> 0 (in place)
         _gemm(L["pose_layer_4"], 0, 1, ACT_NONE),                     # d4 -> 1
         _gemm(L["layer_last"], 1, 0, ACT_LRELU, s),
         _gemm(L["layer_pred"], 0, 1, ACT_NONE, out=out)]
    return u, out


D2 = dict(layers=_d2_layers, program=_d2_program, t16=True)


# ---- 3D critic: pose (B,48) fp32 + KCS (B,32) bf16 -> logit (B,1) ------------------------------------------------
def _d3_layers(D):
    out = [("special_KCS_previous.0", D.special_KCS_previous[0], None), ("previous.0", D.previous[0], None)]
    for b in ("special_KCS_block1", "special_KCS_block2", "special_KCS_block3", "block1", "block2", "block3",
              "merge_block1"):
        blk = getattr(D, b)
        out += [(b + ".fc1", blk.fc1, None), (b + ".fc2", blk.fc2, None)]
    Dd = D.previous[0].weight.shape[0]
    out.append(("merge_previous.0", D.merge_previous[0], [(0, Dd), (Dd, Dd)]))
    out.append(("output", D.output, None))
    return out


def _d3_program(D, L, inputs, M):
    x, kcs = inputs["x"], inputs["kcs"]
    out = torch.empty((M, 1), dtype=torch.float32, device=x.device)
    mp = L["merge_previous.0"]
    if kcs is None:                                          # f16x3: the features are computed from the poses inside the launch
        assert x.dtype == torch.float32
        kload = _unit(LOAD_KCS, dst=1, ld=x.stride(0), g=x)
    else:
        kload = (_unit(LOAD_BF16, dst=1, cols=32, ld=kcs.stride(0), g=kcs) if kcs.dtype == torch.bfloat16 else
                 _unit(LOAD_F32, dst=1, cols=kcs.shape[1], ld=kcs.stride(0), g=kcs))      # f16x3: the fp32 (N,30) features
    u = [kload, _gemm(L["special_KCS_previous.0"], 1, 0, ACT_RELU)]
    _res_blocks(L, u, ("special_KCS_block1", "special_KCS_block2", "special_KCS_block3"))
    # cat(kcs_out, pos_out) -> merge layer, in two halves: LDS cannot hold the KCS branch's 64 KB result beside the two
    # images the pose branch needs, so its share of the merge layer's pre-activation (W[:, :D] kcs_out + bias, 100 wide)
    # waits in buffer 2 as bf16 and the pose branch's share is added to it as a residual (one extra bf16 rounding of a
    # partial sum; the alternative was a 33.5 MB round trip through L2)
    # (f16x3: that share waits in the launch's global workspace instead -- see launch())
    t16 = F_T16 if getattr(mp, "t16", False) else 0
    u.append(_unit(GEMM, flags=t16, src=0, dst=2, ksteps=mp.ksteps[0], n=mp.N, act=ACT_NONE, w=mp.w[0], bias=mp.bias))
    u += [_unit(LOAD_BF16 if x.dtype == torch.bfloat16 else LOAD_F32, dst=1, cols=48, ld=x.stride(0), g=x),
          _gemm(L["previous.0"], 1, 0, ACT_RELU)]
    _res_blocks(L, u, ("block1", "block2", "block3"))
    u.append(_unit(GEMM, flags=t16, src=0, dst=2, res=2, ksteps=mp.ksteps[1], n=mp.N, act=ACT_RELU, w=mp.w[1], bias=mp.zero))
    u.append(_gemm(L["merge_block1.fc1"], 2, 0, ACT_RELU))
    fc2 = _gemm(L["merge_block1.fc2"], 0, 1, ACT_RELU, res=2)
    if hasattr(L["output"], "dot"):
        # the logit layer (100 -> 1) rides in the epilogue of the layer before it
        fc2.flags, fc2.g, fc2.ld, fc2.w2 = F_DOT_OUT, out.data_ptr(), out.stride(0), L["output"].dot.data_ptr()
        u.append(fc2)
    else:                                                    # f16x3: the logit layer is a unit of its own
        u += [fc2, _gemm(L["output"], 1, 0, ACT_NONE, out=out)]
    return u, (out,)


D3 = dict(layers=_d3_layers, program=_d3_program, t16=True)


# ---- forward-with-save: the explicit critic step's forward sweep as ONE launch -----------------------------------
# (critic_step.py: sweep 1 of 4.  Every layer's output also goes to global memory -- the saved activations of the
# reference's autograd graph, R/models_Fk_GAN/model_fk_gan_train.py:177-230 -- but no layer reads its input back from there.)
def _empty16(M, n, dev):
    return torch.empty((M, (n + 15) // 16 * 16), dtype=torch.bfloat16, device=dev)


def _d3s_program(D, L, inputs, M):
    x, kcs = inputs["x"], inputs["kcs"]                       # (M,48) fp32, (M,32) bf16 KCS operand
    dev, Dw = x.device, D.previous[0].weight.shape[0]
    mp = L["merge_previous.0"]
    cat = torch.empty((M, 2 * Dw), dtype=torch.bfloat16, device=dev)
    y = [[_empty16(M, Dw, dev) for _ in range(3)] + [cat[:, b * Dw:(b + 1) * Dw]] for b in range(2)]
    h = [[_empty16(M, Dw, dev) for _ in range(3)] for _ in range(2)]
    m0, mh, m1 = _empty16(M, mp.N, dev), _empty16(M, mp.N, dev), _empty16(M, mp.N, dev)
    logits = torch.empty((M, 1), dtype=torch.float32, device=dev)

    # save_rows: rows of the BLOCK layers' outputs that are written as bf16 images (0: all).  The critic step passes 2B of its
    # 3B rows: the interpolated rows' activations are read by nothing -- their masks are the sign bits, their rows of the
    # buffers receive the tangents (critic_step.py) -- and the G step passes -1 (it only needs masks).  The 100-wide top keeps
    # every row: its masks are read as images.
    sr = inputs.get("save_rows", 0)

    def branch(u, b, first, names):
        u.append(_gemm(L[first], 1, 0, ACT_RELU, save=y[b][0], bits=True, save_rows=sr))
        for i, n in enumerate(names):
            u.append(_gemm(L[n + ".fc1"], 0, 1, ACT_RELU, save=h[b][i], bits=True, save_rows=sr))
            u.append(_gemm(L[n + ".fc2"], 1, 0, ACT_RELU, res=0, save=y[b][i + 1], bits=True, save_rows=sr))

    u = [_unit(LOAD_BF16, dst=1, cols=32, ld=kcs.stride(0), g=kcs)]
    branch(u, 0, "special_KCS_previous.0", ("special_KCS_block1", "special_KCS_block2", "special_KCS_block3"))
    u.append(_unit(GEMM, src=0, dst=2, ksteps=mp.ksteps[0], n=mp.N, act=ACT_NONE, w=mp.w[0], bias=mp.bias))      # (see _d3_program)
    u.append(_unit(LOAD_F32, dst=1, cols=48, ld=x.stride(0), g=x))
    branch(u, 1, "previous.0", ("block1", "block2", "block3"))
    u.append(_unit(GEMM, src=0, dst=2, res=2, ksteps=mp.ksteps[1], n=mp.N, act=ACT_RELU, w=mp.w[1], bias=mp.zero, save=m0))
    u.append(_gemm(L["merge_block1.fc1"], 2, 0, ACT_RELU, save=mh))
    u.append(_gemm(L["merge_block1.fc2"], 0, 1, ACT_RELU, res=2, save=m1))
    u.append(_gemm(L["output"], 1, 0, ACT_NONE, out=logits))
    return u, dict(cat=cat, y=y, h=h, m0=m0, mh=mh, m1=m1, logits=logits)


def _d3s_layers(D):
    out = _d3_layers(D)
    return out


D3S = dict(layers=_d3s_layers, program=_d3s_program)


def _d2s_program(D, L, inputs, M):
    x = inputs["x"]                                           # (M,32) fp32
    dev, Dw, s = x.device, D.pose_layer_1.weight.shape[0], D.slope
    d = [_empty16(M, Dw, dev) for _ in range(5)]
    logits = torch.empty((M, 1), dtype=torch.float32, device=dev)
    u = [_unit(LOAD_F32, dst=1, cols=32, ld=x.stride(0), g=x),
         _gemm(L["pose_layer_1"], 1, 0, ACT_LRELU, s, save=d[0], bits=True, save_rows=inputs.get("save_rows", 0)),
         _gemm(L["pose_layer_2"], 0, 1, ACT_LRELU, s, save=d[1], bits=True, save_rows=inputs.get("save_rows", 0)),    # (see _d3s_program)
         _gemm(L["pose_layer_3"], 1, 0, ACT_LRELU, s, res=0, save=d[2], bits=True, save_rows=inputs.get("save_rows", 0)),
         _gemm(L["pose_layer_4"], 0, 1, ACT_NONE, save=d[3], save_rows=inputs.get("save_rows", 0)),
         _gemm(L["layer_last"], 1, 0, ACT_LRELU, s, save=d[4], bits=True, save_rows=inputs.get("save_rows", 0)),
         _gemm(L["layer_pred"], 0, 1, ACT_NONE, out=logits)]
    return u, dict(d=d, logits=logits)


D2S = dict(layers=_d2_layers, program=_d2s_program)


def step_forward_supported(D):
    """the critics whose explicit step can take its forward sweep from one fused launch: the single-frame critics at a
    hidden width the kernel has shapes for"""
    from .models_Fk_GAN.Fk_discriminator import Fk_2D_Discriminator, Fk_3D_Discriminator
    if type(D) is Fk_3D_Discriminator:
        return supported(D.previous[0].weight.shape[0]) and D.previous[0].weight.shape[0] == 256 and D.merge_previous[0].weight.shape[0] <= 128
    if type(D) is Fk_2D_Discriminator:
        return D.pose_layer_1.weight.shape[0] == 256
    return False


def critic3d_forward_save(D, x, kcs, save_rows=0):
    """x (M,48) fp32 root-relative poses, kcs (M,32) bf16 operand -> saved activations + logits (see _d3s_program)"""
    return _net(D, D3S, "bf16", "+save").run(dict(x=x, kcs=kcs, save_rows=save_rows), x.shape[0])


def critic2d_forward_save(D, x, save_rows=0):
    return _net(D, D2S, "bf16", "+save").run(dict(x=x, save_rows=save_rows), x.shape[0])


def partial_save_ok(rows):
    """may a step whose batch is made of `rows`-row parts (real | fake | interpolated; or the G step's one part) ask the
    forward-with-save programs to leave some parts' block-layer images unwritten?  Only if every consumer of those rows' masks
    reads the sign bits: bits written and consumed, and every part -- hence every launch over 1, 2 or 3 parts -- made of whole
    32-row tiles (a launch over rows that are not is served by the kernels that read the mask IMAGE)"""
    from . import ops
    return SIGN_BITS and ops.DBITS and rows > 0 and rows % 32 == 0


def generator_head(G, z, mode="bf16"):
    return _net(G, GEN, mode).run(dict(z=z.contiguous()), z.shape[0])


def critic2d(D, x, mode="bf16"):
    x = x.reshape(-1, 32).contiguous()
    return _net(D, D2, mode).run(dict(x=x), x.shape[0])


def critics(D3_mod, D2_mod, x3, kcs, x2, mode="bf16"):
    """both critics of one batch in ONE launch (the 3D critic's program, then the 2D critic's, per batch tile): one
    kernel start-up and one dispatch gap less than critic3d() + critic2d().  x3 (N,48) root-relative pose, kcs (N,32)
    bf16 operand (bf16 mode) or (N,30) fp32 features (f16x3), x2 (N,16,2) | (N,32) projection -> (logit3d (N,1), logit2d (N,1))"""
    x3 = x3.reshape(-1, 48).contiguous()                    # fp32, or bf16 as Fk_Generator.sample_for_critics can emit them
    x2 = x2.reshape(-1, 32).contiguous()
    M = x3.shape[0]
    assert x2.shape[0] == M and (kcs.shape[0] == M if kcs is not None else mode == "f16x3")
    u3, (o3,) = D3["program"](D3_mod, _net(D3_mod, D3, mode)._fresh(), dict(x=x3, kcs=kcs), M)
    u2, o2 = D2["program"](D2_mod, _net(D2_mod, D2, mode)._fresh(), dict(x=x2), M)
    launch(u3 + u2, M, mode)
    return o3, o2


def critic3d(D, x, center=False, kcs=None, mode="bf16"):
    """center=True: x is a world/camera-space pose; its root-relative copy and the KCS operand come from one pass.
    kcs: the operand if the caller already has it (Fk_Generator.sample_for_critics)"""
    x = x.reshape(-1, 48).contiguous()
    if mode == "f16x3":
        if center:
            x = ops.center_flip(x.reshape(-1, 16, 3), True, False).reshape(-1, 48)
        if kcs is not None and kcs.dtype != torch.float32:
            kcs = None                                        # (the features come from x inside the launch: LOAD_KCS)
        x = x.float()
    elif kcs is not None:
        pass
    elif center:
        x, kcs = ops.center_kcs_forward(x, 32, True)
    else:
        _, kcs = ops.kcs_forward(x, True, f32=False, bf16_ld=32)
    return _net(D, D3, mode).run(dict(x=x, kcs=kcs), x.shape[0])[0]
