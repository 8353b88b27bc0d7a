"""Deterministic synthetic inputs shared by the golden generator and the tests (no reference code)."""
import numpy as np
import torch


def seeded_state_dict(shapes, seed):
    """name -> tensor, U(-1/sqrt(fan_in), 1/sqrt(fan_in)) from numpy's stream-stable MT19937.
    `shapes`: dict name -> shape tuple in state_dict order ('x.weight' (out,in) followed by 'x.bias')."""
    rs = np.random.RandomState(seed)
    out, fan_in = {}, 1
    for k, shp in shapes.items():
        if k.endswith(".weight"):
            fan_in = shp[1]
        bound = 1.0 / np.sqrt(fan_in)
        out[k] = torch.tensor(((rs.random_sample(shp) * 2 - 1) * bound).astype(np.float32))
    return out


def synth_fk_inputs(N, seed):
    """angles U(-180,180) deg, bone_len U(0.1,0.5) m, root N(0,1) clipped +-10."""
    g = torch.Generator().manual_seed(seed)
    angles = (torch.rand(N, 37, generator=g) * 2 - 1) * 180.0
    bone_len = torch.rand(N, 15, generator=g) * 0.4 + 0.1
    root = torch.randn(N, 3, generator=g).clamp(-10, 10)
    return angles, bone_len, root


def synth_pose16(N, seed):
    """Random non-degenerate 16-joint poses (metres)."""
    g = torch.Generator().manual_seed(seed)
    return torch.randn(N, 16, 3, generator=g) * 0.3


def shapes_generator(D, frames=1):
    s = {"preprocess.0.weight": (D, 128), "preprocess.0.bias": (D,)}
    for b in ("block1", "block2", "block3"):
        for f in ("fc1", "fc2"):
            s["%s.%s.weight" % (b, f)] = (D, D); s["%s.%s.bias" % (b, f)] = (D,)
    s["deconv_out.weight"] = (35 * frames, D); s["deconv_out.bias"] = (35 * frames,)
    return s


def _res(s, name, D):
    for f in ("fc1", "fc2"):
        s["%s.%s.weight" % (name, f)] = (D, D); s["%s.%s.bias" % (name, f)] = (D,)


def shapes_d3(D):
    s = {"previous.0.weight": (D, 48), "previous.0.bias": (D,)}
    for b in ("block1", "block2", "block3"):
        _res(s, b, D)
    s["special_KCS_previous.0.weight"] = (D, 30); s["special_KCS_previous.0.bias"] = (D,)
    for b in ("special_KCS_block1", "special_KCS_block2", "special_KCS_block3"):
        _res(s, b, D)
    s["merge_previous.0.weight"] = (100, 2 * D); s["merge_previous.0.bias"] = (100,)
    _res(s, "merge_block1", 100)
    s["output.weight"] = (1, 100); s["output.bias"] = (1,)
    return s


def shapes_d2(D):
    s = {"pose_layer_1.weight": (D, 32), "pose_layer_1.bias": (D,)}
    for n in ("pose_layer_2", "pose_layer_3", "pose_layer_4", "layer_last"):
        s[n + ".weight"] = (D, D); s[n + ".bias"] = (D,)
    s["layer_pred.weight"] = (1, D); s["layer_pred.bias"] = (1,)
    return s


# ---- compact fixtures of large tensors (critic_step_*_D256: 0.9 M parameters per critic) -----------------------------------
def compact(t, seed, max_full=8192, samples=4096, nproj=32):
    """what a fixture keeps of a large tensor: every (numel // samples)-th element and nproj seeded +-1 projections (a wrong
    gradient moves them with overwhelming probability); small tensors are kept whole.  Deterministic in (shape, seed)."""
    import torch
    f = t.detach().double().reshape(-1).cpu()
    if f.numel() <= max_full:
        return dict(full=f.float())
    stride = f.numel() // samples
    g = torch.Generator().manual_seed(1000003 * seed + f.numel())
    signs = torch.randint(0, 2, (nproj, f.numel()), generator=g, dtype=torch.int8).double() * 2 - 1
    return dict(sample=f[::stride].float(), proj=(signs @ f))


def compact_close(t, ref, seed, atol, rtol, name=""):
    """t against the compact record `ref` of the reference's tensor: sampled elements within atol + rtol * max|ref|, projections
    within the same per-element bound times sqrt(numel) (independent roundings add in quadrature; x4 margin)"""
    got = compact(t, seed)
    if "full" in ref:
        scale = ref["full"].abs().max().item()
        assert (got["full"].double() - ref["full"].double()).abs().max().item() <= atol + rtol * scale, name
        return
    scale = ref["sample"].abs().max().item()
    assert (got["sample"].double() - ref["sample"].double()).abs().max().item() <= atol + rtol * scale, name
    n = t.numel()
    assert (got["proj"] - ref["proj"].double()).abs().max().item() <= 4.0 * (atol + rtol * scale) * n ** 0.5, name


def compact_close_but_flips(t, ref, seed, atol, rtol, name="", max_flips=4, flip_rtol=0.15):
    """compact_close for gradients of LONG batches at fp32-grade arithmetic: every recorded element within atol + rtol * max|ref|, EXCEPT
    at most max_flips elements, which may be off by up to flip_rtol * max|ref| -- the units whose pre-activation lies within fp32
    rounding of zero somewhere in the batch: their ReLU mask falls on the other side than in the reference's run, and ONE flipped
    (row, unit) moves that unit's bias-gradient element by the row's whole cotangent (measured at B = 128 clips, DenseDim 1000: three
    such elements in the 3D motion critic, 6e-5 on gradients of scale 1e-3; with ~1e7 pre-activations per step a handful of them within
    1e-7 of zero is what a normal distribution gives).  The projections keep compact_close's bound (a few flipped elements vanish in a
    sum over the whole tensor)."""
    got = compact(t, seed)
    key = "full" if "full" in ref else "sample"
    scale = ref[key].abs().max().item()
    err = (got[key].double() - ref[key].double()).abs()
    over = err > atol + rtol * scale
    assert int(over.sum()) <= max_flips, (name, int(over.sum()), err.max().item(), scale)
    assert err.max().item() <= atol + flip_rtol * scale, (name, err.max().item(), scale)
    if "proj" in ref:
        n = t.numel()
        assert (got["proj"] - ref["proj"].double()).abs().max().item() <= 4.0 * (atol + rtol * scale) * n ** 0.5 + max_flips * flip_rtol * scale, name
    return int(over.sum())
