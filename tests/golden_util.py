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
