"""Uninitialised-memory probe.  torch.empty / empty_like / new_empty are patched to return buffers filled with a poison
value (a huge finite number, then NaN; 0xFF bytes for integer types), and N training iterations (eager, then hipGraph
replays) are compared bit for bit with an unpoisoned run: any result that depends on memory nobody wrote shows up.
    python tools/poison.py [B] [D] [N]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import argparse
import torch
import golden_util as GU
import test_gpu_graphs as T
import dhaug_amd
from dhaug_amd import graphs
from dhaug_amd.common.camera import camera_params9
from dhaug_amd.common.h36m_dataset import h36m_cameras_extrinsic_params, h36m_cameras_intrinsic_params
from dhaug_amd.models_Fk_GAN import forward_kinematics_DH_model as fkm, model_fk_gan_train as train
from test_gpu_models import make_args

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
D = int(sys.argv[2]) if len(sys.argv) > 2 else 256
N = int(sys.argv[3]) if len(sys.argv) > 3 else 5
ext = h36m_cameras_extrinsic_params["S1"][0]
cam = ([float(v) for v in ext["orientation"]], [float(v) / 1000.0 for v in ext["translation"]], camera_params9(h36m_cameras_intrinsic_params[0]))
M = argparse.Namespace(graphs=graphs, fkm=fkm, train=train, cam=cam)
args = make_args(batch_size=B, Gen_DenseDim=D, Dis_DenseDim_3D=D, Dis_DenseDim_2D=D)
gen = torch.Generator().manual_seed(5)
x3 = GU.synth_pose16(B, seed=3).cuda() + torch.tensor([0.0, 0.0, 4.5], device="cuda")
x2 = ((torch.rand(B, 16, 2, generator=gen) - 0.5) * 1.2).cuda()
cp = torch.zeros(B, 16, device="cuda"); cp[:, 9:13] = torch.tensor(cam[0], device="cuda"); cp[:, 13:16] = torch.tensor(cam[1], device="cuda")
mk = lambda: train.ConstDraws(noise=[torch.randn(B, 128, generator=torch.Generator().manual_seed(1)).cuda()],
                              scaler=[(torch.randint(-200, 200, (B, 8), generator=torch.Generator().manual_seed(2)) / 1000.0).cuda()],
                              alpha=[torch.rand(B, 1, generator=torch.Generator().manual_seed(3)).cuda()])

POISON = [None]
_empty, _empty_like, _new_empty = torch.empty, torch.empty_like, torch.Tensor.new_empty


def _fill(t):
    p = POISON[0]
    if p is not None and t.is_cuda and t.numel():
        if t.dtype.is_floating_point:
            t.fill_(p)
        else:
            t.view(torch.uint8).fill_(255)
    return t


torch.empty = lambda *a, **k: _fill(_empty(*a, **k))
torch.empty_like = lambda *a, **k: _fill(_empty_like(*a, **k))
torch.Tensor.new_empty = lambda self, *a, **k: _fill(_new_empty(self, *a, **k))


def run(graph):
    d = T._build(M, args, D)
    dr = mk()
    out = []
    G = graphs.GraphedGanIteration(train.gan_iteration, args, d, ["S1"], None) if graph else None
    for i in range(N):
        g = i % 5 == 4
        r = G(x3, cp, x2, g, cam, draws=dr) if graph else train.gan_iteration(args, d, x3, cp, x2, ["S1"], None, None, do_g_step=g, camera=cam, draws=dr)
        out.append([r[k].item() if r[k] is not None else None for k in ("D_cost_3D", "D_cost_2D", "G_cost")])
    torch.cuda.synchronize()
    return out, [d[k].flat_param.clone() for k in ("optimizer_d3d", "optimizer_d2d", "optimizer_G")]


for graph in (False, True):
    ref = None
    for p in (None, 3.0e38, -7.0e4, float("nan")):
        POISON[0] = p
        o, ps = run(graph)
        if ref is None:
            ref = (o, ps)
        first = next((i for i in range(N) if o[i] != ref[0][i]), None)
        dp = [float("nan") if torch.isnan(a).any() else (a - b).abs().max().item() for a, b in zip(ps, ref[1])]
        print(f"graph={graph} poison={p}: first differing iteration {first}; max|dp| d3 {dp[0]:.2e} d2 {dp[1]:.2e} G {dp[2]:.2e}; last {o[-1]}", flush=True)
