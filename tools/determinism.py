"""Run the single-frame training iteration several times from identical state and report whether the critics' state
after N iterations is bit-identical between runs, for: eager (critics concurrent / serial) and hipGraph replay.
    python tools/determinism.py [B] [D] [N]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import argparse
import torch
import golden_util as GU
import test_gpu_graphs as T
import dhaug_amd
from dhaug_amd import graphs, critic_step as CS
from dhaug_amd.common.camera import camera_params9
from dhaug_amd.common.h36m_dataset import h36m_cameras_extrinsic_params, h36m_cameras_intrinsic_params
from dhaug_amd.models_Fk_GAN import forward_kinematics_DH_model as fkm, model_fk_gan_train as train
from test_gpu_models import make_args

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
D = int(sys.argv[2]) if len(sys.argv) > 2 else 256
N = int(sys.argv[3]) if len(sys.argv) > 3 else 4
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


def run(mode):
    d = T._build(M, args, D)
    dr = mk()
    train.CONCURRENT_CRITICS = mode != "serial"
    CS.TN_SPLIT = mode in ("split", "graph+split")
    out = []
    if mode.startswith("graph"):
        G = graphs.GraphedGanIteration(train.gan_iteration, args, d, ["S1"], None)
        for i in range(N):
            r = G(x3, cp, x2, False, cam, draws=dr)
            out.append([r[k].item() for k in ("D_cost_3D", "D_cost_2D")])
    else:
        for i in range(N):
            r = train.gan_iteration(args, d, x3, cp, x2, ["S1"], None, None, do_g_step=False, camera=cam, draws=dr)
            out.append([r[k].item() for k in ("D_cost_3D", "D_cost_2D")])
    torch.cuda.synchronize()
    return out, d["optimizer_d3d"].flat_param.clone(), d["optimizer_d2d"].flat_param.clone()


ref = None
REP = int(os.environ.get("REP", "2"))
for mode in ("serial",) + ("concurrent", "split", "graph", "graph+split") * REP:
    o, p3, p2 = run(mode)
    if ref is None:
        ref = (o, p3, p2)
    d3 = (p3 - ref[1]).abs().max().item(); d2 = (p2 - ref[2]).abs().max().item()
    first = next((i for i in range(N) if o[i] != ref[0][i]), None)
    print(f"{mode:11s} max|dp3| {d3:.3e} max|dp2| {d2:.3e} first differing iteration {first} D3 costs {[round(v[0], 6) for v in o]}", flush=True)

# ---- one critic step from identical state, twice: which parameter gradients differ between the two runs?
print("single critic step, gradient differences between two runs (max |dg| / max |g|):")
for key, okey, w in (("model_d3d", "optimizer_d3d", 48), ("model_d2d", "optimizer_d2d", 32)):
    grads = []
    for rep in range(3):
        d = T._build(M, args, D)
        real = (x3 - x3[:, :1]).reshape(B, 48) if w == 48 else x2.reshape(B, 32)
        fake = real.roll(1, 0) * 1.05 + 0.01
        al = torch.rand(B, 1, generator=torch.Generator().manual_seed(3)).cuda()
        sc = CS.critic_step(d[key], d[okey], real.contiguous(), fake.contiguous(), al, 10.0)
        torch.cuda.synchronize()
        g1 = {n: p.grad.clone() for n, p in d[key].named_parameters()}
        p1 = {"P1." + n: p.data.clone() for n, p in d[key].named_parameters()}
        sc2 = CS.critic_step(d[key], d[okey], real.contiguous(), fake.contiguous(), al, 10.0)
        torch.cuda.synchronize()
        g1.update(p1)
        g1.update({"G2." + n: p.grad.clone() for n, p in d[key].named_parameters()})
        grads.append((g1, [float(v) for v in sc[:5]] + [float(v) for v in sc2[:5]]))
    for n in grads[0][0]:
        a, b, c = grads[0][0][n], grads[1][0][n], grads[2][0][n]
        dd = max((a - b).abs().max().item(), (a - c).abs().max().item())
        if dd > 0:
            print(f"  {key} {n:34s} {tuple(a.shape)} max|dg| {dd:.3e} max|g| {a.abs().max().item():.3e} median|g| {a.abs().median().item():.3e}")
    print("  scalars", grads[0][1], grads[1][1])
