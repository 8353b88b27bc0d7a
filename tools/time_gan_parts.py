"""Development aid: GAN iteration time without / with the generator step, and one critic step of each critic."""
import os, sys, time, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dhaug_amd
from dhaug_amd import ops
from dhaug_amd.function_aug.config import synth_args
from dhaug_amd.models_Fk_GAN import model_fk_gan_train as T
from dhaug_amd.models_Fk_GAN.forward_kinematics_DH_model import Forward_Kinematics_DH_Model
from dhaug_amd.common.camera import camera_params9
from dhaug_amd.common.h36m_dataset import h36m_cameras_extrinsic_params, h36m_cameras_intrinsic_params

B, D = 65536, 256
if os.environ.get("CAP"):
    ops._lib.lib().dhaug_set_workgroup_cap(int(os.environ["CAP"]))       # persistent launches on a part of the card
args = synth_args(B, D)
fk = Forward_Kinematics_DH_Model(args, ["S1"], None)
m = T.my_get_poseFk_model(args, None, fk)
ext = h36m_cameras_extrinsic_params["S1"][0]
quat, trans = [float(v) for v in ext["orientation"]], [float(v) / 1000.0 for v in ext["translation"]]
cam9 = camera_params9(h36m_cameras_intrinsic_params[0])
ang = (torch.randn(B, 37, device="cuda") * 40).clamp(-180, 180)
bl = torch.rand(B, 15, device="cuda") * 0.4 + 0.1
rw = ops.fk_forward(ang, bl, torch.randn(B, 3, device="cuda") * 0.3)
rc, r2 = ops.world_to_camera_project(rw, quat, trans, cam9)
cp = torch.zeros(B, 16, device="cuda"); cp[:, 9:13] = torch.tensor(quat, device="cuda"); cp[:, 13:16] = torch.tensor(trans, device="cuda")


def t(fn, n=10, w=3):
    for _ in range(w):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


it = lambda g: T.gan_iteration(args, m, rc, cp, r2, ["S1"], None, None, do_g_step=g, camera=(quat, trans, cam9))
for _ in range(20):
    it(False)
print("iteration without G step %.2f ms, with G step %.2f ms" % (t(lambda: it(False)), t(lambda: it(True))))
real = ops.center_flip(rw, True, False); fake = real + 0.01
S = argparse.Namespace(train_iter_num=0)
print("D3 critic step %.2f ms" % t(lambda: T.train_Fk_discriminator(m["model_d3d"], real, fake, S, None, "a", m["optimizer_d3d"], args)))
print("D2 critic step %.2f ms" % t(lambda: T.train_Fk_discriminator(m["model_d2d"], r2, r2 + 0.01, S, None, "a", m["optimizer_d2d"], args)))
with torch.no_grad():
    print("G sample %.3f ms" % t(lambda: m["model_G"](torch.randn(B, 128, device="cuda"))))
