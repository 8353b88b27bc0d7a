"""development: host-side issue time of the eager single-frame iteration against its GPU time, and any implicit host
synchronisation inside it (torch's sync debug mode warns where one happens).  cProfile of the issue path with --profile."""
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
it = lambda g: T.gan_iteration(args, m, rc, cp, r2, ["S1"], None, None, do_g_step=g, camera=(quat, trans, cam9))
for i in range(20):
    it(i % 5 == 4)
torch.cuda.synchronize()
torch.cuda.set_sync_debug_mode("warn")
it(False); it(True)
torch.cuda.set_sync_debug_mode("default")
torch.cuda.synchronize()
N = 20
t0 = time.perf_counter()
for i in range(N):
    it(i % 5 == 4)
th = time.perf_counter() - t0
torch.cuda.synchronize()
tg = time.perf_counter() - t0
print("host issue %.2f ms / iteration, until the GPU is done %.2f ms / iteration" % (th / N * 1e3, tg / N * 1e3))
if "--profile" in sys.argv:
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
    for i in range(10):
        it(i % 5 == 4)
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(28)
