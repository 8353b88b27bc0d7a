"""Profiling target: single-frame GAN iterations WITH (GS=1) or without (GS=0) the generator step, B = 65 536, D = 256, bf16, eager.
    rocprofv3 --kernel-trace --stats -- python tools/prof_gstep.py     (the difference of the two runs is the G step)"""
import os, sys, time
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
gs = os.environ.get("GS", "1") == "1"
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
it = lambda: T.gan_iteration(args, m, rc, cp, r2, ["S1"], None, None, do_g_step=gs, camera=(quat, trans, cam9))
for _ in range(2): it()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(8): it()
torch.cuda.synchronize()
print("G step %s: %.2f ms per iteration" % ("on" if gs else "off", (time.perf_counter() - t0) / 8 * 1e3))
