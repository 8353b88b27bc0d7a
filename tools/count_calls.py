import os, sys, collections, traceback
sys.path.insert(0, "/root/repo")
import torch
import dhaug_amd
from dhaug_amd import ops, _lib
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
for _ in range(3): it(False)
cnt = collections.Counter(); sites = collections.Counter()
orig = _lib.call
def call(name, *a):
    cnt[name] += 1
    if name in ("dhaug_cast_pad_bf16",):
        st = traceback.extract_stack(limit=7)
        sites[" < ".join("%s:%d" % (os.path.basename(f.filename), f.lineno) for f in reversed(st[:-1]))[:150]] += 1
    return orig(name, *a)
_lib.call = call; ops._lib.call = call
it(False)
torch.cuda.synchronize()
print(sum(cnt.values()), "C-ABI calls in an iteration without G step")
for k, v in cnt.most_common(40): print("%4d %s" % (v, k))
print("cast_pad call sites:")
for k, v in sites.most_common(): print("%3d %s" % (v, k))
