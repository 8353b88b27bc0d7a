"""Development aid: every NT layer-product call (dhaug_gemm_bf16*, single and grouped) of ONE eager video iteration (B = 512 x R = 9, DenseDim 1000)
by entry point and shape -- which layers still travel as single short launches."""
import os, sys, argparse, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dhaug_amd
from dhaug_amd import ops, _lib
from dhaug_amd.function_aug.config import synth_args
from dhaug_amd.models_Fk_GAN import model_fk_gan_train as T, video_GAN_fun as V
from dhaug_amd.models_Fk_GAN.forward_kinematics_DH_model import Forward_Kinematics_DH_Model
from dhaug_amd.common.camera import camera_params9
from dhaug_amd.common.h36m_dataset import h36m_cameras_extrinsic_params, h36m_cameras_intrinsic_params

dev = "cuda"
Bv, Dv, Rv = 512, 1000, 9
Nv = Bv * Rv
ext = h36m_cameras_extrinsic_params["S1"][0]
quat, trans = [float(v) for v in ext["orientation"]], [float(v) / 1000.0 for v in ext["translation"]]
cam9 = camera_params9(h36m_cameras_intrinsic_params[0])
av = synth_args(Bv, Dv, single_or_multi_train_mode="multi", architecture="3,3", video_Dis_DenseDim_3D=Dv, video_Dis_DenseDim_2D=Dv,
                single_dis_warmup_epoch=0)
mv = T.video_mode_my_get_poseFk_model(av, None, Forward_Kinematics_DH_Model(av, ["S1"], None), Rv)
angv = (torch.randn(Nv, 37, device=dev) * 40).clamp(-180, 180)
rwv = ops.fk_forward(angv, torch.rand(Nv, 15, device=dev) * 0.4 + 0.1, torch.randn(Nv, 3, device=dev).clamp(-10, 10) * 0.3)
rcv, r2v = ops.world_to_camera_project(rwv, quat, trans, cam9)
cpv = torch.zeros(Bv, 16, device=dev)
cpv[:, 9:13] = torch.tensor(quat, device=dev)
cpv[:, 13:16] = torch.tensor(trans, device=dev)
mv["model_G"].GAN_generator_get_bone_length(rcv)
v3, v2 = rcv.reshape(Bv, Rv, 16, 3), r2v.reshape(Bv, Rv, 16, 2)
sv = argparse.Namespace(epoch=10, train_iter_num=0)
for i in range(2):
    V.video_gan_iteration(av, mv, v3, cpv, v2, ["S1"], sv, None, do_g_step=False, camera=(quat, trans, cam9))
torch.cuda.synchronize()
log = collections.Counter()
real = _lib.call
def spy(name, *a):
    if name == "dhaug_gemm_bf16":
        log["gemm_bf16            M %6d N %4d K %4d%s%s%s act %d" % (a[14], a[15], a[16], " bias" if a[4] else "", " res" if a[5] else "", " f32out" if a[12] else "", a[17])] += 1
    elif name == "dhaug_gemm_bf16_dmask":
        log["gemm_bf16_dmask      M %6d N %4d K %4d%s" % (a[12], a[13], a[14], " res" if a[4] else "")] += 1
    elif name == "dhaug_gemm_bf16_dmask_pad":
        log["gemm_bf16_dmask_pad  M %6d N %4d K %4d%s" % (a[13], a[14], a[15], " res" if a[4] else "")] += 1
    elif name == "dhaug_gemm_bf16_group":
        d = a[0]
        log["gemm_bf16_group x%d   M %6d N %4d K %4d%s%s" % (a[1], d[0].M, d[0].N, d[0].K, " res" if d[0].residual else "", " mask" if d[0].dmask else "")] += 1
    elif name.startswith("dhaug_gemm") and "tn" not in name:
        log[name] += 1
    return real(name, *a)
_lib.call = spy
V.video_gan_iteration(av, mv, v3, cpv, v2, ["S1"], sv, None, do_g_step=False, camera=(quat, trans, cam9))
torch.cuda.synchronize()
_lib.call = real
for k, v in sorted(log.items()):
    print("%4d x %s" % (v, k))
