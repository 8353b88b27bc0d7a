"""Development aid: what each grouped weight-gradient launch (ops.gemm_tn_group -> dhaug_gemm_tn_group_bf16_phase) of ONE eager video
iteration (B = 512 x R = 9, DenseDim 1000) is made of: per call of the C entry point the layers' (M, N1, N2), their 256 x 256 blocks, and the
stream; then each call timed alone on an idle card."""
import os, sys, argparse, time, ctypes
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
for i in range(3):
    V.video_gan_iteration(av, mv, v3, cpv, v2, ["S1"], sv, None, do_g_step=False, camera=(quat, trans, cam9))
torch.cuda.synchronize()
log = []
real = _lib.call
def spy(name, *a):
    if name == "dhaug_gemm_tn_group_bf16_phase":
        arr, n, ws, phase, stream = a
        layers = [(arr[i].M, arr[i].N1, arr[i].N2, int(arr[i].colsum_a is not None), arr[i].max_workgroups) for i in range(n)]
        keep = (_lib.TnLayer * n)()
        ctypes.memmove(keep, arr, ctypes.sizeof(_lib.TnLayer) * n)
        log.append((layers, phase, stream, keep, n, ws))
    return real(name, *a)
_lib.call = spy
V.video_gan_iteration(av, mv, v3, cpv, v2, ["S1"], sv, None, do_g_step=False, camera=(quat, trans, cam9))
torch.cuda.synchronize()
_lib.call = real
tot = 0.0
for layers, phase, stream, keep, n, ws in log:
    nb = sum(((l[1] + 255) // 256) * ((l[2] + 255) // 256) for l in layers)
    stages = sum(((l[1] + 255) // 256) * ((l[2] + 255) // 256) * (l[0] // 32) for l in layers)
    flops = sum(2.0 * l[0] * l[1] * l[2] for l in layers)
    # the call alone (same descriptors: the gradient slots take one more contribution, nothing reads them afterwards)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st = torch.cuda.current_stream().cuda_stream
    real("dhaug_gemm_tn_group_bf16_phase", keep, n, ws, phase, st)
    e0.record()
    for _ in range(5):
        real("dhaug_gemm_tn_group_bf16_phase", keep, n, ws, phase, st)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 200.0
    tot += us
    shapes = {}
    for l in layers:
        shapes[l[:3]] = shapes.get(l[:3], 0) + 1
    print("phase %d layers %2d blocks %4d stages %7d  %.1f GF  alone %7.1f us = %.2f of the MFMA peak  cap %d  %s" % (
        phase, len(layers), nb, stages, flops / 1e9, us, flops / (us * 1e-6) / 2.5e15, layers[0][4],
        " ".join("%dx(%d,%d,%d)" % (c, *k) for k, c in sorted(shapes.items()))))
print("calls %d, alone in total %.1f us" % (len(log), tot))
