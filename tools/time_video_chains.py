"""Development aid: the video iteration's four critic chains (d3: 2 steps, d2: 2, m3: 4, m2: 4 at B = 512 x R = 9, DenseDim 1000), each
captured as a hipGraph of its own, timed ALONE, one after the other on one stream, and side by side on four streams -- how much of
the iteration's critic phase is chain latency, how much is contention between the chains.
    python tools/time_video_chains.py"""
import os, sys, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dhaug_amd
from dhaug_amd import ops
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

captured = {}


def fake_run(steps, optimizers, interleave, long_rows=False):
    """stands in for run_critic_steps: warm every chain up, capture it on a stream of its own, time the graphs"""
    keys = []
    for k, _ in steps:
        if k not in keys:
            keys.append(k)
    res = {}
    for _ in range(2):
        for i, (k, fn) in enumerate(steps):
            res[i] = fn()
    torch.cuda.synchronize()
    for k in keys:
        st = torch.cuda.Stream()
        g = torch.cuda.CUDAGraph()
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            g.capture_begin(pool=torch.cuda.graph_pool_handle())
            for i, (kk, fn) in enumerate(steps):
                if kk == k:
                    res[i] = fn()
            g.capture_end()
        captured[k] = (st, g, sum(1 for kk, _ in steps if kk == k))
    torch.cuda.synchronize()
    return res


T_run, V.run_critic_steps = V.run_critic_steps, fake_run
V.video_gan_iteration(av, mv, v3, cpv, v2, ["S1"], sv, None, do_g_step=False, camera=(quat, trans, cam9))
V.run_critic_steps = T_run


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


main = torch.cuda.current_stream()
if os.environ.get("CHAIN"):                      # profiling: one chain replayed alone, nothing else (rocprofv3 --kernel-trace --stats)
    st, g, n = captured[os.environ["CHAIN"]]
    torch.cuda.synchronize()
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    sys.exit(0)
alone = {}
for k, (st, g, n) in captured.items():
    alone[k] = timed(lambda: g.replay())
    print("chain %-3s (%d steps) alone: %.3f ms" % (k, n, alone[k]), flush=True)
print("sum of the chains alone: %.3f ms" % sum(alone.values()))


def side_by_side(keys):
    def run():
        for k in keys:
            st, g, _ = captured[k]
            st.wait_stream(main)
            with torch.cuda.stream(st):
                g.replay()
        for k in keys:
            main.wait_stream(captured[k][0])
    return run


ks = list(captured)
print("all four side by side: %.3f ms" % timed(side_by_side(ks)))
for pair in (("m3", "m2"), ("d3", "d2"), ("d3", "m3"), ("d2", "m2"), ("d3", "m2"), ("m3", "d2")):
    if all(k in captured for k in pair):
        print("  %s + %s side by side: %.3f ms (alone %.3f + %.3f)" % (pair[0], pair[1], timed(side_by_side(pair)), alone[pair[0]], alone[pair[1]]))

# host side: how long the CPU is inside hipGraphLaunch for every chain (the replay of a graph walks its nodes on the host)
import time
for k, (st, g, n) in captured.items():
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    g.replay()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("chain %-3s: host inside replay() %.3f ms, until the card is done %.3f ms" % (k, (t1 - t0) * 1e3, (t2 - t0) * 1e3))
