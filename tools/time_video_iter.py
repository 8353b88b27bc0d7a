"""Development aid: the video iteration (B = 512 x R = 9, DenseDim 1000) as bench.py runs it (hipGraphs, forked critic chains), timed per
iteration with a synchronisation behind each: iterations without / with the generator step, and the replay's items one by one."""
import os, sys, argparse, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dhaug_amd
from dhaug_amd import ops
from dhaug_amd.graphs import GraphedGanIteration
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
gv = GraphedGanIteration(V.video_gan_iteration, av, mv, ["S1"], sv)
for i in range(10):
    gv(v3, cpv, v2, i % 5 == 4, (quat, trans, cam9))
torch.cuda.synchronize()
for g in (False, True):
    ts = []
    for _ in range(6):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        gv(v3, cpv, v2, g, (quat, trans, cam9))
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    print("iteration %s the G step, synchronised each: %s ms" % ("WITH" if g else "without", " ".join("%.2f" % t for t in ts)))
t0 = time.perf_counter()
for i in range(20):
    gv(v3, cpv, v2, i % 5 == 4, (quat, trans, cam9))
torch.cuda.synchronize()
print("20 iterations back to back: %.2f ms each" % ((time.perf_counter() - t0) * 1e3 / 20))
# the items of the no-G-step replay, one by one
key = [k for k in gv.graphs if not k[0]] if all(isinstance(k, tuple) for k in gv.graphs) else list(gv.graphs)   # (key[0]: with the G step)
fc = gv.graphs[key[0]]
cur = torch.cuda.current_stream()
for kind, obj in getattr(fc, "items", []):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if kind == "graph":
        obj.replay()
    else:
        for st, g in obj:
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                g.replay()
        for st, _ in obj:
            cur.wait_stream(st)
    torch.cuda.synchronize()
    print("  item %-5s %s: %.3f ms" % (kind, "" if kind == "graph" else "(%d chains)" % len(obj), (time.perf_counter() - t0) * 1e3))

# the forked chains of that replay, each alone on an idle card (what the fork overlaps)
for kind, obj in getattr(fc, "items", []):
    if kind == "graph":
        continue
    for i, (st, g) in enumerate(obj):
        ts = []
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            with torch.cuda.stream(st):
                g.replay()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        print("  chain %d alone: %.3f ms" % (i, min(ts)))
# the same chains dealt to fewer streams (the chains of one stream one after the other): what the overlap of four is worth
for kind, obj in getattr(fc, "items", []):
    if kind == "graph":
        continue
    for part in ([[0], [1], [2], [3]], [[0, 2], [1, 3]], [[1, 2], [0, 3]], [[1], [0, 2, 3]], [[0, 1], [2, 3]], [[0, 1, 2, 3]]):
        ts = []
        for _ in range(4):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for grp in part:
                st = obj[grp[0]][0]
                st.wait_stream(cur)
                with torch.cuda.stream(st):
                    for i in grp:
                        obj[i][1].replay()
            for grp in part:
                cur.wait_stream(obj[grp[0]][0])
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        print("  chains as %s: %.3f ms" % (part, min(ts)))
