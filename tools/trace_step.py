"""Development aid: the kernels of ONE critic step, in order, with durations and overlap.
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ts -o r -- python tools/trace_step.py run
    python tools/trace_step.py show gpurun_out/ts        (-> gpurun_out/trace_step.txt)
`run` executes warm-up steps, then single steps separated by 30 ms of idle card (D3, D3, D2, D2 | video: M3, M2 with VIDEO=1); `show`
splits the trace at the idle gaps and prints the last segment of each kind."""
import os, sys, time, glob, csv, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run():
    import torch
    import dhaug_amd
    from dhaug_amd import ops
    from dhaug_amd.function_aug.config import synth_args
    from dhaug_amd.models_Fk_GAN import model_fk_gan_train as T
    from dhaug_amd.models_Fk_GAN.forward_kinematics_DH_model import Forward_Kinematics_DH_Model
    from dhaug_amd.common.camera import camera_params9
    from dhaug_amd.common.h36m_dataset import h36m_cameras_extrinsic_params, h36m_cameras_intrinsic_params
    B, D = int(os.environ.get("B", 65536)), int(os.environ.get("D", 256))
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
    real = ops.center_flip(rw, True, False); fake = real + 0.01
    S = argparse.Namespace(train_iter_num=0)
    d3 = lambda: T.train_Fk_discriminator(m["model_d3d"], real, fake, S, None, "a", m["optimizer_d3d"], args)
    d2 = lambda: T.train_Fk_discriminator(m["model_d2d"], r2, r2 + 0.01, S, None, "a", m["optimizer_d2d"], args)
    for _ in range(6):
        d3(); d2()
    torch.cuda.synchronize()
    if os.environ.get("GSTEP"):                  # whole iterations: without / with the generator step
        cp = torch.zeros(B, 16, device="cuda"); cp[:, 9:13] = torch.tensor(quat, device="cuda"); cp[:, 13:16] = torch.tensor(trans, device="cuda")
        it = lambda g: T.gan_iteration(args, m, rc, cp, r2, ["S1"], None, None, do_g_step=g, camera=(quat, trans, cam9))
        for _ in range(6):
            it(False)
        it(True); it(False)
        torch.cuda.synchronize()
        seq = (lambda: it(False), lambda: it(True), lambda: it(False), lambda: it(True))
    else:
        seq = (d3, d3, d2, d2)
    for fn in seq:
        time.sleep(0.03)
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        print("step %.3f ms" % ((time.perf_counter() - t0) * 1e3))
    time.sleep(0.03)


def show(d):
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    segs, cur, last_end = [], [], None
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if last_end is not None and s - last_end > 20e6:
            segs.append(cur); cur = []
        cur.append(r)
        last_end = e if last_end is None else max(last_end, e)
    segs.append(cur)
    out = open("gpurun_out/trace_step.txt", "w")
    for seg in segs[-4:]:
        t0 = int(seg[0]["Start_Timestamp"])
        end = max(int(r["End_Timestamp"]) for r in seg)
        out.write("=== segment: %d dispatches, span %.1f us, sum %.1f us\n" % (len(seg), (end - t0) / 1e3, sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg) / 1e3))
        prev_end = t0
        for r in seg:
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            out.write("%9.1f %8.1f us gap %7.1f q%s grid %-8s lds %-6s %s\n" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r.get("Queue_Id", ""), r.get("Grid_Size", r.get("Grid_Size_X", "")),
                      r.get("LDS_Block_Size", ""), r["Kernel_Name"].replace("(anonymous namespace)::", "")[:70]))
            prev_end = max(prev_end, e)
    out.close()
    print(open("gpurun_out/trace_step.txt").read()[-6000:])


if __name__ == "__main__":
    run() if sys.argv[1] == "run" else show(sys.argv[2])
