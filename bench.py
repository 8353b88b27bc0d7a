#!/usr/bin/env python3
"""bench.py -- throughput of the DH-AUG hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload fwd|fk_gen_fwd|gan_step|fk] [--batch 65536]

One "step" = one pass of the hot path over one batch of synthetic input that is already resident in HBM:
    fwd        (default) FK + Gen + D3 + D2 forward, B = 65 536 poses per GPU, D = 256, bf16 dense layers, fp32 FK
               (north_star's target workload; superset of BASELINE.json configs[1])
    fk_gen_fwd FK + Gen forward only (configs[1] exactly)
    gan_step   full single-frame GAN iteration (configs[2]/[3]): 2+2 WGAN-GP critic steps (flip copies), G step every
               5th iteration, fused Adam; with N > 1 one RCCL all-reduce per optimizer step
    fk         the FK kernel alone on B poses
N > 1: one process per GPU (torch.distributed.run), the batch shards across ranks (B per rank, weak scaling); the
forward workloads have no exchange step, gan_step all-reduces the flat gradient bucket of the network being stepped.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_BF16_PEAK_TFLOPS = 2500.0     # dense bf16, MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0              # HBM3E spec, MI355X_MICROARCH.md (6.29 TB/s measured copy ceiling)
FK_BYTES_PER_POSE = 412            # 37+15+3 fp32 in, 48 fp32 out


def mac_per_pose(D):
    gen = 128 * D + 6 * D * D + 35 * D
    d3 = 78 * D + 12 * D * D + 200 * D + 2 * 100 * 100 + 100
    d2 = 32 * D + 4 * D * D + D
    return gen, d3, d2


def event_time(fn, iters, warm, rewarm_s=0.3):
    """average duration of fn (seconds), HIP events on the stream the kernels are launched on; rewarm_s of fn first so
    that the kernel is timed at the clocks it runs at inside the loaded step, not at those left by the previous phase"""
    t_end = time.perf_counter() + rewarm_s
    while time.perf_counter() < t_end:
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
    for _ in range(warm):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e-3 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--prewarm", type=float, default=1.0,
                    help="seconds of the workload run before the W warmup steps, untimed: MI355X clocks take a few hundred ms "
                         "of load to leave their idle state (the same step measures 0.28 ms right after start-up, 0.25 ms warm)")
    ap.add_argument("--workload", default="fwd", choices=["fwd", "fk_gen_fwd", "gan_step", "fk"])
    ap.add_argument("--batch", type=int, default=65536, help="poses per GPU per step")
    ap.add_argument("--dense", type=int, default=256)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true")
    ap.add_argument("--no-roofline", action="store_true", help="skip the per-kernel roofline timing loops (profiling runs)")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: this process becomes the launcher.  It has not touched the GPU (importing
        # torch does not initialise HIP) and never will: it starts N rank processes and relays rank 0's JSON line.
        sys.exit(launch_ranks(a.gpus))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and world != a.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with --nproc-per-node %d)" % (a.gpus, world, a.gpus))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # "nccl" = RCCL over xGMI on ROCm.  DHAUG_DIST_BACKEND=gloo lets the multi-rank path be rehearsed with all
        # ranks on one GPU (RCCL refuses duplicate devices).
        dist.init_process_group(os.environ.get("DHAUG_DIST_BACKEND", "nccl"))
    local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    import dhaug_amd
    from dhaug_amd import ops
    from dhaug_amd.function_aug.config import synth_args
    from dhaug_amd.models_Fk_GAN import model_fk_gan_train as T
    from dhaug_amd.models_Fk_GAN.forward_kinematics_DH_model import Forward_Kinematics_DH_Model
    from dhaug_amd.models_Fk_GAN.Fk_discriminator import score_fake_pair
    from dhaug_amd.common.camera import camera_params9
    from dhaug_amd.common.h36m_dataset import h36m_cameras_extrinsic_params, h36m_cameras_intrinsic_params
    dhaug_amd._lib.lib()

    B, D = a.batch, a.dense
    args = synth_args(B, D)
    torch.manual_seed(1234 + rank)
    fk = Forward_Kinematics_DH_Model(args, ["S1"], None)
    models = T.my_get_poseFk_model(args, None, fk)
    G, D3, D2 = models["model_G"], models["model_d3d"], models["model_d2d"]
    if world > 1:                                # data-parallel replicas start from rank 0's weights
        for k in ("optimizer_G", "optimizer_d3d", "optimizer_d2d"):
            dist.broadcast(models[k].flat_param, 0)

    # synthetic inputs, resident in HBM
    ext = h36m_cameras_extrinsic_params["S1"][0]
    quat, trans = [float(v) for v in ext["orientation"]], [float(v) / 1000.0 for v in ext["translation"]]
    cam9 = camera_params9(h36m_cameras_intrinsic_params[0])
    ang = (torch.randn(B, 37, device=dev) * 40).clamp(-180, 180)
    bl = torch.rand(B, 15, device=dev) * 0.4 + 0.1
    real_world = ops.fk_forward(ang, bl, torch.randn(B, 3, device=dev).clamp(-10, 10) * 0.3)
    real_cam, real_2d = ops.world_to_camera_project(real_world, quat, trans, cam9)
    cam_param = torch.zeros(B, 16, device=dev)
    cam_param[:, 9:13] = torch.tensor(quat, device=dev)
    cam_param[:, 13:16] = torch.tensor(trans, device=dev)
    G.GAN_generator_get_bone_length(real_cam)
    z = torch.randn(B, 128, device=dev)
    root = torch.randn(B, 3, device=dev)
    it = [0]

    def step_fk():
        ops.fk_forward(ang, bl, root)

    def step_fk_gen():
        with torch.no_grad():
            return G(z)

    def step_fwd():
        with torch.no_grad():
            fw, xc, kcs, p2 = G.sample_for_critics(z, (quat, trans, cam9), inputs_bf16=True)   # FK tail + critic inputs, one launch
            l3, l2 = score_fake_pair(D3, D2, xc, kcs, p2)                      # both critics, one launch
        return l3, l2

    def step_gan():
        T.gan_iteration(args, models, real_cam, cam_param, real_2d, ["S1"], summary=None, writer=None,
                        do_g_step=(it[0] % 5 == 4), camera=(quat, trans, cam9))
        it[0] += 1

    steps = {"fk": step_fk, "fk_gen_fwd": step_fk_gen, "fwd": step_fwd, "gan_step": step_gan}

    def timed(fn, k, w):
        for _ in range(w):
            fn()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(k):
            fn()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([t], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            t = tt.item()
        return t

    if a.prewarm > 0:                                   # bring the clocks up (untimed; W warmup steps still follow)
        if world > 1:
            # a step of the training workload holds a collective: every rank must run the same number of them, so the
            # pre-warm is a step count here (~prewarm seconds at the single-GPU rates), not a wall-clock loop
            for _ in range(max(1, int(a.prewarm * (45 if a.workload == "gan_step" else 3500)))):
                steps[a.workload]()
            torch.cuda.synchronize()
        else:
            t_end = time.perf_counter() + a.prewarm
            while time.perf_counter() < t_end:
                for _ in range(20):
                    steps[a.workload]()
                torch.cuda.synchronize()
    t = timed(steps[a.workload], a.steps, a.warmup)
    value = B * world * a.steps / t

    extra = {}
    if not a.no_extra:
        for name in ("fk_gen_fwd", "fwd", "gan_step"):
            if name != a.workload:
                k = max(5, min(a.steps, 50 if name != "gan_step" else 10))
                try:
                    te = timed(steps[name], k, 5 if name == "gan_step" else 3)
                    extra[name + "_poses_per_s"] = B * world * k / te
                    extra[name + "_ms_per_step"] = te / k * 1e3
                except Exception as ex:              # never lose the headline line to an optional measurement
                    extra[name + "_error"] = repr(ex)[:200]

    if rank == 0:
        gen_mac, d3_mac, d2_mac = mac_per_pose(D)
        # dominant kernel of the default workload: the fused D3 forward (one launch, 1.755 MFLOP/pose at D=256)
        from dhaug_amd import fused
        x3 = torch.randn(B, 48, device=dev) * 0.3
        with torch.no_grad():
            tg = event_time(lambda: fused.critic3d(D3, x3), 100, 20)
        kcs_t = event_time(lambda: ops.kcs_forward(x3, True, f32=False, bf16_ld=32), 100, 20)
        tg = max(tg - kcs_t, 1e-9)                       # critic3d() = KCS kernel + fused kernel
        roofline = {"kernel": "fused_mlp_kernel (Fk_3D_Discriminator forward, M=%d, D=%d, 17 layers in one launch)" % (B, D),
                    "bound": "mfma", "achieved": 2.0 * d3_mac * B / tg / 1e12, "peak": MFMA_BF16_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": 2.0 * d3_mac * B / tg / 1e12 / MFMA_BF16_PEAK_TFLOPS,
                    "traffic": pmc_traffic("fused_mlp_kernel", "d3_") if (B, D) == (65536, 256) else None,
                    "avg_us": tg * 1e6, "algorithmic_flop_per_pose": 2 * d3_mac}
        xb = torch.randn(B, D, device=dev).to(torch.bfloat16)
        wb = (torch.randn(D, D, device=dev) / D ** 0.5).to(torch.bfloat16)
        bias = torch.zeros(D, device=dev)
        tl = event_time(lambda: ops.gemm_nt(xb, wb, D, D, bias=bias, res_bf16=xb, act=1, out_bf16=True), 50, 10)
        roofline_layer = {"kernel": "gemm_nt256s_kernel<16,1> (training path: one M=%d, N=K=%d layer, bias+residual+ReLU)" % (B, D),
                          "bound": "hbm", "achieved": (3 * B * D * 2 + D * D * 2) / tl / 1e9, "peak": HBM_PEAK_GBS,
                          "unit": "GB/s", "frac": (3 * B * D * 2 + D * D * 2) / tl / 1e9 / HBM_PEAK_GBS, "traffic": None,
                          "avg_us": tl * 1e6, "tflops": 2.0 * B * D * D / tl / 1e12}
        nfk = 1 << 22
        a4 = (torch.rand(nfk, 37, device=dev) * 2 - 1) * 180
        b4 = torch.rand(nfk, 15, device=dev) * 0.4 + 0.1
        r4 = torch.randn(nfk, 3, device=dev)
        tf = event_time(lambda: ops.fk_forward(a4, b4, r4), 20, 5)
        tf_b = event_time(step_fk, 50, 10)
        roofline_fk = {"kernel": "fk_forward_kernel<0,16,true>", "bound": "hbm", "achieved": FK_BYTES_PER_POSE * nfk / tf / 1e9,
                       "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": FK_BYTES_PER_POSE * nfk / tf / 1e9 / HBM_PEAK_GBS,
                       "traffic": pmc_traffic("fk_forward_kernel<0; 16; true; false>"), "algorithmic_bytes": FK_BYTES_PER_POSE * nfk, "poses_per_launch": nfk, "avg_us": tf * 1e6,
                       "at_batch": {"poses": B, "avg_us": tf_b * 1e6, "achieved": FK_BYTES_PER_POSE * B / tf_b / 1e9}}
        del a4, b4, r4

        cpu = None
        if world == 1 and not a.no_cpu_baseline:
            cpu = cpu_baseline(a.workload, D, {k: v.detach().cpu() for k, v in G.state_dict().items()},
                               {k: v.detach().cpu() for k, v in D3.state_dict().items()},
                               {k: v.detach().cpu() for k, v in D2.state_dict().items()}, quat, trans, cam9)
        flops = {"fk": 2.5e3, "fk_gen_fwd": 2.0 * gen_mac, "fwd": 2.0 * (gen_mac + d3_mac + d2_mac)}.get(a.workload)
        out = {"metric": "augmented poses/sec (FK+GAN step), 16-joint batch=65536", "value": value, "unit": "poses/s",
               "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "prewarm_s": a.prewarm, "ms_per_step": t / a.steps * 1e3,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
               "config": {"workload": {"fwd": "FK+Gen+D3+D2 forward", "fk_gen_fwd": "FK+Gen forward",
                                       "gan_step": "full single-frame GAN iteration (WGAN-GP critics x4, G every 5th, Adam)",
                                       "fk": "FK kernel only"}[a.workload],
                          "batch_per_gpu": B, "global_batch": B * world, "dense_dim": D, "preAngle": True,
                          "fk_dtype": "f32", "dense_dtype": "bf16 MFMA, fp32 accumulate"},
               "roofline": roofline, "roofline_fk": roofline_fk, "roofline_layer": roofline_layer, "cpu_baseline": cpu, "extra": extra}
        if flops:
            out["algorithmic_tflops"] = flops * value / 1e12
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def launch_ranks(n):
    """One child process per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in its environment, as torch.distributed.run
    sets them); children are started fresh -- nothing is exec'ed from a process that holds the GPU.  Returns the exit
    code (0 only if every rank succeeded); rank 0's stdout (the JSON line) is passed through."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    return rc


def pmc_traffic(kernel_substr, prefix=""):
    """HBM bytes per launch of a kernel from the committed rocprofv3 PMC summaries (profiles/r01_pmc_*_summary.csv,
    collected by tools/collect_profiles.sh with separate --pmc FETCH_SIZE / WRITE_SIZE passes).  The largest dispatch
    of the kernel is the one bench.py times; prefix "d3_" selects the passes that ran the 3D critic's launch alone (the
    three networks share one kernel name).  gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts half of the
    bytes of wide coalesced reads -> doubled; both counters are in KiB."""
    import csv
    tot = 0.0
    for tag, mult in ((prefix + "fetch", 2.0), (prefix + "write", 1.0)):
        path = os.path.join(ROOT, "profiles", "r01_pmc_%s_summary.csv" % tag)
        if not os.path.exists(path):
            return None
        hit = [float(r["max"]) for r in csv.DictReader(open(path)) if kernel_substr in r["kernel"]]
        if not hit:
            return None
        tot += mult * max(hit) * 1024.0
    return tot


def usable_cores():
    """host cores this process may actually use (cgroup quota / affinity), not the machine's core count"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return max(1, min(n, 64))


def cpu_baseline(workload, D, sdG, sd3, sd2, quat, trans, cam9):
    """The oracle (CPU restatement, 'port') timed on this box's host cores on a bounded sample of the workload."""
    from oracle import dhaug_oracle as O
    cores = usable_cores()
    torch.set_num_threads(cores)
    Bs = 4096
    g = torch.Generator().manual_seed(0)
    z = torch.randn(Bs, 128, generator=g)
    bl = torch.rand(Bs, 15, generator=g) * 0.4 + 0.1
    sc = torch.randint(-200, 200, (Bs, 8), generator=g) / 1000.0
    q, tr, c9 = torch.tensor([quat]), torch.tensor([trans]), torch.tensor([cam9]).repeat(Bs, 1)
    ang = (torch.rand(Bs, 37, generator=g) * 2 - 1) * 180
    rt = torch.randn(Bs, 3, generator=g)

    def one():
        with torch.no_grad():
            if workload == "fk":
                O.fk_forward16(ang, bl, rt)
                return
            fake, _, _ = O.generator_forward(z, sdG, bl, sc)
            if workload == "fk_gen_fwd":
                return
            fw = fake.reshape(-1, 16, 3)
            O.d3_forward(fw - fw[:, :1], sd3)
            O.d2_forward(O.project_to_2d(O.world_to_camera(fw, q, tr), c9), sd2)

    if workload == "gan_step":
        wl = "fwd"
        workload = "fwd"
        note = "forward part only (FK+Gen+D3+D2); the oracle's full step is timed in tests at small batch"
    else:
        note = ""
    one()
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < 12.0 and n < 2000:
        one()
        n += 1
    dt = time.perf_counter() - t0
    return {"value": Bs * n / dt, "unit": "poses/s", "cores": cores, "kind": "port",
            "sample": "%d batches of %d poses, fp32 torch-CPU oracle, %.1f s %s" % (n, Bs, dt, note)}


if __name__ == "__main__":
    main()
