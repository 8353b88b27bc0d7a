#!/usr/bin/env python3
"""bench.py -- throughput of the DH-AUG hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload fwd|fk_gen_fwd|gan_step|video|fk] [--batch 65536]

One "step" = one pass of the hot path over one batch of synthetic input that is already resident in HBM:
    fwd        (default) FK + Gen + D3 + D2 forward, B = 65 536 poses per GPU, D = 256, fp32 FK
               (north_star's target workload; superset of BASELINE.json configs[1]).  Timed in BOTH arithmetics of the dense
               layers: `value` = bf16 (one MFMA pass), `value_parity` = f16x3 (fp16 hi+lo operands, three MFMA terms --
               the mode that meets the 1e-4 logit tolerance against the fp32 reference, tests/test_gpu_models.py)
    fk_gen_fwd FK + Gen forward only (configs[1] exactly)
    gan_step   full single-frame GAN iteration (configs[2]/[3]): 2+2 WGAN-GP critic steps (flip copies), G step every
               5th iteration, fused Adam; with N > 1 one RCCL all-reduce per optimizer step
    video      multi-frame GAN iteration (configs[4]): B = 512 clips x R = 9 frames, DenseDim 1000, four critics
    fk         the FK kernel alone on B poses
N > 1: one process per GPU.  Launched by torch.distributed.run (RANK / WORLD_SIZE in the environment) the process is a rank;
launched plainly with --gpus N it starts the N ranks itself, before touching the GPU.  The batch shards across ranks
(B per rank, weak scaling); the forward workloads have no exchange step, the training workloads all-reduce the flat
gradient bucket of the network being stepped.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_BF16_PEAK_TFLOPS = 2500.0     # dense bf16 / fp16, MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0              # HBM3E spec, MI355X_MICROARCH.md (6.29 TB/s measured copy ceiling)
FK_BYTES_PER_POSE = 412            # 37+15+3 fp32 in, 48 fp32 out
WORKLOADS = {"fwd": "FK+Gen+D3+D2 forward", "fk_gen_fwd": "FK+Gen forward",
             "gan_step": "full single-frame GAN iteration (WGAN-GP critics x4, G every 5th, Adam)",
             "video": "multi-frame GAN iteration (B clips x R frames; D3, D2 + two motion critics, G every 5th, Adam)",
             "fk": "FK kernel only"}


# ------------------------------------------------------------------------------------------------ FLOP accounting
def mac_per_pose(D, R=1):
    """multiply-accumulates per pose of one forward pass, from the layer shapes"""
    gen = (128 * D + 6 * D * D + 35 * R * D) / R
    d3 = 78 * D + 12 * D * D + 200 * D + 2 * 100 * 100 + 100
    d2 = 32 * D + 4 * D * D + D
    return gen, d3, d2


def mac_motion(D, R):
    """per CLIP: the two motion critics (R/models_Fk_GAN/Fk_discriminator.py:381-587)"""
    blocks = 6 * D * D
    m3 = (R * 15 + (R - 1) * 15 + R * 48 + (R - 1) * 48) * D + 4 * blocks + 4 * D * 100 + 2 * 100 * 100 + 100
    m2 = (R * 32 + (R - 1) * 2) * D + 2 * blocks + 2 * D * 100 + 2 * 100 * 100 + 100
    return m3, m2


def critic_step_flops(mac, mac_first, mac_out, rows):
    """the explicit critic step (dhaug_amd/critic_step.py): forward 3B rows, backward chain 3B (the input layers only on the
    B interpolated rows), tangent sweep B rows (no logit layer), weight gradients 3B rows"""
    return 2.0 * rows * (3 * mac + 3 * (mac - mac_first) + mac_first + (mac - mac_out) + 3 * mac)


def critic_step_bytes(widths, rows, in_cols):
    """ALGORITHMIC HBM bytes of one explicit critic step: every layer output y (bf16) is written once by the forward sweep and
    read once by its weight-gradient contraction, every cotangent gz (bf16) is written once by the backward sweep and read
    once by the contraction, over the 3B rows [real; fake; x_hat] (the tangents replace y on the x_hat rows in place); the
    fp32 inputs are read once.  widths = output widths of all layers.  What the sweeps move on top of this (masks, skips and
    cotangents re-read layer by layer) is the traffic / algorithmic ratio of roofline_step."""
    return 3.0 * rows * (8.0 * sum(widths) + 4.0 * in_cols)


def step_algorithmic_bytes(D, B):
    """one single-frame GAN iteration (2 + 2 critic steps; the sampling pass and the G step every fifth iteration are
    compute-resident: their inputs / outputs only)"""
    d3 = critic_step_bytes([D] * 14 + [100, 100, 100, 1], B, 48 + 30)
    d2 = critic_step_bytes([D] * 5 + [1], B, 32)
    gen = B * (128 * 4 + 48 * 4 + 15 * 4)
    return 2 * d3 + 2 * d2 + 1.2 * gen


def video_algorithmic_bytes(Dd3, Dd2, Dm, B, R):
    """one video GAN iteration (per GPU): 2 + 2 single-frame critic steps over B R frames, 4 + 4 motion-critic steps over B clips
    (the 2D motion critic's penalty runs over frames, its layers over clips).  Per optimizer step: activations / cotangents as in
    critic_step_bytes, plus what scales with the PARAMETERS and dominates at DenseDim 1000 and 512-row batches: the bf16 weights
    read once by each of the four sweeps (8 B), the fp32 weight gradient written once (4 B), Adam (read p, g, m, v; write p, m, v:
    28 B) and the two bf16 operand copies rewritten (4 B) = 44 B per parameter."""
    N = B * R
    p3 = mac_per_pose(Dd3)[1]
    p2 = mac_per_pose(Dd2)[2]
    pm3, pm2 = mac_motion(Dm, R)
    d3 = critic_step_bytes([Dd3] * 14 + [100, 100, 100, 1], N, 48 + 30) + 44.0 * p3
    d2 = critic_step_bytes([Dd2] * 5 + [1], N, 32) + 44.0 * p2
    m3 = critic_step_bytes([Dm] * 28 + [100, 100, 100, 1], B, R * 15 + (R - 1) * 15 + R * 48 + (R - 1) * 48) + 44.0 * pm3
    m2 = critic_step_bytes([Dm] * 14 + [100, 100, 100, 1], B, R * 32 + (R - 1) * 2) + 44.0 * pm2
    gen = N * (48 * 4 + 15 * 4) + B * 128 * 4
    return 2 * d3 + 2 * d2 + 4 * m3 + 4 * m2 + 1.2 * gen


def profile_stats(name, kernel_substr):
    """(mean_us, min_us, calls, file) of a kernel in the TRACKED rocprofv3 summary profiles/<PROFILE_TAG>_<name>_kernel_stats.csv
    (tools/collect_profiles.sh), or None: what the judge recomputes the roofline fractions from"""
    import csv
    path = os.path.join(ROOT, "profiles", "%s_%s_kernel_stats.csv" % (PROFILE_TAG, name))
    if not os.path.exists(path):
        return None
    for r in csv.DictReader(open(path)):
        if kernel_substr in r["Name"]:
            return {"profile_mean_us": float(r["AverageNs"]) * 1e-3, "profile_min_us": float(r["MinNs"]) * 1e-3,
                    "profile_calls": int(r["Calls"]), "profile_file": "profiles/%s_%s_kernel_stats.csv" % (PROFILE_TAG, name)}
    return None


def with_profile(block, name, kernel_substr, work):
    """adds the tracked profile's durations to a roofline block (work = flops or bytes per launch; block["peak"] in T or G
    units per second).  `frac` / `achieved` / `avg_us` stay what THIS run measured with HIP events (a regression of the kernel
    moves them); the figures that follow from the TRACKED profile's launch durations -- what a reader recomputes from
    profiles/, collected under rocprofv3 with the build stamped in <tag>_STAMP.txt -- stand beside them as frac_profile_mean /
    frac_profile_min / achieved_profile_mean.  The two must agree up to the profiler's lower clocks (a few per cent)."""
    ps = profile_stats(name, kernel_substr)
    block["frac_source"] = "HIP events of this run (avg_us)"
    if ps:
        unit = 1e12 if block["unit"] == "TFLOP/s" else 1e9
        block.update(ps)
        block["frac_profile_mean"] = work / (ps["profile_mean_us"] * 1e-6) / unit / block["peak"]
        block["frac_profile_min"] = work / (ps["profile_min_us"] * 1e-6) / unit / block["peak"]
        block["achieved_profile_mean"] = work / (ps["profile_mean_us"] * 1e-6) / unit
        block["profile_agrees"] = bool(abs(block["frac_profile_mean"] / block["frac"] - 1.0) <= 0.15)
    return block


def event_time(fn, iters, warm, rewarm_s=0.3):
    """average duration of fn (seconds), HIP events on the stream the kernels are launched on; rewarm_s of fn first so
    that the kernel is timed at the clocks it runs at inside the loaded step, not at those left by the previous phase"""
    import torch
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


def dist_info(dist, torch, backend, world, local, device_name):
    """what the exchange ran on: backend, world size, the collective library's version and every rank's device -- so that the
    driver can see that RCCL ("nccl" on ROCm) saw N ranks on N devices.  Never raises."""
    try:
        names = [None] * world
        dist.all_gather_object(names, "%s:%d %s" % (os.uname().nodename, local, device_name))
        ver = torch.cuda.nccl.version() if backend == "nccl" else None
        return {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "rccl_version": list(ver) if ver else None,
                "rank_devices": names, "hsa_ipc_legacy": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}
    except Exception as ex:                                       # noqa: BLE001 (reported, not raised)
        return {"error": repr(ex)[:200]}


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
    # poll all ranks: as soon as one exits non-zero the others (stuck in a collective by then) are terminated
    rc, alive = 0, list(procs)
    while alive:
        time.sleep(0.2)
        for p in list(alive):
            c = p.poll()
            if c is None:
                continue
            alive.remove(p)
            rc = max(rc, abs(c))
        if rc != 0 and alive:
            for p in alive:
                p.terminate()                    # (exact children only)
            t_end = time.time() + 10
            for p in alive:
                try:
                    p.wait(timeout=max(0.1, t_end - time.time()))
                except Exception:
                    p.kill()
            break
    return rc


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--reps", type=int, default=5,
                    help="the block of --steps timed steps is repeated this many times (after ONE warm-up): `value` / `ms_per_step` "
                         "are the MEDIAN block's, value_min / value_max the slowest / fastest block's")
    ap.add_argument("--prewarm", type=float, default=1.0,
                    help="seconds of the workload run before the W warmup steps, untimed: MI355X clocks take a few hundred ms "
                         "of load to leave their idle state (the same step measures 0.28 ms right after start-up, 0.25 ms warm)")
    ap.add_argument("--workload", default="fwd", choices=list(WORKLOADS))
    ap.add_argument("--batch", type=int, default=None, help="poses per GPU per step (video: clips per GPU); default 65536 (512)")
    ap.add_argument("--dense", type=int, default=None, help="DenseDim (default 256; video 1000)")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "parity"],
                    help="arithmetic of the dense layers in the forward workloads that `value` is measured in; the other one "
                         "is reported beside it (value_parity / value_bf16)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true")
    ap.add_argument("--no-roofline", action="store_true", help="skip the per-kernel roofline timing loops (profiling runs)")
    ap.add_argument("--graph", default="auto", choices=["auto", "on", "off"],
                    help="training workloads: replay the iteration as captured hipGraphs (auto = on; with several ranks the "
                         "graph is cut at the all-reduces)")
    ap.add_argument("--dry-run", action="store_true",
                    help="rendezvous only: every rank joins the process group, the timing exchange runs, rank 0 prints the line "
                         "(no GPU work; with DHAUG_DIST_BACKEND=gloo this rehearses the N-rank launch path on a CPU box)")
    return ap.parse_args(argv)


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: this process becomes the launcher.  It has not touched the GPU (importing
        # torch does not initialise HIP) and never will: it starts N rank processes and relays rank 0's JSON line.
        sys.exit(launch_ranks(a.gpus))
    # (the pool's host driver only supports dmabuf IPC: without this RCCL's first exchange fails with hipIpcGetMemHandle; it is exported
    # on the boxes already -- kept here for a launcher that starts the ranks with a clean environment)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and world != a.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with --nproc-per-node %d)" % (a.gpus, world, a.gpus))
    backend = os.environ.get("DHAUG_DIST_BACKEND", "nccl")        # "nccl" = RCCL over xGMI on ROCm; gloo: CPU rehearsal
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend)
    if a.dry_run:
        tt = torch.tensor([float(rank + 1)], dtype=torch.float64)
        info = None
        if world > 1:
            dist.barrier()
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            info = dist_info(dist, torch, backend, world, local, "cpu (dry run)")
            dist.destroy_process_group()
        if rank == 0:
            line = {"metric": "augmented poses/sec (FK+GAN step), 16-joint batch=65536", "dry_run": True,
                    "n_gpus": world, "max_over_ranks": tt.item(), "backend": backend if world > 1 else None, "dist": info}
            if a.workload in ("gan_step", "video"):
                # what the exchange step of this workload moves: the flat gradient bucket of every network (host-side: the
                # modules are only constructed), and what a real N-rank run of it prints beside `value`
                from dhaug_amd.function_aug.config import synth_args as _sa
                from dhaug_amd.models_Fk_GAN import Fk_discriminator as _dis, Fk_generator as _gen
                vid = a.workload == "video"
                Dd = a.dense if a.dense is not None else (1000 if vid else 256)
                Rr = 9 if vid else 1
                aa = _sa(a.batch or (512 if vid else 65536), Dd, **(dict(single_or_multi_train_mode="multi", architecture="3,3",
                                                                        video_Dis_DenseDim_3D=Dd, video_Dis_DenseDim_2D=Dd) if vid else {}))
                nets = {"G": (_gen.Video_Fk_Generator(Rr, None, aa, "cpu") if vid else _gen.Fk_Generator(None, aa, "cpu")),
                        "d3d": _dis.Fk_3D_Discriminator("cpu", aa), "d2d": _dis.Fk_2D_Discriminator(aa, 16)}
                if vid:
                    nets["motion_d3d"] = _dis.Video_motion_Fk_3D_Discriminator("cpu", aa, Rr)
                    nets["motion_d2d"] = _dis.Video_motion_Fk_2D_Discriminator("cpu", aa, Rr)
                line["allreduce_bytes_per_optimizer_step"] = {k: 4 * sum(p.numel() for p in m.parameters()) for k, m in nets.items()}
                line["optimizer_steps_per_iteration"] = ({"d3d": 2, "d2d": 2, "G": 0.2} if not vid else
                                                         {"d3d": 2, "d2d": 2, "motion_d3d": 4, "motion_d2d": 4, "G": 0.2})
                line["multi_rank_fields"] = ["value_gan_step", "gan_step_ms_per_step_max_over_ranks", "allreduce_alone",
                                             "allreduce_us_per_optimizer_step", "allreduce_share_of_gan_step_upper_bound", "dist",
                                             "graph_calibration"]
                line["graph_calibration_plan"] = ("eager iteration timed first; segmented graphs tried inside try/except in the same "
                                                  "process; used only if every rank succeeded and it is faster")
                line["hip_graph"] = "segmented (graph | all-reduce | graph ...)" if a.graph != "off" else False
            print(json.dumps(line))
        return
    local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    import dhaug_amd
    from dhaug_amd import _lib, ops, parallel
    from dhaug_amd.function_aug.config import synth_args
    from dhaug_amd.models_Fk_GAN import model_fk_gan_train as T
    from dhaug_amd.models_Fk_GAN import video_GAN_fun as V
    from dhaug_amd.models_Fk_GAN.forward_kinematics_DH_model import Forward_Kinematics_DH_Model
    from dhaug_amd.models_Fk_GAN.Fk_discriminator import score_fake_pair
    from dhaug_amd.common.camera import camera_params9
    from dhaug_amd.common.h36m_dataset import h36m_cameras_extrinsic_params, h36m_cameras_intrinsic_params
    _lib.lib()

    video = a.workload == "video"
    B = a.batch if a.batch is not None else (512 if video else 65536)
    D = a.dense if a.dense is not None else (1000 if video else 256)
    R = 9 if video else 1
    if video:
        args = synth_args(B, D, single_or_multi_train_mode="multi", architecture="3,3", video_Dis_DenseDim_3D=D,
                          video_Dis_DenseDim_2D=D, single_dis_warmup_epoch=0)
    else:
        args = synth_args(B, D)
    torch.manual_seed(1234 + rank)
    fk = Forward_Kinematics_DH_Model(args, ["S1"], None)
    models = T.video_mode_my_get_poseFk_model(args, None, fk, R) if video else T.my_get_poseFk_model(args, None, fk)
    G, D3, D2 = models["model_G"], models["model_d3d"], models["model_d2d"]
    opts = [v for k, v in models.items() if k.startswith("optimizer")]
    if world > 1:                                # data-parallel replicas start from rank 0's weights
        parallel.broadcast_optimizers(opts, 0)
    bucket_bytes = {k[len("optimizer_"):]: v.flat_grad.numel() * 4 for k, v in models.items() if k.startswith("optimizer")}

    # synthetic inputs, resident in HBM
    N = B * R
    ext = h36m_cameras_extrinsic_params["S1"][0]
    quat, trans = [float(v) for v in ext["orientation"]], [float(v) / 1000.0 for v in ext["translation"]]
    cam9 = camera_params9(h36m_cameras_intrinsic_params[0])
    ang = (torch.randn(N, 37, device=dev) * 40).clamp(-180, 180)
    bl = torch.rand(N, 15, device=dev) * 0.4 + 0.1
    real_world = ops.fk_forward(ang, bl, torch.randn(N, 3, device=dev).clamp(-10, 10) * 0.3)
    real_cam, real_2d = ops.world_to_camera_project(real_world, quat, trans, cam9)
    cam_param = torch.zeros(B, 16, device=dev)
    cam_param[:, 9:13] = torch.tensor(quat, device=dev)
    cam_param[:, 13:16] = torch.tensor(trans, device=dev)
    G.GAN_generator_get_bone_length(real_cam)
    z = torch.randn(B, 128, device=dev)
    root = torch.randn(N, 3, device=dev)
    it = [0]
    summary = argparse.Namespace(epoch=10, train_iter_num=0)

    def set_precision(p):
        for m in (G, D3, D2):
            m.precision = p

    def step_fk():
        ops.fk_forward(ang, bl, root)

    def step_fk_gen():
        with torch.no_grad():
            return G(z)

    def step_fwd():
        with torch.no_grad():
            bf = G.precision == "bf16"
            fw, xc, kcs, p2 = G.sample_for_critics(z, (quat, trans, cam9), inputs_bf16=bf)   # FK tail + critic inputs, one launch
            l3, l2 = score_fake_pair(D3, D2, xc, kcs, p2)                                    # both critics, one launch
        return l3, l2

    # one rank: the iteration is one hipGraph; several ranks: graph segments with the all-reduces between them
    # (graphs.SegmentedCall).  --graph off: eager
    use_graph = a.graph in ("on", "auto")
    graphed = None
    if use_graph and a.workload in ("gan_step", "video"):
        from dhaug_amd.graphs import GraphedGanIteration
        graphed = GraphedGanIteration(V.video_gan_iteration if video else T.gan_iteration, args, models, ["S1"], summary)
    vin3 = real_cam.reshape(B, R, 16, 3) if video else real_cam
    vin2 = real_2d.reshape(B, R, 16, 2) if video else real_2d

    mode = {"graph": graphed is not None}                   # (auto, one rank: settled by a calibration below)

    def step_gan():
        g = it[0] % 5 == 4
        if mode["graph"]:
            graphed(vin3, cam_param, vin2, g, (quat, trans, cam9))
        else:
            T.gan_iteration(args, models, real_cam, cam_param, real_2d, ["S1"], summary=None, writer=None,
                            do_g_step=g, camera=(quat, trans, cam9))
        it[0] += 1

    def step_video():
        g = it[0] % 5 == 4
        if mode["graph"]:
            graphed(vin3, cam_param, vin2, g, (quat, trans, cam9))
        else:
            V.video_gan_iteration(args, models, vin3, cam_param, vin2, ["S1"], summary, None, do_g_step=g,
                                  camera=(quat, trans, cam9))
        it[0] += 1

    steps = {"fk": step_fk, "fk_gen_fwd": step_fk_gen, "fwd": step_fwd, "gan_step": step_gan, "video": step_video}

    def timed(fn, k, w):
        for _ in range(w):
            fn()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        c0 = _lib.CALLS[0]
        t0 = time.perf_counter()
        for _ in range(k):
            fn()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t = time.perf_counter() - t0
        calls = (_lib.CALLS[0] - c0) / k
        if world > 1:
            tt = torch.tensor([t], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            t = tt.item()
        return t, calls

    def timed_reps(fn, k, w, reps):
        """`reps` blocks of exactly k steps, each bracketed like timed() (barrier + synchronize on both sides, MAX over ranks);
        one warm-up in front of the first.  Returns (median block time, calls per step, sorted block times)."""
        ts, calls = [], 0.0
        for r in range(max(1, reps)):
            t, calls = timed(fn, k, w if r == 0 else 0)
            ts.append(t)
        ts.sort()
        return ts[len(ts) // 2], calls, ts

    def prewarm(fn, training):
        if a.prewarm <= 0:
            return
        if world > 1 and training:
            # a training step holds a collective: every rank must run the same number of them, so the pre-warm is a step
            # count here (~prewarm seconds at the single-GPU rates), not a wall-clock loop
            for _ in range(max(1, int(a.prewarm * 80))):
                fn()
            torch.cuda.synchronize()
            return
        t_end = time.perf_counter() + a.prewarm
        while time.perf_counter() < t_end:
            for _ in range(20 if not training else 2):
                fn()
            torch.cuda.synchronize()

    training = a.workload in ("gan_step", "video")
    fwd_like = a.workload in ("fwd", "fk_gen_fwd")
    # (--precision parity: the forward workloads in "f16x3", the training workloads in "bf16x6" -- the arithmetic their goldens pass in)
    main_prec = "f16x3" if (a.precision == "parity" and fwd_like) else ("bf16x6" if (a.precision == "parity" and training) else "bf16")
    set_precision(main_prec)
    prewarm(steps[a.workload], training)
    graph_calibration = None
    if training and graphed is not None and a.graph == "auto" and world > 1:
        # several ranks: the segmented-graph form (graphs.SegmentedCall: graph | all-reduce | graph ...) has only ever run over
        # gloo.  The eager iteration is timed FIRST (it is the fallback), then the graph form is tried inside try / except, in
        # this process (a process that has touched the GPU is never replaced); every rank must have succeeded for it to be used.
        cal = {}
        mode["graph"] = False
        it[0] = 0
        tc, _ = timed(steps[a.workload], 10, 5)
        cal["eager_ms"] = tc / 10 * 1e3
        # (1) CAPTURE on every rank, no replay, no eager warm-up in front of it: the capture issues no collective (FusedAdam
        #     records where its all-reduces go), so a rank that fails here leaves no peer waiting inside one;
        # (2) every rank, unconditionally, joins ONE MIN all-reduce of the outcome -- the collective sequence is the same on
        #     the failure path;
        # (3) only if all ranks captured is the graph form (whose replay holds the all-reduces) run and timed.  A rank that
        #     fails INSIDE a replay cannot be recovered from (its peers are in a collective): it reports and exits non-zero,
        #     the launcher ends the others.  Nothing is re-executed in this process.
        ok = 1.0
        try:
            for gstep in (False, True):
                graphed.prepare(vin3, cam_param, vin2, gstep, (quat, trans, cam9), warmup=0)
        except Exception as ex:                                   # noqa: BLE001
            cal["graph_error"] = repr(ex)[:300]
            ok = 0.0
        flag = torch.tensor([ok], device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        ok = flag.item() > 0
        cal["captured_on_every_rank"] = bool(ok)
        if ok:
            try:
                mode["graph"] = True
                it[0] = 0
                tc, _ = timed(steps[a.workload], 10, 10)
                cal["graph_ms"] = tc / 10 * 1e3
            except Exception as ex:                               # noqa: BLE001
                sys.stderr.write("bench.py rank %d: segmented-graph replay failed (%r); peers are inside a collective -- exiting\n" % (rank, ex))
                sys.stderr.flush()
                os._exit(4)
        mode["graph"] = bool(ok and cal.get("graph_ms", 1e30) <= cal["eager_ms"])
        cal["picked"] = "graph" if mode["graph"] else "eager"
        graph_calibration = cal
        it[0] = 0
    if training and graphed is not None and a.graph == "auto" and world == 1:
        # eager or hipGraph?  On a fast host the eager single-frame iteration wins (it overlaps the first part of the weight
        # gradients with the tangent sweep, which a capture cannot: critic_step.TN_SPLIT), on a slow host or at ~1 400
        # launches per iteration (video) the graph does: ten iterations each (two G steps), the faster one is timed
        cal = {}
        for name, flag in (("graph", True), ("eager", False)):
            mode["graph"] = flag
            it[0] = 0
            tc, _ = timed(steps[a.workload], 10, 5)
            cal[name] = tc / 10 * 1e3
        mode["graph"] = cal["graph"] <= cal["eager"]
        graph_calibration = {"graph_ms": cal["graph"], "eager_ms": cal["eager"], "picked": "graph" if mode["graph"] else "eager"}
        it[0] = 0
    t, calls, blocks = timed_reps(steps[a.workload], a.steps, a.warmup, a.reps)
    value = N * world * a.steps / t
    out = {"metric": "augmented poses/sec (FK+GAN step), 16-joint batch=65536", "value": value, "unit": "poses/s",
           "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "prewarm_s": a.prewarm, "ms_per_step": t / a.steps * 1e3,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": {"bf16": "bf16", "f16x3": "f16x3 (fp16 hi+lo operands, fp32 accumulate)",
                     "bf16x6": "bf16x6 (bf16 hi+mid+lo operands, six MFMA terms, fp32 activations)"}[main_prec], "data": "synthetic",
           "config": {"workload": WORKLOADS[a.workload], "batch_per_gpu": B, "frames": R, "poses_per_gpu_per_step": N,
                      "global_batch": N * world, "dense_dim": D, "preAngle": True, "fk_dtype": "f32",
                      "dense_dtype": {"bf16": "bf16 MFMA, fp32 accumulate", "f16x3": "3 x fp16 MFMA (hi+lo operands), fp32 accumulate",
                                      "bf16x6": "6 x bf16 MFMA (hi+mid+lo operands), fp32 accumulate"}[main_prec]},
           "c_abi_calls_per_step": calls, "hip_graph": bool(mode["graph"]), "graph_calibration": graph_calibration,
           # spread of the timed blocks: `value` is the median block; the slowest / fastest block beside it
           "reps": len(blocks), "value_min": N * world * a.steps / blocks[-1], "value_max": N * world * a.steps / blocks[0],
           "ms_per_step_min": blocks[0] / a.steps * 1e3, "ms_per_step_max": blocks[-1] / a.steps * 1e3}

    # the forward workload in the OTHER arithmetic, same inputs (bf16 <-> parity), with its own step time
    if fwd_like:
        other = "bf16" if main_prec == "f16x3" else "f16x3"
        set_precision(other)
        k = max(5, a.steps // 2)
        to, _, ob = timed_reps(steps[a.workload], k, max(3, a.warmup // 2), a.reps)
        key = "value_bf16" if other == "bf16" else "value_parity"
        out[key] = N * world * k / to
        out[key + "_min"], out[key + "_max"] = N * world * k / ob[-1], N * world * k / ob[0]
        out[("ms_per_step_bf16" if other == "bf16" else "ms_per_step_parity")] = to / k * 1e3
        out["parity_mode"] = ("f16x3: logits <= 1e-4 rel / poses <= 1e-5 m vs the fp32 reference "
                              "(tests/test_gpu_models.py::test_fused_forward_vs_reference_golden)")
        set_precision(main_prec)
    if world > 1:
        # what the exchange ran on: the driver sees that the collective library saw N ranks on N devices
        out["dist"] = dist_info(dist, torch, backend, world, local, torch.cuda.get_device_name(local))
    if training and world > 1:
        out["allreduce_bytes_per_optimizer_step"] = bucket_bytes
    if world > 1:
        # the exchange step of the training path, timed alone: all-reduce of the flat gradient buckets (D2 / G / D3 at D = 256,
        # and a DenseDim-1000 motion critic's) over the job's backend ("nccl" = RCCL over xGMI).  Never fails the bench line.
        try:
            import time as _t
            sizes = (1_090_000, 1_750_000, 3_530_000, 100_000_000)
            res = []
            for nb in sizes:
                ar_buf = torch.zeros(nb // 4, dtype=torch.float32, device=dev if backend == "nccl" else "cpu")
                for _ in range(3):
                    dist.all_reduce(ar_buf)
                if backend == "nccl":
                    torch.cuda.synchronize()
                dist.barrier()
                t0 = _t.perf_counter()
                ar_k = 10 if nb < 50_000_000 else 5
                for _ in range(ar_k):
                    dist.all_reduce(ar_buf)
                if backend == "nccl":
                    torch.cuda.synchronize()
                dt = (_t.perf_counter() - t0) / ar_k
                res.append({"bytes": nb // 4 * 4, "us": dt * 1e6, "busbw_GBps": 2.0 * (world - 1) / world * (nb // 4 * 4) / dt / 1e9})
            out["allreduce_alone"] = {"backend": backend, "world": world, "sizes": res}
        except Exception as ex:                                  # noqa: BLE001 (reported, not raised)
            out["allreduce_alone"] = {"error": repr(ex)}

    extra = {}
    if not a.no_extra and not video:
        set_precision("bf16")
        for name in ("fk_gen_fwd", "fwd", "gan_step"):
            if name != a.workload:
                k = max(5, min(a.steps, 50 if name != "gan_step" else 10))
                try:
                    te, ce = timed(steps[name], k, 5 if name == "gan_step" else 3)
                    if name == "gan_step" and world == 1 and a.graph == "auto":
                        # the same iterations as captured hipGraphs: the faster form is the one reported (as --workload gan_step
                        # would pick it); which one wins depends on the box's host (eager issues ~110-130 C-ABI calls per iteration)
                        try:
                            from dhaug_amd.graphs import GraphedGanIteration
                            gg = GraphedGanIteration(T.gan_iteration, args, models, ["S1"], summary)

                            def step_graphed():
                                gg(real_cam, cam_param, real_2d, it[0] % 5 == 4, (quat, trans, cam9))
                                it[0] += 1
                            it[0] = 0
                            tg, cg = timed(step_graphed, k, 10)              # (warm-up covers both graphs: with / without the G step)
                            extra["gan_step_eager_ms_per_step"] = te / k * 1e3
                            extra["gan_step_graph_ms_per_step"] = tg / k * 1e3
                            extra["gan_step_hip_graph"] = bool(tg < te)
                            if tg < te:
                                te, ce = tg, cg
                            del gg
                        except Exception as ex:
                            extra["gan_step_graph_error"] = repr(ex)[:200]
                    extra[name + "_poses_per_s"] = N * world * k / te
                    extra[name + "_ms_per_step"] = te / k * 1e3
                    extra[name + "_c_abi_calls_per_step"] = ce
                except Exception as ex:              # never lose the headline line to an optional measurement
                    extra[name + "_error"] = repr(ex)[:200]
        if "gan_step_ms_per_step" in extra and world == 1:
            # the training step in the arithmetic the loop goldens pass in (bf16x6: six MFMA terms per product, fp32
            # activations, layer by layer; tests/test_gpu_loops.py), eager
            try:
                set_precision("bf16x6")
                it6 = lambda: T.gan_iteration(args, models, real_cam, cam_param, real_2d, ["S1"], summary=None, writer=None,
                                              do_g_step=False, camera=(quat, trans, cam9))
                t6, _ = timed(it6, 3, 1)
                extra["gan_step_parity_ms_per_step"] = t6 / 3 * 1e3
                extra["gan_step_parity_note"] = "bf16x6 critics + generator, critic steps only (no G step in these 3 iterations)"
            except Exception as ex:
                extra["gan_step_parity_error"] = repr(ex)[:200]
            set_precision("bf16")
        set_precision(main_prec)
        if world == 1 and a.workload == "fwd" and D != 1000:
            # the same forward at the reference's DEFAULT width (DenseDim 1000: R/function_aug/config.py:101-109), bf16, layer by
            # layer (the fused programs cover 64 / 128 / 256): trunk of G + both critics on B poses
            try:
                a1k = synth_args(B, 1000)
                m1k = T.my_get_poseFk_model(a1k, None, Forward_Kinematics_DH_Model(a1k, ["S1"], None))
                g1k, d31k, d21k = m1k["model_G"], m1k["model_d3d"], m1k["model_d2d"]
                x31k = torch.randn(B, 16, 3, device=dev) * 0.3
                x21k = torch.rand(B, 16, 2, device=dev) - 0.5

                def fwd1k():
                    with torch.no_grad():
                        g1k.trunk(z); d31k(x31k); d21k(x21k)
                t1k, _ = timed(fwd1k, 10, 3)
                mg, m3_, m2_ = mac_per_pose(1000, 1)
                extra["fwd_D1000_ms_per_step"] = t1k / 10 * 1e3
                extra["fwd_D1000_poses_per_s"] = B * 10 / t1k
                extra["fwd_D1000_frac_of_mfma_peak"] = 2.0 * (mg + m3_ + m2_) * B * 10 / t1k / 2.5e15
                extra["fwd_D1000_note"] = ("G trunk + D3 + D2, DenseDim 1000, bf16, layer GEMMs (256 x 256 x 64 ping-pong kernel, "
                                           "csrc/dhaug_gemm_p8.hip: a 1000-wide layer has 500 flop per activation byte -- matrix-bound "
                                           "layer by layer, no cross-layer fusion needed), eager")
                # the same pass in the arithmetics that meet the 1e-4 logit tolerance at this width (tests/test_gpu_loops.py video_D1000):
                # "f16x3" = IEEE-half pairs, three product terms, as layer GEMMs on the ping-pong tiles (dhaug_gemm_f16x3; the fused parity
                # programs' arithmetic, which stop at DenseDim 256), and "bf16x6" (six bf16 product terms; the training paths' parity mode)
                for prec_, key_ in (("f16x3", "fwd_D1000_parity"), ("bf16x6", "fwd_D1000_bf16x6")):
                    for m_ in (g1k, d31k, d21k):
                        m_.precision = prec_
                    t1p, _ = timed(fwd1k, 3, 1)
                    extra[key_ + "_ms_per_step"] = t1p / 3 * 1e3
                    extra[key_ + "_poses_per_s"] = B * 3 / t1p
                    extra[key_ + "_frac_of_mfma_peak_algorithmic"] = 2.0 * (mg + m3_ + m2_) * B * 3 / t1p / 2.5e15
                extra["fwd_D1000_parity_note"] = ("f16x3 layer GEMMs (fp16 hi + lo operands, K' = 3 K, fp32 activations; narrow layers in bf16x6): "
                                                  "logits <= 1e-4 rel at DenseDim 1000; algorithmic flops -- the kernels execute 3x")
                del m1k, g1k, d31k, d21k
            except Exception as ex:
                extra["fwd_D1000_error"] = repr(ex)[:200]
    if not a.no_extra and not video and world == 1 and a.workload == "fwd":
        # the video iteration at BASELINE.json configs[4]'s per-GPU shape (B = 512 clips x R = 9 frames, DenseDim 1000, the two
        # frame critics + the two motion critics, G step every fifth iteration): eager and as hipGraphs, ten iterations each
        # (two G steps), so that the driver's default line carries it
        try:
            from dhaug_amd.graphs import GraphedGanIteration
            Bv, Dv, Rv = 512, 1000, 9
            Nv = Bv * Rv
            av = synth_args(Bv, Dv, single_or_multi_train_mode="multi", architecture="3,3", video_Dis_DenseDim_3D=Dv,
                            video_Dis_DenseDim_2D=Dv, single_dis_warmup_epoch=0)
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
            itv = [0]

            def video_eager():
                V.video_gan_iteration(av, mv, v3, cpv, v2, ["S1"], sv, None, do_g_step=itv[0] % 5 == 4, camera=(quat, trans, cam9))
                itv[0] += 1

            def video_graph():
                gv(v3, cpv, v2, itv[0] % 5 == 4, (quat, trans, cam9))
                itv[0] += 1
            tve, cve = timed(video_eager, 10, 5)
            itv[0] = 0
            tvg, _ = timed(video_graph, 10, 10)
            tvb = min(tve, tvg) / 10
            extra["video_eager_ms_per_step"], extra["video_graph_ms_per_step"] = tve / 10 * 1e3, tvg / 10 * 1e3
            extra["video_ms_per_step"], extra["video_hip_graph"] = tvb * 1e3, bool(tvg < tve)
            extra["video_poses_per_s"] = Nv / tvb
            extra["video_c_abi_calls_per_step_eager"] = cve
            extra["video_config"] = {"batch_clips": Bv, "frames": Rv, "dense_dim": Dv, "critics": 4}
            by_v = video_algorithmic_bytes(Dv, Dv, Dv, Bv, Rv)
            trv = pmc_step_traffic("video")
            out["roofline_video_step"] = {"kernel": "one video GAN iteration (B = 512 x R = 9, DenseDim 1000): 2 + 2 frame-critic and 4 + 4 motion-critic steps, sampling, G step every 5th",
                                          "bound": "hbm", "achieved": by_v / tvb / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                          "frac": by_v / tvb / 1e9 / HBM_PEAK_GBS, "traffic": trv, "traffic_source": pmc_stamp(),
                                          "traffic_note": "traffic = PMC total of the TRACKED profile (not this run); achieved / frac = this run's time",
                                          "algorithmic_bytes_per_iteration": by_v, "traffic_over_algorithmic": (trv / by_v) if trv else None,
                                          "ms_per_iteration": tvb * 1e3}
            del gv, mv
        except Exception as ex:                      # never lose the headline line to an optional measurement ...
            extra["video_error"] = repr(ex)[:300]
    # ... but never hide its failure either: every optional measurement that raised is named at the top level of the line
    failed = sorted(k for k in extra if k.endswith("_error"))
    if failed:
        out["extra_errors"] = {k: extra[k] for k in failed}
    out["extra"] = extra

    gen_mac, d3_mac, d2_mac = mac_per_pose(D, R)
    # FLOPs of the timed workload from the layer shapes
    if a.workload == "gan_step":
        d3_first, d2_first = 78 * D, 32 * D
        per_it = (2 * critic_step_flops(d3_mac, d3_first, 100, B) + 2 * critic_step_flops(d2_mac, d2_first, D, B)
                  + 2.0 * gen_mac * B                                              # sampling pass
                  + 0.2 * (3 * 2.0 * (gen_mac + d3_mac + d2_mac) * B + 2.0 * (d3_mac + d2_mac) * B))   # G step (+ flipped evaluations)
        flops = per_it
    elif a.workload == "video":
        m3, m2 = mac_motion(D, R)
        dd3, dd2 = mac_per_pose(args.Dis_DenseDim_3D)[1], mac_per_pose(args.Dis_DenseDim_2D)[2]
        gmac = mac_per_pose(args.Gen_DenseDim, R)[0]
        d3_first, d2_first = 78 * args.Dis_DenseDim_3D, 32 * args.Dis_DenseDim_2D
        # the motion critics take the explicit four-sweep schedule too (critic_step.step_m3 / step_m2): input layers only on
        # the B interpolated clips in the backward chain
        m3_first, m2_first = (R * 15 + (R - 1) * 15 + R * 48 + (R - 1) * 48) * D, (R * 32 + (R - 1) * 2) * D
        per_it = (2 * critic_step_flops(dd3, d3_first, 100, N) + 2 * critic_step_flops(dd2, d2_first, args.Dis_DenseDim_2D, N)
                  + 4 * critic_step_flops(m3, m3_first, 100, B) + 4 * critic_step_flops(m2, m2_first, 100, B) + 2.0 * gmac * N
                  + 0.2 * (3 * 2.0 * ((gmac + dd3 + dd2) * N + 2 * (m3 + m2) * B) + 2.0 * ((dd3 + dd2) * N + 2 * (m3 + m2) * B)))
        flops = per_it
    else:
        flops = {"fk": 2.5e3 * N, "fk_gen_fwd": 2.0 * gen_mac * N, "fwd": 2.0 * (gen_mac + d3_mac + d2_mac) * N}[a.workload]
    out["algorithmic_tflops"] = flops * world * a.steps / t / 1e12
    out["algorithmic_flop_per_step_per_gpu"] = flops
    if video:
        by_it = video_algorithmic_bytes(args.Dis_DenseDim_3D, args.Dis_DenseDim_2D, D, B, R)
        tr = pmc_step_traffic("video") if (B, D, R) == (512, 1000, 9) else None
        tv = t / a.steps
        out["roofline_step"] = {"kernel": "one video GAN iteration: 2 + 2 single-frame and 4 + 4 motion critic steps, sampling, G step every 5th",
                                "bound": "hbm", "achieved": by_it / tv / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": by_it / tv / 1e9 / HBM_PEAK_GBS, "traffic": tr, "traffic_source": pmc_stamp(),
                                "algorithmic_bytes_per_iteration": by_it, "traffic_over_algorithmic": (tr / by_it) if tr else None,
                                "ms_per_iteration": tv * 1e3, "tflops": flops / tv / 1e12,
                                "mfma_frac": flops / tv / 1e12 / MFMA_BF16_PEAK_TFLOPS,
                                "note": "whole video iteration (per GPU).  Algorithmic bytes (video_algorithmic_bytes): activations and "
                                        "cotangents written once / read once, and 44 B per parameter and optimizer step (weights once per "
                                        "sweep, dW, Adam, operand copies) -- at DenseDim 1000 and 512-row batches the parameters dominate"}
    else:
        # the single-frame training step (configs[2]): HBM-bound layer sweeps.  From the timed workload if that is gan_step,
        # else from the extra measurement of the same run.
        ts = (t / a.steps) if a.workload == "gan_step" else (extra.get("gan_step_ms_per_step", 0.0) * 1e-3)
        if ts > 0:
            d3_first, d2_first = 78 * D, 32 * D
            fl_it = (2 * critic_step_flops(d3_mac, d3_first, 100, B) + 2 * critic_step_flops(d2_mac, d2_first, D, B) + 2.0 * gen_mac * B
                     + 0.2 * (3 * 2.0 * (gen_mac + d3_mac + d2_mac) * B + 2.0 * (d3_mac + d2_mac) * B))
            by_it = step_algorithmic_bytes(D, B)
            tr = pmc_step_traffic("step") if (B, D) == (65536, 256) else None
            out["roofline_step"] = {"kernel": "one single-frame GAN iteration: 2 + 2 explicit critic steps, sampling pass, G step every 5th",
                                    "bound": "hbm", "achieved": by_it / ts / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                    "frac": by_it / ts / 1e9 / HBM_PEAK_GBS, "traffic": tr, "traffic_source": pmc_stamp(),
                                    "traffic_over_algorithmic": (tr / by_it) if tr else None,
                                    "algorithmic_bytes_per_iteration": by_it, "ms_per_iteration": ts * 1e3,
                                    "poses_per_s": N * world / ts, "tflops": fl_it / ts / 1e12,
                                    "mfma_frac": fl_it / ts / 1e12 / MFMA_BF16_PEAK_TFLOPS,
                                    "note": "algorithmic bytes: every activation / cotangent written once and read once by its "
                                            "weight-gradient contraction (critic_step_bytes); per GPU"}
    if a.workload == "gan_step":
        out["value_gan_step"] = value
        out["gan_step_ms_per_step_max_over_ranks"] = t / a.steps * 1e3
    if "gan_step_ms_per_step" in extra:
        # the forward workloads have no exchange step, so the scaled quantity with the all-reduce in it is printed too -- at
        # N = 1 as well: the 1 -> 8 curve of the training step needs its anchor under the same name
        out["value_gan_step"] = extra["gan_step_poses_per_s"]
        out["gan_step_ms_per_step_max_over_ranks"] = extra["gan_step_ms_per_step"]
    if world > 1 and "gan_step_ms_per_step" in extra:
        ar = {r["bytes"]: r["us"] for r in out.get("allreduce_alone", {}).get("sizes", [])} if isinstance(out.get("allreduce_alone"), dict) else {}
        if ar:
            near = lambda nb: ar[min(ar, key=lambda k: abs(k - nb))]
            per_opt = {k: near(v) for k, v in bucket_bytes.items()}
            out["allreduce_us_per_optimizer_step"] = per_opt
            comm = 2 * per_opt.get("d3d", 0.0) + 2 * per_opt.get("d2d", 0.0) + 0.2 * per_opt.get("G", 0.0)
            out["allreduce_share_of_gan_step_upper_bound"] = comm * 1e-3 / extra["gan_step_ms_per_step"]

    if rank == 0 and not a.no_roofline and not video:
        from dhaug_amd import fused
        set_precision("bf16")
        x3 = torch.randn(B, 48, device=dev) * 0.3
        kcs_b = ops.kcs_forward(x3, True, f32=False, bf16_ld=32)[1]
        kcs_f = ops.kcs_forward(x3, True, f32=True)[0]
        with torch.no_grad():
            tg = event_time(lambda: fused.critic3d(D3, x3, kcs=kcs_b), 100, 20)
            tp = event_time(lambda: fused.critic3d(D3, x3, kcs=kcs_f, mode="f16x3"), 40, 8)
        fl = 2.0 * d3_mac * B
        out["roofline"] = {"kernel": "fused_mlp_kernel (Fk_3D_Discriminator forward, M=%d, D=%d, 17 layers in one launch)" % (B, D),
                           "bound": "mfma", "achieved": fl / tg / 1e12, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                           "frac": fl / tg / 1e12 / MFMA_BF16_PEAK_TFLOPS,
                           "traffic": pmc_traffic("fused_mlp_kernel", "d3_") if (B, D) == (65536, 256) else None,
                           "traffic_source": pmc_stamp(), "avg_us": tg * 1e6, "algorithmic_flop_per_pose": 2 * d3_mac}
        if (B, D) == (65536, 256):
            with_profile(out["roofline"], "d3", "fused_mlp_kernel<false>", fl)
        out["roofline_parity"] = {"kernel": "fused_mlp_x3_kernel (same program, fp16 hi+lo operands: 3 MFMA terms per product)",
                                  "bound": "mfma", "achieved": fl / tp / 1e12, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                                  "frac": fl / tp / 1e12 / MFMA_BF16_PEAK_TFLOPS, "executed_tflops": 3 * fl / tp / 1e12,
                                  "executed_frac": 3 * fl / tp / 1e12 / MFMA_BF16_PEAK_TFLOPS, "traffic": None, "avg_us": tp * 1e6,
                                  "note": "achieved counts ALGORITHMIC flops (the reference's fp32 layers); the kernel executes 3x"}
        if (B, D) == (65536, 256):
            rp = with_profile(out["roofline_parity"], "d3_parity", "fused_mlp_x3_kernel", fl)
            if "frac_profile_mean" in rp:
                rp["executed_frac_profile_mean"] = 3 * rp["frac_profile_mean"]
        xb = torch.randn(B, D, device=dev).to(torch.bfloat16)
        wb = (torch.randn(D, D, device=dev) / D ** 0.5).to(torch.bfloat16)
        bias = torch.zeros(D, device=dev)
        tl = event_time(lambda: ops.gemm_nt(xb, wb, D, D, bias=bias, res_bf16=xb, act=1, out_bf16=True), 50, 10)
        out["roofline_layer"] = {"kernel": "gemm_nt256s_kernel<16,1> (training path: one M=%d, N=K=%d layer, bias+residual+ReLU)" % (B, D),
                                 "bound": "hbm", "achieved": (3 * B * D * 2 + D * D * 2) / tl / 1e9, "peak": HBM_PEAK_GBS,
                                 "unit": "GB/s", "frac": (3 * B * D * 2 + D * D * 2) / tl / 1e9 / HBM_PEAK_GBS, "traffic": None,
                                 "avg_us": tl * 1e6, "tflops": 2.0 * B * D * D / tl / 1e12}
        nfk = 1 << 22
        a4 = (torch.rand(nfk, 37, device=dev) * 2 - 1) * 180
        b4 = torch.rand(nfk, 15, device=dev) * 0.4 + 0.1
        r4 = torch.randn(nfk, 3, device=dev)
        tf = event_time(lambda: ops.fk_forward(a4, b4, r4), 20, 5)
        tf_b = event_time(step_fk, 50, 10)
        out["roofline_fk"] = {"kernel": "fk_forward_kernel<0,16,true>", "bound": "hbm", "achieved": FK_BYTES_PER_POSE * nfk / tf / 1e9,
                              "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": FK_BYTES_PER_POSE * nfk / tf / 1e9 / HBM_PEAK_GBS,
                              "traffic": pmc_traffic("fk_forward_kernel<0; 16; true>", "fk_"), "traffic_source": pmc_stamp(),
                              "algorithmic_bytes": FK_BYTES_PER_POSE * nfk, "poses_per_launch": nfk, "avg_us": tf * 1e6,
                              "at_batch": {"poses": N, "avg_us": tf_b * 1e6, "achieved": FK_BYTES_PER_POSE * N / tf_b / 1e9}}
        with_profile(out["roofline_fk"], "fk", "fk_forward_kernel<0, 16, true>", FK_BYTES_PER_POSE * nfk)
        del a4, b4, r4
        set_precision(main_prec)

    if rank == 0:
        cpu = None
        if world == 1 and not a.no_cpu_baseline:
            sd = lambda m: {k: v.detach().cpu() for k, v in m.state_dict().items()}
            if video:
                cpu = cpu_baseline_video(D, R, sd(G), sd(D3), sd(D2), sd(models["model_motion_d3d"]), sd(models["model_motion_d2d"]),
                                         quat, trans, cam9)
            else:
                cpu = cpu_baseline(a.workload, D, sd(G), sd(D3), sd(D2), quat, trans, cam9, Bs=B)
                out["cpu_baseline_faithful"] = cpu_baseline(a.workload, D, sd(G), sd(D3), sd(D2), quat, trans, cam9, faithful=True,
                                                            Bs=B, seconds=10.0)
                # the same pair at the reference's own CPU-runnable batch (BASELINE.json configs[0]: 1 024): there the op-by-op FK
                # is dispatch-bound (7 035 small ops per call), so the ratio of the two shows what op fusion alone buys on the
                # CPU -- at 65 536 both are memory-bound and the ratio drowns in the noise
                small = {}
                for key, faithful in (("batched", False), ("faithful", True)):
                    r = cpu_baseline(a.workload, D, sd(G), sd(D3), sd(D2), quat, trans, cam9, faithful=faithful, Bs=1024, seconds=4.0)
                    small[key] = {k: r[k] for k in ("value", "unit", "cores", "sample")}
                small["fusion_ratio"] = small["batched"]["value"] / small["faithful"]["value"]
                out["cpu_baseline_b1024"] = small
        out["cpu_baseline"] = cpu
        print(json.dumps(out))
    if world > 1:
        dist.barrier()                           # rank 0's single-rank extras (roofline kernels) end before anybody tears down
        dist.destroy_process_group()


PROFILE_TAG = next((t for t in ("r06", "r05", "r04", "r03") if os.path.exists(os.path.join(ROOT, "profiles", t + "_STAMP.txt"))), "r03")


def pmc_stamp():
    """which committed profile the `traffic` numbers were read from (they are NOT measured by this run)"""
    path = os.path.join(ROOT, "profiles", "%s_pmc_fetch_summary.csv" % PROFILE_TAG)
    stamp = os.path.join(ROOT, "profiles", "%s_STAMP.txt" % PROFILE_TAG)
    if not os.path.exists(path):
        return None
    return {"file": "profiles/%s_pmc_*_summary.csv" % PROFILE_TAG,
            "collected": open(stamp).read().strip() if os.path.exists(stamp) else "unknown",
            "note": "rocprofv3 --pmc passes of tools/collect_profiles.sh, not this run"}


def pmc_step_traffic(tag):
    """HBM bytes per ITERATION of a training workload (tag 'step' / 'video'): sum over all kernels of the committed
    FETCH_SIZE x 2 + WRITE_SIZE passes of tools/collect_profiles.sh (profiles/<tag>_pmc_<step|video>_totals.json)."""
    path = os.path.join(ROOT, "profiles", "%s_pmc_%s_totals.json" % (PROFILE_TAG, tag))
    if not os.path.exists(path):
        return None
    d = json.load(open(path))
    return d.get("hbm_bytes_per_iteration")


def pmc_traffic(kernel_substr, prefix=""):
    """HBM bytes per launch of a kernel from the committed rocprofv3 PMC summaries (profiles/<tag>_pmc_*_summary.csv,
    collected by tools/collect_profiles.sh with separate --pmc FETCH_SIZE / WRITE_SIZE passes).  The largest dispatch
    of the kernel is the one bench.py times; prefix "d3_" selects the passes that ran the 3D critic's launch alone (the
    three networks share one kernel name).  gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts half of the
    bytes of wide coalesced reads -> doubled; both counters are in KiB."""
    import csv
    tot = 0.0
    for tag, mult in ((prefix + "fetch", 2.0), (prefix + "write", 1.0)):
        path = os.path.join(ROOT, "profiles", "%s_pmc_%s_summary.csv" % (PROFILE_TAG, tag))
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


def _time_cpu(one, poses_per_call, seconds=12.0, note="", kind="port"):
    import torch
    cores = usable_cores()
    torch.set_num_threads(cores)
    one()
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds and n < 2000:
        one()
        n += 1
    dt = time.perf_counter() - t0
    return {"value": poses_per_call * n / dt, "unit": "poses/s", "cores": cores, "kind": kind,
            "sample": "%d batches of %d poses, fp32 torch-CPU oracle, %.1f s %s" % (n, poses_per_call, dt, note)}


def cpu_baseline(workload, D, sdG, sd3, sd2, quat, trans, cam9, faithful=False, Bs=65536, seconds=12.0):
    """The oracle (CPU restatement, 'port') timed on this box's host cores on a bounded sample of the workload, at the
    benchmark's own batch size.  faithful=True: the same arithmetic with the FK issued at the reference's op granularity
    (oracle.fk_forward32_op_by_op: 33 per-joint matrix builds with slice writes, 46 sequential bmm on cloned operands,
    per-coordinate scatters -- SURVEY.md section 8d) instead of the batched restatement: the ratio of the two is what
    op fusion alone buys on the CPU; the rest of the GPU / CPU ratio is hardware."""
    import torch
    from oracle import dhaug_oracle as O
    g = torch.Generator().manual_seed(0)
    z = torch.randn(Bs, 128, generator=g)
    bl = torch.rand(Bs, 15, generator=g) * 0.4 + 0.1
    sc = torch.randint(-200, 200, (Bs, 8), generator=g) / 1000.0
    q, tr, c9 = torch.tensor([quat]), torch.tensor([trans]), torch.tensor([cam9]).repeat(Bs, 1)
    ang = (torch.rand(Bs, 37, generator=g) * 2 - 1) * 180
    rt = torch.randn(Bs, 3, generator=g)
    fk32 = O.fk_forward32_op_by_op if faithful else None

    def one():
        with torch.no_grad():
            if workload == "fk":
                (fk32 or O.fk_forward32)(ang, bl, rt)[:, O.H36M_32_TO_16]
                return
            fake, _, _ = O.generator_forward(z, sdG, bl, sc, fk32=fk32)
            if workload == "fk_gen_fwd":
                return
            fw = fake.reshape(-1, 16, 3)
            O.d3_forward(fw - fw[:, :1], sd3)
            O.d2_forward(O.project_to_2d(O.world_to_camera(fw, q, tr), c9), sd2)

    note = "(reference-faithful op-by-op FK)" if faithful else "(batched restatement)"
    if workload == "gan_step":
        workload = "fwd"
        note += " forward part only (FK+Gen+D3+D2); the oracle's full step is timed in tests at small batch"
    return _time_cpu(one, Bs, seconds=seconds, note=note)


def cpu_baseline_video(D, R, sdG, sd3, sd2, sdm3, sdm2, quat, trans, cam9):
    import torch
    from oracle import dhaug_oracle as O
    Bs = 64
    g = torch.Generator().manual_seed(0)
    z = torch.randn(Bs, 128, generator=g)
    bl = torch.rand(Bs * R, 15, generator=g) * 0.4 + 0.1
    sc = torch.randint(-200, 200, (Bs, 8), generator=g) / 1000.0
    q, tr, c9 = torch.tensor([quat]), torch.tensor([trans]), torch.tensor([cam9]).repeat(Bs * R, 1)

    def one():
        with torch.no_grad():
            fake, _, _ = O.generator_forward(z, sdG, bl, sc, frames=R)
            fw = fake.reshape(-1, 16, 3)
            fc = fw - fw[:, :1]
            p2 = O.project_to_2d(O.world_to_camera(fw, q, tr), c9)
            O.d3_forward(fc, sd3); O.d2_forward(p2, sd2)
            O.motion_d3_forward(fc.reshape(-1, 48), sdm3, R); O.motion_d2_forward(p2.reshape(-1, 32), sdm2, R)

    return _time_cpu(one, Bs * R, note="forward part only (video Gen + D3 + D2 + both motion critics, DenseDim %d)" % D)


if __name__ == "__main__":
    main()
