"""RCCL readiness that tests itself: on a box with >= 2 GPUs (skipped otherwise -- the build's boxes have one) two fresh rank
processes, one per device, join a process group with backend "nccl" (= RCCL over xGMI on ROCm) and run the data-parallel
single-frame iteration of BASELINE.json configs[3]: three iterations (the G step on the third) eagerly and as SEGMENTED hipGraphs
(graphs.SegmentedCall: graph | all-reduce | graph ...), each rank on its half of the batch.

Checked by every rank: the replicas' parameters stay bit-identical; eager and segmented-graph runs agree; the weights equal those of
ONE rank run over the full batch (mean of shard gradients == full-batch gradient: rank 0 computes that run before it joins the group);
the group really is 2 ranks on 2 different devices (what bench.py's `dist` field reports).  No scaling number is produced: this only
makes the first multi-GPU box yield evidence without a builder in the loop.  (The same schedule over gloo, two replicas sharing one
card: tests/test_gpu_zz_dp.py; over gloo on CPU: tests/test_dp_gloo.py.)"""
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(rank, world, port, q, backend):
    rehearsal = backend != "nccl"                                  # DHAUG_RCCL_TEST_REHEARSE=1 on a one-GPU box: gloo, both ranks on device 0
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0" if rehearsal else str(rank), LOCAL_WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    import golden_util as GU
    import dhaug_amd
    from dhaug_amd import graphs, parallel, autograd_ops as A
    from dhaug_amd.common.camera import camera_params9
    from dhaug_amd.common.h36m_dataset import h36m_cameras_extrinsic_params, h36m_cameras_intrinsic_params
    from dhaug_amd.models_Fk_GAN import forward_kinematics_DH_model as fkm, model_fk_gan_train as train
    from test_gpu_models import make_args
    if rehearsal and rank > 0:
        import time
        time.sleep(20)                                             # (two processes on one card bring the GPU up one after the other)
    torch.cuda.set_device(0 if rehearsal else rank)
    Bg, D, N = 512, 64, 3
    ext = h36m_cameras_extrinsic_params["S1"][0]
    cam = ([float(v) for v in ext["orientation"]], [float(v) / 1000.0 for v in ext["translation"]],
           camera_params9(h36m_cameras_intrinsic_params[0]))
    x3_all = GU.synth_pose16(Bg, seed=3) + torch.tensor([0.0, 0.0, 4.5])
    x2_all = (torch.rand(Bg, 16, 2, generator=torch.Generator().manual_seed(5)) - 0.5) * 1.2
    noise_all = torch.randn(Bg, 128, generator=torch.Generator().manual_seed(1))
    scaler_all = torch.randint(-200, 200, (Bg, 8), generator=torch.Generator().manual_seed(2)) / 1000.0
    alpha_all = torch.rand(Bg, 1, generator=torch.Generator().manual_seed(3))

    def setup(b, e):
        B = e - b
        args = make_args(batch_size=B, Gen_DenseDim=D, Dis_DenseDim_3D=D, Dis_DenseDim_2D=D)
        cp = torch.zeros(B, 16, device="cuda")
        cp[:, 9:13] = torch.tensor(cam[0], device="cuda"); cp[:, 13:16] = torch.tensor(cam[1], device="cuda")
        mk = lambda: train.ConstDraws(noise=[noise_all[b:e].cuda()], scaler=[scaler_all[b:e].cuda()], alpha=[alpha_all[b:e].cuda()])

        def build():
            d = train.my_get_poseFk_model(args, None, fkm.Forward_Kinematics_DH_Model(args, ["S1"], None))
            for key, shapes, seed in (("model_G", GU.shapes_generator(D), 11), ("model_d3d", GU.shapes_d3(D), 12), ("model_d2d", GU.shapes_d2(D), 13)):
                sd = GU.seeded_state_dict(shapes, seed)
                with torch.no_grad():
                    for k, p in d[key].named_parameters():
                        p.copy_(sd[k].cuda())
                d[key].precision = "bf16x6"                      # fp32-grade: shard means against the full batch at 1e-4-level bounds
            A.bump_weight_epoch()
            return d
        return args, x3_all[b:e].cuda(), cp, x2_all[b:e].cuda(), mk, build

    keys = ("optimizer_d3d", "optimizer_d2d", "optimizer_G")
    full = None
    if rank == 0:
        # ONE rank over the full batch, before this process joins the group (no exchange: world size 1)
        args, x3, cp, x2, mk, build = setup(0, Bg)
        d, dr = build(), mk()
        assert d["optimizer_d3d"].world_size() == 1
        for i in range(N):
            train.gan_iteration(args, d, x3, cp, x2, ["S1"], None, None, do_g_step=(i == N - 1), camera=cam, draws=dr)
        full = {k: d[k].flat_param.detach().float().cpu() for k in keys}
        del d
        torch.cuda.synchronize()
    dist.init_process_group(backend)
    parallel.init_from_env(backend)
    bad = []
    # the group: 2 ranks on 2 different devices (bench.py's `dist` field)
    names = [None] * world
    dist.all_gather_object(names, "%s:%d" % (os.uname().nodename, torch.cuda.current_device()))
    if (not rehearsal and len(set(names)) != world) or dist.get_backend() != backend or dist.get_world_size() != world:
        bad.append(("group", names, dist.get_backend(), dist.get_world_size()))
    b, e = parallel.shard_range(Bg, rank, world)
    args, x3, cp, x2, mk, build = setup(b, e)
    de, dr = build(), mk()
    if de["optimizer_d3d"].world_size() != world:
        bad.append(("optimizer world size", de["optimizer_d3d"].world_size()))
    for i in range(N):
        train.gan_iteration(args, de, x3, cp, x2, ["S1"], None, None, do_g_step=(i == N - 1), camera=cam, draws=dr)
    dg, dr2 = build(), mk()
    G = graphs.GraphedGanIteration(train.gan_iteration, args, dg, ["S1"], None)
    for i in range(N):
        G(x3, cp, x2, i == N - 1, cam, draws=dr2)
    torch.cuda.synchronize()
    calls = list(G.graphs.values())
    if not all(isinstance(c, graphs.SegmentedCall) for c in calls):
        bad.append(("not segmented", [type(c).__name__ for c in calls]))
    steps = {"optimizer_d3d": 2 * N, "optimizer_d2d": 2 * N, "optimizer_G": 1}
    for key in keys:
        if int(dg[key].step_dev.item()) != steps[key] or int(de[key].step_dev.item()) != steps[key]:
            bad.append(("steps", key, int(dg[key].step_dev.item()), int(de[key].step_dev.item())))
        a_, b_ = dg[key].flat_param, de[key].flat_param
        if (a_ - b_).abs().max().item() > 2e-5 * b_.abs().max().item() + 1e-12:
            bad.append(("graph vs eager", key, (a_ - b_).abs().max().item()))
        for which, dd in (("eager", de), ("graph", dg)):
            flat = dd[key].flat_param.detach().clone()
            if rehearsal:
                flat = flat.cpu()
            other = [torch.empty_like(flat) for _ in range(world)]
            dist.all_gather(other, flat)                          # (on the devices: RCCL)
            if not all(torch.equal(o, flat) for o in other):
                bad.append(("replicas diverged", which, key))
    if rank == 0:
        for key in keys:
            err = (de[key].flat_param.detach().float().cpu() - full[key]).abs()
            # what one flipped last bit of a near-zero gradient grows into is +-lr per step in that weight; nearly all weights far closer
            if err.max().item() > 2.05e-4 * steps[key] or torch.quantile(err[:: max(1, err.numel() // 200000)], 0.98).item() > 2e-5:
                bad.append(("two ranks vs one rank over the full batch", key, err.max().item()))
    q.put((rank, bad))
    dist.barrier()
    dist.destroy_process_group()


def _worker(rank, world, port, q, backend):
    try:
        _run(rank, world, port, q, backend)
    except BaseException as ex:                                    # report instead of dying silently
        import traceback
        q.put((rank, [("exception", repr(ex), traceback.format_exc())]))
        raise


def test_two_ranks_on_two_devices_over_rccl():
    rehearse = os.environ.get("DHAUG_RCCL_TEST_REHEARSE") == "1"
    if not torch.cuda.is_available() or (torch.cuda.device_count() < 2 and not rehearse):
        pytest.skip("needs two GPUs (RCCL); this box has %d" % (torch.cuda.device_count() if torch.cuda.is_available() else 0))
    backend = "nccl" if torch.cuda.device_count() >= 2 else "gloo"
    import gc
    import queue as _queue
    import time
    gc.collect()
    torch.cuda.empty_cache()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 35500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, backend)) for r in range(2)]
    for p in procs:
        p.start()
    res = []
    try:
        deadline = time.monotonic() + 600
        while len(res) < len(procs) and time.monotonic() < deadline:
            try:
                res.append(q.get(timeout=2))
            except _queue.Empty:
                if [p for p in procs if p.exitcode not in (None, 0)]:
                    break
        res.sort()
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()                                           # exact child only
    assert [r[0] for r in res] == [0, 1], "rank exit codes %s, reported %s" % ([p.exitcode for p in procs], res)
    for r in res:
        assert not r[1], r
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
