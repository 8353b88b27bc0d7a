"""world_size-2 gloo test (CPU) of the data-parallel exchange: the flat gradient bucket of FusedAdam is summed over
the replicas, parameters stay views of the flat buffer, shards partition the batch.  (The Adam arithmetic itself is a
HIP kernel and is checked on the GPU; on CPU step() must refuse.)"""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import dhaug_amd
    from dhaug_amd import parallel
    from dhaug_amd.optim import FusedAdam
    r, w, _ = parallel.init_from_env("gloo")
    assert (r, w) == (rank, world) and dist.get_backend() == "gloo"
    torch.manual_seed(0)                                   # identical replicas
    net = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 1))
    opt = FusedAdam(net.parameters())
    n = sum(p.numel() for p in net.parameters())
    assert opt.flat_param.numel() == n and all(p.data.data_ptr() >= opt.flat_param.data_ptr() for p in net.parameters())
    # each rank works on its own shard of a global batch
    g = torch.Generator().manual_seed(7)
    X, Y = torch.randn(10, 8, generator=g), torch.randn(10, 1, generator=g)
    b, e = parallel.shard_range(10, rank, world)
    opt.zero_grad()
    loss = ((net(X[b:e]) - Y[b:e]) ** 2).sum()
    loss.backward()
    local = opt.flat_grad.clone()
    ws = opt.exchange()
    assert ws == world
    # reference: gradient of the loss over the whole batch on one process
    torch.manual_seed(0)
    ref = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 1))
    ((ref(X) - Y) ** 2).sum().backward()
    ref_flat = torch.cat([p.grad.reshape(-1) for p in ref.parameters()])
    ok = torch.allclose(opt.flat_grad, ref_flat, atol=1e-5) and not torch.allclose(local, ref_flat, atol=1e-5)
    # module.zero_grad(set_to_none=True) breaks the views; exchange() must pick the new .grad tensors up again
    net.zero_grad(set_to_none=True)
    ((net(X[b:e]) - Y[b:e]) ** 2).sum().backward()
    opt.exchange()
    ok = ok and torch.allclose(opt.flat_grad, ref_flat, atol=1e-5)
    refused = False
    try:
        opt.step()
    except RuntimeError as ex:
        refused = "no CPU fallback" in str(ex)
    q.put((rank, bool(ok), refused, parallel.rank_seed(5, rank)))
    dist.destroy_process_group()


def test_flat_bucket_allreduce_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[0] for r in res] == [0, 1]
    assert all(r[1] for r in res), "summed gradient bucket does not equal the full-batch gradient"
    assert all(r[2] for r in res), "FusedAdam.step() must refuse CPU parameters"
    assert res[0][3] != res[1][3]


def test_shard_range_partitions():
    sys.path.insert(0, ROOT)
    from dhaug_amd import parallel
    for total, world in ((65536, 8), (10, 3), (7, 8), (512, 2)):
        spans = [parallel.shard_range(total, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == total
        assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
        assert max(e - b for b, e in spans) - min(e - b for b, e in spans) <= 1


def test_bench_gpus_flag_launches_ranks():
    """`python bench.py --gpus 2` (no torchrun around it) must become two rank processes and report n_gpus = 2: the
    launcher path rehearsed on CPU with gloo (--dry-run: rendezvous + the max-over-ranks exchange, no GPU work)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["DHAUG_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["max_over_ranks"] == 2.0 and line["backend"] == "gloo"
    # who took part: the collective library saw 2 ranks (with RCCL this is where its version and the devices appear)
    d = line["dist"]
    assert d["backend"] == "gloo" and d["world_size"] == 2 and len(d["rank_devices"]) == 2 and d["rccl_version"] is None
    # a mismatch between --gpus and an externally set WORLD_SIZE is an error, never a silent n_gpus = 1
    env2 = dict(env, WORLD_SIZE="1", RANK="0")
    r2 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run"], env=env2,
                        stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r2.returncode != 0


def test_bench_gan_step_dry_run_reports_the_exchange():
    """`python bench.py --gpus 2 --workload gan_step --dry-run`: the N-rank launch of the TRAINING workload rehearsed on CPU
    with gloo -- the line names the gradient buckets an optimizer step all-reduces (bytes from the constructed modules),
    the optimizer steps per iteration, and the fields a real run prints for the scaled quantity."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["DHAUG_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--workload", "gan_step", "--dry-run"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["backend"] == "gloo"
    by = line["allreduce_bytes_per_optimizer_step"]
    # 436 771 / 881 585 / 271 873 parameters at DenseDim 256
    assert by == {"G": 4 * 436771, "d3d": 4 * 881585, "d2d": 4 * 271873}
    assert line["optimizer_steps_per_iteration"] == {"d3d": 2, "d2d": 2, "G": 0.2}
    for k in ("value_gan_step", "allreduce_us_per_optimizer_step", "allreduce_share_of_gan_step_upper_bound", "dist", "graph_calibration"):
        assert k in line["multi_rank_fields"]
    assert "segmented" in line["hip_graph"]
