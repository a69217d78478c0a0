"""world_size-2 gloo test of the multi-GPU path's host logic: round-robin column
sharding, per-rank stepping, and the gather-to-root of diagnostics.  The column
arithmetic is done by the CPU oracle here (no GPU in this container); the
sharding / gather code is the same module bench.py uses with RCCL."""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

import common as cm


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ntotal, nz, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    sys.path.insert(0, cm.ROOT)
    from mckpp_f90_amd import sharding
    from oracle import orc

    dist.init_process_group("gloo", rank=rank, world_size=world)
    idx = sharding.shard_indices(ntotal, rank, world)
    oc, ob = cm.make_oracle(len(idx), nz, exp_mode=1, index=idx, ntotal=ntotal, nthreads=1)
    for nt in (1, 2):
        orc.physics_driver(oc, ob, nt, nthreads=1)
    parts = sharding.gather_to_root(ob["hmix"], dist)
    tparts = sharding.gather_to_root(ob["T"], dist)
    if rank == 0:
        q.put((sharding.unshard(parts, ntotal), sharding.unshard(tparts, ntotal)))
    dist.barrier()
    dist.destroy_process_group()


def test_round_robin_shard_and_gather_gloo():
    from oracle import orc

    ntotal, nz, world = 37, 40, 2           # odd total: ranks own 19 and 18 columns
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, ntotal, nz, q)) for r in range(world)]
    for p in procs:
        p.start()
    hmix, T = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    oc, ob = cm.make_oracle(ntotal, nz, exp_mode=1, nthreads=1)
    for nt in (1, 2):
        orc.physics_driver(oc, ob, nt, nthreads=1)
    assert np.array_equal(hmix, ob["hmix"])
    assert np.array_equal(T, ob["T"])


def test_shard_indices_partition():
    from mckpp_f90_amd import sharding

    for ntotal, world in ((100000, 8), (37, 2), (5, 8)):
        parts = [sharding.shard_indices(ntotal, r, world) for r in range(world)]
        allidx = np.sort(np.concatenate(parts))
        assert np.array_equal(allidx, np.arange(ntotal))
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
        vals = [p * 2.0 for p in parts]
        assert np.array_equal(sharding.unshard(vals, ntotal), np.arange(ntotal) * 2.0)


def test_bench_starts_its_own_ranks_without_touching_the_gpu():
    """`python bench.py --gpus N` with no launcher around it becomes the launcher: N ranks under
    torch.distributed.run on 127.0.0.1, started before this process imports torch or loads the library (a process
    that has initialised the GPU must not start the ranks).  Without a GPU the ranks refuse to run and the exit
    code comes back."""
    import json
    import subprocess

    bench = os.path.join(cm.ROOT, "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "3", "--print-launch"], capture_output=True,
                       text=True, env=env, timeout=60)
    assert r.returncode == 0, r.stderr
    plan = json.loads(r.stdout.strip().splitlines()[-1])
    assert plan["torch_imported"] is False and plan["library_loaded"] is False
    cmd = plan["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=2" in cmd and "--nnodes=1" in cmd
    assert "--standalone" in cmd and cmd[cmd.index("--local-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "2", "--steps", "3"]
    # a real launch here (no GPU): both ranks say so, the parent hands the failure on
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "2", "--no-cpu-baseline"], capture_output=True,
                       text=True, env=env, timeout=300)
    assert r.returncode != 0
    assert "MI355X" in r.stderr and '"metric"' not in r.stdout


def test_one_handle_leg_of_the_bench_cannot_take_the_line_down():
    """At N>1 rank 0 also drives all shards through ONE handle of the C-ABI (peer copies, cross-device waits): that leg
    runs in a child process, and whatever becomes of it the parent gets a dictionary to put into its line.  Here the
    child has no GPU: it fails, and the failure comes back as text."""
    import argparse
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(cm.ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    a = argparse.Namespace(steps=2, warmup=1, ncol=64, nz=40, grid="uniform", dto=3600.0, diag=1, total_ncol=0)
    out = bench.run_single_process_leg(a, [0, 1])
    assert isinstance(out, dict) and "error" in out and "child process" in out["error"]


def _guard_worker(rank, world, port, fail_rank, fail_where, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import datetime

    import torch
    import torch.distributed as dist

    sys.path.insert(0, cm.ROOT)
    from mckpp_f90_amd import sharding

    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    entered = []

    def local():
        if fail_where == "local" and rank == fail_rank:
            raise RuntimeError("download failed on this rank alone")
        return rank * 10

    def collective(x):
        entered.append(x)
        t = torch.tensor([float(x)])
        dist.all_reduce(t)
        if fail_where == "collective":
            raise RuntimeError("collective failed everywhere")
        return float(t.item())

    out, err = sharding.guarded_block(dist, local, collective, timeout_s=30.0)
    # whatever happened in the block, the ranks are still in step: the next collective matches up
    t = torch.ones(1)
    dist.all_reduce(t)
    q.put((rank, out, err, len(entered), float(t.item())))
    dist.barrier()
    dist.destroy_process_group()


def _run_guard(fail_rank, fail_where):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_guard_worker, args=(r, world, port, fail_rank, fail_where, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def test_a_failure_on_one_rank_cannot_leave_the_others_inside_a_collective():
    """bench.py's N>1 diagnostics (the gather to rank 0, the config3_strong block) run under sharding.guarded_block:
    when the part that touches only one rank (a download, a kernel run) fails on ONE rank, every rank skips the
    collectives - none is left waiting inside a gather until the process group's timeout - and all of them go on in
    step; when nothing fails the result comes back; a failure inside the collectives is reported on every rank."""
    res = _run_guard(fail_rank=1, fail_where="local")
    for rank, out, err, entered, after in res:
        assert out is None and err is not None and entered == 0 and after == 2.0
    assert "this rank alone" in res[1][2] and "another rank" in res[0][2]
    res = _run_guard(fail_rank=-1, fail_where="none")
    for rank, out, err, entered, after in res:
        assert err is None and out == 10.0 and entered == 1 and after == 2.0
    res = _run_guard(fail_rank=-1, fail_where="collective")
    for rank, out, err, entered, after in res:
        assert out is None and "collective failed" in err and after == 2.0


def test_bench_has_the_config3_block_and_guards_it():
    """The N>1 line carries BASELINE configs[3] itself (config3_strong: 1e5 x 100 dealt over the ranks, 200 steps in one
    call past step 60) and the diagnostics gather, both through sharding.guarded_block, and a state gathered on rank 0
    that is incomplete or not finite fails the run ("ok": false, non-zero exit)."""
    src = open(os.path.join(cm.ROOT, "bench.py")).read()
    assert "def config3_strong_block(" in src and 'multi["config3_strong"] = config3_strong_block(' in src
    assert src.count("sharding.guarded_block(") >= 2 and "os._exit(3)" in src
    assert 'out["ok"] = not' in src and "incomplete or not finite" in src
