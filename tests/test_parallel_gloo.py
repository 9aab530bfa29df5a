"""N>1 path on CPU: world_size-2 gloo run of the ray partition + all-gather (dual-space-nerf_amd/parallel.py)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _fake_render(o, d, n, f):
    # any deterministic per-ray function: lets the test check placement after the gather
    s = (o * d).sum(-1) + n - f
    return {"color": torch.stack([s, 2 * s, 3 * s], -1), "disp_map": s + 1, "acc_map": s + 2, "depth_map": s + 3}


def _worker(rank, world, port, R, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("par", os.path.join(root, "dual-space-nerf_amd", "parallel.py"))
    par = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(par)
    g = torch.Generator().manual_seed(7)
    o, d = torch.randn(R, 3, generator=g), torch.randn(R, 3, generator=g)
    n, f = torch.rand(R, generator=g), torch.rand(R, generator=g) + 1
    rp = par.RayParallel()
    s, e = rp.shard(R)
    out = rp.render(_fake_render, o, d, n, f)
    ref = _fake_render(o, d, n, f)
    ok = all(torch.equal(out[k], ref[k]) for k in ref)
    for tile in (1, 3, 64):                      # round-robin tiles: same pixels, every ray owned exactly once
        out2 = rp.render_tiled(_fake_render, o, d, n, f, tile=tile)
        ok = ok and all(torch.equal(out2[k], ref[k]) for k in ref)
        owned = torch.cat([rp.tile_indices(R, tile, r) for r in range(world)])
        ok = ok and torch.equal(torch.sort(owned).values, torch.arange(R))
    # cost-balanced contiguous blocks (round 5, bench.py --strong's default partition): ragged blocks, equal slabs, same pixels; a
    # second frame with the same cuts builds nothing
    for cost in (None, torch.arange(R).float() ** 2 + 0.1, torch.zeros(R), torch.cat([torch.zeros(R - 1), torch.ones(1)])):
        out3 = rp.render_blocks(_fake_render, o, d, n, f, cost=cost)
        ok = ok and all(torch.equal(out3[k], ref[k]) for k in ref)
        b = rp.balanced_bounds(torch.ones(R) if cost is None else cost, world, align=1)
        ok = ok and b[0] == 0 and b[-1] == R and len(b) == world + 1 and all(b[i] <= b[i + 1] for i in range(world))
    builds = rp.plan_builds
    rp.render_blocks(_fake_render, o, d, n, f, cost=None)
    ok = ok and rp.plan_builds == builds
    # ranks that bring DIFFERENT costs (each measured its own) still exchange matching slabs: rank 0's cuts are broadcast inside
    # render_blocks; the same cost object a second time costs no collective and no plan
    mine = torch.arange(R).float() * (1.0 + 3.0 * rank) + (R - torch.arange(R)).float() * (3.0 - 2.0 * rank)
    out4 = rp.render_blocks(_fake_render, o, d, n, f, cost=mine)
    ok = ok and all(torch.equal(out4[k], ref[k]) for k in ref)
    want = rp.balanced_bounds(torch.arange(R).float() + (R - torch.arange(R)).float() * 3.0, world)      # rank 0's cost
    ok = ok and rp._agreed[2] == want
    builds = rp.plan_builds
    rp.render_blocks(_fake_render, o, d, n, f, cost=mine)
    ok = ok and rp.plan_builds == builds
    # (measured re-balancing: the block that took twice as long gives rays away)
    if R >= 8:
        b0 = rp.balanced_bounds(torch.ones(R), 2, align=1)
        b1 = rp.rebalance_bounds(b0, [2.0, 1.0], align=1)
        ok = ok and b1[1] < b0[1]
    q.put((rank, s, e, ok))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("R", [10, 7, 1, 200])
def test_ray_parallel_world2(R):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, R, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, s0, e0, ok0), (r1, s1, e1, ok1) = res
    assert ok0 and ok1
    assert s0 == 0 and e0 == s1 and e1 == R      # contiguous, disjoint, covering


def test_single_process_is_identity():
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("par", os.path.join(root, "dual-space-nerf_amd", "parallel.py"))
    par = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(par)
    rp = par.RayParallel()
    assert rp.world == 1 and rp.shard(9) == (0, 9)
    x = torch.arange(12.0).reshape(4, 3)
    assert rp.gather(x, 4) is x


def _grad_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("par", os.path.join(root, "dual-space-nerf_amd", "parallel.py"))
    par = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(par)
    ps = [torch.nn.Parameter(torch.zeros(3, 4)), torch.nn.Parameter(torch.zeros(5)), torch.nn.Parameter(torch.zeros(2, 2))]
    ps[0].grad = torch.full((3, 4), float(rank + 1))
    ps[1].grad = torch.arange(5.0) * (rank + 1)
    # ps[2] has no gradient on rank 0 (counts as zeros), ones on rank 1
    if rank == 1:
        ps[2].grad = torch.ones(2, 2)
    par.RayParallel().average_gradients(ps)
    ok = (torch.allclose(ps[0].grad, torch.full((3, 4), 1.5)) and torch.allclose(ps[1].grad, torch.arange(5.0) * 1.5)
          and torch.allclose(ps[2].grad, torch.full((2, 2), 0.5)))
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_average_world2():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res)



def _frames_worker(rank, world, port, n_frames, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("par", os.path.join(root, "dual-space-nerf_amd", "parallel.py"))
    par = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(par)
    rp = par.RayParallel()
    P = 12
    rendered = []

    def frame(f):                                 # a frame's packed image as a function of its index only
        rendered.append(f)
        return (torch.arange(P * 6, dtype=torch.float32).reshape(P, 6) + 1000.0 * f)

    frames = rp.render_frames(frame, n_frames, P)
    ok = len(frames) == n_frames and all(torch.equal(frames[f], torch.arange(P * 6, dtype=torch.float32).reshape(P, 6) + 1000.0 * f)
                                         for f in range(n_frames))
    ok = ok and rendered == rp.frames_of(n_frames)             # every rank rendered exactly its own frames, in order
    # strong mode's exchange (bench.py --strong): equal slabs gathered, then put back into ray order
    R, tile = 23, 4
    mine = rp.tile_indices(R, tile)
    slab = rp.tile_slab(R, tile)
    pad = torch.zeros(slab, 2)
    pad[: mine.numel(), 0] = mine.float()
    pad[: mine.numel(), 1] = 10.0 * mine.float()
    allp = torch.empty(world * slab, 2)
    dist.all_gather_into_tensor(allp, pad)
    full = rp.undeal_tiles(allp, R, tile)
    ok = ok and torch.equal(full[:, 0], torch.arange(R, dtype=torch.float32)) and torch.equal(full[:, 1], 10.0 * torch.arange(R, dtype=torch.float32))
    # the index tensors of a partition are built (and uploaded) ONCE per (R, tile, world, device) - a second
    # frame of the same shape enumerates nothing: neither tile_indices nor the plan builder run again
    builds = rp.plan_builds
    calls = {"n": 0}
    orig = rp.tile_indices
    rp.tile_indices = lambda *a, **k: (calls.__setitem__("n", calls["n"] + 1), orig(*a, **k))[1]
    o = torch.arange(R, dtype=torch.float32)[:, None].expand(R, 3)
    fake = lambda o_, d_, n_, f_: {"color": o_ * 2.0, "disp_map": o_[:, 0] + 1, "acc_map": o_[:, 0] + 2, "depth_map": o_[:, 0] + 3}
    first = rp.render_tiled(fake, o, o, o[:, 0], o[:, 0], tile=tile)
    b1, c1 = rp.plan_builds, calls["n"]
    second = rp.render_tiled(fake, o, o, o[:, 0], o[:, 0], tile=tile)
    full2 = rp.undeal_tiles(allp, R, tile)
    ok = ok and rp.plan_builds == b1 == builds and calls["n"] == c1 == 0      # (the plan of this R / tile exists since undeal_tiles above)
    ok = ok and torch.equal(first["color"], o * 2.0) and torch.equal(second["depth_map"], o[:, 0] + 3) and torch.equal(full2, full)
    # a rank without a frame joins the collective with a buffer on the GROUP's device (gloo: the CPU, whatever CUDA says)
    ok = ok and rp._group_device() == torch.device("cpu")
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_frames", [5, 4, 2, 1])      # (1: fewer frames than ranks - rank 1 joins with zeros, nobody raises)
def test_frame_parallel_and_tile_undeal_world2(n_frames):
    """BASELINE configs[4] (a sequence dealt frame by frame over the ranks, one asynchronous all-gather per round) and the
    exchange of configs[3] (round-robin tiles of ONE frame: gather of equal slabs + un-dealing into ray order)"""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_frames_worker, args=(r, 2, port, n_frames, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res)
