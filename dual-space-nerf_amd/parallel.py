"""Ray-parallel multi-GPU execution of the hot path: one process per GPU, rays partitioned into
contiguous blocks, rendered pixels combined by ONE all-gather (RCCL over xGMI on MI355X; gloo in the
CPU tests).  The reference has no distributed code at all; rays are independent
units, so there is no data-path collective besides this exchange of 6 floats per ray
(rgb, disp, acc, depth = 24 B/ray: 786 KB per rank for a 512x512 frame on 8 GPUs - latency-bound, so a
single fused all_gather_into_tensor of the packed [block, 6] tensor is used rather than six small ones).
"""
from __future__ import annotations

import torch
import torch.distributed as dist


class RayParallel:
    def __init__(self, group=None):
        self.group = group
        self.enabled = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if self.enabled else 1
        self.rank = dist.get_rank(group) if self.enabled else 0
        # per (R, tile, world, device): the permutation that un-deals gathered slabs into ray order, this rank's tile indices on the
        # device, the slab size - built once, on the host, uploaded once (they were re-created and re-uploaded per
        # rank on every call: 8 synchronous host-to-device copies per frame at 8 ranks)
        self._plans = {}
        self.plan_builds = 0           # how many plans were built (tests: a second call must not build - or upload - anything)

    def _group_device(self):
        """the device collectives of this group run on: the CPU for gloo, the current CUDA device for nccl (= RCCL)"""
        if self.enabled and dist.get_backend(self.group) == "gloo":
            return torch.device("cpu")
        if torch.cuda.is_available():
            return torch.device("cuda", torch.cuda.current_device())
        return torch.device("cpu")

    def tile_plan(self, R: int, tile: int = 3072, device=None, world: int = None):
        """Cached partition of R rays into round-robin tiles for a group of `world` ranks (default: this group): dict(slab = rows of
        the equal slabs the ranks exchange, mine = this rank's ray indices on `device`, src = for every ray of the frame its row in the
        gathered [world * slab, C] tensor, so that un-dealing is ONE index_select)."""
        world = self.world if world is None else int(world)
        device = torch.device(device) if device is not None else torch.device("cpu")
        key = (int(R), int(tile), world, str(device))
        plan = self._plans.get(key)
        if plan is None:
            self.plan_builds += 1
            idx = [self.tile_indices(R, tile, r, world) for r in range(world)]
            slab = max(i.numel() for i in idx)
            src = torch.empty(R, dtype=torch.int64)
            for r, ir in enumerate(idx):
                src[ir] = r * slab + torch.arange(ir.numel())
            plan = {"slab": slab, "mine": idx[self.rank if self.rank < world else 0].to(device), "src": src.to(device),
                    "counts": [i.numel() for i in idx]}
            self._plans[key] = plan
        return plan

    # ---- cost-balanced contiguous blocks (round 5: the partition of the metric's own 512 x 512 x 64 frame) ----
    # A contiguous block of an image keeps a rank's samples in a compact part of space: the cells of the posed mesh's nearest-face
    # grid it visits are (nearly) its own, so the per-frame list build can be restricted to them and the cell-major search runs on
    # full waves - a round-robin tile share spreads an eighth of the samples over ALL the cells the frame visits.  The blocks are cut
    # where the cumulative per-ray COST (evaluated samples, from a probe frame or from measured share times) crosses k / world.
    @staticmethod
    def balanced_bounds(cost, world: int, align: int = 64):
        """cut points b_0 = 0 <= b_1 <= ... <= b_world = R of `world` contiguous blocks with (nearly) equal sums of cost [R]
        (any non-negative per-ray estimate; all-zero -> equal ray counts).  Cuts are rounded to multiples of `align` rays."""
        c = torch.as_tensor(cost, dtype=torch.float64).reshape(-1).clamp_min(0.0)
        R = c.numel()
        world = int(world)
        if float(c.sum()) <= 0.0:
            c = torch.ones(R, dtype=torch.float64)
        cum = torch.cumsum(c, 0)
        total = float(cum[-1])
        bounds = [0]
        for k in range(1, world):
            b = int(torch.searchsorted(cum, torch.tensor(total * k / world, dtype=torch.float64)))
            b = min(R, max(bounds[-1], ((b + align // 2) // align) * align))
            bounds.append(b)
        bounds.append(R)
        return bounds

    @staticmethod
    def rebalance_bounds(bounds, seconds, align: int = 64):
        """one step of measured re-balancing: block r = [bounds[r], bounds[r+1]) took seconds[r]; with the cost taken as uniform
        inside each block the cuts move to where the cumulative measured time crosses k / world"""
        R = bounds[-1]
        dens = torch.zeros(R, dtype=torch.float64)
        for r, t in enumerate(seconds):
            n = bounds[r + 1] - bounds[r]
            if n > 0:
                dens[bounds[r]:bounds[r + 1]] = float(t) / n
        return RayParallel.balanced_bounds(dens, len(seconds), align)

    def block_plan(self, R: int, bounds, device=None):
        """Cached partition of R rays into the contiguous blocks [bounds[r], bounds[r + 1]): the same dict as tile_plan (slab = rows
        of the equal slabs exchanged = the largest block; mine = this rank's ray indices; src = for every ray its row in the gathered
        [world * slab, C] tensor), so undeal() is the same ONE index_select."""
        world = len(bounds) - 1
        device = torch.device(device) if device is not None else torch.device("cpu")
        key = ("blocks", int(R), tuple(int(b) for b in bounds), str(device))
        plan = self._plans.get(key)
        if plan is None:
            assert bounds[0] == 0 and bounds[-1] == R and all(bounds[i] <= bounds[i + 1] for i in range(world))
            self.plan_builds += 1
            counts = [int(bounds[r + 1] - bounds[r]) for r in range(world)]
            slab = max(counts)
            src = torch.empty(R, dtype=torch.int64)
            for r in range(world):
                src[bounds[r]:bounds[r + 1]] = r * slab + torch.arange(counts[r])
            me = self.rank if self.rank < world else 0
            plan = {"slab": slab, "mine": torch.arange(bounds[me], bounds[me + 1]).to(device), "src": src.to(device), "counts": counts,
                    "bounds": [int(b) for b in bounds]}
            self._plans[key] = plan
        return plan

    @staticmethod
    def undeal(gathered: torch.Tensor, plan: dict, out: torch.Tensor = None) -> torch.Tensor:
        """gathered [world * slab, C] (rank-major slabs) -> [R, C] in ray order through a plan's permutation (tile_plan / block_plan)"""
        if out is None:
            return gathered.index_select(0, plan["src"])
        torch.index_select(gathered, 0, plan["src"], out=out)
        return out

    def block(self, R: int) -> int:
        return (R + self.world - 1) // self.world

    def shard(self, R: int):
        """[start, stop) of this rank's contiguous ray block (last blocks may be short or empty)."""
        b = self.block(R)
        start = min(R, self.rank * b)
        return start, min(R, start + b)

    def gather(self, local: torch.Tensor, R: int) -> torch.Tensor:
        """local [r_local, C] (this rank's block, r_local = stop-start) -> [R, C] on every rank."""
        if self.world == 1:
            return local
        b = self.block(R)
        C = local.shape[1]
        pad = torch.zeros(b, C, dtype=local.dtype, device=local.device)
        pad[: local.shape[0]] = local
        out = torch.empty(self.world * b, C, dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, pad, group=self.group)
        return out[:R]

    # ---- load-balanced partition: tiles of `tile` rays dealt round-robin ----
    # With the transparent skip the cost of a ray depends on where it crosses the body, so contiguous blocks of one
    # frame are uneven (the rows through the torso cost several times the rows above the head).  Dealing small tiles
    # round-robin gives every rank the same mix; the exchange is still ONE all-gather of equal-sized slabs.
    def tile_indices(self, R: int, tile: int = 3072, rank: int = None, world: int = None) -> torch.Tensor:
        """ray indices (ascending) of the tiles owned by `rank` (default: this rank) in a group of `world` ranks (default: this
        group's size; another value lets one process enumerate the shares of a larger job: bench.py --strong --emulate-world)."""
        rank = self.rank if rank is None else rank
        world = self.world if world is None else int(world)
        ntiles = (R + tile - 1) // tile
        mine = torch.arange(rank, max(ntiles, rank), world)      # (a rank beyond the last tile owns nothing)
        idx = (mine[:, None] * tile + torch.arange(tile)[None, :]).reshape(-1)
        return idx[idx < R]

    def tile_slab(self, R: int, tile: int = 3072) -> int:
        """rows of the equal slabs the ranks exchange (= the largest share; rank 0 owns the most)"""
        ntiles = (R + tile - 1) // tile
        n0 = (ntiles + self.world - 1) // self.world          # tiles of rank 0
        last = R - (ntiles - 1) * tile                         # rays of the frame's last tile
        return n0 * tile - ((tile - last) if (ntiles - 1) % self.world == 0 else 0)

    def undeal_tiles(self, gathered: torch.Tensor, R: int, tile: int = 3072, out: torch.Tensor = None, world: int = None) -> torch.Tensor:
        """gathered [world * slab, C] (rank-major slabs as all_gather_into_tensor leaves them) -> [R, C] in ray order: ONE gather
        kernel through the cached permutation (no per-rank loop, no upload after the first call)"""
        plan = self.tile_plan(R, tile, gathered.device, world)
        assert gathered.shape[0] == (self.world if world is None else world) * plan["slab"]
        if out is None:
            return gathered.index_select(0, plan["src"])
        torch.index_select(gathered, 0, plan["src"], out=out)
        return out

    def render_tiled(self, render_fn, ray_o, ray_d, near, far, tile: int = 3072):
        """like render(), with the round-robin tile partition.  NOTE: the geometry-guided sampler takes the FIRST ray's
        origin for the whole batch (utils/pts_utils.py:31), so this is meant for rays of one camera."""
        return self.render_partitioned(render_fn, ray_o, ray_d, near, far, self.tile_plan(ray_o.shape[0], tile, ray_o.device))

    def agreed_bounds(self, bounds):
        """rank 0's cut points on every rank: ONE broadcast of world + 1 integers on the group's device.  The gather of
        render_partitioned() exchanges equal slabs whose size follows from the cuts - ranks that cut differently (a cost measured on
        their own GPU, a borderline comparison falling differently) would post mismatched collectives: a hang or corrupt pixels."""
        if self.world == 1:
            return [int(b) for b in bounds]
        t = torch.tensor([int(b) for b in bounds], dtype=torch.int64, device=self._group_device())
        dist.broadcast(t, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)
        return [int(b) for b in t.cpu()]

    def render_blocks(self, render_fn, ray_o, ray_d, near, far, cost=None):
        """like render(), with contiguous blocks cut for equal cost (cost [R]: any per-ray estimate, e.g. the evaluated samples of a
        previous frame of the sequence; None = equal ray counts).  The ranks need NOT pass identical costs: rank 0's cuts are
        broadcast (one small collective per NEW cost - the cuts of a cost tensor seen before, same object and same
        version, are reused, so a sequence rendered with one estimate pays for it once)."""
        R = ray_o.shape[0]
        key = None if cost is None else (id(cost), getattr(cost, "_version", None), R)
        cached = getattr(self, "_agreed", None)
        if cached is not None and cached[0] == key and cached[1] == R:
            bounds = cached[2]
        else:
            bounds = self.balanced_bounds(torch.ones(R) if cost is None else cost, self.world)
            if cost is not None:                     # (equal ray counts are a function of R and the world size alone)
                bounds = self.agreed_bounds(bounds)
            self._agreed = (key, R, bounds)
        return self.render_partitioned(render_fn, ray_o, ray_d, near, far, self.block_plan(R, bounds, ray_o.device))

    def render_partitioned(self, render_fn, ray_o, ray_d, near, far, plan: dict):
        """this rank renders plan["mine"], ONE all-gather of equal slabs, ONE index_select into ray order (plan: tile_plan / block_plan)"""
        R = ray_o.shape[0]
        dev = ray_o.device
        idx = plan["mine"]
        if idx.numel():
            loc = render_fn(ray_o[idx].contiguous(), ray_d[idx].contiguous(), near[idx].contiguous(), far[idx].contiguous())
            packed = torch.cat([loc["color"], loc["disp_map"][:, None], loc["acc_map"][:, None], loc["depth_map"][:, None]], dim=1)
        else:
            packed = torch.zeros(0, 6, dtype=torch.float32, device=dev)
        if self.world == 1:
            full = packed
        else:
            slab = plan["slab"]
            pad = torch.zeros(slab, 6, dtype=torch.float32, device=dev)
            pad[: packed.shape[0]] = packed
            allp = torch.empty(self.world * slab, 6, dtype=torch.float32, device=dev)
            dist.all_gather_into_tensor(allp, pad, group=self.group)
            full = self.undeal(allp, plan)
        return {"color": full[:, 0:3], "disp_map": full[:, 3], "acc_map": full[:, 4], "depth_map": full[:, 5]}

    # ---- multi-frame batches (BASELINE configs[4]: novel-pose sequences, novel_pose_vis.py:41-66) ----
    def frames_of(self, n_frames: int, rank: int = None):
        """frame indices of `rank` (default: this rank): frames are dealt round-robin, frame f -> rank f % world"""
        rank = self.rank if rank is None else rank
        return list(range(rank, n_frames, self.world))

    def render_frames(self, render_frame_fn, n_frames: int, pixels: int, channels: int = 6, device=None, dtype=torch.float32):
        """Every rank renders its frames of a sequence (render_frame_fn(f) -> [pixels, channels] device tensor, e.g. the packed
        rgb / disp / acc / depth image of Renderer.render_view(device_output=True)); after each round of `world` frames ONE
        all_gather_into_tensor brings that round's images to every rank - issued asynchronously, so the exchange of round k
        overlaps the rendering of round k + 1.  Returns the list of n_frames images in sequence order on every rank.
        Fewer frames than ranks is fine: a rank without a frame in a round joins that round's collective with zeros of `dtype` on
        `device` (default: the device the GROUP's backend runs on - the CPU for gloo, the current CUDA device for nccl - so a
        frameless rank of a gloo group on a CUDA host does not bring a CUDA tensor to a CPU collective) - every rank
        takes the same path through the collectives whatever n_frames is, so no rank can raise while the others wait."""
        if n_frames <= 0:
            return []
        rounds = (n_frames + self.world - 1) // self.world
        outs, works = [], []
        dev = torch.device(device) if device is not None else None
        for r in range(rounds):
            f = r * self.world + self.rank
            img = render_frame_fn(f) if f < n_frames else None
            if dev is None:
                dev = img.device if img is not None else self._group_device()
            if img is None:
                img = torch.zeros(pixels, channels, dtype=dtype, device=dev)
            img = img.reshape(pixels, channels).contiguous()
            if self.world == 1:
                outs.append(img[None])
                continue
            buf = torch.empty(self.world, pixels, channels, dtype=img.dtype, device=dev)
            works.append(dist.all_gather_into_tensor(buf.reshape(self.world * pixels, channels), img, group=self.group, async_op=True))
            outs.append(buf)
        for w in works:
            w.wait()
        frames = [outs[f // self.world][f % self.world] for f in range(n_frames)]
        return frames

    def render(self, render_fn, ray_o, ray_d, near, far):
        """render_fn(ray_o, ray_d, near, far) -> dict(color [r,3], disp_map [r], acc_map [r], depth_map [r])
        on this rank's block; returns the same dict for all R rays on every rank."""
        R = ray_o.shape[0]
        s, e = self.shard(R)
        if e > s:
            loc = render_fn(ray_o[s:e].contiguous(), ray_d[s:e].contiguous(), near[s:e].contiguous(),
                            far[s:e].contiguous())
            packed = torch.cat([loc["color"], loc["disp_map"][:, None], loc["acc_map"][:, None],
                                loc["depth_map"][:, None]], dim=1)
        else:
            packed = torch.zeros(0, 6, dtype=torch.float32, device=ray_o.device)
        full = self.gather(packed, R)
        return {"color": full[:, 0:3], "disp_map": full[:, 3], "acc_map": full[:, 4], "depth_map": full[:, 5]}

    def average_gradients(self, parameters):
        """Data-parallel training (every rank renders its own ray batch of the same step): ONE all-reduce of a
        single flat bucket holding all 33 gradients (500 021 floats = 2 MB - latency-bound on xGMI, so one
        collective instead of 33), then the mean.  Parameters without a gradient contribute zeros."""
        params = [p for p in parameters]
        if self.world == 1 or not params:
            return
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        flat /= self.world
        off = 0
        for p in params:
            n = p.numel()
            g = flat[off:off + n].reshape(p.shape)
            if p.grad is None:
                p.grad = g.clone()
            else:
                p.grad.copy_(g)
            off += n

