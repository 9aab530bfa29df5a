"""Host-side mirror of the reference's Renderer (the upper drop-in boundary).

Same constructor, attributes and methods as /root/reference/can_render.py:14-406 so that
trainer.py:70 (`render.render(batch)`), test.py:60 / validate.py:50 (`render.render_view(batch)`),
utils/visualizer.py:47-66 (`w2l_without_lbs`, `query_volume`, `.canonical_model`) and
validate.py:27 (`render.net.load_state_dict`) can use it unchanged.  Orchestration is Python; every
number is produced by libdsnerf_hip.so (no torch math, no fallback).

What differs from the reference, on purpose (DESIGN.md "boundary"):
  * a frame is rendered in one pass over ALL its rays (the reference loops over 3072-ray chunks with
    an empty_cache + 6 device->host copies each, can_render.py:172-245); one device->host copy at the end;
  * in eval mode the networks are evaluated only on non-transparent samples (their sigma is zeroed and
    their colour has weight 0 in the reference, can_render.py:115-120) - outputs are identical;
  * `render()` in train mode returns tensors attached to ONE autograd node whose backward is
    dsn_render_rays_grad (analytic parameter gradients, csrc/dsn_train.hip) instead of an op-by-op graph;
  * `render_views(batches)` (not in the reference): the per-frame loop of novel_pose_vis.py:41-66 with several frames in
    flight on their own HIP streams.

Eval-mode safety nets: the plain-fp16 density screen runs with a margin CALIBRATED for the loaded
parameters (first eval frame after the parameters changed: `screen_info`), can be audited on every frame
(`screen_audit = True`, `last_screen_audit()`), and samples whose activations leave the fp16 range of the split-fp16
kernels are re-evaluated in exact fp32 inside the library.  `early_stop` ("auto" | True | False): eval frames front to back
in slices of 4-8 samples with ray termination below a transmittance of eps(S, c) = min(2^-20, 1e-4 / (2 (S + 1) max(1, c)))
(DSN_EARLY_STOP: within (S + 1) eps c = 5e-5 ABSOLUTE of the one-pass frame for colours up to the scale c) - "auto" decides per
parameter version from the statistics of the first eval frame, which also measures the colours (c = 2 x the largest colour that
frame weighed; later frames are watched and raise it), without a wait.  The one default-on feature that is error-bounded instead of
bit-identical: `last_frame_info` says per frame whether it ran and with which threshold.
"""
from __future__ import annotations

import warnings
import weakref

import numpy as np
import torch

from . import _lib


def load_bodydata(model_type="smpl", gender="neutral", model_path=""):
    """utils/smpl_utils.py:3-14: unpickle the SMPL model (dict with 'f', 'weights', 'kintree_table')."""
    import os
    import pickle

    if os.path.isdir(model_path):
        model_path = os.path.join(model_path, "{}_{}.pkl".format(model_type.upper(), gender.upper()))
    assert os.path.exists(model_path), "Path {} does not exist!".format(model_path)
    with open(model_path, "rb") as f:
        return pickle.load(f, encoding="latin1")


_OUT_KEYS = ("color", "disp_map", "acc_map", "depth_map", "weights", "z_vals")
SCREEN_AUDIT_EVERY = 8        # Renderer.screen_audit = "auto": one eval frame in this many carries the density screen's audit


class _HostPoolGuard:
    """Keeps torch's intra-op OpenMP pool from starving the thread that feeds the GPU (VERDICT r02 missing #4).

    The reference's callers run torch CPU ops on the main thread between frames (test.py:61-76: torch.clamp, psnr, .cpu()).
    On the GPU boxes such an op wakes the whole libgomp team - torch.get_num_threads() = 128-256 threads - and every one of
    them then busy-waits at the pool's dock for GOMP_SPINCOUNT iterations (milliseconds) for a next parallel region: a few
    hundred CPU-milliseconds of spinning per op, right while render_view stages its uploads, enqueues the frame and waits for
    it (36 ms per 512 x 512 frame instead of 19, BENCH_r02).  libgomp reads its wait policy once, at load, so it cannot be made
    passive from here; but when a parallel region starts with a SMALLER team, libgomp retires the pool's surplus threads at
    once (gomp_team_start: the dock barrier is re-initialised, threads without a new task leave their loop).  So: cap the
    team at `limit` threads, run one tiny parallel region, render, and put the caller's thread count back - the next big
    torch op of the caller re-creates its team (~10 us per thread).  limit = None / pool already small: nothing happens."""

    _poke = None

    def __init__(self, limit):
        self.limit = limit
        self.saved = None

    def __enter__(self):
        lim = self.limit
        if lim is None or not torch.get_num_threads() > max(2, int(lim)):
            return self
        self.saved = torch.get_num_threads()
        torch.set_num_threads(max(2, int(lim)))
        if _HostPoolGuard._poke is None:
            _HostPoolGuard._poke = torch.zeros(1 << 17)          # > 2 x at::internal::GRAIN_SIZE: goes through at::parallel_for
        _HostPoolGuard._poke.add_(1.0)
        return self

    def __exit__(self, *exc):
        if self.saved is not None:
            torch.set_num_threads(self.saved)
        return False


class _RenderRays(torch.autograd.Function):
    """render_rays as one differentiable node: forward = dsn_render_rays_train, backward = dsn_render_rays_grad
    (what loss.backward() does in trainer.py:70-81).  Inputs that are not parameters carry no gradient, as in the
    reference (rays, near/far, xyz, poses are data).

    The forward leaves its activations in the renderer's ONE shared GradWorkspace; `_cache_gen` says whose they are.  A
    backward whose generation is no longer current recomputes them - into the same workspace, so it takes a new generation
    itself: with several renders outstanding every backward but (at most) the latest recomputes, and none reads another
    call's activations.  The parameters go through save_for_backward, so an in-place update between forward and backward
    (optimizer.step()) raises like it does for an op-by-op graph."""

    @staticmethod
    def forward(ctx, renderer, call, *params):
        o, d, near, far, S, jitter, noise, uniform, frame_args = call
        if not hasattr(renderer, "_grad_ws"):
            renderer._grad_ws = _lib.GradWorkspace(renderer.device)
        renderer._cache_gen = getattr(renderer, "_cache_gen", 0) + 1      # whose activations the cache holds
        out = _lib.render_rays(renderer.scene, renderer.net.packed(renderer.device), renderer._ws, o, d, near, far, S,
                               renderer._t_vals(S), jitter, noise, skip_transparent=False, uniform=uniform,
                               train_cache=renderer._grad_ws)
        ctx.save_for_backward(*params)
        ctx.renderer, ctx.call, ctx.gen = renderer, (o, d, noise, frame_args), renderer._cache_gen
        ctx.frame_key = renderer.scene.frame_key
        ctx.z_vals = out["z_vals"]
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(out["z_vals"])
        return tuple(out[k] for k in _OUT_KEYS)

    @staticmethod
    def backward(ctx, g_color, g_disp, g_acc, g_depth, g_weights, g_z):
        r = ctx.renderer
        o, d, noise, frame_args = ctx.call
        xyz, poses, frame, zero_code, ls, rot, rc = frame_args
        saved = ctx.saved_tensors                           # raises if a parameter was modified in place since the forward
        params = [p.detach() for p in saved]
        sd = dict(zip(_lib.PARAM_ORDER, params))
        packed = r.net.packed(r.device)
        cached = ctx.gen == getattr(r, "_cache_gen", -1)
        # another frame was set since this call's forward - or the activations have to be recomputed and the frame was set lazily
        # (its lists belong to the forward's render call: the recomputation's warp needs every cell's)
        if r.scene.frame_key != ctx.frame_key or (not cached and r.scene.lazy):
            r.scene.set_frame(packed, xyz, poses, frame, zero_code, ls, rot, rc)
            r._frame_src = None
        if g_color is None:
            g_color = torch.zeros(o.shape[0], 3, device=r.device)
        if not hasattr(r, "_grad_ws"):
            r._grad_ws = _lib.GradWorkspace(r.device)
        if not cached:
            r._cache_gen = getattr(r, "_cache_gen", 0) + 1  # the recomputation below overwrites the shared workspace
        grads = _lib.render_rays_grad(r.scene, sd, poses, frame, zero_code, o, d, ctx.z_vals, noise, g_color, g_disp, g_acc,
                                      g_depth, g_weights, ws=r._grad_ws, packed=packed, cached=cached)
        r._note_training_range(_lib.grad_range_word(r._grad_ws, *ctx.z_vals.shape))
        grads = [g.to(device=p.device, dtype=p.dtype).reshape(p.shape) for g, p in zip(grads, saved)]
        return (None, None) + tuple(grads)


class _ViewSlot:
    """scene + workspace + stream of one frame in flight (render_views)"""

    def __init__(self, renderer, own_scene):
        self.scene = renderer.scene if not own_scene else _lib.Scene(renderer.canonical_model["vertex"], renderer.face_idx,
                                                                    renderer.device)
        # (a new slot's workspace starts with the record capacity the renderer's frames were seen to need)
        self.ws = renderer._ws if not own_scene else _lib.RenderWorkspace(renderer.device, fraction=renderer._ws.want_fraction)
        self.stream = torch.cuda.Stream(device=renderer.device)


class _Renderers:
    alive = 0          # Renderer objects of this process (the host-pool fit is undone when the last one is gone)
    lowered = None     # (the caller's torch.get_num_threads(), what fit_host_pool lowered it to)


class Renderer:
    def __init__(self, net, fine_net=None, cfg=None, canonical_vertex=None, body_data=None, device=None, host_pool="fit"):
        """`body_data` (optional, not in the reference): dict with 'f' [F,3] (and optionally 'weights',
        'kintree_table') used instead of unpickling cfg.DATASETS.SMPL_PATH - the SMPL file is licensed.
        `host_pool` (not in the reference): "fit" (default) caps torch's intra-op thread pool - a PROCESS-GLOBAL setting - at the
        cgroup's CPU quota, once, with a warning when that lowers the caller's setting (why: _lib.fit_host_pool; the GPU boxes show
        256 hardware threads under a 16-core quota and a pool sized from the former gets the process frozen by the kernel's
        bandwidth control); "keep" leaves the caller's pool alone (also: environment DSN_HOST_POOL=keep), and then only the
        per-frame guard (host_pool_limit, restored after every frame) applies."""
        _lib.require_gpu()
        self.net = net
        self.cfg = cfg
        self.fine_net = fine_net
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.canonical_vertex = canonical_vertex
        self._body_data = body_data
        self.load_body_model(gender="neutral", body_model="smpl", model_path=getattr(cfg.DATASETS, "SMPL_PATH", ""))
        self.sample_points_mode = cfg.MODEL.sample_points_mode
        self._ws = _lib.RenderWorkspace(self.device)
        self._tvals = {}
        self._frame_src = None
        self._slots = []
        self.skip_transparent = True      # eval mode: networks only on non-transparent samples (exact)
        # eval frames: the posed mesh's nearest-face lists are built by the frame's own render call, for the cells its samples visit
        # (DSN_FRAME_LAZY_LISTS: same lists entry for entry, for 48 % of the cells on a whole 512 x 512 frame, a tenth on a rank's
        # block of a partitioned frame).  False: every cell's lists in dsn_set_frame, as rounds 1-4 did.
        self.lazy_lists = True
        self.train_lazy_lists = True
        # eval mode: plain-fp16 density screen in front of the accurate pass.  OPT-IN since round 4: its margin is
        # calibrated for the loaded parameters and audited while it runs - frames are bit-identical with it on or off on everything
        # tested - but that is statistical safety, not a proof, and the one converged checkpoint (w4) calibrates it off anyway.
        # density_screen = True: it runs when its calibration says it is safe and pays (-29 % frame time on an untrained fog).
        self.density_screen = False
        # eval mode: re-check 1/128 of the screened-out samples.  True: every frame (read with last_screen_audit()); "auto" (default):
        # the first frame after a calibration and every SCREEN_AUDIT_EVERY-th one after it, read back WITHOUT a wait at the start
        # of a later frame - a violation (a dropped sample whose accurate density is positive) switches the screen off with a
        # warning.  The margin is calibrated on one pose: this is what watches every other pose in production.
        self.screen_audit = "auto"
        self._audit_probe = None          # (count words, event) of an audited frame still to be looked at
        self._audit_frames = 0
        self.screen_info = None           # what the last calibration of the screen found (PackedParams.calibrate_screen)
        # eval mode: front-to-back slices with ray termination (DSN_EARLY_STOP; pixel error <= (S + 1) eps(S, c) x colour = 5e-5 absolute
        # for colours up to the scale c).  "auto": the first eval frame of a parameter version also counts what termination would
        # leave out (DSN_STOP_STATS) and how large its colours are; from the next frame on it is used if it leaves out at least
        # _lib.EARLY_STOP_MIN_SKIPPED of the non-transparent samples, with c = 2 x the largest colour seen (every
        # SCREEN_AUDIT_EVERY-th sliced frame is looked at again, without a wait: larger colours raise c, with a warning).
        # True: the same probe frame, then sliced whatever it leaves out; False switches it off.
        self.early_stop = "auto"
        self._stop_probe = None           # (packed generation, count words, event) of a frame whose statistics are still to be read
        self._guards = []                 # counter copies of the sliced frames enqueued since the last hand-over (_stop_guard)
        self._guard_pool = []             # page-locked 256-byte buffers for them
        self._one_pass_only = False       # set while a frame that broke its early-stop bound is rendered again
        # "auto": the slice lengths of sliced frames follow the probe frame's statistics (longer slices where few rays end: fewer
        # launches, a few more samples, the same error bound - dsn_render_rays_ex); None: uniform slices of 4 / 8 samples
        self.stop_schedule = "auto"
        self._stop_frames = 0
        # render_view / render_views: torch's intra-op pool is capped at this many threads while a frame is staged, enqueued and
        # awaited (_HostPoolGuard; None = leave the pool alone); and for good at the cgroup's CPU quota (_lib.fit_host_pool: a pool
        # larger than the quota gets the whole process frozen by the kernel's bandwidth control whenever the caller runs a torch op)
        self.host_pool_limit = 8
        if host_pool not in ("fit", "keep"):
            raise ValueError('host_pool must be "fit" or "keep"')
        self.host_pool = _lib.fit_host_pool(keep=(host_pool == "keep"))
        _Renderers.alive += 1
        if self.host_pool[0] != self.host_pool[1] and _Renderers.lowered is None:
            _Renderers.lowered = (self.host_pool[0], self.host_pool[1])      # (the caller's setting, what it was lowered to)

    def __del__(self):
        # the pool setting is the process's, not this object's: when the LAST Renderer goes away and nobody has
        # changed the setting since it was lowered, the caller gets back the value it had
        try:
            _Renderers.alive -= 1
            if _Renderers.alive == 0 and _Renderers.lowered is not None:
                before, now = _Renderers.lowered
                _Renderers.lowered = None
                if torch.get_num_threads() == now:
                    torch.set_num_threads(before)
        except Exception:
            pass

    # ---- mode switches (reference :26-38) ----
    def train(self):
        self.net.training = True
        self.net.train()
        if self.fine_net is not None:
            self.fine_net.training = True
            self.fine_net.train()

    def eval(self):
        self.net.training = False
        self.net.eval()
        if self.fine_net is not None:
            self.fine_net.training = False
            self.fine_net.eval()

    # ---- body model (reference :382-406) ----
    def load_body_model(self, gender, body_model, model_path):
        tmp = self._body_data if self._body_data is not None else load_bodydata(body_model, gender, model_path)
        if "kintree_table" in tmp:
            parents = torch.as_tensor(np.asarray(tmp["kintree_table"])[0].astype(np.int64)).long()
            parents[0] = -1
            self.parents = parents
        if "weights" in tmp:
            self.smpl_blend_weight = torch.as_tensor(np.asarray(tmp["weights"], np.float32))[None].to(self.device)
        self.face_idx = torch.as_tensor(np.asarray(tmp["f"]).astype(np.int64)).long().to(self.device)
        if self.canonical_vertex is not None:
            cv = torch.as_tensor(self.canonical_vertex, dtype=torch.float32).reshape(-1, 3).to(self.device)
            self.canonical_model = {"vertex": cv, "meshes": cv[self.face_idx]}
            self.scene = _lib.Scene(cv, self.face_idx, self.device)

    # ---- helpers ----
    def _t_vals(self, S):
        # torch.linspace on the host, exactly as utils/pts_utils.py:4 does, then uploaded once
        if S not in self._tvals:
            self._tvals[S] = torch.linspace(0.0, 1.0, steps=S).to(self.device)
        return self._tvals[S]

    def _dev(self, t, dtype=torch.float32):
        """Tensor on the device.  Host tensors of a batch (the reference's DataLoader hands over pageable CPU tensors and calls
        .cuda() on them, can_render.py:100-103,138-141) go through a small ring of persistent page-locked staging buffers and an
        asynchronous copy: a pageable .to(device) of a few MB costs 10-20 ms on the GPU boxes (the runtime pins and unpins the
        pages and waits for the stream), 70 ms per 512 x 512 batch (scripts/render_view_probe.py)."""
        if t.is_cuda or t.numel() * t.element_size() < (1 << 16) or t.is_pinned() or self.device.type != "cuda":
            return t.to(device=self.device, dtype=dtype).contiguous()
        n = t.numel()
        ring = getattr(self, "_stage_ring", None)
        if ring is None:
            ring = self._stage_ring = {"i": 0, "slots": [{"host": None, "done": torch.cuda.Event()} for _ in range(8)]}
        slot = ring["slots"][ring["i"]]
        ring["i"] = (ring["i"] + 1) % len(ring["slots"])
        slot["done"].synchronize()                       # the copy that last read this staging buffer has finished
        nbytes = n * dtype.itemsize
        if slot["host"] is None or slot["host"].numel() < nbytes:
            slot["host"] = torch.empty(max(nbytes, 1 << 22), dtype=torch.uint8).pin_memory()
        stage = slot["host"][:nbytes].view(dtype).view(t.shape)
        # host memcpy (with the dtype conversion, if any) through numpy: ONE thread.  torch's copy_ fans a few MB out over its
        # whole intra-op pool - 128-256 threads on the GPU boxes - whose threads then spin beside the thread that feeds the GPU:
        # a 1 MB torch CPU op in front of a frame costs the frame 7-14 ms (scripts/h2h_caller_probe.py)
        if t.device.type == "cpu" and not t.requires_grad and t.dtype != torch.bfloat16:
            np.copyto(stage.numpy(), t.numpy(), casting="unsafe")
        else:
            stage.copy_(t)
        out = torch.empty(t.shape, dtype=dtype, device=self.device)
        out.copy_(stage, non_blocking=True)
        slot["done"].record()
        return out

    # Which tensor the scene's posed mesh came from.  Identity, not address: a DataLoader hands a NEW tensor per batch that
    # routinely reuses the freed address of the previous one, and an in-place update keeps the address - so the
    # key is a weak reference to the tensor object plus its version counter.
    def _mark_frame_src(self, xyz):
        self._frame_src = (weakref.ref(xyz), xyz._version)

    def _is_frame_src(self, xyz):
        src = self._frame_src
        return src is not None and src[0]() is xyz and src[1] == xyz._version

    def _set_frame(self, batch, scene=None, frame=None, lazy=False):
        """dsn_set_frame from a batch.  Returns the arguments as used (device tensors), which a training forward keeps for
        its backward.  lazy: the frame is about to be rendered by _lib.render_rays (eval frames): only the posed mesh's fine grid is
        laid out here and the render call builds the candidate lists of the cells its samples visit (DSN_FRAME_LAZY_LISTS)."""
        scene = self.scene if scene is None else scene
        if frame is None:
            frame = int(torch.as_tensor(batch["frame"]).reshape(-1)[0])
        zero_code, ls, rot, rc = self.net.frame_args(batch)
        xyz = self._dev(batch["xyz"][0])
        poses = batch["poses"][0].to(device=self.device, dtype=torch.float32).contiguous()
        # (fine level only: the samples of rays clipped to the body's bounds never leave it; stage calls on far-away points - w2l
        #  on arbitrary points - still get the exact index, from the exhaustive sweep)
        scene.set_frame(self.net.packed(self.device), xyz, poses, frame, zero_code, ls, rot, rc, fine_only=True,
                        lazy=bool(lazy) and self.lazy_lists)
        if scene is self.scene:
            self._mark_frame_src(batch["xyz"])
        return (xyz, poses, frame, zero_code, ls, rot, rc)

    def _screen_usable(self, frame=None):
        """eval mode: is the density screen on for this frame?  Calibrates its margin for the current parameters first
        (once per parameter version: synchronises then).  Needs the scene's frame state to be set.  frame = (scene, workspace, R, S)
        of a frame whose geometry phase has run: the calibration then uses that frame's own canonical points (what _render_eval
        does on the first eval frame of a parameter version); without it, a cube around the canonical centroids."""
        if not (self.density_screen and self.skip_transparent):
            return False
        packed = self.net.packed(self.device)
        if packed.screen is None:
            info = packed.calibrate_screen(self.scene) if frame is None else packed.calibrate_screen(frame[0], frame=frame[1:])
            self.screen_info = dict(info)
            if not info["safe"]:
                warnings.warn("dsnerf_amd: the plain-fp16 density screen would need a margin of %.3g of the term magnitude for these "
                              "parameters (cap %.3g; largest deviation %.3g): it stays off, every non-transparent sample takes the "
                              "accurate pass" % (_lib.SCREEN_HEADROOM * info["margin_statistic"], _lib.SCREEN_MARGIN_CAP,
                                                 info["deviation"]))
        return packed.screen_pays(self._early_stop_in_use(packed))

    def _early_stop_in_use(self, packed):
        if self.early_stop == "auto":
            return bool(packed.early_stop and packed.early_stop["usable"])
        return bool(self.early_stop)

    def _draws(self, R, S):
        """Train-mode random draws from the CPU default generator in the reference's order:
        torch.rand([1,R,S]) (utils/pts_utils.py:12) then torch.randn([R,S]) (utils/nerf_net_utils.py:31)."""
        want_j = self.net.training and self.cfg.MODEL.perturb > 0.0
        want_n = self.net.training and self.cfg.MODEL.raw_noise_std > 0.0
        if not (want_j or want_n):
            return None, None
        if self.device.type != "cuda":
            raise RuntimeError("dsnerf_amd renders on the GPU only")
        # The draws go straight into a small ring of persistent page-locked staging buffers (out=: same generator
        # stream as a fresh tensor) and are uploaded with one asynchronous copy.  Fresh 2 MB host tensors cost 10-80 ms
        # per step on the GPU boxes (page faults of newly mapped memory + the runtime pinning them for the pageable
        # copy, which also blocks until the stream has drained): scripts/train_host_probe.py.
        n = 2 * R * S
        ring = getattr(self, "_draw_ring", None)
        if ring is None or ring["n"] < n:
            ring = self._draw_ring = {"n": n, "i": 0, "slots": [
                {"host": torch.empty(n, dtype=torch.float32).pin_memory(), "dev": torch.empty(n, device=self.device),
                 "done": torch.cuda.Event()} for _ in range(3)]}
        slot = ring["slots"][ring["i"]]
        ring["i"] = (ring["i"] + 1) % len(ring["slots"])
        slot["done"].synchronize()            # the copy that last read this staging buffer (three draws ago) has finished
        k = 0
        if want_j:
            torch.rand(1, R, S, out=slot["host"][:R * S].view(1, R, S))
            k = R * S
        m = k
        if want_n:
            hn = slot["host"][k:k + R * S].view(R, S)
            torch.randn(R, S, out=hn)
            if float(self.cfg.MODEL.raw_noise_std) != 1.0:
                hn.mul_(self.cfg.MODEL.raw_noise_std)
            m = k + R * S
        slot["dev"][:m].copy_(slot["host"][:m], non_blocking=True)
        slot["done"].record()
        # device-side clones (two 5 us copies): autograd may keep the draws alive for longer than the ring does
        jitter = slot["dev"][:R * S].view(R, S).clone() if want_j else None
        noise = slot["dev"][k:k + R * S].view(R, S).clone() if want_n else None
        return jitter, noise

    # ---- sampling (reference :40-63) ----
    def get_sampling_points(self, ray_o, ray_d, near, far, xyz, mode="GG"):
        """ray_o/ray_d [1,R,3], near/far [1,R] (updated in place in GG mode), xyz [1,V,3] ->
        pts [1,R,S,3], z_vals [1,R,S].  Needs the frame's xyz in the scene: Renderer.render does that;
        direct callers get it set here from `xyz` with the current pose state untouched."""
        S = self.cfg.MODEL.COARSE_RAY_SAMPLING
        R = ray_o.shape[1]
        o, d = self._dev(ray_o[0]), self._dev(ray_d[0])
        n_dev, f_dev = self._dev(near[0]), self._dev(far[0])
        jitter = None
        if self.net.training and self.cfg.MODEL.perturb > 0.0:
            jitter = torch.rand(1, R, S).reshape(R, S).to(self.device)
        if not self._is_frame_src(xyz):
            self._upload_xyz(xyz)
        pts, z = _lib.sample(self.scene, o, d, n_dev, f_dev, S, self._t_vals(S), jitter, want_pts=True, gg=(mode == "GG"))
        if mode == "GG":   # in-place update like the reference (:52-53)
            near[0].copy_(n_dev.to(near.device))
            far[0].copy_(f_dev.to(far.device))
        return pts[None], z[None]

    def _upload_xyz(self, xyz):
        # posed mesh only (callers without a full batch: the pose code is zeroed until _set_frame runs)
        x = self._dev(xyz.reshape(-1, 3))
        poses = torch.zeros(24, 3, device=self.device)
        self.scene.set_frame(self.net.packed(self.device), x, poses, 0, True, None, None, None)
        self._mark_frame_src(xyz)

    def _ensure_mesh(self, batch):
        """the scene must hold batch['xyz'] as its posed mesh (direct callers of the warp / network stages)"""
        xyz = batch["xyz"]
        if self._is_frame_src(xyz) and not self.scene.lazy:      # (a lazily set frame holds lists for its own render call only)
            return
        if "poses" in batch and "frame" in batch:
            self._set_frame(batch)
        else:
            self._upload_xyz(xyz)

    # ---- warp (reference :299-379) ----
    def w2l(self, pts_world, ray_o_W, ray_d_W, batch):
        B, ray, sp, _ = pts_world.shape
        pts_smpl_can, transparent_mask, ray_d_can = self.w2l_without_lbs(
            pts_world, batch, self.canonical_model, ray_d_W=ray_d_W.unsqueeze(2).expand([-1, -1, sp, -1]).reshape(B, -1, 3))
        d = self._dev(ray_d_W.reshape(-1, 3))
        rays = torch.cat([d[:, None, :].expand(-1, sp, -1).reshape(-1, 3), ray_d_can], dim=-1).reshape(B * ray, sp, 6)
        pw = self._dev(pts_world.reshape(B * ray, sp, 3))
        return torch.cat([pw, pts_smpl_can.reshape(-1, sp, 3)], dim=-1), rays, transparent_mask

    def w2l_without_lbs(self, pts_world, batch, canonical_model, ray_d_W=None, floor=-4, ceil=5):
        """pts_world [B,R,S,3] -> (pts_smpl_can [N,3], transparent_mask [B,N] bool[, ray_d_can [N,3]]).
        ray_d_W, when given, is the per-POINT direction tensor [B,N,3] the reference passes."""
        assert floor == -4 and ceil == 5, "the uv clamp range is compiled in (utils/render_utils.py:103)"
        B, ray, sp, _ = pts_world.shape
        self._ensure_mesh(batch)
        pts = self._dev(pts_world.reshape(-1, 3))
        if ray_d_W is not None:
            d = self._dev(ray_d_W.reshape(-1, 3))   # per point -> S=1 addressing
            out = _lib.warp(self.scene, pts, d, 1, want_dir=True)
            return out["x_c"], out["transparent"].bool().reshape(B, -1), out["ray_d_can"]
        out = _lib.warp(self.scene, pts, None, 1, want_dir=False)
        return out["x_c"], out["transparent"].bool().reshape(B, -1)

    # ---- network + compositing on explicit points (reference :65-134) ----
    def _module_info(self, batch_info, frame_idx):
        """batch_info for a call into self.net with this renderer's scene holding the frame (so the module does not build a
        second scene): reference batch_info needs poses, xyz, Th, canonical_model, face_idx (model/spacenet.py:210-266)."""
        bi = dict(batch_info)
        if "xyz" in bi and "poses" in bi:
            if not self._is_frame_src(bi["xyz"]) or self.scene.frame_key is None:
                fi = int(torch.as_tensor(frame_idx).reshape(-1)[0])
                self._set_frame(bi, frame=fi)
            bi["_dsn_scene"] = self.scene
        return bi

    def render_rays(self, pts, rays, z_vals, frame_idx, net, transparent_mask=None, batch_info=None):
        """can_render.py:97-134.  Differentiable w.r.t. the parameters in train mode (DualSpaceNeRF.forward's autograd node
        + torch ops on its outputs would be the reference's graph; here compositing is dsn_composite, which carries no
        graph - train through Renderer.render, whose single node covers the whole path)."""
        if self.net.training and torch.is_grad_enabled() and any(p.requires_grad for p in self.net.parameters()):
            raise RuntimeError("Renderer.render_rays / batchify_pts build no autograd graph in dsnerf_amd (compositing is one HIP "
                               "kernel): train through Renderer.render, or call them under torch.no_grad() / in eval mode")
        pts, rays, z_vals = self._dev(pts), self._dev(rays), self._dev(z_vals)
        rays_d = rays[:, 0, :3].contiguous()
        B, sp = pts.shape[:2]
        noise = None
        if self.net.training and self.cfg.MODEL.raw_noise_std > 0.0:
            noise = (torch.randn(B, sp) * self.cfg.MODEL.raw_noise_std).to(self.device)
        bi = self._module_info(batch_info or {}, frame_idx)
        with torch.no_grad():
            rgbs, density, _ = net(pts.reshape(-1, 6), rays.reshape(-1, 6), frame_idx, batch_info=bi)
        tm = None if transparent_mask is None else transparent_mask.to(self.device).reshape(B, sp).to(torch.uint8).contiguous()
        rgb_map, disp_map, acc_map, weights, depth_map = _lib.composite(
            rgbs.reshape(B, sp, 3).contiguous(), density.reshape(B, sp).contiguous(), tm, z_vals, rays_d, noise)
        return {"color": rgb_map, "disp_map": disp_map, "acc_map": acc_map, "depth_map": depth_map,
                "weights": weights, "z_vals": z_vals}

    def batchify_pts(self, pts, rays, z_vals, frame_idx, chunk=1024 * 32, net=None, batch_info=None):
        net = self.net if net is None else net
        return self.render_rays(pts, rays, z_vals, frame_idx, net=net,
                                transparent_mask=batch_info["transparent_mask"], batch_info=batch_info)

    # ---- the trainer's call (reference :137-168) ----
    def render(self, batch):
        o, d = self._dev(batch["ray_o"][0]), self._dev(batch["ray_d"][0])
        near, far = self._dev(batch["near"][0]), self._dev(batch["far"][0])
        R = o.shape[0]
        S = self.cfg.MODEL.COARSE_RAY_SAMPLING
        self._poll_training_range()
        sd = dict(self.net.named_parameters())
        differentiable = self.net.training and torch.is_grad_enabled() and any(p.requires_grad for p in sd.values())
        # (lazily set for training too since round 6: the training forward builds the lists of the cells its batch visits, or - small
        #  batches - completes every cell's lists itself; `train_lazy_lists = False` keeps the per-step build of every cell in set_frame)
        frame_args = self._set_frame(batch, lazy=(not differentiable) or self.train_lazy_lists)
        jitter, noise = self._draws(R, S)
        if self.sample_points_mode not in ("GG", "uniform"):
            raise Exception("error")   # the reference fails on unknown modes too (get_sampling_points returns nothing)
        uniform = self.sample_points_mode == "uniform"
        if differentiable:
            outs = _RenderRays.apply(self, (o, d, near, far, S, jitter, noise, uniform, frame_args),
                                     *[sd[k] for k in _lib.PARAM_ORDER])
            out = dict(zip(_OUT_KEYS, outs))
        else:
            out = self._render_eval(self.scene, self._ws, o, d, near, far, S, jitter, noise)
            out = self._hand_over(out, lambda: self._render_eval(self.scene, self._ws, o, d, near, far, S, jitter, noise))
        if batch["near"].is_cuda:   # in-place semantics of the reference when the batch already lives on the device
            batch["near"][0].copy_(near)
            batch["far"][0].copy_(far)
        batch["canonical_model"] = self.canonical_model
        batch["face_idx"] = self.face_idx
        return {"coarse": out}

    def _eval_plan(self, noise, screen=None):
        """per-frame decisions of an eval-mode frame (needs the scene's frame state for a first calibration): keyword arguments
        of _lib.render_rays"""
        skip = self.skip_transparent and not self.net.training
        if skip and noise is None and self.early_stop in ("auto", True):
            self._read_stop_probe()      # (first: whether the screen pays depends on it)
        if screen is None:
            screen = skip and noise is None and self._screen_usable()
        packed = self.net.packed(self.device)
        stop, stats = False, False
        if skip and noise is None and not self._one_pass_only:
            if self.early_stop in ("auto", True):
                # the first eval frame of a parameter version is rendered in one pass and measures: what termination would leave out,
                # how large the colours are (-> the threshold's colour scale), the slice schedule.  "auto" then slices when it pays;
                # True slices whenever the colours are finite (round 5: it used to slice from the first frame on with the scale 1, which
                # the hand-over check now refuses for any checkpoint whose colours exceed 1)
                if packed.early_stop is None:
                    if self._stop_probe is None:
                        stats = True
                else:
                    stop = packed.early_stop["usable"] if self.early_stop == "auto" else packed.early_stop.get("finite", True)
            else:
                stop = bool(self.early_stop)
        audit = False
        if screen:
            self._read_audit_probe()
            screen = screen and self.density_screen        # (a violation found just now has switched it off)
            if self.screen_audit == "auto":
                audit = self._audit_probe is None and self._audit_frames % SCREEN_AUDIT_EVERY == 0
                self._audit_frames += 1
            else:
                audit = bool(self.screen_audit)
        sched = packed.early_stop.get("schedule") if (stop and packed.early_stop and self.stop_schedule == "auto") else None
        if sched is not None and sum(sched) != int(self.cfg.MODEL.COARSE_RAY_SAMPLING):
            sched = None
        return {"skip_transparent": skip, "uniform": (self.sample_points_mode == "uniform"), "screen": screen,
                "audit": audit, "early_stop": stop, "stop_stats": stats, "stop_schedule": sched}

    def _render_eval(self, scene, ws, o, d, near, far, S, jitter, noise, screen=None, plan=None, phases=0, out=None):
        packed = self.net.packed(self.device)
        if (plan is None and screen is None and phases == 0 and noise is None and packed.screen is None and self.density_screen
                and self.skip_transparent and not self.net.training):
            # First eval frame of a parameter version: its geometry phase first (sampler, warp: the canonical points of its
            # non-transparent samples), the density screen's margin calibrated on THOSE points, then the rest of the frame.
            geo = {"skip_transparent": True, "uniform": (self.sample_points_mode == "uniform"), "screen": True, "audit": False,
                   "early_stop": False, "stop_stats": False, "stop_schedule": None}
            out = _lib.render_rays(scene, packed, ws, o, d, near, far, S, self._t_vals(S), jitter, noise, phases=_lib.PHASE_GEOMETRY,
                                   out=out, **geo)
            self._screen_usable(frame=(scene, ws, o.shape[0], S))
            phases = _lib.PHASE_FIELD | _lib.PHASE_SHADE
        plan = self._eval_plan(noise, screen) if plan is None else plan
        # what this frame ran with (early stop is the one default-on feature whose output is error-bounded, not
        # bit-identical - a caller can see per frame whether it was in use and with which threshold)
        cs = packed.colour_scale
        self.last_frame_info = {"density_screen": bool(plan["screen"]), "screen_audit": bool(plan["audit"]),
                                "early_stop": bool(plan["early_stop"]), "early_stop_schedule": plan.get("stop_schedule"),
                                "early_stop_eps": _lib.early_stop_eps(S, cs) if plan["early_stop"] else None,
                                "early_stop_colour_scale": cs if plan["early_stop"] else None,
                                "early_stop_bound_x_max_colour": (S + 1) * _lib.early_stop_eps(S, cs) if plan["early_stop"] else 0.0,
                                # (absolute, for colours up to the scale)
                                "early_stop_bound_abs": (S + 1) * _lib.early_stop_eps(S, cs) * cs if plan["early_stop"] else 0.0}
        out = _lib.render_rays(scene, packed, ws, o, d, near, far, S, self._t_vals(S), jitter, noise, phases=phases, out=out,
                               share_cus=getattr(self, "_frames_overlap", False), **plan)
        if plan["stop_stats"] and (phases == 0 or phases & _lib.PHASE_SHADE):
            self._probe_samples = int(o.shape[0]) * int(S)
            self._probe_shape = (int(o.shape[0]), int(S))
            snap = ws.buf[:_lib.CNT_BYTES].clone()      # (stream-ordered: the next frame on this workspace clears the words)
            ev = torch.cuda.Event()
            ev.record()
            self._stop_probe = (packed.generation, snap, ev)
        if plan["early_stop"] and (phases == 0 or phases & _lib.PHASE_SHADE):
            # EVERY sliced frame leaves its counters for the caller's hand-over point (_stop_guard_ok): the bound of early stop holds for
            # colours up to the scale its threshold was computed with - a frame that weighed a larger colour is rendered again in one
            # pass BEFORE it is handed to the caller (round 4 looked at every 8th frame and only raised the scale for
            # later ones)
            self._guards.append(self._stop_guard(ws, cs, int(o.shape[0]) * int(S)))
            while len(self._guards) > 64:      # (a caller of batchify_rays_view / render_rays that never reaches a hand-over point)
                self._guard_pool.append(self._guards.pop(0)[0])
            self._stop_frames += 1
        if plan["audit"] and self.screen_audit == "auto" and (phases == 0 or phases & _lib.PHASE_SHADE):
            snap = ws.buf[:256].clone()
            ev = torch.cuda.Event()
            ev.record()
            self._audit_probe = (snap, ev)
        return out

    def _read_audit_probe(self, wait=False):
        """screen_audit = "auto": look at the counters of an audited frame once it has finished (no wait unless asked)"""
        if self._audit_probe is None:
            return None
        snap, ev = self._audit_probe
        if not wait and not ev.query():
            return None
        ev.synchronize()
        self._audit_probe = None
        return self._judge_audit(snap.view(torch.int32).cpu())

    def _judge_audit(self, c):
        # (word + 6: the audited samples the judging kernel went through - the raw counter at + 0 runs on past the list's capacity)
        res = {"audited": int(c[_lib.CNT_AUDIT + 6]), "violations": int(c[_lib.CNT_AUDIT + 4]),
               "max_sigma": float(c[_lib.CNT_AUDIT + 5:_lib.CNT_AUDIT + 6].view(torch.float32)[0])}
        self.last_audit = res
        if res["violations"] > 0 and self.density_screen:
            self.density_screen = False
            warnings.warn("dsnerf_amd: the density screen dropped samples with positive density (%d of %d audited, max sigma %.3g): "
                          "screen switched off" % (res["violations"], res["audited"], res["max_sigma"]))
        return res

    def _read_stop_probe(self, wait=False):
        """early_stop = "auto": pick up the statistics of the probe frame once it has finished (no wait unless asked)."""
        if self._stop_probe is None:
            return
        gen, snap, ev = self._stop_probe
        if not wait and not ev.query():
            return
        ev.synchronize()
        self._stop_probe = None
        packed = self.net.packed(self.device)
        if packed.generation != gen:
            return
        st = _lib.read_stop_stats(snap)
        frac = st["would_skip"] / max(st["active"], 1)
        # (the probe frame also says how many relu records frames of these parameters need: a dense field gets a larger record array
        #  before its second frame instead of the overflow pass on every frame)
        n_pos = int(snap.view(torch.int32)[_lib.CNT_POS])
        if getattr(self, "_probe_samples", 0) > 0:
            # (the probe frame is one pass; with termination in use the sliced frames list far fewer samples: estimated by what it leaves
            #  out, corrected by the sliced frames themselves at their hand-over (_stop_guards_ok), covered by the exact overflow pass in between)
            will_stop = frac >= _lib.EARLY_STOP_MIN_SKIPPED if self.early_stop == "auto" else bool(self.early_stop)
            self._fit_records(n_pos / float(self._probe_samples) * ((1.0 - frac) if will_stop else 1.0), 1.6 if will_stop else 1.25)
        self._note_colour_max(packed, st["colour_max"], first=True)
        cm = st["colour_max"]
        finite = cm == cm and cm != float("inf")
        packed.early_stop = {"skipped_fraction": frac, "usable": finite and frac >= _lib.EARLY_STOP_MIN_SKIPPED, "finite": finite,
                             "colour_max": st["colour_max"], "colour_scale": packed.colour_scale}
        # the slice schedule of the frames to come (longer slices where few rays end: dsn_render_rays_ex), from the probe frame's histogram
        R_, S_ = getattr(self, "_probe_shape", (0, 0))
        if self.stop_schedule == "auto" and R_ > 0 and snap.numel() >= _lib.CNT_BYTES:
            hist, L = _lib.read_stop_hist(snap, R_, S_)
            lens, ev, un = _lib.choose_stop_schedule(hist, L, S_)
            if len(lens) < hist.shape[1]:
                packed.early_stop.update({"schedule": lens, "schedule_evaluates": ev, "uniform_evaluates": un})

    def _fit_records(self, positive_fraction, headroom=1.25):
        """the relu-record capacity of THIS renderer's workspaces (its own and its view slots') follows what its frames put on the
        sigma > 0 list; each workspace applies the request at the start of its next frame (RenderWorkspace.begin_frame) - never
        between the phases of a frame, and no other renderer's workspace is touched"""
        for ws in [self._ws] + [sl.ws for sl in self._slots if sl.ws is not self._ws]:
            ws.fit_records(positive_fraction, headroom)

    def _note_colour_max(self, packed, cmax, first=False):
        """the early-stop threshold's colour scale follows the largest colour seen: scale = 2 x that (never below 1, never lowered)"""
        if not (cmax == cmax) or cmax == float("inf"):
            # a NaN / inf colour reached a pixel: nothing sensible to scale with - termination off for these parameters
            if packed.early_stop is not None:
                packed.early_stop["usable"] = packed.early_stop["finite"] = False
            warnings.warn("dsnerf_amd: a frame weighed a non-finite colour: early stop stays off for these parameters")
            return
        want = _lib.EARLY_STOP_COLOUR_HEADROOM * cmax
        if first:
            if want > packed.colour_scale:
                packed.set_early_stop_colour_scale(want)
        elif cmax > packed.colour_scale:
            warnings.warn("dsnerf_amd: a sliced frame weighed colours up to %.3g, above the scale %.3g of the early-stop threshold (its "
                          "bound would be %.2g instead of 5e-5): the frame is rendered again in one pass, scale raised to %.3g"
                          % (cmax, packed.colour_scale, 5e-5 * cmax / packed.colour_scale, want))
            packed.set_early_stop_colour_scale(want)
            if packed.early_stop is not None:
                packed.early_stop["colour_scale"] = packed.colour_scale

    def _stop_guard(self, ws, scale, n_samples):
        """enqueue the copy of a sliced frame's counter words (largest colour weighed, samples on the sigma > 0 list) to page-locked
        memory behind the frame, on its stream: (host words, event, colour scale the frame's threshold used, samples, generation)"""
        pool = self._guard_pool
        host = pool.pop() if pool else torch.empty(256, dtype=torch.uint8).pin_memory()
        host.copy_(ws.buf[:256], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        return (host, ev, float(scale), int(n_samples), self.net.packed(self.device).generation)

    def _stop_guards_ok(self, guards):
        """the hand-over check of sliced frames (waits for their counter copies - the frames themselves are done or about to be):
        True when every one of them stayed below the colour scale its threshold assumed.  Otherwise the scale is raised (warning) and
        the caller renders the frame again in one pass.  Also feeds the record capacity with what sliced frames really list."""
        ok = True
        packed = self.net.packed(self.device)
        for host, ev, scale, n_samples, gen in guards:
            ev.synchronize()
            c = host.view(torch.int32)
            cmax = float(c[_lib.CNT_COLOUR_MAX:_lib.CNT_COLOUR_MAX + 1].view(torch.float32)[0])
            n_pos = int(c[_lib.CNT_POS])
            self._guard_pool.append(host)
            if gen != packed.generation:
                continue
            if n_samples > 0:
                self._fit_records(n_pos / float(n_samples))
            if not (cmax <= scale):                       # (NaN / inf included)
                ok = False
                self._note_colour_max(packed, cmax)
        return ok

    def _hand_over(self, first, again, guards=None):
        """`first` is a frame that has been enqueued; before it goes to the caller the early-stop guards of its sliced render calls are
        looked at (their counter copies sit right behind the compositor on the frame's stream).  A frame that weighed a colour above
        the scale its threshold assumed is rendered AGAIN, in one pass (`again()`), and that is what the caller gets: the stated bound
        of early stop - 5e-5 absolute - then holds for every frame handed out, not only while the colours stay where the probe frame
        found them."""
        if guards is None:
            guards, self._guards = self._guards, []
        if not guards or self._stop_guards_ok(guards):
            return first
        self._one_pass_only = True
        try:
            res = again()
        finally:
            self._one_pass_only = False
            self._guards = []
        self.last_frame_info = dict(self.last_frame_info, rendered_again_in_one_pass=True)
        return res

    def last_screen_audit(self, ws=None):
        """what the audit of the last audited eval frame found - synchronises.  dict(audited, violations, max_sigma): `violations`
        audited samples (declared empty by the screen) have an accurate density > 0; they were rendered correctly (audited samples
        take the accurate pass), but their un-audited peers were not, so the screen is switched off for this renderer when it
        happens.  With screen_audit = "auto" this is the pending audited frame if there is one, else the last one judged."""
        if self.screen_audit == "auto":
            res = self._read_audit_probe(wait=True)
            return res if res is not None else getattr(self, "last_audit", None)
        ws = ws or self._ws
        if ws.buf is None:
            return None
        return self._judge_audit(ws.buf[:256].view(torch.int32).cpu())

    # Training steps whose activations / tangents / adjoints left the fp16 range of the split-fp16 kernels have no exact twin to fall
    # back to: the forward counts such samples (workspace word 48), the backward drops their second-order / adjoint terms and counts
    # them (dsn_train.hip, w.small[303]).  Both counters are copied to page-locked memory behind every backward and looked at -
    # without a wait - when the next step begins: a non-zero count is reported as a warning (nobody calls
    # range_overflow_count(), which synchronises).
    def _note_training_range(self, grad_word):
        st = getattr(self, "_range_watch", None)
        if st is None:
            st = self._range_watch = {"host": torch.zeros(2, dtype=torch.int32).pin_memory(), "ev": None, "warned": 0, "steps": 0}
        if st["ev"] is not None and not st["ev"].query():
            return                                    # the previous snapshot has not landed yet: skip this one
        self._poll_training_range()
        st["host"][0:1].copy_(self._ws.buf[4 * _lib.CNT_RANGE:4 * _lib.CNT_RANGE + 4].view(torch.int32), non_blocking=True)
        st["host"][1:2].copy_(grad_word, non_blocking=True)
        st["ev"] = torch.cuda.Event()
        st["ev"].record()

    def _poll_training_range(self):
        st = getattr(self, "_range_watch", None)
        if st is None or st["ev"] is None or not st["ev"].query():
            return
        st["ev"] = None
        fwd, bwd = int(st["host"][0]), int(st["host"][1])
        if fwd or bwd:
            st["steps"] += 1
            if st["warned"] < 5:
                st["warned"] += 1
                warnings.warn("dsnerf_amd: a training step had %d samples whose activations and %d whose tangents / adjoints left the "
                              "fp16 range of the split-fp16 kernels: their contributions to that step's gradients are not exact "
                              "(Renderer.range_overflow_count())" % (fwd, bwd))

    def range_overflow_count(self):
        """train mode: samples of the last render() whose activations / adjoints left the fp16 range of the split-fp16 kernels
        (there is no exact twin of the stored activations: a non-zero count means this step's gradients are not to be trusted;
        synchronises)."""
        if self._ws.buf is None:
            return 0
        return int(self._ws.buf[:256].view(torch.int32)[_lib.CNT_RANGE])

    # ---- whole-image path (reference :172-278) ----
    def batchify_rays_view(self, ray_o, ray_d, near, far, batch, chunk=None, scene=None, ws=None):
        scene = self.scene if scene is None else scene
        ws = self._ws if ws is None else ws
        # the frame's state first: its kernels (0.45 ms of nearest-face lists for the posed mesh) need the 82 KB of vertices only and
        # run while this thread stages the 8 MB of rays through page-locked memory (-0.1 ms per render_view, A/B in one call)
        self._set_frame(batch, scene=scene, lazy=True)
        o, d = self._dev(ray_o[0]), self._dev(ray_d[0])
        n, f = self._dev(near[0]).clone(), self._dev(far[0]).clone()
        S = self.cfg.MODEL.COARSE_RAY_SAMPLING
        R = o.shape[0]
        chunk = R if chunk is None else int(chunk)
        screen = None
        pk = self.net.packed(self.device)
        if scene is not self.scene and pk.screen is not None:      # (not calibrated yet: _render_eval does it on this frame's points)
            screen = self.skip_transparent and not self.net.training and self.density_screen and pk.screen_pays(self._early_stop_in_use(pk))
        outs = []
        for i in range(0, R, chunk):
            j = min(R, i + chunk)
            jitter, noise = self._draws(j - i, S)
            outs.append(self._render_eval(scene, ws, o[i:j].contiguous(), d[i:j].contiguous(), n[i:j].contiguous(),
                                          f[i:j].contiguous(), S, jitter, noise, screen=None if noise is not None else screen))
        coarse = outs[0] if len(outs) == 1 else {k: torch.cat([x[k] for x in outs], 0) for k in outs[0]}
        return coarse, {}

    def _view_images(self, batch, chunk, scene, ws):
        coarse, _ = self.batchify_rays_view(batch["ray_o"], batch["ray_d"], batch["near"], batch["far"], batch, chunk, scene, ws)
        _, H, W, _ = batch["img"].shape
        # utils/render_utils.py:466-472 post_process, done on the device (dsn_image_scatter)
        return _lib.image_scatter(coarse, self._dev(batch["mask_at_box"][0], torch.uint8), H, W)

    def render_view(self, batch, chunk=None, device_output=False):
        """device_output=True (not in the reference) keeps the [H,W,*] images on the GPU - for multi-frame sequences
        (novel_pose_vis.py:41-66) and on-device metrics (`image_metrics`)."""
        with _HostPoolGuard(self.host_pool_limit):
            return self._render_view(batch, chunk, device_output)

    def _render_view(self, batch, chunk, device_output):
        img = self._view_images(batch, chunk, self.scene, self._ws)
        again = lambda: self._view_images(batch, chunk, self.scene, self._ws)
        if device_output:
            return self._hand_over(img, again)
        # four contiguous device images -> one persistent page-locked staging set (asynchronous copies, one synchronisation),
        # then fresh host tensors like the reference's `.cpu()` results.  A pageable `.cpu()` per image goes through the runtime's
        # own bounce buffers and varies between 1 and 20 ms per frame on the GPU boxes (scripts/d2h_probe.py);
        # `device_output=True` avoids the host side altogether.
        keys = ("coarse_color", "coarse_disp", "coarse_acc", "coarse_depth")
        stage = getattr(self, "_d2h_stage", None)
        if stage is None or any(stage[k].shape != img[k].shape for k in keys):
            stage = self._d2h_stage = {k: torch.empty(img[k].shape, dtype=torch.float32).pin_memory() for k in keys}
        for k in keys:
            stage[k].copy_(img[k], non_blocking=True)
        torch.cuda.current_stream(self.device).synchronize()
        checked = self._hand_over(img, again)      # (the guards' copies have landed with the images: no extra wait in the normal case)
        if checked is not img:
            for k in keys:
                stage[k].copy_(checked[k], non_blocking=True)
            torch.cuda.current_stream(self.device).synchronize()
        # fresh pageable tensors (clone() of a page-locked tensor allocates page-locked memory again: 2-3 ms per image), copied
        # by one thread (see _dev)
        return {k: torch.from_numpy(stage[k].numpy().copy()) for k in keys}

    def render_views(self, batches, frames_in_flight=3, device_output=True, chunk=None):
        """The per-frame loop of novel_pose_vis.py:41-66 / test.py:55-64 (`for batch in loader: render.render_view(batch)`) as
        ONE call over an iterable of batches, with `frames_in_flight` frames on their own HIP streams (own scene blob and
        workspace each): the per-frame setup, sampling and warp kernels of frame k+1 run beside the matrix-bound field kernels
        of frame k.  Returns the list of render_view results in order - bit-identical to calling render_view per batch.
        device_output=True keeps the images on the GPU (the D2H copies of host outputs are issued on each frame's stream and
        overlap the next frames too)."""
        with _HostPoolGuard(self.host_pool_limit):
            # (with frames overlapping, the persistent field kernels leave an eighth of the compute units to the neighbours' small
            #  kernels: DSN_SHARE_CUS, -1.4 % per frame; a frame alone keeps them all)
            self._frames_overlap = int(frames_in_flight) > 1
            try:
                return self._render_views(batches, frames_in_flight, device_output, chunk)
            finally:
                self._frames_overlap = False

    def _render_views(self, batches, frames_in_flight, device_output, chunk):
        n = max(1, int(frames_in_flight))
        while len(self._slots) < n:
            self._slots.append(_ViewSlot(self, own_scene=len(self._slots) > 0))
        slots = self._slots[:n]
        cur = torch.cuda.current_stream(self.device)
        self.net.packed(self.device)                         # shared, read-only state is materialised on the caller's stream
        self._t_vals(self.cfg.MODEL.COARSE_RAY_SAMPLING)
        # (the density screen is calibrated by the first frame itself, on its own points: _render_eval; slot 0 is the renderer's own
        #  scene and its frame is enqueued first)
        keys = ("coarse_color", "coarse_disp", "coarse_acc", "coarse_depth")
        results = []
        pending = []                       # (frame index, batch, its early-stop guards)
        self._guards = []
        for k, batch in enumerate(batches):
            slot = slots[k % n]
            slot.stream.wait_stream(cur)
            with torch.cuda.stream(slot.stream):
                img = self._view_images(batch, chunk, slot.scene, slot.ws)
                if self._guards:
                    pending.append((k, batch, self._guards))
                    self._guards = []
                if not device_output:
                    host = {kk: torch.empty(img[kk].shape, dtype=torch.float32).pin_memory() for kk in keys}
                    for kk in keys:
                        host[kk].copy_(img[kk], non_blocking=True)
                    results.append(host)
                else:
                    for t in img.values():
                        t.record_stream(cur)          # consumed on the caller's stream after the join below
                    results.append(img)
        for slot in slots:
            cur.wait_stream(slot.stream)
        if not device_output:
            torch.cuda.current_stream(self.device).synchronize()
        self._frame_src = None      # slot 0 shares the renderer's scene: it now holds the last frame slot 0 rendered
        # hand-over: every sliced frame's guard is looked at (this waits for the frames' compositors - the images of a device_output
        # call are then all but done); a frame that broke its bound is rendered again, alone, in one pass
        for k, batch, guards in pending:
            again = lambda b=batch: self._render_view(b, chunk, device_output)
            checked = self._hand_over(results[k], again, guards=guards)
            if checked is not results[k]:
                self._frame_src = None
                results[k] = checked
        return results

    def image_metrics(self, color_img, batch, clamp=True):
        """test.py:62-71 on the device: clamp to [0,1], psnr with and without mask_at_box against batch["img"].
        Returns a dict of python floats (one 32-byte device->host copy)."""
        img = color_img.to(self.device)
        if clamp:
            img = torch.clamp(img, min=0.0, max=1.0)
        gt = batch["img"][0]
        gt = self._dev(gt, gt.dtype if gt.dtype in (torch.float32, torch.float64) else torch.float32)   # staged upload (6 MB of float64)
        m = _lib.image_psnr(img, gt, self._dev(batch["mask_at_box"][0], torch.uint8)).cpu()
        return {"mse": float(m[0]), "mse_wMask": float(m[1]), "psnr_woMask": float(m[2]), "psnr_wMask": float(m[3])}

    # ---- density query for marching cubes (reference :280-296) ----
    def query_volume(self, pts, code_idx, transparent_mask=None, batch_info={}):
        """batch_info needs only 'poses' (the density-only branch of the reference, model/spacenet.py:223-241)."""
        B, N = pts.shape[:2]
        with torch.no_grad():
            density = self.net(pts, None, code_idx, batch_info, density_only=True)
        if transparent_mask is not None:
            density[transparent_mask.to(density.device).reshape([-1, 1])] = 0
        return density.reshape(B, N, 1)
