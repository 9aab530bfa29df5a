"""ctypes binding of libdsnerf_hip.so (include/dsnerf.h) + thin torch-tensor plumbing.

PyTorch is used for device memory, streams and (elsewhere) torch.distributed only.  There is no
eager / CPU fallback: if the library is missing or a call fails, a RuntimeError is raised.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# DSNERF_LIB: another build of the same library (kernel-variant experiments, scripts/variants.sh)
LIB_PATH = os.environ.get("DSNERF_LIB") or os.path.join(_HERE, "libdsnerf_hip.so")
_lib = None

PARAM_ORDER = [
    "nerf.embedding.weight",
    "nerf.stage1.0.weight", "nerf.stage1.0.bias", "nerf.stage1.2.weight", "nerf.stage1.2.bias",
    "nerf.stage1.4.weight", "nerf.stage1.4.bias", "nerf.stage1.6.weight", "nerf.stage1.6.bias",
    "nerf.stage2.0.weight", "nerf.stage2.0.bias", "nerf.stage2.2.weight", "nerf.stage2.2.bias",
    "nerf.stage2.4.weight", "nerf.stage2.4.bias",
    "nerf.density_net.0.weight", "nerf.density_net.0.bias",
    "nerf.rgb_net.1.weight", "nerf.rgb_net.1.bias", "nerf.rgb_net.3.weight", "nerf.rgb_net.3.bias",
    "lighting_mlp.lights_encoding.0.weight", "lighting_mlp.lights_encoding.0.bias",
    "lighting_mlp.lights_encoding.2.weight", "lighting_mlp.lights_encoding.2.bias",
    "lighting_mlp.lights_encoding.4.weight", "lighting_mlp.lights_encoding.4.bias",
    "pose_mlp.0.weight", "pose_mlp.0.bias", "pose_mlp.2.weight", "pose_mlp.2.bias",
    "pose_mlp.4.weight", "pose_mlp.4.bias",
]

EXPORTS = [
    "dsn_abi_version", "dsn_last_error", "dsn_packed_param_bytes", "dsn_pack_params",
    "dsn_pack_params_host_image", "dsn_scene_bytes", "dsn_set_body", "dsn_set_frame", "dsn_set_frame_ex", "dsn_sample_gg",
    "dsn_sample_uniform", "dsn_warp", "dsn_field", "dsn_field_record_bytes",
    "dsn_field_forward", "dsn_field_reverse", "dsn_shade", "dsn_composite",
    "dsn_render_workspace_bytes", "dsn_render_rays", "dsn_grad_workspace_bytes", "dsn_render_rays_grad",
    "dsn_image_workspace_bytes", "dsn_image_scatter", "dsn_image_psnr", "dsn_debug_screen", "dsn_field_screen", "dsn_lbs_warp", "dsn_render_rays_train", "dsn_debug_nn_stats", "dsn_camera_rays",
    "dsn_pose_state_bytes", "dsn_set_pose", "dsn_light", "dsn_calibrate_workspace_bytes", "dsn_calibrate_screen",
    "dsn_set_screen_margin", "dsn_module_grad", "dsn_early_stop_eps", "dsn_calibrate_screen_frame",
    "dsn_early_stop_eps_scaled", "dsn_set_early_stop_colour_scale", "dsn_nn_header_offsets", "dsn_render_workspace_bytes_for",
    "dsn_render_workspace_record_capacity", "dsn_stop_slice_len", "dsn_stop_stats_slice_len", "dsn_early_stop_colour_headroom", "dsn_render_rays_ex",
    "dsn_render_rays_grad_ex", "dsn_render_rays_train_ex", "dsn_aux_create", "dsn_aux_destroy",
]

SKIP_TRANSPARENT = 1
NN_EXHAUSTIVE = 2
FIELD_FP32 = 4
SAMPLE_UNIFORM = 8
DENSITY_SCREEN = 16           # opt-in since ABI 5 (was NO_SCREEN with the opposite meaning)
SCREEN_AUDIT = 32
EARLY_STOP, STOP_STATS = 64, 128
PHASE_GEOMETRY, PHASE_FIELD, PHASE_SHADE = 256, 512, 1024      # dsn_render_rays: enqueue only these parts of the frame (0 = all)
SHARE_CUS = 2048              # frames in flight: the persistent field kernels take 7/8 of the compute units (dsnerf.h)
CNT_STOP = 56                 # [56] samples left out by ray termination, [57] samples not shaded, [58] STOP_STATS: what early stop would leave out
CNT_BYTES = 8192              # bytes of the workspace's count area (256 diagnostic words + the slice histogram of STOP_STATS from word 256)
CNT_HIST = 256
CNT_COLOUR_MAX = 59           # largest |colour| the compositor of an eval frame weighed (float bits): the scale of the early-stop bound
EARLY_STOP_COLOUR_HEADROOM = 2.0   # the colour scale handed to the library = this x the largest colour seen so far (Renderer / bench.py)
EARLY_STOP_MIN_SKIPPED = 0.04  # Renderer / bench.py: share of the non-transparent samples early stop must leave out before the slicing pays (it costs ~0.5 ms = 3 % of a 512 x 512 x 64 frame when it leaves out nothing)
SCREEN_MARGIN_FLOOR, SCREEN_MARGIN_CAP = 0.002, 0.15      # = F16_SCREEN_FLOOR / F16_SCREEN_CAP of csrc/dsn_field16.hip
SCREEN_HEADROOM = 10.0        # = F16_SCREEN_HEADROOM: every calibration point is this factor in deviation away from a wrong drop
SCREEN_MIN_DROPPED = 0.35     # PackedParams.calibrate_screen: below this share of dropped calibration points the screen stays off
RAYS_ZJU, RAYS_H36M = 0, 1
FRAME_FINE_ONLY = 1           # dsn_set_frame_ex: only the fine nearest-face level of the posed mesh (points beyond it: exhaustive sweep)
FRAME_LAZY_LISTS = 2          # dsn_set_frame_ex: grid geometry only - the frame that uses the level builds the lists of the cells it visits
LAZY_LISTS = 4096             # dsn_render_rays_ex: ... which is this flag (Scene.lazy says whether the scene's frame was set that way)
# int32 words of the render workspace the library leaves diagnostics in (include/dsnerf.h)
CNT_ACTIVE, CNT_POS, CNT_KEEP, CNT_AUDIT, CNT_RANGE = 0, 16, 32, 40, 48
CNT_SEL, CNT_LIT = 13, 14       # DSN_EARLY_STOP frames: slots of the reverse pass, shaded samples (DSN_CNT_SEL / DSN_CNT_LIT)


def lib():
    """Load the HIP library (raises if it has not been built: python dual-space-nerf_amd/build.py)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing - build it with `python dual-space-nerf_amd/build.py` "
                "(hipcc --offload-arch=gfx950). There is no fallback path.")
        L = C.CDLL(LIB_PATH)
        L.dsn_last_error.restype = C.c_char_p
        L.dsn_early_stop_eps.restype = C.c_float
        L.dsn_early_stop_eps_scaled.restype = C.c_float
        L.dsn_early_stop_colour_headroom.restype = C.c_float
        L.dsn_render_workspace_record_capacity.restype = C.c_int64
        L.dsn_render_workspace_record_capacity.argtypes = [C.c_int, C.c_int, C.c_size_t]
        L.dsn_render_workspace_bytes_for.argtypes = [C.c_int, C.c_int, C.c_float]
        for n in ("dsn_packed_param_bytes", "dsn_scene_bytes", "dsn_render_workspace_bytes", "dsn_render_workspace_bytes_for", "dsn_field_record_bytes",
                  "dsn_grad_workspace_bytes", "dsn_image_workspace_bytes", "dsn_pose_state_bytes",
                  "dsn_calibrate_workspace_bytes"):
            getattr(L, n).restype = C.c_size_t
        if L.dsn_abi_version() != 8:
            raise RuntimeError(f"{LIB_PATH} has ABI version {L.dsn_abi_version()}, this binding needs 8 - rebuild it "
                               "(python dual-space-nerf_amd/build.py)")
        _lib = L
    return _lib


def _check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed: {lib().dsn_last_error().decode()}")


def cpu_quota_cores():
    """CPU bandwidth this process may use, in cores, from the cgroup (v2 cpu.max / v1 cpu.cfs_quota_us); None = unlimited / unknown.
    os.cpu_count() and the affinity mask do not see it: the GPU boxes show 256 hardware threads under a quota of 16 cores."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except Exception:
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
            q = int(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            per = int(f.read())
        return None if q <= 0 else q / float(per)
    except Exception:
        return None


_pool_warned = False


def fit_host_pool(keep=False):
    """Cap torch's intra-op pool at the cgroup's CPU quota (minus two cores for the thread that feeds the GPU and the HIP
    runtime's helper threads).  torch sizes its OpenMP pool from the hardware thread count (128-256 on the GPU boxes) whatever the
    quota (16 cores there): every torch CPU op of the CALLER (test.py:61-76 runs torch.clamp / psnr / .cpu() between frames) then
    wakes a team that burns the whole 100 ms quota period in a few milliseconds of busy-waiting at the pool's dock, the kernel
    freezes the container for the rest of the period, and a 19 ms frame takes 60-90 ms (scripts/h2h_guard_probe.py,
    profiles/r03a_h2h_guard.json: nr_throttled counts them; mean 36 ms in BENCH_r02).  No quota, pool already small, or
    DSN_HOST_POOL=keep / keep=True (Renderer(host_pool="keep")): nothing happens.  Lowering the caller's setting is said once, as a
    warning (it is process-global: it also applies to the caller's own CPU work).  Returns (threads before, threads now, quota)."""
    global _pool_warned
    before = torch.get_num_threads()
    quota = cpu_quota_cores()
    if keep or quota is None or os.environ.get("DSN_HOST_POOL", "") == "keep":
        return before, before, quota
    # one process per GPU: the quota is shared by the processes of this node (torchrun exports LOCAL_WORLD_SIZE)
    try:
        local_world = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))
    except ValueError:
        local_world = 1
    want = max(1, int(quota / local_world) - 2)
    if before > want:
        torch.set_num_threads(want)
        if not _pool_warned:
            _pool_warned = True
            import warnings
            warnings.warn(f"dsnerf_amd: torch.set_num_threads({want}) (was {before}): this process may use {quota:g} cores (cgroup quota"
                          f"{', shared by ' + str(local_world) + ' local ranks' if local_world > 1 else ''}) and a larger intra-op pool gets it "
                          "throttled while frames are in flight.  Renderer(host_pool=\"keep\") or DSN_HOST_POOL=keep leave the pool alone.")
    return before, torch.get_num_threads(), quota


def require_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError("dsnerf_amd needs a ROCm device (gfx950); no CPU fallback exists")


def _scratch(nbytes, device):
    """opaque device scratch for the library (scene blobs, workspaces): uninitialised, as the C ABI allows.  DSN_POISON_SCRATCH=1 (tests):
    every byte 0xFF - NaN floats, -1 integers - so that nothing can come to rely on a fresh allocation reading as zeros"""
    t = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
    if os.environ.get("DSN_POISON_SCRATCH"):
        t.fill_(255)
    return t


def _ptr(t, dtype=None):
    if t is None:
        return C.c_void_p(0)
    assert t.is_cuda and t.is_contiguous(), "device-resident contiguous tensor required"
    if dtype is not None:
        assert t.dtype == dtype, (t.dtype, dtype)
    return C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _f32(t, device):
    return t.to(device=device, dtype=torch.float32).contiguous()


class PackedParams:
    """MFMA-ordered device image of the 33 state_dict tensors (dsn_pack_params)."""

    def __init__(self, device):
        require_gpu()
        self.device = torch.device(device)
        self.buf = torch.zeros(lib().dsn_packed_param_bytes(), dtype=torch.uint8, device=self.device)
        self._versions = None
        self._keep = None
        self.generation = 0        # bumped by every re-pack: what calibrations / caches of derived state key on
        self.screen = None         # dict(deviation, margin, overflow_fraction, points, usable) once calibrate_screen() has run
        self.early_stop = None     # dict(skipped_fraction, usable) once a frame's DSN_STOP_STATS have been read (Renderer / bench.py)
        self.colour_scale = 1.0    # colour scale of the early-stop threshold in the packed image (set_early_stop_colour_scale)

    def update(self, state: dict, force=False):
        """state: name -> tensor (any device).  Re-packs only when a tensor changed (data pointer / version counter).
        Edits made through `param.data` do not bump the version counter: pass force=True (or call net.packed(force=True))
        after such an edit."""
        tensors = [state[k] for k in PARAM_ORDER]
        versions = tuple((t.data_ptr(), t._version) for t in tensors)
        if not force and versions == self._versions:
            return self
        dev = [_f32(t.detach(), self.device) for t in tensors]
        ptrs = (C.c_void_p * len(dev))(*[t.data_ptr() for t in dev])
        _check(lib().dsn_pack_params(ptrs, _ptr(self.buf), _stream()), "dsn_pack_params")
        self._keep = dev  # keep sources alive until the pack kernel has run (stream-ordered)
        self._versions = versions
        self.generation += 1
        self.screen = None         # the packed image carries the conservative default margin again
        self.early_stop = None
        self.colour_scale = 1.0    # (dsn_pack_params resets it)
        return self

    def set_early_stop_colour_scale(self, scale: float):
        """colour scale c of DSN_EARLY_STOP's threshold eps(S, c) = min(2^-20, 1e-4 / (2 (S + 1) max(1, c))) for these parameters:
        frames stay within (S + 1) eps c' of the one-pass frame for colours up to c' - half of the 1e-4 bar, absolute, while c' <= c."""
        scale = max(1.0, float(scale))
        _check(lib().dsn_set_early_stop_colour_scale(_ptr(self.buf), C.c_float(scale), _stream()), "dsn_set_early_stop_colour_scale")
        self.colour_scale = scale
        return scale

    def screen_pays(self, early_stop_on: bool = False) -> bool:
        """Is the (calibrated) density screen worth running?  `usable` of calibrate_screen, except that with ray termination in use
        the share of samples it drops counts among the samples that are still evaluated: what termination leaves out is the dense
        interior, which the screen could never drop - dropped / (1 - skipped) >= SCREEN_MIN_DROPPED."""
        sc = self.screen
        if not sc:
            return False
        if not early_stop_on or not self.early_stop or sc.get("dropped_fraction") is None or not sc.get("safe"):
            return bool(sc["usable"])
        f = min(max(float(self.early_stop.get("skipped_fraction", 0.0)), 0.0), 0.99)
        return sc["dropped_fraction"] / (1.0 - f) >= SCREEN_MIN_DROPPED

    def calibrate_screen(self, scene: "Scene", n_points: int = 1 << 20, other_frames=(0, 125, 250, 375, 499), frame=None):
        """Measure the density screen's margin for THESE parameters (dsn_calibrate_screen; synchronises: meant to run once per
        checkpoint, Renderer does it lazily before the first eval-mode frame after the parameters changed): on the scene's
        current frame state with n_points points, and - the first layer's bias depends on the frame's embedding row - with the
        same pose under a spread of other frame codes (`other_frames`, n_points / 4 points each).  The margin leaves every point
        seen a factor 10 of headroom in deviation against a wrong drop (include/dsnerf.h: margin = 10 max(dev - rel / 10), at least
        0.002, +inf above 0.15).  Returns / stores dict(deviation, margin_statistic, margin, overflow_fraction, points,
        dropped_fraction, safe, usable); safe = False: the screen would need a margin above the cap; usable = safe and it
        drops enough samples to pay for itself.
        frame = (RenderWorkspace, R, S) of a frame whose geometry phase has run (render_rays(..., phases=PHASE_GEOMETRY)): the
        calibration points are then the canonical points of that frame's non-transparent samples (+ a 2 cm halo) instead of a cube
        around the canonical centroids - what Renderer and bench.py do: the margin and the share the screen drops are measured on what
        is rendered (dsn_calibrate_screen_frame)."""
        ws = _scratch(lib().dsn_calibrate_workspace_bytes(C.c_int64(n_points)), self.device)
        out = torch.zeros(8, dtype=torch.float32, device=self.device)

        def run(n):
            if frame is not None:       # (render workspace, R, S) of a frame whose geometry phase has run: calibrate on ITS points
                rws, R_, S_ = frame
                _check(lib().dsn_calibrate_screen_frame(_ptr(scene.buf), scene.V, scene.F, _ptr(self.buf), _ptr(rws.buf), int(R_), int(S_),
                                                        C.c_int64(n), _ptr(ws), _ptr(out), _stream()), "dsn_calibrate_screen_frame")
            else:
                _check(lib().dsn_calibrate_screen(_ptr(scene.buf), scene.V, scene.F, _ptr(self.buf), C.c_int64(n), _ptr(ws), _ptr(out),
                                                  _stream()), "dsn_calibrate_screen")
            return [float(v) for v in out.cpu()[:6]]

        state = getattr(scene, "_pose_args", None)          # (poses, frame_idx, zero_code, light_shift, rot, rot_center) of set_frame
        d = t = ovf = 0.0
        total = 0
        if state is not None and not state[2] and other_frames:
            poses, frame_idx, zero_code, ls, r, rc = state
            for f in other_frames:
                if f == frame_idx:
                    continue
                _set_pose(scene.buf, self, poses, None, f, zero_code, ls, r, rc, self.device)
                d2, _, ovf2, n2, _, t2 = run(max(n_points // 4, 1024))
                d, t, ovf, total = max(d, d2), max(t, t2), max(ovf, ovf2), total + n2
            _set_pose(scene.buf, self, poses, None, frame_idx, zero_code, ls, r, rc, self.device)      # back to the frame's own state
        if frame is not None:
            # ... and the cube around the canonical centroids under the frame's own state (the points of ONE frame - let alone
            # of its first chunk - are a sample of what a sequence visits; the margin is that of the joint set, never below the cube's)
            _frame, frame = frame, None
            d2, _, ovf2, n2, _, t2 = run(max(n_points // 4, 1024))
            d, t, ovf, total = max(d, d2), max(t, t2), max(ovf, ovf2), total + n2
            frame = _frame
        # the frame's own state last, with the other states' statistic carried in (out[7]): the margin it leaves in the packed image and
        # the share of points it counts as dropped are those of the joint set
        out[7:8].fill_(t)
        d1, m, ovf1, n1, dropped, t = run(n_points)
        d, ovf, total = max(d, d1), max(ovf, ovf1), total + n1
        safe = bool(m < float("inf"))
        # The screen costs ~0.3 of an accurate forward pass per sample (k_screen16 0.71 us vs k_field16<forward> 2.36 us per
        # thousand samples): it pays only if it drops more than that share.  A network that is dense everywhere near the
        # surface (every calibration point sigma > 0) is better off without it.
        self.screen = {"deviation": d, "margin_statistic": t, "margin": m, "overflow_fraction": ovf, "points": int(total),
                       "dropped_fraction": dropped, "safe": safe, "usable": safe and dropped >= SCREEN_MIN_DROPPED,
                       "points_from": "frame + centroid cube" if frame is not None else "centroid cube"}
        return self.screen

    def set_screen_margin(self, margin: float):
        _check(lib().dsn_set_screen_margin(_ptr(self.buf), C.c_float(margin), _stream()), "dsn_set_screen_margin")
        self.screen = {"deviation": None, "margin": float(margin), "overflow_fraction": None, "points": 0, "dropped_fraction": None,
                       "safe": None, "usable": margin < float("inf")}   # (a margin <= 0 is unsafe: tests use it to provoke the audit)
        return self.screen


class Scene:
    """Body model + per-frame state blob (dsn_set_body / dsn_set_frame)."""

    def __init__(self, canonical_vertex, faces, device):
        require_gpu()
        self.device = torch.device(device)
        canon = _f32(canonical_vertex.reshape(-1, 3), self.device)
        f = faces.to(device=self.device, dtype=torch.int32).contiguous()
        self.V, self.F = canon.shape[0], f.shape[0]
        self.buf = _scratch(lib().dsn_scene_bytes(self.V, self.F), self.device)
        _check(lib().dsn_set_body(_ptr(self.buf), _ptr(canon), _ptr(f), self.V, self.F, _stream()), "dsn_set_body")
        self._keep = (canon, f)
        self.frame_key = None
        # Nearest-face list levels that do not fit their (fixed) capacity are switched off by the build and every query falls through to
        # the next level / the exhaustive sweep - same index, 10-50x slower (silently).  The canonical mesh's levels are
        # built once: looked at here (one synchronisation at set-up); the posed mesh's are rebuilt per frame: their headers are copied
        # out asynchronously every NN_WATCH_EVERY-th frame and looked at when the next frame is set (nn_watch()).
        off = (C.c_size_t * 4)()
        _check(lib().dsn_nn_header_offsets(self.V, self.F, off), "dsn_nn_header_offsets")
        self._nn_off = [int(o) for o in off]
        self._nn_probe = None
        self._nn_host = None
        self._nn_frames = 0
        self.lazy = False              # the current frame was set with DSN_FRAME_LAZY_LISTS (render_rays then passes DSN_LAZY_LISTS)
        self.nn_overflow = {}          # level name -> (entries needed, capacity) of every overflow seen
        st = nn_stats(self)
        for name in ("canon_fine", "canon_coarse"):
            self._judge_level(name, st[name])

    NN_WATCH_EVERY = 16
    _NN_NAMES = ("world_fine", "world_coarse", "canon_fine", "canon_coarse")

    def _judge_level(self, name, rec):
        ncell, ok, total, cap = rec
        if cap <= 0:          # a level that was not built (fine-only frames leave the posed mesh's coarse level switched off)
            return
        if total > cap and name not in self.nn_overflow:
            import warnings
            warnings.warn(f"dsnerf_amd: the {name.replace('_', ' ')} nearest-face level of this mesh needs {total} list entries, its capacity is "
                          f"{cap}: the level is switched off and its queries take the next level / the exhaustive sweep over all {self.F} "
                          "centroids - same results, but the nearest-face search of every frame is several times slower "
                          "(Scene.nn_overflow; a mesh far denser than SMPL's 13 776 faces per body needs larger DSN capacities)")
        if total > cap:
            self.nn_overflow[name] = (int(total), int(cap))

    def nn_watch(self, wait=False):
        """look at the posed mesh's level headers an earlier frame left (no wait unless asked); returns the overflow record"""
        if self._nn_probe is not None:
            snap, ev = self._nn_probe
            if wait or ev.query():
                ev.synchronize()
                self._nn_probe = None
                w = snap.view(torch.int32)
                for i, name in enumerate(self._NN_NAMES[:2]):
                    self._judge_level(name, tuple(int(w[16 * i + k]) for k in (8, 9, 10, 11)))
        return self.nn_overflow

    def _watch_headers(self):
        """copy the posed mesh's two level headers out (asynchronously, into ONE persistent page-locked buffer: ADVICE r04 - a fresh
        pin_memory() per probe is a hipHostMalloc on the hot path) for nn_watch() to look at later"""
        if self._nn_host is None:
            self._nn_host = torch.empty(128, dtype=torch.uint8).pin_memory()
        host = self._nn_host
        for i in range(2):      # world fine / coarse headers, 64 bytes each
            host[64 * i:64 * i + 64].copy_(self.buf[self._nn_off[i]:self._nn_off[i] + 64], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._nn_probe = (host, ev)

    def set_frame(self, packed: PackedParams, xyz, poses, frame_idx: int, zero_code: bool = False,
                  light_shift=None, rot=None, rot_center=None, reuse: bool = False, fine_only: bool = False, lazy: bool = False):
        """reuse=True: skip the call when the scene already holds exactly this frame for these parameters (same tensors at
        the same versions) - used by the backward of a training step, which follows its own forward.
        fine_only=True (DSN_FRAME_FINE_ONLY): build only the fine nearest-face level of the posed mesh - enough for every sample
        of rays clipped to the body's bounds (what Renderer.render / render_view produce); points further than 0.12 m outside the
        posed centroids' box then take the exhaustive sweep (same index).
        lazy=True (DSN_FRAME_LAZY_LISTS, implies fine_only): only the fine grid is laid out; render_rays on this scene then builds
        the lists of the cells ITS samples visit (it passes DSN_LAZY_LISTS while Scene.lazy is set).  Stage calls on a lazily set
        scene (warp, lbs_warp, a training forward) stay exact but take the exhaustive sweep: set the frame without `lazy` for those."""
        vkey = lambda a: None if a is None else (a.data_ptr(), a._version, tuple(a.shape))
        key = (vkey(xyz), vkey(poses), int(frame_idx), bool(zero_code), vkey(light_shift), vkey(rot), vkey(rot_center),
               id(packed), packed._versions, bool(fine_only), bool(lazy))
        if reuse and key == self.frame_key:
            return self
        self.frame_key = key
        # (a lazily set frame's lists - and the header that says whether they fit - are completed by its render call: what the previous
        #  frame left is looked at here, before this frame overwrites it)
        self.nn_watch()
        if self.lazy and self._nn_probe is None and self._nn_frames % self.NN_WATCH_EVERY == 1:
            self._watch_headers()
        self.lazy = bool(lazy)
        xyz = _f32(xyz.reshape(-1, 3), self.device)
        assert xyz.shape[0] == self.V, "xyz must have the body model's vertex count"
        poses = _f32(poses.reshape(24, 3), self.device)
        ls = None if light_shift is None else _f32(light_shift.reshape(-1)[:3], self.device)
        r = None if rot is None else _f32(rot.reshape(-1)[:4], self.device)
        rc = None if rot_center is None else _f32(rot_center.reshape(-1)[:2], self.device)
        _check(lib().dsn_set_frame_ex(_ptr(self.buf), self.V, self.F, _ptr(packed.buf), _ptr(xyz), _ptr(poses),
                                      int(frame_idx), int(bool(zero_code)), _ptr(ls), _ptr(r), _ptr(rc),
                                      (FRAME_LAZY_LISTS | FRAME_FINE_ONLY) if lazy else (FRAME_FINE_ONLY if fine_only else 0), _stream()),
               "dsn_set_frame")
        self._keep_frame = (xyz, poses, ls, r, rc)
        self._pose_args = (poses, int(frame_idx), bool(zero_code), ls, r, rc)
        if not lazy and self._nn_probe is None and self._nn_frames % self.NN_WATCH_EVERY == 0:
            self._watch_headers()
        self._nn_frames += 1
        return self


def _set_pose(buf, packed, poses, pose_feat, frame_idx, zero_code, light_shift, rot, rot_center, device):
    f = lambda a, n: None if a is None else _f32(a.reshape(-1)[:n], device)
    po = None if poses is None else _f32(poses.reshape(24, 3), device)
    pf, ls, r, rc = f(pose_feat, 16), f(light_shift, 3), f(rot, 4), f(rot_center, 2)
    _check(lib().dsn_set_pose(_ptr(buf), _ptr(packed.buf), _ptr(po), _ptr(pf), int(frame_idx), int(bool(zero_code)), _ptr(ls),
                              _ptr(r), _ptr(rc), _stream()), "dsn_set_pose")
    return (po, pf, ls, r, rc)


def scene_set_pose(scene: Scene, packed: PackedParams, poses, frame_idx, zero_code=False, light_shift=None, rot=None,
                   rot_center=None, pose_feat=None):
    """pose code / embedding row / light edits of a scene, mesh untouched (dsn_set_pose)."""
    scene._keep_pose = _set_pose(scene.buf, packed, poses, pose_feat, frame_idx, zero_code, light_shift, rot, rot_center, scene.device)
    scene.frame_key = None     # no longer the state a set_frame(reuse=True) left
    scene._pose_args = None
    return scene


class PoseState:
    """Per-frame network state WITHOUT a body model (dsn_pose_state_bytes): all a density-only query or a stand-alone
    SpaceNet.forward needs.  Quacks like a Scene for `field()` (V = F = 0)."""

    def __init__(self, device):
        require_gpu()
        self.device = torch.device(device)
        self.V = self.F = 0
        self.buf = torch.zeros(lib().dsn_pose_state_bytes(), dtype=torch.uint8, device=self.device)

    def set_pose(self, packed: PackedParams, poses, frame_idx, zero_code=False, light_shift=None, rot=None, rot_center=None,
                 pose_feat=None):
        self._keep = _set_pose(self.buf, packed, poses, pose_feat, frame_idx, zero_code, light_shift, rot, rot_center, self.device)
        return self


def light(packed: PackedParams, normal, xyz_world, view_dir_world, essence, fp32=False):
    """model/spacenet.py:174-188 LightingMLP.forward as a pure function (dsn_light): [N,3] x 4 -> colour [N,3]."""
    dev = packed.device
    a = [_f32(t.reshape(-1, 3), dev) for t in (normal, xyz_world, view_dir_world, essence)]
    N = a[0].shape[0]
    assert all(t.shape[0] == N for t in a), "normal, xyz_world, view_dir_world and essence_feature must have one row per point"
    col = torch.empty(N, 3, dtype=torch.float32, device=dev)
    scratch = _scratch(lib().dsn_pose_state_bytes(), dev)
    _check(lib().dsn_light(_ptr(packed.buf), _ptr(a[0]), _ptr(a[1]), _ptr(a[2]), _ptr(a[3]), C.c_int64(N), _ptr(col), _ptr(scratch),
                           FIELD_FP32 if fp32 else 0, _stream()), "dsn_light")
    return col


def nn_stats(scene: Scene):
    """Diagnostics (synchronising): dict level -> (ncell, ok, total, cap)."""
    out = (C.c_int32 * 16)()
    _check(lib().dsn_debug_nn_stats(_ptr(scene.buf), scene.V, scene.F, out, _stream()), "dsn_debug_nn_stats")
    names = ["world_fine", "world_coarse", "canon_fine", "canon_coarse"]
    return {n: tuple(out[4 * i:4 * i + 4]) for i, n in enumerate(names)}


# ------------------------------------------------------------------------------------------------
# stage calls (device tensors in, device tensors out)
# ------------------------------------------------------------------------------------------------
def sample(scene: Scene, ray_o, ray_d, near, far, S, t_vals, jitter=None, want_pts=True, gg=True):
    """near/far are updated in place (like utils/pts_utils.py:52-53)."""
    R = ray_o.shape[0]
    dev = scene.device
    z = torch.empty(R, S, dtype=torch.float32, device=dev)
    pts = torch.empty(R, S, 3, dtype=torch.float32, device=dev) if want_pts else None
    fn = lib().dsn_sample_gg if gg else lib().dsn_sample_uniform
    _check(fn(_ptr(scene.buf), scene.V, scene.F, _ptr(ray_o, torch.float32), _ptr(ray_d, torch.float32),
              _ptr(near, torch.float32), _ptr(far, torch.float32), R, S, _ptr(t_vals, torch.float32), _ptr(jitter),
              _ptr(z), _ptr(pts), _stream()), "dsn_sample")
    return pts, z


def warp(scene: Scene, pts, ray_d, S, want_dir=True, want_uvh=False, want_active=False, exhaustive=False):
    pts = pts.reshape(-1, 3)
    N = pts.shape[0]
    dev = scene.device
    out = {
        "face_idx": torch.empty(N, dtype=torch.int32, device=dev),
        "transparent": torch.empty(N, dtype=torch.uint8, device=dev),
        "x_c": torch.empty(N, 3, dtype=torch.float32, device=dev),
    }
    uv = h = rdc = lst = cnt = None
    if want_uvh:
        uv = out["uv"] = torch.empty(N, 2, dtype=torch.float32, device=dev)
        h = out["h"] = torch.empty(N, dtype=torch.float32, device=dev)
    if want_dir and ray_d is not None:
        rdc = out["ray_d_can"] = torch.empty(N, 3, dtype=torch.float32, device=dev)
    if want_active:
        lst = out["active_list"] = torch.empty(N, dtype=torch.int32, device=dev)
        cnt = out["active_count"] = torch.zeros(64, dtype=torch.int32, device=dev)
    _check(lib().dsn_warp(_ptr(scene.buf), scene.V, scene.F, _ptr(pts, torch.float32), _ptr(ray_d), C.c_int64(N), S,
                          _ptr(out["face_idx"]), _ptr(uv), _ptr(h), _ptr(out["transparent"]), _ptr(out["x_c"]),
                          _ptr(rdc), _ptr(lst), _ptr(cnt), NN_EXHAUSTIVE if exhaustive else 0, _stream()), "dsn_warp")
    return out


def field(scene: Scene, packed: PackedParams, x_c, want_essence=True, want_grad=True, active=None, fp32=False):
    x_c = x_c.reshape(-1, 3)
    N = x_c.shape[0]
    dev = scene.device
    sigma = torch.zeros(N, dtype=torch.float32, device=dev)
    ess = torch.zeros(N, 3, dtype=torch.float32, device=dev) if want_essence else None
    g = torch.zeros(N, 3, dtype=torch.float32, device=dev) if want_grad else None
    lst, cnt = (None, None) if active is None else active
    _check(lib().dsn_field(_ptr(scene.buf), scene.V, scene.F, _ptr(packed.buf), _ptr(x_c, torch.float32), C.c_int64(N),
                           _ptr(lst), _ptr(cnt), _ptr(sigma), _ptr(ess), _ptr(g), FIELD_FP32 if fp32 else 0, _stream()),
           "dsn_field")
    return sigma, ess, g


def field_forward(scene: Scene, packed: PackedParams, x_c, active=None):
    """sigma, essence for the listed points + the list of points with sigma > 0 (see dsn_field_forward)."""
    x_c = x_c.reshape(-1, 3)
    N = x_c.shape[0]
    dev = scene.device
    sigma = torch.zeros(N, dtype=torch.float32, device=dev)
    ess = torch.zeros(N, 3, dtype=torch.float32, device=dev)
    rec = _scratch(lib().dsn_field_record_bytes(C.c_int64(N)), dev)
    pos = torch.zeros(N, dtype=torch.int32, device=dev)
    pcnt = torch.zeros(1, dtype=torch.int32, device=dev)
    lst, cnt = (None, None) if active is None else active
    _check(lib().dsn_field_forward(_ptr(scene.buf), scene.V, scene.F, _ptr(packed.buf), _ptr(x_c, torch.float32), C.c_int64(N),
                                   _ptr(lst), _ptr(cnt), _ptr(sigma), _ptr(ess), _ptr(rec), _ptr(pos), _ptr(pcnt), _stream()),
           "dsn_field_forward")
    return sigma, ess, rec, (pos, pcnt)


def field_reverse(scene: Scene, packed: PackedParams, x_c, rec, pos, sigma, essence):
    """sigma / essence: the arrays field_forward returned (samples it flagged as outside the fp16 range are re-evaluated
    here in exact fp32, all three outputs)."""
    x_c = x_c.reshape(-1, 3)
    N = x_c.shape[0]
    g = torch.zeros(N, 3, dtype=torch.float32, device=scene.device)
    _check(lib().dsn_field_reverse(_ptr(scene.buf), scene.V, scene.F, _ptr(packed.buf), _ptr(x_c, torch.float32), C.c_int64(N),
                                   _ptr(pos[0]), _ptr(pos[1]), _ptr(rec), _ptr(g), _ptr(sigma, torch.float32),
                                   _ptr(essence, torch.float32), _stream()), "dsn_field_reverse")
    return g


def lbs_warp(scene: Scene, pts, smpl_weights, joint_transforms, bw_type="rigid_center", exhaustive=False):
    """dormant alternate (utils/render_utils.py:352-403 + utils/blend_utils.py:72-81): nearest-face blend weights and inverse
    LBS.  Returns dict(face_idx, weights [N,24], transparent, pts_zero [N,3])."""
    if bw_type not in ("rigid_center", "rigid_interp"):
        raise ValueError("unsupport value: bw_type")
    dev = scene.device
    pts = _f32(pts.reshape(-1, 3), dev)
    N = pts.shape[0]
    W = _f32(smpl_weights.reshape(-1, 24), dev)
    assert W.shape[0] == scene.V
    A = _f32(joint_transforms.reshape(24, 16), dev)
    out = {"face_idx": torch.empty(N, dtype=torch.int32, device=dev), "weights": torch.empty(N, 24, dtype=torch.float32, device=dev),
           "transparent": torch.empty(N, dtype=torch.uint8, device=dev), "pts_zero": torch.empty(N, 3, dtype=torch.float32, device=dev)}
    _check(lib().dsn_lbs_warp(_ptr(scene.buf), scene.V, scene.F, _ptr(pts), C.c_int64(N), _ptr(W), _ptr(A),
                              0 if bw_type == "rigid_center" else 1, _ptr(out["face_idx"]), _ptr(out["weights"]),
                              _ptr(out["transparent"]), _ptr(out["pts_zero"]), NN_EXHAUSTIVE if exhaustive else 0, _stream()),
           "dsn_lbs_warp")
    return out


def screen_debug(scene: Scene, packed: PackedParams, x_c):
    """plain-fp16 density and the magnitude of its terms for every point (dsn_debug_screen)."""
    x_c = x_c.reshape(-1, 3)
    N = x_c.shape[0]
    dev = scene.device
    sg = torch.zeros(N, dtype=torch.float32, device=dev)
    s1 = torch.zeros(N, dtype=torch.float32, device=dev)
    lst = torch.zeros(N, dtype=torch.int32, device=dev)
    cnt = torch.zeros(64, dtype=torch.int32, device=dev)
    _check(lib().dsn_debug_screen(_ptr(scene.buf), scene.V, scene.F, _ptr(packed.buf), _ptr(x_c, torch.float32), C.c_int64(N),
                                  _ptr(sg), _ptr(s1), _ptr(lst), _ptr(cnt), _stream()), "dsn_debug_screen")
    return sg, s1


def shade(scene: Scene, packed: PackedParams, x_c, grad, x_w, ray_d, essence, S, active=None, exhaustive=False, fp32=False):
    x_c = x_c.reshape(-1, 3)
    N = x_c.shape[0]
    dev = scene.device
    idx = torch.zeros(N, dtype=torch.int32, device=dev)
    n_w = torch.zeros(N, 3, dtype=torch.float32, device=dev)
    col = torch.zeros(N, 3, dtype=torch.float32, device=dev)
    lst, cnt = (None, None) if active is None else active
    _check(lib().dsn_shade(_ptr(scene.buf), scene.V, scene.F, _ptr(packed.buf), _ptr(x_c, torch.float32),
                           _ptr(grad.reshape(-1, 3), torch.float32), _ptr(x_w.reshape(-1, 3), torch.float32),
                           _ptr(ray_d, torch.float32), _ptr(essence.reshape(-1, 3), torch.float32), C.c_int64(N), S,
                           _ptr(lst), _ptr(cnt), _ptr(idx), _ptr(n_w), _ptr(col), (NN_EXHAUSTIVE if exhaustive else 0) | (FIELD_FP32 if fp32 else 0),
                           _stream()), "dsn_shade")
    return idx, n_w, col


def composite(colour, sigma, transparent, z_vals, ray_d, noise=None):
    R, S = z_vals.shape
    dev = z_vals.device
    rgb = torch.empty(R, 3, dtype=torch.float32, device=dev)
    disp = torch.empty(R, dtype=torch.float32, device=dev)
    acc = torch.empty(R, dtype=torch.float32, device=dev)
    w = torch.empty(R, S, dtype=torch.float32, device=dev)
    dep = torch.empty(R, dtype=torch.float32, device=dev)
    _check(lib().dsn_composite(_ptr(colour.reshape(-1, 3), torch.float32), _ptr(sigma.reshape(-1), torch.float32),
                               _ptr(transparent), _ptr(z_vals, torch.float32), _ptr(ray_d, torch.float32), _ptr(noise),
                               R, S, _ptr(rgb), _ptr(disp), _ptr(acc), _ptr(w), _ptr(dep), _stream()), "dsn_composite")
    return rgb, disp, acc, w, dep


RECORD_FRACTION_DEFAULT = 0.125      # = DSN_RECORD_FRACTION_DEFAULT (csrc/dsn_api.hip)


class RenderWorkspace:
    """The caller-owned scratch of the fused path, grown on demand.  Its relu-record capacity is ITS OWN (ABI 6: the library reads the
    capacity from the size it is handed; rounds 3-4 kept one process-wide fraction that every workspace re-read at every call).
    `fraction` = share of a big frame's samples the records are sized for; fit_records() asks for more after a frame was seen to
    need it.  A larger request is applied at the START of the next frame only (begin_frame(), called by render_rays in front of a
    frame's geometry phase), never between the phase calls of one frame: the buffer a frame's geometry phase filled is the buffer
    its field and shading phases read."""

    def __init__(self, device, fraction: float = RECORD_FRACTION_DEFAULT):
        self.device = torch.device(device)
        self.buf = None
        self.cap = 0
        self.fraction = float(fraction)      # in force for the buffer
        self.want_fraction = float(fraction)      # asked for: applied by begin_frame()
        self._sized = None

    def fit_records(self, positive_fraction: float, headroom: float = 1.25) -> float:
        """a frame put this share of its samples on the sigma > 0 list: make the record capacity cover `headroom` x that from the next
        frame on (samples beyond the capacity have their forward pass evaluated twice; Renderer / bench.py call this with their probe
        frame's count - with more headroom when the share is an ESTIMATE for sliced frames from a one-pass probe).  Never shrinks."""
        want = min(1.0, float(headroom) * float(positive_fraction))
        if want > self.want_fraction:
            self.want_fraction = want
        return self.want_fraction

    def begin_frame(self):
        self.fraction = max(self.fraction, self.want_fraction)

    def bytes_for(self, R, S):
        return lib().dsn_render_workspace_bytes_for(int(R), int(S), C.c_float(self.fraction))

    def record_capacity(self, R, S):
        """records the buffer holds for an R x S frame (what dsn_render_rays_ex will use)"""
        return int(lib().dsn_render_workspace_record_capacity(int(R), int(S), C.c_size_t(self.cap)))

    def get(self, R, S):
        need = self.bytes_for(R, S)
        if self.buf is None or self.cap < need:
            self.buf = _scratch(need, self.device)
            self.cap = need
        return self.buf


def render_rays(scene: Scene, packed: PackedParams, ws: RenderWorkspace, ray_o, ray_d, near, far, S, t_vals,
                jitter=None, noise=None, skip_transparent=True, want_weights=True, out=None, exhaustive=False,
                fp32=False, uniform=False, screen=False, train_cache=None, audit=False, early_stop=False, stop_stats=False, phases=0,
                share_cus=False, stop_schedule=None):
    """Whole hot path on R rays (can_render.py:137-168).  Returns dict of device tensors.
    phases: 0 = the whole frame; PHASE_GEOMETRY | PHASE_FIELD | PHASE_SHADE = only those parts, on the current stream (the caller
    orders the three calls of a frame with its own events and passes the same `out` / workspace to all of them: PhasePipeline).
    screen: DSN_DENSITY_SCREEN (opt-in: the plain-fp16 density screen in front of the accurate pass, margin as calibrated).
    early_stop: DSN_EARLY_STOP (eval mode: front-to-back slices, rays end once their transmittance is below eps(S, colour scale)).
    stop_stats: DSN_STOP_STATS (count what early stop would leave out; read ws word CNT_STOP + 2)."""
    R = ray_o.shape[0]
    dev = scene.device
    if out is None:
        out = {
            "color": torch.empty(R, 3, dtype=torch.float32, device=dev),
            "disp_map": torch.empty(R, dtype=torch.float32, device=dev),
            "acc_map": torch.empty(R, dtype=torch.float32, device=dev),
            "depth_map": torch.empty(R, dtype=torch.float32, device=dev),
            "z_vals": torch.empty(R, S, dtype=torch.float32, device=dev),
        }
        if want_weights:
            out["weights"] = torch.empty(R, S, dtype=torch.float32, device=dev)
    flags = SKIP_TRANSPARENT if (skip_transparent and noise is None) else 0
    if exhaustive:
        flags |= NN_EXHAUSTIVE
    if fp32:
        flags |= FIELD_FP32
    if uniform:
        flags |= SAMPLE_UNIFORM
    if screen:
        flags |= DENSITY_SCREEN
        if audit:
            flags |= SCREEN_AUDIT
    if early_stop and (flags & SKIP_TRANSPARENT) and not fp32:
        flags |= EARLY_STOP
    if stop_stats:
        flags |= STOP_STATS
    flags |= int(phases) & (PHASE_GEOMETRY | PHASE_FIELD | PHASE_SHADE)
    if share_cus:
        flags |= SHARE_CUS
    if scene.lazy:               # (the training forward takes the flag too since round 6: dsn_render_rays_train's fused geometry)
        flags |= LAZY_LISTS
    if not phases or (int(phases) & PHASE_GEOMETRY):
        ws.begin_frame()          # (a larger record capacity asked for since the last frame: the buffer may be replaced HERE, only here)
    buf = ws.get(R, S)
    if train_cache is not None:      # training forward: dense, and everything its backward needs stays in train_cache
        flags &= ~SKIP_TRANSPARENT
        gbuf = train_cache.get(R, S)
        # (ABI 8: the far canonical search beside the field kernel on the backward's auxiliary stream; DSN_TRAIN_FWD_AUX=0: A/B switch -
        #  the forward on one stream, the backward's chains still on two)
        ax = train_cache.aux() if os.environ.get("DSN_TRAIN_FWD_AUX", "1") != "0" else (None, None, None)
        _check(lib().dsn_render_rays_train_ex(_ptr(scene.buf), scene.V, scene.F, _ptr(packed.buf), _ptr(ray_o, torch.float32),
                                              _ptr(ray_d, torch.float32), _ptr(near, torch.float32), _ptr(far, torch.float32), R, S,
                                              _ptr(t_vals, torch.float32), _ptr(jitter), _ptr(noise), flags, _ptr(out["color"]),
                                              _ptr(out["disp_map"]), _ptr(out["acc_map"]), _ptr(out["depth_map"]),
                                              _ptr(out.get("weights")), _ptr(out["z_vals"]), _ptr(buf), _ptr(gbuf), _stream(),
                                              ax[0], ax[1], ax[2]),
               "dsn_render_rays_train")
        return out
    # stop_schedule (early_stop only): slice lengths chosen from a probe frame's statistics (choose_stop_schedule); None = uniform slices
    sched, n_sched = None, 0
    if stop_schedule is not None and (flags & EARLY_STOP):
        n_sched = len(stop_schedule)
        sched = (C.c_int32 * n_sched)(*[int(x) for x in stop_schedule])
    _check(lib().dsn_render_rays_ex(_ptr(scene.buf), scene.V, scene.F, _ptr(packed.buf), _ptr(ray_o, torch.float32),
                                    _ptr(ray_d, torch.float32), _ptr(near, torch.float32), _ptr(far, torch.float32), R, S,
                                    _ptr(t_vals, torch.float32), _ptr(jitter), _ptr(noise), flags, _ptr(out["color"]),
                                    _ptr(out["disp_map"]), _ptr(out["acc_map"]), _ptr(out["depth_map"]),
                                    _ptr(out.get("weights")), _ptr(out["z_vals"]), _ptr(buf), C.c_size_t(ws.cap), sched, n_sched, _stream()),
           "dsn_render_rays")
    return out


class PhasePipeline:
    """Frames in flight BY PHASE (round 3): two HIP streams, one for the matrix-bound field kernels (density screen, field forward /
    reverse: 13 of a 512 x 512 x 64 frame's 17 ms, one persistent workgroup per compute unit) and one for everything else (per-frame
    list build, sampler, nearest-face search, warp of frame k + 1; normals, lighting, compositing of frame k - 1), which is sized to
    fit into the LDS and registers the field workgroups leave on every compute unit and runs BESIDE them.  With one stream per FRAME
    (round 2) the two frames' phases lock into step - both do geometry, then their field kernels take turns - and 2.3 ms of geometry
    per frame run with no field kernel beside them (profiles/r03b_overlap_two_frame_streams.txt).
    MEASURED (profiles/r03c_overlap_phase.txt, r03c_ab_*.json): no gain on the default frame (16.50 vs 16.26-16.45 ms), 2 % on the
    converged set (10.66 vs 10.90).  The GPU is busy 99.7 % of the frame in both schemes; beside a persistent field workgroup the small
    kernels run 5-10x longer (k_light16 0.40 -> 3.4 ms) and slow the field kernels by 8 %: the geometry is work the chip has to do,
    not latency to hide.  Kept as an option (bench.py --overlap phase) and as the test of the phase flags.

    submit(geometry_fn, field_fn, shade_fn): geometry_fn() is enqueued on the side stream at once, field_fn() on the field stream
    behind it, shade_fn() on the side stream behind field_fn - but one submit later, so that the side stream never waits for a
    field phase with the next frame's geometry queued behind the wait.  flush() enqueues the pending shading; join() makes the
    caller's current stream wait for everything.  The functions enqueue on torch's current stream."""

    def __init__(self, device):
        self.device = torch.device(device)
        self.side = torch.cuda.Stream(device=self.device)
        self.field = torch.cuda.Stream(device=self.device)
        self.pending = None

    def submit(self, geometry_fn, field_fn, shade_fn):
        cur = torch.cuda.current_stream(self.device)
        self.side.wait_stream(cur)                      # inputs produced on the caller's stream
        with torch.cuda.stream(self.side):
            geometry_fn()
            e_g = torch.cuda.Event()
            e_g.record()
        with torch.cuda.stream(self.field):
            self.field.wait_event(e_g)
            field_fn()
            e_f = torch.cuda.Event()
            e_f.record()
        prev, self.pending = self.pending, (shade_fn, e_f)
        if prev is not None:
            self._shade(prev)

    def _shade(self, p):
        shade_fn, e_f = p
        with torch.cuda.stream(self.side):
            self.side.wait_event(e_f)
            shade_fn()

    def flush(self):
        if self.pending is not None:
            self._shade(self.pending)
            self.pending = None

    def join(self):
        self.flush()
        torch.cuda.current_stream(self.device).wait_stream(self.side)


def early_stop_eps(S: int, colour_scale: float = 1.0) -> float:
    """the termination / shading threshold of DSN_EARLY_STOP for rays of S samples and colours up to colour_scale:
    min(2^-20, 1e-4 / (2 (S + 1) max(1, colour_scale))) - the frame stays within (S + 1) eps x the largest colour of the one-pass
    frame (include/dsnerf.h), i.e. within 5e-5 absolute while the colours stay below the scale"""
    return float(lib().dsn_early_stop_eps_scaled(int(S), C.c_float(colour_scale)))


def stop_slice_len(R: int, S: int) -> int:
    """samples per uniform slice of DSN_EARLY_STOP for an R x S frame, from the library itself (dsn_stop_slice_len: what
    dsn_render_rays_ex and the DSN_STOP_STATS histogram use, the DSN_STOP_SLICE experiment override included)"""
    return int(lib().dsn_stop_slice_len(int(R), int(S)))


def read_stop_hist(ws, R: int, S: int):
    """(synchronises) the slice histogram a DSN_STOP_STATS frame left: numpy int64 [K + 1, K], hist[g][k] = non-transparent samples of
    uniform slice k on rays whose first slice with T < eps at its start is g (g = K: never)"""
    import numpy as np
    buf = ws if isinstance(ws, torch.Tensor) else ws.buf
    L = int(lib().dsn_stop_stats_slice_len(int(R), int(S)))      # (the histogram's own slice length: half the uniform slice where K <= 32 allows)
    K = (S + L - 1) // L
    c = buf[4 * CNT_HIST:4 * CNT_HIST + 4 * (K + 1) * K].view(torch.int32).cpu().numpy().astype(np.int64)
    return c.reshape(K + 1, K), L


def choose_stop_schedule(hist, L: int, S: int, round_samples: int = 128 * 224, launch_rounds: float = 0.3, quantise: bool = True,
                         round_weight: float = 0.5):
    """Slice lengths for dsn_render_rays_ex from a probe frame's histogram (read_stop_hist).  A slice that starts at uniform slice a
    and covers slices a .. b evaluates sum_k M[a][k], M[a][k] = samples of slice k on rays still alive at the start of a (sum over
    g > a of hist[g][k]).  Its forward launch runs in ROUNDS of the persistent grid - 128 samples per workgroup, round_samples per
    round (224 workgroups with frames in flight, DSN_SHARE_CUS) - so it costs ceil(samples / round_samples) rounds whatever its last
    round holds, plus about `launch_rounds` of a round for the small launches in front of it (transmittance, list filter).  Dynamic
    programming over the slice borders minimises the sum; slices stay within 64 samples.  round_weight (round 6's last session): the
    price of a slice is round_weight x its whole rounds + (1 - round_weight) x its plain sample count - the workgroups of the
    persistent grid fetch their tiles dynamically and the neighbours' kernels use what a last round leaves idle, so a started round
    does not cost a whole one: with 1.0 (rounds 5-6) a rank's eighth of the bench frame was cut into 2-4 slices and evaluated 20 % more
    samples than it had to; 0.5 measured best on the emulated partition (8 ranks 6.3 -> 6.5 x, the whole frame unchanged).
    quantise=False: round 4's model (samples +
    0.9 rounds per slice), which ignored the half-empty last round - 4 % on a whole 512 x 512 frame (12 rounds per slice), a third of
    the forward time of a rank's eighth of it (1.5 rounds per slice).
    Returns (list of lengths in samples, evaluated samples it predicts, evaluated samples of the uniform schedule)."""
    import numpy as np
    hist = np.asarray(hist, np.int64)
    K = hist.shape[1]
    M = np.zeros((K, K), np.int64)                       # M[a][k]: alive at the start of slice a
    for a in range(K):
        M[a] = hist[a + 1:].sum(0)
    rs = float(round_samples)
    # (DSN_STOP_ROUND_ALPHA / DSN_STOP_LAUNCH_ROUNDS: experiment overrides of round_weight / launch_rounds)
    alpha = float(os.environ.get("DSN_STOP_ROUND_ALPHA", round_weight))
    launch_rounds = float(os.environ.get("DSN_STOP_LAUNCH_ROUNDS", launch_rounds))

    def cost(n):
        if not quantise:
            return 0.9 * 32768.0 + float(n)
        return (float(launch_rounds) + alpha * float(-(-int(n) // int(round_samples))) + (1.0 - alpha) * float(n) / rs) * rs

    best = [0.0] + [float("inf")] * K
    prev = [0] * (K + 1)
    for b in range(1, K + 1):
        for a in range(max(0, b - 64 // max(L, 1)), b):
            if (min(b * L, S) - a * L) > 64:
                continue
            c = best[a] + cost(M[a][a:b].sum())
            if c < best[b] or (c == best[b] and a > prev[b]):      # (ties: the later border - termination checked more often)
                best[b], prev[b] = c, a
    cuts, b = [], K
    while b > 0:
        cuts.append((prev[b], b))
        b = prev[b]
    cuts.reverse()
    lens = [min(b * L, S) - a * L for a, b in cuts]
    evaluated = int(sum(M[a][a:b].sum() for a, b in cuts))
    uniform = int(sum(M[k][k] for k in range(K)))
    return lens, evaluated, uniform


def read_stop_stats(ws):
    """(synchronises) dict(active, would_skip, skipped, unshaded, colour_max) of the last dsn_render_rays on this workspace (or of a
    copy of its first 256 bytes): non-transparent samples; what DSN_STOP_STATS counted (samples early stop would leave out); what
    DSN_EARLY_STOP left out / did not shade; the largest |colour| the frame's compositor weighed."""
    buf = ws if isinstance(ws, torch.Tensor) else ws.buf
    c = buf[:256].view(torch.int32).cpu()
    return {"active": int(c[CNT_ACTIVE]), "would_skip": int(c[CNT_STOP + 2]), "skipped": int(c[CNT_STOP]), "unshaded": int(c[CNT_STOP + 1]),
            "colour_max": float(c[CNT_COLOUR_MAX:CNT_COLOUR_MAX + 1].view(torch.float32)[0])}


class GradWorkspace:
    """Scratch of the training backward (grown on demand; 22 KB per sample)."""

    def __init__(self, device):
        self.device = torch.device(device)
        self.buf = None
        self._aux = None           # (aux stream, fork event, join event) of dsn_render_rays_grad_ex, made on first use

    def aux(self):
        """the second stream + two events the backward's independent chains use (DSN_TRAIN_AUX=0: none - one stream, as rounds 1-5)"""
        if os.environ.get("DSN_TRAIN_AUX", "1") == "0":
            return (None, None, None)
        if self._aux is None:
            h = [C.c_void_p(), C.c_void_p(), C.c_void_p()]
            with torch.cuda.device(self.device):
                _check(lib().dsn_aux_create(C.byref(h[0]), C.byref(h[1]), C.byref(h[2])), "dsn_aux_create")
            self._aux = tuple(C.c_void_p(x.value) for x in h)
        return self._aux

    def __del__(self):
        a, self._aux = getattr(self, "_aux", None), None
        if a is not None:
            try:
                lib().dsn_aux_destroy(*a)
            except Exception:
                pass

    def get(self, R, S):
        need = lib().dsn_grad_workspace_bytes(int(R), int(S))
        if self.buf is None or self.buf.numel() < need:
            self.buf = None
            self.buf = _scratch(need, self.device)
        return self.buf


GRAD_CNT_RANGE = 303          # float word of the training workspace's last 4 KB (dsn_train.hip: w.small) that dsn_render_rays_grad uses as
                              # an int32 counter of samples whose tangent / adjoint pass left the fp16 range (their second-order /
                              # adjoint contributions are dropped from that step's weight gradients)


def grad_range_word(ws: "GradWorkspace", R, S):
    """1-element int32 device view of that counter for a backward over R x S samples (no synchronisation)"""
    need = lib().dsn_grad_workspace_bytes(int(R), int(S))
    off = need - 4096 + 4 * GRAD_CNT_RANGE
    return ws.buf[off:off + 4].view(torch.int32)


def grad_row_counts(ws: "GradWorkspace", R, S):
    """(synchronises) (rows the last training forward evaluated, rows its backward propagated) of R x S samples: the forward skips
    transparent samples whose noise is <= 0 (alpha = 0 exactly), the backward every row whose cotangents are all zero (dsn_train.hip,
    Rows) - two int32 words in the 256 bytes in front of the workspace's last 4 KB"""
    need = lib().dsn_grad_workspace_bytes(int(R), int(S))
    off = need - 4096 - 256
    c = ws.buf[off:off + 8].view(torch.int32).cpu()
    return int(c[0]), int(c[1])


def grad_range_overflow(ws: "GradWorkspace", R, S) -> int:
    """(synchronises) the counter's value after the last dsn_render_rays_grad on this workspace"""
    return int(grad_range_word(ws, R, S).item())


def render_rays_grad(scene: Scene, params, poses, frame_idx, zero_code, ray_o, ray_d, z_vals, noise, d_rgb, d_disp=None,
                     d_acc=None, d_depth=None, d_weights=None, ws: GradWorkspace = None, packed: PackedParams = None,
                     cached: bool = False):
    """Parameter gradients of render_rays' outputs (dsn_render_rays_grad; trainer.py:70-81 loss.backward()).
    params: the 33 tensors in PARAM_ORDER (name -> tensor dict or list).  Returns a list of 33 gradient tensors
    (float32, on the device, shaped like the parameters).  The scene's frame must be set with the same parameters."""
    dev = scene.device
    if isinstance(params, dict):
        params = [params[k] for k in PARAM_ORDER]
    prm = [_f32(p.detach(), dev) for p in params]
    if packed is None:       # MFMA image of the same parameters (the fused forward / reverse kernel streams it)
        packed = PackedParams(dev).update(dict(zip(PARAM_ORDER, prm)))
    grads = [torch.empty_like(p) for p in prm]
    pp = (C.c_void_p * 33)(*[p.data_ptr() for p in prm])
    gp = (C.c_void_p * 33)(*[g.data_ptr() for g in grads])
    R, S = z_vals.shape
    ws = ws or GradWorkspace(dev)
    buf = ws.get(R, S)
    poses = _f32(poses.reshape(24, 3), dev)
    f = lambda a: None if a is None else _f32(a, dev)
    args = [f(ray_o), f(ray_d), f(z_vals), f(noise), f(d_rgb), f(d_disp), f(d_acc), f(d_depth), f(d_weights)]
    ax = ws.aux()
    _check(lib().dsn_render_rays_grad_ex(_ptr(scene.buf), scene.V, scene.F, _ptr(packed.buf), pp, _ptr(poses), int(frame_idx), int(bool(zero_code)),
                                         _ptr(args[0]), _ptr(args[1]), _ptr(args[2]), _ptr(args[3]), int(R), int(S),
                                         _ptr(args[4]), _ptr(args[5]), _ptr(args[6]), _ptr(args[7]), _ptr(args[8]), gp,
                                         _ptr(buf), 1 if cached else 0, _stream(), ax[0], ax[1], ax[2]), "dsn_render_rays_grad")
    scene._keep_grad = (prm, args, poses, packed)
    return grads


def module_grad(scene: Scene, params, poses, frame_idx, zero_code, x_world, x_canon, view_dir, d_colour, d_sigma,
                ws: GradWorkspace = None, packed: PackedParams = None):
    """Parameter gradients of DualSpaceNeRF.forward's (colour, density) on explicit points (dsn_module_grad).  Returns the
    33 gradients in PARAM_ORDER (float32, device)."""
    dev = scene.device
    if isinstance(params, dict):
        params = [params[k] for k in PARAM_ORDER]
    prm = [_f32(p.detach(), dev) for p in params]
    if packed is None:
        packed = PackedParams(dev).update(dict(zip(PARAM_ORDER, prm)))
    grads = [torch.empty_like(p) for p in prm]
    pp = (C.c_void_p * 33)(*[p.data_ptr() for p in prm])
    gp = (C.c_void_p * 33)(*[g.data_ptr() for g in grads])
    xw, xc, vd = (_f32(t.reshape(-1, 3), dev) for t in (x_world, x_canon, view_dir))
    N = xw.shape[0]
    dc = _f32(d_colour.reshape(N, 3), dev)
    ds = _f32(d_sigma.reshape(N), dev)
    zeros = torch.zeros(N, dtype=torch.float32, device=dev)
    ws = ws or GradWorkspace(dev)
    buf = ws.get(N, 1)
    poses = _f32(poses.reshape(24, 3), dev)
    _check(lib().dsn_module_grad(_ptr(scene.buf), scene.V, scene.F, _ptr(packed.buf), pp, _ptr(poses), int(frame_idx),
                                 int(bool(zero_code)), _ptr(xw), _ptr(xc), _ptr(vd), _ptr(zeros), C.c_int64(N), _ptr(dc), _ptr(ds), gp,
                                 _ptr(buf), _stream()), "dsn_module_grad")
    scene._keep_grad = (prm, xw, xc, vd, dc, ds, zeros, poses, packed)
    return grads


def image_scatter(out: dict, mask_at_box, H, W, clamp=False):
    """post_process on the device (utils/render_utils.py:466-472): compacted per-ray outputs -> [H,W,*] images with
    zeros outside mask_at_box.  out: dict with color [R,3], disp_map, acc_map, depth_map [R] (device)."""
    dev = out["color"].device
    mask = mask_at_box.reshape(-1).to(device=dev, dtype=torch.uint8).contiguous()
    R = out["color"].shape[0]
    ws = _scratch(lib().dsn_image_workspace_bytes(H, W), dev)
    img = {k: torch.empty(H, W, c, dtype=torch.float32, device=dev) for k, c in
           (("coarse_color", 3), ("coarse_disp", 1), ("coarse_acc", 1), ("coarse_depth", 1))}
    _check(lib().dsn_image_scatter(_ptr(out["color"], torch.float32), _ptr(out["disp_map"]), _ptr(out["acc_map"]),
                                   _ptr(out["depth_map"]), R, _ptr(mask), H, W, int(bool(clamp)), _ptr(img["coarse_color"]),
                                   _ptr(img["coarse_disp"]), _ptr(img["coarse_acc"]), _ptr(img["coarse_depth"]), _ptr(ws),
                                   _stream()), "dsn_image_scatter")
    return img


def image_psnr(img_rgb, gt, mask_at_box=None):
    """metrics.py:8-21 on the device: returns a float64 device tensor {mse_all, mse_masked, psnr_all, psnr_masked}."""
    dev = img_rgb.device
    H, W = img_rgb.shape[:2]
    gt = gt.reshape(H, W, 3).to(dev).contiguous()
    assert gt.dtype in (torch.float64, torch.float32)
    mask = None if mask_at_box is None else mask_at_box.reshape(-1).to(device=dev, dtype=torch.uint8).contiguous()
    ws = _scratch(lib().dsn_image_workspace_bytes(H, W), dev)
    out = torch.empty(4, dtype=torch.float64, device=dev)
    g64, g32 = (gt, None) if gt.dtype == torch.float64 else (None, gt)
    _check(lib().dsn_image_psnr(_ptr(img_rgb.contiguous(), torch.float32), _ptr(g64), _ptr(g32), _ptr(mask), H, W, _ptr(out),
                                _ptr(ws), _stream()), "dsn_image_psnr")
    return out


def camera_rays(K, R, T, bounds, H, W, device=None, convention="zju"):
    """Whole-image rays + box near/far on the device: convention "zju" = utils/rays_utils.py:16-30, :63-97, :176-184
    (un-normalised directions, padded box), "h36m" = utils/h36m_utils.py:14-28, :61-76, :162-176 (unit directions,
    float32 slab test).
    K, R [3,3], T [3] or [3,1], bounds [2,3]: anything convertible to float64 tensors.  Returns
    (ray_o [H*W,3], ray_d [H*W,3], near [H*W], far [H*W], mask_at_box [H*W] bool), all on the device, uncompacted."""
    require_gpu()
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    f64 = lambda a: torch.as_tensor(a, dtype=torch.float64).reshape(-1).to(dev).contiguous()
    K, R, T, bounds = f64(K), f64(R), f64(T), f64(bounds)
    n = H * W
    ray_o = torch.empty(n, 3, dtype=torch.float32, device=dev)
    ray_d = torch.empty(n, 3, dtype=torch.float32, device=dev)
    near = torch.empty(n, dtype=torch.float32, device=dev)
    far = torch.empty(n, dtype=torch.float32, device=dev)
    mask = torch.empty(n, dtype=torch.uint8, device=dev)
    if convention not in ("zju", "h36m"):
        raise ValueError("convention must be 'zju' or 'h36m'")
    _check(lib().dsn_camera_rays(_ptr(K), _ptr(R), _ptr(T), _ptr(bounds), H, W, RAYS_H36M if convention == "h36m" else RAYS_ZJU,
                                 _ptr(ray_o), _ptr(ray_d), _ptr(near), _ptr(far), _ptr(mask), _stream()), "dsn_camera_rays")
    return ray_o, ray_d, near, far, mask.bool()
