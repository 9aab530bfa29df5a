"""hipGraph capture of one whole frame (set_frame + render_rays) at the small interactive size (128 x 128 x 32,
BASELINE configs[0]) against plain stream launches: the C ABI never allocates or synchronises, so the ~30 launches of a
frame are capturable as they are."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dsnerf_amd
from dsnerf_amd import _lib, synth

def main(hw=128, S=32):
    dev = torch.device("cuda:0")
    canon, faces = synth.make_body()
    sd = synth.make_state_dict()
    xyz = synth.pose_body(canon)
    rays = synth.make_rays(hw, hw, xyz, fit_box=True)
    packed = _lib.PackedParams(dev).update({k: torch.from_numpy(v) for k, v in sd.items()})
    scene = _lib.Scene(torch.from_numpy(canon), torch.from_numpy(faces), dev)
    ws = _lib.RenderWorkspace(dev)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    o, d, n0, f0 = T(rays["ray_o"]), T(rays["ray_d"]), T(rays["near"]), T(rays["far"])
    near, far = n0.clone(), f0.clone()
    tv = torch.linspace(0, 1, S).to(dev)
    dxyz, dposes = T(xyz), T(synth.make_poses())
    out = None
    def frame():
        nonlocal out
        near.copy_(n0); far.copy_(f0)
        scene.set_frame(packed, dxyz, dposes, 5)
        out = _lib.render_rays(scene, packed, ws, o, d, near, far, S, tv, want_weights=False, out=out)
    def timed(fn, reps=50):
        for _ in range(5): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): fn()
        torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / reps
    t_eager = timed(frame)
    ref = {k: v.clone() for k, v in out.items()}
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        frame()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        frame()
    t_graph = timed(g.replay)
    same = all(torch.equal(out[k], ref[k]) or k == "disp_map" for k in ref)
    print(f"{hw}x{hw}x{S}: stream launches {t_eager:.3f} ms/frame, graph replay {t_graph:.3f} ms/frame, identical outputs: {same}")

if __name__ == "__main__":
    main()
    main(512, 64)
