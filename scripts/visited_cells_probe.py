"""How much of the posed mesh's fine nearest-face level does a frame actually query?  Cells of the bench frame's samples against all
cells of the level, weighted by list length (the build cost follows the entries, the super-cell scan the cells)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dsnerf_amd
from dsnerf_amd import _lib, synth

def main(hw=512, S=64):
    dev = torch.device("cuda:0")
    canon, faces = synth.make_body()
    sd = synth.make_state_dict()
    xyz = synth.pose_body(canon)
    rays = synth.make_rays(hw, hw, xyz, fit_box=True)
    packed = _lib.PackedParams(dev).update({k: torch.from_numpy(v) for k, v in sd.items()})
    scene = _lib.Scene(torch.from_numpy(canon), torch.from_numpy(faces), dev)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    o, d, near, far = T(rays["ray_o"]), T(rays["ray_d"]), T(rays["near"]), T(rays["far"])
    scene.set_frame(packed, T(xyz), T(synth.make_poses()), 5)
    tv = torch.linspace(0, 1, S).to(dev)
    pts, z = _lib.sample(scene, o, d, near, far, S, tv)
    torch.cuda.synchronize()
    off = scene._nn_off[0]
    hdr = scene.buf[off:off + 64].cpu()
    lo = hdr[:12].view(torch.float32).numpy(); cell, inv = [float(x) for x in hdr[12:20].view(torch.float32)]
    nx, ny, nz, ncell, ok, total, cap, maxcell = [int(x) for x in hdr[20:52].view(torch.int32)]
    offs = scene.buf[off + 256:off + 256 + 4 * (ncell + 1)].view(torch.int32).cpu().numpy().astype(np.int64)
    lens = np.diff(offs)
    p = pts.reshape(-1, 3)
    ijk = torch.floor((p - torch.tensor(lo, device=dev)) * inv).to(torch.int64)
    inside = ((ijk >= 0) & (ijk < torch.tensor([nx, ny, nz], device=dev))).all(1)
    cid = (ijk[:, 2] * ny + ijk[:, 1]) * nx + ijk[:, 0]
    print(f"grid {nx} x {ny} x {nz} = {ncell} cells of {cell * 100:.2f} cm, ok {ok}, {total} entries (cap {cap}); samples inside the grid "
          f"{float(inside.float().mean()):.3f} of {p.shape[0]}")
    for order in ("zyx", "xyz"):
        c = cid if order == "zyx" else (ijk[:, 0] * ny + ijk[:, 1]) * nz + ijk[:, 2]
        u = torch.unique(c[inside]).cpu().numpy()
        u = u[(u >= 0) & (u < ncell)]
        print(f"  cell order {order}: visited cells {len(u)} = {len(u) / ncell:.3f} of all; their lists hold {lens[u].sum()} = {lens[u].sum() / max(total, 1):.3f} of the entries; "
              f"occupied cells (list length > 0) {int((lens > 0).sum())}")

if __name__ == "__main__":
    main()
