"""Sizes of the super-cells' candidate supersets and of the cells' lists of the posed mesh's fine level (what the list build sweeps)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import dsnerf_amd
from dsnerf_amd import _lib, synth

def a256(x): return (x + 255) & ~255

def main():
    dev = torch.device("cuda:0")
    canon, faces = synth.make_body()
    sd = synth.make_state_dict()
    xyz = synth.pose_body(canon)
    packed = _lib.PackedParams(dev).update({k: torch.from_numpy(v) for k, v in sd.items()})
    scene = _lib.Scene(torch.from_numpy(canon), torch.from_numpy(faces), dev)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    scene.set_frame(packed, T(xyz), T(synth.make_poses()), 5)      # (the full build: every cell)
    torch.cuda.synchronize()
    off = scene._nn_off[0]
    hdr = scene.buf[off:off + 64].cpu()
    cell = float(hdr[12:16].view(torch.float32))
    nx, ny, nz, ncell, ok, total, cap, maxcell = [int(x) for x in hdr[20:52].view(torch.int32)]
    MAXCELL = 65536
    o_offs = off + 256
    o_u2 = o_offs + a256(4 * (MAXCELL + 1))
    o_list = o_u2 + a256(4 * MAXCELL)
    o_sc = o_list + a256(16 * cap)
    offs = scene.buf[o_offs:o_offs + 4 * (ncell + 1)].view(torch.int32).cpu().numpy().astype(np.int64)
    lens = np.diff(offs)
    sx, sy, sz = (nx + 3) // 4, (ny + 3) // 4, (nz + 3) // 4
    sc = scene.buf[o_sc:o_sc + 4 * sx * sy * sz].view(torch.int32).cpu().numpy().astype(np.int64)
    print(f"grid {nx} x {ny} x {nz} = {ncell} cells of {cell * 100:.2f} cm, ok {ok}, {total} entries (cap {cap}); {sx * sy * sz} super-cells")
    q = [0, 10, 25, 50, 75, 90, 99, 100]
    print("superset sizes, percentiles", q, np.percentile(sc, q).astype(int), "mean", sc.mean())
    print("list lengths,   percentiles", q, np.percentile(lens, q).astype(int), "mean", lens.mean())
    # per cell: the superset it sweeps
    ix = np.arange(ncell) // (nz * ny); iy = (np.arange(ncell) // nz) % ny; iz = np.arange(ncell) % nz
    sb = ((ix // 4) * sy + iy // 4) * sz + iz // 4
    n = sc[sb]
    print("superset swept per cell, percentiles", q, np.percentile(n, q).astype(int), "mean", n.mean(), " share of cells with n <= 1024:", (n <= 1024).mean(),
          "<= 2048:", (n <= 2048).mean(), " sum of n over cells", n.sum(), " sum of 64-entry chunks", ((n + 63) // 64).sum())
    print("ratio list / superset: mean", (lens / np.maximum(n, 1)).mean())

if __name__ == "__main__":
    main()
