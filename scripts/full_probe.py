import sys, os, torch, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import dsnerf_amd
from dsnerf_amd import _lib, synth
dev = torch.device("cuda:0")
canon, faces = synth.make_body(); sd = synth.make_state_dict(); xyz = synth.pose_body(canon)
packed = _lib.PackedParams(dev).update({k: torch.from_numpy(v) for k, v in sd.items()})
sc = _lib.Scene(torch.from_numpy(canon), torch.from_numpy(faces), dev)
sc.set_frame(packed, torch.from_numpy(xyz), torch.from_numpy(synth.make_poses()), 5)
N = 8192 * 64
rng = np.random.default_rng(0)
x = torch.from_numpy((canon[rng.integers(0, canon.shape[0], N)] + 0.03 * rng.standard_normal((N, 3))).astype(np.float32)).to(dev)
for _ in range(3): _lib.field(sc, packed, x)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): _lib.field(sc, packed, x)
e1.record(); torch.cuda.synchronize()
print("FULL field16 on %d dense samples: %.3f ms" % (N, e0.elapsed_time(e1) / 10))
