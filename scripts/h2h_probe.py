"""Where does a host-batch -> host-image frame (Renderer.render_view) spend its time, and does other CPU work in the process
(the C oracle's OpenMP pool, torch's intra-op pool) change it?  python scripts/h2h_probe.py"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench, dsnerf_amd
from dsnerf_amd import synth

canon, faces = synth.make_body(); sd = synth.make_state_dict(); poses = synth.make_poses(seed=5); xyz = synth.pose_body(canon, seed=3)
rays = synth.make_rays(512, 512, xyz, fit_box=True)
args = argparse.Namespace(cpu_rays=2048)
dev = torch.device("cuda:0")
h = lambda tag: print(f"{tag:44s} host_to_host {bench.host_to_host(args, dsnerf_amd, synth, dev, canon, faces, sd, xyz, poses, rays, 512, 512, 64):6.2f} ms", flush=True)
h("fresh process")
print("   C oracle", bench.cpu_baseline(synth, canon, faces, xyz, poses, sd, rays, 64, args)["value"], "rays/s")
h("after the C oracle ran (OpenMP pool alive)")
time.sleep(2.0)
h("2 s later")
torch.set_num_threads(16)
h("torch.set_num_threads(16)")
os.environ["OMP_WAIT_POLICY"] = "passive"
