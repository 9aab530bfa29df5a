"""Shared helpers for the test-suite (test infrastructure only)."""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
CASES = ["small_eval", "small_train", "small_novel", "small_rot", "full_eval"]


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def state():
    import dsnerf_amd.synth as synth
    return synth.make_state_dict()


def code_for(g, sd, name):
    c = sd["nerf.embedding.weight"][int(g["frame"])]
    return c * 0 if name == "small_novel" else c   # test.py:193-196 sets net.nerf.w = 0


def light_kw(g):
    kw = {}
    if "light_center" in g.files:
        kw["light_shift"] = g["light_center"] - g["Th"]
    if "rot" in g.files:
        kw["rot"] = g["rot"]
        kw["rot_center"] = g["rot_center"][0, :2]
    return kw


def per_point_dirs(g):
    S = int(g["S"])
    return np.repeat(g["ray_d"][:, None, :], S, 1).reshape(-1, 3)


def maxdiff(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.nanmax(np.abs(a - b))) if a.size else 0.0
