"""Source-level guard: every kernel built on the split-fp16 weight-ring pipeline of dsn_field16.hip (dense16 /
dense16x / the screen's dense loop) - the family whose waves made co-resident waves of OTHER kernels consume registers before their
loads had landed - must take its SIMD's whole register file (DSN_OWN_SIMD / DSN_OWN_SIMD_T with an aggressor bit).  Any
other kernel that issues MFMAs must either carry the guard or be on the list of kernels the round-6 bisect cleared
(scripts/dbg/race_bisect.sh, profiles/r06_coresidency_bisect.txt: 0 differing samples unguarded).  A new matrix kernel therefore
fails this test until it is guarded or measured."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "dual-space-nerf_amd", "csrc")
# cleared by measurement (unguarded: 0 differing samples beside 14 frames' shading / geometry phases, 3 repetitions each)
CLEARED = {"k_t_wgrad16d": "256 + 256 registers: nothing fits beside it; bisect tu4", "k_t_wgrad16p": "bisect tu8", "k_t_lin": "bisect tu16",
           "k_t_wgrad": "bisect tu32", "k_t_wgrad16c": "round 3's kernel, DSN_WGRAD16=c only: 256 + 256 registers"}
MFMA = re.compile(r"__builtin_amdgcn_mfma|MFMA16\(|DSN_MFMA\(|dense16x?<|dense16s<|v_mfma")
GUARD = re.compile(r"DSN_OWN_SIMD\(\)|DSN_OWN_SIMD_T\((\d+)\)|v_mov_b32 v255")


def kernels(path):
    src = open(path).read()
    for m in re.finditer(r"__global__\s+void[^{;]*?\b(k_\w+)\s*\(", src):
        name = m.group(1)
        i = src.index("{", m.end())
        depth, j = 0, i
        while True:
            c = src[j]
            depth += c == "{"
            depth -= c == "}"
            j += 1
            if depth == 0:
                break
        yield name, src[i:j], src


def calls_matrix_pipeline(body, src, seen=None):
    """does the kernel body (or a device function of this file it calls) issue MFMAs?"""
    if MFMA.search(body):
        return True
    seen = seen if seen is not None else set()
    for fn in set(re.findall(r"\b([a-z_][a-z0-9_]*)\s*(?:<[^;{}()]*>)?\(", body)):
        if fn in seen:
            continue
        seen.add(fn)
        m = re.search(r"__device__[^;{]*?\b" + re.escape(fn) + r"\s*\([^;{]*?\)\s*\{", src)
        if not m:
            continue
        i = m.end() - 1
        depth, j = 0, i
        while True:
            depth += src[j] == "{"
            depth -= src[j] == "}"
            j += 1
            if depth == 0:
                break
        if calls_matrix_pipeline(src[i:j], src, seen):
            return True
    return False


def test_every_matrix_kernel_is_guarded_or_cleared():
    aggressor_bits = int(re.search(r"#define DSN_TRAIN_AGGRESSORS (\d+)", open(os.path.join(CSRC, "dsn_common.h")).read()).group(1))
    found = {}
    for f in sorted(os.listdir(CSRC)):
        if not f.endswith(".hip"):
            continue
        for name, body, src in kernels(os.path.join(CSRC, f)):
            if not calls_matrix_pipeline(body, src):
                continue
            g = GUARD.search(body)
            guarded = bool(g) and (g.group(1) is None or (int(g.group(1)) & aggressor_bits) != 0)
            found[name] = guarded
            assert guarded or name in CLEARED, (f"{f}: matrix kernel {name} neither takes its SIMD's register file (DSN_OWN_SIMD) nor is it on "
                                                f"the list of kernels cleared by scripts/dbg/race_bisect.sh")
    # the known family is all there (the scan did not silently find nothing)
    for k in ("k_field16", "k_field", "k_light16", "k_tangent16", "k_adjoint16", "k_screen16"):
        assert found.get(k) is True, (k, found)
    for k in ("k_t_wgrad16d", "k_t_wgrad16p", "k_t_lin", "k_t_wgrad"):
        assert k in found, (k, found)
