"""Source-level guard of THE RULE of csrc/dsn_common.h (round 6): a kernel that issues v_mfma_f32_32x32x16_f16 - gfx950's K = 16 form
of the f16 MFMA - makes co-resident waves of OTHER kernels consume their vector-memory loads before they have landed (DESIGN 4.5;
found by exchanging exactly this instruction for two K = 8 ones in an otherwise identical kernel, profiles/r06_coresidency_bisect.txt),
so it must take its SIMD's whole register file (DSN_OWN_SIMD / DSN_OWN_SIMD_T with a bit of DSN_TRAIN_AGGRESSORS).  Kernels whose MFMAs
are all of another kind (fp32 32x32x2, f16 32x32x8) are exempt by the rule and by measurement (12 repetitions beside 14 frames' shading
and geometry phases: 0 differing samples).  A new kernel on the K = 16 instruction fails this test until it is guarded."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "dual-space-nerf_amd", "csrc")
X16 = re.compile(r"mfma_f32_32x32x16|MFMA16\(|dense16x?<|dense16s<")              # the K = 16 instruction and the pipelines built on it
ANY_MFMA = re.compile(r"__builtin_amdgcn_mfma|MFMA16\(|DSN_MFMA\(|dense16x?<|dense16s<|v_mfma")
GUARD = re.compile(r"DSN_OWN_SIMD\(\)|DSN_OWN_SIMD_T\((\d+)\)|v_mov_b32 v255")
# matrix kernels WITHOUT the K = 16 instruction (what they issue instead): exempt - listed so that the scan's findings are explicit
OTHER_MFMA = {"k_t_lin": "fp32 32x32x2", "k_t_wgrad": "fp32 32x32x2", "k_t_wgrad16p": "f16 32x32x8 (moved off the K = 16 form in round 6)",
              "k_field": "fp32 32x32x2 (guarded anyway: 480 registers)", "k_light": "fp32 32x32x2 (guarded anyway: the fallback path)"}


def _body(src, start):
    i = src.index("{", start)
    depth, j = 0, i
    while True:
        depth += src[j] == "{"
        depth -= src[j] == "}"
        j += 1
        if depth == 0:
            return src[i:j]


def kernels(path):
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"//[^\n]*", " ", src)                 # (comments name the instruction too)
    for m in re.finditer(r"__global__\s+void[^{;]*?\b(k_\w+)\s*\(", src):
        yield m.group(1), _body(src, m.end()), src


def reaches(body, src, pattern, seen=None):
    """does the kernel body, or a device function of this file it calls, match `pattern`?"""
    if pattern.search(body):
        return True
    seen = seen if seen is not None else set()
    for fn in set(re.findall(r"\b([a-z_][a-z0-9_]*)\s*(?:<[^;{}()]*>)?\(", body)):
        if fn in seen:
            continue
        seen.add(fn)
        m = re.search(r"__device__[^;{]*?\b" + re.escape(fn) + r"\s*\([^;{]*?\)\s*\{", src)
        if m and reaches(_body(src, m.end() - 1), src, pattern, seen):
            return True
    return False


def test_every_kernel_on_the_k16_mfma_owns_its_simd():
    aggressor_bits = int(re.search(r"#define DSN_TRAIN_AGGRESSORS (\d+)", open(os.path.join(CSRC, "dsn_common.h")).read()).group(1))
    x16, other = {}, {}
    for f in sorted(os.listdir(CSRC)):
        if not f.endswith(".hip"):
            continue
        for name, body, src in kernels(os.path.join(CSRC, f)):
            if not reaches(body, src, ANY_MFMA):
                continue
            g = GUARD.search(body)
            guarded = bool(g) and (g.group(1) is None or (int(g.group(1)) & aggressor_bits) != 0)
            if reaches(body, src, X16):
                x16[name] = guarded
                assert guarded, f"{f}: {name} issues the K = 16 f16 MFMA and does not own its SIMD (DSN_OWN_SIMD): see csrc/dsn_common.h"
            else:
                other[name] = guarded
                assert name in OTHER_MFMA, f"{f}: matrix kernel {name} is not on the K = 16 instruction and not listed: say what it issues"
    # the scan found the family it is about (it did not silently match nothing)
    assert set(x16) >= {"k_field16", "k_light16", "k_tangent16", "k_adjoint16", "k_screen16", "k_t_wgrad16d", "k_t_wgrad16q"}, x16
    assert set(other) == set(OTHER_MFMA), (other, OTHER_MFMA)
