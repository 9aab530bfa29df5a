"""C-ABI checks that need no GPU: the library loads, exports every symbol include/dsnerf.h declares,
its size helpers are sane and argument errors are reported (never silently ignored)."""
import ctypes as C
import os
import re

import pytest

from helpers import ROOT


@pytest.fixture(scope="module")
def lib():
    import dsnerf_amd
    return dsnerf_amd._lib.lib()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "dsnerf.h")).read()
    return sorted(set(re.findall(r"DSN_EXPORT\s+[\w\s\*]+?\b(dsn_\w+)\s*\(", text)))


def test_header_symbols_exported(lib):
    syms = declared_symbols()
    assert len(syms) >= 16
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/dsnerf.h but not exported"
    import dsnerf_amd
    assert set(dsnerf_amd._lib.EXPORTS) == set(syms)


def test_version_and_sizes(lib):
    assert lib.dsn_abi_version() == 1
    assert lib.dsn_packed_param_bytes() > 3_000_000            # fwd + transposed images of ~0.5 M params
    assert lib.dsn_scene_bytes(6890, 13776) > 13776 * (64 * 2 + 16 * 2)
    assert lib.dsn_scene_bytes(0, 0) == 0
    n = lib.dsn_render_workspace_bytes(1024, 64)
    assert n >= 1024 * 64 * (4 + 1 + 4 + 12 * 5 + 4)
    assert lib.dsn_render_workspace_bytes(0, 64) == 0


def test_errors_are_loud(lib):
    assert lib.dsn_pack_params(None, None, None) != 0
    assert b"dsn_pack_params" in lib.dsn_last_error()
    assert lib.dsn_composite(None, None, None, None, None, None, 4, 4, None, None, None, None, None, None) != 0
    assert lib.dsn_render_rays(None, 1, 1, None, None, None, None, None, 0, 0, None, None, None, 0, None, None, None,
                               None, None, None, None, None) != 0
    assert b"dsn_render_rays" in lib.dsn_last_error()


def test_no_fallback_without_gpu():
    import torch
    import dsnerf_amd
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        dsnerf_amd._lib.PackedParams("cuda")


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "dual-space-nerf_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dp, f)).read()
                assert "liboracle" not in src and "import oracle" not in src and "orc_" not in src, f
