"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/deepim_hip.h declares."""
import os
import re

from conftest import ROOT


def test_header_symbols_exported(hip_lib):
    from lib.hip import capi

    header = open(os.path.join(ROOT, "include", "deepim_hip.h")).read()
    declared = set(re.findall(r"\b(dim_[A-Za-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert declared == set(capi.SIGNATURES), declared ^ set(capi.SIGNATURES)
    for name in declared:
        assert hasattr(hip_lib, name), name


def test_pure_host_queries(hip_lib):
    # no GPU needed: size queries are plain host arithmetic
    assert hip_lib.dim_conv2d_packed_weight_floats(64, 8, 7, 7) == 13 * 32 * 64  # 49 taps, 4 per chunk
    assert hip_lib.dim_conv2d_packed_weight_floats(128, 64, 5, 5) == 25 * 64 * 128
    assert hip_lib.dim_conv2d_workspace_floats(2, 8, 10, 512, 1024, 3, 3, 1, 1, 4) == 4 * 2 * 8 * 10 * 1024
    assert hip_lib.dim_raster_workspace_bytes(2, 100, 480, 640) == 2 * 480 * 640 * 8 + 2 * 100 * 12


def test_product_never_imports_oracle():
    import subprocess
    pkg = os.path.join(ROOT, "mx-deepim_amd")
    out = subprocess.run(["grep", "-rEl", r"^\s*(from|import)\s+oracle", pkg, "--include=*.py"], capture_output=True, text=True)
    assert out.stdout.strip() == "", out.stdout


def test_no_cpu_fallback_message():
    import pytest
    from lib.hip import capi
    import torch

    with pytest.raises(capi.DeepIMHipError):
        capi.dptr(torch.zeros(3))
