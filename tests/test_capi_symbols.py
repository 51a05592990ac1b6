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
    assert hip_lib.dim_conv2d_packed_weight_floats(64, 8, 7, 7) == 13 * 32 * 64 + 153600 // 4  # 49 taps, 4 per chunk; + flow_conv1's three-term image
    assert hip_lib.dim_conv2d_packed_weight_floats(128, 64, 5, 5) == 25 * 64 * 128
    assert hip_lib.dim_conv2d_workspace_floats(2, 8, 10, 512, 1024, 3, 3, 1, 1, 4) == 4 * 2 * 8 * 10 * 1024
    # z-buffer (8 B / pixel) + projected vertices (padded to 256 B) + 256-byte header + covered-pixel list (4 B / pixel)
    assert hip_lib.dim_raster_workspace_bytes(2, 100, 480, 640) == 2 * 480 * 640 * 8 + 2560 + 256 + 2 * 480 * 640 * 4 + 2 * 1200 * 16
    # Winograd paths: packed weights = planes x K x Cout, workspace = planes x tiles x (K + Cout)
    # f32 image + the three-term bf16 image behind it (6 bytes per weight): x 5 / 2
    assert hip_lib.dim_winograd_packed_weight_floats(256, 128, 4) == 36 * 256 * 128 * 5 // 2
    assert hip_lib.dim_winograd_packed_weight_floats(256, 128, 2) == 16 * 256 * 128 * 5 // 2
    assert hip_lib.dim_winograd_workspace_floats(16, 60, 80, 256, 256, 4) == 36 * (16 * 15 * 20) * 512
    assert hip_lib.dim_winograd_workspace_floats(16, 60, 80, 256, 256, 2) == 16 * (16 * 30 * 40) * 512
    assert hip_lib.dim_winograd_workspace_floats(1, 7, 9, 32, 64, 3) == 0  # unsupported tile size
    assert hip_lib.dim_winograd5x5s2_packed_weight_floats(128, 64) == 36 * 128 * 256 * 5 // 2
    assert hip_lib.dim_winograd5x5s2_workspace_floats(16, 240, 320, 64, 128) == 36 * (16 * 30 * 40) * (256 + 128)
    # batches whose transformed tiles would overflow 32-bit byte offsets run in slices: the workspace stops growing with N
    big = hip_lib.dim_winograd5x5s2_workspace_floats(512, 240, 320, 64, 128)
    assert big == hip_lib.dim_winograd5x5s2_workspace_floats(1024, 240, 320, 64, 128)
    assert big * 4 * 256 // (256 + 128) < (1 << 32)  # V of one slice stays below 4 GiB
    assert hip_lib.dim_conv2d_wgrad_winograd_workspace_floats(16, 60, 80, 256, 256, 1, 4) == 36 * 4800 * 512 + 36 * 256 * 256 * 5
    assert hip_lib.dim_fc_fwd_workspace_floats(1024, 8, 10, 256) == 256 * 32 * 256


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
