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


def test_plane_gemm_arithmetic_switch_and_tile_rule(hip_lib):
    """host-only: the switch between the three-term arithmetic and the f32 pipe, and what it changes in the launch plans of the
    encoder's Winograd layers at 16 pairs (Cout, tile rows, planes) -- the rule of csrc/conv.hip, shared by FlowNetHip and dim_refiner"""
    keep = hip_lib.dim_get_winograd_split()
    try:
        assert hip_lib.dim_set_winograd_split(1) == 0 and hip_lib.dim_get_winograd_split() == 1
        three = {(256, 4800, 36): 5, (512, 1280, 36): 5, (128, 19200, 36): 4,    # conv3 / conv3_1, conv4_1, conv2
                 (512, 1200, 81): 5, (512, 320, 81): 4, (512, 320, 36): 7,       # conv4, conv5, conv5_1
                 (1024, 96, 36): 7, (1024, 96, 81): 4, (64, 4800, 36): 3}        # conv6_1, conv6, a 64-channel layer (f32 pipe tile)
        for (cout, rows, planes), tile in three.items():
            assert hip_lib.dim_winograd_gemm_tile_planes(cout, rows, planes) == tile, (cout, rows, planes)
        assert hip_lib.dim_winograd_gemm_tile(512, 320) == hip_lib.dim_winograd_gemm_tile_planes(512, 320, 36)
        assert hip_lib.dim_winograd3x3s2_use(15, 20, 512, 1024) == 1      # conv6 through phase images (255 MB of weight planes)
        assert hip_lib.dim_set_winograd_split(0) == 0 and hip_lib.dim_get_winograd_split() == 0
        assert hip_lib.dim_winograd_gemm_tile_planes(512, 320, 36) == 6 and hip_lib.dim_winograd_gemm_tile_planes(1024, 96, 36) == 7
        assert hip_lib.dim_winograd_gemm_tile_planes(256, 4800, 36) == 5
        assert hip_lib.dim_winograd3x3s2_use(15, 20, 512, 1024) == 0 and hip_lib.dim_winograd3x3s2_use(30, 40, 512, 512) == 1
    finally:
        hip_lib.dim_set_winograd_split(keep)


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
