"""The C-ABI library loads and exports every symbol include/rtgpu.h declares; host-only entry points work
without a GPU and the device entry points fail loudly (no CPU fallback).  CPU only."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import raytracer_amd as ra
from raytracer_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "rtgpu.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound():
    syms = header_symbols()
    assert len(syms) >= 25
    raw = C.CDLL(L.LIB_PATH)
    for s in syms:
        assert hasattr(raw, s), f"{s} declared in rtgpu.h but not exported by librtgpu.so"
        assert s in L._PROTOS, f"{s} has no ctypes prototype"
    assert sorted(L._PROTOS) == syms


def test_struct_layouts_match_header_sizes():
    # sizes computed from the header's field lists (natural alignment)
    assert C.sizeof(L.RenderParams) == 12 * 8 + 7 * 4 + 4 + 8
    assert C.sizeof(L.SceneInfo) == 4 + 4 + 8 * 3 + 4 + 4 + 8 + 8
    assert C.sizeof(L.Stats) == 3 * 8 + 5 * 8


def test_version_and_errors():
    assert b"gfx950" in L.lib.rt_version()
    assert L.lib.rt_scene_info(None, None) != 0
    assert b"null" in L.lib.rt_last_error()


@pytest.mark.skipif(L.lib.rt_device_count() > 0, reason="a GPU is present")
def test_device_entry_points_fail_loudly_without_gpu():
    assert L.lib.rt_create(0) is None
    assert b"no HIP device" in L.lib.rt_last_error()
    with pytest.raises(ra.RtError):
        ra.Context(0)


def test_write_ppm_roundtrip(tmp_path):
    rgb = (np.arange(5 * 7 * 3) % 256).astype(np.uint8).reshape(5, 7, 3)
    path = tmp_path / "x.ppm"
    ra.write_ppm(path, rgb)
    data = path.read_bytes()
    assert data.startswith(b"P6\n7 5\n255\n")  # cpp/rtrender.cpp:77
    assert data[len(b"P6\n7 5\n255\n"):] == rgb.tobytes()
    with pytest.raises(ra.RtError):
        ra.write_ppm(tmp_path / "no_such_dir" / "x.ppm", rgb)


def test_header_is_plain_c99_and_links_against_the_library(tmp_path):
    """The boundary is a C ABI: include/rtgpu.h must compile as strict C99 (no C++, no torch types) and a C
    program using it must link against librtgpu.so and run its host-only entry points."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "use_abi.c"
    src.write_text(
        '#include <stdio.h>\n#include "rtgpu.h"\n'
        "int main(void) {\n"
        "  int r0 = -1, r1 = -1; uint64_t pos = 0;\n"
        "  if (rt_shard_rows(4096, 4096, 64, 3, 8, 0, &r0, &r1, &pos) != 0) return 1;\n"
        '  printf("%s %d %d %llu\\n", rt_version(), r0, r1, (unsigned long long)pos);\n'
        "  return (r0 == 1536 && r1 == 2048) ? 0 : 2;\n}\n")
    exe = tmp_path / "use_abi"
    libdir = os.path.join(root, "raytracer_amd")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(root, "include"),
                           str(src), "-o", str(exe), "-L", libdir, "-lrtgpu", "-Wl,-rpath," + libdir])
    out = subprocess.check_output([str(exe)]).decode()
    assert out.startswith("rtgpu") and " 1536 2048 " in out
