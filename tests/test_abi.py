"""The C-ABI libraries load on a CPU-only machine and export every symbol their headers declare (no compute calls)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", text)))


@pytest.mark.parametrize("header,lib", [("rt_abi.h", "raytracer_amd/librt_mi355x.so"), ("rt_host.h", "raytracer_amd/librt_host.so")])
def test_every_declared_symbol_is_exported(header, lib):
    names = declared_functions(header)
    assert len(names) >= 15
    so = ctypes.CDLL(os.path.join(ROOT, lib))
    missing = [n for n in names if not hasattr(so, n)]
    assert not missing, "declared in include/%s but not exported by %s: %s" % (header, lib, missing)


def test_abi_version_and_struct_sizes():
    from raytracer_amd import _abi
    so = ctypes.CDLL(os.path.join(ROOT, "raytracer_amd/librt_mi355x.so"))
    so.rt_abi_version.restype = ctypes.c_int
    assert so.rt_abi_version() == 3 == _abi.RT_ABI_VERSION
    header = open(os.path.join(ROOT, "include", "rt_abi.h")).read()
    assert "#define RT_ABI_VERSION 3" in header
    assert ctypes.sizeof(_abi.RtGlobals) == 156                # prelude.h:110-139
    offs = {n: getattr(_abi.RtGlobals, n).offset for n, _ in _abi.RtGlobals._fields_}
    assert offs["camera_to_world"] == 16 and offs["camera_aspect"] == 80 and offs["translucent_accumulator_count"] == 88
    assert offs["translucent_emission_bounces"] == 96 and offs["translucent_scattering"] == 112
    assert offs["translucent_absorption"] == 128 and offs["light_color"] == 144
    assert _abi.VERTEX_DTYPE.itemsize == 24 and _abi.SAMPLE_POINT_DTYPE.itemsize == 24 and _abi.HIT_DTYPE.itemsize == 32
    assert ctypes.sizeof(_abi.RtGeometry) == 48 and ctypes.sizeof(_abi.RtInstance) == 68


def test_product_does_not_reference_the_oracle():
    """the product path must never route through oracle/ (test infrastructure)"""
    for base, _, files in os.walk(os.path.join(ROOT, "raytracer_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                text = open(os.path.join(base, f), errors="ignore").read()
                assert not re.search(r"import\s+oracle|from\s+oracle|liboracle|\borc_[a-z]|oracle/", text), f


def test_rt_create_fails_loudly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from raytracer_amd import api
    with pytest.raises(RuntimeError):
        api.Raytracer(0)


def test_committed_counter_passes_belong_to_the_committed_kernel_sources():
    """bench.py prices its roofline with the PMC passes in profiles/pmc_traffic.json and drops them (roofline.frac = null,
    pmc_note says why) when they were taken on other kernel sources: after a change under raytracer_amd/csrc, include/rt_abi.h
    or the Makefile, rerun tools/final_measure.sh pmc (GPU) and tools/pmc_merge.py before committing."""
    import json
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import pmc_summary
    want = pmc_summary.source_hash(root)
    entries = json.load(open(os.path.join(root, "profiles", "pmc_traffic.json")))
    for key in ("c2", "c3", "c4", "c5", "c5_3840x2160"):
        assert entries[key]["source_hash"] == want, "profiles/pmc_traffic.json[%r] is stale (%s, sources are %s)" % (key, entries[key]["source_hash"], want)


def test_frame_instruction_count_from_a_counter_pass():
    """bench.py's whole-frame issue figure: the vector instructions of one pipelined frame are every render kernel's mean
    times its launches over the frames of the pass; set-up kernels and the one counting frame are left out"""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    k = lambda n, insts: {"launches_sampled": n, "SQ_INSTS_VALU": insts}
    ks = {"k_frame_resolve": k(10, 2.0), "k_shade": k(33, 1.0), "k_shade<true, true, 3>": k(10, 30.0), "k_shade<true, true, 0>": k(1, 9.0),
          "k_shade<false, true, 0>": k(2, 7.0), "k_traverse": k(13, 1.0), "k_traverse<false, 8, false, false, false, 3>": k(10, 50.0),
          "k_traverse<true, 1, true, false, false, 3>": k(1, 500.0), "k_traverse<true, 1, false, false, false, 3>": k(2, 600.0),
          "k_sah_bin": k(40, 1000.0), "k_morton": k(3, 1000.0)}
    assert bench.pmc_frame_insts(ks) == pytest.approx(82.0)          # 30 + 50 + 2: the counting frame's five launches and the build are out
    # a frame that gathers: the counting frame is unfused (k_camera_resolve) and takes one share of the once-per-frame kernels
    ks = {"k_frame_resolve": k(4, 1.0), "k_camera_resolve": k(1, 5.0), "k_gather": k(5, 1.0), "k_gather<1, 256>": k(5, 100.0), "k_probe_raygen": k(10, 3.0),
          "k_shade<false, false, 0>": k(4 * 3 + 2, 10.0), "k_shade<true, false, 0>": k(1, 11.0), "k_shade<true, true, 0>": k(4, 20.0),
          "k_traverse<false, 7, false, false, false, 2>": k(12, 40.0), "k_traverse<true, 1, false, false, false, 2>": k(2, 1.0),
          "k_traverse<true, 1, true, false, false, 2>": k(1, 1.0), "k_bn_try": k(99, 1.0)}
    assert bench.pmc_frame_insts(ks) == pytest.approx(1.0 + 100.0 + 2 * 3.0 + 3 * 10.0 + 20.0 + 3 * 40.0)
    assert bench.pmc_frame_insts({"k_shade": k(3, 1.0)}) is None     # no pipelined frame in the pass
