#!/usr/bin/env python3
"""BVH build throughput: BLAS over a synthetic triangle soup (<= 21,845 triangles per geometry, many geometries) and
TLAS over N instances, for both builders."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raytracer_amd import api, _abi, host, scenes

def soup(n, seed):
    rs = np.random.RandomState(seed)
    c = rs.uniform(-1, 1, (n, 1, 3)); t = (c + rs.uniform(-0.01, 0.01, (n, 3, 3))).astype(np.float32)
    v = np.zeros(n * 3, dtype=_abi.VERTEX_DTYPE); v["position"] = t.reshape(-1, 3); v["normal"][:, 2] = 1
    return v, np.arange(n * 3, dtype=np.uint16)

for builder in ("2", "1", "0"):
    os.environ["RT_BVH_BUILDER"] = builder
    rt = api.Raytracer(0)
    rt.update_resolution(64, 64)
    v, i = soup(21845, 1)
    geo = [scenes._geo(v, i, _abi.RT_SHADER_LAMBERT, (1, 1, 1))]
    for _ in range(3):
        t0 = time.perf_counter(); h = rt.build_blas(geo); dt = time.perf_counter() - t0
    st = rt.get_stats()
    print("builder %s: BLAS 21,845 tris: %.3f ms on the stream (%.1f Mtris/s), %.2f ms wall incl. upload" % (builder, st["blas_build_ms"], 21845 / st["blas_build_ms"] / 1e3, dt * 1e3), flush=True)
    for n in (1000, 52743, 500000):
        rs = np.random.RandomState(2)
        inst = [(host.affine_transform(0.01, float(a), (float(x), float(y), float(z))), h) for a, x, y, z in rs.uniform(-1, 1, (n, 4)) * np.array([3.14, 50, 50, 1])]
        t0 = time.perf_counter(); rt.build_tlas(inst); dt = time.perf_counter() - t0
        st = rt.get_stats()
        print("builder %s: TLAS %d instances: %.3f ms on the stream (%.2f Minst/s), %.1f ms wall" % (builder, n, st["tlas_build_ms"], n / st["tlas_build_ms"] / 1e3, dt * 1e3), flush=True)
    rt.close()
