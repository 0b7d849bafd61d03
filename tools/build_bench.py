#!/usr/bin/env python3
"""BVH build throughput (outside the timed loop of bench.py; SURVEY.md 8d reports it separately).
  BLAS: a synthetic soup of 21,845 triangles (the most a 16-bit-indexed geometry without shared vertices holds).
  TLAS: N instances of the bunny (2,844 vertices) on C5's jittered lattice, N = 1,000 / 52,743 / 500,000, with the
        instance boxes taken from every vertex of the instance (`exact`, the default: one wave per instance, N x 2,844
        vertex transforms) and from the 8 corners of the BLAS box (RT_EXACT_INSTANCE_BOXES=0).
Round 1's record used a 65,535-vertex BLAS for the instances: at 52,743 instances the exact boxes were 3.4 G vertex
transforms (most of the build), at 500,000 the library's 4 G-vertex cap switched them off -- which is why 500 k
instances 'built faster' than 52 k there."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raytracer_amd import api, _abi, scenes

def soup(n, seed):
    rs = np.random.RandomState(seed)
    c = rs.uniform(-1, 1, (n, 1, 3)); t = (c + rs.uniform(-0.01, 0.01, (n, 3, 3))).astype(np.float32)
    v = np.zeros(n * 3, dtype=_abi.VERTEX_DTYPE); v["position"] = t.reshape(-1, 3); v["normal"][:, 2] = 1
    return v, np.arange(n * 3, dtype=np.uint16)

names = {"0": "LBVH", "1": "PLOC", "2": "binned SAH (PLOC above 262,144 primitives)", "4": "cheaper of PLOC and SAH (default)"}
fields = {n: scenes.bunny_field(64, 64, grid=int(np.ceil(np.sqrt(n))), count=n) for n in (1000, 52743, 500000)}
for builder in ("0", "1", "2", "4"):
    for exact in ("1", "0"):
        os.environ["RT_BVH_BUILDER"] = builder
        os.environ["RT_EXACT_INSTANCE_BOXES"] = exact
        rt = api.Raytracer(0)
        rt.update_resolution(64, 64)
        if exact == "1":
            v, i = soup(21845, 1)
            geo = [scenes._geo(v, i, _abi.RT_SHADER_LAMBERT, (1, 1, 1))]
            for _ in range(3):
                t0 = time.perf_counter(); rt.build_blas(geo); dt = time.perf_counter() - t0
            st = rt.get_stats()
            print("builder %s (%s): BLAS 21,845 tris: %.3f ms on the stream (%.1f Mtris/s), %.2f ms wall incl. upload" % (
                builder, names[builder], st["blas_build_ms"], 21845 / st["blas_build_ms"] / 1e3, dt * 1e3), flush=True)
        for n, scene in sorted(fields.items()):
            handles = [rt.build_blas(g) for g in scene.blas]
            inst = [(m, handles[b]) for m, b in scene.instances]
            best = None
            for _ in range(2):
                t0 = time.perf_counter(); rt.build_tlas(inst); dt = time.perf_counter() - t0
                ms = rt.get_stats()["tlas_build_ms"]
                best = ms if best is None else min(best, ms)
            print("builder %s: TLAS %7d bunny instances, %s boxes: %.3f ms on the stream (%.1f Minst/s), %.1f ms wall with the host-side instance records" % (
                builder, len(inst), "exact" if exact == "1" else "corner", best, len(inst) / best / 1e3, dt * 1e3), flush=True)
        rt.close()
