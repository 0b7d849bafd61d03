#!/usr/bin/env python3
"""A/B of traversal settings in ONE process (interleaved rounds, cdna guide rule 24).
usage: python tools/ab_traverse.py [specs...]   spec = RT_BVH_BUILDER[:RT_SAH_TRI], e.g. 0  1  1:1.0  1:1.6"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raytracer_amd import api, scenes, distributed, _abi

variants = sys.argv[1:] or ["1", "2"]
workloads = {"c2": lambda: scenes.bunny(1920, 1080, spp=1, bounces=1),
             "c3": lambda: scenes.cornell(1920, 1080, spp=4, bounces=4)}
rounds, steps = 5, 6
for wname, mk in workloads.items():
    ctxs = {}
    for v in variants:
        f = v.split(":")
        os.environ["RT_BVH_BUILDER"] = f[0]
        os.environ["RT_SAH_TRI"] = f[1] if len(f) > 1 else "1.3"
        rt = api.Raytracer(0); sc = mk(); sc.setup(rt)
        rt.set_flags(sc.flags | _abi.RT_FLAG_TIME_KERNELS)
        ctxs[v] = (rt, sc, distributed.FrameChain(1), [0])
    res = {v: {"wall": [], "trav": []} for v in variants}
    for r in range(rounds + 1):
        for v in variants:
            rt, sc, chain, k = ctxs[v]
            rt.reset_stats(); rt.sync()
            t0 = time.perf_counter()
            for _ in range(steps):
                distributed.render_frame(rt, sc.globals, k[0], chain); k[0] += 1
            rt.sync(); dt = time.perf_counter() - t0
            st = rt.get_stats()
            if r > 0:
                rays = st["rays_camera"]
                res[v]["wall"].append(rays / dt / 1e6); res[v]["trav"].append(rays / (st["traverse_ms"] * 1e-3) / 1e6)
    for v in variants:
        print("%s variant %s: wall Mrays/s median %.0f min %.0f | traverse-only Mrays/s median %.0f min %.0f" % (
            wname, v, np.median(res[v]["wall"]), np.min(res[v]["wall"]), np.median(res[v]["trav"]), np.min(res[v]["trav"])), flush=True)
    for v in variants:
        ctxs[v][0].close()
