#!/usr/bin/env python3
"""Pooled traversal (RT_POOL) against the per-lane walk on growing instance fields, image against image, one size at a
time; stops at the first difference.  Run with the bounds-checking build: RT_LIB_PATH=raytracer_amd/librt_mi355x_dbg.so"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raytracer_amd import api, scenes, distributed

def render(pool, scene, frames):
    os.environ["RT_POOL"] = pool
    rt = api.Raytracer(0)
    try:
        scene.setup(rt)
        for f in range(frames):
            distributed.render_frame(rt, scene.globals, f, 7)
        rt.sync()
        st = rt.get_stats()
        return rt.read_accumulator(), st
    finally:
        rt.close()

cases = [(640, 360, 24, 500, 3), (1920, 1080, 48, 2048, 3), (1920, 1080, 230, 52743, 3), (3840, 2160, 230, 52743, 3)]
for w, h, grid, count, bounces in cases[:int(sys.argv[1]) if len(sys.argv) > 1 else len(cases)]:
    sc = scenes.bunny_field(w, h, spp=1, bounces=bounces, grid=grid, count=count)
    ref, st0 = render("0", sc, 3)
    got, st1 = render("2", sc, 3)
    same = np.array_equal(ref.view(np.uint32), got.view(np.uint32))
    print("%dx%d %d instances: rays %d / %d, depth %d+%d stack_in_lds %d, images %s" % (
        w, h, count, st0["rays_camera"], st1["rays_camera"], st1["tlas_depth"], st1["max_blas_depth"], st1["stack_in_lds"],
        "identical" if same else "DIFFER at %d pixels" % int((ref != got).any(axis=2).sum())), flush=True)
    if not same or st0["rays_camera"] != st1["rays_camera"]:
        sys.exit(1)
print("ok")
