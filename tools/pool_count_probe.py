#!/usr/bin/env python3
"""The counting variant of the pooled traversal (RT_FLAG_COUNT_TRAVERSAL) on growing instance fields, under the bounds-checking
build: RT_LIB_PATH=raytracer_amd/librt_mi355x_dbg.so"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import api, scenes, distributed, _abi
cases = [(160, 90, 20, 300, 3), (640, 360, 24, 500, 4), (1920, 1080, 48, 2048, 4), (1920, 1080, 230, 52743, 4), (3840, 2160, 230, 52743, 4)]
for w, h, grid, count, bounces in cases[:int(sys.argv[1]) if len(sys.argv) > 1 else len(cases)]:
    sc = scenes.bunny_field(w, h, spp=1, bounces=bounces, grid=grid, count=count)
    rt = api.Raytracer(0)
    sc.setup(rt, _abi.RT_FLAG_COUNT_TRAVERSAL)
    rt.reset_stats()
    for f in range(2):
        distributed.render_frame(rt, sc.globals, f, 5)
    s = rt.get_stats(); rays = s["rays_camera"]
    print("%dx%d %d: rays %d nodes/ray %.2f tris %.2f insts %.2f | wave steps per 64 rays: node %.1f tri %.1f inst %.1f | SIMD eff node %.2f tri %.2f inst %.2f" % (
        w, h, count, rays, s["nodes_visited"] / rays, s["tris_tested"] / rays, s["instances_entered"] / rays,
        s["wave_node_steps"] / (rays / 64), s["wave_tri_steps"] / (rays / 64), s["wave_instance_steps"] / (rays / 64),
        s["nodes_visited"] / max(1, 64 * s["wave_node_steps"]), s["tris_tested"] / max(1, 64 * s["wave_tri_steps"]),
        s["instances_entered"] / max(1, 64 * s["wave_instance_steps"])), flush=True)
    rt.close()
