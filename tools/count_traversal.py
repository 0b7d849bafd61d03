#!/usr/bin/env python3
"""Per-ray traversal counters and SIMD efficiency of k_traverse for the benchmark workloads."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import api, scenes, distributed, _abi
print('RT_SAH_TRI', os.environ.get('RT_SAH_TRI'), 'RT_BVH_BUILDER', os.environ.get('RT_BVH_BUILDER'))
ALL = {"c2": lambda: scenes.bunny(1920, 1080, 1, 1), "c3": lambda: scenes.cornell(1920, 1080, 2, 4),
       "c4": lambda: scenes.cornell_bunny(1920, 1080, 1, 4), "c5": lambda: scenes.bunny_field(1920, 1080, 1, 4),
       # the primary launches alone (no bounce), and C5 at its benchmark size
       "c2p": lambda: scenes.bunny(1920, 1080, 1, 0), "c5p": lambda: scenes.bunny_field(1920, 1080, 1, 0),
       "c5p4k": lambda: scenes.bunny_field(3840, 2160, 1, 0), "c5_4k": lambda: scenes.bunny_field(3840, 2160, 1, 4)}
for name in (sys.argv[1:] or ["c2", "c3", "c4"]):
    sc = ALL[name]()
    rt = api.Raytracer(0); sc.setup(rt, _abi.RT_FLAG_COUNT_TRAVERSAL)
    rt.reset_stats()
    distributed.render_frame(rt, sc.globals, 0, 5); distributed.render_frame(rt, sc.globals, 1, 5)
    s = rt.get_stats(); rays = s["rays_camera"] + s["rays_probe"]
    print("%s rays %d | per ray: nodes %.2f tris %.2f insts %.2f | per wave-of-64-rays steps: node %.1f tri %.1f inst %.1f | SIMD eff node %.2f tri %.2f inst %.2f | max steps of one ray %d" % (
        name, rays, s["nodes_visited"] / rays, s["tris_tested"] / rays, s["instances_entered"] / rays,
        s["wave_node_steps"] / (rays / 64), s["wave_tri_steps"] / (rays / 64), s["wave_instance_steps"] / (rays / 64),
        s["nodes_visited"] / max(1, 64 * s["wave_node_steps"]), s["tris_tested"] / max(1, 64 * s["wave_tri_steps"]),
        s["instances_entered"] / max(1, 64 * s["wave_instance_steps"]), s["max_ray_steps"]), flush=True)
    rt.close()
