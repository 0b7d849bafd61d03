#!/usr/bin/env python3
"""Traversal counters of the 52,743-instance field bounce by bounce: the same two frames with 0, 1, 2, 3, 4 bounces; the
counters of bounce b are the difference between the runs with b and b - 1 bounces (a path's rays do not depend on the
bounce limit).  usage: count_by_bounce.py [width height]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import api, scenes, distributed, _abi
w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3840, 2160)
keys = ["rays", "nodes_visited", "tris_tested", "instances_entered", "wave_node_steps", "wave_tri_steps", "wave_instance_steps"]
prev = dict((k, 0) for k in keys)
for b in range(5):
    sc = scenes.bunny_field(w, h, 1, b)
    rt = api.Raytracer(0); sc.setup(rt, _abi.RT_FLAG_COUNT_TRAVERSAL)
    rt.reset_stats()
    distributed.render_frame(rt, sc.globals, 0, 5); distributed.render_frame(rt, sc.globals, 1, 5)
    s = rt.get_stats(); s["rays"] = s["rays_camera"] + s["rays_probe"]
    d = dict((k, s[k] - prev[k]) for k in keys); prev = dict((k, s[k]) for k in keys)
    rays = max(1, d["rays"])
    print("bounce %d: rays %9d | per ray: nodes %.2f tris %.2f insts %.2f | wave steps per 64 rays: node %.1f tri %.1f inst %.1f | SIMD eff node %.2f tri %.2f inst %.2f" % (
        b, d["rays"], d["nodes_visited"] / rays, d["tris_tested"] / rays, d["instances_entered"] / rays,
        d["wave_node_steps"] / (rays / 64), d["wave_tri_steps"] / (rays / 64), d["wave_instance_steps"] / (rays / 64),
        d["nodes_visited"] / max(1, 64 * d["wave_node_steps"]), d["tris_tested"] / max(1, 64 * d["wave_tri_steps"]),
        d["instances_entered"] / max(1, 64 * d["wave_instance_steps"])), flush=True)
    rt.close()
