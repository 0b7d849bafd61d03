#!/usr/bin/env python3
"""Serial latency of one traversal step: tiny launches (every wave resident at once), so the kernel time is the
longest wave's dependent chain.  Prints kernel time / max steps of one ray."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import api, scenes, distributed, _abi
ALL = {"c2": lambda w, h: scenes.bunny(w, h, 1, 0), "c3": lambda w, h: scenes.cornell(w, h, 1, 0),
       "c5": lambda w, h: scenes.bunny_field(w, h, 1, 0)}
for name in (sys.argv[1:] or ["c2", "c5"]):
    for (w, h) in [(64, 64), (256, 256), (1920, 1080)]:
        sc = ALL[name](w, h)
        rt = api.Raytracer(0); sc.setup(rt)
        chain = distributed.FrameChain(1)
        for k in range(3): distributed.render_frame(rt, sc.globals, k, chain)
        rt.set_flags(sc.flags | _abi.RT_FLAG_TIME_KERNELS); rt.reset_stats()
        for k in range(3, 13): distributed.render_frame(rt, sc.globals, k, chain)
        rt.sync(); s = rt.get_stats(); ms = s["traverse_ms"] / s["traverse_launches"]; rays = s["rays_camera"] / s["traverse_launches"]
        rt.set_flags(sc.flags | _abi.RT_FLAG_COUNT_TRAVERSAL); rt.reset_stats()
        distributed.render_frame(rt, sc.globals, 13, chain); rt.sync(); s = rt.get_stats()
        r = s["rays_camera"]
        print("%s %dx%d primary only: %.1f us per launch, %d rays, max steps of one ray %d, mean steps/ray %.1f, wave trips/wave %.1f -> %.2f us per step of the longest ray" % (
            name, w, h, ms * 1e3, rays, s["max_ray_steps"], (s["nodes_visited"] + s["tris_tested"] + s["instances_entered"]) / r,
            (s["wave_node_steps"] + s["wave_tri_steps"] + s["wave_instance_steps"]) / (r / 64), ms * 1e3 / max(1, s["max_ray_steps"])), flush=True)
        rt.close()
