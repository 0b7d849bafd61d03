#!/usr/bin/env python3
"""where the pooled traversal's time goes (RT_POOL_DEBUG build): one C5 frame sequence, profile printed by rt_get_stats"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RT_POOL"] = "2"; os.environ["RT_FRAMES_IN_FLIGHT"] = "1"
from raytracer_amd import api, scenes, distributed
w, h = (3840, 2160) if len(sys.argv) < 2 else (int(sys.argv[1]), int(sys.argv[2]))
sc = scenes.bunny_field(w, h, spp=1, bounces=4)
rt = api.Raytracer(0); sc.setup(rt)
for f in range(2): distributed.render_frame(rt, sc.globals, f, 5)
rt.sync(); rt.reset_stats()
for f in range(2, 6): distributed.render_frame(rt, sc.globals, f, 5)
s = rt.get_stats(); print("rays", s["rays_camera"])
rt.close()
