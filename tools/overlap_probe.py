#!/usr/bin/env python3
"""How much do independent launches on separate streams overlap?  N contexts (one stream each) render the same
workload at the same time; aggregate Mrays/s against one context alone."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import api, scenes, distributed
ALL = {"c2": lambda: scenes.bunny(1920, 1080, 1, 1), "c3": lambda: scenes.cornell(1920, 1080, 4, 4),
       "c5": lambda: scenes.bunny_field(1920, 1080, 1, 4), "c2q": lambda: scenes.bunny(960, 540, 1, 1)}
for name in (sys.argv[1:] or ["c2", "c5"]):
    for n in (1, 2, 4):
        ctx = []
        for k in range(n):
            rt = api.Raytracer(0); sc = ALL[name](); sc.setup(rt); ctx.append((rt, sc, distributed.FrameChain(1 + k)))
        steps = 12
        def run(first):
            for f in range(first, first + steps):
                for rt, sc, chain in ctx:
                    distributed.render_frame(rt, sc.globals, f, chain)
            for rt, _, _ in ctx: rt.sync()
        run(0)
        for rt, _, _ in ctx: rt.reset_stats()
        t0 = time.perf_counter(); run(steps); dt = time.perf_counter() - t0
        rays = sum(rt.get_stats()["rays_camera"] for rt, _, _ in ctx)
        print("%s: %d concurrent contexts: %.0f Mrays/s aggregate" % (name, n, rays / dt / 1e6), flush=True)
        for rt, _, _ in ctx: rt.close()
