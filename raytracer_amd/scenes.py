"""The benchmark / parity configurations C1..C5 of BASELINE.md as backend-independent scene
descriptions.  A description is applied to any object with the Backend interface of raytracer_amd._abi
(the HIP product or, in tests, the CPU oracle), so both sides receive bit-identical inputs.

The scenes themselves are assembled by the native host layer (rt_scene_create, raytracer_amd/host/rt_scene.cpp: scene
assembly after src/main.cpp:243-361, the normalising instance transform of :349-360, the defaults of :411-422 /
:496-505) -- the same code the headless frame loop rt_render uses; this module only copies them into numpy."""
import ctypes as C

import numpy as np

from . import _abi, host

LAMBERT, LIGHT, TRANSLUCENT = _abi.RT_SHADER_LAMBERT, _abi.RT_SHADER_LIGHT, _abi.RT_SHADER_TRANSLUCENT


class SceneDesc(object):
    def __init__(self, name, width, height):
        self.name, self.width, self.height = name, width, height
        self.blas = []        # list of lists of geometry dicts
        self.instances = []   # list of (4x4 float32, blas index)
        self.globals = host.default_globals(float(width) / float(height))
        self.flags = _abi.RT_FLAG_COLLECT_SAMPLES | _abi.RT_FLAG_ENABLE_SSS
        self.sample_radius = None   # set => generate_translucent_samples(radius, rng_seed, phase_seed)
        self.rng_seed, self.phase_seed = 1, 2
        self.bssrdf_profile = 0
        self.gather_mode = None     # (hierarchical, max_solid_angle) => set_gather_mode; None = the reference's loop
        self.frames = 1

    def triangle_count(self):
        per_blas = [sum(len(g["indices"]) // 3 for g in geos) for geos in self.blas]
        return sum(per_blas[b] for _, b in self.instances)

    def setup(self, backend, extra_flags=0):
        """Builds the scene on `backend`; returns the number of translucent sample points."""
        backend.update_resolution(self.width, self.height)
        handles = [backend.build_blas(geos) for geos in self.blas]
        backend.build_tlas([(m, handles[b]) for m, b in self.instances])
        backend.set_flags(self.flags | extra_flags)
        backend.set_bssrdf_profile(self.bssrdf_profile)
        backend.set_globals(self.globals)
        if self.gather_mode is not None:
            backend.set_gather_mode(*self.gather_mode)
        points = 0
        if self.sample_radius is not None:
            points = backend.generate_translucent_samples(self.sample_radius, self.rng_seed, self.phase_seed)
        return points


def _geo(verts, idx, shader, color):
    """one geometry of a BLAS for hand-built scene descriptions (tests)"""
    return {"vertices": verts, "indices": idx, "shader": shader, "color": color}


def _identity():
    return np.eye(4, dtype=np.float32)


def from_native(config, width, height, name=None, **options):
    """rt_scene_create(config, ...) copied into a SceneDesc (options: the fields of RtSceneOptions)."""
    opt = host.RtSceneOptions()
    for k, v in options.items():
        if k == "bounces":
            opt.bounces, opt.bounces_set = int(v), 1
        elif v is not None:
            setattr(opt, k, v)
    lib = host.lib()
    h = C.c_void_p(lib.rt_scene_create(config.encode(), _abi.DATA_DIR.encode(), C.c_uint32(width), C.c_uint32(height), C.byref(opt)))
    if not h:
        raise RuntimeError("rt_scene_create(%s) failed" % config)
    try:
        s = SceneDesc(name or config, width, height)
        for b in range(lib.rt_scene_blas_count(h)):
            geos = []
            for k in range(lib.rt_scene_geometry_count(h, b)):
                g = _abi.RtGeometry()
                if lib.rt_scene_geometry(h, C.c_int(b), C.c_int(k), C.byref(g)) != _abi.RT_OK:
                    raise RuntimeError("rt_scene_geometry failed")
                verts = np.empty(g.vertex_count, dtype=_abi.VERTEX_DTYPE)
                idx = np.empty(g.index_count, dtype=np.uint16)
                C.memmove(verts.ctypes.data, g.vertices, g.vertex_count * 24)
                C.memmove(idx.ctypes.data, g.indices, g.index_count * 2)
                geos.append({"vertices": verts, "indices": idx, "shader": int(g.shader), "color": tuple(g.color)})
            s.blas.append(geos)
        n = lib.rt_scene_instance_count(h)
        raw = np.empty(n, dtype=_abi.INSTANCE_DTYPE)
        C.memmove(raw.ctypes.data, lib.rt_scene_instances(h), n * _abi.INSTANCE_DTYPE.itemsize)
        s.instances = [(raw["transform"][i].reshape(4, 4).copy(), int(raw["blas"][i])) for i in range(n)]
        C.memmove(C.byref(s.globals), lib.rt_scene_globals(h), C.sizeof(_abi.RtGlobals))
        s.flags = int(lib.rt_scene_flags(h))
        r = float(lib.rt_scene_sample_radius(h))
        s.sample_radius = r if r > 0.0 else None
        s.bssrdf_profile = int(lib.rt_scene_bssrdf_profile(h))
        a, b = C.c_uint32(), C.c_uint32()
        lib.rt_scene_seeds(h, C.byref(a), C.byref(b))
        s.rng_seed, s.phase_seed = a.value, b.value
        return s
    finally:
        lib.rt_scene_destroy(h)


def debug_cube(width=256, height=256, spp=1, bounces=1):
    """C1: data/debug_cube.obj (rhs=False, Lambert 0.9) scaled x0.4 at the origin under one emissive quad."""
    return from_native("c1", width, height, "c1_debug_cube", spp=spp, bounces=bounces)


def bunny(width=1920, height=1080, spp=1, bounces=1):
    """C2: bunny.obj normalised like main.cpp:352-358 (Lambert 0.9) on a floor quad under a luminaire quad."""
    return from_native("c2", width, height, "c2_bunny", spp=spp, bounces=bounces)


def cornell(width=1920, height=1080, spp=64, bounces=4, translucent_boxes=False, sample_radius=0.025):
    """C3: the cornell box exactly as main.cpp:243-361; boxes Lambert (pure diffuse GI) unless translucent_boxes,
    which gives the scene as shipped (dipole BSSRDF, main.cpp:418-422)."""
    return from_native("c3", width, height, "c3_cornell" + ("_translucent" if translucent_boxes else ""), spp=spp, bounces=bounces,
                       translucent_boxes=1 if translucent_boxes else 0, sample_radius=sample_radius)


def cornell_bunny(width=1920, height=1080, spp=16, bounces=4, profile=0, sample_radius=0.025, bunny_scale=0.6,
                  tabulated=True):
    """C4: cornell walls + a translucent bunny, tabulated skin profile (scale 0.4, fudge 1, eta 1.75)."""
    return from_native("c4", width, height, "c4_cornell_bunny_skin%d" % profile, spp=spp, bounces=bounces, bssrdf_profile=profile,
                       sample_radius=sample_radius, bunny_scale=bunny_scale, dipole=0 if tabulated else 1)


def bunny_field(width=3840, height=2160, spp=1, bounces=4, grid=230, count=52743):
    """C5: `count` instanced bunnies on a grid x grid jittered lattice (seeded xorshift: uniform scale, random Z
    rotation) over a floor quad under a large luminaire; 52,743 x 948 = 50.0 M triangles at the default size."""
    return from_native("c5", width, height, "c5_bunny_field_%d" % count, spp=spp, bounces=bounces, grid=grid, instances=count)


CONFIGS = {"c1": debug_cube, "c2": bunny, "c3": cornell, "c4": cornell_bunny, "c5": bunny_field}
