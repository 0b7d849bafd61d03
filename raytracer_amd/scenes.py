"""The benchmark / parity configurations C1..C5 of BASELINE.md as backend-independent scene
descriptions.  A description is applied to any object with the Backend interface of raytracer_amd._abi
(the HIP product or, in tests, the CPU oracle), so both sides receive bit-identical inputs.

Scene assembly follows src/main.cpp:243-361 (cornell), :349-360 (normalising instance transform) and the defaults of
:411-422 / :496-505; all geometry is loaded by the host library (rt_parse_obj_file)."""
import math

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
    return {"vertices": verts, "indices": idx, "shader": shader, "color": color}


def _identity():
    return np.eye(4, dtype=np.float32)


def debug_cube(width=256, height=256, spp=1, bounces=1):
    """C1: data/debug_cube.obj (rhs=False, Lambert 0.9) scaled x0.4 at the origin under one emissive quad."""
    s = SceneDesc("c1_debug_cube", width, height)
    verts, idx = host.parse_obj_files(host.data_path("debug_cube.obj"), False)
    s.blas.append([_geo(verts, idx, LAMBERT, (0.9, 0.9, 0.9))])
    lv, li = host.quad((-0.5, -0.5, 0.8), (-0.5, 0.5, 0.8), (0.5, 0.5, 0.8), (0.5, -0.5, 0.8))  # faces down
    fv, fi = host.quad((-1.5, -1.5, -0.2), (1.5, -1.5, -0.2), (1.5, 1.5, -0.2), (-1.5, 1.5, -0.2))  # faces up
    s.blas.append([_geo(fv, fi, LAMBERT, (0.9, 0.9, 0.9)), _geo(lv, li, LIGHT, (0.0, 0.0, 0.0))])
    s.instances.append((host.affine_transform(0.4, 0.0, (0.0, 0.0, 0.0)), 0))
    s.instances.append((_identity(), 1))
    s.globals.samples_per_pixel = spp
    s.globals.bounces_per_sample = bounces
    return s


def _bunny():
    aabb = host.aabb_null()
    verts, idx = host.parse_obj_files(host.data_path("bunny.obj"), True, aabb)  # Y-up mesh -> Z-up world
    return verts, idx, aabb


def bunny(width=1920, height=1080, spp=1, bounces=1):
    """C2: bunny.obj normalised like main.cpp:352-358 (Lambert 0.9) on a floor quad under a luminaire quad."""
    s = SceneDesc("c2_bunny", width, height)
    verts, idx, aabb = _bunny()
    s.blas.append([_geo(verts, idx, LAMBERT, (0.9, 0.9, 0.9))])
    m = host.normalizing_transform(aabb)
    zmin = float(aabb[2] * m[2, 2] + m[3, 2])
    fv, fi = host.quad((-1.5, -1.5, zmin), (1.5, -1.5, zmin), (1.5, 1.5, zmin), (-1.5, 1.5, zmin))
    lv, li = host.quad((-0.5, -0.5, 1.2), (-0.5, 0.5, 1.2), (0.5, 0.5, 1.2), (0.5, -0.5, 1.2))
    s.blas.append([_geo(fv, fi, LAMBERT, (0.9, 0.9, 0.9)), _geo(lv, li, LIGHT, (0.0, 0.0, 0.0))])
    s.instances.append((m, 0))
    s.instances.append((_identity(), 1))
    s.globals.samples_per_pixel = spp
    s.globals.bounces_per_sample = bounces
    return s


_CORNELL_WALLS = [
    (["floor.obj", "back.obj", "ceiling.obj"], (0.9, 0.9, 0.9), LAMBERT),   # main.cpp:249-265
    (["redwall.obj"], (0.9, 0.0, 0.0), LAMBERT),                             # :266-280
    (["greenwall.obj"], (0.0, 0.9, 0.0), LAMBERT),                           # :281-295
    (["luminaire.obj"], (0.0, 0.0, 0.0), LIGHT),                             # :296-310
]
_CORNELL_BOXES = [(["largebox.obj"], (0.9, 0.9, 0.9)), (["smallbox.obj"], (0.9, 0.9, 0.9))]  # :311-340


def _cornell_geometries(box_shader, with_boxes=True):
    aabb = host.aabb_null()
    geos = []
    for files, color, shader in _CORNELL_WALLS:
        v, i = host.parse_obj_files([host.data_path("cornell", f) for f in files], True, aabb)
        geos.append(_geo(v, i, shader, color))
    if with_boxes:
        for files, color in _CORNELL_BOXES:
            v, i = host.parse_obj_files([host.data_path("cornell", f) for f in files], True, aabb)
            geos.append(_geo(v, i, box_shader, color))
    return geos, aabb


def cornell(width=1920, height=1080, spp=64, bounces=4, translucent_boxes=False, sample_radius=0.025):
    """C3: the cornell box exactly as main.cpp:243-361; boxes Lambert (pure diffuse GI) unless translucent_boxes,
    which gives the scene as shipped (dipole BSSRDF, main.cpp:418-422)."""
    s = SceneDesc("c3_cornell" + ("_translucent" if translucent_boxes else ""), width, height)
    geos, aabb = _cornell_geometries(TRANSLUCENT if translucent_boxes else LAMBERT)
    s.blas.append(geos)
    s.instances.append((host.normalizing_transform(aabb), 0))
    s.globals.samples_per_pixel = spp
    s.globals.bounces_per_sample = bounces
    if translucent_boxes:
        s.sample_radius = sample_radius
    return s


def cornell_bunny(width=1920, height=1080, spp=16, bounces=4, profile=0, sample_radius=0.025, bunny_scale=0.6,
                  tabulated=True):
    """C4: cornell walls + a translucent bunny, tabulated skin profile (scale 0.4, fudge 1, eta 1.75)."""
    s = SceneDesc("c4_cornell_bunny_skin%d" % profile, width, height)
    geos, aabb = _cornell_geometries(LAMBERT, with_boxes=False)
    s.blas.append(geos)
    room = host.normalizing_transform(aabb)
    s.instances.append((room, 0))
    verts, idx, baabb = _bunny()
    s.blas.append([_geo(verts, idx, TRANSLUCENT, (0.9, 0.9, 0.9))])
    m = host.normalizing_transform(baabb)
    m[:3, :3] *= np.float32(bunny_scale)
    m[3, :3] *= np.float32(bunny_scale)
    floor_z = float(aabb[2] * room[2, 2] + room[3, 2])
    bunny_zmin = float(baabb[2] * m[2, 2] + m[3, 2])
    m[3, 2] += np.float32(floor_z - bunny_zmin + 0.002)
    s.instances.append((m, 1))
    s.globals.samples_per_pixel = spp
    s.globals.bounces_per_sample = bounces
    s.globals.translucent_emission_bounces = 1
    s.globals.translucent_bssrdf_scale = 0.4 if tabulated else 0.0   # main.cpp:558
    s.globals.translucent_bssrdf_fudge = 1.0
    s.globals.translucent_refractive_index = 1.75
    s.sample_radius = sample_radius
    s.bssrdf_profile = profile
    return s


def _xorshift(state):
    state ^= (state << 13) & 0xFFFFFFFF
    state ^= state >> 17
    state ^= (state << 5) & 0xFFFFFFFF
    return state & 0xFFFFFFFF


def bunny_field(width=3840, height=2160, spp=1, bounces=4, grid=230, count=52743, seed=0x2545F491):
    """C5: `count` instanced bunnies on a grid x grid jittered lattice (seeded xorshift: uniform scale, random Z
    rotation) over a floor quad under a large luminaire; 52,743 x 948 = 50.0 M triangles at the default size."""
    s = SceneDesc("c5_bunny_field_%d" % count, width, height)
    verts, idx, aabb = _bunny()
    s.blas.append([_geo(verts, idx, LAMBERT, (0.9, 0.9, 0.9))])
    base = host.normalizing_transform(aabb)
    zmin = float(aabb[2] * base[2, 2] + base[3, 2])
    spacing = 1.3
    half = 0.5 * grid * spacing
    fv, fi = host.quad((-half - 2, -half - 2, 0.0), (half + 2, -half - 2, 0.0), (half + 2, half + 2, 0.0), (-half - 2, half + 2, 0.0))
    lh = 0.75 * half + 1.0
    top = 0.25 * half + 4.0
    lv, li = host.quad((-lh, -lh, top), (-lh, lh, top), (lh, lh, top), (lh, -lh, top))
    s.blas.append([_geo(fv, fi, LAMBERT, (0.9, 0.9, 0.9)), _geo(lv, li, LIGHT, (0.0, 0.0, 0.0))])
    state = seed
    count = min(count, grid * grid)
    for k in range(count):
        gx, gy = k % grid, k // grid
        state = _xorshift(state); jx = (state & 0xFFFF) / 65536.0 - 0.5
        state = _xorshift(state); jy = (state & 0xFFFF) / 65536.0 - 0.5
        state = _xorshift(state); sc = 0.7 + 0.3 * ((state & 0xFFFF) / 65536.0)
        state = _xorshift(state); rot = 2.0 * math.pi * ((state & 0xFFFF) / 65536.0)
        x = (gx + 0.5) * spacing - half + 0.25 * spacing * jx
        y = (gy + 0.5) * spacing - half + 0.25 * spacing * jy
        xf = host.affine_transform(sc, rot, (x, y, -zmin * sc))
        s.instances.append(((base @ xf).astype(np.float32), 0))
    s.instances.append((_identity(), 1))
    s.globals.samples_per_pixel = spp
    s.globals.bounces_per_sample = bounces
    s.globals.frame_rng = 0x12345678
    dist = max(6.0, 0.9 * half)
    host.camera_orbit(s.globals, 0.6, math.radians(25.0), dist, (0.0, 0.0, 0.3), math.radians(40.0), float(width) / float(height))
    return s


CONFIGS = {"c1": debug_cube, "c2": bunny, "c3": cornell, "c4": cornell_bunny, "c5": bunny_field}
