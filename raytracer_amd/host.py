"""Python binding of librt_host.so (include/rt_host.h): OBJ loading, scene transforms, camera, image output."""
import ctypes as C
import os

import numpy as np

from . import _abi

_lib = _abi.load_library(os.path.join(_abi.PKG_DIR, "librt_host.so"))
_lib.rt_mesh_create.restype = C.c_void_p
_lib.rt_mesh_vertices.restype = C.c_void_p
_lib.rt_mesh_indices.restype = C.c_void_p
_lib.rt_mesh_vertex_count.restype = C.c_size_t
_lib.rt_mesh_index_count.restype = C.c_size_t
_lib.rt_aabb_widest.restype = C.c_float
_lib.rt_triangle_area.restype = C.c_float
_lib.rt_next_frame_rng.restype = C.c_uint32
_lib.rt_scene_create.restype = C.c_void_p
_lib.rt_scene_instances.restype = C.c_void_p
_lib.rt_scene_globals.restype = C.c_void_p
_lib.rt_scene_instance_count.restype = C.c_size_t
_lib.rt_scene_flags.restype = C.c_uint32
_lib.rt_scene_sample_radius.restype = C.c_float
for _name in ("rt_scene_destroy", "rt_scene_blas_count", "rt_scene_geometry_count", "rt_scene_geometry", "rt_scene_instance_count",
              "rt_scene_instances", "rt_scene_globals", "rt_scene_flags", "rt_scene_sample_radius", "rt_scene_bssrdf_profile"):
    getattr(_lib, _name).argtypes = [C.c_void_p] + ([C.c_int] if _name == "rt_scene_geometry_count" else
                                                    [C.c_int, C.c_int, C.c_void_p] if _name == "rt_scene_geometry" else [])
_lib.rt_scene_seeds.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]


class RtSceneOptions(C.Structure):
    """RtSceneOptions of include/rt_host.h"""
    _fields_ = [("spp", C.c_uint32), ("bounces", C.c_uint32), ("bounces_set", C.c_int32), ("translucent_boxes", C.c_int32),
                ("bssrdf_profile", C.c_int32), ("dipole", C.c_int32), ("sample_radius", C.c_float), ("bunny_scale", C.c_float),
                ("instances", C.c_uint32), ("grid", C.c_uint32)]


def lib():
    return _lib


def data_path(*parts):
    return os.path.join(_abi.DATA_DIR, *parts)


def aabb_null():
    a = (C.c_float * 6)()
    _lib.rt_aabb_null(a)
    return a


def aabb_widest(aabb):
    return float(_lib.rt_aabb_widest(aabb))


def parse_obj_files(filenames, convert_to_rhs, aabb=None):
    """parse_obj_file (parse_obj.cpp:28) applied to each file in turn, appending to one vertex/index array as
    main.cpp:252-254 does.  Returns (vertices VERTEX_DTYPE, indices uint16); extends `aabb` (ctypes float[6])."""
    if isinstance(filenames, str):
        filenames = [filenames]
    mesh = C.c_void_p(_lib.rt_mesh_create())
    try:
        for f in filenames:
            rc = _lib.rt_parse_obj_file(f.encode(), C.c_int(1 if convert_to_rhs else 0), mesh, aabb)
            if rc != _abi.RT_OK:
                raise RuntimeError("rt_parse_obj_file(%s) failed with status %d" % (f, rc))
        nv, ni = _lib.rt_mesh_vertex_count(mesh), _lib.rt_mesh_index_count(mesh)
        verts = np.empty(nv, dtype=_abi.VERTEX_DTYPE)
        idx = np.empty(ni, dtype=np.uint16)
        C.memmove(verts.ctypes.data, _lib.rt_mesh_vertices(mesh), nv * 24)
        C.memmove(idx.ctypes.data, _lib.rt_mesh_indices(mesh), ni * 2)
        return verts, idx
    finally:
        _lib.rt_mesh_destroy(mesh)


def normalizing_transform(aabb):
    m = (C.c_float * 16)()
    _lib.rt_normalizing_transform(aabb, m)
    return np.array(m, dtype=np.float32).reshape(4, 4)


def affine_transform(scale, rot_z, translation):
    m = (C.c_float * 16)()
    t = (C.c_float * 3)(*[float(x) for x in translation])
    _lib.rt_affine_transform(C.c_float(scale), C.c_float(rot_z), t, m)
    return np.array(m, dtype=np.float32).reshape(4, 4)


def default_globals(aspect):
    g = _abi.RtGlobals()
    _lib.rt_default_globals(C.c_float(aspect), C.byref(g))
    return g


def camera_orbit(g, azimuth, elevation, distance, target, fov_y, aspect):
    t = (C.c_float * 3)(*[float(x) for x in target])
    _lib.rt_camera_orbit(C.c_float(azimuth), C.c_float(elevation), C.c_float(distance), t, C.c_float(fov_y),
                         C.c_float(aspect), C.byref(g))


def next_frame_rng(rng):
    return int(_lib.rt_next_frame_rng(C.c_uint32(rng)))


def load_skin_table(profile):
    """(3, 2000) float32 in r, g, b = data_l, data_m, data_s order (raytracing.cpp:116-121)."""
    return np.fromfile(data_path("skin_%d.f32" % profile), dtype="<f4").reshape(3, -1)


def write_png(filename, rgba8):
    h, w = rgba8.shape[:2]
    a = np.ascontiguousarray(rgba8, dtype=np.uint8)
    return _lib.rt_write_png(filename.encode(), C.c_void_p(a.ctypes.data), C.c_uint32(w), C.c_uint32(h))


def write_ppm(filename, rgba8):
    h, w = rgba8.shape[:2]
    a = np.ascontiguousarray(rgba8, dtype=np.uint8)
    return _lib.rt_write_ppm(filename.encode(), C.c_void_p(a.ctypes.data), C.c_uint32(w), C.c_uint32(h))


def write_pfm(filename, rgba, scale=1.0):
    h, w = rgba.shape[:2]
    a = np.ascontiguousarray(rgba, dtype=np.float32)
    return _lib.rt_write_pfm(filename.encode(), C.c_void_p(a.ctypes.data), C.c_uint32(w), C.c_uint32(h), C.c_float(scale))


def quad(p0, p1, p2, p3):
    """Two triangles (0 1 2, 0 2 3) with the face normal of the first, like a quad face in parse_obj.cpp:133-157."""
    p = np.array([p0, p1, p2, p3], dtype=np.float32)
    n = np.cross(p[1] - p[0], p[2] - p[0]).astype(np.float32)
    n = n / np.float32(np.sqrt(np.dot(n, n)))
    verts = np.zeros(4, dtype=_abi.VERTEX_DTYPE)
    verts["position"] = p
    verts["normal"] = n
    return verts, np.array([0, 1, 2, 0, 2, 3], dtype=np.uint16)
