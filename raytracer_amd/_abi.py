"""ctypes mirror of include/rt_abi.h and include/rt_host.h (struct layouts and library loading).

The product libraries must exist in-tree (built by `make` / __graft_entry__.build()); there is no fallback:
a missing librt_mi355x.so raises at import of raytracer_amd.api.
"""
import ctypes as C
import os

import numpy as np

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
DATA_DIR = os.path.join(PKG_DIR, "data")

RT_ABI_VERSION = 3   # include/rt_abi.h
RT_OK = 0
RT_SHADER_LAMBERT, RT_SHADER_LIGHT, RT_SHADER_TRANSLUCENT = 0, 1, 2
RT_FLAG_COLLECT_SAMPLES = 1 << 0
RT_FLAG_ENABLE_SSS = 1 << 1
RT_FLAG_DOUBLE_DISPATCH = 1 << 2
RT_FLAG_CAPTURE_HITS = 1 << 3
RT_FLAG_COUNT_TRAVERSAL = 1 << 4
RT_FLAG_TIME_KERNELS = 1 << 5
RT_FLAG_ALL_PHASE_GROUPS = 1 << 6
RT_FLAG_TIME_SAMPLED = 1 << 7

VERTEX_DTYPE = np.dtype([("position", "<f4", 3), ("normal", "<f4", 3)])
SAMPLE_POINT_DTYPE = np.dtype([("position", "<f4", 3), ("payload", "<f4", 3)])
HIT_DTYPE = np.dtype([("instance", "<i4"), ("geometry", "<i4"), ("primitive", "<i4"), ("t", "<f4"),
                      ("u", "<f4"), ("v", "<f4"), ("rng_after", "<u4"), ("valid", "<u4")])


class RtGlobals(C.Structure):
    _pack_ = 4
    _fields_ = [
        ("frame_rng", C.c_uint32), ("accumulator_count", C.c_uint32), ("samples_per_pixel", C.c_uint32),
        ("bounces_per_sample", C.c_uint32), ("camera_to_world", C.c_float * 16), ("camera_aspect", C.c_float),
        ("camera_focal_length", C.c_float), ("translucent_accumulator_count", C.c_uint32),
        ("translucent_instance_stride", C.c_uint32), ("translucent_emission_bounces", C.c_int32),
        ("translucent_refractive_index", C.c_float), ("translucent_bssrdf_scale", C.c_float),
        ("translucent_bssrdf_fudge", C.c_float), ("translucent_scattering", C.c_float * 3), ("_pad124", C.c_uint32),
        ("translucent_absorption", C.c_float * 3), ("_pad140", C.c_uint32), ("light_color", C.c_float * 3),
    ]

    def copy(self):
        g = RtGlobals()
        C.memmove(C.byref(g), C.byref(self), C.sizeof(RtGlobals))
        return g


assert C.sizeof(RtGlobals) == 156


class RtGeometry(C.Structure):
    _fields_ = [("vertices", C.c_void_p), ("vertex_count", C.c_size_t), ("indices", C.c_void_p),
                ("index_count", C.c_size_t), ("shader", C.c_int32), ("color", C.c_float * 3)]


class RtInstance(C.Structure):
    _fields_ = [("transform", C.c_float * 16), ("blas", C.c_int32)]


class RtStats(C.Structure):
    _fields_ = [("rays_camera", C.c_uint64), ("rays_probe", C.c_uint64), ("gather_pairs", C.c_uint64),
                ("nodes_visited", C.c_uint64), ("tris_tested", C.c_uint64), ("instances_entered", C.c_uint64),
                ("traverse_launches", C.c_uint64), ("traverse_ms", C.c_double), ("blas_build_ms", C.c_double),
                ("tlas_build_ms", C.c_double), ("blas_triangles", C.c_uint64), ("tlas_instances", C.c_uint64),
                ("wave_node_steps", C.c_uint64), ("wave_tri_steps", C.c_uint64), ("wave_instance_steps", C.c_uint64),
                ("max_ray_steps", C.c_uint64), ("gather_launches", C.c_uint64), ("gather_ms", C.c_double),
                ("traverse_launches_timed", C.c_uint64), ("gather_launches_timed", C.c_uint64),
                ("primary_launches", C.c_uint64), ("primary_launches_timed", C.c_uint64), ("primary_ms", C.c_double),
                ("rays_primary", C.c_uint64), ("bluenoise_ms", C.c_double), ("bluenoise_points", C.c_uint64),
                ("tlas_depth", C.c_uint32), ("max_blas_depth", C.c_uint32), ("stack_in_lds", C.c_uint32), ("flat_depth", C.c_uint32),
                ("fused_bounce0", C.c_uint32), ("fused_tail", C.c_uint32), ("tlas_grid_cells", C.c_uint32)]


def load_library(path):
    if not os.path.exists(path):
        raise ImportError("%s is missing: build it with `make` (or __graft_entry__.build()) first; "
                          "there is no fallback path" % path)
    return C.CDLL(path)


def geometry_array(geometries):
    """geometries: list of dicts {vertices: VERTEX_DTYPE array, indices: uint16 array, shader, color}.
    Returns (ctypes array, keepalive list)."""
    arr = (RtGeometry * len(geometries))()
    keep = []
    for i, ge in enumerate(geometries):
        v = np.ascontiguousarray(ge["vertices"], dtype=VERTEX_DTYPE)
        ix = np.ascontiguousarray(ge["indices"], dtype=np.uint16)
        keep += [v, ix]
        arr[i].vertices = v.ctypes.data
        arr[i].vertex_count = len(v)
        arr[i].indices = ix.ctypes.data
        arr[i].index_count = len(ix)
        arr[i].shader = int(ge["shader"])
        arr[i].color = (C.c_float * 3)(*[float(x) for x in ge["color"]])
    return arr, keep


INSTANCE_DTYPE = np.dtype([("transform", "<f4", 16), ("blas", "<i4")])
assert INSTANCE_DTYPE.itemsize == C.sizeof(RtInstance)


def instance_array(instances):
    """instances: list of (transform 4x4 row-major array-like, blas handle) -> ctypes array of RtInstance."""
    n = len(instances)
    buf = np.zeros(n, dtype=INSTANCE_DTYPE)
    if n:
        buf["transform"] = np.asarray([np.asarray(m, dtype=np.float32).reshape(16) for m, _ in instances], dtype=np.float32)
        buf["blas"] = np.asarray([int(b) for _, b in instances], dtype=np.int32)
    arr = (RtInstance * n).from_buffer_copy(buf.tobytes()) if n else (RtInstance * 0)()
    return arr


class Backend(object):
    """Common wrapper over a library exporting <prefix>update_resolution, <prefix>build_blas, ... with the
    signatures of include/rt_abi.h.  raytracer_amd.api.Raytracer (prefix rt_) is the product; test code may wrap
    another library with the same call shapes in this class so that parity tests drive both the same way."""

    def __init__(self, lib, prefix, handle, last_error=None):
        self.lib, self.prefix, self.h = lib, prefix, handle
        self._last_error = last_error
        self.width = self.height = 0

    def _fn(self, name):
        return getattr(self.lib, self.prefix + name)

    def _check(self, rc, what):
        if rc != RT_OK:
            msg = self._last_error() if self._last_error else ""
            raise RuntimeError("%s%s failed with status %d %s" % (self.prefix, what, rc, msg))

    def _call(self, name, *args):
        fn = self._fn(name)
        fn.restype = C.c_int
        self._check(fn(self.h, *args), name)

    def update_resolution(self, w, h):
        self._call("update_resolution", C.c_uint32(w), C.c_uint32(h))
        self.width, self.height = w, h

    def load_bssrdf_table(self, profile, table):
        """table: float32 array (3, n) in r, g, b order."""
        t = np.ascontiguousarray(table, dtype=np.float32)
        self._call("load_bssrdf_table", C.c_int(profile), C.c_void_p(t[0].ctypes.data), C.c_void_p(t[1].ctypes.data),
                   C.c_void_p(t[2].ctypes.data), C.c_uint32(t.shape[1]))

    def set_bssrdf_profile(self, profile):
        self._call("set_bssrdf_profile", C.c_int(profile))

    def build_blas(self, geometries):
        arr, keep = geometry_array(geometries)
        out = C.c_int32(-1)
        self._call("build_blas", arr, C.c_int(len(geometries)), C.byref(out))
        del keep
        return out.value

    def build_tlas(self, instances):
        arr = instance_array(instances)
        self._call("build_tlas", arr, C.c_int(len(instances)))

    def generate_translucent_samples(self, radius, rng_seed=1, phase_seed=2):
        total = C.c_uint32(0)
        self._call("generate_translucent_samples", C.c_float(radius), C.c_uint32(rng_seed), C.c_uint32(phase_seed),
                   C.byref(total))
        return total.value

    def set_globals(self, g):
        self._call("set_globals", C.byref(g))

    def get_globals(self):
        g = RtGlobals()
        self._call("get_globals", C.byref(g))
        return g

    def set_flags(self, flags):
        self._call("set_flags", C.c_uint32(flags))

    def set_gather_mode(self, hierarchical, max_solid_angle=0.0):
        self._call("set_gather_mode", C.c_int(1 if hierarchical else 0), C.c_float(max_solid_angle))

    def set_tile_split(self, rank, world):
        self._call("set_tile_split", C.c_uint32(rank), C.c_uint32(world))

    def set_probe_split(self, rank, world):
        self._call("set_probe_split", C.c_uint32(rank), C.c_uint32(world))

    def set_probe_exchange(self, fn):
        """fn(stream_ptr, points_ptr, n_points, block_points) -> 0, called after every sample-point pass (include/rt_abi.h);
        None removes it.  The ctypes thunk is kept alive on this object."""
        proto = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32)

        def thunk(user, stream, points, n, block):
            try:
                return int(fn(stream or 0, points or 0, n, block) or 0)
            except Exception:   # an exception must not unwind through the C caller
                import traceback
                traceback.print_exc()
                return 1
        keep = proto(thunk) if fn is not None else proto()   # (no argument: a NULL function pointer)
        f = self._fn("set_probe_exchange")
        f.restype = C.c_int
        f.argtypes = [C.c_void_p, proto, C.c_void_p]
        self._check(f(self.h, keep, None), "set_probe_exchange")
        self._probe_exchange_thunk = keep   # (the previous thunk may be released only now)

    def dispatch_rays(self):
        self._call("dispatch_rays")

    def dispatch_rays_static(self):
        self._call("dispatch_rays_static")

    def read_accumulator(self):
        out = np.empty((self.height, self.width, 4), dtype=np.float32)
        self._call("read_accumulator", C.c_void_p(out.ctypes.data))
        return out

    def read_render_target(self):
        out = np.empty((self.height, self.width, 4), dtype=np.uint8)
        self._call("read_render_target", C.c_void_p(out.ctypes.data))
        return out

    def read_hit_records(self, bounce):
        out = np.empty(self.height * self.width, dtype=HIT_DTYPE)
        self._call("read_hit_records", C.c_uint32(bounce), C.c_void_p(out.ctypes.data))
        return out.reshape(self.height, self.width)

    def translucent_instance_count(self):
        n = C.c_uint32(0)
        self._call("translucent_instance_count", C.byref(n))
        return n.value

    def sample_point_count(self, instance):
        n = C.c_uint32(0)
        self._call("sample_point_count", C.c_uint32(instance), C.byref(n))
        return n.value

    def read_sample_points(self, instance):
        n = self.sample_point_count(instance)
        pts = np.empty(n, dtype=SAMPLE_POINT_DTYPE)
        nrm = np.empty((n, 3), dtype=np.float32)
        if n:
            self._call("read_sample_points", C.c_uint32(instance), C.c_void_p(pts.ctypes.data), C.c_void_p(nrm.ctypes.data))
        return pts, nrm
