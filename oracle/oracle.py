"""TEST INFRASTRUCTURE ONLY: ctypes wrapper of oracle/liboracle.so (the CPU golden model, oracle/rt_oracle.cpp).
Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never by raytracer_amd."""
import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(_HERE))
from raytracer_amd import _abi  # noqa: E402  (struct layouts of include/rt_abi.h only)

_lib = _abi.load_library(os.path.join(_HERE, "liboracle.so"))
_lib.orc_create.restype = C.c_void_p
for _name in ("orc_hash1", "orc_hash2", "orc_hash3", "orc_random_u32"):
    getattr(_lib, _name).restype = C.c_uint32
for _name in ("orc_random01", "orc_random11", "orc_schlick"):
    getattr(_lib, _name).restype = C.c_float


class Oracle(_abi.Backend):
    def __init__(self, brute_force=True, threads=1, load_tables=True):
        super().__init__(_lib, "orc_", C.c_void_p(_lib.orc_create()))
        self.set_options(brute_force, threads)
        if load_tables:
            from raytracer_amd import host
            for profile in (0, 1):
                self.load_bssrdf_table(profile, host.load_skin_table(profile))

    def set_options(self, brute_force, threads):
        self._call("set_options", C.c_int(1 if brute_force else 0), C.c_int(threads))

    def close(self):
        if self.h:
            _lib.orc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def counters(self):
        v = [C.c_uint64() for _ in range(4)]
        self._call("get_counters", *[C.byref(x) for x in v])
        return dict(zip(("rays", "nodes", "tris", "insts"), [x.value for x in v]))

    def reset_counters(self):
        self._call("reset_counters")

    def trace_ray(self, o, d, tmin=1e-6, tmax=1e4):
        out = np.zeros(1, dtype=_abi.HIT_DTYPE)
        o3 = (C.c_float * 3)(*[float(x) for x in o]); d3 = (C.c_float * 3)(*[float(x) for x in d])
        self._call("trace_ray", o3, d3, C.c_float(tmin), C.c_float(tmax), C.c_void_p(out.ctypes.data))
        return out[0]

    def bssrdf_dipole(self, radius):
        out = (C.c_float * 3)()
        _lib.orc_bssrdf_dipole(self.h, C.c_float(radius), out)
        return np.array(out, dtype=np.float32)

    def bssrdf_tabulated(self, radius):
        out = (C.c_float * 3)()
        _lib.orc_bssrdf_tabulated(self.h, C.c_float(radius), out)
        return np.array(out, dtype=np.float32)


# ---- unit taps ------------------------------------------------------------------------------------------------
def hash1(x): return int(_lib.orc_hash1(C.c_uint32(x)))
def hash2(x, y): return int(_lib.orc_hash2(C.c_uint32(x), C.c_uint32(y)))
def hash3(x, y, z): return int(_lib.orc_hash3(C.c_uint32(x), C.c_uint32(y), C.c_uint32(z)))


def random_u32(seed):
    s = C.c_uint32(seed); r = _lib.orc_random_u32(C.byref(s)); return int(r), s.value


def random01(seed):
    s = C.c_uint32(seed); r = _lib.orc_random01(C.byref(s)); return np.float32(r), s.value


def random11(seed):
    s = C.c_uint32(seed); r = _lib.orc_random11(C.byref(s)); return np.float32(r), s.value


def sincos(phi):
    s, c = C.c_float(), C.c_float()
    _lib.orc_sincos(C.c_float(phi), C.byref(s), C.byref(c))
    return np.float32(s.value), np.float32(c.value)


def random_on_sphere(seed):
    s = C.c_uint32(seed); out = (C.c_float * 3)()
    _lib.orc_random_on_sphere(C.byref(s), out)
    return np.array(out, dtype=np.float32), s.value


def random_on_hemisphere(seed, normal):
    s = C.c_uint32(seed); out = (C.c_float * 3)(); n = (C.c_float * 3)(*[float(x) for x in normal])
    _lib.orc_random_on_hemisphere(C.byref(s), n, out)
    return np.array(out, dtype=np.float32), s.value


def schlick(n, c): return np.float32(_lib.orc_schlick(C.c_float(n), C.c_float(c)))


def intersect_triangle(o, d, p0, p1, p2):
    f3 = lambda v: (C.c_float * 3)(*[float(x) for x in v])
    tuv = (C.c_float * 3)()
    hit = _lib.orc_intersect_triangle(f3(o), f3(d), f3(p0), f3(p1), f3(p2), tuv)
    return bool(hit), np.array(tuv, dtype=np.float32)


def invert_affine(m):
    a = (C.c_float * 16)(*[float(x) for x in np.asarray(m, dtype=np.float32).reshape(16)])
    out = (C.c_float * 12)()
    _lib.orc_invert_affine(a, out)
    return np.array(out, dtype=np.float32).reshape(4, 3)


def phase_sequence(seed, count):
    out = (C.c_uint32 * count)()
    _lib.orc_phase_sequence(C.c_uint32(seed), C.c_uint32(count), out)
    return list(out)
