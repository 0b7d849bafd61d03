"""Python binding of librt_mi355x.so (include/rt_abi.h): the MI355X path tracer behind the reference's
Raytracing:: interface (src/raytracing.h:52-87).  Host-side plumbing only; all compute is in the HIP library."""
import ctypes as C
import os

from . import _abi
from ._abi import *  # noqa: F401,F403  (re-export constants and struct types)

_LIB_PATH = os.environ.get("RT_LIB_PATH") or os.path.join(_abi.PKG_DIR, "librt_mi355x.so")  # RT_LIB_PATH: another build of the same library (A/B of compile-time variants)
_lib = _abi.load_library(_LIB_PATH)
_lib.rt_last_error.restype = C.c_char_p
_lib.rt_abi_version.restype = C.c_int
if _lib.rt_abi_version() != _abi.RT_ABI_VERSION:  # struct layouts below (RtStats) belong to one version of include/rt_abi.h
    raise ImportError("%s has ABI version %d, this binding is for version %d: rebuild with `make`"
                      % (_LIB_PATH, _lib.rt_abi_version(), _abi.RT_ABI_VERSION))


def last_error():
    return (_lib.rt_last_error() or b"").decode()


class Raytracer(_abi.Backend):
    """One context per process and GPU (Raytracing::init, raytracing.cpp:73)."""

    def __init__(self, device=0, load_tables=True):
        handle = C.c_void_p()
        dev = (C.c_int * 1)(device)
        _lib.rt_create.restype = C.c_int
        rc = _lib.rt_create(dev, 1, C.byref(handle))
        if rc != _abi.RT_OK:
            raise RuntimeError("rt_create failed with status %d: %s" % (rc, last_error()))
        super().__init__(_lib, "rt_", handle, last_error)
        if load_tables:
            from . import host
            for profile in (0, 1):
                self.load_bssrdf_table(profile, host.load_skin_table(profile))

    def close(self):
        if self.h:
            _lib.rt_destroy.restype = None
            _lib.rt_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, hip_stream_ptr):
        self._call("set_stream", C.c_void_p(hip_stream_ptr))

    def sync(self):
        self._call("sync")

    def get_stats(self):
        s = _abi.RtStats()
        self._call("get_stats", C.byref(s))
        return {name: getattr(s, name) for name, _ in _abi.RtStats._fields_}

    def reset_stats(self):
        self._call("reset_stats")

    def accumulator_device_ptr(self):
        p, n = C.c_void_p(), C.c_size_t()
        self._call("accumulator_device_ptr", C.byref(p), C.byref(n))
        return p.value, n.value

    def resolve_render_target(self):
        self._call("resolve_render_target")

    def resolve_render_target_frames(self, total_frames):
        self._call("resolve_render_target_frames", C.c_uint32(total_frames))

    # tile split: exchange of the owned tiles only (include/rt_abi.h)
    def tile_pack_bytes(self, world):
        n = C.c_size_t()
        self._call("tile_pack_bytes", C.c_uint32(world), C.byref(n))
        return n.value

    def pack_tiles(self, rank, world, dev_ptr):
        self._call("pack_tiles", C.c_uint32(rank), C.c_uint32(world), C.c_void_p(dev_ptr))

    def unpack_tiles(self, rank, world, dev_ptr):
        self._call("unpack_tiles", C.c_uint32(rank), C.c_uint32(world), C.c_void_p(dev_ptr))
