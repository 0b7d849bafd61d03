"""raytracer_amd -- MI355X (gfx950) implementation of the per-ray hot path of gluyas/raytracer.

  raytracer_amd.api     ctypes binding of librt_mi355x.so (HIP kernels behind include/rt_abi.h); needs a GPU
  raytracer_amd.host    ctypes binding of librt_host.so   (CPU scene layer, include/rt_host.h)
  raytracer_amd.scenes  the benchmark configurations C1..C5 of BASELINE.md as backend-independent descriptions
"""
__all__ = ["api", "host", "scenes"]
