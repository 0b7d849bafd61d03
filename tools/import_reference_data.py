#!/usr/bin/env python3
"""Import the reference's *data* (meshes and tabulated BSSRDF profiles) into raytracer_amd/data/.

Only data travels: OBJ meshes are copied verbatim (they are scene inputs), and the two
profile tables data/skin_0.h / data/skin_1.h (C arrays of floats) are converted to raw
little-endian float32 files holding the three radial arrays in the order  data_l, data_m, data_s  (2000 floats
each), which is the r,g,b order the reference uploads them in (raytracing.cpp:116-121).

Run in the authoring container only (needs /root/reference):  python tools/import_reference_data.py
"""
import os, re, shutil, sys
import numpy as np

REF = "/root/reference/data"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "raytracer_amd", "data")


def parse_skin(path):
    text = open(path).read()
    n = int(re.search(r"data_len\s*=\s*(\d+)", text).group(1))
    arrays = {}
    for m in re.finditer(r"float\s+(data_[sml])\[\]\s*=\s*\{([^}]*)\}", text):
        vals = np.array([float(x) for x in m.group(2).replace("\n", " ").split(",") if x.strip()], dtype=np.float32)
        assert len(vals) == n, (path, m.group(1), len(vals))
        arrays[m.group(1)] = vals
    assert set(arrays) == {"data_s", "data_m", "data_l"}
    return np.concatenate([arrays["data_l"], arrays["data_m"], arrays["data_s"]])


def main():
    os.makedirs(os.path.join(OUT, "cornell"), exist_ok=True)
    for name in ("bunny.obj", "debug_cube.obj"):
        shutil.copyfile(os.path.join(REF, name), os.path.join(OUT, name))
    for name in sorted(os.listdir(os.path.join(REF, "cornell"))):
        shutil.copyfile(os.path.join(REF, "cornell", name), os.path.join(OUT, "cornell", name))
    for k in (0, 1):
        parse_skin(os.path.join(REF, "skin_%d.h" % k)).astype("<f4").tofile(os.path.join(OUT, "skin_%d.f32" % k))
    print("imported into", OUT)


if __name__ == "__main__":
    sys.exit(main())
