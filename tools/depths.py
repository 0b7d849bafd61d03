import sys; sys.path.insert(0,'.')
from raytracer_amd import api, scenes
for name, sc in [("c2", scenes.bunny(64,64)), ("c3", scenes.cornell(64,64,1)), ("c4", scenes.cornell_bunny(64,64,1)), ("c5", scenes.bunny_field(64,64))]:
    rt = api.Raytracer(0); sc.setup(rt); s = rt.get_stats(); print(name, s["tlas_depth"], s["max_blas_depth"], s["stack_in_lds"]); rt.close()
