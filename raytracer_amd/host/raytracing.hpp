// raytracing.hpp -- C++ host mirror of the reference's Raytracing:: interface (src/raytracing.h:19-87) on top of
// the C-ABI of librt_mi355x.so.  Same names, argument meaning and call order as the reference so that a caller
// written against src/raytracing.h ports by dropping the D3D12 command-list arguments.  Errors abort, like the
// reference's CHECK_RESULT (prelude.h:185), after printing rt_last_error().
#pragma once
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../include/rt_abi.h"
#include "../../include/rt_host.h"

typedef RtVertex Vertex;   // prelude.h:96-99
typedef RtIndex  Index;    // prelude.h:66
typedef RtGlobals RaytracingGlobals;

enum Shader { Lambert = 0, Light, Translucent, Count }; // raytracing.h:19-25

struct Material { Shader shader; float color[3]; };     // raytracing.h:27-30

template <typename T> struct ArrayView { const T* ptr; size_t len; }; // array.h

struct GeometryInstance {                                // raytracing.h:32-36
    ArrayView<Vertex> vertices;
    ArrayView<Index>  indices;
    Material          material;
};

struct Blas { rt_blas handle; };                         // raytracing.h:38-45
struct BlasInstance { float transform[16]; Blas* blas; };// raytracing.h:47-50

namespace Raytracing {

inline rt_ctx*& ctx() { static rt_ctx* c = nullptr; return c; }
inline RaytracingGlobals g_globals = {};                 // raytracing.h:61
inline bool g_enable_translucent_sample_collection = true; // raytracing.h:63
inline bool g_enable_subsurface_scattering = true;       // raytracing.h:64
inline bool g_reference_double_dispatch = false;         // Q1: camera_rgen twice per frame

inline void check(int rc, const char* what) {
    if (rc != RT_OK) { fprintf(stderr, "%s failed (%d): %s\n", what, rc, rt_last_error()); abort(); }
}

// Raytracing::init (raytracing.cpp:73): creates the context and uploads the BSSRDF tables (3 x n floats: r, g, b)
inline void init(int device, const float* skin0_rgb, const float* skin1_rgb, uint32_t n) {
    if (rt_abi_version() != RT_ABI_VERSION) { fprintf(stderr, "librt_mi355x.so has ABI version %d, this caller was built against %d\n", rt_abi_version(), RT_ABI_VERSION); abort(); }
    check(rt_create(&device, 1, &ctx()), "rt_create");
    if (skin0_rgb) check(rt_load_bssrdf_table(ctx(), 0, skin0_rgb, skin0_rgb + n, skin0_rgb + 2 * n, n), "rt_load_bssrdf_table");
    if (skin1_rgb) check(rt_load_bssrdf_table(ctx(), 1, skin1_rgb, skin1_rgb + n, skin1_rgb + 2 * n, n), "rt_load_bssrdf_table");
}
inline void shutdown() { rt_destroy(ctx()); ctx() = nullptr; }

inline void update_resolution(uint32_t width, uint32_t height) { check(rt_update_resolution(ctx(), width, height), "rt_update_resolution"); }

inline Blas build_blas(ArrayView<GeometryInstance> geometries) { // raytracing.cpp:316
    std::vector<RtGeometry> g(geometries.len);
    for (size_t i = 0; i < geometries.len; i++) {
        const GeometryInstance& s = geometries.ptr[i];
        g[i].vertices = s.vertices.ptr; g[i].vertex_count = s.vertices.len;
        g[i].indices = s.indices.ptr;   g[i].index_count = s.indices.len;
        g[i].shader = (int) s.material.shader;
        for (int k = 0; k < 3; k++) g[i].color[k] = s.material.color[k];
    }
    Blas b;
    check(rt_build_blas(ctx(), g.data(), (int) g.size(), &b.handle), "rt_build_blas");
    return b;
}

inline void build_tlas(ArrayView<BlasInstance> instances) {       // raytracing.cpp:446
    std::vector<RtInstance> in(instances.len);
    for (size_t i = 0; i < instances.len; i++) {
        for (int k = 0; k < 16; k++) in[i].transform[k] = instances.ptr[i].transform[k];
        in[i].blas = instances.ptr[i].blas->handle;
    }
    check(rt_build_tlas(ctx(), in.data(), (int) in.size()), "rt_build_tlas");
    RtGlobals cur; check(rt_get_globals(ctx(), &cur), "rt_get_globals");
    g_globals.translucent_instance_stride = cur.translucent_instance_stride; // raytracing.cpp:461,491
}

inline uint32_t generate_translucent_samples(float radius, uint32_t rng_seed = 1, uint32_t phase_seed = 2) { // raytracing.cpp:553
    uint32_t total = 0;
    check(rt_generate_translucent_samples(ctx(), radius, rng_seed, phase_seed, &total), "rt_generate_translucent_samples");
    return total;
}

inline void dispatch_rays() {                                       // raytracing.cpp:652
    uint32_t flags = (g_enable_translucent_sample_collection ? RT_FLAG_COLLECT_SAMPLES : 0) |
                     (g_enable_subsurface_scattering ? RT_FLAG_ENABLE_SSS : 0) |
                     (g_reference_double_dispatch ? RT_FLAG_DOUBLE_DISPATCH : 0);
    check(rt_set_flags(ctx(), flags), "rt_set_flags");
    check(rt_set_globals(ctx(), &g_globals), "rt_set_globals");
    check(rt_dispatch_rays(ctx()), "rt_dispatch_rays");
    RtGlobals cur; check(rt_get_globals(ctx(), &cur), "rt_get_globals");
    g_globals.accumulator_count = cur.accumulator_count;                         // raytracing.cpp:744
    g_globals.translucent_accumulator_count = cur.translucent_accumulator_count; // raytracing.cpp:745
}

} // namespace Raytracing
