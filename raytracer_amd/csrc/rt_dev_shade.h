// rt_dev_shade.h -- closest-hit / miss shading of one ray (raytracing.hlsl:60-67,170-207,305-349) and the
// camera ray generator (raytracing.hlsl:139-149), as device functions shared by the wavefront kernels.
#pragma once
#include "rt_dev_scene.h"
#include "../../include/rt_abi.h"

namespace rtd {

// 32 B per geometry of a BLAS: the shader record of raytracing.h:11-17 without the D3D identifiers
struct GeomRec {
    u32   index_offset;   // first index of the geometry in the BLAS index buffer (elements)
    u32   first_tri;      // index_offset / 3
    int   shader;         // RT_SHADER_*
    int   translucent_id; // RaytracingLocals.translucent_id, -1 if not translucent
    float color[3];       // RaytracingLocals.color
    u32   _pad;
};

// per instance, shading side
struct InstShade {
    float object_to_world[16];
    const RtVertex* vertices;  // BLAS vertex buffer (concatenated geometries)
    const uint16_t* indices;   // BLAS index buffer, rebased 16-bit (raytracing.cpp:360-366)
    const GeomRec*  geoms;
    const TriRec*   tris;
    u32 instance_id;           // InstanceID()
    u32 vertex_count;          // of `vertices`
};

// one translucent instance's slice of the concatenated sample-point arrays
struct TranslucentSlice {
    u32 offset, count;
    u32 slot;                  // translucent_id * stride + instance_id
    float mean_area;           // TranslucentProperties.samples_mean_area
};

struct ShadeScene {
    const InstShade* insts;
    const int* slot_to_ti;     // slot -> translucent instance index or -1
    u32 slot_count;
    const TranslucentSlice* slices;
};

struct ShadeOut {
    f3 scatter, reflectance, emission;
    float t;
    int geometry, primitive;
    // translucent emission is evaluated by the gather kernel (second hot loop); these feed it
    bool  gather;
    int   ti;                  // translucent instance index
    f3    hit_point;
    float transmitted_fresnel;
};

// raytracing.hlsl:60-67 + prelude.h:195-246
RT_DEV f3 world_space_normal(const InstShade& in, u32 prim_global, float bu, float bv, f3 world_dir) {
    const uint16_t* idx = in.indices + 3u * prim_global;
    u32 i0 = idx[0], i1 = idx[1], i2 = idx[2];
    const RtVertex& v0 = in.vertices[i0];
    const RtVertex& v1 = in.vertices[i1];
    const RtVertex& v2 = in.vertices[i2];
    f3 n0 = mk3(v0.normal[0], v0.normal[1], v0.normal[2]);
    f3 n1 = mk3(v1.normal[0], v1.normal[1], v1.normal[2]);
    f3 n2 = mk3(v2.normal[0], v2.normal[1], v2.normal[2]);
    f3 normal = n0;
    normal = normal + bu * (n1 - n0);
    normal = normal + bv * (n2 - n0);
    normal = xform_dir44(in.object_to_world, normal);
    normal = normal * (-sign_f32(dot3(world_dir, normal)));
    return normalize3(normal);
}

// payload_t_in: the bounce index (or +inf when translucent emission is ignored), raytracing.hlsl:102-107
// emission_only: the caller uses nothing but r.emission and the RNG state (the path ends here whatever the surface
// does): a Lambert hit, whose emission is zero, then skips the normal (three vertex fetches) and the scatter direction,
// and only advances the RNG by the two draws of random_on_sphere
RT_DEV void shade_ray(const ShadeScene& sc, const RtGlobals& g, float fudge, f3 o, f3 d, const HitOut& h,
                      float payload_t_in, u32& rng, ShadeOut& r, bool emission_only = false) {
    r.gather = false; r.ti = -1; r.transmitted_fresnel = 0.0f; r.hit_point = mk3(0.0f, 0.0f, 0.0f);
    r.geometry = -1; r.primitive = -1;
    if (h.inst < 0) { // miss, raytracing.hlsl:201-207
        r.scatter = mk3(0.0f, 0.0f, 0.0f); r.reflectance = r.scatter; r.emission = r.scatter; r.t = inf_f32();
        return;
    }
    const InstShade& in = sc.insts[h.inst];
    const TriRec& tr = in.tris[h.tri];
    const GeomRec& geo = in.geoms[tr.geometry];
    r.geometry = (int) tr.geometry; r.primitive = (int) tr.prim_local;
    f3 color = mk3(geo.color[0], geo.color[1], geo.color[2]);
    r.t = h.t;
    if (geo.shader == RT_SHADER_LAMBERT && emission_only) {
        r.scatter = mk3(0.0f, 0.0f, 0.0f); r.reflectance = r.scatter; r.emission = r.scatter;
        (void) random01(rng); (void) random11(rng); // random.hlsl:70-75: the stream goes on into the pixel's next sample
        return;
    }
    f3 normal = world_space_normal(in, tr.prim_global, h.u, h.v, d);
    if (geo.shader == RT_SHADER_LAMBERT) { // raytracing.hlsl:170-179
        r.scatter = random_on_hemisphere(rng, normal);
        r.reflectance = color * dot3(normal, r.scatter);
        r.emission = mk3(0.0f, 0.0f, 0.0f);
    } else if (geo.shader == RT_SHADER_LIGHT) { // raytracing.hlsl:186-199
        f3 c = any3(color) ? color : mk3(g.light_color[0], g.light_color[1], g.light_color[2]);
        r.scatter = mk3(0.0f, 0.0f, 0.0f);
        r.reflectance = r.scatter;
        r.emission = c * (-dot3(normal, d));
    } else { // translucent, raytracing.hlsl:305-349
        u32 slot = (u32) geo.translucent_id * g.translucent_instance_stride + in.instance_id;
        r.hit_point = o + h.t * d;
        if (payload_t_in <= (float) g.translucent_emission_bounces && fudge != 0.0f) {
            int ti = slot < sc.slot_count ? sc.slot_to_ti[slot] : -1;
            if (ti >= 0 && sc.slices[ti].count > 0) { r.gather = true; r.ti = ti; }
        }
        float n = g.translucent_refractive_index;
        f3 scatter = random_on_hemisphere(rng, normal);
        float lambert = dot3(scatter, normal);
        float incident_fresnel = schlick(n, lambert);
        float transmitted_cosine = sqrtf(1.0f - 1.0f / (n * n) * (1.0f - (-dot3(d, normal))));
        r.transmitted_fresnel = 1.0f - schlick(n, transmitted_cosine);
        r.scatter = scatter;
        r.reflectance = color * lambert * incident_fresnel;
        // diffuse_irradiance == 0 here; when r.gather the gather kernel adds the real term instead
        r.emission = mk3(0.0f, 0.0f, 0.0f) * r.transmitted_fresnel / (RT_TAU_F / 2.0f);
    }
}

// camera ray for pixel (px, py), raytracing.hlsl:141-149; draws two random11 (x first, Q6)
// `camera_origin` = camera_to_world[3].xyz / camera_to_world[3].w: the same for every pixel, so the host divides
// (IEEE division, same quotient) instead of every lane
// `inv_size` = { 1 / width, 1 / height } as the host's IEEE quotients: x / width is then x * inv + one residual correction
// (quotient_by_reciprocal, rt_dev_math.h), the correctly rounded quotient in 3 instructions instead of 11
RT_DEV void camera_ray(const RtGlobals& g, const float* camera_origin, const float* inv_size, u32 px, u32 py, u32 width, u32 height, u32& rng, f3& origin, f3& direction) {
    const float* m = g.camera_to_world;
    origin = mk3(camera_origin[0], camera_origin[1], camera_origin[2]);
    float dx = (float) px + 0.5f, dy = (float) py + 0.5f;
    float jx = random11(rng);
    float jy = random11(rng);
    dx = dx + 0.5f * jx; dy = dy + 0.5f * jy;
    dx = quotient_by_reciprocal(2.0f * dx, (float) width, inv_size[0]) - 1.0f;
    dy = quotient_by_reciprocal(2.0f * dy, (float) height, inv_size[1]) - 1.0f;
    dx = dx * g.camera_aspect;
    dy = dy * -1.0f;
    float dz = -g.camera_focal_length;
    direction = normalize3(xform_dir44(m, mk3(dx, dy, dz)));
}

} // namespace rtd
