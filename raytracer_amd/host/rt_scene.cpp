// rt_scene.cpp -- the benchmark / parity configurations C1..C5 of BASELINE.md as scene descriptions,
// assembled the way the reference's main.cpp assembles its Cornell box: parse_obj_file per part (main.cpp:249-340), one
// BLAS of several geometries, the normalising instance transform (main.cpp:349-360), the defaults of main.cpp:411-422 and
// the reset camera of :496-505.  Part of librt_host.so (include/rt_host.h): pure host code.  Both the headless frame loop
// (rt_render.cpp) and the Python scene module (raytracer_amd/scenes.py) take their scenes from here, so every consumer --
// the HIP library, and in the tests the CPU oracle -- is handed bit-identical inputs.
#include "../../include/rt_host.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace {

struct Geometry {
    rt_mesh* mesh;
    int32_t shader;
    float color[3];
};

} // namespace

struct rt_scene {
    std::string name;
    std::vector<std::vector<Geometry>> blas;
    std::vector<RtInstance> instances; // blas = index into `blas`
    RtGlobals globals;
    uint32_t flags = RT_FLAG_COLLECT_SAMPLES | RT_FLAG_ENABLE_SSS;
    float sample_radius = 0.0f; // > 0: the scene has translucent geometry, generate sample points with this radius
    uint32_t rng_seed = 1, phase_seed = 2;
    int32_t bssrdf_profile = 0;
    uint32_t width = 0, height = 0;
    std::string error;
    ~rt_scene() { for (auto& b : blas) for (auto& g : b) rt_mesh_destroy(g.mesh); }
};

namespace {

const float WHITE[3] = { 0.9f, 0.9f, 0.9f }, BLACK[3] = { 0.0f, 0.0f, 0.0f };

Geometry make_geometry(rt_mesh* mesh, int32_t shader, const float color[3]) {
    Geometry g; g.mesh = mesh; g.shader = shader; memcpy(g.color, color, 12);
    return g;
}

// parse_obj_file applied to each file in turn, appending to one vertex / index array (main.cpp:252-254)
rt_mesh* load(rt_scene* s, const std::string& dir, const std::vector<const char*>& files, int rhs, float aabb[6]) {
    rt_mesh* m = rt_mesh_create();
    for (const char* f : files) {
        const std::string path = dir + "/" + f;
        if (rt_parse_obj_file(path.c_str(), rhs, m, aabb) != RT_OK && s->error.empty()) s->error = "cannot parse " + path;
    }
    return m;
}

// two triangles (0 1 2, 0 2 3) with the face normal of the first, like a quad face in parse_obj.cpp:133-157
rt_mesh* quad(const float p0[3], const float p1[3], const float p2[3], const float p3[3]) {
    rt_mesh* m = rt_mesh_create();
    const float e1[3] = { p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2] }, e2[3] = { p2[0] - p0[0], p2[1] - p0[1], p2[2] - p0[2] };
    float n[3] = { e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0] };
    const float len = sqrtf(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
    for (int k = 0; k < 3; k++) n[k] = n[k] / len;
    rt_mesh_push_vertex(m, p0, n); rt_mesh_push_vertex(m, p1, n); rt_mesh_push_vertex(m, p2, n); rt_mesh_push_vertex(m, p3, n);
    rt_mesh_push_triangle(m, 0, 1, 2); rt_mesh_push_triangle(m, 0, 2, 3);
    return m;
}
rt_mesh* quad_z(float x0, float y0, float x1, float y1, float z, bool up) { // axis-aligned, facing +z (up) or -z
    const float a[3] = { x0, y0, z }, b[3] = { x1, y0, z }, c[3] = { x1, y1, z }, d[3] = { x0, y1, z };
    return up ? quad(a, b, c, d) : quad(a, d, c, b);
}

void identity(float m[16]) { memset(m, 0, 64); m[0] = m[5] = m[10] = m[15] = 1.0f; }

void add_instance(rt_scene* s, const float m[16], int blas) {
    RtInstance in; memcpy(in.transform, m, 64); in.blas = blas;
    s->instances.push_back(in);
}

// lowest z of a box under a transform without rotation: aabb_min.z * m[2][2] + m[3][2]
float floor_height(const float aabb[6], const float m[16]) { return aabb[2] * m[10] + m[14]; }

rt_mesh* load_bunny(rt_scene* s, const std::string& dir, float aabb[6]) {
    rt_aabb_null(aabb);
    return load(s, dir, { "bunny.obj" }, 1, aabb); // the mesh is Y-up: convert_to_rhs lands it in the Z-up world
}

// C1: data/debug_cube.obj (rhs = false, Lambert 0.9) scaled x0.4 at the origin, a floor and one emissive quad above
void build_c1(rt_scene* s, const std::string& dir) {
    float aabb[6]; rt_aabb_null(aabb);
    s->blas.push_back({ make_geometry(load(s, dir, { "debug_cube.obj" }, 0, aabb), RT_SHADER_LAMBERT, WHITE) });
    s->blas.push_back({ make_geometry(quad_z(-1.5f, -1.5f, 1.5f, 1.5f, -0.2f, true), RT_SHADER_LAMBERT, WHITE),
                        make_geometry(quad_z(-0.5f, -0.5f, 0.5f, 0.5f, 0.8f, false), RT_SHADER_LIGHT, BLACK) });
    float m[16]; const float origin[3] = { 0.0f, 0.0f, 0.0f };
    rt_affine_transform(0.4f, 0.0f, origin, m); add_instance(s, m, 0);
    identity(m); add_instance(s, m, 1);
}

// C2: bunny.obj normalised like main.cpp:352-358 (Lambert 0.9) on a floor quad under a luminaire quad
void build_c2(rt_scene* s, const std::string& dir) {
    float aabb[6];
    s->blas.push_back({ make_geometry(load_bunny(s, dir, aabb), RT_SHADER_LAMBERT, WHITE) });
    float m[16]; rt_normalizing_transform(aabb, m);
    const float zmin = floor_height(aabb, m);
    s->blas.push_back({ make_geometry(quad_z(-1.5f, -1.5f, 1.5f, 1.5f, zmin, true), RT_SHADER_LAMBERT, WHITE),
                        make_geometry(quad_z(-0.5f, -0.5f, 0.5f, 0.5f, 1.2f, false), RT_SHADER_LIGHT, BLACK) });
    add_instance(s, m, 0);
    identity(m); add_instance(s, m, 1);
}

// the Cornell box of main.cpp:243-340: one BLAS, geometries in the reference's order
std::vector<Geometry> cornell_geometries(rt_scene* s, const std::string& dir, int box_shader, bool with_boxes, float aabb[6]) {
    const std::string c = dir + "/cornell";
    const float red[3] = { 0.9f, 0.0f, 0.0f }, green[3] = { 0.0f, 0.9f, 0.0f };
    rt_aabb_null(aabb);
    std::vector<Geometry> g;
    g.push_back(make_geometry(load(s, c, { "floor.obj", "back.obj", "ceiling.obj" }, 1, aabb), RT_SHADER_LAMBERT, WHITE)); // :249-265
    g.push_back(make_geometry(load(s, c, { "redwall.obj" }, 1, aabb), RT_SHADER_LAMBERT, red));                            // :266-280
    g.push_back(make_geometry(load(s, c, { "greenwall.obj" }, 1, aabb), RT_SHADER_LAMBERT, green));                        // :281-295
    g.push_back(make_geometry(load(s, c, { "luminaire.obj" }, 1, aabb), RT_SHADER_LIGHT, BLACK));                          // :296-310
    if (with_boxes) {                                                                                                       // :311-340
        g.push_back(make_geometry(load(s, c, { "largebox.obj" }, 1, aabb), box_shader, WHITE));
        g.push_back(make_geometry(load(s, c, { "smallbox.obj" }, 1, aabb), box_shader, WHITE));
    }
    return g;
}

// C3: the Cornell box exactly as main.cpp:243-361; boxes Lambert (pure diffuse GI) unless translucent_boxes, which gives
// the scene as shipped (dipole BSSRDF, main.cpp:418-422)
void build_c3(rt_scene* s, const std::string& dir, const RtSceneOptions& o) {
    float aabb[6];
    s->blas.push_back(cornell_geometries(s, dir, o.translucent_boxes ? RT_SHADER_TRANSLUCENT : RT_SHADER_LAMBERT, true, aabb));
    float m[16]; rt_normalizing_transform(aabb, m); add_instance(s, m, 0);
    if (o.translucent_boxes) s->sample_radius = o.sample_radius > 0.0f ? o.sample_radius : 0.025f; // main.cpp:50
}

// C4: Cornell walls + a translucent bunny standing on the floor, tabulated skin profile (scale 0.4 main.cpp:558, fudge 1,
// eta 1.75)
void build_c4(rt_scene* s, const std::string& dir, const RtSceneOptions& o) {
    float aabb[6], baabb[6];
    s->blas.push_back(cornell_geometries(s, dir, RT_SHADER_LAMBERT, false, aabb));
    float room[16]; rt_normalizing_transform(aabb, room); add_instance(s, room, 0);
    s->blas.push_back({ make_geometry(load_bunny(s, dir, baabb), RT_SHADER_TRANSLUCENT, WHITE) });
    float m[16]; rt_normalizing_transform(baabb, m);
    const float scale = o.bunny_scale > 0.0f ? o.bunny_scale : 0.6f;
    for (int r = 0; r < 4; r++) for (int c = 0; c < 3; c++) m[4 * r + c] = m[4 * r + c] * scale;
    const float floor_z = floor_height(aabb, room), bunny_zmin = floor_height(baabb, m);
    m[14] = m[14] + (float) ((double) floor_z - (double) bunny_zmin + 0.002);
    add_instance(s, m, 1);
    s->globals.translucent_emission_bounces = 1;
    s->globals.translucent_bssrdf_scale = o.dipole ? 0.0f : 0.4f;
    s->globals.translucent_bssrdf_fudge = 1.0f;
    s->globals.translucent_refractive_index = 1.75f;
    s->sample_radius = o.sample_radius > 0.0f ? o.sample_radius : 0.025f;
    s->bssrdf_profile = o.bssrdf_profile;
}

uint32_t xorshift(uint32_t x) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; return x; }

// C5: `instances` bunnies on a grid x grid jittered lattice (seeded xorshift: uniform scale, random rotation about Z)
// over a floor quad under a large luminaire; 52,743 x 948 = 50.0 M instanced triangles at the default size
void build_c5(rt_scene* s, const std::string& dir, const RtSceneOptions& o) {
    const uint32_t grid = o.grid ? o.grid : 230u;
    uint32_t count = o.instances ? o.instances : 52743u;
    if (count > grid * grid) count = grid * grid;
    float aabb[6];
    s->blas.push_back({ make_geometry(load_bunny(s, dir, aabb), RT_SHADER_LAMBERT, WHITE) });
    float base[16]; rt_normalizing_transform(aabb, base);
    const float zmin = floor_height(aabb, base);
    const double spacing = 1.3, half = 0.5 * grid * spacing;
    const double lh = 0.75 * half + 1.0, top = 0.25 * half + 4.0;
    const float fe = (float) (half + 2.0);
    s->blas.push_back({ make_geometry(quad_z(-fe, -fe, fe, fe, 0.0f, true), RT_SHADER_LAMBERT, WHITE),
                        make_geometry(quad_z((float) -lh, (float) -lh, (float) lh, (float) lh, (float) top, false), RT_SHADER_LIGHT, BLACK) });
    uint32_t state = 0x2545F491u;
    s->instances.reserve(count + 1);
    for (uint32_t k = 0; k < count; k++) {
        const uint32_t gx = k % grid, gy = k / grid;
        state = xorshift(state); const double jx = (state & 0xFFFFu) / 65536.0 - 0.5;
        state = xorshift(state); const double jy = (state & 0xFFFFu) / 65536.0 - 0.5;
        state = xorshift(state); const double sc = 0.7 + 0.3 * ((state & 0xFFFFu) / 65536.0);
        state = xorshift(state); const double rot = 2.0 * 3.14159265358979323846 * ((state & 0xFFFFu) / 65536.0);
        const double x = (gx + 0.5) * spacing - half + 0.25 * spacing * jx;
        const double y = (gy + 0.5) * spacing - half + 0.25 * spacing * jy;
        const float t[3] = { (float) x, (float) y, (float) (-(double) zmin * sc) };
        float xf[16], m[16];
        rt_affine_transform((float) sc, (float) rot, t, xf);
        for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) { // base * xf (row vectors: the mesh is normalised first)
            float acc = base[4 * r] * xf[c];
            for (int j = 1; j < 4; j++) acc = acc + base[4 * r + j] * xf[4 * j + c];
            m[4 * r + c] = acc;
        }
        add_instance(s, m, 0);
    }
    float m[16]; identity(m); add_instance(s, m, 1);
    s->globals.frame_rng = 0x12345678u;
    const double dist = 0.9 * half > 6.0 ? 0.9 * half : 6.0;
    const float target[3] = { 0.0f, 0.0f, 0.3f };
    const double deg = 3.14159265358979323846 / 180.0;
    rt_camera_orbit(0.6f, (float) (25.0 * deg), (float) dist, target, (float) (40.0 * deg), (float) s->width / (float) s->height, &s->globals);
}

} // namespace

extern "C" {

rt_scene* rt_scene_create(const char* config, const char* data_dir, uint32_t width, uint32_t height, const RtSceneOptions* options) {
    if (!config || !data_dir || width == 0 || height == 0) return nullptr;
    RtSceneOptions o; memset(&o, 0, sizeof o);
    if (options) o = *options;
    rt_scene* s = new rt_scene();
    s->name = config; s->width = width; s->height = height;
    rt_default_globals((float) width / (float) height, &s->globals);
    const std::string dir = data_dir, c = config;
    uint32_t spp = 1, bounces = 4;
    if (c == "c1") { build_c1(s, dir); bounces = 1; }
    else if (c == "c2") { build_c2(s, dir); bounces = 1; }
    else if (c == "c3") { build_c3(s, dir, o); spp = o.translucent_boxes ? 1u : 64u; }
    else if (c == "c4") { build_c4(s, dir, o); spp = 16; }
    else if (c == "c5") build_c5(s, dir, o);
    else s->error = "unknown configuration " + c + " (c1 .. c5)";
    s->globals.samples_per_pixel = o.spp ? o.spp : spp;
    s->globals.bounces_per_sample = o.bounces_set ? o.bounces : bounces;
    if (!s->error.empty()) { fprintf(stderr, "rt_scene_create: %s\n", s->error.c_str()); delete s; return nullptr; }
    return s;
}
void rt_scene_destroy(rt_scene* s) { delete s; }

int rt_scene_blas_count(const rt_scene* s) { return (int) s->blas.size(); }
int rt_scene_geometry_count(const rt_scene* s, int blas) { return blas >= 0 && blas < (int) s->blas.size() ? (int) s->blas[blas].size() : 0; }
int rt_scene_geometry(const rt_scene* s, int blas, int geometry, RtGeometry* out) {
    if (!out || geometry < 0 || geometry >= rt_scene_geometry_count(s, blas)) return RT_ERR_INVALID;
    const Geometry& g = s->blas[blas][geometry];
    out->vertices = rt_mesh_vertices(g.mesh); out->vertex_count = rt_mesh_vertex_count(g.mesh);
    out->indices = rt_mesh_indices(g.mesh); out->index_count = rt_mesh_index_count(g.mesh);
    out->shader = g.shader; memcpy(out->color, g.color, 12);
    return RT_OK;
}
size_t rt_scene_instance_count(const rt_scene* s) { return s->instances.size(); }
const RtInstance* rt_scene_instances(const rt_scene* s) { return s->instances.data(); }
const RtGlobals* rt_scene_globals(const rt_scene* s) { return &s->globals; }
uint32_t rt_scene_flags(const rt_scene* s) { return s->flags; }
float rt_scene_sample_radius(const rt_scene* s) { return s->sample_radius; }
int rt_scene_bssrdf_profile(const rt_scene* s) { return s->bssrdf_profile; }
void rt_scene_seeds(const rt_scene* s, uint32_t* rng_seed, uint32_t* phase_seed) { if (rng_seed) *rng_seed = s->rng_seed; if (phase_seed) *phase_seed = s->phase_seed; }

} // extern "C"
