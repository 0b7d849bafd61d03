// rt_render -- headless replacement of the reference's frame loop (src/main.cpp:146-792): builds the Cornell scene
// exactly as main.cpp:243-361, applies the defaults of main.cpp:411-422 and the reset camera of :496-505, runs N
// progressive frames with the per-frame reseed of :443-454 and the accumulator logic of :632-658, and writes the
// render target (PPM) and the accumulator (PFM) where the reference writes a PNG (:677-757).
//
//   rt_render [--data DIR] [--width W] [--height H] [--frames N] [--spp S] [--bounces B] [--radius R]
//             [--tabulated SCALE] [--lambert-boxes] [--double-dispatch] [--seed X] [--out PREFIX]
#include <cstring>
#include <string>
#include <vector>

#include "raytracing.hpp"

static std::vector<float> load_f32(const std::string& path) {
    std::vector<float> v;
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", path.c_str()); exit(2); }
    float buf[1024]; size_t n;
    while ((n = fread(buf, 4, 1024, f)) > 0) v.insert(v.end(), buf, buf + n);
    fclose(f);
    return v;
}

int main(int argc, char** argv) {
    std::string data = "raytracer_amd/data", out = "render";
    uint32_t width = 1000, height = 955, frames = 16, spp = 1, bounces = 4, seed = 0x12345678; // main.cpp:162-163,413-414
    float radius = 0.025f, tabulated = 0.0f;                                                   // main.cpp:50,418
    float hg_limit = -1.0f; // >= 0: hierarchical BSSRDF gather with this solid-angle limit (rt_set_gather_mode; not in the reference)
    bool lambert_boxes = false, double_dispatch = false;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto next = [&]() -> const char* { if (i + 1 >= argc) { fprintf(stderr, "%s needs a value\n", a.c_str()); exit(2); } return argv[++i]; };
        if (a == "--data") data = next(); else if (a == "--out") out = next();
        else if (a == "--width") width = (uint32_t) atoi(next()); else if (a == "--height") height = (uint32_t) atoi(next());
        else if (a == "--frames") frames = (uint32_t) atoi(next()); else if (a == "--spp") spp = (uint32_t) atoi(next());
        else if (a == "--bounces") bounces = (uint32_t) atoi(next()); else if (a == "--radius") radius = (float) atof(next());
        else if (a == "--tabulated") tabulated = (float) atof(next()); else if (a == "--seed") seed = (uint32_t) strtoul(next(), nullptr, 0);
        else if (a == "--hierarchical-gather") hg_limit = (float) atof(next());
        else if (a == "--lambert-boxes") lambert_boxes = true; else if (a == "--double-dispatch") double_dispatch = true;
        else { fprintf(stderr, "unknown option %s\n", a.c_str()); return 2; }
    }
    std::vector<float> skin0 = load_f32(data + "/skin_0.f32"), skin1 = load_f32(data + "/skin_1.f32");
    Raytracing::init(0, skin0.data(), skin1.data(), (uint32_t) (skin0.size() / 3));
    Raytracing::update_resolution(width, height);
    if (hg_limit >= 0.0f) Raytracing::check(rt_set_gather_mode(Raytracing::ctx(), 1, hg_limit), "rt_set_gather_mode");

    // SCENE CREATION, main.cpp:240-361
    float cornell_aabb[6]; rt_aabb_null(cornell_aabb);
    struct Part { std::vector<const char*> files; float color[3]; Shader shader; };
    Shader box = lambert_boxes ? Lambert : Translucent;
    std::vector<Part> parts = {
        { { "floor.obj", "back.obj", "ceiling.obj" }, { 0.9f, 0.9f, 0.9f }, Lambert },
        { { "redwall.obj" }, { 0.9f, 0.0f, 0.0f }, Lambert },
        { { "greenwall.obj" }, { 0.0f, 0.9f, 0.0f }, Lambert },
        { { "luminaire.obj" }, { 0.0f, 0.0f, 0.0f }, Light },
        { { "largebox.obj" }, { 0.9f, 0.9f, 0.9f }, box },
        { { "smallbox.obj" }, { 0.9f, 0.9f, 0.9f }, box },
    };
    std::vector<rt_mesh*> meshes;
    std::vector<GeometryInstance> geometries;
    for (auto& p : parts) {
        rt_mesh* m = rt_mesh_create();
        for (const char* f : p.files) {
            std::string path = data + "/cornell/" + f;
            if (rt_parse_obj_file(path.c_str(), 1, m, cornell_aabb) != RT_OK) { fprintf(stderr, "cannot parse %s\n", path.c_str()); return 2; }
        }
        meshes.push_back(m);
        GeometryInstance g;
        g.vertices = { rt_mesh_vertices(m), rt_mesh_vertex_count(m) };
        g.indices = { rt_mesh_indices(m), rt_mesh_index_count(m) };
        g.material.shader = p.shader;
        memcpy(g.material.color, p.color, 12);
        geometries.push_back(g);
    }
    Blas cornell_blas = Raytracing::build_blas({ geometries.data(), geometries.size() });
    for (rt_mesh* m : meshes) rt_mesh_destroy(m);   // host arrays are only borrowed by build_blas (main.cpp:344-346)
    BlasInstance instance; instance.blas = &cornell_blas;
    rt_normalizing_transform(cornell_aabb, instance.transform);   // main.cpp:352-358
    Raytracing::build_tlas({ &instance, 1 });

    // globals, main.cpp:411-422 and the reset camera :496-505
    uint32_t stride = Raytracing::g_globals.translucent_instance_stride;
    rt_default_globals((float) width / (float) height, &Raytracing::g_globals);
    Raytracing::g_globals.translucent_instance_stride = stride;
    Raytracing::g_globals.frame_rng = seed;
    Raytracing::g_globals.samples_per_pixel = spp;
    Raytracing::g_globals.bounces_per_sample = bounces;
    Raytracing::g_globals.translucent_bssrdf_scale = tabulated;
    Raytracing::g_reference_double_dispatch = double_dispatch;

    // MAIN LOOP, main.cpp:427-792
    uint32_t total_points = 0;
    for (uint32_t frame = 0; frame < frames; frame++) {
        Raytracing::g_globals.frame_rng = rt_next_frame_rng(Raytracing::g_globals.frame_rng);   // :443-454
        if (frame == 0) {                                                                        // :632-658
            total_points = Raytracing::generate_translucent_samples(radius);
            Raytracing::g_globals.translucent_accumulator_count = 0;
            Raytracing::g_globals.accumulator_count = 0;
        }
        Raytracing::dispatch_rays();                                                             // :669
    }
    std::vector<uint8_t> rgba8((size_t) width * height * 4);
    std::vector<float> acc((size_t) width * height * 4);
    Raytracing::check(rt_read_render_target(Raytracing::ctx(), rgba8.data()), "rt_read_render_target");
    Raytracing::check(rt_read_accumulator(Raytracing::ctx(), acc.data()), "rt_read_accumulator");
    RtStats st; Raytracing::check(rt_get_stats(Raytracing::ctx(), &st), "rt_get_stats");
    // file name carries the parameters like the reference's capture (main.cpp:740-750)
    char name[512];
    snprintf(name, sizeof name, "%s_%ux%u_f%u_spp%u_b%u_pts%u", out.c_str(), width, height, frames, spp, bounces, total_points);
    rt_write_png((std::string(name) + ".png").c_str(), rgba8.data(), width, height);                 // main.cpp:750
    rt_write_ppm((std::string(name) + ".ppm").c_str(), rgba8.data(), width, height);
    rt_write_pfm((std::string(name) + ".pfm").c_str(), acc.data(), width, height, 1.0f / (float) frames);
    printf("%s: %u frames, %u sample points, %llu camera rays, %llu probe rays, %llu gather pairs\n", name, frames, total_points,
           (unsigned long long) st.rays_camera, (unsigned long long) st.rays_probe, (unsigned long long) st.gather_pairs);
    Raytracing::shutdown();
    return 0;
}
