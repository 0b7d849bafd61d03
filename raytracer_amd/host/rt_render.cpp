// rt_render -- headless replacement of the reference's frame loop (src/main.cpp:146-792): takes one of the scenes of
// rt_scene_create (the Cornell box as shipped, main.cpp:243-361, or a BASELINE.md configuration c1..c5), applies the
// defaults of main.cpp:411-422 and the reset camera of :496-505, and runs N progressive frames with the per-frame
// reseed of :443-454 and the regenerate / reset sequencing of :632-658.  What the reference's debug menu changes from
// frame to frame is scripted with --at (each key raises the same reset flag as its slider, main.cpp:538-627).  Writes the
// render target (PNG / PPM) and the accumulator (PFM) where the reference writes a PNG (:677-757).
//
//   rt_render [--scene cornell|c1|c2|c3|c4|c5] [--data DIR] [--width W] [--height H] [--frames N] [--spp S] [--bounces B]
//             [--radius R] [--tabulated SCALE] [--lambert-boxes] [--profile 0|1] [--instances N] [--double-dispatch]
//             [--seed X] [--at FRAME:KEY=VALUE]... [--pfm mean|raw] [--out PREFIX]
//   --at keys: radius (regenerate the sample points), spp, bounces, emission-bounces, sss, collect, tabulated, fudge
//              (reset the image), light, eta, reset-samples (reset sample points and image), reset (the image)
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

struct Event { uint32_t frame; std::string key; double value; };

int main(int argc, char** argv) {
    std::string data = "raytracer_amd/data", out = "render", scene_name = "cornell", pfm = "mean";
    uint32_t width = 1000, height = 955, frames = 16, seed = 0x12345678; // main.cpp:162-163
    RtSceneOptions opt; memset(&opt, 0, sizeof opt);
    float tabulated = -1.0f;
    float hg_limit = -1.0f; // >= 0: hierarchical BSSRDF gather with this solid-angle limit (rt_set_gather_mode; not in the reference)
    bool lambert_boxes = false, double_dispatch = false, seed_set = false;
    std::vector<Event> events;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto next = [&]() -> const char* { if (i + 1 >= argc) { fprintf(stderr, "%s needs a value\n", a.c_str()); exit(2); } return argv[++i]; };
        if (a == "--data") data = next(); else if (a == "--out") out = next(); else if (a == "--scene") scene_name = next();
        else if (a == "--width") width = (uint32_t) atoi(next()); else if (a == "--height") height = (uint32_t) atoi(next());
        else if (a == "--frames") frames = (uint32_t) atoi(next()); else if (a == "--spp") opt.spp = (uint32_t) atoi(next());
        else if (a == "--bounces") { opt.bounces = (uint32_t) atoi(next()); opt.bounces_set = 1; }
        else if (a == "--radius") opt.sample_radius = (float) atof(next());
        else if (a == "--tabulated") tabulated = (float) atof(next());
        else if (a == "--seed") { seed = (uint32_t) strtoul(next(), nullptr, 0); seed_set = true; }
        else if (a == "--profile") opt.bssrdf_profile = atoi(next()); else if (a == "--instances") opt.instances = (uint32_t) atoi(next());
        else if (a == "--hierarchical-gather") hg_limit = (float) atof(next());
        else if (a == "--lambert-boxes") lambert_boxes = true; else if (a == "--double-dispatch") double_dispatch = true;
        else if (a == "--pfm") pfm = next();
        else if (a == "--at") {
            std::string e = next();
            size_t colon = e.find(':'), eq = e.find('=');
            if (colon == std::string::npos) { fprintf(stderr, "--at wants FRAME:KEY[=VALUE], got %s\n", e.c_str()); return 2; }
            Event ev; ev.frame = (uint32_t) atoi(e.substr(0, colon).c_str());
            ev.key = e.substr(colon + 1, eq == std::string::npos ? std::string::npos : eq - colon - 1);
            ev.value = eq == std::string::npos ? 0.0 : atof(e.substr(eq + 1).c_str());
            events.push_back(ev);
        }
        else { fprintf(stderr, "unknown option %s\n", a.c_str()); return 2; }
    }
    // "cornell" = the scene as shipped: c3 with translucent boxes (dipole), 1 spp (main.cpp:311-340,413,418)
    const bool shipped = scene_name == "cornell";
    if (shipped) { opt.translucent_boxes = lambert_boxes ? 0 : 1; if (!opt.spp) opt.spp = 1; }
    rt_scene* scene = rt_scene_create(shipped ? "c3" : scene_name.c_str(), data.c_str(), width, height, &opt);
    if (!scene) return 2;

    std::vector<float> skin0 = load_f32(data + "/skin_0.f32"), skin1 = load_f32(data + "/skin_1.f32");
    Raytracing::init(0, skin0.data(), skin1.data(), (uint32_t) (skin0.size() / 3));
    Raytracing::update_resolution(width, height);
    if (hg_limit >= 0.0f) Raytracing::check(rt_set_gather_mode(Raytracing::ctx(), 1, hg_limit), "rt_set_gather_mode");
    Raytracing::check(rt_set_bssrdf_profile(Raytracing::ctx(), rt_scene_bssrdf_profile(scene)), "rt_set_bssrdf_profile");

    // SCENE CREATION, main.cpp:240-408: one build_blas per BLAS of the scene, then build_tlas over the instances
    std::vector<Blas> blas((size_t) rt_scene_blas_count(scene));
    for (int b = 0; b < (int) blas.size(); b++) {
        std::vector<GeometryInstance> geometries((size_t) rt_scene_geometry_count(scene, b));
        for (int k = 0; k < (int) geometries.size(); k++) {
            RtGeometry g; Raytracing::check(rt_scene_geometry(scene, b, k, &g), "rt_scene_geometry");
            geometries[k].vertices = { g.vertices, g.vertex_count };
            geometries[k].indices = { g.indices, g.index_count };
            geometries[k].material.shader = (Shader) g.shader;
            memcpy(geometries[k].material.color, g.color, 12);
        }
        blas[b] = Raytracing::build_blas({ geometries.data(), geometries.size() });
    }
    std::vector<BlasInstance> instances(rt_scene_instance_count(scene));
    for (size_t i = 0; i < instances.size(); i++) {
        const RtInstance& in = rt_scene_instances(scene)[i];
        memcpy(instances[i].transform, in.transform, 64);
        instances[i].blas = &blas[(size_t) in.blas];
    }
    Raytracing::build_tlas({ instances.data(), instances.size() });

    // globals, main.cpp:411-422 and the reset camera :496-505 (in the scene description), then the command line
    const uint32_t stride = Raytracing::g_globals.translucent_instance_stride;
    Raytracing::g_globals = *rt_scene_globals(scene);
    Raytracing::g_globals.translucent_instance_stride = stride;
    if (seed_set || Raytracing::g_globals.frame_rng == 0) Raytracing::g_globals.frame_rng = seed; // the reference: GetTickCount64(), main.cpp:411
    if (tabulated >= 0.0f) Raytracing::g_globals.translucent_bssrdf_scale = tabulated;
    Raytracing::g_reference_double_dispatch = double_dispatch;
    uint32_t rng_seed = 1, phase_seed = 2; rt_scene_seeds(scene, &rng_seed, &phase_seed);

    // MAIN LOOP, main.cpp:427-792.  The three flags of main.cpp:46-52 start as the reference's statics do: sample points
    // are generated before the first frame (g_do_regenerate_translucent_samples = 0.025f), the camera reset of frame 0
    // resets the image.
    float do_regenerate = rt_scene_sample_radius(scene), sample_points_radius = 0.0f;
    bool do_reset_translucent = false, do_reset_accumulator = true;
    uint32_t total_points = 0, regenerations = 0, image_resets = 0;
    for (uint32_t frame = 0; frame < frames; frame++) {
        Raytracing::g_globals.frame_rng = rt_next_frame_rng(Raytracing::g_globals.frame_rng);   // :443-454
        for (const Event& e : events) {                                                          // UPDATE, :486-627
            if (e.frame != frame) continue;
            RtGlobals& g = Raytracing::g_globals;
            if (e.key == "radius") do_regenerate = (float) e.value;                                                  // :595
            else if (e.key == "spp") { g.samples_per_pixel = (uint32_t) e.value; do_reset_accumulator = true; }      // :621
            else if (e.key == "bounces") { g.bounces_per_sample = (uint32_t) e.value; do_reset_accumulator = true; } // :622
            else if (e.key == "emission-bounces") { g.translucent_emission_bounces = (int32_t) e.value; do_reset_accumulator = true; } // :555
            else if (e.key == "sss") { Raytracing::g_enable_subsurface_scattering = e.value != 0; do_reset_accumulator = true; }        // :553
            else if (e.key == "collect") { Raytracing::g_enable_translucent_sample_collection = e.value != 0; do_reset_accumulator = true; } // :590
            else if (e.key == "tabulated") { g.translucent_bssrdf_scale = (float) e.value; do_reset_accumulator = true; }            // :559-566
            else if (e.key == "fudge") { g.translucent_bssrdf_fudge = (float) e.value; do_reset_accumulator = true; }                // :567
            else if (e.key == "light") { for (int k = 0; k < 3; k++) g.light_color[k] = (float) e.value; do_reset_translucent = true; } // :544-547
            else if (e.key == "eta") { g.translucent_refractive_index = (float) e.value; do_reset_translucent = true; }              // :562
            else if (e.key == "reset-samples") do_reset_translucent = true;                                                           // :596
            else if (e.key == "reset") do_reset_accumulator = true;                                                                   // :764
            else { fprintf(stderr, "--at: unknown key %s\n", e.key.c_str()); return 2; }
        }
        // PRE-RENDER, :629-658
        if (do_regenerate != 0.0f) {                                                             // :632-646
            total_points = Raytracing::generate_translucent_samples(do_regenerate, rng_seed, phase_seed);
            sample_points_radius = do_regenerate;
            do_regenerate = 0.0f;
            do_reset_translucent = true;
            regenerations++;
        }
        if (do_reset_translucent) {                                                              // :649-653
            do_reset_accumulator = true;
            Raytracing::g_globals.translucent_accumulator_count = 0;
            do_reset_translucent = false;
        }
        if (do_reset_accumulator) {                                                              // :655-658
            Raytracing::g_globals.accumulator_count = 0;
            do_reset_accumulator = false;
            image_resets++;
        }
        Raytracing::dispatch_rays();                                                             // RENDER, :669
    }
    const uint32_t accumulated = Raytracing::g_globals.accumulator_count;
    std::vector<uint8_t> rgba8((size_t) width * height * 4);
    std::vector<float> acc((size_t) width * height * 4);
    Raytracing::check(rt_read_render_target(Raytracing::ctx(), rgba8.data()), "rt_read_render_target");
    Raytracing::check(rt_read_accumulator(Raytracing::ctx(), acc.data()), "rt_read_accumulator");
    RtStats st; Raytracing::check(rt_get_stats(Raytracing::ctx(), &st), "rt_get_stats");
    // the file name carries the parameters like the reference's capture (main.cpp:740-750): scene, radius, accumulated samples, size
    char name[512];
    snprintf(name, sizeof name, "%s_%s-r%.3e-#%u-%ux%u", out.c_str(), scene_name.c_str(), sample_points_radius, accumulated, width, height);
    rt_write_png((std::string(name) + ".png").c_str(), rgba8.data(), width, height);                 // main.cpp:750
    rt_write_ppm((std::string(name) + ".ppm").c_str(), rgba8.data(), width, height);
    // accumulator: the mean over the accumulated frames, or as it stands (the sum) with --pfm raw
    rt_write_pfm((std::string(name) + ".pfm").c_str(), acc.data(), width, height, pfm == "raw" ? 1.0f : 1.0f / (float) (accumulated ? accumulated : 1u));
    printf("%s: %u frames (%u accumulated, %u image resets, %u regenerations), %u sample points, %llu camera rays, %llu probe rays, %llu gather pairs\n",
           name, frames, accumulated, image_resets, regenerations, total_points,
           (unsigned long long) st.rays_camera, (unsigned long long) st.rays_probe, (unsigned long long) st.gather_pairs);
    Raytracing::shutdown();
    rt_scene_destroy(scene);
    return 0;
}
