/* rt_host.h -- C-ABI of librt_host.so: the CPU-side scene layer of gluyas/raytracer restated without
 * Win32 / DirectXMath.  Pure host code, no GPU calls.
 *
 * Replaces: src/parse_obj.cpp:28-176, src/geometry.cpp:21-61, the scene assembly and camera of
 * src/main.cpp:243-361,443-454,486-536, and the PNG capture step main.cpp:677-757 (as PPM/PFM writers).
 */
#ifndef RT_HOST_H
#define RT_HOST_H

#include <stddef.h>
#include <stdint.h>
#include "rt_abi.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Growable vertex/index arrays: Array<Vertex> + Array<Index> of parse_obj.cpp:28. */
typedef struct rt_mesh rt_mesh;
rt_mesh*        rt_mesh_create(void);
void            rt_mesh_destroy(rt_mesh* m);
size_t          rt_mesh_vertex_count(const rt_mesh* m);
size_t          rt_mesh_index_count(const rt_mesh* m);
const RtVertex* rt_mesh_vertices(const rt_mesh* m);
const RtIndex*  rt_mesh_indices(const rt_mesh* m);
/* append a vertex / a triangle (for procedurally built quads) */
int             rt_mesh_push_vertex(rt_mesh* m, const float position[3], const float normal[3]);
int             rt_mesh_push_triangle(rt_mesh* m, uint32_t a, uint32_t b, uint32_t c);

/* parse_obj_file (parse_obj.cpp:28-176): APPENDS to the mesh; v/vt/vn, triangle and quad faces (quad -> 0 1 2,
 * 0 2 3), no vertex de-duplication, face normals when the face has no vn, convert_to_rhs swaps y/z and negates x.
 * aabb (min xyz, max xyz) is extended when non-NULL; initialise it with rt_aabb_null.  Returns RT_OK or
 * RT_ERR_INVALID (file missing / malformed face), where the reference asserts. */
int  rt_parse_obj_file(const char* filename, int convert_to_rhs, rt_mesh* mesh, float aabb[6]);

/* geometry.cpp */
void  rt_aabb_null(float aabb[6]);                                     /* AABB_NULL, geometry.h:21 */
float rt_aabb_widest(const float aabb[6]);                             /* geometry.cpp:57-61 */
float rt_triangle_area(const float a[3], const float b[3], const float c[3]); /* geometry.cpp:21-25 */

/* main.cpp:352-358: uniform scale 1/widest and centre at the origin (XMMatrixAffineTransformation). */
void rt_normalizing_transform(const float aabb[6], float out16[16]);
/* general affine: scale * rotation about Z (radians) * translation, row-vector convention */
void rt_affine_transform(float scale, float rot_z, const float translation[3], float out16[16]);

/* main.cpp:525-535: orbit camera -> camera_to_world, aspect, focal length.  Angles in radians. */
void rt_camera_orbit(float azimuth, float elevation, float distance, const float target[3], float fov_y,
                     float aspect, RtGlobals* g);
/* main.cpp:411-422 defaults (spp 1, bounces 4, emission bounces 1, dipole sigma_s 15, sigma_a 0.1, eta 1.75,
 * light 50) and the reset camera of main.cpp:496-505. */
void rt_default_globals(float aspect, RtGlobals* g);
/* main.cpp:443-454: frame_rng <- wang_hash(frame_rng) */
uint32_t rt_next_frame_rng(uint32_t frame_rng);

/* ---- scene assembly (SCENE CREATION of main.cpp:240-361, generalised to the five configurations of BASELINE.md) ------
 * "c1" debug_cube.obj 256x256-style test scene, "c2" bunny on a floor under a luminaire, "c3" the Cornell box of
 * main.cpp:243-361 (boxes Lambert, or Translucent = as shipped), "c4" Cornell walls + translucent bunny (tabulated
 * BSSRDF), "c5" the instanced bunny field.  The scene owns its meshes; RtGeometry pointers stay valid until
 * rt_scene_destroy (rt_build_blas only borrows them, main.cpp:344-346).  RtInstance.blas is an index into the scene's
 * BLAS list: the caller replaces it with the handle rt_build_blas returned for that list entry. */
typedef struct RtSceneOptions {        /* all zero = the configuration's defaults */
    uint32_t spp;                      /* samples_per_pixel; 0 = default (c3: 64, c4: 16, else 1)            */
    uint32_t bounces; int32_t bounces_set; /* bounces_per_sample when bounces_set (default c1, c2: 1, else 4) */
    int32_t  translucent_boxes;        /* c3: boxes as Shader::Translucent, dipole BSSRDF (main.cpp:311-340,418-422) */
    int32_t  bssrdf_profile;           /* c4: 0 = skin_0, 1 = skin_1                                          */
    int32_t  dipole;                   /* c4: dipole instead of the tabulated profile                         */
    float    sample_radius;            /* c3 translucent / c4: blue-noise radius; 0 = 0.025 (main.cpp:50)     */
    float    bunny_scale;              /* c4: 0 = 0.6                                                         */
    uint32_t instances, grid;          /* c5: 0 = 52,743 instances on a 230 x 230 lattice                     */
} RtSceneOptions;
typedef struct rt_scene rt_scene;
rt_scene*         rt_scene_create(const char* config, const char* data_dir, uint32_t width, uint32_t height, const RtSceneOptions* options);
void              rt_scene_destroy(rt_scene* s);
int               rt_scene_blas_count(const rt_scene* s);
int               rt_scene_geometry_count(const rt_scene* s, int blas);
int               rt_scene_geometry(const rt_scene* s, int blas, int geometry, RtGeometry* out);
size_t            rt_scene_instance_count(const rt_scene* s);
const RtInstance* rt_scene_instances(const rt_scene* s);
const RtGlobals*  rt_scene_globals(const rt_scene* s);      /* defaults of main.cpp:411-422 + the configuration's settings */
uint32_t          rt_scene_flags(const rt_scene* s);        /* rt_set_flags bits */
float             rt_scene_sample_radius(const rt_scene* s);/* > 0: call rt_generate_translucent_samples with it */
int               rt_scene_bssrdf_profile(const rt_scene* s);
void              rt_scene_seeds(const rt_scene* s, uint32_t* rng_seed, uint32_t* phase_seed);

/* image output (stands in for stbi_write_png, main.cpp:750) */
int rt_write_png(const char* filename, const uint8_t* rgba8, uint32_t width, uint32_t height); /* 8-bit RGBA like main.cpp:750 */
int rt_write_ppm(const char* filename, const uint8_t* rgba8, uint32_t width, uint32_t height);
int rt_write_pfm(const char* filename, const float* rgba, uint32_t width, uint32_t height, float scale);

#ifdef __cplusplus
}
#endif
#endif /* RT_HOST_H */
