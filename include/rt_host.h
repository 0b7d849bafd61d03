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

/* image output (stands in for stbi_write_png, main.cpp:750) */
int rt_write_png(const char* filename, const uint8_t* rgba8, uint32_t width, uint32_t height); /* 8-bit RGBA like main.cpp:750 */
int rt_write_ppm(const char* filename, const uint8_t* rgba8, uint32_t width, uint32_t height);
int rt_write_pfm(const char* filename, const float* rgba, uint32_t width, uint32_t height, float scale);

#ifdef __cplusplus
}
#endif
#endif /* RT_HOST_H */
