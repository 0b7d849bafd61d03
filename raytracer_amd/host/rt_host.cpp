// rt_host.cpp -- CPU-side scene layer (OBJ loading, AABB/area helpers, scene transform, camera, image dump).
// Restates src/parse_obj.cpp, src/geometry.cpp and the scene/camera parts of src/main.cpp of gluyas/raytracer
// without Win32 or DirectXMath; see include/rt_host.h for the per-function citations.
#include "../../include/rt_host.h"

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

struct rt_mesh {
    std::vector<RtVertex> vertices;
    std::vector<RtIndex>  indices;
};

namespace {

struct F3 { float x, y, z; };
struct F2 { float x, y; };

// parse_obj.cpp:3-26: up to max_count whitespace separated floats on the rest of the line
unsigned parse_floats(const char*& p, unsigned max_count, float* out) {
    unsigned i = 0;
    while (i < max_count) {
        while (*p && isspace((unsigned char) *p)) {
            if (*p == '\r' || *p == '\n') return i;
            p++;
        }
        if (!*p) return i;
        out[i++] = (float) atof(p);
        while (*p && !isspace((unsigned char) *p)) p++;
    }
    return i;
}

F3 sub(F3 a, F3 b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
F3 crossf(F3 a, F3 b) { return { a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x }; }
float dotf(F3 a, F3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
F3 normalizef(F3 a) {
    float len = sqrtf(dotf(a, a));
    if (len > 0) return { a.x / len, a.y / len, a.z / len };
    return { 0, 0, 0 };
}

void aabb_join(float aabb[6], const float p[3]) { // geometry.cpp:40-44
    for (int k = 0; k < 3; k++) {
        if (p[k] < aabb[k]) aabb[k] = p[k];
        if (p[k] > aabb[3 + k]) aabb[3 + k] = p[k];
    }
}

} // namespace

extern "C" {

rt_mesh* rt_mesh_create(void) { return new rt_mesh(); }
void rt_mesh_destroy(rt_mesh* m) { delete m; }
size_t rt_mesh_vertex_count(const rt_mesh* m) { return m->vertices.size(); }
size_t rt_mesh_index_count(const rt_mesh* m) { return m->indices.size(); }
const RtVertex* rt_mesh_vertices(const rt_mesh* m) { return m->vertices.data(); }
const RtIndex* rt_mesh_indices(const rt_mesh* m) { return m->indices.data(); }

int rt_mesh_push_vertex(rt_mesh* m, const float position[3], const float normal[3]) {
    RtVertex v;
    memcpy(v.position, position, 12); memcpy(v.normal, normal, 12);
    m->vertices.push_back(v);
    return RT_OK;
}
int rt_mesh_push_triangle(rt_mesh* m, uint32_t a, uint32_t b, uint32_t c) {
    m->indices.push_back((RtIndex) a); m->indices.push_back((RtIndex) b); m->indices.push_back((RtIndex) c);
    return RT_OK;
}

int rt_parse_obj_file(const char* filename, int convert_to_rhs, rt_mesh* mesh, float aabb[6]) {
    FILE* file = fopen(filename, "rb");
    if (!file) return RT_ERR_INVALID;
    std::vector<F3> vs, vns;
    std::vector<F2> vts;
    char line[512];
    int status = RT_OK;
    while (fgets(line, sizeof line, file)) {
        const char* p = line;
        if (p[0] == 'v') {
            if (isspace((unsigned char) p[1])) {
                p += 1; F3 v;
                if (parse_floats(p, 3, &v.x) != 3) { status = RT_ERR_INVALID; break; }
                vs.push_back(v);
            } else if (p[1] == 't' && isspace((unsigned char) p[2])) {
                p += 2; F2 vt;
                if (parse_floats(p, 2, &vt.x) != 2) { status = RT_ERR_INVALID; break; }
                vts.push_back(vt);
            } else if (p[1] == 'n' && isspace((unsigned char) p[2])) {
                p += 2; F3 vn;
                if (parse_floats(p, 3, &vn.x) != 3) { status = RT_ERR_INVALID; break; }
                vns.push_back(vn);
            }
        } else if (p[0] == 'f') {
            p += 1;
            unsigned v_idx[4], vt_idx[4], vn_idx[4];
            unsigned nv = 0, nvt = 0, nvn = 0;
            int verts_count = 0;
            bool end_face = false;
            while (verts_count < 4 && !end_face) {
                while (*p && isspace((unsigned char) *p)) {
                    if (*p == '\r' || *p == '\n') { end_face = true; break; }
                    p++;
                }
                if (end_face || !*p) break;
                verts_count++;
                v_idx[nv++] = (unsigned) (atoi(p) - 1);
                // "v", "v/vt", "v//vn", "v/vt/vn"  (parse_obj.cpp:95-118)
                bool end_vertex = false;
                while (*p != '/') {
                    p++;
                    if (!*p || isspace((unsigned char) *p)) { end_vertex = true; break; }
                }
                if (!end_vertex) {
                    p++;
                    if (*p != '/') {
                        vt_idx[nvt++] = (unsigned) (atoi(p) - 1);
                        while (*p != '/') {
                            p++;
                            if (!*p || isspace((unsigned char) *p)) { end_vertex = true; break; }
                        }
                    }
                    if (!end_vertex) {
                        p++;
                        vn_idx[nvn++] = (unsigned) (atoi(p) - 1);
                    }
                }
                while (*p && !isspace((unsigned char) *p)) p++;
            }
            if (verts_count < 3 || (int) nv != verts_count || !(nvt == (unsigned) verts_count || nvt == 0) ||
                !(nvn == (unsigned) verts_count || nvn == 0)) { status = RT_ERR_INVALID; break; }
            bool ok = true;
            for (unsigned i = 0; i < nv; i++) if (v_idx[i] >= vs.size()) ok = false;
            for (unsigned i = 0; i < nvn; i++) if (vn_idx[i] >= vns.size()) ok = false;
            if (!ok) { status = RT_ERR_INVALID; break; }
            (void) vt_idx;
            RtIndex base = (RtIndex) mesh->vertices.size(); // 16-bit, as the reference (parse_obj.cpp:127)
            if (verts_count == 3) {
                RtIndex f[] = { (RtIndex) (base + 0), (RtIndex) (base + 1), (RtIndex) (base + 2) };
                mesh->indices.insert(mesh->indices.end(), f, f + 3);
            } else {
                RtIndex f[] = { (RtIndex) (base + 0), (RtIndex) (base + 1), (RtIndex) (base + 2),
                                (RtIndex) (base + 0), (RtIndex) (base + 2), (RtIndex) (base + 3) };
                mesh->indices.insert(mesh->indices.end(), f, f + 6);
            }
            for (int i = 0; i < verts_count; i++) {
                RtVertex vertex;
                F3 pos = vs[v_idx[i]], nrm;
                if (nvn) nrm = vns[vn_idx[i]];
                else {   // face normal from the first three positions, before the RHS conversion
                    F3 a = vs[v_idx[0]];
                    nrm = normalizef(crossf(sub(vs[v_idx[1]], a), sub(vs[v_idx[2]], a)));
                }
                if (convert_to_rhs) {
                    float t = pos.y; pos.y = pos.z; pos.z = t; pos.x = -pos.x;
                    t = nrm.y; nrm.y = nrm.z; nrm.z = t; nrm.x = -nrm.x;
                }
                vertex.position[0] = pos.x; vertex.position[1] = pos.y; vertex.position[2] = pos.z;
                vertex.normal[0] = nrm.x; vertex.normal[1] = nrm.y; vertex.normal[2] = nrm.z;
                mesh->vertices.push_back(vertex);
                if (aabb) aabb_join(aabb, vertex.position);
            }
        }
    }
    fclose(file);
    return status;
}

void rt_aabb_null(float aabb[6]) {
    const float inf = std::numeric_limits<float>::infinity();
    aabb[0] = aabb[1] = aabb[2] = inf; aabb[3] = aabb[4] = aabb[5] = -inf;
}
float rt_aabb_widest(const float aabb[6]) {
    return fmaxf(fmaxf(aabb[3] - aabb[0], aabb[4] - aabb[1]), aabb[5] - aabb[2]);
}
float rt_triangle_area(const float a[3], const float b[3], const float c[3]) {
    F3 A = { a[0], a[1], a[2] }, B = { b[0], b[1], b[2] }, C = { c[0], c[1], c[2] };
    F3 cr = crossf(sub(B, A), sub(C, A));
    return 0.5f * sqrtf(dotf(cr, cr));
}

void rt_normalizing_transform(const float aabb[6], float m[16]) {
    float scale = (float) (1.0 / rt_aabb_widest(aabb));
    memset(m, 0, 64);
    m[0] = m[5] = m[10] = scale; m[15] = 1.0f;
    for (int k = 0; k < 3; k++) m[12 + k] = (0.5f * (aabb[3 + k] - aabb[k]) - aabb[3 + k]) * scale;
}

void rt_affine_transform(float scale, float rot_z, const float translation[3], float m[16]) {
    float c = cosf(rot_z), s = sinf(rot_z);
    memset(m, 0, 64);
    // row vectors: v' = v * S * Rz + T
    m[0] = scale * c;  m[1] = scale * s;
    m[4] = -scale * s; m[5] = scale * c;
    m[10] = scale; m[15] = 1.0f;
    m[12] = translation[0]; m[13] = translation[1]; m[14] = translation[2];
}

void rt_camera_orbit(float azimuth, float elevation, float distance, const float target[3], float fov_y,
                     float aspect, RtGlobals* g) {
    F3 focus = { target[0], target[1], target[2] };
    F3 eye = { focus.x + -sinf(azimuth) * cosf(elevation) * distance,
               focus.y + -cosf(azimuth) * cosf(elevation) * distance,
               focus.z + sinf(elevation) * distance };
    g->camera_aspect = aspect;
    g->camera_focal_length = 1 / tanf(fov_y / 2);
    // XMMatrixLookAtRH(eye, focus, +Z) followed by XMMatrixInverse: for a rigid view matrix the inverse has the
    // camera axes as rows 0..2 and the eye position as row 3.
    F3 up = { 0, 0, 1 };
    F3 r2 = normalizef(sub(eye, focus));      // camera +Z points away from the focus (camera looks down -Z)
    F3 r0 = normalizef(crossf(up, r2));
    F3 r1 = crossf(r2, r0);
    float* m = g->camera_to_world;
    m[0] = r0.x; m[1] = r0.y; m[2] = r0.z; m[3] = 0;
    m[4] = r1.x; m[5] = r1.y; m[6] = r1.z; m[7] = 0;
    m[8] = r2.x; m[9] = r2.y; m[10] = r2.z; m[11] = 0;
    m[12] = eye.x; m[13] = eye.y; m[14] = eye.z; m[15] = 1;
}

void rt_default_globals(float aspect, RtGlobals* g) {
    memset(g, 0, sizeof *g);
    const double DEGREES = 6.28318530717958647692528676655900577 / 360;
    g->frame_rng = 0;
    g->samples_per_pixel = 1;
    g->bounces_per_sample = 4;
    g->translucent_emission_bounces = 1;
    g->translucent_bssrdf_scale = 0.0f;
    g->translucent_bssrdf_fudge = 1.0f;
    g->translucent_refractive_index = 1.75f;
    for (int k = 0; k < 3; k++) { g->translucent_scattering[k] = 15.0f; g->translucent_absorption[k] = 0.1f; g->light_color[k] = 50.0f; }
    const float target[3] = { 0, 0, -0.06f };
    rt_camera_orbit(0.0f, (float) (9 * DEGREES), 2.5f, target, (float) (30 * DEGREES), aspect, g);
}

uint32_t rt_next_frame_rng(uint32_t rng) {
    rng = (rng ^ 61u) ^ (rng >> 16);
    rng *= 9u;
    rng = rng ^ (rng >> 4);
    rng *= 0x27d4eb2du;
    rng = rng ^ (rng >> 15);
    return rng;
}

int rt_write_ppm(const char* filename, const uint8_t* rgba8, uint32_t width, uint32_t height) {
    FILE* f = fopen(filename, "wb");
    if (!f) return RT_ERR_INVALID;
    fprintf(f, "P6\n%u %u\n255\n", width, height);
    for (size_t i = 0; i < (size_t) width * height; i++) fwrite(rgba8 + 4 * i, 1, 3, f);
    fclose(f);
    return RT_OK;
}

// PNG, 8-bit RGBA, what stbi_write_png(filename, w, h, 4, data, pitch) produces at main.cpp:750 up to the choice of
// compression: the zlib stream here uses stored (uncompressed) deflate blocks, so no compressor is needed
namespace {
uint32_t crc32_update(uint32_t crc, const uint8_t* p, size_t n) {
    static uint32_t table[256]; static bool ready = false;
    if (!ready) {
        for (uint32_t i = 0; i < 256; i++) { uint32_t c = i; for (int k = 0; k < 8; k++) c = (c & 1u) ? 0xedb88320u ^ (c >> 1) : c >> 1; table[i] = c; }
        ready = true;
    }
    for (size_t i = 0; i < n; i++) crc = table[(crc ^ p[i]) & 0xffu] ^ (crc >> 8);
    return crc;
}
void put_be32(std::vector<uint8_t>& v, uint32_t x) { for (int s = 24; s >= 0; s -= 8) v.push_back((uint8_t) (x >> s)); }
bool write_chunk(FILE* f, const char type[4], const std::vector<uint8_t>& data) {
    std::vector<uint8_t> head; put_be32(head, (uint32_t) data.size());
    uint32_t crc = crc32_update(0xffffffffu, (const uint8_t*) type, 4);
    if (!data.empty()) crc = crc32_update(crc, data.data(), data.size());
    std::vector<uint8_t> tail; put_be32(tail, crc ^ 0xffffffffu);
    return fwrite(head.data(), 1, 4, f) == 4 && fwrite(type, 1, 4, f) == 4 &&
           (data.empty() || fwrite(data.data(), 1, data.size(), f) == data.size()) && fwrite(tail.data(), 1, 4, f) == 4;
}
} // namespace

int rt_write_png(const char* filename, const uint8_t* rgba8, uint32_t width, uint32_t height) {
    if (!filename || !rgba8 || width == 0 || height == 0) return RT_ERR_INVALID;
    FILE* f = fopen(filename, "wb");
    if (!f) return RT_ERR_INVALID;
    static const uint8_t signature[8] = { 0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a };
    bool ok = fwrite(signature, 1, 8, f) == 8;
    std::vector<uint8_t> ihdr; put_be32(ihdr, width); put_be32(ihdr, height);
    ihdr.push_back(8); ihdr.push_back(6); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0); // 8-bit RGBA, no interlace
    ok = ok && write_chunk(f, "IHDR", ihdr);
    // raw scanlines: filter byte 0 + row
    const size_t row = (size_t) width * 4, raw_size = (row + 1) * height;
    std::vector<uint8_t> raw(raw_size);
    for (uint32_t y = 0; y < height; y++) { raw[(row + 1) * y] = 0; memcpy(&raw[(row + 1) * y + 1], rgba8 + row * y, row); }
    uint32_t a = 1, b = 0; // Adler-32 of the raw data
    for (size_t i = 0; i < raw_size;) { size_t n = std::min<size_t>(raw_size - i, 5552); for (size_t k = 0; k < n; k++) { a += raw[i + k]; b += a; } a %= 65521u; b %= 65521u; i += n; }
    std::vector<uint8_t> z; z.reserve(raw_size + raw_size / 65535 * 5 + 16);
    z.push_back(0x78); z.push_back(0x01);
    for (size_t i = 0; i < raw_size;) {
        size_t n = std::min<size_t>(raw_size - i, 65535);
        z.push_back(i + n == raw_size ? 1 : 0); // BFINAL, BTYPE = 00 (stored)
        z.push_back((uint8_t) (n & 0xff)); z.push_back((uint8_t) (n >> 8));
        z.push_back((uint8_t) (~n & 0xff)); z.push_back((uint8_t) ((~n >> 8) & 0xff));
        z.insert(z.end(), raw.begin() + i, raw.begin() + i + n);
        i += n;
    }
    put_be32(z, (b << 16) | a);
    ok = ok && write_chunk(f, "IDAT", z) && write_chunk(f, "IEND", std::vector<uint8_t>());
    fclose(f);
    return ok ? RT_OK : RT_ERR_INVALID;
}

int rt_write_pfm(const char* filename, const float* rgba, uint32_t width, uint32_t height, float scale) {
    FILE* f = fopen(filename, "wb");
    if (!f) return RT_ERR_INVALID;
    fprintf(f, "PF\n%u %u\n-1.0\n", width, height);
    for (uint32_t y = height; y-- > 0;) // PFM stores rows bottom to top
        for (uint32_t x = 0; x < width; x++) {
            float rgb[3];
            for (int k = 0; k < 3; k++) rgb[k] = rgba[4 * ((size_t) y * width + x) + k] * scale;
            fwrite(rgb, 4, 3, f);
        }
    fclose(f);
    return RT_OK;
}

} // extern "C"
