// rt_oracle.cpp -- TEST INFRASTRUCTURE ONLY.  Scalar CPU restatement ("golden model") of the per-ray hot path
// of gluyas/raytracer.  Nothing in the product (raytracer_amd/, include/) includes, links or calls this file;
// only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg load the library built from it.
//
// PARITY STATUS: the reference has no tests, golden vectors or fixtures and cannot be
// built or run here (HLSL + D3D12/DXR), so this oracle is pinned only by
//   (a) the shader text it follows line by line (citations below, file:line in the reference tree), and
//   (b) independent numpy restatements of the integer RNG and of the float formulas in tests/.
// The arithmetic the reference leaves to the DXR driver (ray/triangle test, traversal order, texture filter,
// sin/cos) is *defined* here (items S1..S9 in DESIGN.md) => "parity unpinned" for those parts.
//
// Build: see oracle/Makefile (g++ -O2 -ffp-contract=off, no -ffast-math).
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <limits>
#include <thread>
#include <vector>

#include "../include/rt_abi.h"

namespace {

typedef uint32_t u32;
typedef uint64_t u64;
const float F_INF = std::numeric_limits<float>::infinity();

struct V3 {
    float x, y, z;
    float  operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
inline V3 v3(float x, float y, float z) { V3 r = { x, y, z }; return r; }
inline V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline V3 operator*(float s, V3 a) { return v3(s * a.x, s * a.y, s * a.z); }
inline V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
inline V3 operator/(V3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
// S3: dot is evaluated left to right, unfused
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
// S2: normalize(v) = v * (1 / sqrt(dot(v, v))) with IEEE sqrt and divide
inline V3 normalize(V3 a) { float r = 1.0f / sqrtf(dot(a, a)); return a * r; }
inline bool any3(V3 a) { return a.x != 0 || a.y != 0 || a.z != 0; }
inline float as_float(u32 u) { float f; memcpy(&f, &u, 4); return f; }
inline u32 as_uint(float f) { u32 u; memcpy(&u, &f, 4); return u; }
inline V3 ld3(const float* p) { return v3(p[0], p[1], p[2]); }

// ---- random.hlsl ------------------------------------------------------------------------------------------
// random.hlsl:6-15
u32 hash1(u32 seed) {
    seed = (seed ^ 61u) ^ (seed >> 16);
    seed *= 9u;
    seed = seed ^ (seed >> 4);
    seed *= 0x27d4eb2du;
    seed = seed ^ (seed >> 15);
    return seed;
}
// random.hlsl:17-19
u32 hash2(u32 x, u32 y) { return hash1(y) + 31u * hash1(x); }
// random.hlsl:21-23
u32 hash3(u32 x, u32 y, u32 z) { return hash1(z) + 31u * hash2(x, y); }
// random.hlsl:30-36
u32 random_u32(u32& seed) {
    seed ^= (seed << 13);
    seed ^= (seed >> 17);
    seed ^= (seed << 5);
    return seed;
}
// random.hlsl:39-41
float random01(u32& seed) { return as_float(0x3f800000u | (0x007fffffu & random_u32(seed))) - 1.0f; }
// random.hlsl:44-48
float random11(u32& seed) {
    u32 r = random_u32(seed);
    float r01 = as_float(0x3f800000u | (0x007fffffu & r)) - 1.0f;
    return as_float(as_uint(r01) | (r & 0x80000000u));
}

// S1: sin/cos of phi in [0, 2*pi]: quadrant reduction with a 3-term Cody-Waite split of pi/2 and the Cephes
// single-precision minimax polynomials; every operation is a separate IEEE multiply or add (no FMA).
void sincos_spec(float phi, float* s_out, float* c_out) {
    int   k  = (int) (phi * 0.636619772f + 0.5f);
    float fk = (float) k;
    float r  = phi - fk * 1.5703125f;
    r = r - fk * 4.837512969970703125e-4f;
    r = r - fk * 7.54978995489188e-8f;
    float z = r * r;
    float ps = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r + r;
    float pc = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z
             - 0.5f * z + 1.0f;
    switch (k & 3) {
        case 0:  *s_out =  ps; *c_out =  pc; break;
        case 1:  *s_out =  pc; *c_out = -ps; break;
        case 2:  *s_out = -ps; *c_out = -pc; break;
        default: *s_out = -pc; *c_out =  ps; break;
    }
}

const float TAUf = 6.28318530717958647692528676655900577f; // prelude.h:271

// random.hlsl:70-75
V3 random_on_sphere(u32& seed) {
    float phi       = random01(seed) * TAUf;
    float cos_theta = random11(seed);
    float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
    float s, c;
    sincos_spec(phi, &s, &c);
    return v3(sin_theta * c, sin_theta * s, cos_theta);
}
// random.hlsl:85-88
V3 random_on_hemisphere(u32& seed, V3 normal) {
    V3 unit = random_on_sphere(seed);
    float d = 2.0f * dot(normal, unit);
    float m = d < 0.0f ? d : 0.0f; // min(0.0, d)
    return unit - m * normal;
}

// ---- scene ------------------------------------------------------------------------------------------------
struct Geometry {
    u32 index_offset;  // into Blas::indices (elements)
    u32 index_count;
    u32 first_tri;     // index_offset / 3 (BLAS-global primitive index of primitive 0)
    int shader;
    V3  color;
    int translucent_id;
};

struct BvhNode { // oracle-side CPU BVH (binned SAH); test infrastructure for speed and traversal counters only
    float lo[3], hi[3];
    int left, right; // children, or left = -1 -> leaf
    int first, count;
};

struct TranslucentMesh { // TranslucentMesh + BluenoisePreprocess, raytracing.cpp:49-53, bluenoise.h:6-12
    int blas;
    u32 ib_offset;     // elements
    u32 indices_count;
    float total_surface_area;
    std::vector<float> partial_surface_areas;
    V3 aabb_min, aabb_max;
};

struct Blas {
    std::vector<RtVertex> vertices;
    std::vector<u32>      indices;   // rebased 16-bit values (raytracing.cpp:360-366)
    std::vector<Geometry> geometries;
    std::vector<int>      tri_geometry; // BLAS-global primitive -> geometry
    u32 translucent_ids_index, translucent_ids_count;
    std::vector<BvhNode>  nodes;
    std::vector<u32>      bvh_tris;  // BLAS-global primitive indices in leaf order
};

struct Instance {
    float m[16];       // object->world
    float inv[12];     // world->object, rows 0..3 of a 4x3 (row-vector convention): S7
    int   blas;
    u32   instance_id; // InstanceID(): running count per BLAS (raytracing.cpp:472-479,501)
    V3    wlo, whi;    // world AABB (oracle BVH culling only)
};

// Hierarchical evaluation of the sample-point sum (the reference only plans it, todo.txt:1-2).
// Not reference behaviour: an approximation behind orc_set_gather_mode, specified in DESIGN.md section 8 (HG-1..HG-6)
// and restated here independently of the product.
struct HgNode { float centre[3]; float radius2; u32 first, count, right, depth; };
struct HgTree {
    std::vector<HgNode> nodes;      // preorder: the left child of node i is i + 1
    std::vector<u32> order;         // sorted position -> index in generation order
    std::vector<V3> sums;           // per node, refreshed when the read buffer changes
};

struct TranslucentInstance { // raytracing.cpp:54-65
    HgTree hg;
    u32 translucent_id, instance_id;
    float transform[16];
    float scale_factor;
    std::vector<RtSamplePoint> sample_points;        // read buffer
    std::vector<RtSamplePoint> write_sample_points;  // written by translucent_rgen
    std::vector<V3> point_normals;
    u32 samples_count;
    u32 slot;          // translucent_id * stride + instance_id
};

struct Hit {
    int inst, geom, prim; // prim = PrimitiveIndex() within geometry
    u32 prim_global;
    float t, u, v;
};

struct Counters { u64 nodes, tris, insts, rays; };

struct Scene {
    std::vector<Blas> blas;
    std::vector<Instance> instances;
    std::vector<BvhNode> tlas;          // oracle-side top-level BVH over the instances' world boxes (median splits)
    std::vector<int> tlas_instances;    // leaf order
    std::vector<TranslucentMesh> translucent_meshes;
    std::vector<float> translucent_mean_area;      // g_translucent_properties (by translucent_id)
    std::vector<float> properties_by_slot;         // uploaded instanced layout (raytracing.cpp:599-600)
    std::vector<int>   slot_to_instance;           // slot -> index into translucent_instances
    std::vector<TranslucentInstance> translucent_instances;
    RtGlobals g;
    u32 flags;
    u32 width, height;
    std::vector<float>   accumulator;   // RGBA32F
    std::vector<uint8_t> render_target; // RGBA8
    std::vector<std::vector<RtHit> > hit_records; // [bounce][pixel]
    std::vector<float> bssrdf[2];       // 3 * padded texels, interleaved rgb
    u32 bssrdf_texels[2];
    int profile;
    bool brute_force;
    int threads;
    Counters counters;
    u32 tile_rank, tile_world;
    bool hg_enabled; float hg_max_solid_angle;
    Scene() : flags(RT_FLAG_COLLECT_SAMPLES | RT_FLAG_ENABLE_SSS), width(0), height(0), profile(0),
              brute_force(true), threads(1), tile_rank(0), tile_world(1), hg_enabled(false), hg_max_solid_angle(0.0f) {
        memset(&g, 0, sizeof g);
        memset(&counters, 0, sizeof counters);
        bssrdf_texels[0] = bssrdf_texels[1] = 0;
    }
};

// ---- S4: watertight ray/triangle test (Woop, Benthin, Wald 2013), back faces culled -------------------------
struct RayPrep {
    V3 o, d;
    int kx, ky, kz;
    float sx, sy, sz;
};

RayPrep prepare_ray(V3 o, V3 d) {
    RayPrep r;
    r.o = o; r.d = d;
    float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
    int kz = 0; float am = ax;
    if (ay > am) { kz = 1; am = ay; }
    if (az > am) { kz = 2; }
    int kx = kz + 1; if (kx == 3) kx = 0;
    int ky = kx + 1; if (ky == 3) ky = 0;
    if (d[kz] < 0.0f) { int t = kx; kx = ky; ky = t; }
    r.kx = kx; r.ky = ky; r.kz = kz;
    r.sx = d[kx] / d[kz];
    r.sy = d[ky] / d[kz];
    r.sz = 1.0f / d[kz];
    return r;
}

// returns true and (t, u, v) when the ray crosses the FRONT face (dot(cross(v1-v0, v2-v0), d) < 0)
bool intersect_triangle(const RayPrep& r, V3 p0, V3 p1, V3 p2, float* t_out, float* u_out, float* v_out) {
    V3 A = p0 - r.o, B = p1 - r.o, C = p2 - r.o;
    float Ax = A[r.kx] - r.sx * A[r.kz], Ay = A[r.ky] - r.sy * A[r.kz];
    float Bx = B[r.kx] - r.sx * B[r.kz], By = B[r.ky] - r.sy * B[r.kz];
    float Cx = C[r.kx] - r.sx * C[r.kz], Cy = C[r.ky] - r.sy * C[r.kz];
    float U = Cx * By - Cy * Bx;
    float V = Ax * Cy - Ay * Cx;
    float W = Bx * Ay - By * Ax;
    if (U == 0.0f || V == 0.0f || W == 0.0f) {
        double CxBy = (double) Cx * (double) By, CyBx = (double) Cy * (double) Bx;
        U = (float) (CxBy - CyBx);
        double AxCy = (double) Ax * (double) Cy, AyCx = (double) Ay * (double) Cx;
        V = (float) (AxCy - AyCx);
        double BxAy = (double) Bx * (double) Ay, ByAx = (double) By * (double) Ax;
        W = (float) (BxAy - ByAx);
    }
    if (U < 0.0f || V < 0.0f || W < 0.0f) return false;
    float det = U + V + W;
    if (det == 0.0f) return false;
    float Az = r.sz * A[r.kz], Bz = r.sz * B[r.kz], Cz = r.sz * C[r.kz];
    float T = U * Az + V * Bz + W * Cz;
    *t_out = T / det;
    *u_out = V / det; // weight of vertex 1
    *v_out = W / det; // weight of vertex 2
    return true;
}

// S5/S6: closest hit with TMin < t < TMax (exclusive) and ties broken by the lowest (instance, primitive)
inline bool better_hit(float t, int inst, u32 prim_global, const Hit& best) {
    if (t < best.t) return true;
    if (t == best.t && best.inst >= 0) {
        if (inst < best.inst) return true;
        if (inst == best.inst && prim_global < best.prim_global) return true;
    }
    return false;
}

// S7: world->object = inverse of the affine object->world, computed in double (cofactors) and rounded once
void invert_affine(const float* m, float* inv) {
    double a = m[0], b = m[1], c = m[2], d = m[4], e = m[5], f = m[6], g = m[8], h = m[9], i = m[10];
    double A = e * i - f * h, B = c * h - b * i, C = b * f - c * e;
    double D = f * g - d * i, E = a * i - c * g, F = c * d - a * f;
    double G = d * h - e * g, H = b * g - a * h, I = a * e - b * d;
    double det = a * A + b * D + c * G;
    double r[9] = { A / det, B / det, C / det, D / det, E / det, F / det, G / det, H / det, I / det };
    double tx = m[12], ty = m[13], tz = m[14];
    double it[3];
    for (int j = 0; j < 3; j++) it[j] = -(tx * r[0 + j] + ty * r[3 + j] + tz * r[6 + j]);
    for (int k = 0; k < 9; k++) inv[k] = (float) r[k];
    for (int j = 0; j < 3; j++) inv[9 + j] = (float) it[j];
}
inline V3 xform_point43(const float* m, V3 p) { // row vector * 4x3 (rows of 3)
    return v3(p.x * m[0] + p.y * m[3] + p.z * m[6] + m[9],
              p.x * m[1] + p.y * m[4] + p.z * m[7] + m[10],
              p.x * m[2] + p.y * m[5] + p.z * m[8] + m[11]);
}
inline V3 xform_dir43(const float* m, V3 p) {
    return v3(p.x * m[0] + p.y * m[3] + p.z * m[6],
              p.x * m[1] + p.y * m[4] + p.z * m[7],
              p.x * m[2] + p.y * m[5] + p.z * m[8]);
}
inline V3 xform_point44(const float* m, V3 p) { // mul(float4(p, 1), M).xyz
    return v3(p.x * m[0] + p.y * m[4] + p.z * m[8] + m[12],
              p.x * m[1] + p.y * m[5] + p.z * m[9] + m[13],
              p.x * m[2] + p.y * m[6] + p.z * m[10] + m[14]);
}
inline V3 xform_dir44(const float* m, V3 p) {   // mul(float4(p, 0), M).xyz
    return v3(p.x * m[0] + p.y * m[4] + p.z * m[8],
              p.x * m[1] + p.y * m[5] + p.z * m[9],
              p.x * m[2] + p.y * m[6] + p.z * m[10]);
}

// ---- oracle-only CPU BVH (binned SAH).  It never changes an answer: leaves are tested with the same
// intersect_triangle and better_hit, and the box test is padded to be conservative. ---------------------------
struct BuildTri { V3 lo, hi, c; u32 prim; };

void build_bvh_rec(Blas& b, std::vector<BuildTri>& tris, int first, int count, int node_index) {
    V3 lo = v3(F_INF, F_INF, F_INF), hi = v3(-F_INF, -F_INF, -F_INF), clo = lo, chi = hi;
    for (int i = first; i < first + count; i++) {
        const BuildTri& t = tris[i];
        lo = v3(std::min(lo.x, t.lo.x), std::min(lo.y, t.lo.y), std::min(lo.z, t.lo.z));
        hi = v3(std::max(hi.x, t.hi.x), std::max(hi.y, t.hi.y), std::max(hi.z, t.hi.z));
        clo = v3(std::min(clo.x, t.c.x), std::min(clo.y, t.c.y), std::min(clo.z, t.c.z));
        chi = v3(std::max(chi.x, t.c.x), std::max(chi.y, t.c.y), std::max(chi.z, t.c.z));
    }
    {
        BvhNode& n = b.nodes[node_index];
        n.lo[0] = lo.x; n.lo[1] = lo.y; n.lo[2] = lo.z; n.hi[0] = hi.x; n.hi[1] = hi.y; n.hi[2] = hi.z;
        n.left = -1; n.right = -1; n.first = first; n.count = count;
    }
    if (count <= 2) return;
    const int NB = 16;
    int best_axis = -1, best_split = -1; float best_cost = F_INF;
    for (int axis = 0; axis < 3; axis++) {
        float cmin = clo[axis], cmax = chi[axis];
        if (!(cmax > cmin)) continue;
        struct Bin { V3 lo, hi; int n; } bins[NB];
        for (int k = 0; k < NB; k++) { bins[k].lo = v3(F_INF, F_INF, F_INF); bins[k].hi = v3(-F_INF, -F_INF, -F_INF); bins[k].n = 0; }
        float scale = NB / (cmax - cmin);
        for (int i = first; i < first + count; i++) {
            const BuildTri& t = tris[i];
            int k = std::min(NB - 1, std::max(0, (int) ((t.c[axis] - cmin) * scale)));
            Bin& bn = bins[k]; bn.n++;
            bn.lo = v3(std::min(bn.lo.x, t.lo.x), std::min(bn.lo.y, t.lo.y), std::min(bn.lo.z, t.lo.z));
            bn.hi = v3(std::max(bn.hi.x, t.hi.x), std::max(bn.hi.y, t.hi.y), std::max(bn.hi.z, t.hi.z));
        }
        float la[NB], ra[NB]; int ln[NB], rn[NB];
        V3 l = v3(F_INF, F_INF, F_INF), h = v3(-F_INF, -F_INF, -F_INF); int n = 0;
        for (int k = 0; k < NB; k++) {
            n += bins[k].n;
            l = v3(std::min(l.x, bins[k].lo.x), std::min(l.y, bins[k].lo.y), std::min(l.z, bins[k].lo.z));
            h = v3(std::max(h.x, bins[k].hi.x), std::max(h.y, bins[k].hi.y), std::max(h.z, bins[k].hi.z));
            V3 e = h - l; la[k] = n ? (e.x * e.y + e.y * e.z + e.z * e.x) : 0; ln[k] = n;
        }
        l = v3(F_INF, F_INF, F_INF); h = v3(-F_INF, -F_INF, -F_INF); n = 0;
        for (int k = NB - 1; k >= 0; k--) {
            n += bins[k].n;
            l = v3(std::min(l.x, bins[k].lo.x), std::min(l.y, bins[k].lo.y), std::min(l.z, bins[k].lo.z));
            h = v3(std::max(h.x, bins[k].hi.x), std::max(h.y, bins[k].hi.y), std::max(h.z, bins[k].hi.z));
            V3 e = h - l; ra[k] = n ? (e.x * e.y + e.y * e.z + e.z * e.x) : 0; rn[k] = n;
        }
        for (int k = 0; k < NB - 1; k++) {
            if (ln[k] == 0 || rn[k + 1] == 0) continue;
            float cost = la[k] * ln[k] + ra[k + 1] * rn[k + 1];
            if (cost < best_cost) { best_cost = cost; best_axis = axis; best_split = k; }
        }
    }
    int mid;
    if (best_axis < 0) {
        if (count <= 4) return;
        mid = first + count / 2;
    } else {
        V3 e = hi - lo;
        float leaf_cost = (e.x * e.y + e.y * e.z + e.z * e.x) * count;
        if (count <= 4 && leaf_cost <= best_cost + (e.x * e.y + e.y * e.z + e.z * e.x)) return;
        float cmin = clo[best_axis], cmax = chi[best_axis];
        float scale = NB / (cmax - cmin);
        int ax = best_axis, sp = best_split;
        BuildTri* beg = &tris[first];
        BuildTri* m = std::partition(beg, beg + count, [=](const BuildTri& t) {
            int k = std::min(NB - 1, std::max(0, (int) ((t.c[ax] - cmin) * scale)));
            return k <= sp;
        });
        mid = first + (int) (m - beg);
        if (mid == first || mid == first + count) mid = first + count / 2;
    }
    int l = (int) b.nodes.size(); b.nodes.push_back(BvhNode());
    int r = (int) b.nodes.size(); b.nodes.push_back(BvhNode());
    b.nodes[node_index].left = l; b.nodes[node_index].right = r;
    build_bvh_rec(b, tris, first, mid - first, l);
    build_bvh_rec(b, tris, mid, first + count - mid, r);
}

void build_bvh(Blas& b) {
    u32 ntri = (u32) b.indices.size() / 3;
    std::vector<BuildTri> tris(ntri);
    for (u32 i = 0; i < ntri; i++) {
        V3 p0 = ld3(b.vertices[b.indices[3 * i + 0]].position);
        V3 p1 = ld3(b.vertices[b.indices[3 * i + 1]].position);
        V3 p2 = ld3(b.vertices[b.indices[3 * i + 2]].position);
        BuildTri& t = tris[i];
        t.lo = v3(std::min(p0.x, std::min(p1.x, p2.x)), std::min(p0.y, std::min(p1.y, p2.y)), std::min(p0.z, std::min(p1.z, p2.z)));
        t.hi = v3(std::max(p0.x, std::max(p1.x, p2.x)), std::max(p0.y, std::max(p1.y, p2.y)), std::max(p0.z, std::max(p1.z, p2.z)));
        t.c = 0.5f * (t.lo + t.hi);
        t.prim = i;
    }
    b.nodes.clear(); b.nodes.reserve(2 * ntri + 1);
    b.nodes.push_back(BvhNode());
    if (ntri) build_bvh_rec(b, tris, 0, (int) ntri, 0);
    else { BvhNode& n = b.nodes[0]; n.left = -1; n.first = 0; n.count = 0; for (int k = 0; k < 3; k++) { n.lo[k] = F_INF; n.hi[k] = -F_INF; } }
    b.bvh_tris.resize(ntri);
    for (u32 i = 0; i < ntri; i++) b.bvh_tris[i] = tris[i].prim;
}

// conservative slab test (padded); oracle-only
inline bool box_hit(const float* lo, const float* hi, V3 o, V3 inv_d, float tmin, float tmax) {
    float t0 = tmin, t1 = tmax;
    for (int a = 0; a < 3; a++) {
        float ta = (lo[a] - o[a]) * inv_d[a], tb = (hi[a] - o[a]) * inv_d[a];
        float tn = fminf(ta, tb), tf = fmaxf(ta, tb);
        tn = tn - fabsf(tn) * 4e-7f - 1e-30f; tf = tf + fabsf(tf) * 4e-7f + 1e-30f;
        t0 = fmaxf(t0, tn); t1 = fminf(t1, tf); // fmaxf/fminf drop NaNs
    }
    return t0 <= t1;
}

// entry distance of the padded slab test, or +inf for a miss
inline float box_entry(const float* lo, const float* hi, V3 o, V3 inv_d, float tmin, float tmax) {
    float t0 = tmin, t1 = tmax;
    for (int a = 0; a < 3; a++) {
        float ta = (lo[a] - o[a]) * inv_d[a], tb = (hi[a] - o[a]) * inv_d[a];
        float tn = fminf(ta, tb), tf = fmaxf(ta, tb);
        tn = tn - fabsf(tn) * 4e-7f - 1e-30f; tf = tf + fabsf(tf) * 4e-7f + 1e-30f;
        t0 = fmaxf(t0, tn); t1 = fminf(t1, tf);
    }
    return t0 <= t1 ? t0 : F_INF;
}
struct StackEntry { int node; float entry; };

void intersect_blas(const Scene& s, int inst_index, V3 o, V3 d, float tmin, float tmax, Hit& best, Counters& cnt) {
    const Instance& inst = s.instances[inst_index];
    const Blas& b = s.blas[inst.blas];
    // DXR: the ray is transformed into object space; t is preserved (direction is not renormalised)
    V3 oo = xform_point43(inst.inv, o), od = xform_dir43(inst.inv, d);
    RayPrep rp = prepare_ray(oo, od);
    cnt.insts++;
    if (s.brute_force) {
        u32 ntri = (u32) b.indices.size() / 3;
        for (u32 i = 0; i < ntri; i++) {
            V3 p0 = ld3(b.vertices[b.indices[3 * i + 0]].position);
            V3 p1 = ld3(b.vertices[b.indices[3 * i + 1]].position);
            V3 p2 = ld3(b.vertices[b.indices[3 * i + 2]].position);
            float t, u, v;
            cnt.tris++;
            if (!intersect_triangle(rp, p0, p1, p2, &t, &u, &v)) continue;
            if (!(t > tmin && t < tmax)) continue;
            if (!better_hit(t, inst_index, i, best)) continue;
            best.t = t; best.u = u; best.v = v; best.inst = inst_index; best.prim_global = i;
        }
        return;
    }
    V3 inv_d = v3(1.0f / od.x, 1.0f / od.y, 1.0f / od.z);
    StackEntry stack[128]; int sp = 0;
    {
        const BvhNode& root = b.nodes[0];
        float e = box_entry(root.lo, root.hi, oo, inv_d, tmin, best.t < tmax ? best.t : tmax);
        if (e < F_INF) { stack[sp].node = 0; stack[sp].entry = e; sp++; }
    }
    while (sp) {
        StackEntry se = stack[--sp];
        float tcap = best.t < tmax ? best.t : tmax;
        if (se.entry > tcap) continue; // a closer hit was found since this node was pushed
        const BvhNode& n = b.nodes[se.node];
        if (n.left < 0) {
            for (int k = n.first; k < n.first + n.count; k++) {
                u32 i = b.bvh_tris[k];
                V3 p0 = ld3(b.vertices[b.indices[3 * i + 0]].position);
                V3 p1 = ld3(b.vertices[b.indices[3 * i + 1]].position);
                V3 p2 = ld3(b.vertices[b.indices[3 * i + 2]].position);
                float t, u, v;
                cnt.tris++;
                if (!intersect_triangle(rp, p0, p1, p2, &t, &u, &v)) continue;
                if (!(t > tmin && t < tmax)) continue;
                if (!better_hit(t, inst_index, i, best)) continue;
                best.t = t; best.u = u; best.v = v; best.inst = inst_index; best.prim_global = i;
            }
        } else {
            // one expanded inner node == one 64-byte two-child node fetched in the GPU layout (N_node):
            // both child boxes are tested here and the nearer child is visited first
            cnt.nodes++;
            const BvhNode& l = b.nodes[n.left]; const BvhNode& r = b.nodes[n.right];
            float el = box_entry(l.lo, l.hi, oo, inv_d, tmin, tcap), er = box_entry(r.lo, r.hi, oo, inv_d, tmin, tcap);
            if (sp + 2 > 128) abort();
            StackEntry a = { n.left, el }, c = { n.right, er };
            if (er < el) { StackEntry t = a; a = c; c = t; } // a = nearer
            if (c.entry < F_INF) stack[sp++] = c;
            if (a.entry < F_INF) stack[sp++] = a;
        }
    }
}

// top-level BVH over instance boxes: median split of the box centres along the widest axis, one instance per leaf.
// The answer does not depend on it (S5: ties go to the lowest instance / primitive whatever the visiting order).
int build_tlas_rec(Scene& s, int first, int count) {
    int id = (int) s.tlas.size();
    s.tlas.push_back(BvhNode());
    BvhNode n;
    for (int k = 0; k < 3; k++) { n.lo[k] = F_INF; n.hi[k] = -F_INF; }
    float clo[3] = { F_INF, F_INF, F_INF }, chi[3] = { -F_INF, -F_INF, -F_INF };
    for (int i = first; i < first + count; i++) {
        const Instance& in = s.instances[s.tlas_instances[i]];
        for (int k = 0; k < 3; k++) {
            n.lo[k] = std::min(n.lo[k], in.wlo[k]); n.hi[k] = std::max(n.hi[k], in.whi[k]);
            float c = 0.5f * (in.wlo[k] + in.whi[k]);
            clo[k] = std::min(clo[k], c); chi[k] = std::max(chi[k], c);
        }
    }
    n.first = first; n.count = count; n.left = -1; n.right = -1;
    if (count > 1) {
        int axis = 0;
        if (chi[1] - clo[1] > chi[axis] - clo[axis]) axis = 1;
        if (chi[2] - clo[2] > chi[axis] - clo[axis]) axis = 2;
        int mid = first + count / 2;
        std::nth_element(s.tlas_instances.begin() + first, s.tlas_instances.begin() + mid, s.tlas_instances.begin() + first + count,
                         [&](int a, int b) {
                             float ca = s.instances[a].wlo[axis] + s.instances[a].whi[axis], cb = s.instances[b].wlo[axis] + s.instances[b].whi[axis];
                             return ca < cb || (ca == cb && a < b);
                         });
        n.left = build_tlas_rec(s, first, mid - first);
        n.right = build_tlas_rec(s, mid, first + count - mid);
    }
    s.tlas[id] = n;
    return id;
}
void build_tlas_bvh(Scene& s) {
    s.tlas.clear();
    s.tlas_instances.resize(s.instances.size());
    for (size_t i = 0; i < s.instances.size(); i++) s.tlas_instances[i] = (int) i;
    if (!s.instances.empty()) build_tlas_rec(s, 0, (int) s.instances.size());
}

// TraceRay (raytracing.hlsl:108-112): closest hit over all instances, RAY_FLAG_CULL_BACK_FACING_TRIANGLES
Hit trace_ray(const Scene& s, V3 o, V3 d, float tmin, float tmax, Counters& cnt) {
    Hit best; best.inst = -1; best.geom = -1; best.prim = -1; best.prim_global = 0; best.t = F_INF; best.u = best.v = 0;
    cnt.rays++;
    if (s.brute_force) {
        for (int i = 0; i < (int) s.instances.size(); i++) intersect_blas(s, i, o, d, tmin, tmax, best, cnt);
    } else if (!s.tlas.empty()) {
        V3 inv_d = v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
        StackEntry stack[128]; int sp = 0;
        {
            float e = box_entry(s.tlas[0].lo, s.tlas[0].hi, o, inv_d, tmin, tmax);
            if (e < F_INF) { stack[sp].node = 0; stack[sp].entry = e; sp++; }
        }
        while (sp) {
            StackEntry se = stack[--sp];
            float tcap = best.t < tmax ? best.t : tmax;
            if (se.entry > tcap) continue;
            const BvhNode& n = s.tlas[se.node];
            if (n.left < 0) { intersect_blas(s, s.tlas_instances[n.first], o, d, tmin, tmax, best, cnt); continue; }
            cnt.nodes++; // an expanded top-level node, counted like the bottom-level ones (N_node)
            const BvhNode& l = s.tlas[n.left]; const BvhNode& r = s.tlas[n.right];
            float el = box_entry(l.lo, l.hi, o, inv_d, tmin, tcap), er = box_entry(r.lo, r.hi, o, inv_d, tmin, tcap);
            if (sp + 2 > 128) abort();
            StackEntry a = { n.left, el }, c = { n.right, er };
            if (er < el) { StackEntry t = a; a = c; c = t; }
            if (c.entry < F_INF) stack[sp++] = c;
            if (a.entry < F_INF) stack[sp++] = a;
        }
    }
    if (best.inst >= 0) {
        const Blas& b = s.blas[s.instances[best.inst].blas];
        best.geom = b.tri_geometry[best.prim_global];
        best.prim = (int) (best.prim_global - b.geometries[best.geom].first_tri);
    }
    return best;
}

// ---- prelude.h helpers ------------------------------------------------------------------------------------
// prelude.h:241-246
V3 get_interpolated_normal(const RtVertex* v0, const RtVertex* v1, const RtVertex* v2, float bx, float by) {
    V3 n0 = ld3(v0->normal), n1 = ld3(v1->normal), n2 = ld3(v2->normal);
    V3 normal = n0;
    normal = normal + bx * (n1 - n0);
    normal = normal + by * (n2 - n0);
    return normal;
}
// prelude.h:248-265
void get_barycentrics(V3 p0, V3 p1, V3 p2, V3 position, float* bx, float* by) {
    V3 v0 = p1 - p0, v1 = p2 - p0, v2 = position - p0;
    float d00 = dot(v0, v0), d01 = dot(v0, v1), d11 = dot(v1, v1), d20 = dot(v2, v0), d21 = dot(v2, v1);
    float denom = d00 * d11 - d01 * d01;
    *bx = (d11 * d20 - d01 * d21) / denom;
    *by = (d00 * d21 - d01 * d20) / denom;
}
inline float sign_f(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }

// raytracing.hlsl:60-67 (load_3x16bit_indices prelude.h:195-218 reduces to three index reads)
V3 get_world_space_normal(const Scene& s, const Hit& h, V3 world_dir) {
    const Instance& inst = s.instances[h.inst];
    const Blas& b = s.blas[inst.blas];
    const u32* idx = &b.indices[3 * h.prim_global];
    V3 normal = get_interpolated_normal(&b.vertices[idx[0]], &b.vertices[idx[1]], &b.vertices[idx[2]], h.u, h.v);
    normal = xform_dir44(inst.m, normal);                 // mul(float4(normal, 0), ObjectToWorld4x3())
    normal = normal * (-sign_f(dot(world_dir, normal)));  // towards the viewer
    normal = normalize(normal);
    return normal;
}

// raytracing.hlsl:211-222
float schlick(float refractive_index, float cosine) {
    float r0 = (refractive_index - 1.0f) / (refractive_index + 1.0f);
    r0 *= r0;
    float fresnel;
    fresnel = 1.0f - cosine;
    fresnel *= fresnel * fresnel * fresnel * fresnel;
    fresnel *= 1.0f - r0;
    fresnel += r0;
    return fresnel;
}

// S8: the 1-D texture fetch of raytracing.hlsl:275: linear filter at texel coordinate u*N - 0.5, border = 0
V3 sample_bssrdf_table(const Scene& s, float u) {
    const std::vector<float>& tab = s.bssrdf[s.profile];
    int n = (int) s.bssrdf_texels[s.profile];
    float x = u * (float) n - 0.5f;
    float fl = floorf(x);
    float f = x - fl;
    if (!(fl >= -1.0f && fl < (float) n)) return v3(0, 0, 0); // also NaN
    int i0 = (int) fl, i1 = i0 + 1;
    V3 a = (i0 >= 0 && i0 < n) ? ld3(&tab[3 * i0]) : v3(0, 0, 0);
    V3 b = (i1 >= 0 && i1 < n) ? ld3(&tab[3 * i1]) : v3(0, 0, 0);
    return a + f * (b - a);
}

// raytracing.hlsl:272-276
V3 eval_bssrdf_tabulated(const Scene& s, float radius) {
    float z = s.g.translucent_bssrdf_scale * s.g.translucent_bssrdf_scale;
    float f = s.g.translucent_bssrdf_fudge;
    return (f * sample_bssrdf_table(s, radius / s.g.translucent_bssrdf_scale)) / z;
}

// raytracing.hlsl:278-301 (per channel)
float dipole_channel(float sigma_s, float sigma_a, float eta, float radius) {
    float attenuation = sigma_s + sigma_a;
    float mean_free_path = 1.0f / attenuation;
    float albedo = sigma_s / attenuation;
    float effective_attenuation = sqrtf(3.0f * sigma_s * sigma_a);
    float diffuse_fresnel = -1.440f / (eta * eta) + 0.710f / eta + 0.668f + 0.0636f * eta;
    float z_real = mean_free_path;
    float d_real = radius + z_real;
    float c_real = z_real * (effective_attenuation + 1.0f / d_real);
    float z_virtual = mean_free_path * (1.0f + 1.25f * (1.0f + diffuse_fresnel) / (1.0f - diffuse_fresnel));
    float d_virtual = radius + z_virtual;
    float c_virtual = z_virtual * (effective_attenuation + 1.0f / d_virtual);
    float m_real = c_real * expf(-effective_attenuation * d_real) / (d_real * d_real);
    float m_virtual = c_virtual * expf(-effective_attenuation * d_virtual) / (d_virtual * d_virtual);
    float r = albedo / (2.0f * TAUf) * (m_real + m_virtual);
    return r > 0.0f ? r : 0.0f; // max(0, r); NaN -> 0 as HLSL max
}
V3 eval_bssrdf_dipole(const Scene& s, float radius) {
    const RtGlobals& g = s.g;
    return v3(dipole_channel(g.translucent_scattering[0], g.translucent_absorption[0], g.translucent_refractive_index, radius),
              dipole_channel(g.translucent_scattering[1], g.translucent_absorption[1], g.translucent_refractive_index, radius),
              dipole_channel(g.translucent_scattering[2], g.translucent_absorption[2], g.translucent_refractive_index, radius));
}

// ---- hit shaders ------------------------------------------------------------------------------------------
struct RayPayload { // raytracing.hlsl:71-77
    u32 rng; float t; V3 scatter, reflectance, emission;
};

// raytracing.hlsl:170-179
void lambert_chit(const Scene& s, const Geometry& geo, const Hit& h, V3 dir, RayPayload& p) {
    V3 normal = get_world_space_normal(s, h, dir);
    p.scatter = random_on_hemisphere(p.rng, normal);
    p.reflectance = geo.color * dot(normal, p.scatter);
    p.emission = v3(0, 0, 0);
    p.t = h.t;
}
// raytracing.hlsl:186-199
void light_chit(const Scene& s, const Geometry& geo, const Hit& h, V3 dir, RayPayload& p) {
    V3 normal = get_world_space_normal(s, h, dir);
    V3 color = any3(geo.color) ? geo.color : ld3(s.g.light_color);
    p.scatter = v3(0, 0, 0);
    p.reflectance = v3(0, 0, 0);
    p.emission = color * (-dot(normal, dir));
    p.t = h.t;
}
// raytracing.hlsl:201-207
void miss(RayPayload& p) {
    p.scatter = v3(0, 0, 0); p.reflectance = v3(0, 0, 0); p.emission = v3(0, 0, 0); p.t = F_INF;
}
// HG-1..HG-4: Morton order of the points, implicit midpoint tree in preorder, leaves of <= 8 points
u32 hg_part1by2(u32 x) {
    x &= 0x3ffu; x = (x | (x << 16)) & 0x030000ffu; x = (x | (x << 8)) & 0x0300f00fu;
    x = (x | (x << 4)) & 0x030c30c3u; x = (x | (x << 2)) & 0x09249249u;
    return x;
}
u32 hg_build_node(HgTree& t, const std::vector<RtSamplePoint>& pts, u32 first, u32 count, u32 depth) {
    u32 id = (u32) t.nodes.size();
    t.nodes.push_back(HgNode());
    double sum[3] = { 0, 0, 0 };
    for (u32 j = 0; j < count; j++) for (int a = 0; a < 3; a++) sum[a] += (double) pts[t.order[first + j]].position[a];
    HgNode n;
    for (int a = 0; a < 3; a++) n.centre[a] = (float) (sum[a] / (double) count);
    n.radius2 = 0.0f;
    for (u32 j = 0; j < count; j++) {
        V3 d = ld3(pts[t.order[first + j]].position) - ld3(n.centre);
        float d2 = dot(d, d);
        if (d2 > n.radius2) n.radius2 = d2;
    }
    n.first = first; n.count = count; n.right = 0; n.depth = depth;
    if (count > 8) {
        u32 half = (count + 1) / 2;
        hg_build_node(t, pts, first, half, depth + 1);
        n.right = hg_build_node(t, pts, first + half, count - half, depth + 1);
    }
    t.nodes[id] = n;
    return id;
}
void hg_build(TranslucentInstance& ti) {
    HgTree& t = ti.hg;
    t.nodes.clear(); t.order.clear(); t.sums.clear();
    const std::vector<RtSamplePoint>& pts = ti.sample_points;
    u32 n = (u32) pts.size();
    if (!n) return;
    float lo[3], hi[3];
    for (int a = 0; a < 3; a++) { lo[a] = pts[0].position[a]; hi[a] = pts[0].position[a]; }
    for (u32 k = 1; k < n; k++) for (int a = 0; a < 3; a++) {
        lo[a] = fminf(lo[a], pts[k].position[a]); hi[a] = fmaxf(hi[a], pts[k].position[a]);
    }
    std::vector<std::pair<u32, u32> > keyed(n);
    for (u32 k = 0; k < n; k++) {
        u32 q[3];
        for (int a = 0; a < 3; a++) {
            float e = hi[a] - lo[a];
            q[a] = e > 0.0f ? (u32) fminf(fmaxf((pts[k].position[a] - lo[a]) / e * 1024.0f, 0.0f), 1023.0f) : 0u;
        }
        keyed[k] = std::make_pair(hg_part1by2(q[0]) | (hg_part1by2(q[1]) << 1) | (hg_part1by2(q[2]) << 2), k);
    }
    std::sort(keyed.begin(), keyed.end());
    t.order.resize(n);
    for (u32 k = 0; k < n; k++) t.order[k] = keyed[k].second;
    hg_build_node(t, pts, 0, n, 0);
}
// HG-5: node sums of the read buffer's payloads (leaf: sequential in sorted order from 0; inner: left + right)
V3 hg_sum_node(HgTree& t, const std::vector<RtSamplePoint>& pts, u32 id) {
    const HgNode& n = t.nodes[id];
    V3 sum = v3(0, 0, 0);
    if (n.right == 0) for (u32 j = 0; j < n.count; j++) sum = sum + ld3(pts[t.order[n.first + j]].payload);
    else { V3 l = hg_sum_node(t, pts, id + 1); V3 r = hg_sum_node(t, pts, n.right); sum = l + r; }
    t.sums[id] = sum;
    return sum;
}
void hg_refresh(TranslucentInstance& ti) {
    if (ti.hg.nodes.empty()) return;
    ti.hg.sums.assign(ti.hg.nodes.size(), v3(0, 0, 0));
    hg_sum_node(ti.hg, ti.sample_points, 0);
}
V3 eval_bssrdf(const Scene& s, float fudge, float radius) {
    if (s.g.translucent_bssrdf_scale != 0.0f) {
        float z = s.g.translucent_bssrdf_scale * s.g.translucent_bssrdf_scale;
        return (fudge * sample_bssrdf_table(s, radius / s.g.translucent_bssrdf_scale)) / z;
    }
    return eval_bssrdf_dipole(s, radius);
}
// HG-6: preorder walk, left before right; a node stands for its points when the hit is outside its bounding sphere
// and it subtends less than the solid-angle limit: count * mean_area < limit * distance^2
V3 hg_gather(const Scene& s, const TranslucentInstance& ti, float mean_area, float fudge, V3 hit_point, u64* pairs) {
    const HgTree& t = ti.hg;
    V3 sum = v3(0, 0, 0);
    if (t.nodes.empty()) return sum;
    u32 stack[64]; int sp = 0;
    stack[sp++] = 0;
    while (sp > 0) {
        u32 id = stack[--sp];
        const HgNode& n = t.nodes[id];
        if (n.right == 0) {
            for (u32 j = 0; j < n.count; j++) {
                const RtSamplePoint& p = ti.sample_points[t.order[n.first + j]];
                V3 dpos = ld3(p.position) - hit_point;
                sum = sum + eval_bssrdf(s, fudge, sqrtf(dot(dpos, dpos))) * ld3(p.payload);
            }
            if (pairs) *pairs += n.count;
            continue;
        }
        V3 d = ld3(n.centre) - hit_point;
        float d2 = dot(d, d);
        if (d2 > n.radius2 && (float) n.count * mean_area < s.hg_max_solid_angle * d2) {
            sum = sum + eval_bssrdf(s, fudge, sqrtf(d2)) * t.sums[id];
            if (pairs) *pairs += 1;
        } else { stack[sp++] = n.right; stack[sp++] = id + 1; }
    }
    return sum;
}

// raytracing.hlsl:305-349
void translucent_chit(const Scene& s, const Geometry& geo, const Hit& h, V3 org, V3 dir, RayPayload& p, u64* pairs) {
    const Instance& inst = s.instances[h.inst];
    u32 index = (u32) geo.translucent_id * s.g.translucent_instance_stride + inst.instance_id;
    V3 normal = get_world_space_normal(s, h, dir);
    V3 hit_point = org + h.t * dir;
    V3 diffuse_irradiance = v3(0, 0, 0);
    float fudge = (s.flags & RT_FLAG_ENABLE_SSS) ? s.g.translucent_bssrdf_fudge : 0.0f; // raytracing.cpp:653-657
    if (p.t <= (float) s.g.translucent_emission_bounces && fudge != 0.0f) {
        const TranslucentInstance* ti = NULL;
        if (index < s.slot_to_instance.size() && s.slot_to_instance[index] >= 0) ti = &s.translucent_instances[s.slot_to_instance[index]];
        u32 samples_count = ti ? ti->samples_count : 0;
        if (s.hg_enabled && ti) {
            diffuse_irradiance = hg_gather(s, *ti, s.properties_by_slot[index], fudge, hit_point, pairs);
            samples_count = 0;
        }
        for (u32 i = 0; i < samples_count; i++) {
            const RtSamplePoint& sp = ti->sample_points[i];
            V3 dpos = ld3(sp.position) - hit_point;
            float radius = sqrtf(dot(dpos, dpos));
            V3 bssrdf;
            if (s.g.translucent_bssrdf_scale != 0.0f) {
                float z = s.g.translucent_bssrdf_scale * s.g.translucent_bssrdf_scale;
                bssrdf = (fudge * sample_bssrdf_table(s, radius / s.g.translucent_bssrdf_scale)) / z;
            } else bssrdf = eval_bssrdf_dipole(s, radius);
            diffuse_irradiance = diffuse_irradiance + bssrdf * ld3(sp.payload);
        }
        if (pairs && !s.hg_enabled) *pairs += samples_count;
        diffuse_irradiance = diffuse_irradiance / (float) (s.g.translucent_accumulator_count + 1u);
    }
    float n = s.g.translucent_refractive_index;
    V3 scatter = random_on_hemisphere(p.rng, normal);
    float lambert = dot(scatter, normal);
    float incident_cosine = lambert;
    float incident_fresnel = schlick(n, incident_cosine);
    float transmitted_cosine = sqrtf(1.0f - 1.0f / (n * n) * (1.0f - (-dot(dir, normal))));
    float transmitted_fresnel = 1.0f - schlick(n, transmitted_cosine);
    p.scatter = scatter;
    p.reflectance = geo.color * lambert * incident_fresnel;
    p.emission = diffuse_irradiance * transmitted_fresnel / (TAUf / 2.0f);
    p.t = h.t;
}

struct PathTap { std::vector<RtHit>* records; u32 path; }; // parity tap (sample 0 only)

// raytracing.hlsl:92-126.  Returns rgba; origin/direction are inout exactly as in the shader.
void trace_path_sample(const Scene& s, u32& rng, V3& origin, V3& direction, float tmin, float tmax,
                       bool ignore_translucent_emission, float* result, Counters& cnt, u64* pairs,
                       std::vector<std::vector<RtHit> >* tap, u32 path) {
    RayPayload payload; payload.rng = rng; payload.t = 0;
    payload.scatter = payload.reflectance = payload.emission = v3(0, 0, 0);
    V3 radiance = v3(0, 0, 0), reflectance = v3(1, 1, 1);
    u32 bounce_index;
    for (bounce_index = 0; bounce_index <= s.g.bounces_per_sample; bounce_index++) {
        payload.t = (float) bounce_index;
        if (ignore_translucent_emission) payload.t = F_INF;
        Hit h = trace_ray(s, origin, direction, tmin, tmax, cnt);
        if (h.inst < 0) miss(payload);
        else {
            const Geometry& geo = s.blas[s.instances[h.inst].blas].geometries[h.geom];
            if (geo.shader == RT_SHADER_LAMBERT) lambert_chit(s, geo, h, direction, payload);
            else if (geo.shader == RT_SHADER_LIGHT) light_chit(s, geo, h, direction, payload);
            else translucent_chit(s, geo, h, origin, direction, payload, pairs);
        }
        if (tap && bounce_index < tap->size()) {
            RtHit& r = (*tap)[bounce_index][path];
            r.instance = h.inst; r.geometry = h.geom; r.primitive = h.prim; r.t = h.inst < 0 ? F_INF : h.t;
            r.u = h.u; r.v = h.v; r.rng_after = payload.rng; r.valid = 1;
        }
        radiance = radiance + payload.emission * reflectance;
        reflectance = reflectance * payload.reflectance;
        if (!any3(payload.reflectance)) break;
        origin = origin + payload.t * direction;
        direction = payload.scatter;
    }
    result[0] = radiance.x; result[1] = radiance.y; result[2] = radiance.z;
    result[3] = !(bounce_index == 0 && std::isinf(payload.t)) ? 1.0f : 0.0f;
    rng = payload.rng;
}

inline uint8_t to_unorm8(float x) { // S9: D3D float -> UNORM8 (NaN -> 0, clamp, round to nearest)
    if (!(x > 0.0f)) return 0;
    if (x > 1.0f) x = 1.0f;
    return (uint8_t) (int) (x * 255.0f + 0.5f);
}

// raytracing.hlsl:128-163 for one pixel
void camera_rgen_pixel(Scene& s, u32 px, u32 py, bool capture, Counters& cnt, u64* pairs) {
    const RtGlobals& g = s.g;
    u32 rng = hash3(px, py, g.frame_rng * (u32) (g.accumulator_count != 0));
    const float tmin = 0.000001f, tmax = 10000.0f;
    float acc[4] = { 0, 0, 0, 0 };
    u32 pixel = py * s.width + px;
    for (u32 sample_index = 0; sample_index < g.samples_per_pixel; sample_index++) {
        const float* c2w = g.camera_to_world;
        V3 origin = v3(c2w[12], c2w[13], c2w[14]) / c2w[15];
        float dx = (float) px + 0.5f, dy = (float) py + 0.5f;
        float jx = random11(rng); // Q6: x first
        float jy = random11(rng);
        dx = dx + 0.5f * jx; dy = dy + 0.5f * jy;
        dx = 2.0f * dx / (float) s.width - 1.0f;
        dy = 2.0f * dy / (float) s.height - 1.0f;
        dx = dx * g.camera_aspect;
        dy = dy * -1.0f;
        float dz = -g.camera_focal_length;
        V3 direction = normalize(xform_dir44(c2w, v3(dx, dy, dz)));
        float r[4];
        trace_path_sample(s, rng, origin, direction, tmin, tmax, false, r, cnt, pairs,
                          (capture && sample_index == 0) ? &s.hit_records : NULL, pixel);
        for (int k = 0; k < 4; k++) acc[k] += r[k];
    }
    for (int k = 0; k < 4; k++) acc[k] /= (float) g.samples_per_pixel;
    if (g.accumulator_count != 0) for (int k = 0; k < 4; k++) acc[k] += s.accumulator[4 * pixel + k];
    for (int k = 0; k < 4; k++) {
        s.render_target[4 * pixel + k] = to_unorm8(sqrtf(acc[k] / (float) (g.accumulator_count + 1u)));
        s.accumulator[4 * pixel + k] = acc[k];
    }
}

// raytracing.hlsl:224-265 for one (sample point, translucent instance)
void translucent_rgen_point(Scene& s, TranslucentInstance& ti, u32 x, Counters& cnt) {
    const RtGlobals& g = s.g;
    float mean_area = s.properties_by_slot[ti.slot];
    RtSamplePoint sample_point = ti.write_sample_points[x];
    V3 normal = ti.point_normals[x];
    V3 payload = ld3(sample_point.payload);
    if (g.translucent_accumulator_count == 0) payload = v3(0, 0, 0);
    u32 rng = hash3(x, ti.slot, g.frame_rng * (u32) (g.translucent_accumulator_count != 0));
    const float tmin = 0.0001f, tmax = 10000.0f;
    V3 transmitted_irradiance = v3(0, 0, 0);
    for (u32 i = 0; i < g.samples_per_pixel; i++) {
        V3 origin = ld3(sample_point.position);
        V3 direction = random_on_hemisphere(rng, normal);
        float r[4];
        trace_path_sample(s, rng, origin, direction, tmin, tmax, true, r, cnt, NULL, NULL, 0);
        V3 radiance = v3(r[0], r[1], r[2]);
        float cosine = -dot(direction, normal); // direction was modified by trace_path_sample (inout)
        float fresnel = 1.0f - schlick(g.translucent_refractive_index, cosine);
        transmitted_irradiance = transmitted_irradiance + radiance * cosine * fresnel;
    }
    payload = payload + (transmitted_irradiance * mean_area) / (TAUf / 2.0f * (float) g.samples_per_pixel);
    sample_point.payload[0] = payload.x; sample_point.payload[1] = payload.y; sample_point.payload[2] = payload.z;
    ti.write_sample_points[x] = sample_point;
}

template <typename F> void parallel_for(int threads, u32 n, F f) {
    if (threads <= 1 || n < 2) { for (u32 i = 0; i < n; i++) f(i, 0); return; }
    std::atomic<u32> next(0);
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; t++) pool.emplace_back([&, t]() {
        for (;;) { u32 i = next.fetch_add(1); if (i >= n) break; f(i, t); }
    });
    for (auto& th : pool) th.join();
}

void camera_pass(Scene& s) {
    bool capture = (s.flags & RT_FLAG_CAPTURE_HITS) != 0;
    if (capture) {
        s.hit_records.assign(s.g.bounces_per_sample + 1, std::vector<RtHit>());
        RtHit none; memset(&none, 0, sizeof none); none.instance = -1; none.geometry = -1; none.primitive = -1;
        for (auto& v : s.hit_records) v.assign((size_t) s.width * s.height, none);
    }
    int nt = std::max(1, s.threads);
    std::vector<Counters> cnt(nt); std::vector<u64> pairs(nt, 0);
    for (auto& c : cnt) memset(&c, 0, sizeof c);
    parallel_for(nt, s.height, [&](u32 py, int t) {
        for (u32 px = 0; px < s.width; px++) {
            if (s.tile_world > 1) {
                u32 tiles_x = (s.width + 31) / 32;
                u32 tile = (py / 32) * tiles_x + (px / 32);
                if (tile % s.tile_world != s.tile_rank) continue;
            }
            camera_rgen_pixel(s, px, py, capture, cnt[t], &pairs[t]);
        }
    });
    for (int t = 0; t < nt; t++) { s.counters.nodes += cnt[t].nodes; s.counters.tris += cnt[t].tris; s.counters.insts += cnt[t].insts; s.counters.rays += cnt[t].rays; }
}

void probe_pass(Scene& s) {
    int nt = std::max(1, s.threads);
    std::vector<Counters> cnt(nt);
    for (auto& c : cnt) memset(&c, 0, sizeof c);
    for (auto& ti : s.translucent_instances) {
        TranslucentInstance* tip = &ti;
        parallel_for(nt, ti.samples_count, [&, tip](u32 x, int t) { translucent_rgen_point(s, *tip, x, cnt[t]); });
    }
    for (int t = 0; t < nt; t++) { s.counters.nodes += cnt[t].nodes; s.counters.tris += cnt[t].tris; s.counters.insts += cnt[t].insts; s.counters.rays += cnt[t].rays; }
}

// ---- blue-noise sample points: bluenoise.cpp:69-147,263-351 + bluenoise.hlsl:44-245 -----------------------------
struct InitialSamplePoint { V3 position; u32 triangle_id; u32 cell_id; };
struct HashEntry { u32 cell_id; u32 first_index; V3 selected; };

// geometry.cpp:21-25
float triangle_area(V3 a, V3 b, V3 c) {
    V3 cr = cross(b - a, c - a);
    return 0.5f * sqrtf(dot(cr, cr));
}

// bluenoise.cpp:69-93
void preprocess_mesh_data(const Blas& b, TranslucentMesh& m) {
    m.total_surface_area = 0;
    m.partial_surface_areas.clear();
    m.aabb_min = v3(F_INF, F_INF, F_INF); m.aabb_max = v3(-F_INF, -F_INF, -F_INF);
    for (u32 i = 0; i < m.indices_count / 3; i++) {
        V3 p[3];
        for (int k = 0; k < 3; k++) p[k] = ld3(b.vertices[b.indices[m.ib_offset + 3 * i + k]].position);
        m.total_surface_area += triangle_area(p[0], p[1], p[2]);
        m.partial_surface_areas.push_back(m.total_surface_area);
        for (int k = 0; k < 3; k++) {
            m.aabb_min = v3(std::min(m.aabb_min.x, p[k].x), std::min(m.aabb_min.y, p[k].y), std::min(m.aabb_min.z, p[k].z));
            m.aabb_max = v3(std::max(m.aabb_max.x, p[k].x), std::max(m.aabb_max.y, p[k].y), std::max(m.aabb_max.z, p[k].z));
        }
    }
}

struct BluenoiseGrid {
    float rejection_radius, cell_width;
    V3 origin; u32 dims[3];
    u32 upper_bound, initial_count, buckets;
};

// bluenoise.cpp:119-147
BluenoiseGrid bluenoise_grid(const TranslucentMesh& m, const float* transform, float radius, float* scale_out) {
    BluenoiseGrid gr;
    float scale = 0;
    for (int i = 0; i < 3; i++) { V3 row = ld3(&transform[4 * i]); scale += sqrtf(dot(row, row)); }
    scale /= 3;
    *scale_out = scale;
    gr.rejection_radius = radius / scale;
    gr.cell_width = (float) ((double) gr.rejection_radius / sqrt(3.0));
    V3 size = m.aabb_max - m.aabb_min;
    V3 dimensions = v3(ceilf(size.x / gr.cell_width + 0.5f), ceilf(size.y / gr.cell_width + 0.5f), ceilf(size.z / gr.cell_width + 0.5f));
    gr.origin = m.aabb_min - 0.5f * (gr.cell_width * dimensions - size);
    gr.dims[0] = (u32) dimensions.x; gr.dims[1] = (u32) dimensions.y; gr.dims[2] = (u32) dimensions.z;
    double rr = gr.rejection_radius;
    gr.upper_bound = (u32) ceil((double) m.total_surface_area / (0.5 * 6.28318530717958647692528676655900577 * 0.25 * rr * rr));
    gr.initial_count = 1;
    while (gr.initial_count < 16 * gr.upper_bound) gr.initial_count *= 2;
    gr.buckets = gr.upper_bound;
    return gr;
}

// bluenoise.hlsl:48-58
u32 cell_id_from_position(const BluenoiseGrid& gr, V3 p) {
    V3 c = (p - gr.origin) / gr.cell_width;
    float fx = floorf(c.x), fy = floorf(c.y), fz = floorf(c.z);
    u32 cx = fx > 0.0f ? (u32) fx : 0u, cy = fy > 0.0f ? (u32) fy : 0u, cz = fz > 0.0f ? (u32) fz : 0u;
    return cx + cy * gr.dims[0] + cz * gr.dims[0] * gr.dims[1];
}

// phase order: Fisher-Yates with xorshift32 in place of rand() (bluenoise.cpp:306-308)
void phase_sequence(u32 seed, u32 count, u32* seq) {
    u32 state = hash1(seed) | 1u;
    for (u32 i = 0; i < count; i++) seq[i] = i;
    for (u32 i = 0; i < count; i++) { u32 r = random_u32(state) % (count - i); std::swap(seq[r], seq[count - 1 - i]); }
}

u32 generate_sample_points(Scene& s, TranslucentInstance& ti, float radius, u32 rng_seed, u32 phase_seed) {
    const TranslucentMesh& mesh = s.translucent_meshes[ti.translucent_id];
    const Blas& b = s.blas[mesh.blas];
    BluenoiseGrid gr = bluenoise_grid(mesh, ti.transform, radius, &ti.scale_factor);
    u32 triangles_count = mesh.indices_count / 3;
    u32 N0 = gr.initial_count;
    std::vector<InitialSamplePoint> pts(N0);
    // generate_initial_sample_points, bluenoise.hlsl:60-103
    for (u32 index = 0; index < N0; index++) {
        u32 rng = hash2(index, rng_seed);
        float x = random01(rng) * mesh.total_surface_area;
        u32 min_index = 0, max_index = triangles_count - 1;
        while (min_index != max_index) {
            u32 i = (min_index + max_index) / 2;
            if (x > mesh.partial_surface_areas[i]) min_index = i + 1; else max_index = i;
        }
        InitialSamplePoint sp; sp.triangle_id = min_index;
        const u32* idx = &b.indices[mesh.ib_offset + 3 * sp.triangle_id];
        V3 p0 = ld3(b.vertices[idx[0]].position), p1 = ld3(b.vertices[idx[1]].position), p2 = ld3(b.vertices[idx[2]].position);
        float ux = random01(rng);
        float uy = random01(rng);
        ux = sqrtf(ux); uy = uy * ux; ux = 1.0f - ux;
        sp.position = p0 + ux * (p1 - p0) + uy * (p2 - p0);
        sp.cell_id = cell_id_from_position(gr, sp.position);
        pts[index] = sp;
    }
    // sort_initial_sample_points (bluenoise.hlsl:105-129): the order among equal cell ids is unspecified in the
    // reference (bitonic network); defined here as the stable order, i.e. by (cell_id, initial index)
    std::stable_sort(pts.begin(), pts.end(), [](const InitialSamplePoint& a, const InitialSamplePoint& b) { return a.cell_id < b.cell_id; });
    // build_hashtable (bluenoise.hlsl:133-158): 5 entries per bucket; which cells overflow depends on atomic order
    // in the reference; defined here as: the 5 smallest cell ids of each bucket are kept
    std::vector<std::vector<HashEntry> > table(gr.buckets);
    u32 entry_count = 0, overflow_count = 0;
    for (u32 i = 0; i < N0; i++) {
        if (i > 0 && pts[i - 1].cell_id == pts[i].cell_id) continue;
        std::vector<HashEntry>& bucket = table[pts[i].cell_id % gr.buckets];
        if (bucket.size() >= 5) { overflow_count++; continue; }
        HashEntry e; e.cell_id = pts[i].cell_id; e.first_index = i; e.selected = v3(F_INF, F_INF, F_INF);
        bucket.push_back(e); entry_count++;
    }
    (void) overflow_count;
    u32 phases = (s.flags & RT_FLAG_ALL_PHASE_GROUPS) ? 27u : 24u;
    u32 seq[27];
    phase_sequence(phase_seed, phases, seq);
    u32 capacity = entry_count;
    ti.sample_points.clear(); ti.point_normals.clear();
    auto search = [&](u32 cell_id) -> HashEntry* {
        std::vector<HashEntry>& bucket = table[cell_id % gr.buckets];
        for (auto& e : bucket) if (e.cell_id == cell_id) return &e;
        return (HashEntry*) NULL;
    };
    u32 gx = (gr.dims[0] + 2) / 3, gy = (gr.dims[1] + 2) / 3, gz = (gr.dims[2] + 2) / 3;
    float rr2 = gr.rejection_radius * gr.rejection_radius;
    u32 prev_count = 0;
    for (u32 trial_index = 0;; trial_index++) { // bluenoise.cpp:327-351
        for (u32 ph = 0; ph < phases; ph++) {
            u32 pg = seq[ph];
            u32 pgx = pg % 3, pgy = (pg / 3) % 3, pgz = pg / 9;
            // generate_sample_points, bluenoise.hlsl:180-245; new points are appended in dispatch-id order
            // (x fastest), which makes the reference's atomic append deterministic
            for (u32 iz = 0; iz < gz; iz++) for (u32 iy = 0; iy < gy; iy++) for (u32 ix = 0; ix < gx; ix++) {
                if (ti.sample_points.size() >= capacity) continue;
                u32 cx = 3 * ix + pgx, cy = 3 * iy + pgy, cz = 3 * iz + pgz;
                if (cx >= gr.dims[0] || cy >= gr.dims[1] || cz >= gr.dims[2]) continue;
                u32 cell_id = cx + cy * gr.dims[0] + cz * gr.dims[0] * gr.dims[1];
                HashEntry* entry = search(cell_id);
                if (!entry) continue;
                if (!std::isinf(entry->selected.x)) continue;
                u64 ti_index = (u64) entry->first_index + trial_index;
                if (ti_index >= N0) continue; // out-of-bounds structured read in the reference: defined as exhausted
                const InitialSamplePoint& trial = pts[ti_index];
                if (trial.cell_id != cell_id) continue;
                bool reject = false;
                for (int xo = -2; xo <= 2 && !reject; xo++) for (int yo = -2; yo <= 2 && !reject; yo++) for (int zo = -2; zo <= 2 && !reject; zo++) {
                    if ((xo == 0 && yo == 0 && zo == 0) || (abs(xo) == 2 && abs(yo) == 2 && abs(zo) == 2)) continue;
                    int nx = (int) cx + xo, ny = (int) cy + yo, nz = (int) cz + zo;
                    if (nx < 0 || ny < 0 || nz < 0 || nx >= (int) gr.dims[0] || ny >= (int) gr.dims[1] || nz >= (int) gr.dims[2]) continue;
                    HashEntry* ne = search((u32) nx + (u32) ny * gr.dims[0] + (u32) nz * gr.dims[0] * gr.dims[1]);
                    if (ne) {
                        V3 d = ne->selected - trial.position;
                        if (dot(d, d) <= rr2) reject = true;
                    }
                }
                if (reject) continue;
                entry->selected = trial.position;
                RtSamplePoint sp;
                V3 wp = xform_point44(ti.transform, trial.position);
                sp.position[0] = wp.x; sp.position[1] = wp.y; sp.position[2] = wp.z;
                sp.payload[0] = sp.payload[1] = sp.payload[2] = 0;
                ti.sample_points.push_back(sp);
                const u32* idx = &b.indices[mesh.ib_offset + 3 * trial.triangle_id];
                const RtVertex* v0 = &b.vertices[idx[0]]; const RtVertex* v1 = &b.vertices[idx[1]]; const RtVertex* v2 = &b.vertices[idx[2]];
                float bx, by;
                get_barycentrics(ld3(v0->position), ld3(v1->position), ld3(v2->position), trial.position, &bx, &by);
                V3 normal = get_interpolated_normal(v0, v1, v2, bx, by);
                normal = normalize(xform_dir44(ti.transform, normal));
                ti.point_normals.push_back(normal);
            }
        }
        if (ti.sample_points.size() == prev_count) break;
        prev_count = (u32) ti.sample_points.size();
    }
    ti.samples_count = (u32) ti.sample_points.size();
    return ti.samples_count;
}

} // namespace

// ---- C interface (same shapes as include/rt_abi.h with an orc_ prefix) ---------------------------------------
extern "C" {

struct orc_scene { Scene s; };

orc_scene* orc_create(void) { return new orc_scene(); }
void orc_destroy(orc_scene* o) { delete o; }

int orc_set_options(orc_scene* o, int brute_force, int threads) {
    o->s.brute_force = brute_force != 0; o->s.threads = threads > 0 ? threads : 1; return 0;
}

int orc_update_resolution(orc_scene* o, uint32_t w, uint32_t h) { // raytracing.cpp:143-187
    Scene& s = o->s; s.width = w; s.height = h;
    s.accumulator.assign((size_t) w * h * 4, 0.0f); s.render_target.assign((size_t) w * h * 4, 0);
    return 0;
}

int orc_load_bssrdf_table(orc_scene* o, int profile, const float* r, const float* g, const float* b, uint32_t n) {
    if (profile < 0 || profile > 1) return 1;
    uint32_t padded = ((n + 255u) / 256u) * 256u; // round_up(data_len, D3D12_TEXTURE_DATA_PITCH_ALIGNMENT), raytracing.cpp:113
    std::vector<float>& t = o->s.bssrdf[profile];
    t.assign((size_t) padded * 3, 0.0f);
    for (uint32_t i = 0; i < n; i++) { t[3 * i] = r[i]; t[3 * i + 1] = g[i]; t[3 * i + 2] = b[i]; }
    o->s.bssrdf_texels[profile] = padded;
    return 0;
}
int orc_set_bssrdf_profile(orc_scene* o, int profile) { if (profile < 0 || profile > 1) return 1; o->s.profile = profile; return 0; }

// raytracing.cpp:316-444
int orc_build_blas(orc_scene* o, const RtGeometry* geometries, int n, rt_blas* out) {
    Scene& s = o->s;
    Blas b;
    b.translucent_ids_index = (u32) s.translucent_mean_area.size();
    b.translucent_ids_count = 0;
    for (int gi = 0; gi < n; gi++) {
        const RtGeometry& ge = geometries[gi];
        Geometry geo;
        geo.index_offset = (u32) b.indices.size(); geo.index_count = (u32) ge.index_count; geo.first_tri = geo.index_offset / 3;
        geo.shader = ge.shader; geo.color = v3(ge.color[0], ge.color[1], ge.color[2]); geo.translucent_id = -1;
        for (size_t k = 0; k < ge.index_count; k++) b.indices.push_back((u32) (uint16_t) (b.vertices.size() + ge.indices[k]));
        for (size_t k = 0; k < ge.vertex_count; k++) b.vertices.push_back(ge.vertices[k]);
        for (size_t k = 0; k < ge.index_count / 3; k++) b.tri_geometry.push_back(gi);
        b.geometries.push_back(geo);
    }
    if (b.vertices.size() > 65535) return RT_ERR_LIMIT; // raytracing.cpp:367
    int blas_index = (int) s.blas.size();
    for (int gi = 0; gi < n; gi++) {
        if (b.geometries[gi].shader != RT_SHADER_TRANSLUCENT) continue;
        b.translucent_ids_count++;
        b.geometries[gi].translucent_id = (int) s.translucent_mean_area.size();
        TranslucentMesh m; m.blas = blas_index; m.ib_offset = b.geometries[gi].index_offset; m.indices_count = b.geometries[gi].index_count;
        preprocess_mesh_data(b, m);
        s.translucent_meshes.push_back(m);
        s.translucent_mean_area.push_back(0.0f);
    }
    build_bvh(b);
    s.blas.push_back(b);
    *out = blas_index;
    return 0;
}

// raytracing.cpp:446-551
int orc_build_tlas(orc_scene* o, const RtInstance* instances, int n) {
    Scene& s = o->s;
    s.translucent_instances.clear();
    s.g.translucent_instance_stride = 0;
    s.instances.clear();
    std::vector<u32> counts(s.blas.size(), 0);
    for (int i = 0; i < n; i++) {
        if (instances[i].blas < 0 || instances[i].blas >= (int) s.blas.size()) return RT_ERR_INVALID;
        Instance in; memcpy(in.m, instances[i].transform, sizeof in.m); in.blas = instances[i].blas;
        counts[in.blas] += 1;
        in.instance_id = counts[in.blas] - 1;
        invert_affine(in.m, in.inv);
        const Blas& b = s.blas[in.blas];
        in.wlo = v3(F_INF, F_INF, F_INF); in.whi = v3(-F_INF, -F_INF, -F_INF);
        for (size_t k = 0; k < b.indices.size(); k++) {
            V3 p = xform_point44(in.m, ld3(b.vertices[b.indices[k]].position));
            in.wlo = v3(std::min(in.wlo.x, p.x), std::min(in.wlo.y, p.y), std::min(in.wlo.z, p.z));
            in.whi = v3(std::max(in.whi.x, p.x), std::max(in.whi.y, p.y), std::max(in.whi.z, p.z));
        }
        V3 pad = 1e-5f * (in.whi - in.wlo) + v3(1e-6f, 1e-6f, 1e-6f);
        in.wlo = in.wlo - pad; in.whi = in.whi + pad;
        if (b.translucent_ids_count > 0) {
            for (u32 j = 0; j < b.translucent_ids_count; j++) {
                TranslucentInstance ti;
                ti.translucent_id = b.translucent_ids_index + j; ti.instance_id = in.instance_id;
                memcpy(ti.transform, in.m, sizeof ti.transform); ti.scale_factor = 0; ti.samples_count = 0; ti.slot = 0;
                s.translucent_instances.push_back(ti);
            }
            s.g.translucent_instance_stride = std::max(s.g.translucent_instance_stride, counts[in.blas]);
        }
        s.instances.push_back(in);
    }
    build_tlas_bvh(s);
    return 0;
}

// raytracing.cpp:553-650
int orc_generate_translucent_samples(orc_scene* o, float radius, uint32_t rng_seed, uint32_t phase_seed, uint32_t* total) {
    Scene& s = o->s;
    *total = 0;
    if (s.g.translucent_instance_stride == 0) return 0;
    u32 stride = s.g.translucent_instance_stride;
    s.properties_by_slot.assign(s.translucent_mean_area.size() * stride, 0.0f);
    s.slot_to_instance.assign(s.translucent_mean_area.size() * stride, -1);
    for (size_t i = 0; i < s.translucent_instances.size(); i++) {
        TranslucentInstance& ti = s.translucent_instances[i];
        const TranslucentMesh& mesh = s.translucent_meshes[ti.translucent_id];
        generate_sample_points(s, ti, radius, rng_seed + (u32) i, phase_seed);
        float mean_area = ti.scale_factor * ti.scale_factor * mesh.total_surface_area / (float) ti.samples_count;
        s.translucent_mean_area[ti.translucent_id] = mean_area;
        *total += ti.samples_count;
        ti.slot = ti.translucent_id * stride + ti.instance_id;
        s.properties_by_slot[ti.slot] = mean_area;
        s.slot_to_instance[ti.slot] = (int) i;
        ti.write_sample_points = ti.sample_points; // clone, raytracing.cpp:607-621
        hg_build(ti);
        hg_refresh(ti);
    }
    return 0;
}

int orc_set_globals(orc_scene* o, const RtGlobals* g) {
    uint32_t stride = o->s.g.translucent_instance_stride;
    o->s.g = *g; o->s.g.translucent_instance_stride = stride; return 0;
}
int orc_get_globals(orc_scene* o, RtGlobals* g) { *g = o->s.g; return 0; }
int orc_set_flags(orc_scene* o, uint32_t flags) { o->s.flags = flags; return 0; }
int orc_set_gather_mode(orc_scene* o, int hierarchical, float max_solid_angle) {
    if (hierarchical && !(max_solid_angle >= 0.0f)) return RT_ERR_INVALID;
    o->s.hg_enabled = hierarchical != 0; o->s.hg_max_solid_angle = max_solid_angle; return 0;
}
int orc_set_tile_split(orc_scene* o, uint32_t rank, uint32_t world) { o->s.tile_rank = rank; o->s.tile_world = world ? world : 1; return 0; }

// raytracing.cpp:652-747.  Default (no RT_FLAG_DOUBLE_DISPATCH): probe pass -> copy -> one camera pass.
int orc_dispatch_rays(orc_scene* o) {
    Scene& s = o->s;
    if (s.width == 0 || s.instances.empty()) return RT_ERR_STATE;
    bool collect = (s.flags & RT_FLAG_COLLECT_SAMPLES) != 0 && !s.translucent_instances.empty() && !s.properties_by_slot.empty();
    if (collect) probe_pass(s);
    if (s.flags & RT_FLAG_DOUBLE_DISPATCH) camera_pass(s);
    if (collect) for (auto& ti : s.translucent_instances) ti.sample_points = ti.write_sample_points; // raytracing.cpp:723-728
    if (s.hg_enabled) for (auto& ti : s.translucent_instances) hg_refresh(ti);
    camera_pass(s);
    s.g.accumulator_count += 1;
    s.g.translucent_accumulator_count += ((s.flags & RT_FLAG_COLLECT_SAMPLES) != 0) ? 1u : 0u;
    return 0;
}

int orc_read_accumulator(orc_scene* o, float* rgba) { memcpy(rgba, o->s.accumulator.data(), o->s.accumulator.size() * 4); return 0; }
int orc_read_render_target(orc_scene* o, uint8_t* rgba8) { memcpy(rgba8, o->s.render_target.data(), o->s.render_target.size()); return 0; }
int orc_read_hit_records(orc_scene* o, uint32_t bounce, RtHit* out) {
    if (bounce >= o->s.hit_records.size()) return RT_ERR_INVALID;
    memcpy(out, o->s.hit_records[bounce].data(), o->s.hit_records[bounce].size() * sizeof(RtHit)); return 0;
}
int orc_translucent_instance_count(orc_scene* o, uint32_t* n) { *n = (uint32_t) o->s.translucent_instances.size(); return 0; }
int orc_sample_point_count(orc_scene* o, uint32_t instance, uint32_t* n) {
    if (instance >= o->s.translucent_instances.size()) return RT_ERR_INVALID;
    *n = o->s.translucent_instances[instance].samples_count; return 0;
}
int orc_read_sample_points(orc_scene* o, uint32_t instance, RtSamplePoint* points, float* normals) {
    if (instance >= o->s.translucent_instances.size()) return RT_ERR_INVALID;
    const TranslucentInstance& ti = o->s.translucent_instances[instance];
    if (points) memcpy(points, ti.sample_points.data(), ti.sample_points.size() * sizeof(RtSamplePoint));
    if (normals) for (size_t i = 0; i < ti.point_normals.size(); i++) { normals[3 * i] = ti.point_normals[i].x; normals[3 * i + 1] = ti.point_normals[i].y; normals[3 * i + 2] = ti.point_normals[i].z; }
    return 0;
}
int orc_get_counters(orc_scene* o, uint64_t* rays, uint64_t* nodes, uint64_t* tris, uint64_t* insts) {
    *rays = o->s.counters.rays; *nodes = o->s.counters.nodes; *tris = o->s.counters.tris; *insts = o->s.counters.insts; return 0;
}
int orc_reset_counters(orc_scene* o) { memset(&o->s.counters, 0, sizeof o->s.counters); return 0; }

// ---- unit-level taps used by tests/ to pin the oracle against independent numpy restatements ---------------------
uint32_t orc_hash1(uint32_t x) { return hash1(x); }
uint32_t orc_hash2(uint32_t x, uint32_t y) { return hash2(x, y); }
uint32_t orc_hash3(uint32_t x, uint32_t y, uint32_t z) { return hash3(x, y, z); }
uint32_t orc_random_u32(uint32_t* seed) { return random_u32(*seed); }
float orc_random01(uint32_t* seed) { return random01(*seed); }
float orc_random11(uint32_t* seed) { return random11(*seed); }
void orc_sincos(float phi, float* s, float* c) { sincos_spec(phi, s, c); }
void orc_random_on_sphere(uint32_t* seed, float* out) { V3 v = random_on_sphere(*seed); out[0] = v.x; out[1] = v.y; out[2] = v.z; }
void orc_random_on_hemisphere(uint32_t* seed, const float* n, float* out) { V3 v = random_on_hemisphere(*seed, ld3(n)); out[0] = v.x; out[1] = v.y; out[2] = v.z; }
float orc_schlick(float n, float c) { return schlick(n, c); }
void orc_bssrdf_dipole(orc_scene* o, float radius, float* rgb) { V3 v = eval_bssrdf_dipole(o->s, radius); rgb[0] = v.x; rgb[1] = v.y; rgb[2] = v.z; }
void orc_bssrdf_tabulated(orc_scene* o, float radius, float* rgb) { V3 v = eval_bssrdf_tabulated(o->s, radius); rgb[0] = v.x; rgb[1] = v.y; rgb[2] = v.z; }
int orc_intersect_triangle(const float* o3, const float* d3, const float* p0, const float* p1, const float* p2, float* tuv) {
    RayPrep r = prepare_ray(ld3(o3), ld3(d3));
    return intersect_triangle(r, ld3(p0), ld3(p1), ld3(p2), &tuv[0], &tuv[1], &tuv[2]) ? 1 : 0;
}
void orc_invert_affine(const float* m16, float* inv12) { invert_affine(m16, inv12); }
void orc_phase_sequence(uint32_t seed, uint32_t count, uint32_t* seq) { phase_sequence(seed, count, seq); }
// single ray through the scene (camera TMin/TMax), for BVH-vs-brute-force checks
int orc_trace_ray(orc_scene* o, const float* o3, const float* d3, float tmin, float tmax, RtHit* out) {
    Counters c; memset(&c, 0, sizeof c);
    Hit h = trace_ray(o->s, ld3(o3), ld3(d3), tmin, tmax, c);
    out->instance = h.inst; out->geometry = h.geom; out->primitive = h.prim; out->t = h.t; out->u = h.u; out->v = h.v; out->rng_after = 0; out->valid = 1;
    return 0;
}

} // extern "C"
