// rt_dev_math.h -- per-ray arithmetic of the MI355X path tracer (device functions).
//
// Everything that decides where a ray goes (RNG, samplers, normals, the ray/triangle test) is written with
// separate IEEE multiplies and adds (the library is compiled with -ffp-contract=off) so that closest-hit ids
// are reproducible bit for bit; see DESIGN.md "Arithmetic contract" S1..S9.  Citations are file:line in the
// reference tree (gluyas/raytracer).
//
// Device code only: the header is compiled by hipcc alone (its host pass merely parses it).
#pragma once
#include <stdint.h>
#include <math.h>
#include <string.h>

#if !defined(__HIPCC__)
#error "rt_dev_math.h holds gfx950 device code: compile with hipcc"
#endif
#define RT_DEV __device__ __forceinline__
#if defined(__HIP_DEVICE_COMPILE__)
// explicit address spaces: pointers that were loaded from memory (instance records) or merged in a phi are "flat" to
// the compiler, and flat_load/flat_store are slower than global_load / ds_read and serialise the wait counters
#define RT_GLOBAL __attribute__((address_space(1)))
#define RT_LDS    __attribute__((address_space(3)))
#else
#define RT_GLOBAL
#define RT_LDS
#endif

namespace rtd {

typedef uint32_t u32;

// wave votes.  The ballot of a compare IS the compare's lane mask; the ballot of a conjunction of two compares is lowered as
// v_cndmask 0, 1 + v_cmp_ne (two vector instructions per vote, in the hottest loop of the flat walk): any_lane_both takes the
// two ballots and combines them as scalars instead.
#if defined(__HIP_DEVICE_COMPILE__)
RT_DEV bool any_lane(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }
RT_DEV bool all_lanes(bool p) { return __builtin_amdgcn_ballot_w64(!p) == 0ull; }
RT_DEV bool any_lane_both(bool p, bool q) { return (__builtin_amdgcn_ballot_w64(p) & __builtin_amdgcn_ballot_w64(q)) != 0ull; }
#else
RT_DEV bool any_lane(bool p) { return p; }
RT_DEV bool all_lanes(bool p) { return p; }
RT_DEV bool any_lane_both(bool p, bool q) { return p && q; }
#endif

struct f3 { float x, y, z; };

RT_DEV f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
RT_DEV f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
RT_DEV f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
RT_DEV f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
RT_DEV f3 operator*(float s, f3 a) { return mk3(s * a.x, s * a.y, s * a.z); }
RT_DEV f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
RT_DEV f3 operator/(f3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }
RT_DEV float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }          // S3
RT_DEV f3 normalize3(f3 a) { float r = 1.0f / sqrtf(dot3(a, a)); return a * r; }      // S2
RT_DEV bool any3(f3 a) { return a.x != 0.0f || a.y != 0.0f || a.z != 0.0f; }
RT_DEV float comp3(f3 a, int k) { return k == 0 ? a.x : (k == 1 ? a.y : a.z); }
RT_DEV float as_f32(u32 u) { return __builtin_bit_cast(float, u); }
RT_DEV u32 as_u32(float f) { return __builtin_bit_cast(u32, f); }
RT_DEV float inf_f32() { return as_f32(0x7f800000u); }

// ---- random.hlsl ------------------------------------------------------------------------------------------
RT_DEV u32 wang_hash(u32 seed) {                      // random.hlsl:6-15
    seed = (seed ^ 61u) ^ (seed >> 16);
    seed *= 9u;
    seed = seed ^ (seed >> 4);
    seed *= 0x27d4eb2du;
    seed = seed ^ (seed >> 15);
    return seed;
}
RT_DEV u32 hash_u2(u32 x, u32 y) { return wang_hash(y) + 31u * wang_hash(x); }                   // random.hlsl:17-19
RT_DEV u32 hash_u3(u32 x, u32 y, u32 z) { return wang_hash(z) + 31u * hash_u2(x, y); }            // random.hlsl:21-23
RT_DEV u32 xorshift32(u32& s) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; }              // random.hlsl:30-36
RT_DEV float random01(u32& s) { return as_f32(0x3f800000u | (0x007fffffu & xorshift32(s))) - 1.0f; } // :39-41
RT_DEV float random11(u32& s) {                                                                  // :44-48
    u32 r = xorshift32(s);
    float r01 = as_f32(0x3f800000u | (0x007fffffu & r)) - 1.0f;
    return as_f32(as_u32(r01) | (r & 0x80000000u));
}

// S1: sin and cos of phi in [0, 2 pi]
RT_DEV void sincos_2pi(float phi, float& s_out, float& c_out) {
    int   k  = (int) (phi * 0.636619772f + 0.5f);
    float fk = (float) k;
    float r  = phi - fk * 1.5703125f;
    r = r - fk * 4.837512969970703125e-4f;
    r = r - fk * 7.54978995489188e-8f;
    float z = r * r;
    float ps = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r + r;
    float pc = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z
             - 0.5f * z + 1.0f;
    int q = k & 3;
    float s = (q & 1) ? pc : ps;
    float c = (q & 1) ? ps : pc;
    s_out = (q & 2) ? -s : s;
    c_out = (q == 1 || q == 2) ? -c : c;
}

#define RT_TAU_F 6.28318530717958647692528676655900577f

RT_DEV f3 random_on_sphere(u32& seed) {               // random.hlsl:70-75
    float phi       = random01(seed) * RT_TAU_F;
    float cos_theta = random11(seed);
    float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
    float s, c;
    sincos_2pi(phi, s, c);
    return mk3(sin_theta * c, sin_theta * s, cos_theta);
}
RT_DEV f3 random_on_hemisphere(u32& seed, f3 normal) { // random.hlsl:85-88
    f3 unit = random_on_sphere(seed);
    float d = 2.0f * dot3(normal, unit);
    float m = d < 0.0f ? d : 0.0f;
    return unit - m * normal;
}

RT_DEV float sign_f32(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }

RT_DEV float schlick(float refractive_index, float cosine) { // raytracing.hlsl:211-222
    float r0 = (refractive_index - 1.0f) / (refractive_index + 1.0f);
    r0 *= r0;
    float fresnel = 1.0f - cosine;
    fresnel *= fresnel * fresnel * fresnel * fresnel;
    fresnel *= 1.0f - r0;
    fresnel += r0;
    return fresnel;
}

// ---- transforms (row vectors) -------------------------------------------------------------------------------
// 4x3 stored as 12 floats: rows 0..2 = linear part, row 3 = translation
RT_DEV f3 xform_point43(const float* m, f3 p) {
    return mk3(p.x * m[0] + p.y * m[3] + p.z * m[6] + m[9],
               p.x * m[1] + p.y * m[4] + p.z * m[7] + m[10],
               p.x * m[2] + p.y * m[5] + p.z * m[8] + m[11]);
}
RT_DEV f3 xform_dir43(const float* m, f3 p) {
    return mk3(p.x * m[0] + p.y * m[3] + p.z * m[6],
               p.x * m[1] + p.y * m[4] + p.z * m[7],
               p.x * m[2] + p.y * m[5] + p.z * m[8]);
}
RT_DEV f3 xform_dir44(const float* m, f3 p) {          // mul(float4(p, 0), M).xyz
    return mk3(p.x * m[0] + p.y * m[4] + p.z * m[8],
               p.x * m[1] + p.y * m[5] + p.z * m[9],
               p.x * m[2] + p.y * m[6] + p.z * m[10]);
}
RT_DEV f3 xform_point44(const float* m, f3 p) {        // mul(float4(p, 1), M).xyz
    return mk3(p.x * m[0] + p.y * m[4] + p.z * m[8] + m[12],
               p.x * m[1] + p.y * m[5] + p.z * m[9] + m[13],
               p.x * m[2] + p.y * m[6] + p.z * m[10] + m[14]);
}

// ---- S4: watertight ray/triangle test, back faces culled -----------------------------------------------------
// S3 (Woop et al.): kz = the axis of the largest |d| (the first one on ties), kx = kz + 1, ky = kz + 2 (mod 3), kx and
// ky exchanged when d[kz] < 0 (keeps the winding); the ray is sheared and scaled so that it runs along +z.
// The triangles are stored three times, with their coordinates in the orders (1, 2, 0), (2, 0, 1) and (0, 1, 2)
// (TriRec, rt_dev_scene.h), and a ray reads the copy of its kz: the per-vertex permutation (18 selects per triangle)
// is done once per ray, on its origin.  The exchange of kx and ky is not applied to the coordinates either: with the
// x and y roles exchanged an edge function Cx By - Cy Bx becomes Cy Bx - Cx By = (-Cx) By - Cy (-Bx) -- each product
// has one x and one y factor, a - b is a + (-b) and the sum commutes -- so the test runs in the unexchanged order with
// the three sheared x coordinates negated (`flip`): every operation of the contract gets the same operands (up to
// their order in commutative operations), hence the same t, u, v, bit for bit, signs of zeros included.
struct RayPrep {
    f3 o;              // origin in the order (kz + 1, kz + 2, kz)
    int kz;
    u32 flip;          // 0x80000000 when d[kz] < 0
    float sx, sy, sz;  // d[kz + 1] / d[kz], d[kz + 2] / d[kz], 1 / d[kz]
};

// SHARED_DIVISOR: see below; for callers that prepare the rays of a whole wave together (the flat walk)
template <bool SHARED_DIVISOR = false>
RT_DEV RayPrep prepare_ray(f3 o, f3 d) {
    RayPrep r;
    const float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
    const bool y_over_x = ay > ax;
    const bool z_over = az > (y_over_x ? ay : ax);
    const int kz = z_over ? 2 : (y_over_x ? 1 : 0);
    const float dz = z_over ? d.z : (y_over_x ? d.y : d.x);
    const float d1 = z_over ? d.x : (y_over_x ? d.z : d.y); // d[kz + 1]
    const float d2 = z_over ? d.y : (y_over_x ? d.x : d.z); // d[kz + 2]
    r.o = mk3(z_over ? o.x : (y_over_x ? o.z : o.y), z_over ? o.y : (y_over_x ? o.x : o.z), z_over ? o.z : (y_over_x ? o.y : o.x));
    r.kz = kz;
    r.flip = dz < 0.0f ? 0x80000000u : 0u;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(RT_NO_SHARED_DIV)
    // The three quotients share their divisor.  The compiler's IEEE division is v_div_scale (both operands), v_rcp, two
    // Newton steps on the reciprocal, the quotient with two residual corrections, v_div_fmas, v_div_fixup: 11 instructions,
    // of which the scaling and the fix-up do nothing unless an operand, the reciprocal or the quotient comes near the ends
    // of the exponent range.  With all three operands in [2^-60, 2^60] (|d1|, |d2| <= |dz| by the choice of kz) they are
    // idle, and the same sequence, operation for operation, needs the reciprocal's part once: 17 instructions for the three
    // quotients instead of 33, the same bits.  A wave with any lane outside the range (a zero component, say) divides as before.
    // Same-box A/B (profiles/r02_ab_shared_divisor.txt): C2 +1.0 %, C3 +0.45 %; at the instance entries of the two-level walk,
    // where a few lanes of a wave prepare at a time, the vote costs more than the divisions saved (C5 -0.3 %): not used there.
    const float small = fminf(fabsf(d1), fabsf(d2));
    if (SHARED_DIVISOR && all_lanes(fabsf(dz) <= 0x1p60f && small >= 0x1p-60f)) {
        float rc = __builtin_amdgcn_rcpf(dz);
        rc = __builtin_fmaf(__builtin_fmaf(-dz, rc, 1.0f), rc, rc);
        float q = d1 * rc;
        q = __builtin_fmaf(__builtin_fmaf(-dz, q, d1), rc, q);
        r.sx = __builtin_fmaf(__builtin_fmaf(-dz, q, d1), rc, q);
        q = d2 * rc;
        q = __builtin_fmaf(__builtin_fmaf(-dz, q, d2), rc, q);
        r.sy = __builtin_fmaf(__builtin_fmaf(-dz, q, d2), rc, q);
        q = __builtin_fmaf(__builtin_fmaf(-dz, rc, 1.0f), rc, rc);
        r.sz = __builtin_fmaf(__builtin_fmaf(-dz, q, 1.0f), rc, q);
        return r;
    }
#endif
    r.sx = d1 / dz;
    r.sy = d2 / dz;
    r.sz = 1.0f / dz;
    return r;
}

// The barycentrics of a candidate are returned as the numerators V, W and the denominator det: only the closest hit
// of a ray needs u = V / det and v = W / det, so the caller divides once per ray (same operands, same quotient).
// p0, p1, p2: the vertices from the triangle copy of the ray's kz (coordinates in the order of r.o)
RT_DEV bool intersect_triangle(const RayPrep& r, f3 p0, f3 p1, f3 p2, float& t_out, float& v_num, float& w_num, float& det_out) {
    f3 A = p0 - r.o, B = p1 - r.o, C = p2 - r.o;
    float Akz = A.z, Bkz = B.z, Ckz = C.z;
    float Ax = A.x - r.sx * Akz, Ay = A.y - r.sy * Akz;
    float Bx = B.x - r.sx * Bkz, By = B.y - r.sy * Bkz;
    float Cx = C.x - r.sx * Ckz, Cy = C.y - r.sy * Ckz;
    Ax = as_f32(as_u32(Ax) ^ r.flip); Bx = as_f32(as_u32(Bx) ^ r.flip); Cx = as_f32(as_u32(Cx) ^ r.flip); // kx and ky exchanged (see RayPrep)
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
    float Az = r.sz * Akz, Bz = r.sz * Bkz, Cz = r.sz * Ckz;
    float T = U * Az + V * Bz + W * Cz;
    t_out = T / det;
    v_num = V; w_num = W; det_out = det;
    return true;
}

// One triangle candidate against the closest hit so far (S4-S6): intersect_triangle and the closest-hit rule in one
// piece, written as nested conditions.  Each condition is a wave-level branch (an exec mask: scalar work), and what sits
// behind it -- the division, the tie rule, the seven moves of an update -- costs vector instructions only when some lane
// gets there; the values a rejected candidate would have carried are never made up.  The exact-zero test of the edge
// functions is one min3 of magnitudes instead of three compares.  Same operands, same results as the two separate steps.
struct ClosestHit {
    float t, u_num, v_num, det; // u = u_num / det, v = v_num / det once the ray is finished (S4); t starts at TMax
    int inst, tri;              // inst < 0: no hit yet
    u32 prim_global;
};
RT_DEV void triangle_candidate(const RayPrep& r, f3 p0, f3 p1, f3 p2, float tmin, int inst, int tri, u32 prim_global, ClosestHit& best) {
    f3 A = p0 - r.o, B = p1 - r.o, C = p2 - r.o;
    float Akz = A.z, Bkz = B.z, Ckz = C.z;
    float Ax = A.x - r.sx * Akz, Ay = A.y - r.sy * Akz;
    float Bx = B.x - r.sx * Bkz, By = B.y - r.sy * Bkz;
    float Cx = C.x - r.sx * Ckz, Cy = C.y - r.sy * Ckz;
    Ax = as_f32(as_u32(Ax) ^ r.flip); Bx = as_f32(as_u32(Bx) ^ r.flip); Cx = as_f32(as_u32(Cx) ^ r.flip); // kx and ky exchanged (see RayPrep)
    float U = Cx * By - Cy * Bx;
    float V = Ax * Cy - Ay * Cx;
    float W = Bx * Ay - By * Ax;
    if (fminf(fminf(fabsf(U), fabsf(V)), fabsf(W)) == 0.0f) { // U == 0 || V == 0 || W == 0 (a NaN is not a zero either way)
        double CxBy = (double) Cx * (double) By, CyBx = (double) Cy * (double) Bx;
        U = (float) (CxBy - CyBx);
        double AxCy = (double) Ax * (double) Cy, AyCx = (double) Ay * (double) Cx;
        V = (float) (AxCy - AyCx);
        double BxAy = (double) Bx * (double) Ay, ByAx = (double) By * (double) Ax;
        W = (float) (BxAy - ByAx);
    }
    if (U < 0.0f || V < 0.0f || W < 0.0f) return;
    const float det = U + V + W;
    if (det == 0.0f) return;
    const float Az = r.sz * Akz, Bz = r.sz * Bkz, Cz = r.sz * Ckz;
    const float T = U * Az + V * Bz + W * Cz;
    const float t = T / det;
    if (!(t > tmin)) return;    // t < TMax follows from the rule below: best.t starts at TMax and only decreases
    bool better = t < best.t;
    if (t == best.t && best.inst >= 0) // exact tie: the lowest (instance, primitive) wins (S5)
        better = inst < best.inst || (inst == best.inst && prim_global < best.prim_global);
    if (better) { best.t = t; best.u_num = V; best.v_num = W; best.det = det; best.inst = inst; best.tri = tri; best.prim_global = prim_global; }
}

// ---- BSSRDF ---------------------------------------------------------------------------------------------------
// S8: linear filtering of the zero-bordered 1-D table at texel coordinate u*N - 0.5 (raytracing.hlsl:275)
RT_DEV f3 sample_bssrdf_table(const float* tab /* rgb interleaved */, int n, float u) {
    float x = u * (float) n - 0.5f;
    float fl = floorf(x);
    float f = x - fl;
    if (!(fl >= -1.0f && fl < (float) n)) return mk3(0.0f, 0.0f, 0.0f);
    int i0 = (int) fl, i1 = i0 + 1;
    f3 a = mk3(0.0f, 0.0f, 0.0f), b = a;
    if (i0 >= 0 && i0 < n) a = mk3(tab[3 * i0], tab[3 * i0 + 1], tab[3 * i0 + 2]);
    if (i1 >= 0 && i1 < n) b = mk3(tab[3 * i1], tab[3 * i1 + 1], tab[3 * i1 + 2]);
    return a + f * (b - a);
}

// raytracing.hlsl:278-301, one channel.  The radius-independent terms are hoisted by the caller (DipoleConst).
struct DipoleConst {
    float z_real, z_virtual, sigma_tr, scale; // scale = albedo / (2 tau)
    float exp2_rate, a_real, a_virtual;       // dipole_eval_fast only: -sigma_tr log2(e), scale z exp(-sigma_tr z) per pole
};
RT_DEV DipoleConst dipole_constants(float sigma_s, float sigma_a, float eta) {
    DipoleConst c;
    float attenuation = sigma_s + sigma_a;
    float mean_free_path = 1.0f / attenuation;
    float albedo = sigma_s / attenuation;
    c.sigma_tr = sqrtf(3.0f * sigma_s * sigma_a);
    float diffuse_fresnel = -1.440f / (eta * eta) + 0.710f / eta + 0.668f + 0.0636f * eta;
    c.z_real = mean_free_path;
    c.z_virtual = mean_free_path * (1.0f + 1.25f * (1.0f + diffuse_fresnel) / (1.0f - diffuse_fresnel));
    c.scale = albedo / (2.0f * RT_TAU_F);
    c.exp2_rate = -c.sigma_tr * 1.44269504088896341f;
    c.a_real = c.scale * c.z_real * expf(-c.sigma_tr * c.z_real);
    c.a_virtual = c.scale * c.z_virtual * expf(-c.sigma_tr * c.z_virtual);
    return c;
}
RT_DEV float dipole_eval(const DipoleConst& c, float radius) {
    float d_real = radius + c.z_real;
    float c_real = c.z_real * (c.sigma_tr + 1.0f / d_real);
    float d_virtual = radius + c.z_virtual;
    float c_virtual = c.z_virtual * (c.sigma_tr + 1.0f / d_virtual);
    float m_real = c_real * expf(-c.sigma_tr * d_real) / (d_real * d_real);
    float m_virtual = c_virtual * expf(-c.sigma_tr * d_virtual) / (d_virtual * d_virtual);
    float r = c.scale * (m_real + m_virtual);
    return r > 0.0f ? r : 0.0f;
}

// Hardware approximations (1-2 ulp), used only where the result is covered by a tolerance and never decides a hit id
// or an RNG draw: the conservative box test and the BSSRDF gather (radiance, 1e-4).
RT_DEV float fast_rcp(float x) {
    return __builtin_amdgcn_rcpf(x);
}
RT_DEV float fast_sqrt(float x) {
    return __builtin_amdgcn_sqrtf(x);
}
// a / b, correctly rounded, from inv_b = RN(1 / b): q = RN(a inv_b) is within an ulp of the quotient, r = a - b q is exact in
// one fused multiply-add, and RN(q + r inv_b) is the correctly rounded quotient (Markstein 1990; the theorem's exception,
// a divisor whose significand is all ones, and results near the ends of the exponent range are the caller's to exclude).
// Used for the pixel -> screen division by the image size (integers <= 32768, numerators below 2^17): checked against
// the IEEE quotient on 2.4e8 operand pairs over all divisors up to 8192 (no mismatch).
RT_DEV float quotient_by_reciprocal(float a, float b, float inv_b) {
    const float q = a * inv_b;
    return __builtin_fmaf(__builtin_fmaf(-b, q, a), inv_b, q);
}
// x / y where the caller supplies inv_y = 1 / y (IEEE) of a divisor shared by many quotients: one Newton step on the
// product.  Three instructions instead of the ~11 of a division; equals the correctly rounded quotient except in rare
// half-way cases (then 1 ulp off), for finite operands without overflow -- good for tolerance-covered values only.
RT_DEV float div_by_shared(float x, float y, float inv_y) {
    float q = x * inv_y;
    float r = __builtin_fmaf(-y, q, x);
    return __builtin_fmaf(r, inv_y, q);
}
// sqrt(x), x >= 0 and normal: hardware reciprocal square root refined by one Heron step (5 instructions instead of
// the ~12 of the correctly rounded library sequence); at most 1 ulp off, exact for 0
RT_DEV float sqrt_refined(float x) {
    float y = __builtin_amdgcn_rsqf(x);
    float s = x * y, h = 0.5f * y;
    float r = __builtin_fmaf(-s, s, x);
    float t = __builtin_fmaf(r, h, s);
    return x > 0.0f ? t : 0.0f;
}
RT_DEV float fast_exp(float x) {
    return __builtin_amdgcn_exp2f(x * 1.44269504088896341f);
}
// dipole_eval with hardware rcp / exp2 and exp(-sigma (r + z)) split as exp(-sigma r) exp(-sigma z), so that both poles
// share one exponential per pair and the per-pole factors come from dipole_constants: the same function to a few ulp
RT_DEV float dipole_eval_fast(const DipoleConst& c, float radius) {
    float ir = fast_rcp(radius + c.z_real), iv = fast_rcp(radius + c.z_virtual);
    float e = __builtin_amdgcn_exp2f(radius * c.exp2_rate);
    float m_real = (c.sigma_tr + ir) * (ir * ir);
    float m_virtual = (c.sigma_tr + iv) * (iv * iv);
    float r = e * (c.a_real * m_real + c.a_virtual * m_virtual);
    return r > 0.0f ? r : 0.0f;
}

RT_DEV unsigned char to_unorm8(float x) {             // S9
    if (!(x > 0.0f)) return 0;
    if (x > 1.0f) x = 1.0f;
    return (unsigned char) (int) (x * 255.0f + 0.5f);
}

} // namespace rtd
