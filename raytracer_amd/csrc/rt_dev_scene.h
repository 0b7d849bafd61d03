// rt_dev_scene.h -- acceleration-structure layout in HBM and the closest-hit traversal (the work the reference
// hands to the DXR driver through TraceRay, raytracing.hlsl:108-112, and
// BuildRaytracingAccelerationStructure, raytracing.cpp:435,547).
#pragma once
#include "rt_dev_math.h"

namespace rtd {

// 64 B: one binary BVH node = the boxes of its two children + two links.  Fetched as 4 x 16 B.
//   link >= 0 : index of an internal node
//   link <  0 : leaf.  BLAS: ~link = (first_tri << 3) | (tri_count - 1), TLAS: ~link = instance index
//   an absent child has the box RT_EMPTY_BOX and is never entered
struct alignas(16) BvhNode {
    float lo0[3], hi0[3], lo1[3], hi1[3];
    int   c0, c1;
    float bound;   // node 0 only: max |coordinate| of the whole tree (box-test padding)
    int   depth;   // node 0 only: depth of the tree (stack budget check on the host)
};
#define RT_EMPTY_COORD 3.0e38f
#define RT_LEAF_MAX_TRIS 4
#define RT_STACK_MARKER ((int) 0x80000000)

// 48 B: triangle in leaf order, fetched as 3 x 16 B
struct alignas(16) TriRec {
    float v0[3], v1[3], v2[3];
    u32 prim_global; // primitive index in the concatenated index buffer of the BLAS (geometry order)
    u32 geometry;    // GeometryIndex()
    u32 prim_local;  // PrimitiveIndex()
};

// 64 B: what traversal needs per instance
struct alignas(16) InstRec {
    float world_to_object[12]; // S7
    const BvhNode* nodes;
    u32   tris_offset;         // byte offset of the BLAS's TriRec array from `nodes` (one allocation per BLAS)
    float bound;               // nodes[0].bound, here so that entering an instance costs no dependent load
};

struct SceneView {
    const BvhNode* tlas_nodes;
    const InstRec* insts;
};

struct HitOut {
    float t, u, v;
    int inst;        // TLAS leaf = index into the rt_build_tlas array, -1 = miss
    int tri;         // TriRec index inside the instance's BLAS
    u32 prim_global;
};

struct TraverseCounters { u32 nodes, tris, insts; u32 node_trips, tri_trips, inst_trips; }; // *_trips: once per wave per step

RT_DEV bool first_active_lane() {
#if defined(__HIP_DEVICE_COMPILE__)
    return __lane_id() == (unsigned) (__ffsll((long long) __ballot(1)) - 1);
#else
    return true;
#endif
}

// parameters of the (non-spec, conservative) ray/box test for one level
//
// Per axis k the ray enters a slab through its plane on the side the ray comes from and leaves through the other one;
// which is which depends only on the sign of d_k, so the two planes are picked with a select and each distance is one
// fused multiply-add,  near_k = P_near * (1/d_k) + cn_k,   far_k = P_far * (1/d_k) + cf_k,
// with cn_k = -(o_k / d_k) - pad_k and cf_k = -(o_k / d_k) + pad_k.  17 vector instructions per box; the form
// (P - o) * (1/d) with min / max per axis and the padding applied afterwards took 27.
//
// pad_k makes the test conservative:
//  * 2^-21 (|o_k| + bound) |1/d_k| covers the rounding of this test itself.  |P| <= bound (the largest coordinate of
//    the level's boxes), so |P - o_k| |1/d_k| <= (bound + |o_k|) |1/d_k|; the reciprocal is good to 2^-23 relative, the
//    product o_k * (1/d_k), the constant and the fused multiply-add to 2^-24 each of a quantity no larger than that:
//    1.25 * 2^-22 in all.
//  * 4e-6 (bound + max|o|) / max|d| covers the rounding of the *triangle* test, whose t is a quotient of sums of terms
//    of size (extent + |o|) / max|d| (S4): two hits closer together than that may order differently than their boxes.
// A zero component of d gives 1/d_k = +-1e18 instead of +-inf: nothing overflows or turns into NaN for coordinates
// below 1e20, the slab still says "inside" or "outside" to within pad_k, and no ray can pass every box (a NaN or
// infinite box distance used to let one ray in ~8 M walk a whole 52 k-instance TLAS and hold a launch for 30 ms).
struct BoxRay {
    float idx, idy, idz;   // 1 / d, magnitude clamped to 1e18
    float cnx, cny, cnz;   // constants of the entry distances
    float cfx, cfy, cfz;   // constants of the exit distances
    bool px, py, pz;       // 1/d_k >= 0: the ray enters slab k through its lower plane
};

#define RT_BOX_ID_LIMIT 1.0e18f
RT_DEV BoxRay make_box_ray(f3 o, f3 d, float bound) {
    BoxRay b;
    b.idx = fminf(fmaxf(fast_rcp(d.x), -RT_BOX_ID_LIMIT), RT_BOX_ID_LIMIT);
    b.idy = fminf(fmaxf(fast_rcp(d.y), -RT_BOX_ID_LIMIT), RT_BOX_ID_LIMIT);
    b.idz = fminf(fmaxf(fast_rcp(d.z), -RT_BOX_ID_LIMIT), RT_BOX_ID_LIMIT);
    b.px = b.idx >= 0.0f; b.py = b.idy >= 0.0f; b.pz = b.idz >= 0.0f;
    const float omax = fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fabsf(o.z));
    const float imin = fminf(fminf(fabsf(b.idx), fabsf(b.idy)), fabsf(b.idz)); // 1 / largest |component of d|
    const float pad = 4.0e-6f * (bound + omax) * imin;
    const float padx = __builtin_fmaf(4.77e-7f * (fabsf(o.x) + bound), fabsf(b.idx), pad);
    const float pady = __builtin_fmaf(4.77e-7f * (fabsf(o.y) + bound), fabsf(b.idy), pad);
    const float padz = __builtin_fmaf(4.77e-7f * (fabsf(o.z) + bound), fabsf(b.idz), pad);
    const float ox = o.x * b.idx, oy = o.y * b.idy, oz = o.z * b.idz;
    b.cnx = -ox - padx; b.cny = -oy - pady; b.cnz = -oz - padz;
    b.cfx = padx - ox;  b.cfy = pady - oy;  b.cfz = padz - oz;
    return b;
}

// hit iff the padded slab intervals and [tmin, tmax] have a common point; `entry` orders the two children of a node
RT_DEV bool box_test(const BoxRay& r, const float* lo, const float* hi, float tmin, float tmax, float& entry) {
    const float nx = __builtin_fmaf(r.px ? lo[0] : hi[0], r.idx, r.cnx), fx = __builtin_fmaf(r.px ? hi[0] : lo[0], r.idx, r.cfx);
    const float ny = __builtin_fmaf(r.py ? lo[1] : hi[1], r.idy, r.cny), fy = __builtin_fmaf(r.py ? hi[1] : lo[1], r.idy, r.cfy);
    const float nz = __builtin_fmaf(r.pz ? lo[2] : hi[2], r.idz, r.cnz), fz = __builtin_fmaf(r.pz ? hi[2] : lo[2], r.idz, r.cfz);
    const float tn = fmaxf(fmaxf(nx, ny), nz);
    const float tf = fminf(fminf(fx, fy), fz);
    entry = tn;
    return fmaxf(tn, tmin) <= fminf(tf, tmax);
}

// 16-byte loads: nodes, triangles and instance records are fetched as whole 16-byte quads (coalescer-friendly,
// 4 / 3 / 4 load instructions per record instead of one per float)
struct alignas(16) Quad { float x, y, z, w; };
RT_DEV Quad ldq(const RT_GLOBAL void* base, int quad_index) { return ((const RT_GLOBAL Quad*) base)[quad_index]; }

#define RT_TRAVERSAL_DONE ((int) 0x80000001)

// ---- closest-hit traversal ------------------------------------------------------------------------------------------
// Closest hit with TMin < t < TMax, back faces culled, ties broken by the lowest (instance, primitive) (S5, S6).
// "while-while": every lane keeps descending inner nodes until it holds a leaf, then the wave processes leaves together.
//
// The per-ray state is a struct so that kernels can set a ray up (trace_begin) and run it (trace_run) separately.
struct TraceState {
    HitOut best;
    bool have_hit, bottom;
    int cur_inst;  // instance being traversed (bottom level)
    int cur;       // node link in hand, RT_TRAVERSAL_DONE when the ray is finished
    float det;     // denominator of best.u / best.v, which hold the numerators until the ray is finished
    const RT_GLOBAL BvhNode* nodes;
    const RT_GLOBAL TriRec* tris;
    BoxRay br;
    RayPrep rp;
};

RT_DEV float tlas_bound_of(const SceneView& sc) { return ((const RT_GLOBAL BvhNode*) sc.tlas_nodes)[0].bound; }

RT_DEV void trace_top_level(const SceneView& sc, f3 o, f3 d, TraceState& S) {
    S.bottom = false;
    S.nodes = (const RT_GLOBAL BvhNode*) sc.tlas_nodes;
    S.br = make_box_ray(o, d, tlas_bound_of(sc));
}

// DXR semantics: inside an instance the ray is in object space and t is preserved
RT_DEV void trace_bottom_level(const SceneView& sc, f3 o, f3 d, TraceState& S) {
    const RT_GLOBAL InstRec* in = (const RT_GLOBAL InstRec*) sc.insts + S.cur_inst;
    Quad m0 = ldq(in, 0), m1 = ldq(in, 1), m2 = ldq(in, 2), m3 = ldq(in, 3);
    float w2o[12] = { m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w, m2.x, m2.y, m2.z, m2.w };
    f3 oo = xform_point43(w2o, o);
    f3 od = xform_dir43(w2o, d);
    S.rp = prepare_ray(oo, od);
    unsigned long long pn = ((unsigned long long) as_u32(m3.y) << 32) | as_u32(m3.x);
    S.nodes = (const RT_GLOBAL BvhNode*) pn; S.tris = (const RT_GLOBAL TriRec*) (pn + as_u32(m3.z));
    S.br = make_box_ray(oo, od, m3.w);
    S.bottom = true;
}

RT_DEV void trace_begin(const SceneView& sc, f3 o, f3 d, float tmax, TraceState& S) {
    S.best.t = tmax; S.best.u = 0.0f; S.best.v = 0.0f; S.best.inst = -1; S.best.tri = -1; S.best.prim_global = 0;
    S.have_hit = false; S.cur_inst = -1; S.tris = nullptr; S.det = 1.0f;
    S.rp.o = o; S.rp.kx = 0; S.rp.ky = 1; S.rp.kz = 2; S.rp.sx = S.rp.sy = S.rp.sz = 0.0f;
    trace_top_level(sc, o, d, S);
    // a NaN in the ray makes every triangle test fail (S4 compares with NaN are false): the answer is a miss, and
    // it must not cost a walk through the whole tree (NaN box tests pass conservatively).  Such rays do occur:
    // a shading normal perpendicular to the ray is scaled by sign(0) = 0 (raytracing.hlsl:64-65).
    // (an infinite component would put NaNs into the box test: every triangle test fails on it as well)
    const float big = 3.0e38f;
    bool finite = fabsf(o.x) <= big && fabsf(o.y) <= big && fabsf(o.z) <= big && fabsf(d.x) <= big && fabsf(d.y) <= big && fabsf(d.z) <= big;
    S.cur = finite ? 0 : RT_TRAVERSAL_DONE;
}

// Runs the ray to the end; S.best is then final (t = +inf for a miss).
// `Stack` provides reset() / top() / put(v, really) / push(v) / pop() and the stack pointer `sp`; COUNT enables the
// traversal counters.  The bottom stack entry is a RT_TRAVERSAL_DONE sentinel, so a pop never has to ask whether the
// stack is empty, and "back to the top level" (RT_STACK_MARKER) and "finished" are handled like leaves.
//
// The node step is straight-line code: the entry a pop would return is read together with the node (it is needed by
// the lanes whose boxes both miss), the far child is stored above the top of the stack whether or not it is kept, and
// the stack pointer moves by (both hit) - (none hit).  A wave pays every path that any of its lanes takes, so the
// branchy form (pop / descend / push as separate exec-masked regions, ~60 scalar instructions per step) cost the sum
// of all of them on most steps.
template <class Stack, bool COUNT>
RT_DEV void trace_run(const SceneView& sc, f3 o, f3 d, float tmin, float tmax, Stack& st, TraceState& S, TraverseCounters& cnt) {
    st.reset();
    for (;;) {
        // ---- inner nodes ---------------------------------------------------------------------------------------
        while (S.cur >= 0) {
            const RT_GLOBAL BvhNode* n = S.nodes + S.cur;
            Quad q0 = ldq(n, 0), q1 = ldq(n, 1), q2 = ldq(n, 2), q3 = ldq(n, 3);
            const int top = st.top();
            if (COUNT) { cnt.nodes++; if (first_active_lane()) cnt.node_trips++; }
            float lo0[3] = { q0.x, q0.y, q0.z }, hi0[3] = { q0.w, q1.x, q1.y };
            float lo1[3] = { q1.z, q1.w, q2.x }, hi1[3] = { q2.y, q2.z, q2.w };
            float e0, e1;
            const bool h0 = box_test(S.br, lo0, hi0, tmin, S.best.t, e0);
            const bool h1 = box_test(S.br, lo1, hi1, tmin, S.best.t, e1);
            const int c0 = (int) as_u32(q3.x), c1 = (int) as_u32(q3.y);
            const bool both = h0 && h1, any = h0 || h1, first0 = e0 <= e1;
            st.put(first0 ? c1 : c0, both);       // the far child, kept only when both boxes are hit
            int next = h0 ? c0 : c1;              // the only child hit ...
            next = both ? (first0 ? c0 : c1) : next; // ... or the nearer of the two
            S.cur = any ? next : top;
            st.sp += (both ? 1 : 0) - (any ? 0 : 1);
        }
        if (S.cur == RT_TRAVERSAL_DONE) break;
        if (S.cur == RT_STACK_MARKER) {
            // ---- the instance is finished: back to the top level ---------------------------------------------------
            trace_top_level(sc, o, d, S);
            S.cur = st.pop();
        } else if (!S.bottom) {
            // ---- TLAS leaf: enter the instance ---------------------------------------------------------------------
            S.cur_inst = ~S.cur;
            if (COUNT) { cnt.insts++; if (first_active_lane()) cnt.inst_trips++; }
            trace_bottom_level(sc, o, d, S);
            st.push(RT_STACK_MARKER);
            S.cur = 0;
        } else {
            // ---- BLAS leaf: up to 8 triangles ------------------------------------------------------------------
            int v = ~S.cur;
            int first = v >> 3, count = (v & 7) + 1;
            for (int k = 0; k < count; k++) {
                const RT_GLOBAL TriRec* tr = S.tris + first + k;
                Quad a = ldq(tr, 0), b = ldq(tr, 1), c = ldq(tr, 2);
                if (COUNT) { cnt.tris++; if (first_active_lane()) cnt.tri_trips++; }
                float t = 0.0f, vn = 0.0f, wn = 0.0f, det = 1.0f;
                const bool hit = intersect_triangle(S.rp, mk3(a.x, a.y, a.z), mk3(a.w, b.x, b.y), mk3(b.z, b.w, c.x), t, vn, wn, det);
                const u32 prim_global = as_u32(c.y);
                // closest hit, ties to the lowest (instance, primitive); written without short-circuit branches
                const bool in_range = (t > tmin) & (t < tmax);
                const bool first_of_tie = (S.cur_inst < S.best.inst) | ((S.cur_inst == S.best.inst) & (prim_global < S.best.prim_global));
                const bool better = (t < S.best.t) | (S.have_hit & (t == S.best.t) & first_of_tie);
                if (hit & in_range & better) {
                    S.best.t = t; S.best.u = vn; S.best.v = wn; S.det = det; S.best.inst = S.cur_inst; S.best.tri = first + k;
                    S.best.prim_global = prim_global; S.have_hit = true;
                }
            }
            S.cur = st.pop();
        }
    }
    if (S.have_hit) { S.best.u = S.best.u / S.det; S.best.v = S.best.v / S.det; } // S4: u = V / det, v = W / det
    else S.best.t = inf_f32();
}

} // namespace rtd
