// rt_dev_scene.h -- acceleration-structure layout in HBM and the closest-hit traversal (the work the reference
// hands to the DXR driver through TraceRay, raytracing.hlsl:108-112, and
// BuildRaytracingAccelerationStructure, raytracing.cpp:435,547).
#pragma once
#include "rt_dev_math.h"

namespace rtd {

// 64 B: one binary BVH node as the builders write it = the boxes of its two children + two links.
//   link >= 0 : index of an internal node (PackedNode: twice the index, i.e. its offset in 16-byte units)
//   link <  0 : leaf.  BLAS: ~link = (first_tri << 3) | (tri_count - 1), TLAS: ~link = instance index
//   an absent child has the box RT_EMPTY_BOX and is never entered
// Traversal reads the 32-byte PackedNode made from it (below).
struct alignas(16) BvhNode {
    float lo0[3], hi0[3], lo1[3], hi1[3];
    int   c0, c1;
    float bound;   // node 0 only: max |coordinate| of the whole tree (box-test padding)
    int   depth;   // node 0 only: depth of the tree (stack budget check on the host)
};
#define RT_EMPTY_COORD 3.0e38f
#define RT_LEAF_MAX_TRIS 4
#define RT_STACK_MARKER ((int) 0x80000000)

// 32 B: the node traversal fetches (2 x 16 B instead of 4: for incoherent rays every 16-byte load of a lane is a
// separate trip through the texture-address unit, which was ~70 % busy with the 64-byte node).  The planes of the two
// child boxes are IEEE half floats in the tree's own frame q = (P - ctr) * inv_h (|q| <= 1 inside the tree's box), rounded
// outward (lower planes down, upper planes up: the box only grows, the box test is conservative anyway).  Child 0 sits in
// the low half of each word and child 1 in the high half, so one select per word picks the entry (or exit) plane of
// both children and v_fma_mix_f32 converts the half while it computes the slab distance.  An absent child has
// +infinity planes.  Measured cost of the 11-bit planes: nodes / triangles / instances per ray +0.1 % (bunny), +2 %
// instance entries (52,743-instance field).
struct alignas(16) PackedNode {
    u32 lox, hix, loy, hiy, loz, hiz;
    int c0, c1;
};
// frame of a tree + what the host needs to know about it (written by the packing kernel)
struct TreeInfo {
    float ctr[3], inv_h[3];
    float bound;   // max |coordinate| of the tree in its own (object / world) space
    int   depth;
};
typedef _Float16 rt_half2 __attribute__((ext_vector_type(2)));
RT_DEV float half_lo(u32 w) { return (float) __builtin_bit_cast(rt_half2, w).x; }
RT_DEV float half_hi(u32 w) { return (float) __builtin_bit_cast(rt_half2, w).y; }

// 48 B: triangle in leaf order, fetched as 3 x 16 B.  A BLAS holds its n triangles three times, records [c n, (c + 1) n)
// with the coordinates of every vertex in the order (c + 1, c + 2, c) mod 3: a ray whose dominant direction axis is kz
// reads copy kz and needs no permutation (RayPrep, rt_dev_math.h).  Everything else is the same in the three copies.
struct alignas(16) TriRec {
    float v0[3], v1[3], v2[3];
    u32 prim_global; // primitive index in the concatenated index buffer of the BLAS (geometry order)
    u32 geometry;    // GeometryIndex()
    u32 prim_local;  // PrimitiveIndex()
};

// 96 B: what traversal needs per instance (6 x 16 B)
struct alignas(16) InstRec {
    float world_to_object[12]; // S7
    const PackedNode* nodes;
    u32   tris_offset;         // byte offset of the BLAS's TriRec array from `nodes` (one allocation per BLAS)
    float bound;               // TreeInfo.bound of the BLAS, here so that entering an instance costs no dependent load
    float ctr[3], inv_h[3];    // frame of the BLAS's packed nodes
    u32   tris_copy_bytes;     // size of one copy of the TriRec array (copy kz starts at tris_offset + kz * tris_copy_bytes)
    float pad;
};

// Flat scenes (a handful of instances, rt_build_tlas): ONE tree over the world-space boxes of all instanced triangles.
// 48 B: a triangle of that tree in leaf order, vertices in its instance's OBJECT space (the triangle test runs there, S7),
// three copies by coordinate order like TriRec.  The tree keeps the instances in separate subtrees (lbvh_flat_build), and
// the triangles of instance 0 come first: a leaf (links as in a BLAS) belongs to instance 1 iff its first triangle is at
// or after SceneView::flat_first1.
struct alignas(16) FlatTri {
    float v0[3], v1[3], v2[3];
    u32 prim_global; // primitive index in the BLAS index buffer: the tie-break key of S5
    u32 tri;         // TriRec index inside the instance's BLAS (what the hit record names)
    u32 inst;        // index into the rt_build_tlas array
};
#define RT_FLAT_MAX_INSTANCES 2
#define RT_FLAT_MAX_TRIS (1u << 20) // triangles of all instances together (leaf links keep 24 bits for the first one)

// Grid top level (rt_api.hip build_tlas_grid): scenes of many instances of similar size spread evenly (the 52,743-instance
// field) get, besides the top-level tree, a uniform grid over the instances' world boxes; cell c lists the instances whose
// box (grown by grid_slack) overlaps it, entries grid_start[c] .. grid_start[c + 1].  32 B, fetched as 2 x 16 B.
struct alignas(16) GridEntry {
    float lo[3]; u32 inst;   // index into the rt_build_tlas array
    float hi[3]; u32 pad;
};
#define RT_GRID_MAX_RES 1024
#define RT_GRID_STACK_ROWS 20    // LDS stack rows of the grid kernels: sentinel + bottom-level depth + the row above the top

#define RT_INLINE_MAX_TRIS 8
struct SceneView {
    const PackedNode* tlas_nodes;
    const InstRec* insts;
    float tlas_ctr[3], tlas_inv_h[3], tlas_bound;
    // Inlined instance (inl_inst < 0: none): one instance of at most RT_INLINE_MAX_TRIS triangles -- a floor with a luminaire,
    // the walls of a room: a few large triangles whose instance box holds the whole scene, so that every ray would enter it
    // (and, its origin being inside, enter it first) -- is not in the top level at all: every ray tests its triangles
    // before the walk, all lanes together, in the instance's object space like any other triangle test (trace_begin).  The
    // walk then starts with that hit's t as its bound, as it did when the instance came first, and the top level holds the
    // other n_insts - 1 instances: its primitive p is instance p, except p == inl_inst, which is the last instance
    // (rt_build_tlas moves the last box into the free slot).
    int inl_inst;
    // the same for a flat scene of two instances of which one is that small (a mesh on a floor): the flat tree then holds the
    // other one alone (instance flat_inst0) and walks like a one-instance scene, 8 waves per SIMD (trace_flat<1, .., INL>)
    int flat_inl_inst;
    u32 flat_inst0;
    u32 inl_count;           // triangles of the inlined instance
    u32 n_insts;
    const TriRec* inl_tris;  // its TriRec array (three copies of inl_copy_bytes)
    u32 inl_copy_bytes;
    float inl_w2o[12];
    float inl_cam_o[3];      // the camera origin in its object space (as flat_cam_o)
    // world boxes of those triangles, grown by 1e-4 of the scene's reach (rt_api.hip set_inline_boxes): a triangle whose box
    // no ray of the wave meets is not tested, and a wave that meets none of them does not prepare its rays for the instance
    float inl_box[RT_INLINE_MAX_TRIS][6];
    u32 inl_box_same;        // bit j: the box of triangle j is the box of triangle j - 1 (the two halves of a quad)
    // flat tree (flat_instances = 0: none, traverse the two levels)
    const PackedNode* flat_nodes;
    const FlatTri* flat_tris;
    u32 flat_copy_bytes;     // size of one copy of the FlatTri array
    u32 flat_instances;
    u32 flat_first1;         // first triangle of instance 1 (the triangle count when there is one instance)
    float flat_ctr[3], flat_inv_h[3], flat_bound;
    float flat_w2o[RT_FLAT_MAX_INSTANCES][12]; // world_to_object of the instances (InstRec), here so that they arrive as scalars
    float flat_cam_o[RT_FLAT_MAX_INSTANCES][3]; // the camera origin in the object space of each instance (xform_point43 done by the host: same operations, same result)
    // grid top level (grid_cells = 0: none, walk tlas_nodes): see GridEntry / trace_run_grid
    const u32* grid_start;       // [cells + 1] first entry of each cell
    const GridEntry* grid_entries;
    u32 grid_cells, grid_n_entries;
    u32 grid_res[3];             // cells per axis, each 1 .. RT_GRID_MAX_RES
    float grid_lo[3], grid_cell[3], grid_inv_cell[3];
    float grid_slack;            // world-space slack of the grid's decisions (the entries' boxes are grown by it)
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
// The test runs in the frame of the tree's packed nodes: q = (P - ctr) * inv_h per axis, an affine map per axis, which
// keeps the ray parameter t.  Per axis k the ray enters a slab through its plane on the side the ray comes from and
// leaves through the other one; which is which depends only on the sign of d_k, so the planes are picked with a select
// (one per packed word, i.e. for both children at once) and each distance is one fused multiply-add,
//     near_k = q_near * (1/d_k) + cn_k,   far_k = q_far * (1/d_k) + cf_k      (o, d in the frame)
// with cn_k = -(o_k / d_k) - pad_k and cf_k = -(o_k / d_k) + pad_k: 14 vector instructions per box; the form
// (P - o) * (1/d) with min / max per axis and the padding applied afterwards took 27.
//
// pad_k makes the test conservative:
//  * 2^-20 (|o_k| + 1) |1/d_k| covers the rounding of this test itself.  |q| <= 1, so |q - o_k| |1/d_k| <=
//    (1 + |o_k|) |1/d_k|; origin and direction in the frame carry 2^-23 and 2^-24 relative, the reciprocal 2^-23, the
//    product o_k * (1/d_k), the constant and the fused multiply-add 2^-24 each of a quantity no larger than that:
//    2^-21 in all.
//  * 4e-6 (bound + max|o|) / max|d| (object / world space quantities) covers the rounding of the *triangle* test, whose
//    t is a quotient of sums of terms of size (extent + |o|) / max|d| (S4): two hits closer together than that may
//    order differently than their boxes.
// A zero component of d gives 1/d_k = +-1e18 instead of +-inf: nothing overflows or turns into NaN, the slab still says
// "inside" or "outside" to within pad_k, and no ray can pass every box (a NaN or infinite box distance used to let one
// ray in ~8 M walk a whole 52 k-instance TLAS and hold a launch for 30 ms).
struct BoxRay {
    float idx, idy, idz;   // 1 / d in the frame, magnitude within [1e-30, 1e18]
    float cnx, cny, cnz;   // constants of the entry distances
    float cfx, cfy, cfz;   // constants of the exit distances
    bool px, py, pz;       // 1/d_k >= 0: the ray enters slab k through its lower plane
};

#define RT_BOX_ID_LIMIT 1.0e18f
RT_DEV float box_rcp(float d) {
    const float r = fast_rcp(d);
    // magnitude clamped to [1e-30, 1e18] (never 0: an absent child's infinite planes must stay infinite), sign kept:
    // one median and one bit-field insert; a NaN (from a NaN direction, rejected in trace_begin) would become 1e-30
    return copysignf(__builtin_amdgcn_fmed3f(fabsf(r), 1.0e-30f, RT_BOX_ID_LIMIT), r);
}
// o, d, bound: the level's own space (world or object); ctr, inv_h: the frame of its packed nodes
RT_DEV BoxRay make_box_ray(f3 o, f3 d, float bound, f3 ctr, f3 inv_h) {
    BoxRay b;
    const float omax = fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fabsf(o.z));
    const float dmax = fmaxf(fmaxf(fabsf(d.x), fabsf(d.y)), fabsf(d.z));
    const float pad = 4.0e-6f * (bound + omax) * fminf(fast_rcp(dmax), RT_BOX_ID_LIMIT);
    const f3 oq = (o - ctr) * inv_h, dq = d * inv_h;
    b.idx = box_rcp(dq.x); b.idy = box_rcp(dq.y); b.idz = box_rcp(dq.z);
    b.px = b.idx >= 0.0f; b.py = b.idy >= 0.0f; b.pz = b.idz >= 0.0f;
    const float padx = __builtin_fmaf(9.54e-7f * (fabsf(oq.x) + 1.0f), fabsf(b.idx), pad);
    const float pady = __builtin_fmaf(9.54e-7f * (fabsf(oq.y) + 1.0f), fabsf(b.idy), pad);
    const float padz = __builtin_fmaf(9.54e-7f * (fabsf(oq.z) + 1.0f), fabsf(b.idz), pad);
    const float ox = oq.x * b.idx, oy = oq.y * b.idy, oz = oq.z * b.idz;
    b.cnx = -ox - padx; b.cny = -oy - pady; b.cnz = -oz - padz;
    b.cfx = padx - ox;  b.cfy = pady - oy;  b.cfz = padz - oz;
    return b;
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
    HitOut best;     // filled in when the ray is finished
    ClosestHit ch;   // the closest hit so far (triangle_candidate, rt_dev_math.h)
    bool bottom;
    int cur_inst;  // instance being traversed (bottom level)
    int cur;       // node link in hand, RT_TRAVERSAL_DONE when the ray is finished
    const RT_GLOBAL PackedNode* nodes;
    const RT_GLOBAL TriRec* tris;
    BoxRay br;
    RayPrep rp;
};

RT_DEV void trace_top_level(const SceneView& sc, f3 o, f3 d, TraceState& S) {
    S.bottom = false;
    S.nodes = (const RT_GLOBAL PackedNode*) sc.tlas_nodes;
    S.br = make_box_ray(o, d, sc.tlas_bound, mk3(sc.tlas_ctr[0], sc.tlas_ctr[1], sc.tlas_ctr[2]),
                        mk3(sc.tlas_inv_h[0], sc.tlas_inv_h[1], sc.tlas_inv_h[2]));
}

// DXR semantics: inside an instance the ray is in object space and t is preserved
RT_DEV void trace_bottom_level(const SceneView& sc, f3 o, f3 d, TraceState& S) {
    const RT_GLOBAL InstRec* in = (const RT_GLOBAL InstRec*) sc.insts + S.cur_inst;
    Quad m0 = ldq(in, 0), m1 = ldq(in, 1), m2 = ldq(in, 2), m3 = ldq(in, 3), m4 = ldq(in, 4), m5 = ldq(in, 5);
    float w2o[12] = { m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w, m2.x, m2.y, m2.z, m2.w };
    f3 oo = xform_point43(w2o, o);
    f3 od = xform_dir43(w2o, d);
    S.rp = prepare_ray(oo, od);
    unsigned long long pn = ((unsigned long long) as_u32(m3.y) << 32) | as_u32(m3.x);
    S.nodes = (const RT_GLOBAL PackedNode*) pn;
    S.tris = (const RT_GLOBAL TriRec*) (pn + as_u32(m3.z) + (unsigned long long) ((u32) S.rp.kz * as_u32(m5.z))); // the triangle copy of this ray's kz
    S.br = make_box_ray(oo, od, m3.w, mk3(m4.x, m4.y, m4.z), mk3(m4.w, m5.x, m5.y));
    S.bottom = true;
}

// back at the top level after an instance: the box-test constants of the world-space ray come back from the stack's
// side store (9 LDS reads) instead of being recomputed (4 reciprocals and ~45 other instructions per instance left)
template <class Stack>
RT_DEV void trace_resume_top_level(const SceneView& sc, const Stack& st, TraceState& S) {
    S.bottom = false;
    S.nodes = (const RT_GLOBAL PackedNode*) sc.tlas_nodes;
    float v[9];
    st.load_side(v);
    BoxRay& b = S.br;
    b.idx = v[0]; b.idy = v[1]; b.idz = v[2]; b.cnx = v[3]; b.cny = v[4]; b.cnz = v[5]; b.cfx = v[6]; b.cfy = v[7]; b.cfz = v[8];
    b.px = b.idx >= 0.0f; b.py = b.idy >= 0.0f; b.pz = b.idz >= 0.0f;
}

RT_DEV void trace_begin(const SceneView& sc, f3 o, f3 d, float tmax, TraceState& S, bool tree_top_level = true) {
    S.ch.t = tmax; S.ch.u_num = 0.0f; S.ch.v_num = 0.0f; S.ch.det = 1.0f; S.ch.inst = -1; S.ch.tri = -1; S.ch.prim_global = 0;
    S.cur_inst = -1; S.tris = nullptr;
    S.rp.o = o; S.rp.kz = 2; S.rp.flip = 0u; S.rp.sx = S.rp.sy = S.rp.sz = 0.0f;
    if (tree_top_level) trace_top_level(sc, o, d, S);
    else { // grid top level (trace_run_grid): no box-test constants for the world-space ray
        S.bottom = false; S.nodes = nullptr;
        S.br.idx = S.br.idy = S.br.idz = 1.0f; S.br.cnx = S.br.cny = S.br.cnz = 0.0f; S.br.cfx = S.br.cfy = S.br.cfz = 0.0f; S.br.px = S.br.py = S.br.pz = true;
    }
    // a NaN in the ray makes every triangle test fail (S4 compares with NaN are false): the answer is a miss, and
    // it must not cost a walk through the whole tree (NaN box tests pass conservatively).  Such rays do occur:
    // a shading normal perpendicular to the ray is scaled by sign(0) = 0 (raytracing.hlsl:64-65).
    // (an infinite component would put NaNs into the box test: every triangle test fails on it as well)
    const float big = 3.0e38f;
    bool finite = fabsf(o.x) <= big && fabsf(o.y) <= big && fabsf(o.z) <= big && fabsf(d.x) <= big && fabsf(d.y) <= big && fabsf(d.z) <= big;
    S.cur = finite ? 0 : RT_TRAVERSAL_DONE;
}

// One step down the tree: fetches node `cur` (>= 0), tests the two child boxes against [tmin, best_t] and returns the link
// to go on with -- the nearer child hit (the other one goes on the stack), or the top of the stack when neither is.
// Straight-line code: the entry a pop would return is read together with the node (it is needed by the lanes whose
// boxes both miss), the far child is stored above the top of the stack whether or not it is kept, and the stack pointer
// moves by (both hit) - (none hit).  A wave pays every path that any of its lanes takes, so the branchy form (pop /
// descend / push as separate exec-masked regions, ~60 scalar instructions per step) cost the sum of all of them.
// BYTE_LINKS: `nodes` is the same for all lanes and inner links are byte offsets (flat tree): the load takes the base as
// a scalar and the link as its 32-bit offset, no address arithmetic at all.
template <class Stack, bool COUNT, bool BYTE_LINKS = false>
RT_DEV int node_step(const RT_GLOBAL PackedNode* nodes, const BoxRay& r, int cur, float tmin, float best_t, Stack& st, TraverseCounters& cnt,
                     bool* hit0 = nullptr, bool* hit1 = nullptr) {
    // otherwise links to inner nodes count 16-byte units (k_pack_nodes): base + (cur << 4) is one v_lshl_add_u64, whose
    // shift goes up to 4; a node index would need a 64-bit shift and a 64-bit add
    const RT_GLOBAL PackedNode* n = (const RT_GLOBAL PackedNode*) ((const RT_GLOBAL char*) nodes + (BYTE_LINKS ? (unsigned long long) (u32) cur : ((unsigned long long) (u32) cur << 4)));
    const Quad q0 = ldq(n, 0), q1 = ldq(n, 1);
    const int top = st.top();
    if (COUNT) { cnt.nodes++; if (first_active_lane()) cnt.node_trips++; }
    // entry / exit planes of both children, one packed word per axis
    const u32 nwx = as_u32(r.px ? q0.x : q0.y), fwx = as_u32(r.px ? q0.y : q0.x);
    const u32 nwy = as_u32(r.py ? q0.z : q0.w), fwy = as_u32(r.py ? q0.w : q0.z);
    const u32 nwz = as_u32(r.pz ? q1.x : q1.y), fwz = as_u32(r.pz ? q1.y : q1.x);
    const float e0 = fmaxf(fmaxf(__builtin_fmaf(half_lo(nwx), r.idx, r.cnx), __builtin_fmaf(half_lo(nwy), r.idy, r.cny)), __builtin_fmaf(half_lo(nwz), r.idz, r.cnz));
    const float x0 = fminf(fminf(__builtin_fmaf(half_lo(fwx), r.idx, r.cfx), __builtin_fmaf(half_lo(fwy), r.idy, r.cfy)), __builtin_fmaf(half_lo(fwz), r.idz, r.cfz));
    const float e1 = fmaxf(fmaxf(__builtin_fmaf(half_hi(nwx), r.idx, r.cnx), __builtin_fmaf(half_hi(nwy), r.idy, r.cny)), __builtin_fmaf(half_hi(nwz), r.idz, r.cnz));
    const float x1 = fminf(fminf(__builtin_fmaf(half_hi(fwx), r.idx, r.cfx), __builtin_fmaf(half_hi(fwy), r.idy, r.cfy)), __builtin_fmaf(half_hi(fwz), r.idz, r.cfz));
    // hit iff the padded slab intervals and [tmin, best t] have a common point
    const bool h0 = fmaxf(e0, tmin) <= fminf(x0, best_t);
    const bool h1 = fmaxf(e1, tmin) <= fminf(x1, best_t);
    if (hit0) { *hit0 = h0; *hit1 = h1; } // which children the ray enters (trace_flat: the root's children are the instances)
    const int c0 = (int) as_u32(q1.z), c1 = (int) as_u32(q1.w);
    const bool both = h0 && h1, any = h0 || h1, first0 = e0 <= e1;
    st.put(first0 ? c1 : c0, both);       // the far child, kept only when both boxes are hit
    int next = h0 ? c0 : c1;              // the only child hit ...
    next = both ? (first0 ? c0 : c1) : next; // ... or the nearer of the two
    st.sp += (both ? 1 : 0) - (any ? 0 : 1);
    return any ? next : top;
}

// The triangles of the inlined instance (SceneView::inl_inst / flat_inl_inst), before the walk, all lanes together.
// The box test is a plain slab test in world space with the hardware reciprocal: the boxes are grown far beyond its
// rounding (1e-4 of the scene's reach against ~1e-7 of the coordinates) and the comparison has a relative slack of 1e-5 on
// top, so a ray that the triangle test would accept is never turned away; which triangles get tested changes no result.
template <bool COUNT, bool CAMERA_ORIGIN = false>
RT_DEV void inlined_triangles_first(const SceneView& sc, f3 o, f3 d, float tmin, int inst, ClosestHit& best, TraverseCounters& cnt) {
    u32 need = 0; // bit j: some ray of the wave meets the box of triangle j (wave-uniform)
    {
        const float ix = fast_rcp(d.x), iy = fast_rcp(d.y), iz = fast_rcp(d.z);
        bool in_box = false;
        for (u32 j = 0; j < sc.inl_count; j++) {
            if (!((sc.inl_box_same >> j) & 1u)) {
                const float* b = sc.inl_box[j];
                const float x0 = (b[0] - o.x) * ix, x1 = (b[3] - o.x) * ix;
                const float y0 = (b[1] - o.y) * iy, y1 = (b[4] - o.y) * iy;
                const float z0 = (b[2] - o.z) * iz, z1 = (b[5] - o.z) * iz;
                const float en = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fmaxf(fminf(z0, z1), tmin));
                const float ex = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fminf(fmaxf(z0, z1), best.t));
                in_box = en <= __builtin_fmaf(1.0e-5f, fabsf(en) + fabsf(ex), ex); // [tmin, closest hit so far] of the ray against the box, with slack
            }
            if (any_lane(in_box)) need |= 1u << j;
        }
    }
    if (need == 0u) return;
    const f3 oo = CAMERA_ORIGIN ? mk3(sc.inl_cam_o[0], sc.inl_cam_o[1], sc.inl_cam_o[2]) : xform_point43(sc.inl_w2o, o);
    const RayPrep irp = prepare_ray<true>(oo, xform_dir43(sc.inl_w2o, d));
    const RT_GLOBAL TriRec* tr = (const RT_GLOBAL TriRec*) ((const RT_GLOBAL char*) sc.inl_tris + (size_t) ((u32) irp.kz * sc.inl_copy_bytes));
    for (u32 j = 0; j < sc.inl_count; j++) {
        if (!((need >> j) & 1u)) continue;
        const Quad a = ldq(tr + j, 0), b = ldq(tr + j, 1), c = ldq(tr + j, 2);
        if (COUNT) { cnt.tris++; if (first_active_lane()) cnt.tri_trips++; }
        triangle_candidate(irp, mk3(a.x, a.y, a.z), mk3(a.w, b.x, b.y), mk3(b.z, b.w, c.x), tmin, inst, (int) j, as_u32(c.y), best);
    }
}

// Runs the ray to the end; S.best is then final (t = +inf for a miss).
// `Stack` provides reset() / top() / put(v, really) / push(v) / pop(), the stack pointer `sp` and a side store of nine
// floats (save_side / load_side) for the top-level box-test constants; COUNT enables the
// traversal counters.  The bottom stack entry is a RT_TRAVERSAL_DONE sentinel, so a pop never has to ask whether the
// stack is empty, and "back to the top level" (RT_STACK_MARKER) and "finished" are handled like leaves.
//
// (node step: node_step above)
template <class Stack, bool COUNT>
RT_DEV void trace_run(const SceneView& sc, f3 o, f3 d, float tmin, float tmax, Stack& st, TraceState& S, TraverseCounters& cnt) {
    st.reset();
    { const BoxRay& b = S.br; const float v[9] = { b.idx, b.idy, b.idz, b.cnx, b.cny, b.cnz, b.cfx, b.cfy, b.cfz }; st.save_side(v); }
    if (sc.inl_inst >= 0 && S.cur != RT_TRAVERSAL_DONE) { // the triangles of the inlined instance first (SceneView::inl_inst); uniform but for NaN rays
        inlined_triangles_first<COUNT>(sc, o, d, tmin, sc.inl_inst, S.ch, cnt);
        if (sc.n_insts < 2u) S.cur = RT_TRAVERSAL_DONE; // nothing else in the scene (cannot happen: such scenes get a flat tree)
    }
    for (;;) {
        // ---- inner nodes ---------------------------------------------------------------------------------------
        while (S.cur >= 0) S.cur = node_step<Stack, COUNT>(S.nodes, S.br, S.cur, tmin, S.ch.t, st, cnt);
        if (S.cur == RT_TRAVERSAL_DONE) break;
        if (S.cur == RT_STACK_MARKER) {
            // ---- the instance is finished: back to the top level ---------------------------------------------------
            trace_resume_top_level(sc, st, S);
            S.cur = st.pop();
        } else if (!S.bottom) {
            // ---- TLAS leaf: enter the instance ---------------------------------------------------------------------
            const int p = ~S.cur;
            S.cur_inst = p == sc.inl_inst ? (int) sc.n_insts - 1 : p; // the slot of the inlined instance holds the last one (SceneView)
            if (COUNT) { cnt.insts++; if (first_active_lane()) cnt.inst_trips++; }
            trace_bottom_level(sc, o, d, S);
            st.push(RT_STACK_MARKER);
            S.cur = 0;
        } else {
            // ---- BLAS leaf: up to 8 triangles ------------------------------------------------------------------
            int v = ~S.cur;
            int first = v >> 3, count = (v & 7) + 1;
            const RT_GLOBAL TriRec* tr = S.tris + first;
            for (int k = 0; k < count; k++, tr++) {
                Quad a = ldq(tr, 0), b = ldq(tr, 1), c = ldq(tr, 2);
                if (COUNT) { cnt.tris++; if (first_active_lane()) cnt.tri_trips++; }
                triangle_candidate(S.rp, mk3(a.x, a.y, a.z), mk3(a.w, b.x, b.y), mk3(b.z, b.w, c.x), tmin, S.cur_inst, first + k, as_u32(c.y), S.ch);
            }
            S.cur = st.pop();
        }
    }
    S.best.inst = S.ch.inst; S.best.tri = S.ch.tri; S.best.prim_global = S.ch.prim_global;
    if (S.ch.inst >= 0) { S.best.t = S.ch.t; S.best.u = S.ch.u_num / S.ch.det; S.best.v = S.ch.v_num / S.ch.det; } // S4: u = V / det, v = W / det
    else { S.best.t = inf_f32(); S.best.u = 0.0f; S.best.v = 0.0f; }
}

// ---- two-level scenes with a grid top level -------------------------------------------------------------------------------
// The same closest hit as trace_run; the top level is walked as a uniform grid instead of a tree.  A ray in a tree over
// 52,743 instances takes 17 of its 29 node steps in the top level -- sixteen of them just to get from the root down to the
// leaf around its own origin, and every ancestor's box has to be tested on the way -- for 1.26 instances entered.  A grid
// finds the cell of a point in O(1): the ray is clipped to the grid's box, starts in the cell of its first point there and
// steps from cell to cell (3D DDA), testing the world boxes of the instances listed in each cell; it stops when the closest
// hit so far lies before the end of the current cell.
//  * An instance overlapping several cells is listed in all of them and entered once: in the cell that holds the point
//    where the ray enters its box (t_entry >= t_in - slack; before that cell the ray is not in the box, in a later cell
//    the instance has been entered already).  With the slack an instance may be entered twice (rarely; the closest hit
//    does not care), never not at all: the boxes of the entries and the cells an instance is listed in are grown by
//    grid_slack, far more than the rounding of the cell walk can misplace a point.
//  * Order and termination are by t along the ray, so S5 (closest hit, ties to the lowest ids) holds as in any other walk:
//    the walk ends only when best t < the current cell's exit t - slack, and an instance holding a hit at best t has its
//    entry point at or before best t, i.e. in a cell already visited.
// The bottom level is walked as in trace_run (same node_step, same triangle loop); the stack holds bottom-level entries only
// and its sentinel is the "back to the top level" marker, so an instance entry pushes nothing.
struct GridWalk {
    u32 cell;              // x | y << 10 | z << 20
    float tnx, tny, tnz;   // t at which the ray leaves the current cell through its next x / y / z plane
    float t_in;            // t at which it entered the current cell
    float t_end;           // t at which it leaves the grid's box
    u32 e, e_end;          // entries of the current cell still to test
    float idx, idy, idz;   // 1 / d, magnitude within [1e-30, 1e18] (box_rcp)
    float slack_t;         // grid_slack as a distance along the ray
};

RT_DEV u32 grid_cell_index(const SceneView& sc, u32 cell) {
    const u32 x = cell & 1023u, y = (cell >> 10) & 1023u, z = cell >> 20;
    const u32 i = (z * sc.grid_res[1] + y) * sc.grid_res[0] + x;
    return i < sc.grid_cells ? i : sc.grid_cells - 1u; // (cannot be out of range; a wrong cell costs time, a wrong address the GPU)
}
RT_DEV void grid_load_cell(const SceneView& sc, GridWalk& g) {
    const RT_GLOBAL u32* st = (const RT_GLOBAL u32*) sc.grid_start + grid_cell_index(sc, g.cell);
    const u32 a = st[0], b = st[1];
    g.e = a < sc.grid_n_entries ? a : sc.grid_n_entries;
    g.e_end = b < sc.grid_n_entries ? b : sc.grid_n_entries;
}
// returns false when the ray misses the grid's box within [tmin, best_t]
RT_DEV bool grid_begin(const SceneView& sc, f3 o, f3 d, float tmin, float best_t, GridWalk& g) {
    g.idx = box_rcp(d.x); g.idy = box_rcp(d.y); g.idz = box_rcp(d.z);
    const float hx = sc.grid_lo[0] + (float) sc.grid_res[0] * sc.grid_cell[0], hy = sc.grid_lo[1] + (float) sc.grid_res[1] * sc.grid_cell[1], hz = sc.grid_lo[2] + (float) sc.grid_res[2] * sc.grid_cell[2];
    const float ax = (sc.grid_lo[0] - o.x) * g.idx, bx = (hx - o.x) * g.idx;
    const float ay = (sc.grid_lo[1] - o.y) * g.idy, by = (hy - o.y) * g.idy;
    const float az = (sc.grid_lo[2] - o.z) * g.idz, bz = (hz - o.z) * g.idz;
    const float tn = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
    const float tf = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
    const float idmin = fminf(fminf(fabsf(g.idx), fabsf(g.idy)), fabsf(g.idz));
    g.slack_t = sc.grid_slack * idmin;
    const float t0 = fmaxf(tn, tmin);
    g.t_end = tf + g.slack_t;
    g.t_in = t0;
    g.e = 0; g.e_end = 0;
    if (!(t0 <= fminf(g.t_end, best_t))) return false; // (NaN: false)
    // the cell of the ray's first point in the box, clamped into the grid
    const f3 p = mk3(__builtin_fmaf(d.x, t0, o.x), __builtin_fmaf(d.y, t0, o.y), __builtin_fmaf(d.z, t0, o.z));
    int c[3]; float tnext[3];
    const float pk[3] = { p.x, p.y, p.z }, ok[3] = { o.x, o.y, o.z }, idk[3] = { g.idx, g.idy, g.idz };
    for (int k = 0; k < 3; k++) {
        const float q = (pk[k] - sc.grid_lo[k]) * sc.grid_inv_cell[k];
        const float qc = fminf(fmaxf(q, 0.0f), (float) (sc.grid_res[k] - 1u)); // (NaN -> 0)
        c[k] = (int) qc;
        const float plane = sc.grid_lo[k] + (float) (c[k] + (idk[k] >= 0.0f ? 1 : 0)) * sc.grid_cell[k];
        tnext[k] = (plane - ok[k]) * idk[k];
    }
    g.cell = (u32) c[0] | ((u32) c[1] << 10) | ((u32) c[2] << 20);
    g.tnx = tnext[0]; g.tny = tnext[1]; g.tnz = tnext[2];
    grid_load_cell(sc, g);
    return true;
}
// Leaves the current cell; false when the walk is over (the closest hit lies before the cell's end, or the ray leaves the grid).
RT_DEV bool grid_next_cell(const SceneView& sc, float best_t, GridWalk& g) {
    const float t_out = fminf(fminf(g.tnx, g.tny), g.tnz);
    if (!(t_out - g.slack_t <= best_t) || !(t_out <= g.t_end)) return false;
    // the axis whose plane comes first (ties: x before y before z; the cells a tie skips are a slack away and listed alike)
    const bool sx = g.tnx <= g.tny && g.tnx <= g.tnz, sy = !sx && g.tny <= g.tnz;
    const float id = sx ? g.idx : sy ? g.idy : g.idz;
    const int shift = sx ? 0 : sy ? 10 : 20;
    const u32 res = sx ? sc.grid_res[0] : sy ? sc.grid_res[1] : sc.grid_res[2];
    const float cs = sx ? sc.grid_cell[0] : sy ? sc.grid_cell[1] : sc.grid_cell[2];
    const int ck = (int) ((g.cell >> shift) & 1023u) + (id >= 0.0f ? 1 : -1);
    if (ck < 0 || ck >= (int) res) return false; // out of the grid
    g.cell = (g.cell & ~(1023u << shift)) | ((u32) ck << shift);
    const float dt = cs * fabsf(id);
    if (sx) g.tnx += dt; else if (sy) g.tny += dt; else g.tnz += dt;
    g.t_in = t_out;
    grid_load_cell(sc, g);
    return true;
}
// Tests entry g.e of the current cell (g.e < g.e_end) and moves past it; true: the ray has to enter instance `inst` now.
RT_DEV bool grid_test_entry(const SceneView& sc, f3 o, float tmin, float best_t, GridWalk& g, int& inst) {
    const RT_GLOBAL GridEntry* en = (const RT_GLOBAL GridEntry*) sc.grid_entries + g.e;
    const Quad q0 = ldq(en, 0), q1 = ldq(en, 1);
    g.e++;
    const float ax = (q0.x - o.x) * g.idx, bx = (q1.x - o.x) * g.idx;
    const float ay = (q0.y - o.y) * g.idy, by = (q1.y - o.y) * g.idy;
    const float az = (q0.z - o.z) * g.idz, bz = (q1.z - o.z) * g.idz;
    const float te = fmaxf(fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz)), tmin);
    const float tx = fminf(fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz)), best_t);
    inst = (int) as_u32(q0.w);
    // the box is met within [tmin, closest hit so far] (it is grown by grid_slack: the comparison itself needs no more than
    // its relative slack), and this is the cell in which the ray enters it
    return te <= __builtin_fmaf(1.0e-5f, fabsf(te) + fabsf(tx), tx) && te >= g.t_in - g.slack_t;
}

template <class Stack, bool COUNT>
RT_DEV void trace_run_grid(const SceneView& sc, f3 o, f3 d, float tmin, float tmax, Stack& st, TraceState& S, TraverseCounters& cnt) {
    st.lds[0] = RT_STACK_MARKER; st.sp = 0; // the sentinel of the bottom-level stack: "back to the top level"
    if (sc.inl_inst >= 0 && S.cur != RT_TRAVERSAL_DONE) inlined_triangles_first<COUNT>(sc, o, d, tmin, sc.inl_inst, S.ch, cnt);
    GridWalk g;
    bool walking = S.cur != RT_TRAVERSAL_DONE && grid_begin(sc, o, d, tmin, S.ch.t, g);
    S.cur = walking ? RT_STACK_MARKER : RT_TRAVERSAL_DONE;
    S.bottom = false;
    for (;;) {
        // ---- bottom-level inner nodes ------------------------------------------------------------------------------
        while (S.cur >= 0) S.cur = node_step<Stack, COUNT>(S.nodes, S.br, S.cur, tmin, S.ch.t, st, cnt);
        if (S.cur == RT_TRAVERSAL_DONE) break;
        if (S.cur == RT_STACK_MARKER) {
            // ---- top level: the next instance to enter, or the end of the walk -----------------------------------------
            int inst = -1;
            for (;;) {
                if (g.e >= g.e_end) { // (a cell may be empty: the lane steps again on the next trip)
                    if (COUNT) { cnt.nodes++; if (first_active_lane()) cnt.node_trips++; } // a cell step counts as a node step
                    if (!grid_next_cell(sc, S.ch.t, g)) { S.cur = RT_TRAVERSAL_DONE; break; }
                }
                if (g.e < g.e_end) {
                    if (COUNT) { cnt.nodes++; if (first_active_lane()) cnt.node_trips++; } // and so does a box test
                    if (grid_test_entry(sc, o, tmin, S.ch.t, g, inst)) break;
                }
            }
            if (S.cur == RT_TRAVERSAL_DONE) break;
            // ---- enter the instance ----------------------------------------------------------------------------------------
            S.cur_inst = (u32) inst < sc.n_insts ? inst : 0;
            if (COUNT) { cnt.insts++; if (first_active_lane()) cnt.inst_trips++; }
            trace_bottom_level(sc, o, d, S);
            S.cur = 0; st.sp = 0; // the stack is the marker alone again (the pop that returned it left sp at -1)
        } else {
            // ---- BLAS leaf: up to 8 triangles ------------------------------------------------------------------
            int v = ~S.cur;
            int first = v >> 3, count = (v & 7) + 1;
            const RT_GLOBAL TriRec* tr = S.tris + first;
            for (int k = 0; k < count; k++, tr++) {
                Quad a = ldq(tr, 0), b = ldq(tr, 1), c = ldq(tr, 2);
                if (COUNT) { cnt.tris++; if (first_active_lane()) cnt.tri_trips++; }
                triangle_candidate(S.rp, mk3(a.x, a.y, a.z), mk3(a.w, b.x, b.y), mk3(b.z, b.w, c.x), tmin, S.cur_inst, first + k, as_u32(c.y), S.ch);
            }
            S.cur = st.pop();
        }
    }
    S.best.inst = S.ch.inst; S.best.tri = S.ch.tri; S.best.prim_global = S.ch.prim_global;
    if (S.ch.inst >= 0) { S.best.t = S.ch.t; S.best.u = S.ch.u_num / S.ch.det; S.best.v = S.ch.v_num / S.ch.det; } // S4: u = V / det, v = W / det
    else { S.best.t = inf_f32(); S.best.u = 0.0f; S.best.v = 0.0f; }
}

// ---- flat scenes: one tree, no instance entry --------------------------------------------------------------------------
// The same answer as trace_run (S4-S7: every triangle is tested in its instance's object space with the ray taken there
// by world_to_object), found in ONE world-space tree over all instanced triangles (rt_build_tlas builds it when the scene
// has at most RT_FLAT_MAX_INSTANCES instances: every BASELINE configuration but the instance field).  What the two-level
// walk pays per instance ENTRY -- transform, ray shear with three divisions, box-test constants, the way back to the top
// level: ~200 vector instructions, executed by the whole wave whenever one lane enters -- is paid here once per instance
// and ray, by all lanes together, before the walk: the RayPrep of each of the K instances stays in registers and a leaf
// picks the one its tag names.  The box test runs in world space all the way down (one BoxRay), the node base is the
// same for all lanes, and the walk has no level changes (no marker, no side store).
template <int K>
struct FlatPreps { RayPrep rp[K]; };
template <int K>
RT_DEV RayPrep flat_pick(const FlatPreps<K>& p, u32 inst) {
    if (K == 1) return p.rp[0];
    const bool one = inst != 0u;
    const RayPrep &a = p.rp[0], &b = p.rp[K - 1];
    RayPrep r;
    r.o = mk3(one ? b.o.x : a.o.x, one ? b.o.y : a.o.y, one ? b.o.z : a.o.z);
    r.kz = one ? b.kz : a.kz; r.flip = one ? b.flip : a.flip;
    r.sx = one ? b.sx : a.sx; r.sy = one ? b.sy : a.sy; r.sz = one ? b.sz : a.sz;
    return r;
}
template <int K, class Stack, bool COUNT, bool SPECULATE, bool CAMERA_ORIGIN = false, bool INL = false>
RT_DEV void trace_flat(const SceneView& sc, f3 o, f3 d, float tmin, float tmax, Stack& st, HitOut& out, TraverseCounters& cnt) {
    static_assert(K >= 1 && K <= RT_FLAT_MAX_INSTANCES && RT_FLAT_MAX_INSTANCES == 2, "flat_pick selects between two preparations");
    ClosestHit best; best.t = tmax; best.u_num = 0.0f; best.v_num = 0.0f; best.det = 1.0f; best.inst = -1; best.tri = -1; best.prim_global = 0;
    const float big = 3.0e38f; // NaN or infinite rays miss without a walk (trace_begin)
    const bool finite = fabsf(o.x) <= big && fabsf(o.y) <= big && fabsf(o.z) <= big && fabsf(d.x) <= big && fabsf(d.y) <= big && fabsf(d.z) <= big;
    const BoxRay br = make_box_ray(o, d, sc.flat_bound, mk3(sc.flat_ctr[0], sc.flat_ctr[1], sc.flat_ctr[2]), mk3(sc.flat_inv_h[0], sc.flat_inv_h[1], sc.flat_inv_h[2]));
    st.reset();
    int cur = finite ? 0 : RT_TRAVERSAL_DONE;
    const RT_GLOBAL PackedNode* nodes = (const RT_GLOBAL PackedNode*) sc.flat_nodes;
    // Two instances: node 0 joins their subtrees (lbvh_flat_build: child i = instance i), so the first step of the walk, taken
    // here, says which instances a ray can reach at all, and the wave prepares the ray (transform, shear with three divisions:
    // ~95 vector instructions each) only for the instances that one of its rays enters -- a wave of sky or floor pixels skips
    // one or both.  The step is the one the loop would have taken first: same stack, same order, same counters.
    static_assert(!INL || K == 1, "the inlined instance goes with a tree of one instance");
    if (INL && cur != RT_TRAVERSAL_DONE) inlined_triangles_first<COUNT, CAMERA_ORIGIN>(sc, o, d, tmin, sc.flat_inl_inst, best, cnt); // (SceneView::flat_inl_inst)
    bool enters[2] = { true, true };
    if (K > 1) {
        enters[0] = false; enters[1] = false;
        if (cur >= 0) cur = node_step<Stack, COUNT, true>(nodes, br, cur, tmin, best.t, st, cnt, &enters[0], &enters[1]);
    } else if (INL) { // the root's step first here too: a wave none of whose rays reaches the mesh's box (or gets past the hit above) prepares nothing
        bool e0 = false, e1 = false;
        if (cur >= 0) cur = node_step<Stack, COUNT, true>(nodes, br, cur, tmin, best.t, st, cnt, &e0, &e1);
        enters[0] = e0 || e1;
    }
    FlatPreps<K> preps;
    for (int i = 0; i < K; i++) {
        RayPrep& rp = preps.rp[i];
        rp.o = mk3(0.0f, 0.0f, 0.0f); rp.kz = 0; rp.flip = 0u; rp.sx = 0.0f; rp.sy = 0.0f; rp.sz = 0.0f;
        if ((K > 1 || INL) && !any_lane(enters[i])) continue; // wave-uniform
        // the matrices are kernel arguments (scalar loads); CAMERA_ORIGIN: o is the camera's for every lane and the host has transformed it
        const f3 oo = CAMERA_ORIGIN ? mk3(sc.flat_cam_o[i][0], sc.flat_cam_o[i][1], sc.flat_cam_o[i][2]) : xform_point43(sc.flat_w2o[i], o);
        rp = prepare_ray<true>(oo, xform_dir43(sc.flat_w2o[i], d));
#if defined(__HIP_DEVICE_COMPILE__)
        // computed here, once per ray: left alone the compiler keeps d[kz] instead of 1 / d[kz] and divides again at every leaf
        asm volatile("" : "+v"(rp.sx), "+v"(rp.sy), "+v"(rp.sz));
#endif
    }
    int pending = RT_TRAVERSAL_DONE; // a leaf put aside (RT_TRAVERSAL_DONE: none)
    for (;;) {
        int leaf;
        if (SPECULATE) {
            // Speculative traversal (Aila & Laine 2009), for incoherent rays: the wave stays in the node loop while some lane has
            // no leaf to test yet; a lane that has one put aside walks on towards its next leaf meanwhile instead of idling
            // (node trips per wave of C2 15.3 -> 12.5, of C4 32.6 -> 25.7; a leaf put aside has not shortened the ray yet:
            // nodes per ray 6.73 -> 6.83).  Coherent rays reach their leaves together and only pay the vote: PRIMARY launches
            // walk without it.
            for (;;) {
                const bool none_aside = pending == RT_TRAVERSAL_DONE, walking = cur != RT_TRAVERSAL_DONE;
                const bool need = none_aside && walking;
                if (!any_lane_both(none_aside, walking)) break;
                if (cur >= 0) cur = node_step<Stack, COUNT, true>(nodes, br, cur, tmin, best.t, st, cnt);
                else if (need) { pending = cur; cur = st.pop(); }
            }
            if (pending == RT_TRAVERSAL_DONE) break; // nothing put aside: the walk has ended
            leaf = pending; pending = RT_TRAVERSAL_DONE;
        } else {
            while (cur >= 0) cur = node_step<Stack, COUNT, true>(nodes, br, cur, tmin, best.t, st, cnt);
            if (cur == RT_TRAVERSAL_DONE) break;
            leaf = cur;
        }
        // ---- leaf: up to 8 triangles of one instance ---------------------------------------------------------------
        const u32 v = (u32) ~leaf;
        const u32 first = v >> 3, count = (v & 7u) + 1u;
        const u32 inst = K > 1 ? (first >= sc.flat_first1 ? 1u : 0u) : (INL ? sc.flat_inst0 : 0u);
        const RayPrep rp = flat_pick<K>(preps, K > 1 ? inst : 0u);
        // the three copies are < 2^32 bytes together: the triangle's address is the array's (scalar) base + a 32-bit offset, like the node's
        static_assert((unsigned long long) RT_FLAT_MAX_TRIS * sizeof(FlatTri) * 3ull < (1ull << 32), "32-bit triangle offsets");
        const u32 off0 = __umul24(first, (u32) sizeof(FlatTri)) + (u32) rp.kz * sc.flat_copy_bytes; // first < 2^20: a 24-bit multiply-add
        for (u32 k = 0; k < count; k++) {
            const RT_GLOBAL FlatTri* tr = (const RT_GLOBAL FlatTri*) ((const RT_GLOBAL char*) sc.flat_tris + (unsigned long long) (off0 + k * (u32) sizeof(FlatTri)));
            const Quad a = ldq(tr, 0), b = ldq(tr, 1), c = ldq(tr, 2);
            if (COUNT) { cnt.tris++; if (first_active_lane()) cnt.tri_trips++; }
            triangle_candidate(rp, mk3(a.x, a.y, a.z), mk3(a.w, b.x, b.y), mk3(b.z, b.w, c.x), tmin, (int) inst, (int) as_u32(c.z), as_u32(c.y), best);
        }
        if (!SPECULATE) cur = st.pop();
    }
    out.inst = best.inst; out.tri = best.tri; out.prim_global = best.prim_global;
    if (best.inst >= 0) { out.t = best.t; out.u = best.u_num / best.det; out.v = best.v_num / best.det; } // S4: u = V / det, v = W / det
    else { out.t = inf_f32(); out.u = 0.0f; out.v = 0.0f; }
}

} // namespace rtd
