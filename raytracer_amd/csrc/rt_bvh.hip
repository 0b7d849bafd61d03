// rt_bvh.hip -- GPU LBVH build for the bottom level (triangles) and top level (instances).
// Replaces the DXR driver builds issued at raytracing.cpp:411-436 (BLAS) and raytracing.cpp:523-548 (TLAS).
//
// Pipeline: primitive boxes -> scene bounds -> 30-bit Morton codes -> stable radix sort -> Karras 2012
// hierarchy -> bottom-up refit with arrival counters -> 64-byte two-child traversal nodes with leaves of up to
// RT_LEAF_MAX_TRIS triangles.
#include "rt_internal.h"

namespace rti {

using namespace rtd;

static constexpr int BT = 256;
static inline dim3 grid_for(u32 n, int bt = BT) { return dim3((n + bt - 1) / bt); }

__global__ void k_triangle_boxes(const RtVertex* __restrict__ verts, const uint16_t* __restrict__ indices, u32 n,
                                 Aabb* __restrict__ boxes) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Aabb b;
    for (int k = 0; k < 3; k++) { b.lo[k] = 3.4e38f; b.hi[k] = -3.4e38f; }
    for (int c = 0; c < 3; c++) {
        const float* p = verts[indices[3 * i + c]].position;
        for (int k = 0; k < 3; k++) { b.lo[k] = fminf(b.lo[k], p[k]); b.hi[k] = fmaxf(b.hi[k], p[k]); }
    }
    boxes[i] = b;
}

// world box of an instance from its vertices (one wave per instance): for a rotated mesh this is markedly tighter than
// the box of the eight transformed corners of the object-space box, so neighbouring instances overlap less in the TLAS
__global__ __launch_bounds__(64) void k_instance_boxes_exact(const InstShade* __restrict__ shade, u32 n, Aabb* __restrict__ boxes) {
    const u32 i = blockIdx.x;
    if (i >= n) return;
    const InstShade& in = shade[i];
    float lo[3] = { 3.4e38f, 3.4e38f, 3.4e38f }, hi[3] = { -3.4e38f, -3.4e38f, -3.4e38f };
    for (u32 j = threadIdx.x; j < in.vertex_count; j += 64u) {
        const float* p = in.vertices[j].position;
        f3 w = xform_point44(in.object_to_world, mk3(p[0], p[1], p[2]));
        float wv[3] = { w.x, w.y, w.z };
        for (int k = 0; k < 3; k++) { lo[k] = fminf(lo[k], wv[k]); hi[k] = fmaxf(hi[k], wv[k]); }
    }
    for (int off = 32; off >= 1; off >>= 1)
        for (int k = 0; k < 3; k++) { lo[k] = fminf(lo[k], __shfl_xor(lo[k], off)); hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], off)); }
    if (threadIdx.x == 0) {
        float amax = 0.0f;
        for (int k = 0; k < 3; k++) amax = fmaxf(amax, fmaxf(fabsf(lo[k]), fabsf(hi[k])));
        // the world-space box test and the object-space triangle test round differently: pad generously
        float pad = amax * 1.0e-5f + 1.0e-30f;
        Aabb b;
        for (int k = 0; k < 3; k++) { b.lo[k] = lo[k] - pad; b.hi[k] = hi[k] + pad; }
        boxes[i] = b;
    }
}

__global__ void k_instance_boxes(const InstShade* __restrict__ shade, const InstRec* __restrict__ insts, u32 n,
                                 Aabb* __restrict__ boxes) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // object-space box of the whole BLAS, back from the frame of its packed nodes (a shade generous)
    float lo[3], hi[3];
    for (int k = 0; k < 3; k++) {
        float h = 1.00001f / insts[i].inv_h[k];
        lo[k] = insts[i].ctr[k] - h; hi[k] = insts[i].ctr[k] + h;
    }
    Aabb b;
    for (int k = 0; k < 3; k++) { b.lo[k] = 3.4e38f; b.hi[k] = -3.4e38f; }
    float amax = 0.0f;
    for (int c = 0; c < 8; c++) {
        f3 p = mk3((c & 1) ? hi[0] : lo[0], (c & 2) ? hi[1] : lo[1], (c & 4) ? hi[2] : lo[2]);
        f3 w = xform_point44(shade[i].object_to_world, p);
        float wv[3] = { w.x, w.y, w.z };
        for (int k = 0; k < 3; k++) { b.lo[k] = fminf(b.lo[k], wv[k]); b.hi[k] = fmaxf(b.hi[k], wv[k]); amax = fmaxf(amax, fabsf(wv[k])); }
    }
    // the world-space box test and the object-space triangle test round differently: pad generously
    float pad = amax * 1.0e-5f + 1.0e-30f;
    for (int k = 0; k < 3; k++) { b.lo[k] -= pad; b.hi[k] += pad; }
    boxes[i] = b;
}

// one workgroup: union of all boxes
__global__ __launch_bounds__(1024) void k_scene_box(const Aabb* __restrict__ boxes, u32 n, Aabb* __restrict__ out) {
    __shared__ float slo[3][1024], shi[3][1024];
    float lo[3] = { 3.4e38f, 3.4e38f, 3.4e38f }, hi[3] = { -3.4e38f, -3.4e38f, -3.4e38f };
    for (u32 i = threadIdx.x; i < n; i += 1024)
        for (int k = 0; k < 3; k++) { lo[k] = fminf(lo[k], boxes[i].lo[k]); hi[k] = fmaxf(hi[k], boxes[i].hi[k]); }
    for (int k = 0; k < 3; k++) { slo[k][threadIdx.x] = lo[k]; shi[k][threadIdx.x] = hi[k]; }
    __syncthreads();
    for (u32 off = 512; off >= 1; off >>= 1) {
        if (threadIdx.x < off)
            for (int k = 0; k < 3; k++) {
                slo[k][threadIdx.x] = fminf(slo[k][threadIdx.x], slo[k][threadIdx.x + off]);
                shi[k][threadIdx.x] = fmaxf(shi[k][threadIdx.x], shi[k][threadIdx.x + off]);
            }
        __syncthreads();
    }
    if (threadIdx.x == 0) for (int k = 0; k < 3; k++) { out->lo[k] = slo[k][0]; out->hi[k] = shi[k][0]; }
}

__global__ void k_morton(const Aabb* __restrict__ boxes, const Aabb* __restrict__ scene, u32 n, u32* __restrict__ keys,
                         u32* __restrict__ vals) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    keys[i] = morton30(boxes[i], *scene);
    vals[i] = i;
}

__global__ void k_hierarchy(const u32* __restrict__ keys, int n, BuildNode* __restrict__ bn, int* __restrict__ leaf_parent) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    int first, last, split;
    lbvh_node_range(keys, n, i, first, last, split);
    int left = (split == first) ? ~split : split;
    int right = (split + 1 == last) ? ~(split + 1) : split + 1;
    bn[i].left = left; bn[i].right = right; bn[i].first = first; bn[i].last = last;
    if (left < 0) leaf_parent[~left] = i; else bn[left].parent = i;
    if (right < 0) leaf_parent[~right] = i; else bn[right].parent = i;
    if (i == 0) bn[0].parent = -1;
}

RT_DEV float load_agent(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// bottom-up: the second thread to arrive at a node merges its children's boxes and continues upwards
RT_DEV float half_area(const Aabb& b) {
    float ex = b.hi[0] - b.lo[0], ey = b.hi[1] - b.lo[1], ez = b.hi[2] - b.lo[2];
    return ex * ey + ey * ez + ez * ex;
}
// surface-area-heuristic constants: one two-child node fetch+test vs one triangle fetch+test on gfx950
#define RT_SAH_NODE 1.0f

// cost[] holds the un-normalised SAH cost of the best cut of each subtree; collapsed[] marks subtrees that are
// cheaper as one leaf of <= leaf_max primitives
__global__ void k_refit(const Aabb* __restrict__ boxes, const u32* __restrict__ perm, int n, const BuildNode* __restrict__ bn,
                        const int* __restrict__ leaf_parent, Aabb* node_boxes, int* depth, u32* flags, float* cost,
                        u32* collapsed, int leaf_max, float sah_tri) {
    int leaf = blockIdx.x * blockDim.x + threadIdx.x;
    if (leaf >= n) return;
    int node = leaf_parent[leaf];
    while (node >= 0) {
        __threadfence(); // release what this thread wrote for the level below
        u32 old = atomicAdd(&flags[node], 1u);
        if (old == 0) return;
        __threadfence(); // acquire the sibling's writes
        Aabb b; int dep = 0; float child_cost = 0.0f;
        for (int k = 0; k < 3; k++) { b.lo[k] = 3.4e38f; b.hi[k] = -3.4e38f; }
        int child[2] = { bn[node].left, bn[node].right };
        for (int c = 0; c < 2; c++) {
            if (child[c] < 0) {
                const Aabb& cb = boxes[perm[~child[c]]];
                for (int k = 0; k < 3; k++) { b.lo[k] = fminf(b.lo[k], cb.lo[k]); b.hi[k] = fmaxf(b.hi[k], cb.hi[k]); }
                child_cost += sah_tri * half_area(cb);
            } else {
                child_cost += load_agent(&cost[child[c]]);
                const Aabb* cb = &node_boxes[child[c]];
                for (int k = 0; k < 3; k++) { b.lo[k] = fminf(b.lo[k], load_agent(&cb->lo[k])); b.hi[k] = fmaxf(b.hi[k], load_agent(&cb->hi[k])); }
                int cd = __hip_atomic_load(&depth[child[c]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                dep = cd > dep ? cd : dep;
            }
        }
        node_boxes[node] = b;
        depth[node] = dep + 1;
        int size = bn[node].last - bn[node].first + 1;
        float a = half_area(b);
        float split_cost = RT_SAH_NODE * a + child_cost, leaf_cost = sah_tri * (float) size * a;
        bool col = size <= leaf_max && leaf_cost <= split_cost && size < n; // the root stays a node
        collapsed[node] = col ? 1u : 0u;
        cost[node] = col ? leaf_cost : split_cost;
        node = bn[node].parent;
    }
}

RT_DEV Aabb child_box(int child, const Aabb* boxes, const u32* perm, const Aabb* node_boxes) {
    return child < 0 ? boxes[perm[~child]] : node_boxes[child];
}

__global__ void k_emit(const Aabb* __restrict__ boxes, const u32* __restrict__ perm, int n, const BuildNode* __restrict__ bn,
                       const Aabb* __restrict__ node_boxes, const int* __restrict__ depth, const Aabb* __restrict__ scene,
                       const u32* __restrict__ collapsed, int tlas, BvhNode* __restrict__ out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    BvhNode nd;
    Aabb b0 = child_box(bn[i].left, boxes, perm, node_boxes), b1 = child_box(bn[i].right, boxes, perm, node_boxes);
    set_box(nd.lo0, nd.hi0, b0);
    set_box(nd.lo1, nd.hi1, b1);
    nd.c0 = lbvh_child_link(bn[i].left, bn, perm, collapsed, tlas != 0);
    nd.c1 = lbvh_child_link(bn[i].right, bn, perm, collapsed, tlas != 0);
    nd.bound = 0.0f; nd.depth = 0;
    if (i == 0) {
        float m = 0.0f;
        for (int k = 0; k < 3; k++) m = fmaxf(m, fmaxf(fabsf(scene->lo[k]), fabsf(scene->hi[k])));
        nd.bound = m; nd.depth = depth[0];
    }
    out[i] = nd;
}

// n == 1: a root whose first child is the only leaf
__global__ void k_emit_single(const Aabb* __restrict__ scene, const u32* __restrict__ perm, int n, int tlas, BvhNode* __restrict__ out) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    BvhNode nd;
    set_box(nd.lo0, nd.hi0, *scene);
    set_empty_box(nd.lo1, nd.hi1);
    nd.c0 = tlas ? ~((int) perm[0]) : ~((0 << 3) | (n - 1));
    nd.c1 = nd.c0; // never entered: its box is empty
    float m = 0.0f;
    for (int k = 0; k < 3; k++) m = fmaxf(m, fmaxf(fabsf(scene->lo[k]), fabsf(scene->hi[k])));
    nd.bound = m; nd.depth = 1;
    out[0] = nd;
}

__global__ void k_trirecs(const u32* __restrict__ perm, const RtVertex* __restrict__ verts, const uint16_t* __restrict__ indices,
                          const u32* __restrict__ tri_geometry, const GeomRec* __restrict__ geoms, u32 n, TriRec* __restrict__ out) {
    u32 j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= 3u * n) return;
    const u32 copy = j / n, i = j - copy * n; // copy c: coordinates in the order (c + 1, c + 2, c) mod 3 (TriRec)
    u32 prim = perm[i];
    TriRec t;
    const float* p0 = verts[indices[3 * prim + 0]].position;
    const float* p1 = verts[indices[3 * prim + 1]].position;
    const float* p2 = verts[indices[3 * prim + 2]].position;
    for (int k = 0; k < 3; k++) { const int s = (int) ((copy + 1u + (u32) k) % 3u); t.v0[k] = p0[s]; t.v1[k] = p1[s]; t.v2[k] = p2[s]; }
    t.prim_global = prim;
    t.geometry = tri_geometry[prim];
    t.prim_local = prim - geoms[t.geometry].first_tri;
    out[j] = t;
}

// ---- flat tree (rt_dev_scene.h: FlatTri, trace_flat): one tree over the world boxes of all instanced triangles ------------
// world boxes of the triangles of instance `inst` alone -> boxes[0 .. n_tris of that instance)
__global__ void k_flat_boxes(FlatSources src, u32 inst, Aabb* __restrict__ boxes) {
    const FlatSource& in = src.s[inst];
    u32 prim = blockIdx.x * blockDim.x + threadIdx.x;
    if (prim >= in.n_tris) return;
    Aabb b;
    for (int k = 0; k < 3; k++) { b.lo[k] = 3.4e38f; b.hi[k] = -3.4e38f; }
    float amax = 0.0f;
    for (int c = 0; c < 3; c++) {
        const float* p = in.verts[in.indices[3 * prim + c]].position;
        f3 w = xform_point44(in.o2w, mk3(p[0], p[1], p[2]));
        float wv[3] = { w.x, w.y, w.z };
        for (int k = 0; k < 3; k++) { b.lo[k] = fminf(b.lo[k], wv[k]); b.hi[k] = fmaxf(b.hi[k], wv[k]); amax = fmaxf(amax, fabsf(wv[k])); }
    }
    // the world-space box test and the object-space triangle test round differently (world_to_object is the rounded inverse
    // of this matrix: a few 2^-24 of the coordinates): pad generously, like the instance boxes of the top level
    const float pad = amax * 1.0e-5f + 1.0e-30f;
    for (int k = 0; k < 3; k++) { b.lo[k] -= pad; b.hi[k] += pad; }
    boxes[prim] = b;
}
// the tree of one instance (nodes[0 .. n_nodes), leaves and perm numbering its own triangles from 0) takes its place in the
// joint arrays: nodes from `base`, triangles from `first`
__global__ void k_flat_join_part(const BvhNode* __restrict__ nodes, u32 n_nodes, u32 base, u32 first, u32 n_tris, const Aabb* __restrict__ scene,
                                 BvhNode* __restrict__ join, Aabb* __restrict__ sub_box, u32* __restrict__ perm) {
    u32 j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n_nodes) {
        BvhNode nd = nodes[j]; // (slots of collapsed subtrees hold garbage; nothing links to them)
        int* link[2] = { &nd.c0, &nd.c1 };
        for (int c = 0; c < 2; c++) {
            if (*link[c] >= 0) *link[c] += (int) base;
            else { const u32 v = (u32) ~*link[c]; *link[c] = (int) ~((((v >> 3) + first) << 3) | (v & 7u)); }
        }
        join[base + j] = nd;
    }
    if (j < n_tris) perm[j] += first;
    if (j == 0) *sub_box = *scene;
}
// two instances: node 0 holds the boxes of the two trees; sub_box[2] = the box of everything (the frame of the packed nodes)
__global__ void k_flat_join_top(BvhNode* __restrict__ join, Aabb* __restrict__ sub_box, u32 base0, u32 base1) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    BvhNode nd;
    set_box(nd.lo0, nd.hi0, sub_box[0]);
    set_box(nd.lo1, nd.hi1, sub_box[1]);
    nd.c0 = (int) base0; nd.c1 = (int) base1;
    nd.bound = fmaxf(join[base0].bound, join[base1].bound);
    nd.depth = 1 + (join[base0].depth > join[base1].depth ? join[base0].depth : join[base1].depth);
    join[0] = nd;
    Aabb u;
    for (int k = 0; k < 3; k++) { u.lo[k] = fminf(sub_box[0].lo[k], sub_box[1].lo[k]); u.hi[k] = fmaxf(sub_box[0].hi[k], sub_box[1].hi[k]); }
    sub_box[2] = u;
}
__global__ void k_flat_tris(FlatSources src, const u32* __restrict__ perm, u32 n, FlatTri* __restrict__ out) {
    u32 j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= 3u * n) return;
    const u32 copy = j / n, i = j - copy * n; // copy c: coordinates in the order (c + 1, c + 2, c) mod 3 (TriRec)
    const u32 g = perm[i];
    u32 inst = 0;
    for (u32 k = 1; k < src.n; k++) if (g >= src.s[k].first) inst = k;
    const FlatSource& in = src.s[inst];
    const u32 prim = g - in.first;
    FlatTri t;
    const float* p0 = in.verts[in.indices[3 * prim + 0]].position;
    const float* p1 = in.verts[in.indices[3 * prim + 1]].position;
    const float* p2 = in.verts[in.indices[3 * prim + 2]].position;
    for (int k = 0; k < 3; k++) { const int s = (int) ((copy + 1u + (u32) k) % 3u); t.v0[k] = p0[s]; t.v1[k] = p1[s]; t.v2[k] = p2[s]; }
    t.prim_global = prim;
    t.tri = in.inv_perm[prim];
    t.inst = inst;
    out[j] = t;
}
__global__ void k_invert_perm(const u32* __restrict__ perm, u32 n, u32* __restrict__ inv) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) inv[perm[i]] = i;
}
int lbvh_invert_perm(hipStream_t s, const u32* perm, u32 n, u32* inv) {
    if (n == 0) return RT_OK;
    hipLaunchKernelGGL(k_invert_perm, grid_for(n), dim3(BT), 0, s, perm, n, inv);
    RT_HIP(hipGetLastError());
    return RT_OK;
}

static int lbvh_build_impl(hipStream_t s, LbvhWorkspace& w, PlocWorkspace* pw, SahWorkspace* sw, int builder, float sah_tri, u32 n, int leaf_max, bool tlas, BvhNode* nodes_out, u32* perm_out);
__global__ void k_pack_nodes(const Aabb* __restrict__ scene, const BvhNode* __restrict__ nodes, u32 n_nodes, PackedNode* __restrict__ out, TreeInfo* __restrict__ info, int link_scale);

// One tree per instance over the world boxes of its triangles, with the same builders as the two levels, joined under one
// top node (two instances) and packed in the frame of the box of everything.  A tree over all triangles together came out
// worse (bunny on a floor: 7.4 instead of 5.9 node steps per ray): the few scene-sized triangles of a floor or a room sit
// in the middle of the small ones' centroids, where no single split plane isolates them, and every level until they are
// alone costs a scene-sized box that all rays enter.  Keeping the instances apart also keeps every leaf in one instance.
int lbvh_flat_build(hipStream_t s, LbvhWorkspace& w, PlocWorkspace* pw, SahWorkspace* sw, int builder, float sah_tri, const FlatSources& src,
                    BvhNode* join, Aabb* sub_boxes, u32* perm, FlatTri* tris, PackedNode* nodes_out) {
    if (src.n < 1 || src.n > RT_FLAT_MAX_INSTANCES) { set_error("lbvh_flat_build: bad instance count %u", src.n); return RT_ERR_INVALID; }
    u32 base[RT_FLAT_MAX_INSTANCES], total_nodes = src.n > 1 ? 1u : 0u, total_tris = 0;
    for (u32 i = 0; i < src.n; i++) {
        const u32 n = src.s[i].n_tris, nn = n > 1 ? n - 1 : 1;
        base[i] = total_nodes; total_nodes += nn; total_tris += n;
        hipLaunchKernelGGL(k_flat_boxes, grid_for(n), dim3(BT), 0, s, src, i, w.boxes);
        int rc = lbvh_build_impl(s, w, pw, sw, builder, sah_tri, n, RT_LEAF_MAX_TRIS, false, w.tree_nodes, perm + src.s[i].first);
        if (rc != RT_OK) return rc;
        hipLaunchKernelGGL(k_flat_join_part, grid_for(n > nn ? n : nn), dim3(BT), 0, s, (const BvhNode*) w.tree_nodes, nn, base[i], src.s[i].first, n,
                           (const Aabb*) w.scene_box, join, sub_boxes + i, perm + src.s[i].first);
    }
    if (src.n > 1) hipLaunchKernelGGL(k_flat_join_top, dim3(1), dim3(64), 0, s, join, sub_boxes, base[0], base[1]);
    hipLaunchKernelGGL(k_pack_nodes, grid_for(total_nodes), dim3(BT), 0, s, (const Aabb*) (sub_boxes + (src.n > 1 ? 2 : 0)), (const BvhNode*) join, total_nodes, nodes_out, w.tree_info, 32); // inner links are byte offsets (trace_flat: one base for all lanes)
    hipLaunchKernelGGL(k_flat_tris, grid_for(3 * (size_t) total_tris), dim3(BT), 0, s, src, (const u32*) perm, total_tris, tris);
    RT_HIP(hipGetLastError());
    return RT_OK;
}

int lbvh_workspace_reserve(LbvhWorkspace& w, u32 n) {
    if (n <= w.capacity) return RT_OK;
    lbvh_workspace_free(w);
    u32 cap = n < 1024 ? 1024 : n;
    RT_HIP(hipMalloc(&w.boxes, sizeof(Aabb) * cap));
    RT_HIP(hipMalloc(&w.node_boxes, sizeof(Aabb) * cap));
    RT_HIP(hipMalloc(&w.scene_box, sizeof(Aabb)));
    RT_HIP(hipMalloc(&w.keys, 4 * (size_t) cap));
    RT_HIP(hipMalloc(&w.vals, 4 * (size_t) cap));
    RT_HIP(hipMalloc(&w.keys_tmp, 4 * (size_t) cap));
    RT_HIP(hipMalloc(&w.vals_tmp, 4 * (size_t) cap));
    RT_HIP(hipMalloc(&w.build_nodes, sizeof(BuildNode) * cap));
    RT_HIP(hipMalloc(&w.leaf_parent, 4 * (size_t) cap));
    RT_HIP(hipMalloc(&w.flags, 4 * (size_t) cap));
    RT_HIP(hipMalloc(&w.depth, 4 * (size_t) cap));
    RT_HIP(hipMalloc(&w.cost, 4 * (size_t) cap));
    RT_HIP(hipMalloc(&w.collapsed, 4 * (size_t) cap));
    RT_HIP(hipMalloc(&w.sort_scratch, sort_scratch_bytes(cap)));
    RT_HIP(hipMalloc(&w.tree_nodes, sizeof(BvhNode) * cap));
    RT_HIP(hipMalloc(&w.tree_info, sizeof(TreeInfo)));
    w.capacity = cap;
    return RT_OK;
}

void lbvh_workspace_free(LbvhWorkspace& w) {
    void* ptrs[] = { w.boxes, w.node_boxes, w.scene_box, w.keys, w.vals, w.keys_tmp, w.vals_tmp, w.build_nodes,
                     w.leaf_parent, w.flags, w.depth, w.cost, w.collapsed, w.sort_scratch, w.tree_nodes, w.tree_info };
    for (void* p : ptrs) if (p) (void) hipFree(p);
    w = LbvhWorkspace();
}

int lbvh_triangle_boxes(hipStream_t s, LbvhWorkspace& w, const RtVertex* verts, const uint16_t* indices, u32 n_tris) {
    if (n_tris == 0) return RT_OK;
    hipLaunchKernelGGL(k_triangle_boxes, grid_for(n_tris), dim3(BT), 0, s, verts, indices, n_tris, w.boxes);
    RT_HIP(hipGetLastError());
    return RT_OK;
}

__global__ void k_move_box(Aabb* __restrict__ boxes, u32 dst, u32 src) {
    if (blockIdx.x == 0 && threadIdx.x == 0) boxes[dst] = boxes[src];
}
int lbvh_move_box(hipStream_t s, LbvhWorkspace& w, u32 dst, u32 src) {
    hipLaunchKernelGGL(k_move_box, dim3(1), dim3(64), 0, s, w.boxes, dst, src);
    RT_HIP(hipGetLastError());
    return RT_OK;
}
int lbvh_instance_boxes(hipStream_t s, LbvhWorkspace& w, const InstShade* shade, const InstRec* insts, u32 n, bool exact) {
    if (n == 0) return RT_OK;
    if (exact) hipLaunchKernelGGL(k_instance_boxes_exact, dim3(n), dim3(64), 0, s, shade, n, w.boxes);
    else hipLaunchKernelGGL(k_instance_boxes, grid_for(n), dim3(BT), 0, s, shade, insts, n, w.boxes);
    RT_HIP(hipGetLastError());
    return RT_OK;
}

// ---- packing: 64-byte build nodes -> 32-byte traversal nodes (PackedNode, rt_dev_scene.h) ---------------------------
// frame of a tree: centre and reciprocal half extent of its box.  A flat or very thin axis is scaled as if it were
// 1/4096 of the widest one (its planes then sit near q = 0, where halves are densest), so that rays keep ordinary
// magnitudes in the frame.
RT_DEV void tree_frame(const Aabb& box, float* ctr, float* inv_h) {
    float h[3], hmax = 0.0f;
    for (int k = 0; k < 3; k++) {
        ctr[k] = 0.5f * box.lo[k] + 0.5f * box.hi[k];
        h[k] = fmaxf(box.hi[k] - ctr[k], ctr[k] - box.lo[k]);
        hmax = fmaxf(hmax, h[k]);
    }
    for (int k = 0; k < 3; k++) inv_h[k] = 1.0f / fmaxf(fmaxf(h[k], hmax * (1.0f / 4096.0f)), 1.0e-30f);
}
// largest half <= x / smallest half >= x
RT_DEV u32 half_floor(float x) {
    _Float16 h = (_Float16) x;
    unsigned short u = __builtin_bit_cast(unsigned short, h);
    if ((float) h > x) u = (u & 0x7fffu) == 0 ? (unsigned short) 0x8001u : ((u & 0x8000u) ? u + 1 : u - 1);
    return u;
}
RT_DEV u32 half_ceil(float x) {
    _Float16 h = (_Float16) x;
    unsigned short u = __builtin_bit_cast(unsigned short, h);
    if ((float) h < x) u = (u & 0x7fffu) == 0 ? (unsigned short) 0x0001u : ((u & 0x8000u) ? u - 1 : u + 1);
    return u;
}
// planes of one child box on axis k in the frame, rounded outward; the float computation of q is itself good to a few
// 2^-24 of (|q| + 1), far below the half spacing, and is pushed outward by 2^-20 (|q| + 1) before the rounding
RT_DEV void pack_planes(float lo, float hi, float ctr, float inv_h, u32& qlo, u32& qhi) {
    if (lo == RT_EMPTY_COORD) { qlo = 0x7c00u; qhi = 0x7c00u; return; } // absent child: +infinity
    float a = (lo - ctr) * inv_h, b = (hi - ctr) * inv_h;
    a -= 9.54e-7f * (fabsf(a) + 1.0f); b += 9.54e-7f * (fabsf(b) + 1.0f);
    qlo = half_floor(a); qhi = half_ceil(b);
}
__global__ void k_pack_nodes(const Aabb* __restrict__ scene, const BvhNode* __restrict__ nodes, u32 n_nodes,
                             PackedNode* __restrict__ out, TreeInfo* __restrict__ info, int link_scale) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes) return;
    float ctr[3], inv_h[3];
    tree_frame(*scene, ctr, inv_h);
    const BvhNode nd = nodes[i];
    u32 w[6];
    for (int k = 0; k < 3; k++) {
        u32 l0, h0, l1, h1;
        pack_planes(nd.lo0[k], nd.hi0[k], ctr[k], inv_h[k], l0, h0);
        pack_planes(nd.lo1[k], nd.hi1[k], ctr[k], inv_h[k], l1, h1);
        w[2 * k] = l0 | (l1 << 16); w[2 * k + 1] = h0 | (h1 << 16);
    }
    PackedNode pn;
    pn.lox = w[0]; pn.hix = w[1]; pn.loy = w[2]; pn.hiy = w[3]; pn.loz = w[4]; pn.hiz = w[5];
    // inner links: link_scale = 2: 16-byte units (trace_run: base + (link << 4) is one instruction), 32: byte offsets (trace_flat)
    pn.c0 = nd.c0 >= 0 ? link_scale * nd.c0 : nd.c0; pn.c1 = nd.c1 >= 0 ? link_scale * nd.c1 : nd.c1;
    out[i] = pn;
    if (i == 0) {
        TreeInfo t;
        for (int k = 0; k < 3; k++) { t.ctr[k] = ctr[k]; t.inv_h[k] = inv_h[k]; }
        t.bound = nd.bound; t.depth = nd.depth;
        *info = t;
    }
}

static int lbvh_build_impl(hipStream_t s, LbvhWorkspace& w, PlocWorkspace* pw, SahWorkspace* sw, int builder, float sah_tri, u32 n, int leaf_max, bool tlas, BvhNode* nodes_out, u32* perm_out);
int lbvh_build(hipStream_t s, LbvhWorkspace& w, PlocWorkspace* pw, SahWorkspace* sw, int builder, float sah_tri, u32 n, int leaf_max, bool tlas, PackedNode* nodes_out, u32* perm_out) {
    int rc = lbvh_build_impl(s, w, pw, sw, builder, sah_tri, n, leaf_max, tlas, w.tree_nodes, perm_out);
    if (rc != RT_OK) return rc;
    const u32 n_nodes = n > 1 ? n - 1 : 1;
    hipLaunchKernelGGL(k_pack_nodes, grid_for(n_nodes), dim3(BT), 0, s, w.scene_box, (const BvhNode*) w.tree_nodes, n_nodes, nodes_out, w.tree_info, 2);
    RT_HIP(hipGetLastError());
    return RT_OK;
}
static int lbvh_build_impl(hipStream_t s, LbvhWorkspace& w, PlocWorkspace* pw, SahWorkspace* sw, int builder, float sah_tri, u32 n, int leaf_max, bool tlas, BvhNode* nodes_out, u32* perm_out) {
    if (n == 0 || n > w.capacity) { set_error("lbvh_build: bad primitive count %u", n); return RT_ERR_INVALID; }
    hipLaunchKernelGGL(k_scene_box, dim3(1), dim3(1024), 0, s, w.boxes, n, w.scene_box);
    hipLaunchKernelGGL(k_morton, grid_for(n), dim3(BT), 0, s, w.boxes, w.scene_box, n, w.keys, w.vals);
    int rc = radix_sort_pairs(s, w.keys, w.vals, w.keys_tmp, w.vals_tmp, w.sort_scratch, n, 30);
    if (rc != RT_OK) return rc;
    RT_HIP(hipMemcpyAsync(perm_out, w.vals, (size_t) n * 4, hipMemcpyDeviceToDevice, s));
    if (n == 1) {
        hipLaunchKernelGGL(k_emit_single, dim3(1), dim3(64), 0, s, w.scene_box, perm_out, (int) n, tlas ? 1 : 0, nodes_out);
        RT_HIP(hipGetLastError());
        return RT_OK;
    }
    if ((builder == 2 || (builder == 3 && !tlas)) && sw && n <= SAH_BUILD_MAX) return sah_build(s, w, *sw, n, leaf_max, sah_tri, tlas, nodes_out, perm_out);
    if (builder == 4 && sw && pw && n <= SAH_BUILD_MAX) {
        // neither build wins everywhere (SAH: large meshes and instance fields; PLOC: a handful of big triangles), and
        // both are cheap next to the frames traced through them: build both, keep the cheaper tree
        rc = ploc_build(s, w, *pw, n, leaf_max, sah_tri, tlas, nodes_out, perm_out); if (rc != RT_OK) return rc;
        rc = sah_workspace_reserve(*sw, n); if (rc != RT_OK) return rc;
        rc = sah_build(s, w, *sw, n, leaf_max, sah_tri, tlas, sw->alt_nodes, sw->alt_perm); if (rc != RT_OK) return rc;
        float cost_ploc = 0.0f, cost_sah = 0.0f;
        RT_HIP(hipMemcpyAsync(&cost_ploc, w.cost + (n - 2), 4, hipMemcpyDeviceToHost, s)); // the root is PLOC's last node
        RT_HIP(hipMemcpyAsync(&cost_sah, sw->counters + 3, 4, hipMemcpyDeviceToHost, s));
        RT_HIP(hipStreamSynchronize(s));
        if (cost_sah < cost_ploc) {
            RT_HIP(hipMemcpyAsync(nodes_out, sw->alt_nodes, sizeof(BvhNode) * (size_t) (n - 1), hipMemcpyDeviceToDevice, s));
            RT_HIP(hipMemcpyAsync(perm_out, sw->alt_perm, 4 * (size_t) n, hipMemcpyDeviceToDevice, s));
        }
        return RT_OK;
    }
    if (builder >= 1 && pw) return ploc_build(s, w, *pw, n, leaf_max, sah_tri, tlas, nodes_out, perm_out);
    RT_HIP(hipMemsetAsync(w.flags, 0, (size_t) n * 4, s));
    RT_HIP(hipMemsetAsync(w.depth, 0, (size_t) n * 4, s));
    hipLaunchKernelGGL(k_hierarchy, grid_for(n - 1), dim3(BT), 0, s, w.keys, (int) n, w.build_nodes, w.leaf_parent);
    hipLaunchKernelGGL(k_refit, grid_for(n), dim3(BT), 0, s, w.boxes, perm_out, (int) n, w.build_nodes, w.leaf_parent,
                       w.node_boxes, w.depth, w.flags, w.cost, w.collapsed, leaf_max, sah_tri);
    hipLaunchKernelGGL(k_emit, grid_for(n - 1), dim3(BT), 0, s, w.boxes, perm_out, (int) n, w.build_nodes, w.node_boxes,
                       w.depth, w.scene_box, w.collapsed, tlas ? 1 : 0, nodes_out);
    RT_HIP(hipGetLastError());
    return RT_OK;
}

int lbvh_write_trirecs(hipStream_t s, const u32* perm, const RtVertex* verts, const uint16_t* indices,
                       const u32* tri_geometry, const GeomRec* geoms, u32 n_tris, TriRec* out) {
    if (n_tris == 0) return RT_OK;
    hipLaunchKernelGGL(k_trirecs, grid_for(3 * (size_t) n_tris), dim3(BT), 0, s, perm, verts, indices, tri_geometry, geoms, n_tris, out);
    RT_HIP(hipGetLastError());
    return RT_OK;
}

} // namespace rti
