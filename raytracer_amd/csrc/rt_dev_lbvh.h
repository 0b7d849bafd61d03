// rt_dev_lbvh.h -- per-element pieces of the GPU LBVH build (Morton codes, Karras 2012 hierarchy, node emission).
// The reference delegates this to the DXR driver (BuildRaytracingAccelerationStructure, raytracing.cpp:435,547).
#pragma once
#include "rt_dev_scene.h"

namespace rtd {

struct Aabb { float lo[3], hi[3]; };

RT_DEV u32 expand_bits10(u32 v) { // 10 bits -> every third bit
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}

// 30-bit Morton code of the box centre relative to the scene bounds
RT_DEV u32 morton30(const Aabb& b, const Aabb& scene) {
    u32 q[3];
    for (int k = 0; k < 3; k++) {
        float c = 0.5f * (b.lo[k] + b.hi[k]);
        float e = scene.hi[k] - scene.lo[k];
        float x = e > 0.0f ? (c - scene.lo[k]) / e : 0.0f;
        x = x * 1024.0f;
        int xi = (int) x;
        q[k] = (u32) (xi < 0 ? 0 : (xi > 1023 ? 1023 : xi));
    }
    return (expand_bits10(q[0]) << 2) | (expand_bits10(q[1]) << 1) | expand_bits10(q[2]);
}

RT_DEV int clz32(u32 x) {
#if defined(__HIPCC__)
    return x ? __clz((int) x) : 32;
#else
    return x ? __builtin_clz(x) : 32;
#endif
}

// common-prefix length of sorted keys i and j (ties broken by the index, so all keys are distinct)
RT_DEV int lbvh_delta(const u32* keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    u32 a = keys[i], b = keys[j];
    if (a != b) return clz32(a ^ b);
    return 32 + clz32((u32) i ^ (u32) j);
}

// Karras 2012: internal node i (0 <= i < n-1) covers the sorted range [first, last] and splits after `split`
RT_DEV void lbvh_node_range(const u32* keys, int n, int i, int& first, int& last, int& split) {
    int d = (lbvh_delta(keys, n, i, i + 1) - lbvh_delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
    int dmin = lbvh_delta(keys, n, i, i - d);
    int lmax = 2;
    while (lbvh_delta(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (lbvh_delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    int j = i + l * d;
    int dnode = lbvh_delta(keys, n, i, j);
    int s = 0;
    int t = l;
    do {
        t = (t + 1) >> 1;
        if (lbvh_delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    split = i + s * d + (d < 0 ? -1 : 0);
    first = i < j ? i : j;
    last = i < j ? j : i;
}

// Build-time node: children are internal nodes (>= 0) or sorted leaves (~leaf index)
struct BuildNode {
    int left, right;
    int first, last;
    int parent;
};

RT_DEV void aabb_union(Aabb& a, const Aabb& b) {
    for (int k = 0; k < 3; k++) { a.lo[k] = fminf(a.lo[k], b.lo[k]); a.hi[k] = fmaxf(a.hi[k], b.hi[k]); }
}

// link of one child for the traversal node: an internal child whose subtree was marked `collapsed` (cheaper as one
// leaf by the surface-area heuristic, at most 8 primitives) becomes a leaf over its contiguous sorted range.
// tlas: leaf link = ~(original instance index); TLAS subtrees are never collapsed (leaf_max 1).
RT_DEV int lbvh_child_link(int child, const BuildNode* bn, const u32* perm, const u32* collapsed, bool tlas) {
    int first, count;
    if (child < 0) { first = ~child; count = 1; }
    else {
        if (!collapsed[child]) return child;
        first = bn[child].first; count = bn[child].last - bn[child].first + 1;
    }
    if (tlas) return ~((int) perm[first]);
    return ~((first << 3) | (count - 1));
}

RT_DEV void set_box(float* lo, float* hi, const Aabb& b) {
    for (int k = 0; k < 3; k++) { lo[k] = b.lo[k]; hi[k] = b.hi[k]; }
}
RT_DEV void set_empty_box(float* lo, float* hi) {
    for (int k = 0; k < 3; k++) { lo[k] = RT_EMPTY_COORD; hi[k] = RT_EMPTY_COORD; }
}

} // namespace rtd
