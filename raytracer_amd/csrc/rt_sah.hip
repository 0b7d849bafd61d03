// rt_sah.hip -- BVH topology by top-down binned surface-area splits (builder 2) on gfx950.
//
// The tree is built breadth first, one level per step.  A step has three kernels:
//   k_sah_bin    one thread per primitive: adds its box to one of 16 bins per axis of the node that owns it;
//   k_sah_decide one thread per node of the level: sweeps its bins for the cheapest split, decides between a leaf and
//                an inner node;
//   k_sah_number one workgroup: numbers the new inner nodes and their children with a prefix scan (the numbering is
//                breadth first and the same from run to run) and writes the 64-byte build nodes;
//   k_sah_move   one thread per primitive: moves it to its child's side of the node's range.
// Against the agglomerative PLOC build (rt_ploc.hip) this is the slower build (a few launches per level) and the
// better tree: on the reference's bunny it cut nodes visited per ray by 10 % and traversal time by 12 % (DESIGN.md
// section 7).  Bottom-level trees are small (a BLAS has at most 65,535 vertices, raytracing.cpp:367), so the per-level
// launch latency, not the work, is what the build costs: about a millisecond.
#include "rt_internal.h"

namespace rti {

using namespace rtd;

static constexpr int BT = 256;
static constexpr int NBINS = 16;
static constexpr u32 NO_TASK = 0xffffffffu;
static inline dim3 grid_for(u32 n, int bt = BT) { return dim3((n + bt - 1) / bt); }

// order-preserving map float -> u32, so that integer atomicMin / atomicMax order floats
RT_DEV u32 f2key(float f) { u32 b = as_u32(f); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); }
RT_DEV float key2f(u32 k) { u32 b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k; return __uint_as_float(b); }

struct SahBin { u32 n; u32 lo[3]; u32 hi[3]; u32 pad; };          // 32 B
struct SahTask {                                                   // a node of the current level
    u32 first, count;      // its range of the primitive order
    int parent_slot;       // 2 * parent node + child number, -1 for the root
    u32 depth;
    u32 cb_lo[3], cb_hi[3]; // bounds of the primitives' box centres, as keys (filled by k_sah_move of the level above)
    float lo[3], hi[3];     // the node's box
};
struct SahDecision {       // what k_sah_split decided for a task
    int axis;              // -2 leaf, -1 split in the middle of the range, 0..2 split on that axis
    int split_bin;         // primitives with bin < split_bin go left
    u32 n_left;
    u32 child_task;        // index of the left child task in the next level (right = +1)
    u32 cnt_left, cnt_right; // cursors of k_sah_move
};

RT_DEV int sah_bin_of(float c, float lo, float hi) {
    float e = hi - lo;
    int b = (int) ((c - lo) / e * (float) NBINS);
    return b < 0 ? 0 : (b > NBINS - 1 ? NBINS - 1 : b);
}

__global__ void k_sah_clear(SahBin* __restrict__ bins, const u32* __restrict__ counters) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= counters[0] * 3u * NBINS) return; // the bins of this level's tasks
    SahBin b; b.n = 0; b.pad = 0;
    for (int k = 0; k < 3; k++) { b.lo[k] = 0xffffffffu; b.hi[k] = 0u; }
    bins[i] = b;
}

RT_DEV u32 wave_min_u32(u32 v) { for (int off = 32; off >= 1; off >>= 1) { u32 o = __shfl_xor(v, off); v = o < v ? o : v; } return v; }
RT_DEV u32 wave_max_u32(u32 v) { for (int off = 32; off >= 1; off >>= 1) { u32 o = __shfl_xor(v, off); v = o > v ? o : v; } return v; }

__global__ void k_sah_init(const Aabb* __restrict__ boxes, u32 n, u32* __restrict__ idx, u32* __restrict__ slot_task, SahTask* __restrict__ task0,
                           u32* __restrict__ counters) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < n;
    if (live) { idx[i] = i; slot_task[i] = 0u; }
    const Aabb b = live ? boxes[i] : Aabb();
    for (int k = 0; k < 3; k++) { // one atomic per wave: every primitive of the tree targets the same six words
        const u32 key = f2key(0.5f * (b.lo[k] + b.hi[k]));
        const u32 mn = wave_min_u32(live ? key : 0xffffffffu), mx = wave_max_u32(live ? key : 0u);
        if ((threadIdx.x & 63u) == 0 && mn <= mx) { atomicMin(&task0->cb_lo[k], mn); atomicMax(&task0->cb_hi[k], mx); }
    }
    if (!live) return;
    if (i == 0) { counters[0] = 1u; /* tasks of this level */ counters[1] = 0u; /* inner nodes so far */ counters[2] = 0u; /* depth */ }
}

__global__ void k_sah_bin(const Aabb* __restrict__ boxes, const u32* __restrict__ idx, const u32* __restrict__ slot_task, u32 n,
                          const SahTask* __restrict__ tasks, SahBin* __restrict__ bins) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u32 t = slot_task[i];
    if (t == NO_TASK) return;
    const SahTask& T = tasks[t];
    if (T.count < 2u) return;
    const Aabb b = boxes[idx[i]];
    for (int axis = 0; axis < 3; axis++) {
        const float lo = key2f(T.cb_lo[axis]), hi = key2f(T.cb_hi[axis]);
        if (!(hi > lo)) continue;
        SahBin* bin = bins + ((size_t) t * 3 + axis) * NBINS + sah_bin_of(0.5f * (b.lo[axis] + b.hi[axis]), lo, hi);
        atomicAdd(&bin->n, 1u);
        for (int k = 0; k < 3; k++) { atomicMin(&bin->lo[k], f2key(b.lo[k])); atomicMax(&bin->hi[k], f2key(b.hi[k])); }
    }
}

// the first levels (at most 16 nodes, so all primitives of the tree aim at <= 768 bins): bins are accumulated per
// workgroup in LDS and flushed with one global atomic per touched word; with direct global atomics the binning of the
// root of a 52,743-instance TLAS took 4.4 ms
static constexpr u32 BIN_LDS_TASKS = 16;
__global__ __launch_bounds__(BT) void k_sah_bin_lds(const Aabb* __restrict__ boxes, const u32* __restrict__ idx, const u32* __restrict__ slot_task, u32 n,
                                                    const SahTask* __restrict__ tasks, SahBin* __restrict__ bins, const u32* __restrict__ counters) {
    __shared__ u32 s_n[BIN_LDS_TASKS * 3 * NBINS], s_lo[BIN_LDS_TASKS * 3 * NBINS][3], s_hi[BIN_LDS_TASKS * 3 * NBINS][3];
    const u32 n_bins = counters[0] * 3u * NBINS;
    for (u32 b = threadIdx.x; b < n_bins; b += BT) { s_n[b] = 0; for (int k = 0; k < 3; k++) { s_lo[b][k] = 0xffffffffu; s_hi[b][k] = 0u; } }
    __syncthreads();
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const u32 t = slot_task[i];
        if (t != NO_TASK && tasks[t].count >= 2u) {
            const SahTask& T = tasks[t];
            const Aabb b = boxes[idx[i]];
            for (int axis = 0; axis < 3; axis++) {
                const float lo = key2f(T.cb_lo[axis]), hi = key2f(T.cb_hi[axis]);
                if (!(hi > lo)) continue;
                const u32 bin = (t * 3u + (u32) axis) * NBINS + (u32) sah_bin_of(0.5f * (b.lo[axis] + b.hi[axis]), lo, hi);
                atomicAdd(&s_n[bin], 1u);
                for (int k = 0; k < 3; k++) { atomicMin(&s_lo[bin][k], f2key(b.lo[k])); atomicMax(&s_hi[bin][k], f2key(b.hi[k])); }
            }
        }
    }
    __syncthreads();
    for (u32 b = threadIdx.x; b < n_bins; b += BT) {
        if (s_n[b] == 0u) continue;
        atomicAdd(&bins[b].n, s_n[b]);
        for (int k = 0; k < 3; k++) { atomicMin(&bins[b].lo[k], s_lo[b][k]); atomicMax(&bins[b].hi[k], s_hi[b][k]); }
    }
}

RT_DEV float sah_half_area(const float* lo, const float* hi) {
    float x = hi[0] - lo[0], y = hi[1] - lo[1], z = hi[2] - lo[2];
    return x * y + y * z + z * x;
}

// Split decisions of a level, one thread per task (any number of workgroups): sweeps the bins of the task for the
// cheapest split, decides between a leaf and an inner node and leaves the decision and the boxes of the two sides for
// k_sah_number.
struct SahSides { float l[6], r[6]; };
__global__ __launch_bounds__(BT) void k_sah_decide(const SahTask* __restrict__ tasks, const SahBin* __restrict__ bins, SahDecision* __restrict__ dec,
                                                   SahSides* __restrict__ sides, u32* __restrict__ counters, float sah_tri, int leaf_max) {
    const u32 n_tasks = counters[0];
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    u32 max_depth = 0;
    if (t < n_tasks) {
        SahDecision D;
        D.axis = -2; D.split_bin = 0; D.n_left = 0; D.child_task = 0; D.cnt_left = 0; D.cnt_right = 0;
        float lbox[6] = { 0, 0, 0, 0, 0, 0 }, rbox[6] = { 0, 0, 0, 0, 0, 0 };
        const SahTask T = tasks[t];
        max_depth = T.depth;
        if (T.count >= 2u) {
            float best = 3.0e38f, area = 0.0f;
            for (int axis = 0; axis < 3; axis++) {
                const SahBin* b = bins + ((size_t) t * 3 + axis) * NBINS;
                if (!(key2f(T.cb_hi[axis]) > key2f(T.cb_lo[axis]))) continue;
                // areas of the suffixes, then a forward sweep
                float ra[NBINS]; u32 rn[NBINS];
                float lo[3] = { 3.0e38f, 3.0e38f, 3.0e38f }, hi[3] = { -3.0e38f, -3.0e38f, -3.0e38f }; u32 nn = 0;
                for (int k = NBINS - 1; k > 0; k--) {
                    if (b[k].n) for (int c = 0; c < 3; c++) { lo[c] = fminf(lo[c], key2f(b[k].lo[c])); hi[c] = fmaxf(hi[c], key2f(b[k].hi[c])); }
                    nn += b[k].n; rn[k] = nn; ra[k] = nn ? sah_half_area(lo, hi) : 0.0f;
                }
                if (b[0].n) for (int c = 0; c < 3; c++) { lo[c] = fminf(lo[c], key2f(b[0].lo[c])); hi[c] = fmaxf(hi[c], key2f(b[0].hi[c])); }
                area = sah_half_area(lo, hi); // the whole node (the same for every axis with a spread)
                for (int c = 0; c < 3; c++) { lo[c] = 3.0e38f; hi[c] = -3.0e38f; } nn = 0;
                for (int k = 0; k < NBINS - 1; k++) {
                    if (b[k].n) for (int c = 0; c < 3; c++) { lo[c] = fminf(lo[c], key2f(b[k].lo[c])); hi[c] = fmaxf(hi[c], key2f(b[k].hi[c])); }
                    nn += b[k].n;
                    if (nn == 0u || rn[k + 1] == 0u) continue;
                    float cost = sah_half_area(lo, hi) * (float) nn + ra[k + 1] * (float) rn[k + 1];
                    if (cost < best) { best = cost; D.axis = axis; D.split_bin = k + 1; D.n_left = nn; }
                }
            }
            if (D.axis >= 0) {
                const float leaf_cost = sah_tri * (float) T.count * area, split_cost = area + sah_tri * best;
                if (T.parent_slot >= 0 && (int) T.count <= leaf_max && leaf_cost <= split_cost) D.axis = -2;
            } else if (T.parent_slot < 0 || (int) T.count > leaf_max) { // leaf_max <= 8, the most a leaf link encodes
                D.axis = -1; D.n_left = (T.count + 1u) / 2u; // all centres coincide: cut the range in the middle
            }
            if (D.axis >= 0) { // boxes of the two sides
                const SahBin* b = bins + ((size_t) t * 3 + D.axis) * NBINS;
                for (int c = 0; c < 3; c++) { lbox[c] = 3.0e38f; lbox[3 + c] = -3.0e38f; rbox[c] = 3.0e38f; rbox[3 + c] = -3.0e38f; }
                for (int k = 0; k < NBINS; k++) {
                    if (!b[k].n) continue;
                    float* o = k < D.split_bin ? lbox : rbox;
                    for (int c = 0; c < 3; c++) { o[c] = fminf(o[c], key2f(b[k].lo[c])); o[3 + c] = fmaxf(o[3 + c], key2f(b[k].hi[c])); }
                }
            }
        } // (a single primitive at the root is emitted by the caller)
        if (D.axis < 0) // leaf, or middle cut of coinciding primitives: both halves keep the node's box
            for (int c = 0; c < 3; c++) { lbox[c] = T.lo[c]; lbox[3 + c] = T.hi[c]; rbox[c] = T.lo[c]; rbox[3 + c] = T.hi[c]; }
        dec[t] = D;
        SahSides sd;
        for (int c = 0; c < 6; c++) { sd.l[c] = lbox[c]; sd.r[c] = rbox[c]; }
        sides[t] = sd;
    }
    // depth of the tree so far
    for (int off = 32; off >= 1; off >>= 1) { u32 o = __shfl_xor(max_depth, off); max_depth = o > max_depth ? o : max_depth; }
    if ((threadIdx.x & 63u) == 0 && max_depth) atomicMax(&counters[2], max_depth);
}

// One workgroup of 1024 threads numbers the new inner nodes of the level and their child tasks with a prefix scan over
// the decisions (breadth first, the same from run to run) and writes the nodes and the tasks of the next level.
__global__ __launch_bounds__(1024) void k_sah_number(const SahTask* __restrict__ tasks, SahDecision* __restrict__ dec, const SahSides* __restrict__ sides,
                                                     SahTask* __restrict__ next, BvhNode* __restrict__ nodes, u32* __restrict__ counters) {
    __shared__ u32 s_wave[16];
    __shared__ u32 s_carry;
    const u32 n_tasks = counters[0], node_base = counters[1];
    const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (u32 t0 = 0; t0 < n_tasks; t0 += 1024u) {
        const u32 t = t0 + threadIdx.x;
        const bool live = t < n_tasks;
        SahDecision D; D.axis = -2; D.n_left = 0;
        if (live) D = dec[t];
        const bool split = live && D.axis >= -1;
        // exclusive scan of `split` over the tasks, in task order
        const unsigned long long m = __ballot(split);
        if (lane == 0) s_wave[wave] = (u32) __popcll(m);
        __syncthreads();
        u32 rank = s_carry + (u32) __popcll(m & ((1ull << lane) - 1ull)), total = 0;
        for (u32 w = 0; w < 16u; w++) { u32 c = s_wave[w]; if (w < wave) rank += c; total += c; }
        if (live) {
            const SahTask T = tasks[t];
            int link;
            if (split) {
                const SahSides sd = sides[t];
                const u32 id = node_base + rank;
                link = (int) id;
                dec[t].child_task = 2u * rank;
                SahTask L, R;
                L.first = T.first; L.count = D.n_left; L.parent_slot = (int) (2u * id); L.depth = T.depth + 1u;
                R.first = T.first + D.n_left; R.count = T.count - D.n_left; R.parent_slot = (int) (2u * id + 1u); R.depth = T.depth + 1u;
                for (int c = 0; c < 3; c++) { L.cb_lo[c] = 0xffffffffu; L.cb_hi[c] = 0u; R.cb_lo[c] = 0xffffffffu; R.cb_hi[c] = 0u; }
                for (int c = 0; c < 3; c++) { L.lo[c] = sd.l[c]; L.hi[c] = sd.l[3 + c]; R.lo[c] = sd.r[c]; R.hi[c] = sd.r[3 + c]; }
                next[2u * rank] = L; next[2u * rank + 1u] = R;
                BvhNode nd;
                for (int c = 0; c < 3; c++) { nd.lo0[c] = sd.l[c]; nd.hi0[c] = sd.l[3 + c]; nd.lo1[c] = sd.r[c]; nd.hi1[c] = sd.r[3 + c]; }
                nd.c0 = 0; nd.c1 = 0; nd.bound = 0.0f; nd.depth = 0;
                nodes[id] = nd;
            } else link = ~(int) ((T.first << 3) | (T.count - 1u));
            if (T.parent_slot >= 0) {
                BvhNode* p = nodes + (T.parent_slot >> 1);
                if (T.parent_slot & 1) p->c1 = link; else p->c0 = link;
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) s_carry += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) { counters[0] = 2u * s_carry; counters[1] = node_base + s_carry; }
}

__global__ void k_sah_move(const Aabb* __restrict__ boxes, const u32* __restrict__ idx, const u32* __restrict__ slot_task, u32 n,
                           const SahTask* __restrict__ tasks, SahDecision* __restrict__ dec, SahTask* __restrict__ next,
                           u32* __restrict__ idx_out, u32* __restrict__ slot_task_out) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < n;
    const u32 t = live ? slot_task[i] : NO_TASK, prim = live ? idx[i] : 0u;
    bool moving = false, left = false;
    Aabb b = Aabb();
    if (live) {
        if (t == NO_TASK || dec[t].axis == -2) { idx_out[i] = prim; slot_task_out[i] = NO_TASK; } // settled in a leaf
        else {
            moving = true;
            const SahDecision& D = dec[t];
            const SahTask& T = tasks[t];
            b = boxes[prim];
            if (D.axis >= 0) left = sah_bin_of(0.5f * (b.lo[D.axis] + b.hi[D.axis]), key2f(T.cb_lo[D.axis]), key2f(T.cb_hi[D.axis])) < D.split_bin;
            else left = (i - T.first) < D.n_left;
        }
    }
    // Primitives of a node are contiguous, so in the upper levels a whole wave belongs to one node: then the wave takes
    // its places in the two child ranges with one atomic per side and reduces the children's centroid bounds itself
    // (per-lane atomics on the few words of a big node cost 2.5 ms for the root of a 52,743-instance TLAS).
    const unsigned long long m_move = __ballot(moving);
    if (m_move == 0ull) return;
    const u32 lane = threadIdx.x & 63u;
    const int leader = __ffsll((long long) m_move) - 1;
    const u32 t0 = (u32) __shfl((int) t, leader);
    const bool uniform = __ballot(moving && t == t0) == m_move;
    u32 keys[3];
    for (int k = 0; k < 3; k++) keys[k] = f2key(0.5f * (b.lo[k] + b.hi[k]));
    if (uniform) {
        SahDecision& D = dec[t0];
        const SahTask& T = tasks[t0];
        const unsigned long long m_left = __ballot(moving && left), m_right = m_move & ~m_left;
        u32 base_l = 0, base_r = 0;
        if ((int) lane == leader) {
            if (m_left) base_l = atomicAdd(&D.cnt_left, (u32) __popcll(m_left));
            if (m_right) base_r = atomicAdd(&D.cnt_right, (u32) __popcll(m_right));
        }
        base_l = (u32) __shfl((int) base_l, leader); base_r = (u32) __shfl((int) base_r, leader);
        const unsigned long long below = (1ull << lane) - 1ull;
        if (moving) {
            const u32 pos = left ? T.first + base_l + (u32) __popcll(m_left & below) : T.first + D.n_left + base_r + (u32) __popcll(m_right & below);
            idx_out[pos] = prim; slot_task_out[pos] = D.child_task + (left ? 0u : 1u);
        }
        for (int side = 0; side < 2; side++) {
            const bool mine = moving && (left == (side == 0));
            if ((side == 0 ? m_left : m_right) == 0ull) continue;
            SahTask& C = next[D.child_task + (u32) side];
            for (int k = 0; k < 3; k++) {
                const u32 mn = wave_min_u32(mine ? keys[k] : 0xffffffffu), mx = wave_max_u32(mine ? keys[k] : 0u);
                if ((int) lane == leader) { atomicMin(&C.cb_lo[k], mn); atomicMax(&C.cb_hi[k], mx); }
            }
        }
        return;
    }
    if (!moving) return;
    SahDecision& D = dec[t];
    const SahTask& T = tasks[t];
    const u32 pos = left ? T.first + atomicAdd(&D.cnt_left, 1u) : T.first + D.n_left + atomicAdd(&D.cnt_right, 1u);
    const u32 child = D.child_task + (left ? 0u : 1u);
    idx_out[pos] = prim; slot_task_out[pos] = child;
    SahTask& C = next[child];
    for (int k = 0; k < 3; k++) { atomicMin(&C.cb_lo[k], keys[k]); atomicMax(&C.cb_hi[k], keys[k]); }
}

// root record; for a top-level tree the one-primitive leaf links (first << 3) become instance links
__global__ void k_sah_finish(const Aabb* __restrict__ scene, const u32* __restrict__ counters, const u32* __restrict__ idx, int tlas,
                             BvhNode* __restrict__ nodes) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= counters[1]) return;
    if (tlas) {
        int c0 = nodes[i].c0, c1 = nodes[i].c1;
        if (c0 < 0) nodes[i].c0 = ~(int) idx[(u32) (~c0) >> 3];
        if (c1 < 0) nodes[i].c1 = ~(int) idx[(u32) (~c1) >> 3];
    }
    if (i == 0) {
        float m = 0.0f;
        for (int k = 0; k < 3; k++) m = fmaxf(m, fmaxf(fabsf(scene->lo[k]), fabsf(scene->hi[k])));
        // counters[2] = depth of the deepest task, and the deepest tasks are leaves: as many inner levels (root = level 1)
        nodes[0].bound = m; nodes[0].depth = counters[2] > 1u ? (int) counters[2] : 1;
    }
}

// surface-area cost of the finished tree, the quantity the PLOC build keeps in its cost array: sum over inner nodes of
// their half area + sah_tri * (primitives * half area) over leaves.  One workgroup, fixed summation order.
__global__ __launch_bounds__(1024) void k_sah_cost(const BvhNode* __restrict__ nodes, const u32* __restrict__ counters, float sah_tri, int tlas,
                                                   float* __restrict__ out) {
    __shared__ float s_sum[1024];
    const u32 n_nodes = counters[1];
    float sum = 0.0f;
    for (u32 i = threadIdx.x; i < n_nodes; i += 1024u) {
        const BvhNode nd = nodes[i];
        float lo[3], hi[3];
        for (int k = 0; k < 3; k++) { lo[k] = fminf(nd.lo0[k], nd.lo1[k]); hi[k] = fmaxf(nd.hi0[k], nd.hi1[k]); }
        sum += sah_half_area(lo, hi);
        if (nd.c0 < 0) sum += sah_tri * (float) (tlas ? 1u : ((u32) (~nd.c0) & 7u) + 1u) * sah_half_area(nd.lo0, nd.hi0);
        if (nd.c1 < 0) sum += sah_tri * (float) (tlas ? 1u : ((u32) (~nd.c1) & 7u) + 1u) * sah_half_area(nd.lo1, nd.hi1);
    }
    s_sum[threadIdx.x] = sum;
    __syncthreads();
    for (u32 off = 512; off >= 1; off >>= 1) {
        if (threadIdx.x < off) s_sum[threadIdx.x] += s_sum[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = s_sum[0];
}

int sah_workspace_reserve(SahWorkspace& w, u32 n) {
    if (n <= w.capacity) return RT_OK;
    sah_workspace_free(w);
    u32 cap = n < 1024 ? 1024 : n;
    // at most cap / 2 + 1 tasks of a level hold two or more primitives, but every task index needs its bins
    RT_HIP(hipMalloc(&w.bins, sizeof(SahBin) * 3 * NBINS * (size_t) cap));
    RT_HIP(hipMalloc(&w.tasks[0], sizeof(SahTask) * (size_t) cap));
    RT_HIP(hipMalloc(&w.tasks[1], sizeof(SahTask) * (size_t) cap));
    RT_HIP(hipMalloc(&w.decisions, sizeof(SahDecision) * (size_t) cap));
    RT_HIP(hipMalloc(&w.sides, sizeof(SahSides) * (size_t) cap));
    RT_HIP(hipMalloc(&w.counters, 4 * sizeof(u32)));
    RT_HIP(hipMalloc(&w.alt_nodes, sizeof(BvhNode) * (size_t) cap));
    RT_HIP(hipMalloc(&w.alt_perm, 4 * (size_t) cap));
    w.capacity = cap;
    return RT_OK;
}
void sah_workspace_free(SahWorkspace& w) {
    void* ptrs[] = { w.bins, w.tasks[0], w.tasks[1], w.decisions, w.sides, w.counters, w.alt_nodes, w.alt_perm };
    for (void* p : ptrs) if (p) (void) hipFree(p);
    w = SahWorkspace();
}

// Builds over w.boxes[0..n), n >= 2; scene box in w.scene_box.  nodes_out: n - 1 entries; perm_out: leaf order.
int sah_build(hipStream_t s, LbvhWorkspace& w, SahWorkspace& sw, u32 n, int leaf_max, float sah_tri, bool tlas, BvhNode* nodes_out, u32* perm_out) {
    if (leaf_max > 8) leaf_max = 8;
    if (tlas) leaf_max = 1;
    int rc = sah_workspace_reserve(sw, n); if (rc) return rc;
    SahTask root; memset(&root, 0, sizeof root);
    root.first = 0; root.count = n; root.parent_slot = -1; root.depth = 0;
    for (int k = 0; k < 3; k++) { root.cb_lo[k] = 0xffffffffu; root.cb_hi[k] = 0u; }
    RT_HIP(hipMemcpyAsync(sw.tasks[0], &root, sizeof root, hipMemcpyHostToDevice, s));
    RT_HIP(hipMemcpyAsync((char*) sw.tasks[0] + offsetof(SahTask, lo), w.scene_box, sizeof(Aabb), hipMemcpyDeviceToDevice, s)); // root box
    u32* idx[2] = { w.vals, w.vals_tmp }; u32* slot[2] = { w.keys, w.keys_tmp };
    hipLaunchKernelGGL(k_sah_init, grid_for(n), dim3(BT), 0, s, w.boxes, n, idx[0], slot[0], (SahTask*) sw.tasks[0], sw.counters);
    // The host reads the task count back only every fourth level (a read-back is a stream synchronisation); the kernels
    // take it from device memory, so the up to three levels launched after the tree has closed are no-ops.
    u32 n_tasks = 1; int cur = 0;
    for (int level = 0; level < 96 && n_tasks > 0; level++) {
        const u32 task_bound = level < 20 ? ((1u << level) < n ? (1u << level) : n) : n; // a level has at most 2^level tasks
        hipLaunchKernelGGL(k_sah_clear, grid_for(task_bound * 3 * NBINS), dim3(BT), 0, s, (SahBin*) sw.bins, (const u32*) sw.counters);
        if (task_bound <= BIN_LDS_TASKS)
            hipLaunchKernelGGL(k_sah_bin_lds, grid_for(n), dim3(BT), 0, s, w.boxes, idx[cur], slot[cur], n, (const SahTask*) sw.tasks[cur], (SahBin*) sw.bins, (const u32*) sw.counters);
        else
            hipLaunchKernelGGL(k_sah_bin, grid_for(n), dim3(BT), 0, s, w.boxes, idx[cur], slot[cur], n, (const SahTask*) sw.tasks[cur], (SahBin*) sw.bins);
        hipLaunchKernelGGL(k_sah_decide, grid_for(task_bound), dim3(BT), 0, s, (const SahTask*) sw.tasks[cur], (const SahBin*) sw.bins,
                           (SahDecision*) sw.decisions, (SahSides*) sw.sides, sw.counters, sah_tri, leaf_max);
        hipLaunchKernelGGL(k_sah_number, dim3(1), dim3(1024), 0, s, (const SahTask*) sw.tasks[cur], (SahDecision*) sw.decisions,
                           (const SahSides*) sw.sides, (SahTask*) sw.tasks[cur ^ 1], nodes_out, sw.counters);
        hipLaunchKernelGGL(k_sah_move, grid_for(n), dim3(BT), 0, s, w.boxes, idx[cur], slot[cur], n, (const SahTask*) sw.tasks[cur],
                           (SahDecision*) sw.decisions, (SahTask*) sw.tasks[cur ^ 1], idx[cur ^ 1], slot[cur ^ 1]);
        cur ^= 1;
        if ((level & 3) == 3) {
            RT_HIP(hipMemcpyAsync(&n_tasks, sw.counters, 4, hipMemcpyDeviceToHost, s));
            RT_HIP(hipStreamSynchronize(s));
        }
    }
    if (n_tasks != 0) { set_error("sah_build: the tree did not close after 96 levels"); return RT_ERR_LIMIT; }
    // cost first: k_sah_finish rewrites the leaf links of a top-level tree (counters[3] holds the cost as a float)
    hipLaunchKernelGGL(k_sah_cost, dim3(1), dim3(1024), 0, s, (const BvhNode*) nodes_out, (const u32*) sw.counters, sah_tri, tlas ? 1 : 0, (float*) (sw.counters + 3));
    hipLaunchKernelGGL(k_sah_finish, grid_for(n), dim3(BT), 0, s, w.scene_box, sw.counters, idx[cur], tlas ? 1 : 0, nodes_out);
    RT_HIP(hipMemcpyAsync(perm_out, idx[cur], (size_t) n * 4, hipMemcpyDeviceToDevice, s));
    RT_HIP(hipGetLastError());
    return RT_OK;
}

} // namespace rti
