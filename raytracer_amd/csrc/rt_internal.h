// rt_internal.h -- host-side declarations shared by the .hip translation units of librt_mi355x.so
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/rt_abi.h"
#include "rt_dev_lbvh.h"
#include "rt_dev_shade.h"

namespace rti {

using rtd::u32;

void set_error(const char* fmt, ...);

#define RT_HIP(call)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            rti::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return RT_ERR_DEVICE;                                                                 \
        }                                                                                         \
    } while (0)

// ---- device radix sort (rt_sort.hip) -----------------------------------------------------------------------
// stable LSD sort of (key, value) pairs on `key_bits` low bits; scratch sizes from sort_scratch_bytes
size_t sort_scratch_bytes(u32 n);
int radix_sort_pairs(hipStream_t s, u32* keys, u32* vals, u32* keys_tmp, u32* vals_tmp, void* scratch, u32 n,
                     int key_bits); // result ends in keys/vals

// ---- LBVH (rt_bvh.hip) ---------------------------------------------------------------------------------------
struct LbvhWorkspace {
    rtd::Aabb* boxes = nullptr;      // per primitive
    rtd::Aabb* node_boxes = nullptr; // per internal node
    rtd::Aabb* scene_box = nullptr;
    u32 *keys = nullptr, *vals = nullptr, *keys_tmp = nullptr, *vals_tmp = nullptr;
    rtd::BuildNode* build_nodes = nullptr;
    int* leaf_parent = nullptr;
    u32* flags = nullptr;
    int* depth = nullptr;
    float* cost = nullptr;
    u32* collapsed = nullptr;
    void* sort_scratch = nullptr;
    rtd::BvhNode* tree_nodes = nullptr; // the tree as the builders write it (64-byte nodes), packed afterwards
    rtd::TreeInfo* tree_info = nullptr;
    u32 capacity = 0;
};
int lbvh_workspace_reserve(LbvhWorkspace& w, u32 n);
void lbvh_workspace_free(LbvhWorkspace& w);

struct PlocWorkspace { // rt_ploc.hip
    int* cid[2] = { nullptr, nullptr };
    rtd::Aabb* cbox[2] = { nullptr, nullptr };
    int* nn = nullptr;
    unsigned long long *excl = nullptr, *sums = nullptr;
    u32* totals = nullptr;
    u32* leaf_first = nullptr;
    u32 capacity = 0;
};
int ploc_workspace_reserve(PlocWorkspace& w, u32 n);
void ploc_workspace_free(PlocWorkspace& w);
int ploc_build(hipStream_t s, LbvhWorkspace& w, PlocWorkspace& pw, u32 n, int leaf_max, float sah_tri, bool tlas, rtd::BvhNode* nodes_out, u32* perm_out);

struct SahWorkspace { // rt_sah.hip (builder 2: top-down binned SAH)
    void* bins = nullptr; void* tasks[2] = { nullptr, nullptr }; void* decisions = nullptr;
    u32* counters = nullptr;                                   // tasks of the level, inner nodes, depth, tree cost (float)
    void* sides = nullptr; // boxes of the two sides of every split of a level (k_sah_decide -> k_sah_number)
    rtd::BvhNode* alt_nodes = nullptr; u32* alt_perm = nullptr; // second tree when two builders compete (builder 4)
    u32 capacity = 0;
};
int sah_workspace_reserve(SahWorkspace& w, u32 n);
void sah_workspace_free(SahWorkspace& w);
int sah_build(hipStream_t s, LbvhWorkspace& w, SahWorkspace& sw, u32 n, int leaf_max, float sah_tri, bool tlas, rtd::BvhNode* nodes_out, u32* perm_out);

// boxes for the n triangles of a BLAS -> w.boxes
int lbvh_triangle_boxes(hipStream_t s, LbvhWorkspace& w, const RtVertex* verts, const uint16_t* indices, u32 n_tris);
// world boxes of n instances (8 corners of each BLAS root box through object_to_world, padded) -> w.boxes
// exact: boxes from every vertex of the instance's BLAS (tighter); otherwise from the corners of its object-space box
int lbvh_instance_boxes(hipStream_t s, LbvhWorkspace& w, const rtd::InstShade* shade, const rtd::InstRec* insts, u32 n, bool exact);
// w.boxes[dst] = w.boxes[src] (rt_build_tlas: the last instance's box takes the slot of the inlined instance, SceneView::inl_inst)
int lbvh_move_box(hipStream_t s, LbvhWorkspace& w, u32 dst, u32 src);
// builds the hierarchy over w.boxes[0..n); nodes_out has max(1, n-1) entries; perm_out[n] = sorted -> original.
// builder: 0 = Karras LBVH hierarchy, 1 = PLOC agglomerative clustering (pw required), 2 = top-down binned SAH for
// up to SAH_BUILD_MAX primitives (sw required; larger inputs fall back to PLOC), 3 = SAH for bottom-level trees and
// PLOC for the top level, 4 = both PLOC and SAH, keeping the tree with the lower surface-area cost
// sah_tri: cost of one triangle test relative to one two-child node step in the surface-area heuristic
static constexpr u32 SAH_BUILD_MAX = 1u << 18;
// builds into w.tree_nodes, packs the tree into nodes_out (max(n - 1, 1) nodes) and leaves its TreeInfo in w.tree_info
int lbvh_build(hipStream_t s, LbvhWorkspace& w, PlocWorkspace* pw, SahWorkspace* sw, int builder, float sah_tri, u32 n, int leaf_max, bool tlas, rtd::PackedNode* nodes_out, u32* perm_out);
// TriRec array in leaf order
int lbvh_write_trirecs(hipStream_t s, const u32* perm, const RtVertex* verts, const uint16_t* indices,
                       const u32* tri_geometry, const rtd::GeomRec* geoms, u32 n_tris, rtd::TriRec* out);

// flat tree over the triangles of all instances of a small scene (rt_dev_scene.h: FlatTri, trace_flat)
struct FlatSource { const RtVertex* verts; const uint16_t* indices; const u32* inv_perm; /* primitive -> TriRec index of the BLAS */ float o2w[16]; u32 first, n_tris; };
struct FlatSources { FlatSource s[RT_FLAT_MAX_INSTANCES]; u32 n; }; // s[k].first = triangles of the instances before k
int lbvh_invert_perm(hipStream_t s, const u32* perm, u32 n, u32* inv);
// builds the flat tree: `join` (scratch) holds 1 + sum of max(n_i - 1, 1) BvhNodes, `sub_boxes` (scratch) RT_FLAT_MAX_INSTANCES + 1 boxes,
// `perm` all triangles, `tris` three copies of them, `nodes_out` as many nodes as `join`; TreeInfo ends in w.tree_info
int lbvh_flat_build(hipStream_t s, LbvhWorkspace& w, PlocWorkspace* pw, SahWorkspace* sw, int builder, float sah_tri, const FlatSources& src,
                    rtd::BvhNode* join, rtd::Aabb* sub_boxes, u32* perm, rtd::FlatTri* tris, rtd::PackedNode* nodes_out);

// ---- wavefront pipeline (rt_kernels.hip) ---------------------------------------------------------------------
struct RayQueue { float *ox, *oy, *oz, *dx, *dy, *dz; u32* path; };
struct HitBuf { float *t, *u, *v; int *inst, *tri; };
struct PathState {
    u32* rng;
    float *thr_x, *thr_y, *thr_z;
    float *rad_x, *rad_y, *rad_z, *alpha;
    float *fd_x, *fd_y, *fd_z;           // probe pass: direction after trace_path_sample (raytracing.hlsl:258)
    float *sum_x, *sum_y, *sum_z, *sum_a; // per-dispatch sample sum
};
// state of a path in flight that travels with its ray (slot-indexed, ping-pong like the queues): used by the fused camera
// pass, where k_shade groups the surviving rays by direction octant, so that reads and writes stay coalesced
struct SlotState { u32* rng; float *thr_x, *thr_y, *thr_z, *rad_x, *rad_y, *rad_z; };
struct GatherQueue { u32* path; float *hx, *hy, *hz, *tf, *kx, *ky, *kz; int* ti; };

// hierarchical gather (rt_set_gather_mode): per-instance tree over the sample points, DESIGN.md section 8
struct alignas(16) HgNode { float cx, cy, cz, radius2; u32 first, count, right, depth; }; // first: within the slice
struct HgSlice { u32 node_offset, node_count, max_depth, pad; };
struct HgView {
    const HgNode* nodes;      // all slices concatenated; child links are relative to the slice's node_offset
    float4* sums;             // per node: sum of payloads (refreshed by k_hg_sums)
    const float4* pos;        // per point, sorted order (same offsets as the slices of the sample-point arrays)
    float4* pay;              // per point, sorted order, refreshed by k_hg_sums
    const u32* order;         // sorted position -> index in generation order (within the slice)
    const HgSlice* slices;
    float max_solid_angle;
    u32 enabled;
};

struct DeviceStats { // lives in device memory
    unsigned long long rays_camera, rays_probe, gather_pairs, nodes, tris, insts, node_trips, tri_trips, inst_trips, max_ray_steps;
    unsigned long long pool_err[8]; unsigned long long pool_prof[18]; // (RT_POOL_DEBUG: where the pooled kernel's time goes, summed over waves) // RT_POOL_DEBUG builds: first bounds violation the pooled traversal caught (code, value)
    unsigned long long hg_overflow; // k_gather_hg ran out of walk stack and dropped a subtree: the host reports RT_ERR_LIMIT (rt_sync, rt_get_stats, reads)
};

struct PassArgs {
    RtGlobals g;
    float camera_origin[3];      // camera_to_world[3].xyz / camera_to_world[3].w (raytracing.hlsl:143), divided once on the host
    u32* task_counter;           // persistent traversal: 64 task counters + 1 done counter, 128 bytes apart; all zero between launches
    u32 persistent_slots;        // wave slots of the device for the traversal kernel (CUs x 4 SIMDs x 5 waves), 0 = never persistent
    u32 persistent;              // bit 0: queued (secondary / probe) launches persistent, bit 1: primary launches too
    u32 pool_groups;             // > 0: queued launches of two-level scenes run the pooled traversal (rt_dev_pool.h) on this many persistent workgroups (RT_POOL)
    unsigned long long dbg_blas_lo, dbg_blas_hi; u32 dbg_tlas_bytes; // RT_POOL_DEBUG builds: address range of all bottom-level trees, size of the top-level node array
    u32 fuse_primary, flat_rows; // bounce 0 of the fused camera pass walks and shades in one kernel (k_shade<.., TRACE>); stack rows its walk needs (flat tree depth + 2)
    u32 lazy_render_target;      // the resolve leaves the render target alone (rt_api.hip derives it from the accumulator when it is read)
    u32 defer_resolve;           // fused camera pass: finished paths leave their pixel's sample sum in ps.sum_* for k_frame_resolve
    u32 stack_in_lds;            // the trees are shallow enough for the LDS rows of the traversal stack alone (LaneStack<false>)
    u32 flat_waves7;             // two-instance flat tree shallow enough for 22 LDS rows: 7 waves per SIMD (RT_FLAT_WAVES7)
    u32 flat_waves8;             // one-instance flat tree shallow enough for 20 LDS rows: traversal kernels at 8 waves per SIMD
    u32 wave_major;              // queued traverse launches: task -> (segment, wave) mapping, see k_traverse (RT_WAVE_MAJOR)
    float sort_box[6]; float sort_chord; u32 sort_box_on; // k_shade rank_sorted: new rays are queued by the length of their way through this world box, longest first: 1 = flat scenes, the box of the instance with the most triangles (classes of its diagonal), 2 = the top level's box (classes around sort_chord); RT_SORT_BOX=0: neither
    float inv_size[2];           // 1 / width, 1 / height (IEEE quotients of the host: camera_ray)
    u32 xcd_bands;               // traverse launches: the workgroups of one XCD take neighbouring tiles / segments, see k_traverse (RT_XCD_BANDS)
    u32 ray_sort;                // fused camera pass: survivors of a bounce are grouped by direction octant inside their segment (RT_RAY_SORT)
    SlotState st[2];             // fused camera pass: path state by queue slot
    float fudge;                 // translucent_bssrdf_fudge or 0 when SSS is disabled (raytracing.cpp:653-657)
    u32 width, height;
    u32 tile_rank, tile_world, tiles_x, tiles_y, n_local_tiles;
    u32 tiles_x_magic;           // floor(2^32 / tiles_x) + 1 when tile / tiles_x == umulhi(tile, magic) for every tile, else 0
    u32 n_paths;                 // capacity of the path arrays
    u32 probe;                   // 1 = translucent_rgen pass
    float tmin, tmax;
    rtd::SceneView scene;
    rtd::ShadeScene shade;
    RayQueue q[2];
    HitBuf hits;
    PathState ps;
    GatherQueue gq;
    u32 n_segments;              // queue segments of 1024 rays in this pass (tiles, or ceil(n_points / 1024))
    u32 gather_enabled;
    u32* seg_count[2];           // rays per segment of the two ping-pong queues
    u32* gseg_count;             // gather items per segment
    u32* goff;                   // [n_segments + 1] exclusive scan of gseg_count (k_gather_offsets)
    const float4* sp_pay;        // payloads of sp_read in the Morton order of the points (k_sort_payload, k_gather)
    // batched gather (rt_api.hip run_pass, RT_GATHER_BATCH): the gather items of up to bg_rows shade launches of a frame are kept --
    // row r of the nine gather-queue arrays starts at slot r * bg_stride, its per-segment counts at bg_counts[r * bg_cols] -- and
    // evaluated by ONE k_gather launch (goff then numbers the items of all rows: bg_rows * bg_cols + 1 entries, then the block
    // counter of that launch); 0 = every shade launch is followed by its own gather launch
    u32 bg_rows, bg_cols, bg_stride, bg_groups, bg_block; // bg_groups workgroups of bg_block threads
    u32 bg_to_sums;              // k_gather_apply adds the terms to the pixels' sample sums (fused camera pass) instead of the paths' radiance
    const u32* bg_counts;
    int* stack_spill;            // [48][n_paths] overflow of the per-lane traversal stack
    DeviceStats* stats;
    RtHit* capture;              // [(bounces+1) * width*height] or null
    float* accumulator;          // RGBA32F
    uint8_t* render_target;      // RGBA8
    // translucent sample points (all instances concatenated)
    RtSamplePoint* sp_read;
    RtSamplePoint* sp_write;
    const float* sp_normal;      // 3 per point
    const u32* sp_slice;         // per point -> slice index
    u32 n_points;
    u32 probe_lo, probe_hi;      // translucent_rgen dispatches trace the points [probe_lo, probe_hi) (all of them unless rt_set_probe_split)
    HgView hg;
    const float4* gather_delta;  // k_gather<3>: differences of consecutive scaled texels, fetched through L1 (launch_gather_delta); null: both texels from LDS
    const float* bssrdf_table;   // rgb interleaved
    int bssrdf_texels;
};

int launch_probe_raygen(hipStream_t s, const PassArgs& a, u32 sample_index);
// primary: bounce 0 of the camera pass (rays derived from pixels; in_queue unused)
// tail_shade (queued launches, last bounce of the fused camera pass): the kernel shades its hits itself, no launch_shade follows
int launch_traverse(hipStream_t s, const PassArgs& a, u32 in_queue, bool count_traversal, bool primary, u32 sample_index, bool tail_shade = false);
// fused: finished camera paths resolve their pixel in the shade kernel (no launch_camera_resolve afterwards)
// walk (queued bounces of the fused pass, flat one-instance scenes: PassArgs::fuse_primary): the kernel walks its rays itself, no launch_traverse before it
int launch_shade(hipStream_t s, const PassArgs& a, u32 in_queue, u32 bounce, u32 sample_index, bool primary, bool fused, bool walk = false);
int launch_gather(hipStream_t s, const PassArgs& a);
// batched gather: one launch for the items of a.bg_rows shade launches (a.gq = row 0), then their terms added to the paths' radiance row by row
int launch_gather_batched(hipStream_t s, const PassArgs& a);
int launch_gather_delta(hipStream_t s, const float* table, int n, float factor, float4* delta);
int launch_sort_payload(hipStream_t s, const RtSamplePoint* src, const u32* order, const u32* slice_of, const rtd::TranslucentSlice* slices, u32 n, float4* pay);
int launch_hg_sums(hipStream_t s, const PassArgs& a, u32 n_slices); // refresh HgView::pay / sums from sp_read
int launch_camera_resolve(hipStream_t s, const PassArgs& a);
int launch_frame_resolve(hipStream_t s, const PassArgs& a); // pipelined frames: adds the sample sums the fused shades left in ps.sum_* (a.defer_resolve)
int launch_probe_resolve(hipStream_t s, const PassArgs& a);
int launch_fill_capture(hipStream_t s, const PassArgs& a);
// tile split: the accumulator pixels of the tiles of rank a.tile_rank of a.tile_world, packed (1,024 float4 per tile) / written back
int launch_pack_tiles(hipStream_t s, const PassArgs& a, void* packed);
int launch_unpack_tiles(hipStream_t s, const PassArgs& a, const void* packed);
int launch_resolve_only(hipStream_t s, const PassArgs& a); // raytracing.hlsl:161 after an external accumulator sum

// ---- blue-noise sample points (rt_bluenoise.hip) ----------------------------------------------------------------
struct BluenoiseParams { // ComputeGlobals, bluenoise.h:16-32
    u32 rng_seed, triangles_count, indices_offset;
    float total_surface_area;
    float transform[16];
    float rejection_radius;
    float grid_origin[3];
    float grid_cell_width;
    u32 grid_dims[3];
    u32 buckets;
    u32 initial_count;
    u32 phases;
    u32 phase_sequence[27];
};
// returns the number of points; points/normals are device arrays with capacity `capacity`
int bluenoise_generate(hipStream_t s, const BluenoiseParams& p, const RtVertex* verts, const uint16_t* indices,
                       const float* partial_areas, RtSamplePoint** points_out, float** normals_out, u32* count_out);

} // namespace rti
