// rt_api.hip -- context, scene management and the C-ABI of librt_mi355x.so (include/rt_abi.h).
// Host logic mirrors src/raytracing.cpp of the reference (build_blas :316-444, build_tlas :446-551,
// generate_translucent_samples :553-650, dispatch_rays :652-747) with HIP memory and streams in place of D3D12.
#include "rt_internal.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <limits>
#include <mutex>

using namespace rtd;
using namespace rti;

namespace rti {
static thread_local char g_error[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap; va_start(ap, fmt);
    vsnprintf(g_error, sizeof g_error, fmt, ap);
    va_end(ap);
}
} // namespace rti

namespace {

struct TranslucentMesh { // raytracing.cpp:49-53 + BluenoisePreprocess (bluenoise.h:6-12)
    int blas;
    u32 ib_offset, indices_count;
    float total_surface_area;
    float aabb_min[3], aabb_max[3];
    float* d_partial_areas;
};

struct Blas {
    RtVertex* d_verts = nullptr; uint16_t* d_indices = nullptr; GeomRec* d_geoms = nullptr; u32* d_tri_geometry = nullptr;
    PackedNode* d_nodes = nullptr; TriRec* d_tris = nullptr; /* inside the d_nodes allocation */ u32* d_perm = nullptr; TreeInfo info = {};
    u32* d_inv_perm = nullptr; // primitive (index-buffer order) -> TriRec index: what a hit found in the flat tree reports
    std::vector<RtVertex> verts; std::vector<uint16_t> indices; std::vector<GeomRec> geoms;
    u32 n_tris = 0; int depth = 0;
    u32 translucent_ids_index = 0, translucent_ids_count = 0;
};

struct TranslucentInstance { // raytracing.cpp:54-65
    u32 translucent_id, instance_id;
    float transform[16];
    float scale_factor;
    u32 samples_count;
};

template <typename T> void dev_free(T*& p) { if (p) { (void) hipFree(p); p = nullptr; } }

} // namespace

struct rt_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    LbvhWorkspace lbvh; PlocWorkspace ploc; SahWorkspace sah; int bvh_builder = 4; float sah_tri = 0.5f; // RT_BVH_BUILDER: 0 LBVH, 1 PLOC, 2 binned SAH, 3 SAH for BLAS + PLOC for the TLAS, 4 the cheaper of PLOC and SAH (default); RT_SAH_TRI
    bool exact_instance_boxes = true; // RT_EXACT_INSTANCE_BOXES=0: TLAS boxes from the corners of the BLAS box (tuning aid)
    std::vector<Blas> blas;
    std::vector<TranslucentMesh> translucent_meshes;
    std::vector<float> translucent_mean_area; // g_translucent_properties
    std::vector<TranslucentInstance> translucent_instances;
    // TLAS
    u32 n_instances = 0;
    InstRec* d_inst_rec = nullptr; InstShade* d_inst_shade = nullptr; PackedNode* d_tlas_nodes = nullptr; TreeInfo tlas_info = {}; u32* d_tlas_perm = nullptr;
    int tlas_depth = 0, max_blas_depth = 0; bool stack_lds_ok = true; // RT_STACK_LDS=0: always the variant with the global overflow (tuning aid)
    // flat tree: scenes of at most RT_FLAT_MAX_INSTANCES instances get ONE tree over all instanced triangles besides the two
    // levels (trace_flat, rt_dev_scene.h); flat_instances = 0: none.  RT_FLAT=0 switches it off (tuning aid, two-level tests).
    bool stack_deep_ok = true; // RT_STACK_DEEP=0: deep trees always on the spilling stack (tuning aid)
    bool flat_waves7 = true; /* RT_FLAT_WAVES7 */
    bool flat_enabled = true, flat_waves8 = true; u32 flat_instances = 0, flat_tris = 0, flat_first1 = 0;
    // inlined instance of the two-level walk (SceneView::inl_inst); RT_INLINE=0 switches it off (tuning aid)
    float sort_chord_pct = 0.5f; /* RT_SORT_CHORD */ int sort_box = 1; float flat_box[RT_FLAT_MAX_INSTANCES][6] = {}; u32 flat_big = 0; // RT_SORT_BOX
    int flat_inl_inst = -1; u32 flat_inst0 = 0; bool flat_inline_enabled = true; /* RT_FLAT_INLINE */
    float inl_box[RT_INLINE_MAX_TRIS][6] = {}; u32 inl_box_same = 0;
    bool inline_enabled = true; int inl_inst = -1; u32 inl_count = 0; const TriRec* inl_tris = nullptr; u32 inl_copy_bytes = 0; float inl_w2o[12] = {};
    PackedNode* d_flat_nodes = nullptr; FlatTri* d_flat_tris = nullptr; /* inside the d_flat_nodes allocation */ u32* d_flat_perm = nullptr;
    TreeInfo flat_info = {}; float flat_w2o[RT_FLAT_MAX_INSTANCES][12] = {}; double flat_build_ms = 0;
    // translucent sample points, all instances concatenated
    u32 n_points = 0;
    RtSamplePoint *d_sp_read = nullptr, *d_sp_write = nullptr; float* d_sp_normal = nullptr; u32* d_sp_slice = nullptr;
    TranslucentSlice* d_slices = nullptr; int* d_slot_to_ti = nullptr; u32 slot_count = 0;
    std::vector<TranslucentSlice> slices;
    // hierarchical gather (rt_set_gather_mode)
    HgNode* d_hg_nodes = nullptr; float4* d_hg_sums = nullptr; float4* d_hg_pos = nullptr; float4* d_hg_pay = nullptr;
    u32* d_hg_order = nullptr; HgSlice* d_hg_slices = nullptr; bool hg_enabled = false, hg_built = false; float hg_max_solid_angle = 0.0f;
    // frame state
    RtGlobals g; u32 flags = RT_FLAG_COLLECT_SAMPLES | RT_FLAG_ENABLE_SSS;
    u32 width = 0, height = 0;
    float* d_accumulator = nullptr; uint8_t* d_render_target = nullptr;
    RtHit* d_capture = nullptr; size_t capture_records = 0; u32 capture_bounces = 0;
    u32 tile_rank = 0, tile_world = 1;
    // sample-point pass partitioned over ranks (rt_set_probe_split) and the caller's exchange of the payload blocks after it (rt_set_probe_exchange)
    u32 probe_rank = 0, probe_world = 1; rt_probe_exchange_fn probe_exchange = nullptr; void* probe_exchange_user = nullptr;
    // path buffers
    u32 n_paths = 0; float* d_arena = nullptr;
    RayQueue q[2]; HitBuf hits; PathState ps; SlotState st[2]; GatherQueue gq; int* d_stack_spill = nullptr;
    u32* d_task_counter = nullptr; u32 persistent_slots = 0; int persistent = 0; // RT_PERSISTENT: 0 never (default), bit 0 queued launches, bit 1 primary launches
    // grid top level (SceneView::grid_*, trace_run_grid): RT_TLAS_GRID = 0 never, 1 where the instances suit it (default), 2 wherever the stacks allow
    int tlas_grid = 0; u32* d_grid_start = nullptr; GridEntry* d_grid_entries = nullptr;
    u32 grid_cells = 0, grid_n_entries = 0, grid_res[3] = { 0, 0, 0 }; float grid_lo[3] = {}, grid_cell[3] = {}, grid_slack = 0.0f, grid_density = 1.0f;
    int pool = 0; u32 cu_count = 0; // RT_POOL=2: pooled traversal for the queued launches of two-level scenes (rt_dev_pool.h), workgroups per CU; off by default (profiles/r03_ab_pool_c5.txt: right, and 2-4x slower)
    // the BSSRDF gather of a bounce on a side stream, beside the next bounce's traversal (run_pass): one side stream and two
    // events per lane, made with the lane (ensure_gather_side); RT_GATHER_SIDE=1 -- measured, not kept (profiles/r03_ab_gather_side.txt): default in stream order
    int gather_side = 0; hipStream_t gather_stream[2] = { nullptr, nullptr }; hipEvent_t gather_ev[2][2] = { { nullptr, nullptr }, { nullptr, nullptr } };
    int cur_lane = 0; bool gather_pending = false;
    // batched gather (run_pass; RT_GATHER_BATCH=0 switches it off, RT_GATHER_BATCH_MB bounds a lane's buffers): the gather items
    // of a frame's shade launches are kept, row after row, and evaluated by one launch per frame (or per `rows` shade launches)
    // instead of one small launch per bounce -- per lane: nine arrays of rows x n_paths items, the rows' per-segment counts, the
    // numbering of the items (+ the launch's block counter)
    struct BatchGather { float* d_arena = nullptr; u32* d_counts = nullptr; u32* d_off = nullptr; u32 rows = 0, stride = 0, cols_cap = 0; };
    BatchGather bg[2]; int gather_batch = 1; size_t gather_batch_mb = 3072; bool no_alloc_call = false; int gather_fused = 1; /* RT_GATHER_FUSED: frames with the batched gather run the fused camera pass */ int gather_groups = 2, gather_block = 256; /* RT_GATHER_GROUPS, RT_GATHER_BLOCK: workgroups per CU of the batched launch, threads of each */
    int fuse_tail = 1; // RT_FUSE_TAIL: the last bounce's traversal kernel of the fused camera pass shades its hits itself
    int fuse_queued = 1 << 30; // RT_FUSE_QUEUED: the queued bounces (up to this one) of the fused camera pass as one kernel each (walk + shade) in flat one-instance scenes; 0 = none
    int fuse_primary = 1; // RT_FUSE_PRIMARY: bounce 0 of the fused camera pass as one kernel in flat one-instance scenes (k_shade<.., TRACE>)
    int wave_major = 1; // RT_WAVE_MAJOR
    int xcd_bands = 1;  // RT_XCD_BANDS
    int ray_sort = 1; // RT_RAY_SORT: group the surviving rays of a bounce by direction octant (fused camera pass)
    u32* d_seg_count = nullptr; u32 n_segments_cap = 0; DeviceStats* d_stats = nullptr;
    // frames in flight: consecutive camera-only dispatches alternate between two lanes (a stream and a set of path buffers
    // each), so that the HBM-bound shade kernels of one frame run beside the issue-bound traversal of the other and the
    // tails of the launches are filled; the fused shades then leave the pixels' sample sums in ps.sum_* and one small
    // kernel per frame (k_frame_resolve), which waits for the frame before, adds them to the accumulator (run_pass).
    // Lane 0 is `stream` and the members above; lane 1's stream and buffers sit here and are swapped in for the
    // duration of its dispatch.
    struct PathBuffers { float* d_arena; u32* d_seg_count; RayQueue q[2]; HitBuf hits; PathState ps; SlotState st[2]; GatherQueue gq; int* d_stack_spill; };
    PathBuffers alt; hipStream_t alt_stream = nullptr; hipEvent_t lane_done[2] = { nullptr, nullptr };
    int frames_in_flight = 2;              // RT_FRAMES_IN_FLIGHT: 1 = every dispatch on `stream`, in order
    bool pipelined = false;                // the calls since the last quiesce() were pipelined dispatches: lane 1 may hold work
    int next_lane = 0; bool have_previous = false;
    // frames with the sample-point pass or the gather are pipelined too: each lane gathers from its own copy of the sample
    // points (d_sp_read / d_sp_read_alt), the passes that update d_sp_write and the copies out of it run in frame order
    // (probe_done), the rest of a frame beside the frame before.  sp_version counts the updates of d_sp_write.
    float4 *d_sp_pay = nullptr, *d_sp_pay_alt = nullptr; // payloads of d_sp_read / d_sp_read_alt in Morton order (k_gather)
    RtSamplePoint* d_sp_read_alt = nullptr; hipEvent_t probe_done[2] = { nullptr, nullptr }; int last_probe_lane = -1;
    uint64_t sp_version = 0, sp_read_version[2] = { 0, 0 };
    // With the batched gather nothing of a frame's camera pass reads the sample points before its gather launch at the end, so the
    // sample-point pass (and the copy out of it) of a pipelined frame runs on a stream of its own, with path buffers of its own
    // (a few segments: one path per point), beside the camera pass; the frame's gather launch waits for it (probe_wait).  The
    // passes of consecutive frames stay in order by themselves: one stream.  RT_PROBE_SIDE=0: on the frame's lane, before its camera pass
    int probe_side = 1; hipStream_t probe_stream = nullptr; PathBuffers probe_pb; u32 probe_paths = 0, probe_cap = 0;
    hipEvent_t probe_wait = nullptr; bool lane_ran[2] = { false, false }; bool last_probe_side = false; // (last_probe_lane's pass ran on the probe stream)
    hipEvent_t accumulate_after = nullptr; // run_pass: the frame's resolve waits for this (the previous frame's end)
    bool defer_resolve = false;            // run_pass: pipelined frame (k_frame_resolve instead of the resolve inside k_shade)
    // pipelined frames do not write the 8-bit render target; rt_read_render_target derives it from the accumulator
    // (k_resolve_only: the same expression on the same sums) when the last frames left it behind
    bool render_target_stale = false; u32 render_target_frames = 0; bool last_pass_lazy = false; // (run_pass: the camera pass left the render target to the next read)
    // BSSRDF tables
    float* d_table[2] = { nullptr, nullptr }; int table_texels[2] = { 0, 0 }; int profile = 0;
    // differences of consecutive scaled texels for k_gather<3> (ensure_gather_delta): valid for (table, factor) below
    float4* d_gather_delta = nullptr; const float* delta_table = nullptr; float delta_factor = 0.0f; int delta_texels = 0; int gather_l1 = 0; /* RT_GATHER_L1=1: measured, not kept (profiles/r03_ab_gather_l1.txt) */
    // stats
    std::vector<std::pair<hipEvent_t, hipEvent_t>> timed, timed_primary, timed_gather; double traverse_ms = 0, traverse_primary_ms = 0, gather_ms = 0;
    uint64_t traverse_launches = 0, gather_launches = 0, primary_rays = 0, traverse_launches_timed = 0, gather_launches_timed = 0;
    uint64_t primary_launches = 0, primary_launches_timed = 0; // the bounce-0 launches among the traverse launches (coherent rays derived from pixels)
    uint64_t dispatch_index = 0; // rt_dispatch_rays calls so far (RT_FLAG_TIME_SAMPLED times every 8th)
    double blas_build_ms = 0, tlas_build_ms = 0; uint64_t blas_triangles = 0;
    void* d_static_scratch = nullptr; // this GPU's copy of the static scratch array of rt_dispatch_rays_static (resolved at rt_create)
    double bluenoise_ms = 0; uint64_t bluenoise_points = 0; // wall time and result of the last rt_generate_translucent_samples
    hipEvent_t build_ev[2] = { nullptr, nullptr }; // bracket the builds of rt_build_blas / rt_build_tlas (created once: no error path can leak them)
    rt_ctx() { memset(&g, 0, sizeof g); memset(q, 0, sizeof q); memset(&hits, 0, sizeof hits); memset(&ps, 0, sizeof ps); memset(st, 0, sizeof st); memset(&gq, 0, sizeof gq); memset(&alt, 0, sizeof alt); memset(&probe_pb, 0, sizeof probe_pb); }
};

namespace {

// S7: world->object from object->world, in double (cofactors), rounded once
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

void swap_path_buffers(rt_ctx* c) {
    rt_ctx::PathBuffers& o = c->alt;
    std::swap(c->d_arena, o.d_arena); std::swap(c->d_seg_count, o.d_seg_count); std::swap(c->hits, o.hits); std::swap(c->ps, o.ps);
    std::swap(c->gq, o.gq); std::swap(c->d_stack_spill, o.d_stack_spill);
    for (int k = 0; k < 2; k++) { std::swap(c->q[k], o.q[k]); std::swap(c->st[k], o.st[k]); }
}

// float arrays of one set of path buffers: 2 queues x (6 + path), hits, path state, slot state x 2, gather queue, stack spill
constexpr int PATH_ARRAYS = 14 + 5 + 15 + 14 + 9 + 48;
size_t path_stride(u32 segs) { return ((size_t) segs * 1024 + 63) & ~(size_t) 63; }
size_t seg_count_words(u32 segs) { return 4 * (size_t) segs + 1; } // 2 queues, gather, gather offsets

// points the context's current set of path buffers into `arena` (PATH_ARRAYS * path_stride(segs) floats) and
// `seg_count` (seg_count_words(segs) words)
void carve_path_buffers(rt_ctx* c, float* arena, u32* seg_count, u32 segs) {
    const size_t stride = path_stride(segs);
    c->d_arena = arena; c->d_seg_count = seg_count;
    float* p = arena;
    auto take = [&]() { float* r = p; p += stride; return r; };
    for (int k = 0; k < 2; k++) {
        c->q[k].ox = take(); c->q[k].oy = take(); c->q[k].oz = take();
        c->q[k].dx = take(); c->q[k].dy = take(); c->q[k].dz = take(); c->q[k].path = (u32*) take();
    }
    c->hits.t = take(); c->hits.u = take(); c->hits.v = take(); c->hits.inst = (int*) take(); c->hits.tri = (int*) take();
    c->ps.rng = (u32*) take();
    c->ps.thr_x = take(); c->ps.thr_y = take(); c->ps.thr_z = take();
    c->ps.rad_x = take(); c->ps.rad_y = take(); c->ps.rad_z = take(); c->ps.alpha = take();
    c->ps.fd_x = take(); c->ps.fd_y = take(); c->ps.fd_z = take();
    c->ps.sum_x = take(); c->ps.sum_y = take(); c->ps.sum_z = take(); c->ps.sum_a = take(); // consecutive: k_shade / k_frame_resolve also use them as one float4 array
    for (int k = 0; k < 2; k++) {
        c->st[k].rng = (u32*) take(); c->st[k].thr_x = take(); c->st[k].thr_y = take(); c->st[k].thr_z = take();
        c->st[k].rad_x = take(); c->st[k].rad_y = take(); c->st[k].rad_z = take();
    }
    c->gq.path = (u32*) take(); c->gq.hx = take(); c->gq.hy = take(); c->gq.hz = take(); c->gq.tf = take();
    c->gq.kx = take(); c->gq.ky = take(); c->gq.kz = take(); c->gq.ti = (int*) take();
    c->d_stack_spill = (int*) p; p += 48 * stride;
    c->n_paths = (u32) stride; c->n_segments_cap = segs;
}

// allocates the current set of path buffers for `segs` segments of 1024 queue slots
int alloc_path_buffers(rt_ctx* c, u32 segs) {
    u32* seg_count = nullptr; float* arena = nullptr;
    RT_HIP(hipMalloc(&seg_count, sizeof(u32) * seg_count_words(segs)));
    c->d_seg_count = seg_count; // owned from here on (freed with the context even if the next allocation fails)
    RT_HIP(hipMemset(seg_count, 0, sizeof(u32) * seg_count_words(segs)));
    RT_HIP(hipMalloc(&arena, path_stride(segs) * 4 * PATH_ARRAYS));
    carve_path_buffers(c, arena, seg_count, segs);
    return RT_OK;
}

// callers have quiesced the context: nothing is in flight on either lane
int ensure_path_buffers(rt_ctx* c) {
    u32 tiles = ((c->width + 31) / 32) * ((c->height + 31) / 32);
    u32 segs = std::max(tiles, (c->n_points + 1023) / 1024);
    if (segs == 0) segs = 1;
    if ((uint64_t) segs * 1024 <= c->n_paths) return RT_OK;
    dev_free(c->d_arena); dev_free(c->d_seg_count);
    dev_free(c->alt.d_arena); dev_free(c->alt.d_seg_count); // lane 1's set is allocated again, at the new size, when a dispatch first needs it
    memset(&c->alt, 0, sizeof c->alt);
    c->n_paths = 0;
    return alloc_path_buffers(c, segs);
}

// lane 1 of the frame pipeline: its stream, its buffers, the two end-of-frame events
int ensure_second_lane(rt_ctx* c) {
    if (!c->alt_stream) RT_HIP(hipStreamCreateWithFlags(&c->alt_stream, hipStreamNonBlocking));
    for (int k = 0; k < 2; k++) if (!c->lane_done[k]) RT_HIP(hipEventCreateWithFlags(&c->lane_done[k], hipEventDisableTiming));
    for (int k = 0; k < 2; k++) if (!c->probe_done[k]) RT_HIP(hipEventCreateWithFlags(&c->probe_done[k], hipEventDisableTiming));
    if (c->n_points && !c->d_sp_read_alt) {
        RT_HIP(hipMalloc(&c->d_sp_read_alt, sizeof(RtSamplePoint) * c->n_points));
        RT_HIP(hipMalloc(&c->d_sp_pay_alt, sizeof(float4) * c->n_points));
        c->sp_read_version[1] = ~0ull; // filled from d_sp_write by the first frame that gathers from it
    }
    if (c->alt.d_arena) return RT_OK;
    swap_path_buffers(c);
    int rc = alloc_path_buffers(c, c->n_segments_cap);
    swap_path_buffers(c);
    return rc;
}

// the path buffers of the sample-point pass on its own stream (rt_ctx::probe_side) <-> the context's current set
void swap_probe_buffers(rt_ctx* c) {
    rt_ctx::PathBuffers& o = c->probe_pb;
    std::swap(c->d_arena, o.d_arena); std::swap(c->d_seg_count, o.d_seg_count); std::swap(c->hits, o.hits); std::swap(c->ps, o.ps);
    std::swap(c->gq, o.gq); std::swap(c->d_stack_spill, o.d_stack_spill);
    for (int k = 0; k < 2; k++) { std::swap(c->q[k], o.q[k]); std::swap(c->st[k], o.st[k]); }
    std::swap(c->n_paths, c->probe_paths); std::swap(c->n_segments_cap, c->probe_cap);
}
int ensure_probe_side(rt_ctx* c) {
    if (!c->probe_stream) RT_HIP(hipStreamCreateWithFlags(&c->probe_stream, hipStreamNonBlocking));
    const u32 segs = std::max((c->n_points + 1023u) / 1024u, 1u);
    if (c->probe_pb.d_arena && c->probe_cap >= segs) return RT_OK;
    RT_HIP(hipStreamSynchronize(c->probe_stream));
    dev_free(c->probe_pb.d_arena); dev_free(c->probe_pb.d_seg_count); memset(&c->probe_pb, 0, sizeof c->probe_pb); c->probe_paths = 0; c->probe_cap = 0;
    swap_probe_buffers(c);
    int rc = alloc_path_buffers(c, segs);
    swap_probe_buffers(c);
    return rc;
}

// every entry point other than a pipelined dispatch starts here: lane 1 drained, so that work on `stream` is ordered
// after everything before it as it would be with one stream
int quiesce(rt_ctx* c) {
    if (c->pipelined) {
        c->pipelined = false;
        RT_HIP(hipStreamSynchronize(c->alt_stream));
        RT_HIP(hipStreamSynchronize(c->own_stream));
        if (c->probe_stream) RT_HIP(hipStreamSynchronize(c->probe_stream));
        c->probe_wait = nullptr;
    }
    return RT_OK;
}

void free_sample_points(rt_ctx* c) {
    dev_free(c->d_sp_read); dev_free(c->d_sp_write); dev_free(c->d_sp_normal); dev_free(c->d_sp_slice);
    dev_free(c->d_sp_read_alt); dev_free(c->d_sp_pay); dev_free(c->d_sp_pay_alt); c->sp_version = 0; c->sp_read_version[0] = 0; c->sp_read_version[1] = ~0ull;
    dev_free(c->d_slices); dev_free(c->d_slot_to_ti);
    dev_free(c->d_hg_nodes); dev_free(c->d_hg_sums); dev_free(c->d_hg_pos); dev_free(c->d_hg_pay); dev_free(c->d_hg_order); dev_free(c->d_hg_slices);
    c->hg_built = false;
    c->n_points = 0; c->slot_count = 0; c->slices.clear();
}

void fill_pass_args(rt_ctx* c, PassArgs& a, bool probe) {
    memset(&a, 0, sizeof a);
    a.g = c->g;
    a.ray_sort = c->ray_sort ? 1u : 0u;
    a.wave_major = c->wave_major ? 1u : 0u;
    a.xcd_bands = c->xcd_bands ? 1u : 0u;
    a.sort_box_on = (c->sort_box && c->flat_instances != 0u) ? 1u : 0u; // flat scenes: the box of the instance with the most triangles (of the only one: the scene's)
    memcpy(a.sort_box, c->flat_box[c->flat_big], sizeof a.sort_box);
    a.sort_chord = 0.0f;
    if (c->sort_box && c->sort_chord_pct > 0.0f && c->flat_instances == 0u && c->n_instances > 0u) {
        // two levels: the box of the top level (the instances but the inlined one); a ray with a long way through it meets many instances
        float diag = 0.0f;
        for (int k = 0; k < 3; k++) {
            const float h = 1.0f / c->tlas_info.inv_h[k];
            a.sort_box[k] = c->tlas_info.ctr[k] - h; a.sort_box[3 + k] = c->tlas_info.ctr[k] + h; diag += 4.0f * h * h;
        }
        a.sort_chord = 0.01f * c->sort_chord_pct * sqrtf(diag);
        a.sort_box_on = 2u;
    }
    // rows 0..16 of the per-lane LDS stack of the 6-wave kernels, or rows 0..32 of the 5-wave "deep" kernels (rt_kernels.hip)
    a.stack_in_lds = !c->stack_lds_ok ? 0u : c->tlas_depth + c->max_blas_depth + 2 <= 16 ? 1u : (c->stack_deep_ok && c->tlas_depth + c->max_blas_depth + 2 <= 32) ? 2u : 0u;
    a.task_counter = c->d_task_counter; a.persistent_slots = c->persistent_slots;
    // persistent traversal waves pay ~2.5 us per 64-ray task and win back the dispatcher's idle slots; since the dispatched
    // kernels run 6 waves per SIMD in wave-major order that no longer pays on any benchmark scene (k_traverse, rt_kernels.hip)
    a.persistent = c->persistent > 0 ? (u32) c->persistent : 0u;
    a.pool_groups = c->pool > 0 ? (u32) c->pool * c->cu_count : 0u;
    a.flat_rows = c->flat_instances ? (u32) c->flat_info.depth + 2u : 0u;
    // one kernel for bounce 0: flat trees of ONE instance (with or without the small one tested first), shallow enough for two
    // 1,024-thread workgroups per CU (18 rows of 4 KB + the ranking's 4 KB each); the counting and capturing variants keep the two kernels
    a.fuse_primary = (c->fuse_primary && !probe && c->flat_instances == 1u && a.flat_rows <= 18u && c->persistent <= 0 &&
                      !(c->flags & (RT_FLAG_COUNT_TRAVERSAL))) ? 1u : 0u;
    a.dbg_blas_lo = ~0ull; a.dbg_blas_hi = 0ull;
    for (const Blas& b : c->blas) {
        const u32 nn = b.n_tris > 1 ? b.n_tris - 1 : 1;
        const unsigned long long lo = (unsigned long long) b.d_nodes, hi = lo + nn * sizeof(PackedNode) + 3ull * b.n_tris * sizeof(TriRec);
        a.dbg_blas_lo = std::min(a.dbg_blas_lo, lo); a.dbg_blas_hi = std::max(a.dbg_blas_hi, hi);
    }
    { const u32 np = c->n_instances - (c->inl_inst >= 0 ? 1u : 0u); a.dbg_tlas_bytes = (np > 1 ? np - 1 : 1) * (u32) sizeof(PackedNode); }
    a.st[0] = c->st[0]; a.st[1] = c->st[1];
    for (int k = 0; k < 3; k++) a.camera_origin[k] = c->g.camera_to_world[12 + k] / c->g.camera_to_world[15];
    a.fudge = (c->flags & RT_FLAG_ENABLE_SSS) ? c->g.translucent_bssrdf_fudge : 0.0f;
    a.width = c->width; a.height = c->height;
    a.inv_size[0] = 1.0f / (float) c->width; a.inv_size[1] = 1.0f / (float) c->height;
    a.tile_rank = c->tile_rank; a.tile_world = c->tile_world;
    a.tiles_x = (c->width + 31) / 32; a.tiles_y = (c->height + 31) / 32;
    u32 n_tiles = a.tiles_x * a.tiles_y;
    a.tiles_x_magic = (a.tiles_x > 1 && (uint64_t) n_tiles * a.tiles_x < (1ull << 32)) ? (u32) ((1ull << 32) / a.tiles_x) + 1u : 0u;
    a.n_local_tiles = n_tiles > c->tile_rank ? (n_tiles - c->tile_rank + c->tile_world - 1) / c->tile_world : 0;
    a.n_paths = c->n_paths;
    a.probe = probe ? 1u : 0u;
    a.tmin = probe ? 0.0001f : 0.000001f; // raytracing.hlsl:249 / :133
    a.tmax = 10000.0f;
    a.scene.tlas_nodes = c->d_tlas_nodes; a.scene.insts = c->d_inst_rec;
    for (int k = 0; k < 3; k++) { a.scene.tlas_ctr[k] = c->tlas_info.ctr[k]; a.scene.tlas_inv_h[k] = c->tlas_info.inv_h[k]; }
    a.scene.tlas_bound = c->tlas_info.bound;
    a.scene.flat_inl_inst = c->flat_instances ? c->flat_inl_inst : -1; a.scene.flat_inst0 = c->flat_inst0;
    a.scene.inl_inst = c->inl_inst; a.scene.inl_count = c->inl_count; a.scene.n_insts = c->n_instances; a.scene.inl_tris = c->inl_tris; a.scene.inl_copy_bytes = c->inl_copy_bytes;
    memcpy(a.scene.inl_w2o, c->inl_w2o, sizeof c->inl_w2o);
    memcpy(a.scene.inl_box, c->inl_box, sizeof c->inl_box); a.scene.inl_box_same = c->inl_box_same;
    for (int k = 0; k < 3; k++) { const float* m = c->inl_w2o; const float* p = a.camera_origin; a.scene.inl_cam_o[k] = p[0] * m[k] + p[1] * m[3 + k] + p[2] * m[6 + k] + m[9 + k]; } // xform_point43, operation for operation
    a.scene.flat_instances = c->flat_instances;
    a.scene.grid_cells = (c->flat_instances || c->persistent > 0) ? 0u : c->grid_cells; a.scene.grid_n_entries = c->grid_n_entries;
    a.scene.grid_start = c->d_grid_start; a.scene.grid_entries = c->d_grid_entries;
    for (int k = 0; k < 3; k++) { a.scene.grid_res[k] = c->grid_res[k]; a.scene.grid_lo[k] = c->grid_lo[k]; a.scene.grid_cell[k] = c->grid_cell[k]; a.scene.grid_inv_cell[k] = c->grid_cell[k] > 0.0f ? 1.0f / c->grid_cell[k] : 0.0f; }
    a.scene.grid_slack = c->grid_slack;
    a.flat_waves8 = (c->flat_waves8 && c->flat_instances == 1 && c->flat_info.depth + 2 <= 20) ? 1u : 0u;
    a.flat_waves7 = (c->flat_waves7 && c->flat_instances == 2 && c->flat_info.depth + 2 <= 22) ? 1u : 0u; // the two-instance kernels fit 72 VGPRs since the ray preparation moved behind the root step
    if (c->flat_instances) {
        a.scene.flat_nodes = c->d_flat_nodes; a.scene.flat_tris = c->d_flat_tris; a.scene.flat_copy_bytes = (u32) (c->flat_tris * sizeof(FlatTri));
        for (int k = 0; k < 3; k++) { a.scene.flat_ctr[k] = c->flat_info.ctr[k]; a.scene.flat_inv_h[k] = c->flat_info.inv_h[k]; }
        a.scene.flat_bound = c->flat_info.bound;
        a.scene.flat_first1 = c->flat_first1;
        memcpy(a.scene.flat_w2o, c->flat_w2o, sizeof c->flat_w2o);
        for (u32 i = 0; i < c->flat_instances; i++) { // xform_point43 (rt_dev_math.h) of the camera origin, operation for operation
            const float* m = c->flat_w2o[i]; const float* p = a.camera_origin;
            for (int k = 0; k < 3; k++) a.scene.flat_cam_o[i][k] = p[0] * m[k] + p[1] * m[3 + k] + p[2] * m[6 + k] + m[9 + k];
        }
    }
    a.shade.insts = c->d_inst_shade; a.shade.slot_to_ti = c->d_slot_to_ti; a.shade.slot_count = c->slot_count;
    a.shade.slices = c->d_slices;
    a.q[0] = c->q[0]; a.q[1] = c->q[1]; a.hits = c->hits; a.ps = c->ps; a.gq = c->gq;
    a.n_segments = probe ? (c->n_points + 1023) / 1024 : a.n_local_tiles;
    a.gather_enabled = (!probe && c->n_points > 0 && a.fudge != 0.0f) ? 1u : 0u;
    a.seg_count[0] = c->d_seg_count; a.seg_count[1] = c->d_seg_count + c->n_segments_cap;
    a.gseg_count = c->d_seg_count + 2 * (size_t) c->n_segments_cap;
    a.goff = c->d_seg_count + 3 * (size_t) c->n_segments_cap;
    a.stack_spill = c->d_stack_spill;
    a.sp_pay = c->d_sp_pay;
    a.stats = c->d_stats;
    a.capture = nullptr;
    a.accumulator = c->d_accumulator; a.render_target = c->d_render_target;
    a.sp_read = c->d_sp_read; a.sp_write = c->d_sp_write; a.sp_normal = c->d_sp_normal; a.sp_slice = c->d_sp_slice;
    a.n_points = c->n_points;
    { const u32 block = (c->n_points + c->probe_world - 1) / c->probe_world; // equal blocks, the last one short (or empty)
      a.probe_lo = std::min(c->probe_rank * block, c->n_points); a.probe_hi = std::min(a.probe_lo + block, c->n_points); }
    a.hg.nodes = c->d_hg_nodes; a.hg.sums = c->d_hg_sums; a.hg.pos = c->d_hg_pos; a.hg.pay = c->d_hg_pay;
    a.hg.order = c->d_hg_order; a.hg.slices = c->d_hg_slices; a.hg.max_solid_angle = c->hg_max_solid_angle;
    a.hg.enabled = (c->hg_enabled && c->hg_built) ? 1u : 0u;
    a.bssrdf_table = c->d_table[c->profile]; a.bssrdf_texels = c->table_texels[c->profile];
    a.gather_delta = (c->gather_l1 && c->d_gather_delta && c->delta_table == a.bssrdf_table && c->delta_texels == a.bssrdf_texels &&
                      c->delta_factor == a.fudge / (c->g.translucent_bssrdf_scale * c->g.translucent_bssrdf_scale)) ? c->d_gather_delta : nullptr;
}

// pixels of this rank's tiles that lie inside the image
uint64_t local_pixel_count(const PassArgs& a) {
    uint64_t n = 0;
    for (u32 k = 0; k < a.n_local_tiles; k++) {
        u32 tile = k * a.tile_world + a.tile_rank, tx = tile % a.tiles_x, ty = tile / a.tiles_x;
        u32 w = std::min(32u, a.width - tx * 32u), h = std::min(32u, a.height - ty * 32u);
        n += (uint64_t) w * h;
    }
    return n;
}

bool timing_now(const rt_ctx* c) {
    return (c->flags & RT_FLAG_TIME_KERNELS) != 0 && (!(c->flags & RT_FLAG_TIME_SAMPLED) || c->dispatch_index % 8u == 0u);
}

int timed_traverse(rt_ctx* c, const PassArgs& a, u32 inq, bool primary, u32 sample_index, bool tail_shade = false) {
    bool timing = timing_now(c);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (timing) {
        RT_HIP(hipEventCreate(&e0)); RT_HIP(hipEventCreate(&e1));
        RT_HIP(hipEventRecord(e0, c->stream));
        c->traverse_launches_timed++;
    }
    int rc = launch_traverse(c->stream, a, inq, (c->flags & RT_FLAG_COUNT_TRAVERSAL) != 0, primary, sample_index, tail_shade);
    if (timing) { RT_HIP(hipEventRecord(e1, c->stream)); (primary ? c->timed_primary : c->timed).emplace_back(e0, e1); if (primary) c->primary_launches_timed++; }
    c->traverse_launches++;
    if (primary) { c->primary_launches++; c->primary_rays += local_pixel_count(a); } // bounce-0 rays are not queued, so the device does not count them
    return rc;
}

// bounce 0 as one kernel (PassArgs::fuse_primary): counted and timed as the frame's primary traversal launch (its duration
// now includes the bounce-0 shade)
int timed_primary_fused(rt_ctx* c, const PassArgs& a, u32 sample_index) {
    bool timing = timing_now(c);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (timing) {
        RT_HIP(hipEventCreate(&e0)); RT_HIP(hipEventCreate(&e1));
        RT_HIP(hipEventRecord(e0, c->stream));
        c->traverse_launches_timed++; c->primary_launches_timed++;
    }
    int rc = launch_shade(c->stream, a, 0u, 0u, sample_index, true, true);
    if (timing) { RT_HIP(hipEventRecord(e1, c->stream)); c->timed_primary.emplace_back(e0, e1); }
    c->traverse_launches++; c->primary_launches++; c->primary_rays += local_pixel_count(a);
    return rc;
}

// a queued bounce as one kernel (walk + shade): counted and timed as that bounce's traversal launch
int timed_queued_fused(rt_ctx* c, const PassArgs& a, u32 inq, u32 bounce, u32 sample_index) {
    bool timing = timing_now(c);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (timing) {
        RT_HIP(hipEventCreate(&e0)); RT_HIP(hipEventCreate(&e1));
        RT_HIP(hipEventRecord(e0, c->stream));
        c->traverse_launches_timed++;
    }
    int rc = launch_shade(c->stream, a, inq, bounce, sample_index, false, true, true);
    if (timing) { RT_HIP(hipEventRecord(e1, c->stream)); c->timed.emplace_back(e0, e1); }
    c->traverse_launches++;
    return rc;
}

int timed_gather(rt_ctx* c, const PassArgs& a, hipStream_t stream, bool batched = false) {
    bool timing = timing_now(c);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (timing) {
        RT_HIP(hipEventCreate(&e0)); RT_HIP(hipEventCreate(&e1));
        RT_HIP(hipEventRecord(e0, stream));
        c->gather_launches_timed++;
    }
    int rc = batched ? launch_gather_batched(stream, a) : launch_gather(stream, a);
    if (timing) { RT_HIP(hipEventRecord(e1, stream)); c->timed_gather.emplace_back(e0, e1); }
    c->gather_launches++;
    return rc;
}

// The gather of bounce b reads what the shade of bounce b queued and adds into the radiance of the paths; the traversal of
// bounce b + 1 touches neither.  With a side stream at hand (the context's own streams: ensure_gather_side) the gather runs
// there, beside that traversal, and the next kernel that reads or writes the paths' radiance or the gather queue -- the shade
// of bounce b + 1, the next sample's bounce-0 shade, the resolve -- waits for it (gather_join).  Same kernels, same operands,
// same order of the additions into every path: same bits.
int gather_fork(rt_ctx* c, const PassArgs& a) {
    hipStream_t side = c->gather_side && (c->stream == c->own_stream || c->stream == c->alt_stream) ? c->gather_stream[c->cur_lane] : nullptr;
    if (!side) return timed_gather(c, a, c->stream);
    RT_HIP(hipEventRecord(c->gather_ev[c->cur_lane][0], c->stream));
    RT_HIP(hipStreamWaitEvent(side, c->gather_ev[c->cur_lane][0], 0));
    int rc = timed_gather(c, a, side); if (rc) return rc;
    RT_HIP(hipEventRecord(c->gather_ev[c->cur_lane][1], side));
    c->gather_pending = true;
    return RT_OK;
}
int gather_join(rt_ctx* c) {
    if (!c->gather_pending) return RT_OK;
    c->gather_pending = false;
    RT_HIP(hipStreamWaitEvent(c->stream, c->gather_ev[c->cur_lane][1], 0));
    return RT_OK;
}
int ensure_gather_side(rt_ctx* c, int lanes) {
    for (int l = 0; l < lanes; l++) {
        if (!c->gather_stream[l]) RT_HIP(hipStreamCreateWithFlags(&c->gather_stream[l], hipStreamNonBlocking));
        for (int k = 0; k < 2; k++) if (!c->gather_ev[l][k]) RT_HIP(hipEventCreateWithFlags(&c->gather_ev[l][k], hipEventDisableTiming));
    }
    return RT_OK;
}

int drain_timers(rt_ctx* c) {
    for (auto& pr : c->timed_gather) {
        RT_HIP(hipEventSynchronize(pr.second));
        float ms = 0; RT_HIP(hipEventElapsedTime(&ms, pr.first, pr.second));
        c->gather_ms += ms;
        (void) hipEventDestroy(pr.first); (void) hipEventDestroy(pr.second);
    }
    c->timed_gather.clear();
    for (auto& pr : c->timed) {
        RT_HIP(hipEventSynchronize(pr.second));
        float ms = 0; RT_HIP(hipEventElapsedTime(&ms, pr.first, pr.second));
        c->traverse_ms += ms;
        (void) hipEventDestroy(pr.first); (void) hipEventDestroy(pr.second);
    }
    c->timed.clear();
    for (auto& pr : c->timed_primary) {
        RT_HIP(hipEventSynchronize(pr.second));
        float ms = 0; RT_HIP(hipEventElapsedTime(&ms, pr.first, pr.second));
        c->traverse_ms += ms; c->traverse_primary_ms += ms;
        (void) hipEventDestroy(pr.first); (void) hipEventDestroy(pr.second);
    }
    c->timed_primary.clear();
    return RT_OK;
}

// Buffers of the batched gather for the frame about to run on lane `lane` (c->stream is that lane's stream): as many rows as
// the frame has gathering shade launches, as far as the budget goes.  *out = nullptr: this frame gathers launch by launch
// (fewer than two rows, the hierarchical gather, RT_GATHER_BATCH=0, rt_dispatch_rays_static -- which must not allocate).
int ensure_batch_gather(rt_ctx* c, int lane, rt_ctx::BatchGather** out) {
    *out = nullptr;
    if (!c->gather_batch || c->no_alloc_call || c->g.translucent_emission_bounces < 0 || c->n_paths == 0) return RT_OK;
    const uint64_t per_sample = (uint64_t) std::min<int64_t>(c->g.translucent_emission_bounces, c->g.bounces_per_sample) + 1u;
    const uint64_t want = per_sample * c->g.samples_per_pixel;
    const uint64_t row_bytes = (uint64_t) 9 * 4 * c->n_paths;
    const uint64_t rows = std::min<uint64_t>(std::min<uint64_t>(want, (uint64_t) c->gather_batch_mb * (1ull << 20) / row_bytes), (uint64_t) 0xffffffffu / c->n_paths);
    if (rows < 2 || rows < per_sample) return RT_OK; // (the rows of a sample are flushed together)
    rt_ctx::BatchGather& b = c->bg[lane];
    if (b.rows < rows || b.stride != c->n_paths || b.cols_cap != c->n_segments_cap) {
        RT_HIP(hipStreamSynchronize(c->stream)); // the lane's last frame may still read the old buffers
        dev_free(b.d_arena); dev_free(b.d_counts); dev_free(b.d_off); b.rows = 0;
        if (hipMalloc(&b.d_arena, rows * row_bytes) != hipSuccess || hipMalloc(&b.d_counts, sizeof(u32) * rows * c->n_segments_cap) != hipSuccess ||
            hipMalloc(&b.d_off, sizeof(u32) * (rows * c->n_segments_cap + 2)) != hipSuccess) {
            // no room for the rows (they are an optimisation, not a need): this context gathers launch by launch from here on
            (void) hipGetLastError();
            dev_free(b.d_arena); dev_free(b.d_counts); dev_free(b.d_off);
            c->gather_batch = 0;
            return RT_OK;
        }
        b.rows = (u32) rows; b.stride = c->n_paths; b.cols_cap = c->n_segments_cap;
    }
    *out = &b;
    return RT_OK;
}

// the loop of trace_path_sample (raytracing.hlsl:101-120) over all paths of a pass, wavefront style
int run_pass(rt_ctx* c, bool probe) {
    PassArgs a; fill_pass_args(c, a, probe);
    bool capture = !probe && (c->flags & RT_FLAG_CAPTURE_HITS);
    if (capture) {
        size_t need = (size_t) (c->g.bounces_per_sample + 1) * c->width * c->height;
        if (need > c->capture_records) {
            dev_free(c->d_capture);
            RT_HIP(hipMalloc(&c->d_capture, need * sizeof(RtHit)));
            c->capture_records = need;
        }
        c->capture_bounces = c->g.bounces_per_sample + 1;
        a.capture = c->d_capture;
        int rc = launch_fill_capture(c->stream, a); if (rc) return rc;
    }
    bool gather = a.gather_enabled != 0;
    // Batched gather: the shade launches that queue gather items write them to a row of their own; one k_gather launch evaluates
    // the rows of the whole frame (or of as many samples as there are rows for) and k_gather_apply adds the terms to the paths'
    // radiance / the pixels' sample sums in the order of the rows.  The terms are the ones the launch-by-launch gather adds; a
    // path's radiance sums them after, instead of between, the emission of its later bounces (radiance only: the 1e-4 contract;
    // measured 1e-7).
    rt_ctx::BatchGather* bg = nullptr;
    if (gather && !a.hg.enabled) { int brc = ensure_batch_gather(c, c->cur_lane, &bg); if (brc) return brc; }
    const u32 rows_per_sample = (gather && c->g.translucent_emission_bounces >= 0) ? (u32) std::min<int64_t>(c->g.translucent_emission_bounces, c->g.bounces_per_sample) + 1u : 0u;
    // camera pass: no ray-generation kernel (bounce 0 derives rays from pixels), and unless a gather launch adds emission to the
    // paths' radiance between the bounces, no resolve kernel either: finished paths fold their sample into the pixel's sum in
    // k_shade (`fused`).  With the batched gather that holds for frames with translucent meshes too: the terms are added to the
    // sample sums after the last bounce, and the frame's resolve kernel adds the sums to the accumulator.
    const bool fused_gather = gather && bg && c->gather_fused && !capture && !(c->flags & RT_FLAG_COUNT_TRAVERSAL) && c->persistent <= 0;
    bool fused = !probe && (!gather || fused_gather);
    a.defer_resolve = ((c->defer_resolve || fused_gather) && fused) ? 1u : 0u;
    // (k_frame_resolve skips the pixels whose sample adds nothing, so it cannot keep the 8-bit target current: derived when it is read)
    a.lazy_render_target = (c->defer_resolve || fused_gather) ? 1u : 0u;
    if (!probe) c->last_pass_lazy = a.lazy_render_target != 0u;
    PassArgs ab = a; u32 bg_row = 0;
    auto gq_row = [&](PassArgs& x, u32 row) { // the gather queue of x: row `row` of the batch buffers
        const size_t rs = (size_t) bg->rows * bg->stride; float* p = bg->d_arena + (size_t) row * bg->stride;
        x.gq.path = (u32*) p; x.gq.hx = p + rs; x.gq.hy = p + 2 * rs; x.gq.hz = p + 3 * rs; x.gq.tf = p + 4 * rs;
        x.gq.kx = p + 5 * rs; x.gq.ky = p + 6 * rs; x.gq.kz = p + 7 * rs; x.gq.ti = (int*) (p + 8 * rs);
        x.gseg_count = bg->d_counts + (size_t) row * a.n_segments;
    };
    // the frame's sample-point pass on its own stream (rt_dispatch_rays): what reads the points waits for it
    auto wait_probe = [&]() -> int {
        if (!c->probe_wait) return RT_OK;
        hipEvent_t e = c->probe_wait; c->probe_wait = nullptr;
        RT_HIP(hipStreamWaitEvent(c->stream, e, 0));
        return RT_OK;
    };
    if (!probe && gather && !bg) { int wrc = wait_probe(); if (wrc) return wrc; } // the gather launch of the first bounce already reads them
    auto flush_rows = [&]() -> int {
        if (!bg || !bg_row) return RT_OK;
        { int wrc = wait_probe(); if (wrc) return wrc; }
        ab.bg_rows = bg_row; bg_row = 0;
        return timed_gather(c, ab, c->stream, true);
    };
    if (bg) {
        gq_row(ab, 0);
        ab.bg_cols = a.n_segments; ab.bg_stride = bg->stride; ab.bg_counts = bg->d_counts; ab.goff = bg->d_off;
        ab.bg_to_sums = fused_gather ? 1u : 0u;
        ab.bg_block = (u32) c->gather_block; ab.bg_groups = (u32) c->gather_groups * c->cu_count; // two 256-thread workgroups per CU: 2 waves per SIMD.  The LDS (35 KB each) would allow four, and alone on the GPU the launch is fastest with four; beside the other frame's short, latency-bound launches two leave them the room they need (C4 8.55 against 8.1 Grays/s, profiles/r03_ab_gather_batch.txt)
    }
    for (u32 s = 0; s < c->g.samples_per_pixel; s++) {
        int rc = RT_OK;
        if (probe) { rc = launch_probe_raygen(c->stream, a, s); if (rc) return rc; }
        // (the rows of a sample are flushed together, between samples: every path of the samples before has ended, so the pixels' sums exist)
        if (bg && bg_row + rows_per_sample > bg->rows) { rc = flush_rows(); if (rc) return rc; }
        for (u32 b = 0; b <= c->g.bounces_per_sample; b++) {
            bool primary = !probe && b == 0;
            const bool gathers = gather && (int) b <= c->g.translucent_emission_bounces; // the translucent hits of this bounce emit (raytracing.hlsl:321)
            PassArgs as_row; const PassArgs* as = &a;
            if (bg && gathers) { as_row = a; gq_row(as_row, bg_row++); as = &as_row; } // this launch queues gather items: into the next row
            if (primary && fused && a.fuse_primary) { rc = timed_primary_fused(c, *as, s); if (rc) return rc; } // walk + shade in one kernel
            else if (gather && !fused) { // (the gather of the bounce before may still run on its side stream: the traversal does not care, the shade waits)
                rc = timed_traverse(c, a, b & 1u, primary, s); if (rc) return rc;
                if (!bg) { rc = gather_join(c); if (rc) return rc; }
                rc = launch_shade(c->stream, *as, b & 1u, b, s, primary, fused); if (rc) return rc;
            }
            else {
                // the last bounce of the fused pass ends every path: its traversal kernel shades the hits itself (k_traverse, `tail`)
                const bool tail_shade = c->fuse_tail && fused && !primary && !capture && b == c->g.bounces_per_sample && a.pool_groups == 0u && c->flat_instances != 2u &&
                                        !(c->flags & RT_FLAG_COUNT_TRAVERSAL) && c->persistent <= 0 && !gathers;
                // the queued bounces in between, in flat one-instance scenes: one kernel that walks and shades, like bounce 0 (RT_FUSE_QUEUED)
                const bool walk_and_shade = (int) b <= c->fuse_queued && fused && !primary && !tail_shade && a.fuse_primary && !capture;
                if (walk_and_shade) { rc = timed_queued_fused(c, *as, b & 1u, b, s); if (rc) return rc; }
                else {
                    rc = timed_traverse(c, a, b & 1u, primary, s, tail_shade); if (rc) return rc;
                    if (!tail_shade) { rc = launch_shade(c->stream, *as, b & 1u, b, s, primary, fused); if (rc) return rc; }
                }
            }
            if (gathers && !bg) { rc = gather_fork(c, a); if (rc) return rc; }
        }
    }
    { int jrc = gather_join(c); if (jrc) return jrc; } // (a sample whose last bounce gathers: nothing after it has waited yet)
    { int frc = flush_rows(); if (frc) return frc; }
    if (probe) return launch_probe_resolve(c->stream, a);
    if (fused && !a.defer_resolve) return RT_OK;
    // pipelined frames: pixels are added to in frame order, by the frame's resolve kernel alone; all the tracing, shading
    // and gathering above overlaps the frame before
    if (c->accumulate_after) RT_HIP(hipStreamWaitEvent(c->stream, c->accumulate_after, 0));
    return fused ? launch_frame_resolve(c->stream, a) : launch_camera_resolve(c->stream, a);
}

// the hierarchical gather dropped part of a walk (DeviceStats::hg_overflow): results since the last check are wrong
int check_gather_overflow(rt_ctx* c) {
    if (!(c->hg_enabled && c->hg_built)) return RT_OK;
    unsigned long long flag = 0;
    RT_HIP(hipMemcpy(&flag, &c->d_stats->hg_overflow, sizeof flag, hipMemcpyDeviceToHost));
    if (!flag) return RT_OK;
    RT_HIP(hipMemset(&c->d_stats->hg_overflow, 0, sizeof flag));
    set_error("hierarchical gather: a walk overflowed its %d-entry stack; the frames since the last check are incomplete", 24);
    return RT_ERR_LIMIT;
}

float triangle_area(const float* a, const float* b, const float* c) { // geometry.cpp:21-25
    float e1[3] = { b[0] - a[0], b[1] - a[1], b[2] - a[2] }, e2[3] = { c[0] - a[0], c[1] - a[1], c[2] - a[2] };
    float cr[3] = { e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0] };
    return 0.5f * sqrtf(cr[0] * cr[0] + cr[1] * cr[1] + cr[2] * cr[2]);
}

// phase order: Fisher-Yates with xorshift32 in place of rand() (bluenoise.cpp:306-308)
void phase_sequence(u32 seed, u32 count, u32* seq) {
    u32 state = seed;
    state = (state ^ 61u) ^ (state >> 16); state *= 9u; state = state ^ (state >> 4); state *= 0x27d4eb2du; state = state ^ (state >> 15);
    state |= 1u;
    for (u32 i = 0; i < count; i++) seq[i] = i;
    for (u32 i = 0; i < count; i++) {
        state ^= state << 13; state ^= state >> 17; state ^= state << 5;
        u32 r = state % (count - i);
        std::swap(seq[r], seq[count - 1 - i]);
    }
}


void free_blas(Blas& b) {
    dev_free(b.d_verts); dev_free(b.d_indices); dev_free(b.d_geoms); dev_free(b.d_tri_geometry);
    dev_free(b.d_nodes); b.d_tris = nullptr; dev_free(b.d_perm); dev_free(b.d_inv_perm);
}

// world boxes of the (at most RT_INLINE_MAX_TRIS) triangles of an inlined instance, in TriRec order, for SceneView::inl_box:
// grown by 1e-4 of the largest coordinate, far beyond the rounding of the slab test that uses them and of the
// object-space triangle test behind it
int set_inline_boxes(rt_ctx* c, const Blas& b, const float* o2w) {
    TriRec t[RT_INLINE_MAX_TRIS];
    const u32 k = b.n_tris;
    if (k == 0 || k > RT_INLINE_MAX_TRIS) { set_error("set_inline_boxes: %u triangles", k); return RT_ERR_INVALID; }
    RT_HIP(hipStreamSynchronize(c->stream));
    RT_HIP(hipMemcpy(t, b.d_tris + 2 * (size_t) k, k * sizeof(TriRec), hipMemcpyDeviceToHost)); // copy 2: coordinates in their natural order
    double amax = 0.0;
    double lo[RT_INLINE_MAX_TRIS][3], hi[RT_INLINE_MAX_TRIS][3];
    for (u32 j = 0; j < k; j++) {
        const float* v[3] = { t[j].v0, t[j].v1, t[j].v2 };
        for (int a = 0; a < 3; a++) { lo[j][a] = 1e300; hi[j][a] = -1e300; }
        for (int q = 0; q < 3; q++)
            for (int a = 0; a < 3; a++) {
                const double w = (double) v[q][0] * o2w[a] + (double) v[q][1] * o2w[4 + a] + (double) v[q][2] * o2w[8 + a] + o2w[12 + a];
                lo[j][a] = std::min(lo[j][a], w); hi[j][a] = std::max(hi[j][a], w); amax = std::max(amax, fabs(w));
            }
    }
    const double pad = 1.0e-4 * amax + 1.0e-30;
    c->inl_box_same = 0;
    for (u32 j = 0; j < k; j++) {
        for (int a = 0; a < 3; a++) { c->inl_box[j][a] = (float) (lo[j][a] - pad); c->inl_box[j][3 + a] = (float) (hi[j][a] + pad); }
        if (j > 0 && memcmp(c->inl_box[j], c->inl_box[j - 1], sizeof c->inl_box[j]) == 0) c->inl_box_same |= 1u << j;
    }
    return RT_OK;
}

void free_tlas_grid(rt_ctx* c) {
    dev_free(c->d_grid_start); dev_free(c->d_grid_entries);
    c->grid_cells = 0; c->grid_n_entries = 0;
}

// Grid top level (rt_dev_scene.h, trace_run_grid): a uniform grid over the world boxes of the top level's n_prims instances
// (lbvh.boxes, as rt_build_tlas left them: slot inl_inst holds the last instance).  Built on the host from the boxes the GPU
// computed; cell size from the number of instances (grid_density cells per instance), axes along which the instances do
// not spread get one cell.  Scenes it does not suit keep the tree alone: few instances, instances that span many cells
// (mode 1), bottom-level trees too deep for the grid kernels' stack rows.
int build_tlas_grid(rt_ctx* c, u32 n, u32 n_prims, int max_blas_depth) {
    free_tlas_grid(c);
    if (c->tlas_grid <= 0 || n_prims < 2) return RT_OK;
    if (max_blas_depth + 2 > RT_GRID_STACK_ROWS) return RT_OK;
    if (c->tlas_grid == 1 && n_prims < 4096) return RT_OK;
    const auto t_begin = std::chrono::steady_clock::now();
    std::vector<Aabb> box(n_prims);
    RT_HIP(hipMemcpy(box.data(), c->lbvh.boxes, sizeof(Aabb) * n_prims, hipMemcpyDeviceToHost));
    double lo[3] = { 1e300, 1e300, 1e300 }, hi[3] = { -1e300, -1e300, -1e300 };
    u32 valid = 0;
    for (u32 p = 0; p < n_prims; p++) {
        const Aabb& b = box[p];
        bool ok = true;
        for (int k = 0; k < 3; k++) ok = ok && b.lo[k] <= b.hi[k] && fabsf(b.lo[k]) < 1.0e30f && fabsf(b.hi[k]) < 1.0e30f;
        if (!ok) continue; // an instance without triangles: in no cell
        valid++;
        for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], (double) b.lo[k]); hi[k] = std::max(hi[k], (double) b.hi[k]); }
    }
    if (valid < 2) return RT_OK;
    double e[3], diag2 = 0.0, maxabs = 0.0;
    for (int k = 0; k < 3; k++) { e[k] = hi[k] - lo[k]; diag2 += e[k] * e[k]; maxabs = std::max(maxabs, std::max(fabs(lo[k]), fabs(hi[k]))); }
    const double slack = 2.0e-5 * (sqrt(diag2) + maxabs);
    if (!(slack > 0.0)) return RT_OK;
    for (int k = 0; k < 3; k++) { lo[k] -= 2.0 * slack; hi[k] += 2.0 * slack; e[k] = hi[k] - lo[k]; }
    // cell size: `cells` cells over the axes along which the instances spread
    const double cells = (double) valid * c->grid_density;
    bool thin[3] = { false, false, false };
    double size = 0.0;
    for (int iter = 0; iter < 3; iter++) {
        int m = 0; double vol = 1.0;
        for (int k = 0; k < 3; k++) if (!thin[k]) { m++; vol *= e[k]; }
        size = m == 3 ? cbrt(vol / cells) : m == 2 ? sqrt(vol / cells) : vol / cells;
        bool changed = false;
        for (int k = 0; k < 3; k++) if (!thin[k] && m > 1 && e[k] < 1.5 * size) { thin[k] = true; changed = true; break; } // one axis at a time, thinnest first below
        if (!changed) break;
    }
    u32 res[3]; double cell[3];
    for (int k = 0; k < 3; k++) {
        res[k] = thin[k] ? 1u : (u32) std::min(std::max(ceil(e[k] / size), 1.0), (double) RT_GRID_MAX_RES);
        cell[k] = e[k] / (double) res[k];
    }
    const uint64_t n_cells = (uint64_t) res[0] * res[1] * res[2];
    if (n_cells > (64ull << 20)) return RT_OK;
    // cells of each instance's grown box
    auto cell_range = [&](const Aabb& b, int k, u32& c0, u32& c1) {
        const double a = ((double) b.lo[k] - slack - lo[k]) / cell[k], z = ((double) b.hi[k] + slack - lo[k]) / cell[k];
        c0 = (u32) std::min(std::max(floor(a), 0.0), (double) (res[k] - 1));
        c1 = (u32) std::min(std::max(floor(z), 0.0), (double) (res[k] - 1));
    };
    std::vector<u32> start(n_cells + 1, 0u);
    uint64_t total = 0;
    for (u32 p = 0; p < n_prims; p++) {
        const Aabb& b = box[p];
        if (!(b.lo[0] <= b.hi[0] && b.lo[1] <= b.hi[1] && b.lo[2] <= b.hi[2]) || !(fabsf(b.lo[0]) < 1.0e30f)) continue;
        u32 c0[3], c1[3];
        for (int k = 0; k < 3; k++) cell_range(b, k, c0[k], c1[k]);
        const uint64_t span = (uint64_t) (c1[0] - c0[0] + 1) * (c1[1] - c0[1] + 1) * (c1[2] - c0[2] + 1);
        if (c->tlas_grid == 1 && span > 64) return RT_OK; // an instance much larger than the others: the tree handles it better
        total += span;
        if (total > (1ull << 28)) return RT_OK;
        for (u32 z = c0[2]; z <= c1[2]; z++) for (u32 y = c0[1]; y <= c1[1]; y++) for (u32 x = c0[0]; x <= c1[0]; x++)
            start[((uint64_t) z * res[1] + y) * res[0] + x + 1] += 1u;
    }
    if (total == 0 || (c->tlas_grid == 1 && total > 8ull * valid)) return RT_OK;
    for (uint64_t i = 0; i < n_cells; i++) start[i + 1] += start[i];
    std::vector<GridEntry> entries(total);
    std::vector<u32> fill(start.begin(), start.end() - 1);
    for (u32 p = 0; p < n_prims; p++) {
        const Aabb& b = box[p];
        if (!(b.lo[0] <= b.hi[0] && b.lo[1] <= b.hi[1] && b.lo[2] <= b.hi[2]) || !(fabsf(b.lo[0]) < 1.0e30f)) continue;
        u32 c0[3], c1[3];
        for (int k = 0; k < 3; k++) cell_range(b, k, c0[k], c1[k]);
        GridEntry en;
        for (int k = 0; k < 3; k++) { en.lo[k] = (float) ((double) b.lo[k] - slack); en.hi[k] = (float) ((double) b.hi[k] + slack); }
        en.inst = (c->inl_inst >= 0 && (int) p == c->inl_inst) ? n - 1u : p; // the slot of the inlined instance holds the last one
        en.pad = 0u;
        for (u32 z = c0[2]; z <= c1[2]; z++) for (u32 y = c0[1]; y <= c1[1]; y++) for (u32 x = c0[0]; x <= c1[0]; x++)
            entries[fill[((uint64_t) z * res[1] + y) * res[0] + x]++] = en;
    }
    RT_HIP(hipMalloc(&c->d_grid_start, sizeof(u32) * (n_cells + 1)));
    RT_HIP(hipMalloc(&c->d_grid_entries, sizeof(GridEntry) * total));
    RT_HIP(hipMemcpy(c->d_grid_start, start.data(), sizeof(u32) * (n_cells + 1), hipMemcpyHostToDevice));
    RT_HIP(hipMemcpy(c->d_grid_entries, entries.data(), sizeof(GridEntry) * total, hipMemcpyHostToDevice));
    for (int k = 0; k < 3; k++) { c->grid_res[k] = res[k]; c->grid_lo[k] = (float) lo[k]; c->grid_cell[k] = (float) cell[k]; }
    c->grid_slack = (float) slack;
    c->grid_cells = (u32) n_cells; c->grid_n_entries = (u32) total;
    c->tlas_build_ms += std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    return RT_OK;
}

void free_flat_tree(rt_ctx* c) {
    c->flat_inl_inst = -1; c->flat_inst0 = 0;
    dev_free(c->d_flat_nodes); c->d_flat_tris = nullptr; dev_free(c->d_flat_perm);
    c->flat_instances = 0; c->flat_tris = 0; c->flat_build_ms = 0;
}

// One tree over the world boxes of all instanced triangles, for scenes of one or two instances (every BASELINE
// configuration but the instance field): the walk then has no instance entries (trace_flat, rt_dev_scene.h).  Built with
// the same builders as the two levels; the triangles keep their object-space vertices (S7: the test runs there).
// Not built -- the two-level walk stays -- when the scene is larger than that or the tree too deep for the LDS stack.
int build_flat_tree(rt_ctx* c, const RtInstance* instances, int n, const std::vector<InstRec>& rec) {
    if (!c->flat_enabled || n > RT_FLAT_MAX_INSTANCES) return RT_OK;
    FlatSources src; memset(&src, 0, sizeof src);
    uint64_t total = 0;
    // two instances of which one has at most RT_INLINE_MAX_TRIS triangles (a mesh on a floor): the tree holds the other one
    // alone and every ray tests the small one's triangles first (SceneView::flat_inl_inst)
    int small = -1;
    if (n == 2 && c->flat_inline_enabled) {
        const u32 t0 = c->blas[instances[0].blas].n_tris, t1 = c->blas[instances[1].blas].n_tris;
        if (t1 <= RT_INLINE_MAX_TRIS && t0 > t1) small = 1; else if (t0 <= RT_INLINE_MAX_TRIS && t1 > t0) small = 0;
    }
    const int first_inst = small == 0 ? 1 : 0, n_all = n;
    if (small >= 0) { instances += first_inst; n = 1; } // (rec[] keeps the caller's numbering: rec[first_inst] below)
    for (int i = 0; i < n; i++) {
        const Blas& b = c->blas[instances[i].blas];
        FlatSource& fs = src.s[i];
        fs.verts = b.d_verts; fs.indices = b.d_indices; fs.inv_perm = b.d_inv_perm;
        memcpy(fs.o2w, instances[i].transform, 64);
        fs.first = (u32) total; fs.n_tris = b.n_tris;
        total += b.n_tris;
    }
    src.n = (u32) n;
    if (total > RT_FLAT_MAX_TRIS) return RT_OK;
    const u32 nt = (u32) total;
    u32 n_nodes = n > 1 ? 1u : 0u, largest = 0;
    for (int i = 0; i < n; i++) { n_nodes += src.s[i].n_tris > 1 ? src.s[i].n_tris - 1 : 1; largest = std::max(largest, src.s[i].n_tris); }
    int rc = lbvh_workspace_reserve(c->lbvh, largest); if (rc) return rc;
    RT_HIP(hipMalloc(&c->d_flat_nodes, n_nodes * sizeof(PackedNode) + 3 * (size_t) nt * sizeof(FlatTri)));
    c->d_flat_tris = (FlatTri*) (c->d_flat_nodes + n_nodes);
    RT_HIP(hipMalloc(&c->d_flat_perm, 4 * (size_t) nt));
    BvhNode* d_join = nullptr; // scratch of the build, freed on every way out of the lambda below
    RT_HIP(hipMalloc(&d_join, n_nodes * sizeof(BvhNode) + (RT_FLAT_MAX_INSTANCES + 1) * sizeof(Aabb)));
    auto build = [&]() -> int {
        Aabb* d_sub = (Aabb*) (d_join + n_nodes);
        RT_HIP(hipEventRecord(c->build_ev[0], c->stream));
        int brc = lbvh_flat_build(c->stream, c->lbvh, &c->ploc, &c->sah, c->bvh_builder, c->sah_tri, src, d_join, d_sub, c->d_flat_perm, c->d_flat_tris, c->d_flat_nodes);
        if (brc) return brc;
        RT_HIP(hipEventRecord(c->build_ev[1], c->stream));
        RT_HIP(hipMemcpyAsync(&c->flat_info, c->lbvh.tree_info, sizeof c->flat_info, hipMemcpyDeviceToHost, c->stream));
        RT_HIP(hipMemcpyAsync(c->flat_box, d_sub, (n > 1 ? 2 : 1) * sizeof c->flat_box[0], hipMemcpyDeviceToHost, c->stream)); // world boxes of the instances in the tree
        c->flat_big = (n > 1 && src.s[1].n_tris > src.s[0].n_tris) ? 1u : 0u;
        return RT_OK;
    };
    rc = build();
    { hipError_t e = hipStreamSynchronize(c->stream); (void) hipFree(d_join); if (rc) return rc; RT_HIP(e); }
    float ms = 0; RT_HIP(hipEventElapsedTime(&ms, c->build_ev[0], c->build_ev[1]));
    // k_traverse<.., FLAT> keeps the whole stack in its 26 LDS rows: sentinel + one entry per level + the row above the top
    if (c->flat_info.depth + 2 > 26) { free_flat_tree(c); return RT_OK; }
    for (int i = 0; i < n; i++) memcpy(c->flat_w2o[i], rec[first_inst + i].world_to_object, 48);
    c->flat_inl_inst = small; c->flat_inst0 = (u32) first_inst;
    if (small >= 0) {
        const Blas& b = c->blas[(instances - first_inst)[small].blas];
        c->inl_count = b.n_tris; c->inl_tris = b.d_tris; c->inl_copy_bytes = (u32) (b.n_tris * sizeof(TriRec));
        memcpy(c->inl_w2o, rec[small].world_to_object, 48);
        int brc = set_inline_boxes(c, b, (instances - first_inst)[small].transform); if (brc) { c->flat_inl_inst = -1; return brc; }
    }
    (void) n_all;
    c->flat_instances = (u32) n; c->flat_tris = nt; c->flat_build_ms = ms; c->flat_first1 = n > 1 ? src.s[1].first : nt;
    return RT_OK;
}

// device half of rt_build_blas: uploads the concatenated geometry and builds the bottom-level tree on c->stream
int build_blas_on_device(rt_ctx* c, Blas& b, const std::vector<u32>& tri_geometry) {
    u32 n_nodes = b.n_tris > 1 ? b.n_tris - 1 : 1;
    RT_HIP(hipMalloc(&b.d_verts, b.verts.size() * sizeof(RtVertex)));
    RT_HIP(hipMalloc(&b.d_indices, b.indices.size() * sizeof(uint16_t)));
    RT_HIP(hipMalloc(&b.d_geoms, b.geoms.size() * sizeof(GeomRec)));
    RT_HIP(hipMalloc(&b.d_tri_geometry, tri_geometry.size() * 4));
    // nodes and triangles share one allocation: an instance record addresses both with one pointer + a 32-bit offset
    RT_HIP(hipMalloc(&b.d_nodes, n_nodes * sizeof(PackedNode) + 3 * (size_t) b.n_tris * sizeof(TriRec))); // three copies of the triangles (TriRec)
    b.d_tris = (TriRec*) (b.d_nodes + n_nodes);
    RT_HIP(hipMalloc(&b.d_perm, b.n_tris * 4));
    RT_HIP(hipMalloc(&b.d_inv_perm, b.n_tris * 4));
    RT_HIP(hipMemcpyAsync(b.d_verts, b.verts.data(), b.verts.size() * sizeof(RtVertex), hipMemcpyHostToDevice, c->stream));
    RT_HIP(hipMemcpyAsync(b.d_indices, b.indices.data(), b.indices.size() * sizeof(uint16_t), hipMemcpyHostToDevice, c->stream));
    RT_HIP(hipMemcpyAsync(b.d_geoms, b.geoms.data(), b.geoms.size() * sizeof(GeomRec), hipMemcpyHostToDevice, c->stream));
    RT_HIP(hipMemcpyAsync(b.d_tri_geometry, tri_geometry.data(), tri_geometry.size() * 4, hipMemcpyHostToDevice, c->stream));
    int rc = lbvh_workspace_reserve(c->lbvh, b.n_tris); if (rc) return rc;
    RT_HIP(hipEventRecord(c->build_ev[0], c->stream));
    rc = lbvh_triangle_boxes(c->stream, c->lbvh, b.d_verts, b.d_indices, b.n_tris); if (rc) return rc;
    rc = lbvh_build(c->stream, c->lbvh, &c->ploc, &c->sah, c->bvh_builder, c->sah_tri, b.n_tris, RT_LEAF_MAX_TRIS, false, b.d_nodes, b.d_perm); if (rc) return rc;
    rc = lbvh_write_trirecs(c->stream, b.d_perm, b.d_verts, b.d_indices, b.d_tri_geometry, b.d_geoms, b.n_tris, b.d_tris); if (rc) return rc;
    rc = lbvh_invert_perm(c->stream, b.d_perm, b.n_tris, b.d_inv_perm); if (rc) return rc;
    RT_HIP(hipEventRecord(c->build_ev[1], c->stream));
    RT_HIP(hipMemcpyAsync(&b.info, c->lbvh.tree_info, sizeof b.info, hipMemcpyDeviceToHost, c->stream));
    RT_HIP(hipStreamSynchronize(c->stream)); // host arrays are borrowed only for this call
    float ms = 0; RT_HIP(hipEventElapsedTime(&ms, c->build_ev[0], c->build_ev[1]));
    c->blas_build_ms = ms;
    return RT_OK;
}

} // namespace

extern "C" {

const char* rt_last_error(void) { return g_error; }
int rt_abi_version(void) { return RT_ABI_VERSION; }

// ---- rt_dispatch_rays_static: the same frame with every scratch buffer in a static device array --------------------
// No counterpart in the reference.  For callers that cannot allocate (HIP graph capture, serving loops, harnesses that
// forbid hipMalloc): rt_dispatch_rays keeps its ray queues, hit records, path state, segment counters and traversal
// stack overflow in buffers that it (re)allocates on demand; here they live in `scratch` below, which every context on
// a GPU shares.  Host code never uses the array's name as an address: rt_create resolves it per device
// (hipGetSymbolAddress) and everything goes through that pointer.
static constexpr u32 RT_STATIC_MAX_SEGMENTS = 64;  // 65,536 queue slots: frames up to 64 tiles of 32x32 pixels (e.g. 256x256) and 65,536 sample points
static constexpr size_t RT_STATIC_SEG_BYTES = 4096; // seg_count_words(64) * 4 = 1,028 bytes, padded: the arena behind it stays 256-byte aligned
static constexpr size_t RT_STATIC_SCRATCH_BYTES = RT_STATIC_SEG_BYTES + (size_t) RT_STATIC_MAX_SEGMENTS * 1024 * 4 * PATH_ARRAYS; // 27,529,216
static __device__ __align__(256) unsigned char scratch[RT_STATIC_SCRATCH_BYTES];

namespace {
constexpr int RT_MAX_DEVICES = 64;
void* g_scratch_ptr[RT_MAX_DEVICES]; // device address of `scratch` on each GPU this process has created a context on
std::mutex g_scratch_mutex;

// with device `device` current: resolves this GPU's copy of `scratch` once
int resolve_static_scratch(int device, void** out) {
    if (device < 0 || device >= RT_MAX_DEVICES) { set_error("static scratch: device %d out of range", device); return RT_ERR_INVALID; }
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        void* p = nullptr;
        RT_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        if (!p) { set_error("static scratch: hipGetSymbolAddress returned a null address"); return RT_ERR_DEVICE; }
        g_scratch_ptr[device] = p;
    }
    *out = g_scratch_ptr[device];
    return RT_OK;
}
} // namespace

namespace {
int create_context(rt_ctx* c) {
    if (const char* v = getenv("RT_BVH_BUILDER")) c->bvh_builder = atoi(v);
    if (const char* v = getenv("RT_SAH_TRI")) c->sah_tri = (float) atof(v);
    if (const char* v = getenv("RT_RAY_SORT")) c->ray_sort = atoi(v);
    if (const char* v = getenv("RT_WAVE_MAJOR")) c->wave_major = atoi(v);
    if (const char* v = getenv("RT_FUSE_PRIMARY")) c->fuse_primary = atoi(v);
    if (const char* v = getenv("RT_FUSE_TAIL")) c->fuse_tail = atoi(v);
    if (const char* v = getenv("RT_FUSE_QUEUED")) c->fuse_queued = atoi(v);
    if (const char* v = getenv("RT_GATHER_SIDE")) c->gather_side = atoi(v);
    if (const char* v = getenv("RT_POOL")) c->pool = atoi(v);
    if (const char* v = getenv("RT_GATHER_L1")) c->gather_l1 = atoi(v);
    if (const char* v = getenv("RT_GATHER_BATCH")) c->gather_batch = atoi(v);
    if (const char* v = getenv("RT_GATHER_FUSED")) c->gather_fused = atoi(v);
    if (const char* v = getenv("RT_PROBE_SIDE")) c->probe_side = atoi(v);
    if (const char* v = getenv("RT_GATHER_BLOCK")) { const int n = atoi(v); if (n == 256 || n == 512 || n == 1024) c->gather_block = n; }
    if (const char* v = getenv("RT_GATHER_GROUPS")) { const int n = atoi(v); if (n >= 1 && n <= 16) c->gather_groups = n; }
    if (const char* v = getenv("RT_GATHER_BATCH_MB")) { const long mb = atol(v); if (mb > 0) c->gather_batch_mb = (size_t) mb; }
    if (const char* v = getenv("RT_XCD_BANDS")) c->xcd_bands = atoi(v);
    if (const char* v = getenv("RT_SORT_BOX")) c->sort_box = atoi(v);
    if (const char* v = getenv("RT_SORT_CHORD")) c->sort_chord_pct = (float) atof(v);
    if (const char* v = getenv("RT_INLINE")) c->inline_enabled = atoi(v) != 0;
    if (const char* v = getenv("RT_TLAS_GRID")) c->tlas_grid = atoi(v);
    if (const char* v = getenv("RT_GRID_DENSITY")) { c->grid_density = (float) atof(v); if (!(c->grid_density > 0.01f && c->grid_density < 100.0f)) c->grid_density = 1.0f; }
    if (const char* v = getenv("RT_FLAT_INLINE")) c->flat_inline_enabled = atoi(v) != 0;
    if (const char* v = getenv("RT_PERSISTENT")) c->persistent = atoi(v);
    if (const char* v = getenv("RT_STACK_LDS")) c->stack_lds_ok = atoi(v) != 0;
    if (const char* v = getenv("RT_FLAT")) c->flat_enabled = atoi(v) != 0;
    if (const char* v = getenv("RT_STACK_DEEP")) c->stack_deep_ok = atoi(v) != 0;
    if (const char* v = getenv("RT_FLAT_WAVES8")) c->flat_waves8 = atoi(v) != 0;
    if (const char* v = getenv("RT_FLAT_WAVES7")) c->flat_waves7 = atoi(v) != 0;
    if (const char* v = getenv("RT_FRAMES_IN_FLIGHT")) c->frames_in_flight = atoi(v);
    if (const char* v = getenv("RT_EXACT_INSTANCE_BOXES")) c->exact_instance_boxes = atoi(v) != 0;
    hipDeviceProp_t prop;
    RT_HIP(hipGetDeviceProperties(&prop, c->device));
    c->persistent_slots = (u32) prop.multiProcessorCount * 4u * 5u;
    c->cu_count = (u32) prop.multiProcessorCount;
    if (c->pool == 1 || c->pool > 2) c->pool = 2; // two 256-ray pools per CU: what their LDS allows
    RT_HIP(hipMalloc(&c->d_task_counter, 65 * 128));
    RT_HIP(hipMemset(c->d_task_counter, 0, 65 * 128));
    RT_HIP(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    c->stream = c->own_stream;
    RT_HIP(hipMalloc(&c->d_stats, sizeof(DeviceStats)));
    RT_HIP(hipMemset(c->d_stats, 0, sizeof(DeviceStats)));
    for (int k = 0; k < 2; k++) RT_HIP(hipEventCreate(&c->build_ev[k]));
    return resolve_static_scratch(c->device, &c->d_static_scratch); // rt_dispatch_rays_static
}
} // namespace

int rt_create(const int* devices, int n, rt_ctx** out) {
    if (!out || !devices || n < 1) { set_error("rt_create: bad arguments"); return RT_ERR_INVALID; }
    if (n != 1) { set_error("rt_create: one process drives one GPU (n must be 1)"); return RT_ERR_UNSUPPORTED; }
    int count = 0;
    RT_HIP(hipGetDeviceCount(&count));
    if (count < 1) { set_error("rt_create: no HIP device; this library has no CPU fallback"); return RT_ERR_DEVICE; }
    if (devices[0] < 0 || devices[0] >= count) { set_error("rt_create: device %d out of range", devices[0]); return RT_ERR_INVALID; }
    RT_HIP(hipSetDevice(devices[0]));
    rt_ctx* c = new rt_ctx();
    c->device = devices[0];
    int rc = create_context(c);
    if (rc != RT_OK) { rt_destroy(c); return rc; } // frees whatever the failed set-up had already made
    *out = c;
    return RT_OK;
}

void rt_destroy(rt_ctx* c) {
    if (!c) return;
    (void) hipSetDevice(c->device);
    (void) quiesce(c);
    if (c->stream) (void) hipStreamSynchronize(c->stream);
    (void) drain_timers(c);
    dev_free(c->alt.d_arena); dev_free(c->alt.d_seg_count);
    for (auto& b : c->bg) { dev_free(b.d_arena); dev_free(b.d_counts); dev_free(b.d_off); }
    if (c->probe_stream) { (void) hipStreamSynchronize(c->probe_stream); (void) hipStreamDestroy(c->probe_stream); }
    dev_free(c->probe_pb.d_arena); dev_free(c->probe_pb.d_seg_count);
    if (c->alt_stream) (void) hipStreamDestroy(c->alt_stream);
    for (int l = 0; l < 2; l++) { if (c->gather_stream[l]) { (void) hipStreamSynchronize(c->gather_stream[l]); (void) hipStreamDestroy(c->gather_stream[l]); }
                                  for (int k = 0; k < 2; k++) if (c->gather_ev[l][k]) (void) hipEventDestroy(c->gather_ev[l][k]); }
    for (int k = 0; k < 2; k++) if (c->lane_done[k]) (void) hipEventDestroy(c->lane_done[k]);
    for (int k = 0; k < 2; k++) if (c->probe_done[k]) (void) hipEventDestroy(c->probe_done[k]);
    for (auto& b : c->blas) free_blas(b);
    for (auto& m : c->translucent_meshes) dev_free(m.d_partial_areas);
    free_sample_points(c);
    dev_free(c->d_inst_rec); dev_free(c->d_inst_shade); dev_free(c->d_tlas_nodes); dev_free(c->d_tlas_perm);
    free_flat_tree(c); free_tlas_grid(c);
    dev_free(c->d_task_counter);
    dev_free(c->d_accumulator); dev_free(c->d_render_target); dev_free(c->d_capture); dev_free(c->d_arena);
    dev_free(c->d_seg_count); dev_free(c->d_stats); dev_free(c->d_table[0]); dev_free(c->d_table[1]); dev_free(c->d_gather_delta);
    lbvh_workspace_free(c->lbvh); ploc_workspace_free(c->ploc); sah_workspace_free(c->sah);
    for (int k = 0; k < 2; k++) if (c->build_ev[k]) (void) hipEventDestroy(c->build_ev[k]);
    if (c->own_stream) (void) hipStreamDestroy(c->own_stream);
    delete c;
}

int rt_set_stream(rt_ctx* c, void* s) {
    if (!c) return RT_ERR_INVALID;
    { int rc = quiesce(c); if (rc) return rc; }
    RT_HIP(hipStreamSynchronize(c->stream));
    c->stream = s ? (hipStream_t) s : c->own_stream;
    return RT_OK;
}

int rt_update_resolution(rt_ctx* c, uint32_t w, uint32_t h) {
    // each side <= 32768 keeps tiles * 1024 queue slots (and every pixel index) inside 32 bits
    if (!c || w == 0 || h == 0 || w > 32768u || h > 32768u || (uint64_t) w * h > (1ull << 28)) { set_error("rt_update_resolution: bad size %ux%u (each side <= 32768, at most 2^28 pixels)", w, h); return RT_ERR_INVALID; }
    RT_HIP(hipSetDevice(c->device));
    { int rc = quiesce(c); if (rc) return rc; }
    RT_HIP(hipStreamSynchronize(c->stream));
    dev_free(c->d_accumulator); dev_free(c->d_render_target);
    c->width = w; c->height = h;
    RT_HIP(hipMalloc(&c->d_accumulator, (size_t) w * h * 16));
    RT_HIP(hipMalloc(&c->d_render_target, (size_t) w * h * 4));
    RT_HIP(hipMemsetAsync(c->d_accumulator, 0, (size_t) w * h * 16, c->stream));
    RT_HIP(hipMemsetAsync(c->d_render_target, 0, (size_t) w * h * 4, c->stream));
    c->render_target_stale = false;
    return ensure_path_buffers(c);
}

int rt_load_bssrdf_table(rt_ctx* c, int profile, const float* r, const float* g, const float* b, uint32_t n) {
    if (!c || profile < 0 || profile > 1 || !r || !g || !b || n == 0) { set_error("rt_load_bssrdf_table: bad arguments"); return RT_ERR_INVALID; }
    { int qrc = quiesce(c); if (qrc) return qrc; }
    RT_HIP(hipSetDevice(c->device));
    u32 padded = ((n + 255u) / 256u) * 256u; // raytracing.cpp:113
    std::vector<float> t((size_t) padded * 3, 0.0f);
    for (u32 i = 0; i < n; i++) { t[3 * i] = r[i]; t[3 * i + 1] = g[i]; t[3 * i + 2] = b[i]; }
    RT_HIP(hipStreamSynchronize(c->stream));
    dev_free(c->d_table[profile]);
    c->delta_table = nullptr; // (a new table may get the old one's address)
    RT_HIP(hipMalloc(&c->d_table[profile], t.size() * 4));
    RT_HIP(hipMemcpy(c->d_table[profile], t.data(), t.size() * 4, hipMemcpyHostToDevice));
    c->table_texels[profile] = (int) padded;
    return RT_OK;
}

int rt_set_bssrdf_profile(rt_ctx* c, int profile) {
    if (!c || profile < 0 || profile > 1) return RT_ERR_INVALID;
    c->profile = profile;
    return RT_OK;
}

int rt_build_blas(rt_ctx* c, const RtGeometry* geometries, int n, rt_blas* out) {
    if (!c || !geometries || n < 1 || !out) { set_error("rt_build_blas: bad arguments"); return RT_ERR_INVALID; }
    { int qrc = quiesce(c); if (qrc) return qrc; }
    RT_HIP(hipSetDevice(c->device));
    Blas b;
    b.translucent_ids_index = (u32) c->translucent_mean_area.size();
    std::vector<u32> tri_geometry;
    for (int gi = 0; gi < n; gi++) {
        const RtGeometry& ge = geometries[gi];
        if (!ge.vertices || !ge.indices || ge.index_count % 3 != 0 || ge.shader < 0 || ge.shader >= RT_SHADER_COUNT) {
            set_error("rt_build_blas: geometry %d is malformed", gi); return RT_ERR_INVALID;
        }
        GeomRec rec;
        rec.index_offset = (u32) b.indices.size(); rec.first_tri = rec.index_offset / 3;
        rec.shader = ge.shader; rec.translucent_id = -1; // assigned below, raytracing.cpp:337
        rec.color[0] = ge.color[0]; rec.color[1] = ge.color[1]; rec.color[2] = ge.color[2]; rec._pad = 0;
        for (size_t k = 0; k < ge.index_count; k++) {
            if (ge.indices[k] >= ge.vertex_count) { set_error("rt_build_blas: geometry %d index out of range", gi); return RT_ERR_INVALID; }
            b.indices.push_back((uint16_t) (b.verts.size() + ge.indices[k])); // rebase, raytracing.cpp:360-366
        }
        b.verts.insert(b.verts.end(), ge.vertices, ge.vertices + ge.vertex_count);
        for (size_t k = 0; k < ge.index_count / 3; k++) tri_geometry.push_back((u32) gi);
        b.geoms.push_back(rec);
    }
    if (b.verts.size() > 65535) { set_error("rt_build_blas: %zu vertices exceed the 16-bit index range", b.verts.size()); return RT_ERR_LIMIT; } // raytracing.cpp:367
    b.n_tris = (u32) b.indices.size() / 3;
    // InstRec addresses the three TriRec copies with 32-bit byte offsets: nodes + 3 copies must stay below 4 GiB
    if (b.n_tris == 0 || b.n_tris > (1u << 24)) { set_error("rt_build_blas: bad triangle count %u (limit %u)", b.n_tris, 1u << 24); return RT_ERR_LIMIT; }
    int blas_index = (int) c->blas.size();
    // translucent geometries, raytracing.cpp:377-398 + Bluenoise::preprocess_mesh_data, bluenoise.cpp:69-93
    std::vector<TranslucentMesh> new_meshes;
    for (int gi = 0; gi < n; gi++) {
        if (b.geoms[gi].shader != RT_SHADER_TRANSLUCENT) continue;
        b.translucent_ids_count++;
        b.geoms[gi].translucent_id = (int) (c->translucent_mean_area.size() + new_meshes.size());
        TranslucentMesh m;
        m.blas = blas_index; m.ib_offset = b.geoms[gi].index_offset; m.indices_count = (u32) geometries[gi].index_count;
        m.total_surface_area = 0; m.d_partial_areas = nullptr;
        const float inf = std::numeric_limits<float>::infinity();
        for (int k = 0; k < 3; k++) { m.aabb_min[k] = inf; m.aabb_max[k] = -inf; }
        std::vector<float> partial;
        for (u32 i = 0; i < m.indices_count / 3; i++) {
            const float* p[3];
            for (int k = 0; k < 3; k++) p[k] = b.verts[b.indices[m.ib_offset + 3 * i + k]].position;
            m.total_surface_area += triangle_area(p[0], p[1], p[2]);
            partial.push_back(m.total_surface_area);
            for (int k = 0; k < 3; k++) for (int x = 0; x < 3; x++) {
                m.aabb_min[x] = std::min(m.aabb_min[x], p[k][x]); m.aabb_max[x] = std::max(m.aabb_max[x], p[k][x]);
            }
        }
        if (!partial.empty()) {
            hipError_t e = hipMalloc(&m.d_partial_areas, partial.size() * 4);
            if (e == hipSuccess) e = hipMemcpy(m.d_partial_areas, partial.data(), partial.size() * 4, hipMemcpyHostToDevice);
            if (e != hipSuccess) {
                set_error("rt_build_blas: area table upload failed: %s", hipGetErrorString(e));
                dev_free(m.d_partial_areas);
                for (auto& done : new_meshes) dev_free(done.d_partial_areas);
                return RT_ERR_DEVICE;
            }
        }
        new_meshes.push_back(m);
    }
    // upload + build; a failure frees everything this call allocated
    int rc = build_blas_on_device(c, b, tri_geometry);
    if (rc != RT_OK) {
        free_blas(b);
        for (auto& m : new_meshes) dev_free(m.d_partial_areas);
        return rc;
    }
    b.depth = b.info.depth;
    c->blas_triangles += b.n_tris;
    for (auto& m : new_meshes) { c->translucent_meshes.push_back(m); c->translucent_mean_area.push_back(0.0f); }
    c->blas.push_back(std::move(b));
    *out = blas_index;
    return RT_OK;
}

int rt_build_tlas(rt_ctx* c, const RtInstance* instances, int n) {
    if (!c || !instances || n < 1) { set_error("rt_build_tlas: bad arguments"); return RT_ERR_INVALID; }
    { int qrc = quiesce(c); if (qrc) return qrc; }
    RT_HIP(hipSetDevice(c->device));
    for (int i = 0; i < n; i++)
        if (instances[i].blas < 0 || instances[i].blas >= (int) c->blas.size()) { set_error("rt_build_tlas: instance %d has a bad BLAS handle", i); return RT_ERR_INVALID; }
    RT_HIP(hipStreamSynchronize(c->stream));
    // reset translucent mesh instances, raytracing.cpp:451-461
    free_sample_points(c);
    c->translucent_instances.clear();
    c->g.translucent_instance_stride = 0;
    std::vector<u32> counts(c->blas.size(), 0);
    std::vector<InstRec> rec(n); std::vector<InstShade> shade(n);
    int max_blas_depth = 0;
    uint64_t vertex_work = 0; // vertices to transform for exact instance boxes
    for (int i = 0; i < n; i++) {
        const Blas& b = c->blas[instances[i].blas];
        counts[instances[i].blas] += 1;
        u32 instance_id = counts[instances[i].blas] - 1; // raytracing.cpp:472-479
        if (b.translucent_ids_count > 0) {
            for (u32 j = 0; j < b.translucent_ids_count; j++) {
                TranslucentInstance ti;
                ti.translucent_id = b.translucent_ids_index + j; ti.instance_id = instance_id;
                memcpy(ti.transform, instances[i].transform, 64); ti.scale_factor = 0; ti.samples_count = 0;
                c->translucent_instances.push_back(ti);
            }
            c->g.translucent_instance_stride = std::max(c->g.translucent_instance_stride, counts[instances[i].blas]);
        }
        invert_affine(instances[i].transform, rec[i].world_to_object);
        rec[i].nodes = b.d_nodes; rec[i].tris_offset = (u32) ((const char*) b.d_tris - (const char*) b.d_nodes); rec[i].bound = b.info.bound; rec[i].tris_copy_bytes = (u32) (b.n_tris * sizeof(TriRec));
        for (int k = 0; k < 3; k++) { rec[i].ctr[k] = b.info.ctr[k]; rec[i].inv_h[k] = b.info.inv_h[k]; }
        rec[i].pad = 0.0f;
        memcpy(shade[i].object_to_world, instances[i].transform, 64);
        shade[i].vertices = b.d_verts; shade[i].indices = b.d_indices; shade[i].geoms = b.d_geoms; shade[i].tris = b.d_tris;
        shade[i].instance_id = instance_id; shade[i].vertex_count = (u32) b.verts.size();
        vertex_work += b.verts.size();
        max_blas_depth = std::max(max_blas_depth, b.depth);
    }
    dev_free(c->d_inst_rec); dev_free(c->d_inst_shade); dev_free(c->d_tlas_nodes); dev_free(c->d_tlas_perm);
    free_flat_tree(c); free_tlas_grid(c);
    c->n_instances = 0; c->tlas_depth = 0; // no top level until this call has succeeded: a failure below leaves rt_dispatch_rays answering RT_ERR_STATE
    // the instance to inline (SceneView::inl_inst): among those of at most RT_INLINE_MAX_TRIS triangles the one that reaches
    // furthest (a floor under 50,000 meshes, a room around them); scenes that get a flat tree have no use for it
    c->inl_inst = -1; c->inl_count = 0; c->inl_tris = nullptr; c->inl_copy_bytes = 0;
    u32 inl_k = 0;
    if (c->inline_enabled && n > RT_FLAT_MAX_INSTANCES) {
        float reach_best = 0.0f;
        for (int i = 0; i < n; i++) {
            const Blas& b = c->blas[instances[i].blas];
            if (b.n_tris == 0 || b.n_tris > RT_INLINE_MAX_TRIS) continue;
            const float* m = instances[i].transform;
            float reach = 0.0f;
            for (int k = 0; k < 3; k++) reach = std::max(reach, fabsf(m[12 + k]) + b.info.bound * (fabsf(m[k]) + fabsf(m[4 + k]) + fabsf(m[8 + k])));
            if (c->inl_inst < 0 || reach > reach_best) { c->inl_inst = i; reach_best = reach; inl_k = b.n_tris; }
        }
    }
    const u32 n_prims = (u32) n - (inl_k ? 1u : 0u); // top-level primitives: the instances but the inlined one
    u32 n_nodes = n_prims > 1 ? n_prims - 1 : 1;
    RT_HIP(hipMalloc(&c->d_inst_rec, sizeof(InstRec) * n));
    RT_HIP(hipMalloc(&c->d_inst_shade, sizeof(InstShade) * n));
    RT_HIP(hipMalloc(&c->d_tlas_nodes, sizeof(PackedNode) * n_nodes));
    RT_HIP(hipMalloc(&c->d_tlas_perm, 4 * (size_t) n_prims));
    RT_HIP(hipMemcpyAsync(c->d_inst_rec, rec.data(), sizeof(InstRec) * n, hipMemcpyHostToDevice, c->stream));
    RT_HIP(hipMemcpyAsync(c->d_inst_shade, shade.data(), sizeof(InstShade) * n, hipMemcpyHostToDevice, c->stream));
    // lbvh_instance_boxes writes the boxes of all n instances (the inlined one's too: lbvh_move_box reads boxes[n - 1]);
    // the build below then covers the first n_prims of them
    int rc = lbvh_workspace_reserve(c->lbvh, (u32) n); if (rc) return rc;
    RT_HIP(hipEventRecord(c->build_ev[0], c->stream));
    const bool exact_boxes = c->exact_instance_boxes && vertex_work <= (4ull << 30);
    rc = lbvh_instance_boxes(c->stream, c->lbvh, c->d_inst_shade, c->d_inst_rec, (u32) n, exact_boxes); if (rc) return rc;
    if (c->inl_inst >= 0) {
        const Blas& b = c->blas[instances[c->inl_inst].blas];
        if (c->inl_inst != n - 1) { rc = lbvh_move_box(c->stream, c->lbvh, (u32) c->inl_inst, (u32) n - 1u); if (rc) { c->inl_inst = -1; return rc; } }
        c->inl_count = inl_k; c->inl_tris = b.d_tris; c->inl_copy_bytes = (u32) (b.n_tris * sizeof(TriRec));
        memcpy(c->inl_w2o, rec[c->inl_inst].world_to_object, 48);
        rc = set_inline_boxes(c, b, instances[c->inl_inst].transform); if (rc) { c->inl_inst = -1; return rc; }
    }
    rc = lbvh_build(c->stream, c->lbvh, &c->ploc, &c->sah, c->bvh_builder, c->sah_tri, n_prims, 1, true, c->d_tlas_nodes, c->d_tlas_perm); if (rc) { c->inl_inst = -1; return rc; }
    RT_HIP(hipEventRecord(c->build_ev[1], c->stream));
    RT_HIP(hipMemcpyAsync(&c->tlas_info, c->lbvh.tree_info, sizeof c->tlas_info, hipMemcpyDeviceToHost, c->stream));
    RT_HIP(hipStreamSynchronize(c->stream));
    float ms = 0; RT_HIP(hipEventElapsedTime(&ms, c->build_ev[0], c->build_ev[1]));
    c->tlas_build_ms = ms;
    if (c->tlas_info.depth + 1 + max_blas_depth > 64) {
        set_error("rt_build_tlas: BVH too deep for the traversal stack (tlas %d + blas %d)", c->tlas_info.depth, max_blas_depth);
        return RT_ERR_LIMIT;
    }
    rc = build_flat_tree(c, instances, n, rec);
    if (rc != RT_OK) { free_flat_tree(c); return rc; }
    c->tlas_build_ms += c->flat_build_ms;
    if (!c->flat_instances) { rc = build_tlas_grid(c, (u32) n, n_prims, max_blas_depth); if (rc != RT_OK) { free_tlas_grid(c); return rc; } }
    c->tlas_depth = c->tlas_info.depth; c->max_blas_depth = max_blas_depth;
    c->n_instances = (u32) n;
    return RT_OK;
}

namespace { int hg_build_all(rt_ctx* c); } // hierarchical gather trees, defined below

int rt_generate_translucent_samples(rt_ctx* c, float radius, uint32_t rng_seed, uint32_t phase_seed, uint32_t* total) {
    if (!c || !total || !(radius > 0.0f)) { set_error("rt_generate_translucent_samples: bad arguments"); return RT_ERR_INVALID; }
    { int qrc = quiesce(c); if (qrc) return qrc; }
    *total = 0;
    if (c->n_instances == 0) { set_error("rt_generate_translucent_samples: call rt_build_tlas first"); return RT_ERR_STATE; }
    if (c->g.translucent_instance_stride == 0) return RT_OK; // raytracing.cpp:554
    RT_HIP(hipSetDevice(c->device));
    RT_HIP(hipStreamSynchronize(c->stream));
    const auto wall0 = std::chrono::steady_clock::now(); // the call blocks: its wall time is the cost of the blue-noise set-up (todo.txt:4)
    free_sample_points(c);
    u32 stride = c->g.translucent_instance_stride;
    c->slot_count = (u32) c->translucent_mean_area.size() * stride;
    std::vector<int> slot_to_ti(c->slot_count, -1);
    std::vector<RtSamplePoint*> d_points(c->translucent_instances.size(), nullptr);
    std::vector<float*> d_normals(c->translucent_instances.size(), nullptr);
    u32 n_total = 0;
    int status = RT_OK;
    for (size_t i = 0; i < c->translucent_instances.size() && status == RT_OK; i++) {
        TranslucentInstance& ti = c->translucent_instances[i];
        const TranslucentMesh& mesh = c->translucent_meshes[ti.translucent_id];
        const Blas& b = c->blas[mesh.blas];
        BluenoiseParams p; memset(&p, 0, sizeof p);
        p.rng_seed = rng_seed + (u32) i;
        p.indices_offset = mesh.ib_offset;
        p.triangles_count = mesh.indices_count / 3;
        p.total_surface_area = mesh.total_surface_area;
        memcpy(p.transform, ti.transform, 64);
        { // define cell grid, bluenoise.cpp:119-147
            float scale = 0;
            for (int r = 0; r < 3; r++) { const float* row = &ti.transform[4 * r]; scale += sqrtf(row[0] * row[0] + row[1] * row[1] + row[2] * row[2]); }
            scale /= 3;
            ti.scale_factor = scale;
            p.rejection_radius = radius / scale;
            p.grid_cell_width = (float) ((double) p.rejection_radius / sqrt(3.0));
            float dims[3];
            for (int k = 0; k < 3; k++) {
                float size = mesh.aabb_max[k] - mesh.aabb_min[k];
                dims[k] = ceilf(size / p.grid_cell_width + 0.5f);
                p.grid_origin[k] = mesh.aabb_min[k] - 0.5f * (p.grid_cell_width * dims[k] - size);
                p.grid_dims[k] = (u32) dims[k];
            }
            double rr = p.rejection_radius;
            double ub = ceil((double) mesh.total_surface_area / (0.5 * 6.28318530717958647692528676655900577 * 0.25 * rr * rr));
            if (!(ub >= 1.0) || ub > 1.0e7 || (double) p.grid_dims[0] * p.grid_dims[1] * p.grid_dims[2] > 2.0e9) {
                set_error("rt_generate_translucent_samples: radius %g gives an unreasonable point budget", radius);
                status = RT_ERR_LIMIT; break;
            }
            u32 upper = (u32) ub;
            p.initial_count = 1;
            while (p.initial_count < 16 * upper) p.initial_count *= 2;
            p.buckets = upper;
        }
        p.phases = (c->flags & RT_FLAG_ALL_PHASE_GROUPS) ? 27u : 24u;
        phase_sequence(phase_seed, p.phases, p.phase_sequence);
        u32 count = 0;
        status = bluenoise_generate(c->stream, p, b.d_verts, b.d_indices, mesh.d_partial_areas, &d_points[i], &d_normals[i], &count);
        if (status != RT_OK) break;
        ti.samples_count = count;
        float mean_area = ti.scale_factor * ti.scale_factor * mesh.total_surface_area / (float) count; // raytracing.cpp:595
        c->translucent_mean_area[ti.translucent_id] = mean_area;
        TranslucentSlice sl; sl.offset = n_total; sl.count = count;
        sl.slot = ti.translucent_id * stride + ti.instance_id; sl.mean_area = mean_area;
        c->slices.push_back(sl);
        slot_to_ti[sl.slot] = (int) i;
        n_total += count;
    }
    if (status == RT_OK && n_total > 0) {
        c->n_points = n_total;
        hipError_t e = hipSuccess;
        if ((e = hipMalloc(&c->d_sp_read, sizeof(RtSamplePoint) * n_total)) != hipSuccess ||
            (e = hipMalloc(&c->d_sp_write, sizeof(RtSamplePoint) * n_total)) != hipSuccess ||
            (e = hipMalloc(&c->d_sp_normal, 12 * (size_t) n_total)) != hipSuccess ||
            (e = hipMalloc(&c->d_sp_slice, 4 * (size_t) n_total)) != hipSuccess) {
            set_error("hipMalloc failed: %s", hipGetErrorString(e)); status = RT_ERR_DEVICE;
        }
        std::vector<u32> slice_of(n_total);
        for (size_t i = 0; i < c->slices.size() && status == RT_OK; i++) {
            const TranslucentSlice& sl = c->slices[i];
            if (!sl.count) continue;
            for (u32 k = 0; k < sl.count; k++) slice_of[sl.offset + k] = (u32) i;
            if (hipMemcpyAsync(c->d_sp_read + sl.offset, d_points[i], sizeof(RtSamplePoint) * sl.count, hipMemcpyDeviceToDevice, c->stream) != hipSuccess ||
                hipMemcpyAsync(c->d_sp_write + sl.offset, d_points[i], sizeof(RtSamplePoint) * sl.count, hipMemcpyDeviceToDevice, c->stream) != hipSuccess || // clone, raytracing.cpp:607-621
                hipMemcpyAsync(c->d_sp_normal + 3 * (size_t) sl.offset, d_normals[i], 12 * (size_t) sl.count, hipMemcpyDeviceToDevice, c->stream) != hipSuccess) {
                set_error("sample point copy failed"); status = RT_ERR_DEVICE;
            }
        }
        if (status == RT_OK && hipMemcpyAsync(c->d_sp_slice, slice_of.data(), 4 * (size_t) n_total, hipMemcpyHostToDevice, c->stream) != hipSuccess) status = RT_ERR_DEVICE;
        if (status == RT_OK) (void) hipStreamSynchronize(c->stream);
    }
    if (status == RT_OK && !c->slices.empty()) {
        if (hipMalloc(&c->d_slices, sizeof(TranslucentSlice) * c->slices.size()) != hipSuccess ||
            hipMalloc(&c->d_slot_to_ti, 4 * (size_t) std::max<u32>(c->slot_count, 1)) != hipSuccess ||
            hipMemcpy(c->d_slices, c->slices.data(), sizeof(TranslucentSlice) * c->slices.size(), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(c->d_slot_to_ti, slot_to_ti.data(), 4 * (size_t) c->slot_count, hipMemcpyHostToDevice) != hipSuccess) {
            set_error("translucent table upload failed"); status = RT_ERR_DEVICE;
        }
    }
    for (auto p : d_points) if (p) (void) hipFree(p);
    for (auto p : d_normals) if (p) (void) hipFree(p);
    if (status != RT_OK) { free_sample_points(c); return status; }
    *total = n_total;
    c->bluenoise_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
    c->bluenoise_points = n_total;
    int hrc = hg_build_all(c); if (hrc) return hrc;
    return ensure_path_buffers(c);
}

namespace {
// HG-1..HG-4: Morton order of one instance's points, midpoint tree in preorder, leaves <= 8 points
u32 hg_spread(u32 x) {
    x &= 0x3ffu; x = (x | (x << 16)) & 0x030000ffu; x = (x | (x << 8)) & 0x0300f00fu;
    x = (x | (x << 4)) & 0x030c30c3u; x = (x | (x << 2)) & 0x09249249u;
    return x;
}
struct HgBuilder {
    const RtSamplePoint* pts; const u32* order; std::vector<HgNode>* nodes; u32 max_depth;
    u32 build(u32 first, u32 count, u32 depth) {
        u32 id = (u32) nodes->size();
        nodes->push_back(HgNode());
        if (depth > max_depth) max_depth = depth;
        double sum[3] = { 0, 0, 0 };
        for (u32 j = 0; j < count; j++) for (int a = 0; a < 3; a++) sum[a] += (double) pts[order[first + j]].position[a];
        HgNode n;
        n.cx = (float) (sum[0] / (double) count); n.cy = (float) (sum[1] / (double) count); n.cz = (float) (sum[2] / (double) count);
        n.radius2 = 0.0f;
        for (u32 j = 0; j < count; j++) {
            const float* p = pts[order[first + j]].position;
            float dx = p[0] - n.cx, dy = p[1] - n.cy, dz = p[2] - n.cz;
            float d2 = dx * dx + dy * dy + dz * dz;
            if (d2 > n.radius2) n.radius2 = d2;
        }
        n.first = first; n.count = count; n.right = 0; n.depth = depth;
        if (count > 8) {
            u32 half = (count + 1) / 2;
            build(first, half, depth + 1);
            n.right = build(first + half, count - half, depth + 1);
        }
        (*nodes)[id] = n;
        return id;
    }
};

int hg_build_all(rt_ctx* c) {
    c->hg_built = false;
    if (c->n_points == 0) return RT_OK;
    std::vector<RtSamplePoint> pts(c->n_points);
    RT_HIP(hipMemcpy(pts.data(), c->d_sp_read, sizeof(RtSamplePoint) * c->n_points, hipMemcpyDeviceToHost));
    std::vector<HgNode> nodes; std::vector<HgSlice> hs(c->slices.size());
    std::vector<u32> order(c->n_points); std::vector<float4> pos(c->n_points);
    for (size_t si = 0; si < c->slices.size(); si++) {
        const TranslucentSlice& sl = c->slices[si];
        hs[si].node_offset = (u32) nodes.size(); hs[si].node_count = 0; hs[si].max_depth = 0; hs[si].pad = 0;
        if (!sl.count) continue;
        const RtSamplePoint* p = pts.data() + sl.offset;
        float lo[3], hi[3];
        for (int a = 0; a < 3; a++) { lo[a] = p[0].position[a]; hi[a] = p[0].position[a]; }
        for (u32 k = 1; k < sl.count; k++) for (int a = 0; a < 3; a++) {
            lo[a] = fminf(lo[a], p[k].position[a]); hi[a] = fmaxf(hi[a], p[k].position[a]);
        }
        std::vector<std::pair<u32, u32>> keyed(sl.count);
        for (u32 k = 0; k < sl.count; k++) {
            u32 q[3];
            for (int a = 0; a < 3; a++) {
                float e = hi[a] - lo[a];
                q[a] = e > 0.0f ? (u32) fminf(fmaxf((p[k].position[a] - lo[a]) / e * 1024.0f, 0.0f), 1023.0f) : 0u;
            }
            keyed[k] = std::make_pair(hg_spread(q[0]) | (hg_spread(q[1]) << 1) | (hg_spread(q[2]) << 2), k);
        }
        std::sort(keyed.begin(), keyed.end());
        u32* ord = order.data() + sl.offset;
        for (u32 k = 0; k < sl.count; k++) {
            ord[k] = keyed[k].second;
            const float* q = p[ord[k]].position;
            pos[sl.offset + k] = make_float4(q[0], q[1], q[2], 0.0f);
        }
        std::vector<HgNode> local;
        HgBuilder b { p, ord, &local, 0 };
        b.build(0, sl.count, 0);
        if (b.max_depth > 22) { set_error("hierarchical gather: tree of instance %zu is %u levels deep", si, b.max_depth); return RT_ERR_LIMIT; }
        hs[si].node_count = (u32) local.size(); hs[si].max_depth = b.max_depth;
        nodes.insert(nodes.end(), local.begin(), local.end());
    }
    if (nodes.empty()) return RT_OK;
    RT_HIP(hipMalloc(&c->d_hg_nodes, sizeof(HgNode) * nodes.size()));
    RT_HIP(hipMalloc(&c->d_hg_sums, sizeof(float4) * nodes.size()));
    RT_HIP(hipMalloc(&c->d_hg_pos, sizeof(float4) * c->n_points));
    RT_HIP(hipMalloc(&c->d_hg_pay, sizeof(float4) * c->n_points));
    RT_HIP(hipMalloc(&c->d_hg_order, 4 * (size_t) c->n_points));
    RT_HIP(hipMalloc(&c->d_hg_slices, sizeof(HgSlice) * hs.size()));
    RT_HIP(hipMemcpy(c->d_hg_nodes, nodes.data(), sizeof(HgNode) * nodes.size(), hipMemcpyHostToDevice));
    RT_HIP(hipMemset(c->d_hg_sums, 0, sizeof(float4) * nodes.size()));
    RT_HIP(hipMemcpy(c->d_hg_pos, pos.data(), sizeof(float4) * c->n_points, hipMemcpyHostToDevice));
    RT_HIP(hipMemset(c->d_hg_pay, 0, sizeof(float4) * c->n_points));
    RT_HIP(hipMemcpy(c->d_hg_order, order.data(), 4 * (size_t) c->n_points, hipMemcpyHostToDevice));
    RT_HIP(hipMemcpy(c->d_hg_slices, hs.data(), sizeof(HgSlice) * hs.size(), hipMemcpyHostToDevice));
    RT_HIP(hipMalloc(&c->d_sp_pay, sizeof(float4) * c->n_points));
    { int prc = launch_sort_payload(c->stream, c->d_sp_read, c->d_hg_order, c->d_sp_slice, c->d_slices, c->n_points, c->d_sp_pay); if (prc) return prc; }
    RT_HIP(hipStreamSynchronize(c->stream));
    c->hg_built = true;
    return RT_OK;
}
} // namespace

int rt_set_gather_mode(rt_ctx* c, int hierarchical, float max_solid_angle) {
    if (!c || (hierarchical && !(max_solid_angle >= 0.0f))) { set_error("rt_set_gather_mode: bad arguments"); return RT_ERR_INVALID; }
    c->hg_enabled = hierarchical != 0; c->hg_max_solid_angle = max_solid_angle;
    return RT_OK;
}

int rt_set_globals(rt_ctx* c, const RtGlobals* g) {
    if (!c || !g) return RT_ERR_INVALID;
    u32 stride = c->g.translucent_instance_stride;
    c->g = *g;
    c->g.translucent_instance_stride = stride;
    return RT_OK;
}
int rt_get_globals(rt_ctx* c, RtGlobals* g) { if (!c || !g) return RT_ERR_INVALID; *g = c->g; return RT_OK; }
int rt_set_flags(rt_ctx* c, uint32_t flags) { if (!c) return RT_ERR_INVALID; c->flags = flags; return RT_OK; }

int rt_set_tile_split(rt_ctx* c, uint32_t rank, uint32_t world) {
    if (!c || world == 0 || rank >= world) { set_error("rt_set_tile_split: bad rank/world"); return RT_ERR_INVALID; }
    c->tile_rank = rank; c->tile_world = world;
    return RT_OK;
}

namespace {
int validate_dispatch(rt_ctx* c) {
    if (!c) return RT_ERR_INVALID;
    if (c->width == 0 || c->n_instances == 0) { set_error("rt_dispatch_rays: needs rt_update_resolution and rt_build_tlas first"); return RT_ERR_STATE; }
    if (c->g.samples_per_pixel == 0 || c->g.bounces_per_sample > 64) { set_error("rt_dispatch_rays: samples_per_pixel must be >= 1 and bounces <= 64"); return RT_ERR_INVALID; }
    if (c->g.translucent_bssrdf_scale != 0.0f && c->n_points > 0 && !c->d_table[c->profile]) {
        set_error("rt_dispatch_rays: tabulated BSSRDF selected but profile %d was never loaded", c->profile); return RT_ERR_STATE;
    }
    return RT_OK;
}

// the current lane's read copy of the sample points <- d_sp_write (raytracing.cpp:723-728), and their payloads in Morton order
int refresh_read_points(rt_ctx* c) {
    RT_HIP(hipMemcpyAsync(c->d_sp_read, c->d_sp_write, sizeof(RtSamplePoint) * c->n_points, hipMemcpyDeviceToDevice, c->stream));
    if (c->hg_built && c->d_sp_pay) return launch_sort_payload(c->stream, c->d_sp_write, c->d_hg_order, c->d_sp_slice, c->d_slices, c->n_points, c->d_sp_pay);
    return RT_OK;
}

// translucent_rgen (raytracing.cpp:677-687) on c->stream, and -- multi-GPU with a partitioned pass -- the caller's exchange of the
// payload blocks, enqueued on the same stream right behind it: everything that reads the points afterwards is ordered after both
int sample_pass(rt_ctx* c) {
    int rc = run_pass(c, true); if (rc) return rc;
    if (c->probe_exchange) {
        const u32 block = (c->n_points + c->probe_world - 1) / c->probe_world;
        const int xrc = c->probe_exchange(c->probe_exchange_user, (void*) c->stream, (void*) c->d_sp_write, c->n_points, block);
        if (xrc) { set_error("rt_dispatch_rays: the sample-point exchange callback failed with %d", xrc); return RT_ERR_DEVICE; }
    }
    return RT_OK;
}

// one frame on c->stream with the context's current set of path buffers, nothing else in flight (the order of
// raytracing.cpp:652-747)
int dispatch_serial(rt_ctx* c) {
    const bool collect = (c->flags & RT_FLAG_COLLECT_SAMPLES) && c->n_points > 0;
    int rc = RT_OK;
    if (c->n_points && c->sp_read_version[0] != c->sp_version) { // the last pipelined frame that collected samples ran on lane 1
        { int prc = refresh_read_points(c); if (prc) return prc; }
        c->sp_read_version[0] = c->sp_version;
    }
    auto refresh_hg = [&]() -> int { // node sums of the payloads the camera pass is about to read
        if (!(c->hg_enabled && c->hg_built)) return RT_OK;
        PassArgs a; fill_pass_args(c, a, false);
        return launch_hg_sums(c->stream, a, (u32) c->slices.size());
    };
    if (collect) { rc = sample_pass(c); if (rc) return rc; }                          // raytracing.cpp:677-687
    if (c->flags & RT_FLAG_DOUBLE_DISPATCH) { rc = refresh_hg(); if (rc) return rc; }
    if (c->flags & RT_FLAG_DOUBLE_DISPATCH) { rc = run_pass(c, false); if (rc) return rc; } // :688-695
    if (collect) {
        { int prc = refresh_read_points(c); if (prc) return prc; } // :723-728
        c->sp_read_version[0] = ++c->sp_version;
    }
    rc = refresh_hg(); if (rc) return rc;
    rc = run_pass(c, false); if (rc) return rc;                                       // :734-741
    c->dispatch_index++;
    c->g.accumulator_count += 1;                                                      // :744
    c->g.translucent_accumulator_count += (c->flags & RT_FLAG_COLLECT_SAMPLES) ? 1u : 0u; // :745
    c->render_target_stale = c->last_pass_lazy; c->render_target_frames = c->g.accumulator_count;
    return RT_OK;
}
} // namespace

namespace {
// the table of texel differences k_gather<3> reads belongs to one (profile table, fudge / scale^2) pair: rebuilt when the
// globals or the profile change, with nothing in flight (both lanes may be reading the old one)
int ensure_gather_delta(rt_ctx* c) {
    const float scale = c->g.translucent_bssrdf_scale;
    const float* table = c->d_table[c->profile]; const int n = c->table_texels[c->profile];
    if (!c->gather_l1 || scale == 0.0f || !table || n > 2048 || c->n_points == 0) return RT_OK;
    const float factor = ((c->flags & RT_FLAG_ENABLE_SSS) ? c->g.translucent_bssrdf_fudge : 0.0f) / (scale * scale);
    if (c->d_gather_delta && c->delta_table == table && c->delta_texels == n && c->delta_factor == factor) return RT_OK;
    { int rc = quiesce(c); if (rc) return rc; }
    RT_HIP(hipStreamSynchronize(c->stream));
    if (!c->d_gather_delta) RT_HIP(hipMalloc(&c->d_gather_delta, sizeof(float4) * (2048 + 4)));
    int rc = launch_gather_delta(c->stream, table, n, factor, c->d_gather_delta); if (rc) return rc;
    RT_HIP(hipStreamSynchronize(c->stream));
    c->delta_table = table; c->delta_texels = n; c->delta_factor = factor;
    return RT_OK;
}
} // namespace

int rt_dispatch_rays(rt_ctx* c) {
    { int vrc = validate_dispatch(c); if (vrc) return vrc; }
    RT_HIP(hipSetDevice(c->device));
    { int grc = ensure_gather_delta(c); if (grc) return grc; }
    bool collect = (c->flags & RT_FLAG_COLLECT_SAMPLES) && c->n_points > 0;
    const bool gather = c->n_points > 0 && (c->flags & RT_FLAG_ENABLE_SSS) && c->g.translucent_bssrdf_fudge != 0.0f;
    // Frames are pipelined over the two lanes.  A frame touches its lane's path buffers and copy of the sample points,
    // and three things it shares with its neighbours, each in frame order: the accumulator and the render target (the
    // frame's resolve kernel, after the resolve of the frame before: run_pass), and d_sp_write (the sample-point pass
    // and the copy out of it, after those of the frame before: probe_done).
    const bool hierarchical = c->hg_enabled && c->hg_built && c->n_points > 0; // node sums are shared state: one frame at a time
    if (c->frames_in_flight > 1 && c->stream == c->own_stream && !hierarchical && c->persistent <= 0 &&
        !(c->flags & (RT_FLAG_DOUBLE_DISPATCH | RT_FLAG_CAPTURE_HITS | RT_FLAG_COUNT_TRAVERSAL))) {
        if (!c->pipelined) { // everything enqueued so far is finished before lane 1 starts beside it
            RT_HIP(hipStreamSynchronize(c->stream));
            int rc = ensure_path_buffers(c); if (rc) return rc;
            rc = ensure_second_lane(c); if (rc) return rc;
            if (c->gather_side && gather) { rc = ensure_gather_side(c, 2); if (rc) return rc; }
            c->pipelined = true; c->next_lane = 0; c->have_previous = false; c->last_probe_lane = -1; c->lane_ran[0] = c->lane_ran[1] = false; c->probe_wait = nullptr;
        }
        const int lane = c->next_lane;
        if (c->gather_side && gather && !c->gather_stream[1]) { int grc = ensure_gather_side(c, 2); if (grc) return grc; } // (the gather came on after the lanes were made)
        c->cur_lane = lane;
        auto swap_lane = [&]() { if (lane) { swap_path_buffers(c); std::swap(c->stream, c->alt_stream); std::swap(c->d_sp_read, c->d_sp_read_alt); std::swap(c->d_sp_pay, c->d_sp_pay_alt); } };
        swap_lane();
        int rc = RT_OK;
        auto hip_ok = [&](hipError_t e, const char* what) { if (e != hipSuccess && rc == RT_OK) { set_error("rt_dispatch_rays: %s failed: %s", what, hipGetErrorString(e)); rc = RT_ERR_DEVICE; } };
        const bool side = c->probe_side && c->gather_batch && gather && c->n_points > 0;
        if (side && (collect || c->sp_read_version[lane] != c->sp_version)) {
            // the pass on the probe stream, with its own path buffers; the stream keeps the passes of consecutive frames in order.  It
            // writes this lane's copy of the points, which the gather of the lane's frame before may still read
            rc = ensure_probe_side(c);
            if (rc == RT_OK) {
                swap_probe_buffers(c); std::swap(c->stream, c->probe_stream);
                if (c->lane_ran[lane]) hip_ok(hipStreamWaitEvent(c->stream, c->lane_done[lane], 0), "event wait");
                if (c->last_probe_lane >= 0 && !c->last_probe_side) hip_ok(hipStreamWaitEvent(c->stream, c->probe_done[c->last_probe_lane], 0), "event wait"); // (the pass before ran on its lane)
                if (rc == RT_OK && collect) { rc = sample_pass(c); if (rc == RT_OK) c->sp_version++; } // raytracing.cpp:677-687
                if (rc == RT_OK) rc = refresh_read_points(c); // :723-728
                if (rc == RT_OK) { c->sp_read_version[lane] = c->sp_version; hip_ok(hipEventRecord(c->probe_done[lane], c->stream), "event record"); c->last_probe_lane = lane; c->last_probe_side = true; }
                std::swap(c->stream, c->probe_stream); swap_probe_buffers(c);
                if (rc == RT_OK) c->probe_wait = c->probe_done[lane];
            }
        }
        else if (collect || (gather && c->sp_read_version[lane] != c->sp_version)) {
            if (c->last_probe_lane >= 0 && (c->last_probe_lane != lane || c->last_probe_side)) hip_ok(hipStreamWaitEvent(c->stream, c->probe_done[c->last_probe_lane], 0), "event wait");
            if (rc == RT_OK && collect) { rc = sample_pass(c); if (rc == RT_OK) c->sp_version++; } // raytracing.cpp:677-687
            if (rc == RT_OK) rc = refresh_read_points(c); // :723-728
            if (rc == RT_OK) { c->sp_read_version[lane] = c->sp_version; hip_ok(hipEventRecord(c->probe_done[lane], c->stream), "event record"); c->last_probe_lane = lane; c->last_probe_side = false; }
        }
        if (rc == RT_OK) {
            c->accumulate_after = c->have_previous ? c->lane_done[lane ^ 1] : nullptr;
            c->defer_resolve = true;
            rc = run_pass(c, false);                                                  // :734-741
            c->defer_resolve = false; c->accumulate_after = nullptr;
        }
        if (rc == RT_OK && c->probe_wait) { hip_ok(hipStreamWaitEvent(c->stream, c->probe_wait, 0), "event wait"); c->probe_wait = nullptr; } // (a frame without a gather launch: the lane's end still means its points are written)
        if (rc == RT_OK) { hip_ok(hipEventRecord(c->lane_done[lane], c->stream), "event record"); c->lane_ran[lane] = true; }
        swap_lane();
        c->cur_lane = 0;
        if (rc) return rc;
        c->next_lane ^= 1; c->have_previous = true;
        c->dispatch_index++;
        c->g.accumulator_count += 1;                                                  // :744
        c->g.translucent_accumulator_count += (c->flags & RT_FLAG_COLLECT_SAMPLES) ? 1u : 0u; // :745
        c->render_target_stale = true; c->render_target_frames = c->g.accumulator_count;
        return RT_OK;
    }
    int rc = quiesce(c); if (rc) return rc;
    rc = ensure_path_buffers(c); if (rc) return rc;
    if (c->gather_side && gather && c->stream == c->own_stream) { rc = ensure_gather_side(c, 1); if (rc) return rc; }
    c->cur_lane = 0;
    return dispatch_serial(c);
}

int rt_dispatch_rays_static(rt_ctx* c) {
    { int vrc = validate_dispatch(c); if (vrc) return vrc; }
    if (!c->d_static_scratch) { set_error("rt_dispatch_rays_static: the static scratch was not resolved at rt_create"); return RT_ERR_STATE; }
    if (c->flags & (RT_FLAG_CAPTURE_HITS | RT_FLAG_TIME_KERNELS)) {
        set_error("rt_dispatch_rays_static: RT_FLAG_CAPTURE_HITS and RT_FLAG_TIME_KERNELS allocate (capture buffer, events); use rt_dispatch_rays");
        return RT_ERR_UNSUPPORTED;
    }
    const u32 tiles = ((c->width + 31) / 32) * ((c->height + 31) / 32);
    const u32 segs = std::max(std::max(tiles, (c->n_points + 1023) / 1024), 1u);
    if (segs > RT_STATIC_MAX_SEGMENTS) {
        set_error("rt_dispatch_rays_static: %ux%u pixels / %u sample points need %u queue segments, the static scratch holds %u "
                  "(at most 64 tiles of 32x32 pixels and 65,536 sample points); use rt_dispatch_rays", c->width, c->height, c->n_points, segs, RT_STATIC_MAX_SEGMENTS);
        return RT_ERR_LIMIT;
    }
    RT_HIP(hipSetDevice(c->device));
    int rc = quiesce(c); if (rc) return rc; // a no-op unless rt_dispatch_rays left frames in flight on the context's own streams
    // the context's own buffers step aside for this call
    rt_ctx::PathBuffers own; memset(&own, 0, sizeof own);
    const u32 own_paths = c->n_paths, own_cap = c->n_segments_cap;
    auto exchange = [&]() {
        std::swap(c->d_arena, own.d_arena); std::swap(c->d_seg_count, own.d_seg_count); std::swap(c->hits, own.hits); std::swap(c->ps, own.ps);
        std::swap(c->gq, own.gq); std::swap(c->d_stack_spill, own.d_stack_spill);
        for (int k = 0; k < 2; k++) { std::swap(c->q[k], own.q[k]); std::swap(c->st[k], own.st[k]); }
    };
    exchange();
    unsigned char* base = (unsigned char*) c->d_static_scratch;
    carve_path_buffers(c, (float*) (base + RT_STATIC_SEG_BYTES), (u32*) base, RT_STATIC_MAX_SEGMENTS);
    // the array keeps the last call's contents: the segment fills start at zero like freshly allocated ones
    hipError_t e = hipMemsetAsync(base, 0, RT_STATIC_SEG_BYTES, c->stream);
    if (e != hipSuccess) { set_error("rt_dispatch_rays_static: hipMemsetAsync failed: %s", hipGetErrorString(e)); rc = RT_ERR_DEVICE; }
    c->no_alloc_call = true; // (the batched gather would allocate its rows)
    if (rc == RT_OK) rc = dispatch_serial(c);
    c->no_alloc_call = false;
    exchange();
    c->n_paths = own_paths; c->n_segments_cap = own_cap;
    return rc;
}

int rt_sync(rt_ctx* c) {
    if (!c) return RT_ERR_INVALID;
    { int rc = quiesce(c); if (rc) return rc; }
    RT_HIP(hipStreamSynchronize(c->stream));
    return check_gather_overflow(c);
}

int rt_read_accumulator(rt_ctx* c, float* rgba) {
    if (!c || !rgba || !c->d_accumulator) return RT_ERR_INVALID;
    { int qrc = quiesce(c); if (qrc) return qrc; }
    RT_HIP(hipMemcpyAsync(rgba, c->d_accumulator, (size_t) c->width * c->height * 16, hipMemcpyDeviceToHost, c->stream));
    RT_HIP(hipStreamSynchronize(c->stream));
    return check_gather_overflow(c);
}
int rt_read_render_target(rt_ctx* c, uint8_t* rgba8) {
    if (!c || !rgba8 || !c->d_render_target) return RT_ERR_INVALID;
    { int qrc = quiesce(c); if (qrc) return qrc; }
    if (c->render_target_stale) {
        PassArgs a; fill_pass_args(c, a, false);
        a.g.accumulator_count = c->render_target_frames;
        int rc = launch_resolve_only(c->stream, a); if (rc) return rc;
        c->render_target_stale = false;
    }
    RT_HIP(hipMemcpyAsync(rgba8, c->d_render_target, (size_t) c->width * c->height * 4, hipMemcpyDeviceToHost, c->stream));
    RT_HIP(hipStreamSynchronize(c->stream));
    return RT_OK;
}
int rt_read_hit_records(rt_ctx* c, uint32_t bounce, RtHit* out) {
    if (!c || !out || !c->d_capture || bounce >= c->capture_bounces) { set_error("rt_read_hit_records: nothing captured for bounce %u", bounce); return RT_ERR_INVALID; }
    { int qrc = quiesce(c); if (qrc) return qrc; }
    size_t n = (size_t) c->width * c->height;
    RT_HIP(hipMemcpyAsync(out, c->d_capture + bounce * n, n * sizeof(RtHit), hipMemcpyDeviceToHost, c->stream));
    RT_HIP(hipStreamSynchronize(c->stream));
    return RT_OK;
}
int rt_translucent_instance_count(rt_ctx* c, uint32_t* n) { if (!c || !n) return RT_ERR_INVALID; *n = (u32) c->translucent_instances.size(); return RT_OK; }
int rt_sample_point_count(rt_ctx* c, uint32_t instance, uint32_t* n) {
    if (!c || !n || instance >= c->translucent_instances.size()) return RT_ERR_INVALID;
    *n = c->translucent_instances[instance].samples_count; return RT_OK;
}
int rt_read_sample_points(rt_ctx* c, uint32_t instance, RtSamplePoint* points, float* normals) {
    if (!c || instance >= c->slices.size()) return RT_ERR_INVALID;
    { int qrc = quiesce(c); if (qrc) return qrc; }
    const TranslucentSlice& sl = c->slices[instance];
    if (points) RT_HIP(hipMemcpyAsync(points, c->d_sp_write + sl.offset, /* == the read copy between dispatches; lane 0's may be one pipelined frame behind */ sizeof(RtSamplePoint) * sl.count, hipMemcpyDeviceToHost, c->stream));
    if (normals) RT_HIP(hipMemcpyAsync(normals, c->d_sp_normal + 3 * (size_t) sl.offset, 12 * (size_t) sl.count, hipMemcpyDeviceToHost, c->stream));
    RT_HIP(hipStreamSynchronize(c->stream));
    return RT_OK;
}

int rt_get_stats(rt_ctx* c, RtStats* out) {
    if (!c || !out) return RT_ERR_INVALID;
    { int qrc = quiesce(c); if (qrc) return qrc; }
    RT_HIP(hipStreamSynchronize(c->stream));
    int rc = drain_timers(c); if (rc) return rc;
    DeviceStats ds;
    RT_HIP(hipMemcpy(&ds, c->d_stats, sizeof ds, hipMemcpyDeviceToHost));
#if defined(RT_POOL_DEBUG)
    if (ds.pool_err[0]) {
        fprintf(stderr, "RT_POOL_DEBUG: pooled traversal caught violation code %llu value 0x%llx", ds.pool_err[0], ds.pool_err[1]);
        for (int k = 2; k < 8; k++) fprintf(stderr, " 0x%llx", ds.pool_err[k]);
        PassArgs a; fill_pass_args(c, a, false);
        fprintf(stderr, " | blas range 0x%llx..0x%llx tlas bytes %u\n", a.dbg_blas_lo, a.dbg_blas_hi, a.dbg_tlas_bytes);
    }
    else fprintf(stderr, "RT_POOL_DEBUG: no violation\n");
    if (ds.pool_prof[17]) {
        const unsigned long long* p = ds.pool_prof; const double w = (double) p[17];
        fprintf(stderr, "RT_POOL_DEBUG profile over %llu waves: rounds/wave %.0f | cycles per wave: scheduling %.0f barrier-wait %.0f batches NODE %.0f LEAF %.0f INST %.0f FREE %.0f | rounds without a batch %.0f | "
                        "batches NODE %llu (%.1f rays, %.2f steps) LEAF %llu (%.1f) INST %llu (%.1f) FREE %llu (%.1f)\n",
                p[17], p[0] / w, p[1] / w, p[2] / w, p[5] / w, p[6] / w, p[7] / w, p[8] / w, p[3] / w,
                p[9], p[9] ? (double) p[13] / p[9] : 0.0, p[9] ? (double) p[4] / p[9] : 0.0, p[10], p[10] ? (double) p[14] / p[10] : 0.0,
                p[11], p[11] ? (double) p[15] / p[11] : 0.0, p[12], p[12] ? (double) p[16] / p[12] : 0.0);
    }
#endif
    memset(out, 0, sizeof *out);
    out->rays_camera = ds.rays_camera + c->primary_rays; out->rays_probe = ds.rays_probe; out->gather_pairs = ds.gather_pairs;
    out->nodes_visited = ds.nodes; out->tris_tested = ds.tris; out->instances_entered = ds.insts;
    out->wave_node_steps = ds.node_trips; out->wave_tri_steps = ds.tri_trips; out->wave_instance_steps = ds.inst_trips; out->max_ray_steps = ds.max_ray_steps;
    out->traverse_launches = c->traverse_launches; out->traverse_ms = c->traverse_ms;
    out->gather_launches = c->gather_launches; out->gather_ms = c->gather_ms;
    out->traverse_launches_timed = c->traverse_launches_timed; out->gather_launches_timed = c->gather_launches_timed;
    out->primary_launches = c->primary_launches; out->primary_launches_timed = c->primary_launches_timed;
    out->primary_ms = c->traverse_primary_ms; out->rays_primary = c->primary_rays;
    out->bluenoise_ms = c->bluenoise_ms; out->bluenoise_points = c->bluenoise_points;
    out->tlas_depth = (u32) c->tlas_depth; out->max_blas_depth = (u32) c->max_blas_depth;
    out->tlas_grid_cells = (c->flat_instances || c->persistent > 0) ? 0u : c->grid_cells;
    { PassArgs a; fill_pass_args(c, a, false); out->stack_in_lds = a.stack_in_lds;
      const bool batched = c->gather_batch && c->gather_fused && !a.hg.enabled && (c->bg[0].rows || c->bg[1].rows); // frames with the batched gather run the fused pass too (run_pass)
      const bool fused_pass = (a.gather_enabled == 0u || batched) && !(c->flags & (RT_FLAG_COUNT_TRAVERSAL | RT_FLAG_CAPTURE_HITS));
      out->fused_bounce0 = (fused_pass && a.fuse_primary) ? 1u : 0u;
      out->fused_tail = (fused_pass && c->fuse_tail && a.pool_groups == 0u && c->flat_instances != 2u && c->persistent <= 0) ? 1u : 0u; }
    out->flat_depth = c->flat_instances ? (uint32_t) c->flat_info.depth : 0u;
    out->blas_build_ms = c->blas_build_ms; out->tlas_build_ms = c->tlas_build_ms;
    out->blas_triangles = c->blas_triangles; out->tlas_instances = c->n_instances;
    return RT_OK;
}
int rt_reset_stats(rt_ctx* c) {
    if (!c) return RT_ERR_INVALID;
    { int qrc = quiesce(c); if (qrc) return qrc; }
    RT_HIP(hipStreamSynchronize(c->stream));
    int rc = drain_timers(c); if (rc) return rc;
    RT_HIP(hipMemset(c->d_stats, 0, sizeof(DeviceStats)));
    c->traverse_ms = 0; c->traverse_launches = 0; c->primary_rays = 0; c->gather_ms = 0; c->gather_launches = 0;
    c->traverse_primary_ms = 0; c->primary_launches = 0; c->primary_launches_timed = 0;
    c->traverse_launches_timed = 0; c->gather_launches_timed = 0;
    return RT_OK;
}

int rt_accumulator_device_ptr(rt_ctx* c, void** dev_ptr, size_t* bytes) {
    if (!c || !dev_ptr || !bytes || !c->d_accumulator) return RT_ERR_INVALID;
    { int qrc = quiesce(c); if (qrc) return qrc; }
    *dev_ptr = c->d_accumulator; *bytes = (size_t) c->width * c->height * 16;
    return RT_OK;
}

int rt_resolve_render_target(rt_ctx* c) {
    if (!c || !c->d_accumulator) return RT_ERR_INVALID;
    { int qrc = quiesce(c); if (qrc) return qrc; }
    PassArgs a; fill_pass_args(c, a, false);
    c->render_target_stale = false;
    return launch_resolve_only(c->stream, a);
}

int rt_resolve_render_target_frames(rt_ctx* c, uint32_t total_frames) {
    if (!c || !c->d_accumulator || total_frames == 0) { set_error("rt_resolve_render_target_frames: bad arguments"); return RT_ERR_INVALID; }
    { int qrc = quiesce(c); if (qrc) return qrc; }
    PassArgs a; fill_pass_args(c, a, false);
    a.g.accumulator_count = total_frames; // the context's own count (this rank's share of the frames) stays as it is
    c->render_target_stale = false;
    return launch_resolve_only(c->stream, a);
}

namespace {
// PassArgs naming the tiles of rank `rank` of `world` (thread_pixel, rt_kernels.hip)
void tile_args(rt_ctx* c, PassArgs& a, u32 rank, u32 world) {
    fill_pass_args(c, a, false);
    const u32 n_tiles = a.tiles_x * a.tiles_y;
    a.tile_rank = rank; a.tile_world = world;
    a.n_local_tiles = n_tiles > rank ? (n_tiles - rank + world - 1) / world : 0;
}
} // namespace

int rt_tile_pack_bytes(rt_ctx* c, uint32_t world, size_t* bytes) {
    if (!c || !bytes || world == 0 || c->width == 0) { set_error("rt_tile_pack_bytes: bad arguments"); return RT_ERR_INVALID; }
    const size_t n_tiles = (size_t) ((c->width + 31) / 32) * ((c->height + 31) / 32);
    *bytes = ((n_tiles + world - 1) / world) * 1024 * 16; // the most tiles any rank owns: the same size on every rank
    return RT_OK;
}

int rt_pack_tiles(rt_ctx* c, uint32_t rank, uint32_t world, void* dev_packed) {
    if (!c || !dev_packed || world == 0 || rank >= world || !c->d_accumulator) { set_error("rt_pack_tiles: bad arguments"); return RT_ERR_INVALID; }
    RT_HIP(hipSetDevice(c->device));
    { int qrc = quiesce(c); if (qrc) return qrc; }
    PassArgs a; tile_args(c, a, rank, world);
    size_t bytes = 0; (void) rt_tile_pack_bytes(c, world, &bytes);
    if ((size_t) a.n_local_tiles * 1024 * 16 < bytes) // a rank with one tile less than the largest share: the tail is defined too
        RT_HIP(hipMemsetAsync((char*) dev_packed + (size_t) a.n_local_tiles * 1024 * 16, 0, bytes - (size_t) a.n_local_tiles * 1024 * 16, c->stream));
    return launch_pack_tiles(c->stream, a, dev_packed);
}

int rt_set_probe_split(rt_ctx* c, uint32_t rank, uint32_t world) {
    if (!c || world == 0 || rank >= world) { set_error("rt_set_probe_split: bad arguments"); return RT_ERR_INVALID; }
    { int qrc = quiesce(c); if (qrc) return qrc; }
    c->probe_rank = rank; c->probe_world = world;
    return RT_OK;
}

int rt_set_probe_exchange(rt_ctx* c, rt_probe_exchange_fn fn, void* user) {
    if (!c) return RT_ERR_INVALID;
    { int qrc = quiesce(c); if (qrc) return qrc; }
    c->probe_exchange = fn; c->probe_exchange_user = user;
    return RT_OK;
}

int rt_unpack_tiles(rt_ctx* c, uint32_t rank, uint32_t world, const void* dev_packed) {
    if (!c || !dev_packed || world == 0 || rank >= world || !c->d_accumulator) { set_error("rt_unpack_tiles: bad arguments"); return RT_ERR_INVALID; }
    RT_HIP(hipSetDevice(c->device));
    { int qrc = quiesce(c); if (qrc) return qrc; }
    PassArgs a; tile_args(c, a, rank, world);
    c->render_target_stale = true; c->render_target_frames = c->g.accumulator_count;
    return launch_unpack_tiles(c->stream, a, dev_packed);
}

} // extern "C"
