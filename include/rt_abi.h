/* rt_abi.h -- C-ABI of librt_mi355x.so, the MI355X (gfx950) replacement for the per-ray hot path of
 * gluyas/raytracer.
 *
 * What it replaces (all citations are file:line in the reference tree):
 *   - the GPU shaders  src/raytracing.hlsl:60-349, src/random.hlsl:6-88, src/bluenoise.hlsl:44-245
 *   - the DXR driver work those shaders call (TraceRay raytracing.hlsl:108-112, the BLAS/TLAS builds
 *     raytracing.cpp:411-436,523-548, the 1-D texture fetch raytracing.hlsl:275)
 *   - behind the host interface of src/raytracing.h:52-87 and src/bluenoise.h:69-88.
 *
 * Conventions (readme.md:15-17): row-major 4x4 matrices, row vectors premultiplied (v' = v * M, translation in
 * row 3), right-handed, world Z-up, camera looks down -Z.
 *
 * Every entry point returns an int status (RT_OK == 0) instead of the reference's abort() (prelude.h:185).
 * All work of a context takes effect in call order, as if on one HIP stream; calls are asynchronous unless stated
 * ("blocks").  On the context's own stream consecutive rt_dispatch_rays calls overlap on the GPU (two frames
 * in flight; image and sample points are updated in call order: same results); on a stream given with rt_set_stream everything runs on that stream.
 * The context is not thread-safe (the reference keeps this state in namespace globals, raytracing.cpp:11-71).
 * Plain pointers and sizes only: no torch / HIP types appear in any signature.
 */
#ifndef RT_ABI_H
#define RT_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: RtStats as it stands below (it grew after version 1 shipped), rt_dispatch_rays_static, rt_resolve_render_target_frames,
 *    rt_tile_pack_bytes / rt_pack_tiles / rt_unpack_tiles.  A caller compares rt_abi_version() with the RT_ABI_VERSION it was
 *    built against before it passes any struct (raytracer_amd/api.py and host/raytracing.hpp do). */
/* 3: rt_set_probe_split, rt_set_probe_exchange (no struct changed). */
#define RT_ABI_VERSION 3

/* ---- status codes ------------------------------------------------------------------------------------- */
enum {
    RT_OK              = 0,
    RT_ERR_INVALID     = 1, /* bad argument (null pointer, bad handle, bad size)                          */
    RT_ERR_DEVICE      = 2, /* a HIP call failed; rt_last_error() has the text                            */
    RT_ERR_STATE       = 3, /* call made in the wrong order (e.g. dispatch before build_tlas)             */
    RT_ERR_LIMIT       = 4, /* a reference limit was exceeded (e.g. > 65535 vertices, raytracing.cpp:367) */
    RT_ERR_UNSUPPORTED = 5
};

/* ---- parameter structs: same layout as the reference (prelude.h:96-148) -------------------------------- */
#pragma pack(push, 4)

typedef struct RtVertex {          /* prelude.h:96-99, 24 B */
    float position[3];
    float normal[3];
} RtVertex;

typedef uint16_t RtIndex;          /* prelude.h:66 */

typedef struct RtSamplePoint {     /* prelude.h:101-104, 24 B */
    float position[3];
    float payload[3];              /* incident flux */
} RtSamplePoint;

typedef struct RtGlobals {         /* RaytracingGlobals, prelude.h:110-139, 156 B */
    uint32_t frame_rng;                      /*   0 */
    uint32_t accumulator_count;              /*   4 */
    uint32_t samples_per_pixel;              /*   8 */
    uint32_t bounces_per_sample;             /*  12 */
    float    camera_to_world[16];            /*  16  row-major, row 3 = camera position */
    float    camera_aspect;                  /*  80 */
    float    camera_focal_length;            /*  84 */
    uint32_t translucent_accumulator_count;  /*  88 */
    uint32_t translucent_instance_stride;    /*  92  (written by rt_build_tlas, raytracing.cpp:461,491) */
    int32_t  translucent_emission_bounces;   /*  96 */
    float    translucent_refractive_index;   /* 100 */
    float    translucent_bssrdf_scale;       /* 104  0 => dipole, else tabulated (raytracing.hlsl:327) */
    float    translucent_bssrdf_fudge;       /* 108 */
    float    translucent_scattering[3];      /* 112 */
    uint32_t _pad124;
    float    translucent_absorption[3];      /* 128 */
    uint32_t _pad140;
    float    light_color[3];                 /* 144 */
} RtGlobals;

typedef struct RtLocals {          /* RaytracingLocals, prelude.h:141-144 */
    float   color[3];
    int32_t translucent_id;
} RtLocals;

#pragma pack(pop)

/* Shader enum, raytracing.h:19-25 */
enum { RT_SHADER_LAMBERT = 0, RT_SHADER_LIGHT = 1, RT_SHADER_TRANSLUCENT = 2, RT_SHADER_COUNT = 3 };

/* GeometryInstance + Material, raytracing.h:27-36.  Arrays are borrowed for the duration of rt_build_blas only
 * (the reference frees them right after the call, main.cpp:344-346). */
typedef struct RtGeometry {
    const RtVertex* vertices;
    size_t          vertex_count;
    const RtIndex*  indices;       /* triangle list, indices local to `vertices` */
    size_t          index_count;
    int32_t         shader;        /* RT_SHADER_* */
    float           color[3];
} RtGeometry;

typedef int32_t rt_blas;           /* handle returned by rt_build_blas (Blas, raytracing.h:38-45) */

/* BlasInstance, raytracing.h:47-50 */
typedef struct RtInstance {
    float   transform[16];         /* object->world, row-major, row vectors */
    rt_blas blas;
} RtInstance;

/* Parity tap: one record per (path, bounce) of the first sample of the last camera dispatch. */
typedef struct RtHit {
    int32_t  instance;             /* index into the rt_build_tlas array, -1 = miss                         */
    int32_t  geometry;             /* GeometryIndex() within the BLAS                                       */
    int32_t  primitive;            /* PrimitiveIndex() within the geometry                                  */
    float    t;                    /* RayTCurrent(); +inf on a miss                                         */
    float    u, v;                 /* attr.barycentrics (weights of vertex 1 and 2)                         */
    uint32_t rng_after;            /* payload.rng after the hit shader ran                                  */
    uint32_t valid;                /* 0 = the path was not alive at this bounce                             */
} RtHit;

typedef struct RtStats {
    uint64_t rays_camera;          /* TraceRay-equivalents issued by camera dispatches since rt_reset_stats */
    uint64_t rays_probe;           /* ... by translucent_rgen dispatches                                    */
    uint64_t gather_pairs;         /* (hit, sample point) pairs evaluated by translucent_chit               */
    uint64_t nodes_visited;        /* only counted while rt_set_flags(..., RT_FLAG_COUNT_TRAVERSAL) is set  */
    uint64_t tris_tested;
    uint64_t instances_entered;
    uint64_t traverse_launches;
    double   traverse_ms;          /* HIP-event time of the traverse kernels (only with RT_FLAG_TIME_KERNELS) */
    double   blas_build_ms;        /* HIP-event time of the last rt_build_blas                              */
    double   tlas_build_ms;
    uint64_t blas_triangles;       /* triangles in all BLASes                                               */
    uint64_t tlas_instances;
    uint64_t wave_node_steps;      /* with RT_FLAG_COUNT_TRAVERSAL: node / triangle / instance steps counted once per   */
    uint64_t wave_tri_steps;       /* 64-lane wave; lane steps / (64 * wave steps) = SIMD efficiency of the traversal    */
    uint64_t wave_instance_steps;
    uint64_t max_ray_steps;        /* RT_FLAG_COUNT_TRAVERSAL: most node + triangle + instance steps taken by one ray */
    uint64_t gather_launches;      /* launches of the BSSRDF gather kernel (translucent_chit's loop, raytracing.hlsl:323-331): one per frame */
                                   /* (or per few samples) with the batched gather, else one per gathering bounce      */
    double   gather_ms;            /* their summed duration with RT_FLAG_TIME_KERNELS (HIP events on the context's stream) */
    uint64_t traverse_launches_timed; /* launches that traverse_ms covers (all of them, or every 8th dispatch with RT_FLAG_TIME_SAMPLED) */
    uint64_t gather_launches_timed;   /* launches that gather_ms covers */
    uint64_t primary_launches;        /* the bounce-0 launches among traverse_launches: coherent rays derived from pixels; the rest */
    uint64_t primary_launches_timed;  /* are queued launches of scattered ("secondary", incoherent) or sample-point rays          */
    double   primary_ms;              /* the part of traverse_ms spent in the timed bounce-0 launches                             */
    uint64_t rays_primary;            /* the part of rays_camera traced by bounce-0 launches                                      */
    double   bluenoise_ms;            /* host wall time of the last rt_generate_translucent_samples (it blocks), all instances    */
    uint64_t bluenoise_points;        /* sample points it produced                                                                */
    uint32_t tlas_depth;              /* inner-node levels of the top-level tree and of the deepest bottom-level tree in it: the  */
    uint32_t max_blas_depth;          /* traversal stack holds at most tlas_depth + max_blas_depth + 2 entries                    */
    uint32_t stack_in_lds;            /* 1 = that fits the 17 LDS rows per lane of the 6-wave kernels, 2 = the 33 rows of the 5-wave  */
                                      /* kernels: traversal runs without the global overflow path; 0 = with it                     */
    uint32_t flat_depth;              /* > 0: the scene (<= 2 instances) is traversed in ONE tree over all instanced triangles, of   */
                                      /* this many inner-node levels, instead of the two levels; 0 = two levels */
    uint32_t fused_bounce0;           /* 1 = camera dispatches (without the BSSRDF gather, or with the batched one) walk AND shade bounce 0 in one kernel (flat one-instance */
                                      /* scenes): that launch is counted in primary_launches / primary_ms, and no separate bounce-0 shade runs   */
    uint32_t fused_tail;              /* 1 = their last bounce is shaded inside its traversal kernel                                             */
    uint32_t tlas_grid_cells;         /* > 0: the top level of a two-level scene is walked as a uniform grid of that many cells (many evenly spread   */
                                      /* instances: the 52,743-instance field), the top-level tree stays for the other kernels; RT_TLAS_GRID=0: never */
} RtStats;

/* rt_set_flags bits */
enum {
    RT_FLAG_COLLECT_SAMPLES   = 1 << 0, /* g_enable_translucent_sample_collection (raytracing.cpp:70)       */
    RT_FLAG_ENABLE_SSS        = 1 << 1, /* g_enable_subsurface_scattering          (raytracing.cpp:71)      */
    RT_FLAG_DOUBLE_DISPATCH   = 1 << 2, /* mimic the reference frame: camera_rgen twice (raytracing.cpp:688-741) */
    RT_FLAG_CAPTURE_HITS      = 1 << 3, /* record RtHit for every bounce of sample 0 (parity tap)            */
    RT_FLAG_COUNT_TRAVERSAL   = 1 << 4, /* count nodes/triangles/instances visited (slower kernel variant)   */
    RT_FLAG_TIME_KERNELS      = 1 << 5, /* bracket every traverse launch with HIP events                     */
    RT_FLAG_ALL_PHASE_GROUPS  = 1 << 6, /* blue-noise: run all 27 phase groups, not the reference's 24       */
    RT_FLAG_TIME_SAMPLED      = 1 << 7  /* with RT_FLAG_TIME_KERNELS: bracket the launches of every 8th rt_dispatch_rays only
                                           (the event packets of a fully timed 0.28 ms frame cost 13 us of it)           */
};

typedef struct rt_ctx rt_ctx;

/* ---- lifetime ------------------------------------------------------------------------------------------ */
/* Replaces Device::init + Raytracing::init + Bluenoise::init (device.cpp:17, raytracing.cpp:73, bluenoise.cpp:25).
 * One process drives one GPU: n must be 1. */
int  rt_create(const int* devices, int n, rt_ctx** out);
void rt_destroy(rt_ctx* ctx);
const char* rt_last_error(void);
int  rt_abi_version(void);

/* Run on a caller-owned HIP stream (hipStream_t passed as void*); NULL = the context's own stream. */
int  rt_set_stream(rt_ctx* ctx, void* hip_stream);

/* ---- Raytracing:: interface (raytracing.h:66-85) ------------------------------------------------------- */
/* Raytracing::update_resolution (raytracing.cpp:143-187): (re)allocates the RGBA8 target and RGBA32F accumulator. */
int  rt_update_resolution(rt_ctx* ctx, uint32_t width, uint32_t height);

/* Raytracing::build_blas (raytracing.cpp:316-444): concatenates the geometries, rebases the 16-bit indices,
 * creates one shader record per geometry and builds the bottom-level BVH on the GPU.  Blocks until the
 * host arrays have been copied. */
int  rt_build_blas(rt_ctx* ctx, const RtGeometry* geometries, int n, rt_blas* out);

/* Raytracing::build_tlas (raytracing.cpp:446-551): resets all translucent instances, assigns InstanceID per
 * BLAS in array order, sets translucent_instance_stride, builds the top-level BVH on the GPU. */
int  rt_build_tlas(rt_ctx* ctx, const RtInstance* instances, int n);

/* Raytracing::generate_translucent_samples (raytracing.cpp:553-650) + Bluenoise::generate_sample_points
 * (bluenoise.cpp:95-358).  `rng_seed` replaces rand() at bluenoise.cpp:112 (instance i uses rng_seed + i),
 * `phase_seed` replaces the rand() shuffle at bluenoise.cpp:308.  Blocks.  *total = total sample points. */
int  rt_generate_translucent_samples(rt_ctx* ctx, float radius, uint32_t rng_seed, uint32_t phase_seed,
                                     uint32_t* total);

/* Raytracing::g_globals (raytracing.h:61).  translucent_instance_stride in *g is ignored (owned by build_tlas). */
int  rt_set_globals(rt_ctx* ctx, const RtGlobals* g);
int  rt_get_globals(rt_ctx* ctx, RtGlobals* g);
int  rt_set_flags(rt_ctx* ctx, uint32_t flags);
/* The g_bssrdf 1-D texture (raytracing.cpp:111-140): n radial samples per channel (r = data_l, g = data_m,
 * b = data_s), zero-padded by the library to a multiple of 256 texels (2000 -> 2048).  profile 0 = data/skin_0.h
 * (the one the reference includes, raytracing.cpp:112), profile 1 = data/skin_1.h.  Blocks. */
int  rt_load_bssrdf_table(rt_ctx* ctx, int profile, const float* r, const float* g, const float* b, uint32_t n);
int  rt_set_bssrdf_profile(rt_ctx* ctx, int profile);
/* Hierarchical evaluation of translucent_chit's sum over the sample points (raytracing.hlsl:323-331).  The reference
 * has only the O(points) loop and lists the hierarchy as future work (todo.txt:1-2); this is SURVEY.md 8 row f4.
 * hierarchical = 0 (default): the reference's loop, point by point in index order.
 * hierarchical = 1: a tree over each instance's points (Morton order, midpoint splits, leaves of <= 8 points); a node
 * whose points lie in a sphere not containing the hit and that subtends less than max_solid_angle steradians
 * (count * mean_area < max_solid_angle * distance^2) contributes bssrdf(distance to its centroid) * sum of payloads
 * (Jensen & Buhler 2002).  Results change (an approximation; 0 visits every point, in tree order); the exact
 * definition is DESIGN.md section 8.  Takes effect from the next rt_dispatch_rays. */
int  rt_set_gather_mode(rt_ctx* ctx, int hierarchical, float max_solid_angle);

/* Raytracing::dispatch_rays (raytracing.cpp:652-747): [translucent_rgen] -> copy -> camera_rgen, then
 * accumulator_count += 1 and translucent_accumulator_count += collect (raytracing.cpp:744-745).  Asynchronous; up to
 * two dispatches are in flight at once (not with RT_FLAG_DOUBLE_DISPATCH / CAPTURE_HITS / COUNT_TRAVERSAL, the
 * hierarchical gather, or a stream set by rt_set_stream): every other call waits for both. */
int  rt_dispatch_rays(rt_ctx* ctx);

/* No counterpart in the reference.  For callers that cannot allocate device memory -- code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc: the frame of rt_dispatch_rays, one at a time on the context's
 * stream (rt_set_stream's, if set), with every scratch buffer (ray queues, hit records, path state, segment counters,
 * traversal-stack overflow) in a 27.5 MB static device array instead of buffers allocated on demand.  Calls no
 * hipMalloc / hipFree / event or stream creation.  Limits: at most 64 tiles of 32x32 pixels (e.g. 256x256) and 65,536
 * sample points, else RT_ERR_LIMIT; RT_FLAG_CAPTURE_HITS / RT_FLAG_TIME_KERNELS give RT_ERR_UNSUPPORTED (they
 * allocate).  Same results as rt_dispatch_rays, bit for bit -- except in frames with the BSSRDF gather, where rt_dispatch_rays
 * batches the gather launches of a frame in rows it allocates (RT_GATHER_BATCH) and this entry point gathers bounce by bounce:
 * the same terms added in another order, equal to rounding (1e-7 relative; bit for bit with RT_GATHER_BATCH=0).  All contexts
 * on one GPU share the array: calls that overlap on one GPU (two contexts, or two streams) race. */
int  rt_dispatch_rays_static(rt_ctx* ctx);

/* ---- synchronisation and read-back -------------------------------------------------------------------- */
int  rt_sync(rt_ctx* ctx);
int  rt_read_accumulator(rt_ctx* ctx, float* rgba);           /* g_sample_accumulator, W*H*4 floats; blocks */
int  rt_read_render_target(rt_ctx* ctx, uint8_t* rgba8);      /* g_render_target, W*H*4 bytes; blocks       */
int  rt_read_hit_records(rt_ctx* ctx, uint32_t bounce, RtHit* out /* W*H */);
/* translucent instance i (order of creation in rt_build_tlas): points (read buffer) and normals */
int  rt_translucent_instance_count(rt_ctx* ctx, uint32_t* n);
int  rt_sample_point_count(rt_ctx* ctx, uint32_t instance, uint32_t* n);
int  rt_read_sample_points(rt_ctx* ctx, uint32_t instance, RtSamplePoint* points, float* normals /* 3 per point */);
int  rt_get_stats(rt_ctx* ctx, RtStats* out);
int  rt_reset_stats(rt_ctx* ctx);

/* ---- multi-GPU helpers (no counterpart in the reference, which is single-GPU) ------------ */
/* Tile split: this context renders only 32x32 tiles t with t % world == rank; other pixels are left untouched
 * (zero after a reset) so that a sum over ranks reproduces the single-GPU image bit for bit. */
int  rt_set_tile_split(rt_ctx* ctx, uint32_t rank, uint32_t world);
/* Device pointers for collectives run by the caller (RCCL through torch.distributed): */
int  rt_accumulator_device_ptr(rt_ctx* ctx, void** dev_ptr, size_t* bytes);
/* Re-run the resolve step (raytracing.hlsl:161) after the caller has summed accumulators across ranks. */
int  rt_resolve_render_target(rt_ctx* ctx);
/* The same with an explicit frame count: after a sample-set split every rank's accumulator_count is its own share of the
 * frames, while the summed accumulator holds total_frames of them.  The context's globals are left as they are. */
int  rt_resolve_render_target_frames(rt_ctx* ctx, uint32_t total_frames);
/* Tile split: collecting the image by exchanging only the tiles each rank owns (1 / world of the frame per rank) instead
 * of summing whole frames.  rt_tile_pack_bytes: size of one rank's packed tiles -- ceil(tiles / world) x 1,024 pixels x
 * 16 B, the same on every rank, so that an all_gather / gather of equal blocks works.  rt_pack_tiles: copies the
 * accumulator pixels of the tiles of `rank` (tile t with t % world == rank, in increasing t; inside a tile in the order
 * of the renderer's threads; slots outside the image and a missing last tile are zero) into the caller's device buffer.
 * rt_unpack_tiles: the reverse, into this context's accumulator (the 8-bit target is re-derived at the next read).
 * Stream-ordered on the context's stream like everything else; the caller synchronises (rt_sync) before handing the
 * buffer to a collective on another stream.  Plain copies: bit-identical to the sum of the ranks' frames. */
int  rt_tile_pack_bytes(rt_ctx* ctx, uint32_t world, size_t* bytes);
int  rt_pack_tiles(rt_ctx* ctx, uint32_t rank, uint32_t world, void* dev_packed);
int  rt_unpack_tiles(rt_ctx* ctx, uint32_t rank, uint32_t world, const void* dev_packed);
/* Tile split with the sample-point pass partitioned too (by default every rank runs the whole pass -- it is
 * 0.2 % of a C4 frame -- and no rank needs another's payloads).  rt_set_probe_split: the translucent_rgen dispatches of this
 * context trace only block `rank` of `world` equal blocks of the sample points (all instances concatenated in creation order;
 * block = ceil(n / world) points, the last one short); the payloads of the other blocks are left as they are.
 * rt_set_probe_exchange: `fn` is called by rt_dispatch_rays right after it has enqueued a sample-point pass, with the HIP
 * stream the pass runs on (hipStream_t as void*), the device array of all n points (RtSamplePoint[n], the write buffer,
 * raytracing.cpp:607-621) and the block size: it enqueues ON THAT STREAM (or completes before it returns) an all-gather
 * of the ranks' blocks into the array -- block r = points [r * block, min((r + 1) * block, n)) -- and returns 0.  What reads
 * the points afterwards (the copy of raytracing.cpp:723-728, the frame's gather launch) is ordered behind it; in pipelined
 * frames that stream is the library's sample-point stream, so the exchange runs beside the frame's camera pass.  A point's
 * payload is computed by exactly one rank with the one-GPU arithmetic: after the exchange every rank holds the one-GPU
 * payloads bit for bit.  fn = NULL: no exchange (default).  raytracer_amd/distributed.py: install_probe_exchange. */
typedef int (*rt_probe_exchange_fn)(void* user, void* hip_stream, void* dev_points, uint32_t n_points, uint32_t block_points);
int  rt_set_probe_split(rt_ctx* ctx, uint32_t rank, uint32_t world);
int  rt_set_probe_exchange(rt_ctx* ctx, rt_probe_exchange_fn fn, void* user);

#ifdef __cplusplus
}
#endif
#endif /* RT_ABI_H */
