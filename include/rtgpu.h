/*
 * rtgpu.h -- C ABI of librtgpu.so: the MI355X (gfx950) render hot path of
 * AnalogArnold/rayTracer behind plain pointers and sizes.
 *
 * Every entry point names the reference interface it replaces (paths under the
 * reference checkout).  All arrays are caller-owned, C-contiguous, IEEE binary64
 * unless stated.  Functions returning int return 0 on success and a negative
 * RT_E_* code on failure; rt_last_error() then holds a message (thread-local).
 * No exceptions cross the ABI.  A context is bound to one GPU and may be used by
 * one host thread at a time.  There is no CPU fallback: without a GPU, or without
 * the HIP runtime, the device entry points fail with RT_E_NO_DEVICE.
 */
#ifndef RTGPU_H
#define RTGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_OK 0
#define RT_E_INVALID (-1)   /* bad argument */
#define RT_E_NO_DEVICE (-2) /* no usable HIP device / HIP call failed */
#define RT_E_NO_SCENE (-3)  /* render before rt_upload_scene */
#define RT_E_IO (-4)        /* cannot open output file */
#define RT_E_LIMIT (-5)     /* scene exceeds an encoding limit */
#define RT_E_NOMEM (-6)
#define RT_E_BUSY (-7)      /* rt_render_static: another context's launch still uses this GPU's scratch array */

typedef struct rt_ctx rt_ctx;     /* device context: one per GPU */
typedef struct rt_scene rt_scene; /* host-side acceleration structure of ONE timestep, flattened */

const char *rt_last_error(void);
const char *rt_version(void);
int rt_device_count(void);
/* Host threads this process uses for the BVH build pool (and the host jump-state helper): RT_HOST_THREADS if
 * set, else the cores the process may run on divided by LOCAL_WORLD_SIZE (ranks on this node), clamped to
 * 1..16.  OMP_NUM_THREADS is ignored on purpose (torchrun sets it to 1 for multi-rank jobs).  The reference
 * has one thread; no counterpart. */
int rt_host_threads(void);
/* The RT_* tuning / comparison variables are read from the environment once, by the first
 * rt_create (or the first host-only call that needs one); no render, upload or driver call reads the environment.
 * This re-reads them -- for tests and benchmarks that switch a variable between two calls of one process.  Call it
 * only while no other call of this library is in flight.  No counterpart in the reference (it has no knobs). */
int rt_reload_options(void);
/* The names of those variables, comma-separated (static storage). */
const char *rt_option_names(void);
/* Page-locked host memory for rt_download targets (hipHostMalloc): device-to-host copies into pageable memory
 * run at ~10 GB/s, into pinned memory at the link's rate.  NULL on failure.  No counterpart in the reference
 * (its frame never leaves the host). */
void *rt_host_alloc(size_t bytes);
void rt_host_free(void *p);

/* ------------------------------------------------------------------ */
/* Host BVH build + flatten.                                           */
/* Replaces: TLAS build_acceleration_structures(coords, colours,      */
/*   timestep, timestep_count)  cpp/rtbvh.h:235-238, rtbvh.cpp:590-685 */
/* (binned SAH, 8 bins, BLAS leaf <= 4, TLAS leaf <= 2; tree identical */
/* to the reference's, node for node).                                 */
/*   coords[m]  : (T, E_m, 3, 3)   colors[m] : (T, E_m, 3)             */
/*   elem_counts[m] = E_m (the reference reads it from                 */
/*   face_colors.shape(1), rtbvh.cpp:615).  E_m >= 1.                  */
/* n_threads: meshes are built concurrently on up to n_threads host    */
/* threads (0: the process's pool of rt_host_threads() threads, big    */
/* trees split into tasks: least latency of ONE build; < 0: the same   */
/* pool, every tree built by one thread: least CPU time, for callers   */
/* that keep several builds in flight on threads of their own, as      */
/* rt_render_scene does); results do not depend on it.                 */
/* Returns NULL (rt_last_error) on invalid arguments, and when a        */
/* triangle's centroid -- or, with more than two meshes, the centre of  */
/* a mesh's box -- is a NaN or an infinity in a mesh that is split at   */
/* all (more than 4 triangles): the reference's bin index               */
/* static_cast<int>(t * NUM_BINS), rtbvh.cpp:184, is undefined there.   */
/* ------------------------------------------------------------------ */
rt_scene *rt_scene_build(int n_meshes, const double *const *coords, const double *const *colors,
                         const int64_t *elem_counts, int timestep, int n_threads);
void rt_scene_free(rt_scene *scene);

typedef struct {
    int32_t n_meshes;       /* = number of BLAS slots */
    int32_t n_tlas_nodes;
    int64_t n_blas_nodes;   /* sum over slots, reference numbering */
    int64_t n_tris;         /* sum over meshes */
    int64_t n_dev_nodes;    /* records in the device node array (includes alignment pads) */
    int32_t stack_need;     /* traversal stack entries the kernel needs for this tree */
    int32_t max_leaf_tris;  /* largest leaf (unbounded when a split fails, rtbvh.cpp:295-297) */
    int64_t midpoint_fallbacks; /* number of "SAH splitting failed" events, rtbvh.cpp:128 */
    double build_seconds;
} rt_scene_info_t;
int rt_scene_info(const rt_scene *scene, rt_scene_info_t *out);
/* Test seam, host-only.  The builder partitions long ranges (>= 65,536 elements) with a chunk-parallel closed form of
 * the reference's sequential two-pointer loop (cpp/rtbvh.cpp:271-285); this runs both on `cases` random class patterns
 * of up to max_n elements with random chunk lengths and returns how many differ (0 = identical permutations). */
int64_t rt_partition_selftest(int64_t cases, uint64_t seed, int max_n);

/* Test seam, host-only.  The jump-ahead works modulo the characteristic polynomial phi of MT19937's one-word
 * transition (degree 19937).  phi is compiled in (312 words, bit i of word i / 64 = coefficient of x^i);
 * recompute != 0 derives it afresh from the generator's output with Berlekamp-Massey instead (tens of ms): the two
 * must be equal.  No counterpart in the reference (std::mt19937 never jumps). */
int rt_mt_phi(uint64_t *out312, int recompute);

/* Introspection of the host tree in the reference's own numbering (parity tests). */
int rt_scene_tlas(const rt_scene *scene, double *boxes6, int32_t *blas_count, int32_t *left,
                  int32_t *min_blas_idx);
/* slot = position in tlas.blases as the reference stores it (rtbvh.cpp:544-561) */
int rt_scene_slot_info(const rt_scene *scene, int slot, int32_t *mesh, int32_t *n_nodes, int32_t *n_elems,
                       int64_t *tri_base);
/* index of the slot's root in the device node array: node i of the slot (reference numbering) is record
 * root + i, and an internal node's traversal descriptor is the record index of its left child */
int64_t rt_scene_slot_dev_root(const rt_scene *scene, int slot);
int rt_scene_slot_nodes(const rt_scene *scene, int slot, double *boxes6, int32_t *left, int64_t *count,
                        int32_t *min_idx);
int rt_scene_slot_indices(const rt_scene *scene, int slot, int32_t *sorted_to_elem);
/* prim id (as returned by the kernels) -> (mesh, element); arrays of n_tris */
int rt_scene_prim_tables(const rt_scene *scene, int32_t *prim_mesh, int32_t *prim_elem);

/* ------------------------------------------------------------------ */
/* Device context                                                      */
/* ------------------------------------------------------------------ */
rt_ctx *rt_create(int device);
void rt_destroy(rt_ctx *ctx);
/* Upload the flattened tree (once per frame/timestep; replaces the TLAS by-value
 * local of cpp/rtmain.cpp:50 as the thing render_ppm_image reads). */
int rt_upload_scene(rt_ctx *ctx, const rt_scene *scene);

/* ------------------------------------------------------------------ */
/* Sampler: the reference's process-global `static std::mt19937        */
/* generator` (seed 5489, never reseeded) + uniform_real_distribution  */
/* cpp/rtmathutils.h:15-19.  The context keeps the stream position in  */
/* 32-bit outputs; rt_render() consumes H*W*spp*4 words per full image */
/* exactly like render_ppm_image does.  Jitter pairs are generated on  */
/* the GPU from jump-ahead states (GF(2) polynomial jump, also on the GPU). */
/* ------------------------------------------------------------------ */
int rt_rng_seed(rt_ctx *ctx, uint32_t seed);          /* position 0 of mt19937(seed) */
int rt_rng_seek(rt_ctx *ctx, uint64_t word_position); /* absolute position in the seeded stream */
uint64_t rt_rng_tell(const rt_ctx *ctx);
/* Test seam: the next n values random_double()-0.5 would return, generated on the
 * GPU by the same kernel rt_render uses, in `n_streams` jump-ahead streams.  Advances
 * the position by 2*n words. */
int rt_rng_jitter(rt_ctx *ctx, int64_t n, int n_streams, double *out_host);
/* Test seam: the context's device-resident generator state (the 624 raw words preceding the current
 * position), as maintained by the GPU jump kernels. */
int rt_rng_state(rt_ctx *ctx, uint32_t *state624);
/* Host-only helpers (no GPU needed): the raw 624-word generator state that precedes
 * `word_position` of mt19937(seed), and n such states `stride` words apart.  The state is
 * the window x[p-624..p-1] of untempered words; the bits of the oldest word below bit 31
 * are not part of the generator's state and are unspecified. */
int rt_mt19937_state_at(uint32_t seed, uint64_t word_position, uint32_t *state624);
int rt_mt19937_stream_states(uint32_t seed, uint64_t base_position, uint64_t stride, int n,
                             uint32_t *states);

/* ------------------------------------------------------------------ */
/* Render one image (or a band of its rows).                           */
/* Replaces the body of render_ppm_image(...)  cpp/rtrender.h:22-29,   */
/* rtrender.cpp:35-67 (everything but the file write) and its callees  */
/* return_ray_color / intersect_TLAS / intersect_BLAS / intersect_AABB */
/* / intersect_bvh_triangles  (rtrender.cpp:13-33,                     */
/* rtrayintersection.cpp:36-228).                                      */
/* ------------------------------------------------------------------ */
#define RT_SAMPLER_MT19937 0  /* reference stream from the context's position */
#define RT_SAMPLER_EXPLICIT 1 /* caller supplies the (ox, oy) offsets */
#define RT_SAMPLER_PHILOX 2   /* counter-based FAST sampler, NOT the reference's stream (reported separately):
                                 sample s of the stream (position/4) = Philox4x32-10({s,0,0},{seed,0}); no state */

#define RT_FLAG_HIT_IDS 1u  /* keep per-sample prim ids (int32, -1 = miss) */
#define RT_FLAG_COUNTERS 2u /* count nodes/triangles visited (slower; for the roofline model) */
#define RT_FLAG_HIT_T 4u    /* keep per-sample closest t (f64, +inf = miss) */
#define RT_FLAG_FP32 8u     /* fp32 FAST PATH: the same frame in binary32 (north_star: "an fp32 fast path reported
                               separately with its tolerance").  Not the parity path: hit ids may differ for rays
                               grazing an edge, intensities are close.  No reference counterpart. */

typedef struct {
    double camera_center[3];
    double pixel_00_center[3];
    double pixel_spacing[6]; /* 2x3 row-major */
    int32_t image_height, image_width, samples;
    int32_t row_begin, row_end; /* rows [row_begin,row_end) are rendered; 0,H for the whole image */
    int32_t sampler;
    uint32_t flags;
    /* RT_SAMPLER_EXPLICIT: host array of (row_end-row_begin)*W*spp pairs, sample order (j,i,k) */
    const double *jitter;
} rt_render_params;

/* Results stay in context-owned device buffers until rt_download / the next rt_render.
 * With RT_SAMPLER_MT19937 the stream position advances by (row_end-row_begin)*W*spp*4;
 * rows before row_begin are NOT skipped automatically (callers that shard an image by
 * row bands seek to frame_base + row_begin*W*spp*4 themselves). */
int rt_render(rt_ctx *ctx, const rt_render_params *params);

/* ------------------------------------------------------------------ */
/* rt_render for callers that CANNOT ALLOCATE.  No counterpart in the   */
/* reference (which allocates per leaf visit); it exists for callers    */
/* that must not reach hipMalloc: tight serving loops, harnesses that   */
/* forbid allocation, processes close to the device-memory limit.       */
/* (NOT capturable in a HIP graph: the call waits for this context's    */
/* previous launch, reads event times and launches on the context's     */
/* own stream.)                                                         */
/* Same kernels and results as rt_render (packet path), but nothing is  */
/* allocated or freed: the camera-relative node / TLAS / triangle       */
/* records the prep kernels write, the pixel blocks' cuts when they fit */
/* behind them, and the visit counters live in a file-scope device      */
/* array of RT_STATIC_SCRATCH_BYTES that every GPU the library runs on  */
/* holds until the process ends; jitter and outputs are the CALLER's    */
/* device memory:                                                       */
/*   d_jitter   : (row_end-row_begin)*W*spp (ox,oy) pairs, order (j,i,k) */
/*   d_intensity: rows*W f64     d_rgb: rows*W*3 u8                     */
/*   d_hit_prim : rows*W*spp i32, required with RT_FLAG_HIT_IDS else NULL */
/* params->sampler must be RT_SAMPLER_EXPLICIT (params->jitter is       */
/* ignored); flags: RT_FLAG_HIT_IDS, RT_FLAG_COUNTERS.  The scene is    */
/* the one last given to rt_upload_scene.  Limits (RT_E_LIMIT, with the */
/* sizes in rt_last_error): 256 + 64*n_tlas + 64*n_dev_nodes +          */
/* 128*n_tris bytes (each part rounded up to 256) must fit the array    */
/* (about 400,000 triangles), and the tree must fit the packet kernel   */
/* (stack_need <= 128).  The call returns when its kernels are          */
/* enqueued on the context's stream (after waiting for the context's    */
/* PREVIOUS launch): rt_sync() before reading the outputs;              */
/* rt_last_stats reports its kernel time, counters and whether cuts     */
/* were used (finest_cut_level).  The MT19937 stream position and the   */
/* outputs held for rt_download are untouched.  All contexts on one GPU */
/* share that GPU's array: a call made while ANOTHER context's launch   */
/* into it has not finished returns RT_E_BUSY and launches nothing      */
/* (calls are serialised by a per-process lock; processes sharing a GPU */
/* have an array each).                                                 */
/* ------------------------------------------------------------------ */
#define RT_STATIC_SCRATCH_BYTES (64u << 20)
int rt_render_static(rt_ctx *ctx, const rt_render_params *params, const double *d_jitter, double *d_intensity,
                     uint8_t *d_rgb, int32_t *d_hit_prim);

/* Copy results of the last rt_render to host; any pointer may be NULL.
 *   intensity: rows*W  f64  gray/spp before quantisation (rtrender.cpp:61-62)
 *   rgb      : rows*W*3 u8  the P6 payload (rtrender.cpp:63-65)
 *   hit_prim : rows*W*spp i32, needs RT_FLAG_HIT_IDS
 *   hit_t    : rows*W*spp f64, needs RT_FLAG_HIT_T */
int rt_download(rt_ctx *ctx, double *intensity, uint8_t *rgb, int32_t *hit_prim, double *hit_t);

typedef struct {
    double render_kernel_ms; /* HIP-event time of the traversal kernel, on its own stream */
    double jitter_kernel_ms; /* HIP-event time of the MT19937 jitter kernel */
    double host_jump_ms;     /* host time spent on jump-ahead states */
    uint64_t rays;
    uint64_t nodes_visited; /* RT_FLAG_COUNTERS: TLAS+BLAS nodes slab-tested */
    uint64_t tris_tested;   /* RT_FLAG_COUNTERS */
    uint64_t rays_hit;      /* RT_FLAG_COUNTERS */
    uint64_t wave_node_fetches; /* RT_FLAG_COUNTERS: node records fetched per wave, summed (packet kernel) */
    uint64_t wave_tri_fetches;  /* RT_FLAG_COUNTERS: triangle records fetched per wave, summed (packet kernel) */
    /* running totals since rt_create over the launches whose events have been read (every rt_render reads its
     * predecessor's; rt_last_stats reads the last one's): a loop of K rt_render calls followed by ONE
     * rt_last_stats gives the K launches' kernel time without a sync per step */
    double render_kernel_ms_total;
    uint64_t render_launches_timed;
    uint64_t jitter_streams; /* MT19937 sampler: streams the launch's offsets were generated on (0: other samplers) */
    int64_t finest_cut_level; /* packet kernels: level (log2 of the block side in pixels) of the finest cuts the launch's
                                 walk started from, -1 when it walked from the TLAS root (test seam, see rt_last_cuts) */
} rt_stats;
int rt_last_stats(const rt_ctx *ctx, rt_stats *out);
int rt_sync(rt_ctx *ctx);

/* Test seams for the cuts (no counterpart in the reference, whose walk always starts at node 0,
 * cpp/rtrayintersection.cpp:203).  Per block of (1 << level)^2 pixels 32 words: a header -- bit 31 valid, bit 16
 * CUT_SIMPLE (the list holds only leaf descriptors and trimmed-leaf pairs), bits 13..8 the three axis kinds, bits 7..0
 * the entry count -- and up to 31 entries in visit order at which the walk of that block's rays starts instead of at
 * the TLAS root: a plain descriptor (every ray of the block passes the node), CUT_NODE | index (BLAS node tested per
 * ray; with CUT_TRIM, finest level only, followed by the descriptor of the leaf's live triangle range) or
 * DESC_TLAS | index (TLAS node tested per ray).  Blocks are numbered row-major, *tiles_x per row; levels
 * *level_lo .. *level_top exist, the top one being max(render tile, 8 x 8 pixels) -- many blocks, each refined from
 * the TLAS root -- not a single block for the image.
 * rt_last_cuts: those of the last rt_render, from the device.  Returns the number of blocks of `level` (0 when
 * the last render made no cuts), or a negative error; words = NULL only queries.
 * rt_tile_cuts_host: host-only twin (the same code compiled for the host) for this scene, camera and image (only
 * camera, sizes, samples and rows of `params` are read): lets the CPU suite check the cuts against a brute-force
 * walk and the GPU suite compare rt_last_cuts with it word for word. */
int rt_last_cuts(rt_ctx *ctx, int level, uint32_t *words, int64_t capacity_words, int32_t *tiles_x, int32_t *level_lo,
                 int32_t *level_top);
int rt_tile_cuts_host(const rt_scene *scene, const rt_render_params *params, int level, uint32_t *words,
                      int64_t capacity_words, int32_t *tiles_x, int32_t *level_lo, int32_t *level_top);

/* Raw device pointers of the last render's outputs (for callers that keep the
 * framebuffer on the GPU, e.g. a peer gather).  NULL if not produced. */
int rt_device_outputs(rt_ctx *ctx, void **intensity, void **rgb, void **hit_prim);

/* ------------------------------------------------------------------ */
/* Arbitrary rays against the uploaded scene.                          */
/* Test seam for return_ray_color(ray, TLAS) cpp/rtrender.h:19-20 and  */
/* intersect_TLAS  cpp/rtrayintersection.h:34-37.  dirs are used as    */
/* given (not normalised).  Outputs may be NULL.                       */
/* ------------------------------------------------------------------ */
int rt_trace_rays(rt_ctx *ctx, int64_t n, const double *origins, const double *dirs, double *t,
                  double *bary_uvw, double *color, int32_t *prim);

/* ------------------------------------------------------------------ */
/* Whole-job driver.                                                   */
/* Replaces render_scene(...) cpp/rtmain.cpp:29-73 == the nanobind     */
/* export cpp_render_scene (rtmain.cpp:157-160): for each timestep     */
/* build+upload, for each camera render and write                      */
/* <dir>/rtimage_<t>_cam<c>.ppm, printing the reference's stdout lines.*/
/* The sampler stream is process-global and never reseeded, like the   */
/* reference's static generator.                                       */
/* rank/world shard the frames f = t*n_cameras + c (see rt_shard_frames) */
/* (no collective; every rank seeks the stream to its frame).          */
/* ------------------------------------------------------------------ */
int rt_render_scene(int image_height, int image_width, int number_of_samples, const char *output_directory,
                    int timestep_count, int n_meshes, const double *const *coords,
                    const double *const *colors, const int64_t *elem_counts, int n_cameras,
                    const double *camera_centers, const double *pixel_00_centers,
                    const double *pixel_spacings, int device, int rank, int world);

/* Position (in 32-bit words) of the never-reseeded stream rt_render_scene draws from on `device`: the
 * reference's process-global `static std::mt19937 generator` (cpp/rtmathutils.h:16) persists across
 * render_scene calls, and so does this.  0 before the first call.  Test seam; the reference's generator
 * is not observable. */
uint64_t rt_driver_stream_position(int device);

/* Host-only: the frame schedule rt_render_scene uses.  For rank `rank` of `world` it lists the frames
 * f = t*n_cameras + c it renders -- whole timesteps, t % world == rank, when timestep_count >= world (one
 * host BVH build per timestep, on one rank), else single frames, f % world == rank -- and the position of
 * each in the global RNG stream (stream_base + f*frame_words).  Returns the number of frames written (<= capacity), or a
 * negative error.  frames/positions may be NULL to query the count only. */
int64_t rt_shard_frames(int timestep_count, int n_cameras, int rank, int world, uint64_t stream_base,
                        uint64_t frame_words, int64_t capacity, int64_t *frames, uint64_t *positions);

/* Host-only: the row-band schedule for ONE big frame rendered by several GPUs (the tree is
 * replicated, the image split into contiguous row bands, no collective but the gather of the bands).
 * Rank `rank` of `world` renders rows [*row_begin, *row_end) -- possibly empty -- with rt_render's
 * row_begin/row_end after rt_rng_seek(*stream_position): the band starts at
 * frame_base + row_begin*W*spp*4 of the reference's stream (cpp/rtrender.cpp:47-60 draws row-major),
 * so the assembled image is the one a single rank renders.  Band boundaries fall on multiples of 8 rows
 * when image_height >= 8*world.  No counterpart in the reference (it has one thread and one frame loop). */
int rt_shard_rows(int image_height, int image_width, int number_of_samples, int rank, int world, uint64_t frame_base,
                  int *row_begin, int *row_end, uint64_t *stream_position);

/* P6 writer: "P6\nW H\n255\n" + payload (rtrender.cpp:69-81). Prints the
 * reference's messages ("Failed to open the output file." / " Done."). */
int rt_write_ppm(const char *path, int width, int height, const uint8_t *rgb);

#ifdef __cplusplus
}
#endif
#endif /* RTGPU_H */
