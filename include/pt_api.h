/*
 * pt_api.h - C ABI of the MI355X path-tracing backend (libpt_hip.so).
 *
 * The reference renderer (JacobScheff/Path-Tracing) has no FFI: its renderer
 * boundary is the wgpu bind group of seven storage buffers, the shader's
 * compile-time constants and one draw per frame.  Each entry point below
 * replaces one piece of that contract; the reference location it stands in
 * for is cited at the declaration.  Plain C types only: the header is meant to
 * be consumed by bindgen / cgo / ctypes as it stands.
 *
 * Conventions
 *   - every function returns PT_OK (0) or a negative PT_E* code and never
 *     aborts; pt_last_error() returns the message of the last failure on that
 *     context (or of the last failed pt_create when ctx is NULL);
 *   - a pt_ctx is bound to one GPU and one HIP stream and is single-threaded;
 *     distinct contexts may be driven from distinct host threads;
 *   - scene arrays are COPIED at upload (like wgpu create_buffer_init), the
 *     caller keeps ownership;
 *   - all floats are IEEE binary32, files are little-endian.
 */
#ifndef PT_API_H
#define PT_API_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PT_OK 0
#define PT_EINVAL (-1)   /* bad argument / malformed scene file */
#define PT_EHIP (-2)     /* a HIP runtime call failed (message has the HIP error string) */
#define PT_ENOMEM (-3)
#define PT_EIO (-4)      /* file could not be read */
#define PT_ESTATE (-5)   /* call made in the wrong order (e.g. render before upload) */

/* pt_params.flags */
#define PT_FLAG_EXACT_MATH 1u  /* log/cos/pow from portable IEEE-only polynomials: radiance is then
                                  reproducible bit for bit on a CPU; default is the hardware
                                  transcendental instructions */
#define PT_FLAG_COUNT_WORK 2u  /* kernels also count box tests / triangle tests / ray casts
                                  (pt_get_counters); slower, for roofline accounting */

#define PT_FLAG_WIDE_BVH 4u    /* traverse a four-wide tree folded from the node file at upload (children of inner
                                  children pulled up): half as many dependent node visits.  Same boxes, same ray_box and
                                  ray_triangle arithmetic, so every reported hit is a hit the shader's functions confirm
                                  with the same distance, but the visit order is not the shader's: the closest-hit id can
                                  differ on exact distance ties, and a triangle that the shader's walk loses because a
                                  folded-away box rejects the ray can be found.  Not combinable with PT_FLAG_COUNT_WORK.
                                  No counterpart in the reference (binary walk, shader.wgsl:169-235). */

typedef struct pt_ctx pt_ctx;

/*
 * Everything the reference fixes at pipeline-build time as WGSL `const`s:
 * src/shaders/shader.wgsl:24-37 (counts, bounces, rays per pixel, screen size,
 * fov), :48-54 (environment) and the triangle material hard-coded in
 * ray_triangle, :283-286.  Counts of spheres/triangles/nodes are taken from
 * pt_upload_scene instead.
 */
typedef struct pt_params {
    uint32_t width, height;          /* screen_size, shader.wgsl:35 */
    uint32_t rays_per_pixel;         /* shader.wgsl:34 */
    uint32_t max_bounce_count;       /* shader.wgsl:33; the bounce loop is inclusive (N+1 casts) */
    float fov_degrees;               /* shader.wgsl:36 (60), converted with pi = 3.14159 */
    float tri_color[3];              /* shader.wgsl:283 */
    float tri_emission_color[3];     /* shader.wgsl:284 */
    float tri_emission_strength;     /* shader.wgsl:285 */
    float tri_smoothness;            /* shader.wgsl:286 */
    float sky_horizon[3];            /* shader.wgsl:48 */
    float sky_zenith[3];             /* shader.wgsl:49 */
    float ground[3];                 /* shader.wgsl:50 */
    float sun_dir[3];                /* shader.wgsl:51 (not normalised) */
    float sun_intensity;             /* shader.wgsl:52 */
    float sun_focus;                 /* shader.wgsl:53 */
    uint32_t use_environment_lighting; /* shader.wgsl:54 */
    float far_distance;              /* shader.wgsl:142,177 (1000000.0): distance of "no hit yet"; nothing at or beyond it is hit */
    uint32_t bvh_max_depth;          /* shader.wgsl:30: the shader's stack holds bvh_max_depth + 1 entries (:170).  0 = take the
                                        depth of the uploaded node file; otherwise a node file deeper than this is refused at
                                        upload (the shader would overrun its stack) */
    uint32_t flags;                  /* PT_FLAG_* */
} pt_params;

/* Work counters (PT_FLAG_COUNT_WORK).  They follow the reference's traversal
 * (shader.wgsl:169-235): box_tests = ray_box evaluations, tri_tests =
 * ray_triangle evaluations, ray_casts = calculate_ray_collision calls the
 * reference would make (one per bounce of every sample). */
typedef struct pt_counters {
    uint64_t ray_casts;
    uint64_t box_tests;
    uint64_t tri_tests;
    uint64_t paths;
    /* the part of the three counts above that the bounce/path kernel itself
     * executed (casts after the primary one); used for that kernel's roofline */
    uint64_t path_kernel_ray_casts;
    uint64_t path_kernel_box_tests;
    uint64_t path_kernel_tri_tests;
} pt_counters;

/* HIP-event timing of the kernels launched by the render calls since the last
 * pt_reset_timing (events are recorded around every kernel on the stream it runs on).  Consecutive
 * batches of frames run on two streams and overlap, so the sums below can exceed the wall time; the
 * primary-ray kernel runs only when camera, scene or pixel shard have changed since the batch slot's
 * last use. */
typedef struct pt_timing {
    uint64_t frames;           /* frames rendered */
    uint64_t batches;          /* launches of each kernel (frames are rendered in batches) */
    double primary_ms;         /* primary-ray kernel, summed */
    double path_ms;            /* bounce/path kernel (the dominant one), summed */
    double resolve_ms;         /* per-pixel resolve + accumulation kernel, summed */
} pt_timing;

/* Fills *p with the values the committed reference source uses
 * (shader.wgsl:24-54, 283-286): 1200x600, 20 rays/pixel, 10 bounces, fov 60. */
int pt_default_params(pt_params *p);

/* Replaces State::new's device/pipeline creation (src/main.rs:144-190, 270-274):
 * selects GPU `device_ordinal`, creates the stream and the zeroed accumulation
 * buffer (main.rs:365-379; 16 bytes per pixel as the shader addresses it,
 * shader.wgsl:41). */
int pt_create(int device_ordinal, const pt_params *params, pt_ctx **out);

/* Replaces the sphere / triangle / BVH storage buffers, bindings 0, 5 and 6
 * (src/main.rs:286-355, shader.wgsl:39,44,45).  spheres: sphere_count x 12 f32
 * (x y z radius r g b er eg eb emission_strength smoothness, main.rs:286);
 * triangles: triangle_count x 9 f32 (.bin contents); bvh: node_count x 9 f32
 * (_bvh.bin contents: min.xyz max.xyz triangle_index triangle_count
 * child_index, prepare_data/src/main.rs:181-192).  sphere_count may be 0.
 * The node file is followed exactly as the shader follows it (including
 * unreachable nodes and inner nodes whose ranges disagree with their
 * children); links that leave the arrays are rejected with PT_EINVAL. */
int pt_upload_scene(pt_ctx *ctx, const float *spheres, uint32_t sphere_count,
                    const float *triangles, uint32_t triangle_count,
                    const float *bvh, uint32_t node_count);

/* Convenience for the include_bytes! pair of src/main.rs:323,342. */
int pt_load_scene_files(pt_ctx *ctx, const char *triangles_bin, const char *bvh_bin,
                        const float *spheres, uint32_t sphere_count);

/* Refit: the triangles of the uploaded scene moved (same count, same order), the tree keeps its shape.
 * No counterpart in the reference, whose tree is built offline (prepare_data/src/main.rs); the result is
 * what pt_upload_scene would hold for `triangles` and the uploaded node file with every node's box replaced
 * by the tight bounds of the triangles in its range (the rule the builder's boxes follow, main.rs:16-20,
 * :108-117), bit for bit, computed on the GPU in the resident records: triangles: triangle_count x 9 f32 as in
 * pt_upload_scene.  Spheres, camera and accumulation buffer are untouched (reset the accumulation as after a
 * camera move).  PT_EINVAL for a node file in which a reachable inner node does not cover exactly the ranges
 * of its two children, A then B (files of the builders here and of the reference's current builder do; its
 * legacy dragon_8k file does not), and for non-finite coordinates; PT_ESTATE with PT_FLAG_WIDE_BVH.
 * Synchronous. */
int pt_refit_scene(pt_ctx *ctx, const float *triangles);

/* Rebuild: new triangles (any count >= 1, any order), a new tree, everything on the GPU: the tree of
 * pt_bvh_build_gpu(triangles, max_depth) and the records pt_upload_scene would make from its two files, without
 * the files ever existing on the host.  No counterpart in the reference (offline builder, prepare_data/src/main.rs).
 * Afterwards the context holds exactly what pt_upload_scene(ctx, <the spheres it had>, triangles_out, ..., nodes_out,
 * ...) of that build would have left, so triangle ids in pt_primary_hits are positions in the builder's triangle
 * order.  Spheres, camera and accumulation buffer are untouched; a context without a scene gets one without spheres.
 * *build_ms (may be NULL): GPU time from the end of the upload of the coordinates.  Same refusals as
 * pt_bvh_build_gpu; PT_ESTATE with PT_FLAG_WIDE_BVH.  Synchronous. */
int pt_rebuild_scene(pt_ctx *ctx, const float *triangles, uint32_t triangle_count, int max_depth, float *build_ms);

/* Replaces the camera position / rotation buffers, bindings 3 and 4, written
 * every frame by State::render (src/main.rs:455-464).  Rotation in degrees,
 * applied as in rotate_vector (shader.wgsl:327-348).  Takes effect for the frames enqueued afterwards;
 * frames already enqueued keep the pose they were enqueued with.  A pose equal to the current one bit for
 * bit changes nothing (the primary rays' results are kept across calls while the pose stands). */
int pt_set_camera(pt_ctx *ctx, const float position[3], const float rotation_degrees[3]);

/* Replaces the frame_count upload + the two draws + `frame_count += 1` of
 * State::render (src/main.rs:448-452, 498-510), n_frames times: frame f uses
 * frame_count = first_frame_count + f and folds its per-pixel mean into the
 * accumulation buffer with weight 1/(frame_count+1) (shader.wgsl:104-106), so
 * frame_count 0 overwrites the buffer.  Asynchronous: the call enqueues and returns (pt_synchronize,
 * pt_wait_in_flight, pt_read_accum wait).  Batches of frames of consecutive calls alternate between two
 * streams of the context, so that one starts while the one before it drains; the frames are folded into
 * the accumulation buffer in call and frame order all the same. */
int pt_render_frames(pt_ctx *ctx, uint32_t first_frame_count, uint32_t n_frames);

/* pt_render_frames for callers that cannot allocate device memory (code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  Same
 * result as pt_render_frames, bit for bit, but the per-batch working state (hit list, control words,
 * per-pixel tables, sample buffer) lives in a static device array of pt_static_scratch_bytes()
 * (64 MiB) that every GPU the library runs on holds until the process ends, and the call itself
 * calls no hipMalloc / hipFree / hipEventCreate: it only enqueues memsets, kernels and waits on events created by pt_create, on the
 * context's first stream.  Scene, camera and accumulation buffer are the context's, as in
 * pt_render_frames.  Needs 8448 + 36 * width * height + 16 * roundup64(width * height) *
 * rays_per_pixel bytes for a one-frame batch (more frames per batch are used when they fit);
 * PT_EINVAL with the two byte counts in the message when that exceeds the array.  The kernels are not
 * timed (pt_get_timing does not see them).  The layout is sized for the worst case (every pixel a hit
 * pixel), so a large image with few hits is refused although it would fit.  All contexts of one GPU
 * share that GPU's array, so their static renders must not overlap in time: run them one after the
 * other (pt_synchronize between contexts).  A call that finds another context's static render still
 * running on its GPU returns PT_ESTATE and renders nothing (the check is an event query, created at
 * pt_create); calls of one context are ordered by its stream. */
int pt_render_frames_static(pt_ctx *ctx, uint32_t first_frame_count, uint32_t n_frames);
size_t pt_static_scratch_bytes(void);

/* Frame-sharded form for multi-GPU renders: ADDS the per-pixel mean of frames
 * first_frame_count, +frame_stride, +2*frame_stride ... (n_frames of them) to
 * the accumulation buffer instead of keeping a running mean.  Summing the
 * buffers of all ranks and multiplying by 1/total_frames gives the running
 * mean of shader.wgsl:104-106 up to f32 rounding.  No reference counterpart
 * (the reference is single-GPU). */
int pt_render_frames_sum(pt_ctx *ctx, uint32_t first_frame_count, uint32_t n_frames,
                         uint32_t frame_stride);

/* accum *= s (used by the root rank after the cross-GPU sum). */
int pt_scale_accum(pt_ctx *ctx, float s);

/* Reads binding 2, the accumulation buffer (shader.wgsl:41,105-106): row-major,
 * row 0 on top, stride_floats = 3 (packed RGB) or 4 (the shader's 16-byte
 * pixel stride).  Synchronises the stream first. */
int pt_read_accum(pt_ctx *ctx, float *out, uint32_t stride_floats);

/* Writes the accumulation buffer from host memory (same layouts as pt_read_accum).  The buffer and
 * the frame count ARE the resumable state of a progressive render (src/main.rs:365-379, 448-452):
 * save the image and the number of frames in it, later write it back and continue with
 * pt_render_frames(ctx, frames_done, n).  No reference counterpart (the reference keeps the
 * state on the GPU only).  Synchronises the stream first. */
int pt_write_accum(pt_ctx *ctx, const float *in, uint32_t stride_floats);

/* Zero-fills the accumulation buffer (its initial state, main.rs:365-379). */
int pt_reset_accum(pt_ctx *ctx);

/* Lets the caller own the accumulation buffer: device_ptr must point to
 * width*height*4 floats of device memory on this context's GPU (for example a
 * torch tensor that a collective will reduce in place).  NULL switches back to
 * the internal buffer.  pt_accum_device_ptr returns the buffer in use. */
int pt_bind_accum(pt_ctx *ctx, void *device_ptr);
void *pt_accum_device_ptr(pt_ctx *ctx);

/* Stream-ordered hand-over of the accumulation buffer between the context and a HIP stream of the
 * caller's (hip_stream is a hipStream_t; NULL is the default stream), without a wait on the host.  No
 * reference counterpart: wgpu orders the accesses to a buffer within its one queue itself
 * (src/main.rs:498-508); here a caller that copies or reduces the image on a stream of its own (a
 * collective on the torch stream, a copy into a display surface) says where its work fits in.
 *   pt_accum_release: work enqueued on hip_stream from now on runs after everything the context has
 *                     enqueued so far that reads or writes the accumulation buffer.
 *   pt_accum_acquire: whatever the context enqueues from now on that reads or writes the accumulation
 *                     buffer (the frame folds of later render calls, pt_scale_accum, pt_reset_accum)
 *                     runs after the work enqueued on hip_stream so far; path kernels of later render
 *                     calls, which do not touch the buffer, start without waiting.  The calls that
 *                     synchronise (pt_synchronize, pt_read_accum, ...) wait for that work too.
 * A loop  render; release(s); <copy the image on s>; acquire(s)  keeps the next render's kernels
 * running while the copy of this one is still to come.  Both calls also hand what the two streams hold to
 * the GPU (a stream wait or an event record alone may stay in the runtime's batch until the next launch or
 * query on its stream, and a host that then blocks outside HIP - in a collective's completion, say - would
 * wait for work the GPU was never given). */
int pt_accum_release(pt_ctx *ctx, void *hip_stream);
int pt_accum_acquire(pt_ctx *ctx, void *hip_stream);

/* A second output for the NEXT render call (pt_render_frames, pt_render_frames_sum, pt_render_frames_static), and for
 * that call only: the kernel that folds the call's last frames into the accumulation buffer writes every pixel's value
 * to device_ptr as well (width*height*4 floats of device memory on this context's GPU, 16-byte aligned, not the
 * accumulation buffer itself), so a caller that hands each image on (to a collective, to a display) while the next
 * render is already running needs no copy kernel and no clearing between renders.  Pixels of other ranks' shards
 * (pt_set_pixel_shard) are written as zeros.  move != 0: the accumulation buffer is left holding ZEROS where the
 * call folded frames, the image exists in device_ptr only; a pt_render_frames_sum call then starts the next sum
 * from nothing without a pt_reset_accum in between.  A call of no frames writes nothing.  NULL withdraws the
 * request.  The write is ordered like the frame folds (pt_accum_release / pt_accum_acquire apply to it).  No reference
 * counterpart (the reference presents its one buffer, src/main.rs:466-508). */
int pt_next_render_output(pt_ctx *ctx, void *device_ptr, uint32_t move);

/* Debug / parity output with no reference counterpart (the shader's HitInfo,
 * shader.wgsl:7-16, has no primitive id): closest hit of every pixel's primary
 * ray.  prim_id = triangle index >= 0, -(1+sphere index) for spheres, INT32_MIN
 * for a miss; distance as HitInfo.distance (1e6 on a miss).  Host arrays of
 * width*height elements. */
int pt_primary_hits(pt_ctx *ctx, int32_t *prim_id, float *distance);

/* Blocks until everything the context has enqueued is done. */
int pt_synchronize(pt_ctx *ctx);

/* Frame pacing.  The render calls only enqueue work (the reference's State::render returns after
 * queue.submit, src/main.rs:506-508, and the surface lets a frame or two be in flight before
 * get_current_texture blocks, :466): a host that waits for every frame before it builds the next one
 * leaves the GPU idle while each path kernel drains.  pt_wait_in_flight(ctx, 1) returns when at most the
 * batch of frames enqueued last is still running, so the tick loop  set_camera; render_frames(f, 1);
 * wait_in_flight(1)  always has the next frame queued behind the current one; 0 is pt_synchronize; larger
 * values return at once (a context runs at most two batches at a time). */
int pt_wait_in_flight(pt_ctx *ctx, uint32_t max_in_flight);

int pt_get_counters(pt_ctx *ctx, pt_counters *out);  /* synchronises; needs PT_FLAG_COUNT_WORK */
int pt_reset_counters(pt_ctx *ctx);
/* Diagnostics of the path kernel (PT_FLAG_COUNT_WORK), summed over waves: [0] wave lifetime in
 * shader clocks, [1] clocks inside the traversal loop, [2] traversal steps, [3] lanes at inner
 * nodes summed over steps, [4] lanes whose leaf code ran, [5] steps in which leaf code ran,
 * [6] outer iterations, [7] lanes shaded summed over outer iterations, [8] longest wave lifetime,
 * [9] longest time a wave went on after it found the queue empty, [10] that time summed over
 * waves, [11] ticks of the 100 MHz device clock from the earliest wave start to the first "queue
 * empty", [12] to the last wave end (one launch), [13..15] reserved.  No reference counterpart. */
int pt_get_debug_stats(pt_ctx *ctx, uint64_t out[16]);
int pt_get_timing(pt_ctx *ctx, pt_timing *out);      /* synchronises */
int pt_reset_timing(pt_ctx *ctx);

/* Scene facts derived at upload: depth of the node file walked from the root
 * (the shader needs bvh_max_depth >= this, shader.wgsl:30,170), reachable inner
 * nodes and leaves. */
int pt_scene_info(pt_ctx *ctx, uint32_t *bvh_depth, uint32_t *inner_nodes, uint32_t *leaves);

/* Parity probe: evaluates the small shader functions on the device, one thread per input, so that
 * they can be compared one by one with a CPU restatement.  out receives 16 floats per input:
 * [0] random_value(seed) (shader.wgsl:351-358), [1] random_normal (:361-366), [2..4]
 * random_direction (:369-375), [5..7] random_hemisphere_direction(normal, seed) (:378-382, unused by
 * the shader itself), [8..10] get_environment_light(dir) (:391-404), [11..13] rotate_vector(dir)
 * with the camera rotation of pt_set_camera (:327-348), [14..15] zero.  dirs holds n x 3 floats.
 * Transcendentals follow PT_FLAG_EXACT_MATH of the context.  Needs no scene. */
int pt_probe_functions(pt_ctx *ctx, const uint32_t *seeds, const float *dirs, const float normal[3],
                       uint32_t n, float *out);

/* Parity probe for the intersection functions, one thread per input: rays n x 6 (origin, dir), boxes
 * n x 6 (min, max), triangles n x 9, spheres n x 4 (centre, radius).  out receives 8 floats per
 * input: [0] ray_box (shader.wgsl:237-256) with the plain divider, [1] the same through the
 * reciprocal form the traversal uses when its predicates hold (NaN when they do not), [2] did_hit
 * and [3] distance of ray_triangle (:258-290), [4] did_hit and [5] distance of ray_sphere
 * (:292-325), distances meaningful where did_hit is set, [6..7] zero.  Needs no scene. */
int pt_probe_intersections(pt_ctx *ctx, const float *rays, const float *boxes, const float *triangles,
                           const float *spheres, uint32_t n, float *out);

/* Pixel sharding for multi-GPU renders: after this call the context renders only the 8x8-pixel
 * tiles with (tile_x + tile_y) % world_size == rank and leaves every other pixel of the
 * accumulation buffer untouched.  Each pixel keeps the reference's running mean (shader.wgsl:
 * 104-106) on the rank that owns it, so adding the zero-initialised buffers of all ranks gives
 * the single-GPU image bit for bit.  (0, 1) restores the whole image.  pt_primary_hits always
 * covers the whole image.  No reference counterpart (the reference is single-GPU). */
int pt_set_pixel_shard(pt_ctx *ctx, uint32_t rank, uint32_t world_size);

/* Replaces Rust drop of State. */
void pt_destroy(pt_ctx *ctx);

/* Replaces the unwrap()/expect() panics of the reference (src/main.rs:152-170). */
const char *pt_last_error(pt_ctx *ctx);

/*
 * Offline BVH builder, replaces prepare_data/src/main.rs:14-136 + 179-220:
 * top-down, 10 candidate planes x 3 axes per node, split by triangle centre,
 * stops at max_depth or when the split does not lower the cost.  Pure host
 * code.  triangles_out (triangle_count x 9 f32) receives the reordered
 * triangles the reference writes back over its input .bin; *nodes_out is
 * allocated by the library (free with pt_free) and holds *node_count x 9 f32
 * in _bvh.bin layout.  The output is bit-identical to the reference builder's.
 */
int pt_bvh_build(const float *triangles_in, uint32_t triangle_count, int max_depth,
                 float *triangles_out, float **nodes_out, uint32_t *node_count);
void pt_free(void *p);

/*
 * The same builder on the GPU: same arguments, same output bytes as pt_bvh_build (and so as the
 * reference tool, prepare_data/src/main.rs:14-136 + 179-220), computed by gfx950 kernels on device
 * `device_ordinal`, one pass of the triangle array per tree level instead of 30 scans per node.  For scenes
 * that are rebuilt while a program runs; the reference builds offline only.  *build_ms (may be NULL)
 * receives the GPU time between the end of the upload and the start of the download.  Synchronous.
 * Triangles with a non-finite coordinate or a negative zero are refused with PT_EINVAL (for those the
 * reference's output depends on the order of its scans; pt_bvh_build handles them).
 * pt_bvh_build_gpu_error(): message of the calling thread's last failed call.
 */
int pt_bvh_build_gpu(int device_ordinal, const float *triangles_in, uint32_t triangle_count, int max_depth,
                     float *triangles_out, float **nodes_out, uint32_t *node_count, float *build_ms);
const char *pt_bvh_build_gpu_error(void);
/* pt_bvh_build_gpu and pt_rebuild_scene keep their scratch buffers (up to 1 GiB per process) for the next build, so
 * that a program that rebuilds every frame does not pay hipMalloc / hipFree each time; this returns them to the driver. */
void pt_bvh_build_gpu_release_cache(void);

/*
 * ASCII STL -> .bin converter, replaces prepare_data/data_formatter.js:9-18.
 * Every line containing "vertex" contributes the numbers that follow its first
 * "vertex " (split at single spaces, JavaScript Number() rules: an empty piece
 * is 0, a non-number is NaN), each multiplied by `scale` in double precision
 * (data_formatter.js:5) and rounded once to f32.  *out is allocated by the
 * library (free with pt_free) and holds *n_floats values, 9 per triangle for a
 * well-formed file; like the script, nothing here checks that count.  A line
 * with "vertex" but without "vertex " makes the script throw: PT_EINVAL here.
 * pt_stl_format_file reads stl_path and writes the values as little-endian f32
 * to bin_path (data_formatter.js:15-18).  Pure host code.
 */
int pt_stl_format(const char *stl_text, size_t length, double scale, float **out, uint64_t *n_floats);
int pt_stl_format_file(const char *stl_path, const char *bin_path, double scale, uint64_t *n_floats);
/* Message of the calling thread's last failed pt_stl_format* call. */
const char *pt_stl_format_error(void);

/*
 * Camera motion of the reference host (pure host code), replaces State::update_camera
 * (src/main.rs:45-115) and its rotate_vector (src/main.rs:117-142).
 * keys = [W, S, D, A, Space, ShiftLeft, I, K, L, J, O, U] (main.rs:39, 710-721), non-zero =
 * pressed.  One call = one tick: the pressed keys move the camera by `speed` along and turn it by
 * `rot_speed` degrees about its local axes (CAMERA_SPEED = CAMERA_ROT_SPEED = 1.0, main.rs:17-18),
 * position and rotation_degrees are updated in place.  *any_key (may be NULL) receives 1 when a
 * key is pressed: the caller must then restart the accumulation with frame_count = 0
 * (main.rs:47-49) and pass the new pose to pt_set_camera.
 */
int pt_update_camera(const uint8_t keys[12], float speed, float rot_speed,
                     float position[3], float rotation_degrees[3], int *any_key);
int pt_host_rotate_vector(const float v[3], const float rotation_degrees[3], float out[3]);

#ifdef __cplusplus
}
#endif
#endif /* PT_API_H */
