/*
 * CPU oracle for the path-tracing hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a function-for-function restatement, in plain C and strict IEEE
 * binary32 arithmetic (no FMA contraction, no fast-math), of the reference
 * fragment shader  src/shaders/shader.wgsl  plus the per-frame protocol of
 * src/main.rs:444-523.  Nothing in the shipped library links, loads or calls
 * this code: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may use it, and only as the checker.
 *
 * How it is pinned.  The reference has no tests and no golden vectors, and cannot be executed as its authors
 * run it (WGSL through wgpu/naga from a Rust binary; no Rust toolchain, no Vulkan here).  Its shader TEXT can be
 * executed: tests/golden/wgsl_interp.py interprets src/shaders/shader.wgsl statement by statement in strict
 * binary32 arithmetic, and tests/golden/wgsl_golden.npz holds what the shader's own functions return (random_*,
 * rotate_vector, get_environment_light, ray_box, ray_triangle, ray_sphere, calculate_ray_collision, trace, and
 * fs_main over three whole frames of a 32 x 18 image of the teapot scene).  tests/test_wgsl_golden.py requires
 * this oracle (and, on a GPU, the HIP kernels) to reproduce every one of those values bit for bit.  What that
 * does NOT pin is the rounding of the built-ins WGSL leaves to the implementation (list below): interpreter,
 * oracle and kernels share those choices, naga and a GPU driver may make others.  Also pinned by reference-produced
 * data: teapot.bin / teapot_bvh.bin are reproduced bit for bit by the builder restatement in bvh_build_oracle.c.
 *
 * WGSL built-ins whose results the WGSL specification leaves
 * implementation-defined are fixed here as follows (the HIP kernels use the
 * identical rules):
 *   dot(a,b)        = a.x*b.x + a.y*b.y + a.z*b.z, left to right, no FMA
 *   normalize(v)    = v * (1 / sqrt(dot(v,v)))  with correctly rounded / and sqrt
 *   reflect(i,n)    = i - (2*dot(n,i)) * n
 *   min/max         = IEEE minNum/maxNum (a NaN operand is ignored)
 *   smoothstep      = t*t*(3-2*t), t = clamp((x-e0)/(e1-e0),0,1)
 *   u32(f32)        = truncation toward zero
 *   log/cos/pow/tan = libm in the default build; with -DORC_SHARED_MATH the
 *                     portable polynomial versions in pt_portable_math.h (the
 *                     same source the HIP kernels compile in their
 *                     "exact-math" mode, so results match bit for bit)
 */
#ifndef PT_ORACLE_H
#define PT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float x, y, z; } orc_v3;

/* shader.wgsl:7-16 plus one debug field (prim_id) that the shader lacks. */
typedef struct {
    int32_t did_hit;
    float distance;
    orc_v3 position;
    orc_v3 normal;
    orc_v3 color;
    orc_v3 emission_color;
    float emission_strength;
    float smoothness;
    int32_t prim_id; /* triangle index >= 0, -(1+sphere) for spheres, INT32_MIN = none */
} orc_hit;

/* shader.wgsl:18-21 */
typedef struct { orc_v3 origin, dir; } orc_ray;

/* The shader's compile-time constants (shader.wgsl:24-37, 48-54, 283-286). */
typedef struct {
    uint32_t width, height;
    uint32_t rays_per_pixel;
    uint32_t max_bounce_count;
    uint32_t stack_entries;     /* bvh_max_depth + 1 (shader.wgsl:170); checked, not used for storage */
    float fov_degrees;          /* 60; converted with pi = 3.14159 (shader.wgsl:36) */
    float tri_color[3];         /* shader.wgsl:283 */
    float tri_emission_color[3];/* shader.wgsl:284 */
    float tri_emission_strength;/* shader.wgsl:285 */
    float tri_smoothness;       /* shader.wgsl:286 */
    float sky_horizon[3], sky_zenith[3], ground[3]; /* shader.wgsl:48-50 */
    float sun_dir[3];           /* shader.wgsl:51, not normalised */
    float sun_intensity, sun_focus; /* shader.wgsl:52-53 */
    uint32_t use_environment_lighting; /* shader.wgsl:54 */
    float far_distance;         /* 1000000.0: "no hit yet" distance of calculate_ray_collision and ray_triangle_bvh (shader.wgsl:142,177) */
} orc_params;

typedef struct {
    const float *spheres;   uint32_t sphere_count;   /* 12 f32 each, main.rs:286 */
    const float *triangles; uint32_t triangle_count; /* 9 f32 each */
    const float *bvh;       uint32_t node_count;     /* 9 f32 each */
} orc_scene;

/* Work counters that define the algorithmic bytes of a ray cast. */
typedef struct {
    uint64_t ray_casts;   /* calls of calculate_ray_collision */
    uint64_t box_tests;   /* ray_box evaluations */
    uint64_t tri_tests;   /* ray_triangle evaluations */
    uint64_t nodes_popped;
    uint64_t max_stack;   /* largest stack occupancy seen */
    uint64_t paths;       /* calls of trace */
} orc_counters;

void orc_default_params(orc_params *p);

/* scalar pieces, exported for unit tests */
float orc_random_value(uint32_t seed);                         /* shader.wgsl:351-358 */
float orc_random_normal(uint32_t seed);                        /* shader.wgsl:361-366 */
void  orc_random_direction(uint32_t seed, float out[3]);       /* shader.wgsl:369-375 */
void  orc_random_hemisphere_direction(const float normal[3], uint32_t seed, float out[3]);   /* shader.wgsl:378-382, dead code in the shader */
void  orc_rotate_vector(const float v[3], const float angles_deg[3], float out[3]); /* shader.wgsl:327-348 */
float orc_ray_box(const float origin[3], const float dir[3], const float box[6]);   /* shader.wgsl:237-256 */
void  orc_ray_triangle(const orc_params *p, const float origin[3], const float dir[3],
                       const float tri[9], orc_hit *out);      /* shader.wgsl:258-290 */
void  orc_ray_sphere(const float origin[3], const float dir[3],
                     const float sphere[12], orc_hit *out);    /* shader.wgsl:292-325 */
void  orc_environment_light(const orc_params *p, const float dir[3], float out[3]); /* shader.wgsl:391-404 */
void  orc_screen_size(const orc_params *p, float *screen_width, float *screen_height); /* shader.wgsl:68-69 */
void  orc_primary_ray(const orc_params *p, const float cam_pos[3], const float cam_rot_deg[3],
                      uint32_t px, uint32_t py, float origin[3], float dir[3], uint32_t *pixel_index); /* shader.wgsl:80-95 */

/* traversal + closest hit (shader.wgsl:139-235); counters may be NULL */
void orc_ray_triangle_bvh(const orc_params *p, const orc_scene *s, const float origin[3],
                          const float dir[3], orc_hit *out, orc_counters *c);
void orc_calculate_ray_collision(const orc_params *p, const orc_scene *s, const float origin[3],
                                 const float dir[3], orc_hit *out, orc_counters *c);
/* brute force over every triangle in index order (no BVH), for property tests */
void orc_ray_triangles_brute(const orc_params *p, const orc_scene *s, const float origin[3],
                             const float dir[3], orc_hit *out);

/* shader.wgsl:111-137 */
void orc_trace(const orc_params *p, const orc_scene *s, const float origin[3], const float dir[3],
               uint32_t seed, uint32_t frame_count, float out[3], orc_counters *c);

/* One frame (shader.wgsl:78-109) over pixel rows [y0,y1) and columns [x0,x1) with
 * stride sx/sy (1 = every pixel).  accum has 4 floats per pixel (16-byte stride,
 * shader.wgsl:41), row-major, row 0 on top.  Returns 0, or -1 on bad arguments. */
int orc_render_frame(const orc_params *p, const orc_scene *s, const float cam_pos[3],
                     const float cam_rot_deg[3], uint32_t frame_count, float *accum,
                     uint32_t x0, uint32_t x1, uint32_t sx, uint32_t y0, uint32_t y1, uint32_t sy,
                     int nthreads, orc_counters *c);

/* main.rs:444-523 protocol: frames first_frame .. first_frame+n_frames-1 in order. */
int orc_render_frames(const orc_params *p, const orc_scene *s, const float cam_pos[3],
                      const float cam_rot_deg[3], uint32_t first_frame, uint32_t n_frames,
                      float *accum, int nthreads, orc_counters *c);

/* Per-frame means summed (not running-mean) over frames first, first+stride, ... (n of them):
 * the quantity a frame-sharded rank produces before the cross-rank sum. */
int orc_render_frames_sum(const orc_params *p, const orc_scene *s, const float cam_pos[3],
                          const float cam_rot_deg[3], uint32_t first_frame, uint32_t n_frames,
                          uint32_t frame_stride, float *sum, int nthreads, orc_counters *c);

/* closest-hit id and distance of the primary ray of every pixel */
int orc_primary_hits(const orc_params *p, const orc_scene *s, const float cam_pos[3],
                     const float cam_rot_deg[3], int32_t *prim_id, float *distance, int nthreads);

/* walk the node file from the root; returns depth (root = 0) or -1 if a link is
 * out of range / the walk exceeds 4096 levels (cycle) */
int orc_bvh_depth(const float *bvh, uint32_t node_count);

int orc_uses_shared_math(void);

#ifdef __cplusplus
}
#endif
#endif
