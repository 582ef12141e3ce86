/* Internal declarations shared by the host side (pt_host.cpp) and the gfx950
 * kernels (pt_kernels.hip).  Not part of the public ABI. */
#ifndef PT_INTERNAL_H
#define PT_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stdint.h>

/* Child reference stored in an inner record and on the traversal stack.
 *   bit 31 clear: index of an inner record
 *   bit 31 set  : leaf; bits 0-23 first triangle, bits 24-30 triangle count,
 *                 count field 127 = "look the count up in big_leaf_count[first]" */
#define PT_REF_LEAF 0x80000000u
#define PT_REF_COUNT_ESCAPE 127u
#define PT_REF_NONE 0xffffffffu     /* not a valid reference (a leaf of >= 127 triangles cannot start at triangle 2^24 - 1) */

/* One inner node of the reference file, repacked so that ONE 64-byte record
 * holds everything a visit needs: both child boxes and both child references.
 * (The reference reads the two 36-byte child nodes, 72 bytes at 36-byte
 * stride, and then re-reads a child when it is popped.)
 *   q0 = (A.min.x, A.min.y, A.max.x, A.max.y)
 *   q1 = (B.min.x, B.min.y, B.max.x, B.max.y)
 *   q2 = (A.min.z, A.max.z, B.min.z, B.max.z)
 *   q3 = (refA, refB, 0, 0) as raw bits
 * (x,y pairs and z pairs feed packed-f32 instructions without register shuffles)
 * Box values are copied verbatim from the file, so ray_box sees the same
 * operands as the shader. */
struct PTInnerRecord { float4 q[4]; };

/* Four-wide node (PT_FLAG_WIDE_BVH): an inner node of the file with its inner children replaced by THEIR
 * children, so one visit tests up to four boxes.  Two box blocks in the layout of PTInnerRecord (children 0, 1
 * in q0-q2, children 2, 3 in q3-q5), q6 = the four child references as raw bits (0xffffffff = no such child;
 * its box block then repeats child 0's box), q7 unused.  Box values are the file's, verbatim. */
struct PTWideRecord { float4 q[8]; };

/* One triangle in file order, with the ray-independent part of ray_triangle
 * (shader.wgsl:262-264) hoisted: a, ab = b - a, ac = c - a, n = cross(ab, ac),
 * computed on the host with the same f32 operations the shader uses.
 *   q0 = (a.x, a.y, a.z, ab.x)  q1 = (ab.y, ab.z, ac.x, ac.y)  q2 = (ac.z, n.x, n.y, n.z) */
struct PTTriRecord { float4 q[3]; };

/* A pixel whose primary ray hit something; consumed by the path kernel. */
struct PTHitRecord {
    uint32_t pixel;     /* y * width + x */
    int32_t prim_id;    /* triangle index, or -(1+sphere) */
    float distance;
    uint32_t pad;
};

/* Unsigned division by a run-time constant (Granlund-Montgomery): n / d for any 32-bit n,
 * d >= 1, as one mulhi, one add and two shifts.  Used to split a path-queue index into
 * (hit pixel, frame, sample) without integer division sequences. */
struct PTFastDiv { uint32_t d, mul, sh1, sh2; };
static inline PTFastDiv pt_fastdiv_make(uint32_t d) {
    PTFastDiv f;
    uint32_t l = 0;
    while ((1ull << l) < d) l++;
    f.d = d;
    f.mul = (uint32_t)((((1ull << l) - d) << 32) / d + 1);
    f.sh1 = l < 1 ? l : 1;
    f.sh2 = l > 0 ? l - 1 : 0;
    return f;
}

/* control words in device memory, zeroed before each batch of frames */
/* The path kernel's work queue has PT_QUEUE_PARTS heads (a multiple of 8, at most 64: PT_QUEUE_PARTS / 8 per XCD),
 * PT_HEAD_STRIDE words apart: every wave makes one atomic per round on a head, and atomics on one cache line - one
 * memory channel - are served one after the other (with eight heads in one line the benchmark scenes were bound by
 * exactly that). */
#ifndef PT_QUEUE_PARTS
#define PT_QUEUE_PARTS 32
#endif
#ifndef PT_HEAD_STRIDE
#define PT_HEAD_STRIDE 64
#endif
struct PTControl {
    uint32_t hit_count;     /* number of PTHitRecord written by the primary kernel */
    uint32_t pad0[PT_HEAD_STRIDE - 1];
    uint32_t queue_head[PT_QUEUE_PARTS * PT_HEAD_STRIDE];   /* head of part p: queue_head[p * PT_HEAD_STRIDE] */
};

struct PTKernelParams {
    uint32_t width, height, rays_per_pixel, max_bounce_count;
    uint32_t frame_first;          /* frame_count of the first frame of this batch */
    uint32_t frame_stride;         /* frame_count step between the frames of the batch */
    uint32_t batch_frames;         /* frames rendered by this batch of launches */
    PTFastDiv div_width;           /* pixel -> row */
    PTFastDiv div_rpp, div_batch;  /* queue index = (hit * batch_frames + frame) * rays_per_pixel + sample */
    uint32_t shard_rank, shard_world;   /* pixel sharding: this context renders the 8x8 tiles with (tile_x + tile_y) % world == rank */
    uint32_t accum_mode;           /* 0: running mean (shader.wgsl:104-106), 1: add the frame mean */
    float width_f, height_f, screen_w, screen_h;
    float cam_pos[3];
    float cos_a, sin_a, cos_b, sin_b, cos_c, sin_c;
    float tri_color[3], tri_emission_color[3], tri_emission_strength, tri_smoothness;
    float sky_horizon[3], sky_zenith[3], ground[3], neg_sun[3], sun_intensity, sun_focus;
    uint32_t use_env;
    float far_distance;            /* "no hit yet" distance (shader.wgsl:142,177) */
    uint32_t sphere_count;
    uint32_t root_ref;
    uint32_t stack_entries;        /* LDS traversal stack depth per lane (>= depth of the node file) */
    uint32_t xcd_queues;           /* 1: a block starts on the queue part of the XCD it runs on; 0: blockIdx % 8 */
    uint32_t leaf_min;             /* lanes that must be at a leaf before the triangle code runs (see trav_wave_step) */
    uint32_t leaf_shift;           /* ... or when they are at least 1/2^leaf_shift of the traversing lanes */
    uint32_t refill_min;           /* path kernel: lanes that must wait for shading before the wave leaves the traversal loop */
    uint32_t two_paths;            /* path kernel: two paths per lane, state in `pathbuf` (path_kernel_mb) */
    uint32_t starve_min;           /* ... a round also runs when this many lanes have nothing to walk and a buffer to shade */
    float4 *pathbuf;               /* two-paths-per-lane state: pt_pathbuf_bytes() */
    uint32_t pathbuf_bytes;
    uint32_t quad_fetch;           /* path kernel: the four lanes of a quad fetch each other's inner records (one cache line per gather and quad) */
    uint32_t fast_div;             /* every box bound is 0 or in [2^-56, 2^90]: ray_box may use reciprocals */
    const PTInnerRecord *inner;
    uint32_t tris_bytes;           /* size of the triangle-record array (at most 2^24 x 48 bytes) */
    uint32_t inner_bytes;          /* size of the inner-record array (below 2^32: the upload refuses 2^26 or more inner records) */
    const PTTriRecord *tris;
    const PTWideRecord *wide;      /* four-wide nodes, or null: the traversal follows the binary records */
    uint32_t wide_bytes;
    const float *spheres;
    const uint32_t *big_leaf_count;
    float4 *accum;
    float4 *mirror;                /* resolve kernel: the folded image also goes here (pt_next_render_output), or null */
    uint32_t mirror_move;          /* ... and the accumulation buffer is left holding zeros */
    PTHitRecord *hits;
    PTControl *ctrl;
    uint32_t *spill;               /* path kernel: traversal-stack entries beyond the LDS part, entry k of lane g at spill[k*spill_stride + g] */
    uint32_t spill_stride;
    float4 *samples;               /* radiance of every path; 64 hit pixels interleaved (sample_slot) */
    uint32_t *pixel_hit;           /* per pixel: index into hits, or 0xffffffff when the primary ray missed */
    float4 *pixel_const;           /* per miss pixel: its frame mean (the same in every frame) */
    int32_t *out_ids;
    float *out_dist;
    unsigned long long *counters;  /* 8 words, see flush_counters */
};

/* launchers implemented in pt_kernels.hip; kp.stack_entries sizes the LDS traversal stacks */
hipError_t pt_launch_primary(const PTKernelParams &kp, bool exact_math,
                             bool count_work, bool aov_only, hipStream_t stream);
/* blocks_per_cu_cap: 0 = as many blocks per CU as stay resident (the persistent grid), else at most that many */
hipError_t pt_launch_paths(const PTKernelParams &kp, bool exact_math,
                           bool count_work, int num_cus, int blocks_per_cu_cap, hipStream_t stream);
/* blocks per CU of the path kernel's full grid */
int pt_path_blocks_per_cu_max(void);
hipError_t pt_launch_resolve(const PTKernelParams &kp, hipStream_t stream);
hipError_t pt_launch_scale(float4 *accum, uint32_t n_pixels, float s, hipStream_t stream);
hipError_t pt_launch_probe_intersections(const float *rays, const float *boxes, const float *tris, const float *spheres,
                                         uint32_t n, float *out, hipStream_t stream);
hipError_t pt_launch_probe(const PTKernelParams &kp, bool exact, const uint32_t *seeds, const float *dirs,
                           const float normal[3], uint32_t n, float *out, hipStream_t stream);
/* refit, implemented in pt_refit.hip: triangle records from 9-float triangles; the child boxes of the inner
 * records order[begin .. begin + count) (all of one height) from their children */
hipError_t pt_launch_refit_tris(const float *tri9, uint32_t n, PTTriRecord *out, hipStream_t stream);
hipError_t pt_launch_refit_level(PTInnerRecord *inner, const uint32_t *order, uint32_t begin, uint32_t count, const float *tri9,
                                 const uint32_t *big, uint32_t *not_tame, hipStream_t stream);
int pt_max_stack_entries(void);
uint32_t pt_lane_stride(int num_cus);   /* upper bound of the path kernel's grid size in lanes */
uint32_t pt_spill_entries(uint32_t stack_entries);
size_t pt_pathbuf_bytes(int num_cus);   /* size of PTKernelParams::pathbuf */

#endif
