/*
 * gfx950 kernels of the path-tracing hot path.
 *
 * What is computed is fixed by the reference fragment shader
 * src/shaders/shader.wgsl (function by function citations below); how it is
 * computed is MI355X-specific:
 *
 *   primary_kernel  one lane per pixel, one wave per 8x8 tile.  Every sample of
 *                   a pixel shares the same primary ray (the shader has no
 *                   sub-pixel jitter, shader.wgsl:80-101) and the camera is fixed
 *                   within a call, so the first cast is done once per pixel and
 *                   batch of frames.  Miss pixels get their constant radiance
 *                   here; hit pixels are appended to a compact list.
 *   path_kernel     persistent waves, one launch per batch of frames; each lane
 *                   owns one path (hit pixel, frame, sample) from its first
 *                   bounce to termination and refills itself from a global queue
 *                   (wave ballot + one atomic per refill), so lanes stay busy
 *                   although paths have very different lengths.  BVH traversal is
 *                   resumable per lane; stacks live in LDS, one column per lane.
 *   resolve_kernel  per pixel: adds the samples of every frame of the batch in
 *                   the shader's order and applies the progressive accumulation
 *                   (shader.wgsl:99-106) frame by frame.
 *
 * All arithmetic that decides a hit is plain IEEE binary32 in the shader's
 * evaluation order with FMA contraction off, so closest-hit ids and distances
 * are reproducible bit for bit on a CPU.
 */
#include "pt_internal.h"
#include <cstdlib>
#include "pt_portable_math.h"
#include "pt_exact_div.h"

#pragma clang fp contract(off)

#define PT_BOX_MISS 999999999.0f   /* shader.wgsl:254 */
#define PT_PI 3.14159f             /* shader.wgsl:36,365 */
#define PT_NONE INT32_MIN
#define PT_PIXEL_MISS 0xffffffffu       /* pixel_hit: primary ray hit nothing */
#define PT_PIXEL_NOT_MINE 0xfefefefeu   /* pixel_hit: pixel belongs to another rank's shard (the host fills the array with 0xfe bytes) */

#define PT_WAVES_PER_BLOCK 4
#define PT_BLOCK (64 * PT_WAVES_PER_BLOCK)
#ifndef PT_PATH_WAVES_PER_BLOCK
#define PT_PATH_WAVES_PER_BLOCK 4            /* the path kernel uses no block-level communication: any size works */
#endif
#define PT_PATH_BLOCK (64 * PT_PATH_WAVES_PER_BLOCK)

struct V3 { float x, y, z; };
__device__ __forceinline__ V3 mk(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator*(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ float dot3(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross3(V3 a, V3 b) {
    return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
/* normalize(v) = v * (1/sqrt(dot(v,v))), correctly rounded / and sqrt */
__device__ __forceinline__ V3 normalize3(V3 a) {
    float inv = 1.0f / sqrtf(dot3(a, a));
    return a * inv;
}
__device__ __forceinline__ V3 ld3(const float *p) { return mk(p[0], p[1], p[2]); }
/* shader.wgsl:385-388 */
__device__ __forceinline__ V3 lerp3(V3 a, V3 b, float t) { return a * (1.0f - t) + b * t; }
/* IEEE minNum / maxNum: v_min_f32 / v_max_f32 */
__device__ __forceinline__ float fmin_(float a, float b) { return __builtin_fminf(a, b); }
__device__ __forceinline__ float fmax_(float a, float b) { return __builtin_fmaxf(a, b); }

/* ---- transcendental functions: two interchangeable sets ---- */
template <bool EXACT> __device__ __forceinline__ float t_log(float x) {
    if (EXACT) return pm_logf(x);
    return __builtin_amdgcn_logf(x) * 0.69314718056f;           /* v_log_f32 is log2 */
}
template <bool EXACT> __device__ __forceinline__ float t_cos(float x) {
    if (EXACT) return pm_cosf(x);
    return __builtin_amdgcn_cosf(x * 0.15915494309f);           /* v_cos_f32 takes revolutions */
}
template <bool EXACT> __device__ __forceinline__ float t_pow(float x, float y) {
    if (EXACT) return pm_powf(x, y);
    return __builtin_amdgcn_exp2f(y * __builtin_amdgcn_logf(x));
}

struct Counters { unsigned long long casts, boxes, tris, paths; };

/* ---- RNG, shader.wgsl:351-375 ---- */
__device__ __forceinline__ float random_value(uint32_t state) {
    state = state ^ (state << 13);
    state = state ^ (state >> 17);
    state = state ^ (state << 5);
    return (float)state / 4294967296.0f;
}
template <bool EXACT> __device__ __forceinline__ float random_normal(uint32_t seed) {
    float u1 = random_value(seed);
    float u2 = random_value(seed * 7462u);
    return sqrtf(-2.0f * t_log<EXACT>(u1)) * t_cos<EXACT>(2.0f * PT_PI * u2);
}
template <bool EXACT> __device__ __forceinline__ V3 random_direction(uint32_t seed) {
    float x = random_normal<EXACT>(seed);
    float y = random_normal<EXACT>(seed * 379u);
    float z = random_normal<EXACT>(seed * 123u);
    return normalize3(mk(x, y, z));
}

/* ---- camera, shader.wgsl:80-95 and 327-348 (trig values come from the host) ---- */
/* random_hemisphere_direction, shader.wgsl:378-382: dead code in the shader, reachable here through
 * the function probe only.  WGSL sign(): 1, 0 or -1. */
template <bool EXACT> __device__ __forceinline__ V3 random_hemisphere_direction(V3 normal, uint32_t seed) {
    V3 dir = random_direction<EXACT>(seed);
    float d = dot3(normal, dir);
    float sg = (d > 0.0f) ? 1.0f : ((d < 0.0f) ? -1.0f : 0.0f);
    return dir * sg;
}

__device__ __forceinline__ V3 rotate_vector(const PTKernelParams &P, V3 r) {
    float x = r.x, y = r.y, z = r.z;
    float cos_a = P.cos_a, sin_a = P.sin_a, cos_b = P.cos_b, sin_b = P.sin_b, cos_c = P.cos_c, sin_c = P.sin_c;
    float x_rot = x * cos_c * cos_b + y * (cos_c * sin_b * sin_a - sin_c * cos_a) + z * (cos_c * sin_b * cos_a + sin_c * sin_a);
    float y_rot = x * sin_c * cos_b + y * (sin_c * sin_b * sin_a + cos_c * cos_a) + z * (sin_c * sin_b * cos_a - cos_c * sin_a);
    float z_rot = -x * sin_b + y * cos_b * sin_a + z * cos_b * cos_a;
    return mk(x_rot, y_rot, z_rot);
}

__device__ __forceinline__ void primary_ray(const PTKernelParams &P, uint32_t px, uint32_t py,
                                            V3 &dir, uint32_t &pixel_index) {
    float pos_x = (float)px + 0.5f, pos_y = (float)py + 0.5f;
    float u = (2.0f * pos_x / P.width_f - 1.0f) * P.screen_w / 2.0f;
    float v = (1.0f - 2.0f * pos_y / P.height_f) * P.screen_h / 2.0f;
    pixel_index = (uint32_t)(pos_x + pos_y * P.width_f);       /* f32 arithmetic, shader.wgsl:83 */
    dir = rotate_vector(P, normalize3(mk(u, v, -1.0f)));
}

/* ---- ray_box, shader.wgsl:237-256, for BOTH children of an inner node at once ----
 * The two boxes ride in the two halves of packed-f32 instructions (v_pk_add_f32,
 * v_pk_mul_f32, v_pk_fma_f32), which is why the record stores them interleaved.
 * FAST selects how the quotients are formed: the plain correctly rounded
 * division, or pt_div_by_reciprocal2's multiply + three FMAs with y = 1/d and its
 * low part precomputed per ray (same bits).  The caller picks FAST only for rays and
 * scenes that satisfy the predicates in pt_exact_div.h. */
typedef float f2 __attribute__((ext_vector_type(2)));
typedef unsigned int u4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 mk2(float a, float b) { f2 r; r.x = a; r.y = b; return r; }
__device__ __forceinline__ float4 u4_as_f4(u4_t w) { return make_float4(__uint_as_float(w.x), __uint_as_float(w.y), __uint_as_float(w.z), __uint_as_float(w.w)); }
__device__ __forceinline__ u4_t f4_as_u4(float4 v) { u4_t w; w.x = __float_as_uint(v.x); w.y = __float_as_uint(v.y); w.z = __float_as_uint(v.z); w.w = __float_as_uint(v.w); return w; }

template <bool FAST>
__device__ __forceinline__ f2 div2(f2 n, f2 b, f2 y, f2 yl) {
    if (FAST) {
#ifdef PT_ABL_APPROX_DIV    /* timing probe only: the uncorrected product (results are no longer the shader's) */
        return n * y;
#endif
        /* pt_div_by_reciprocal2, two quotients at a time */
        f2 t = n * yl;
        f2 q = __builtin_elementwise_fma(n, y, t);
        f2 r = __builtin_elementwise_fma(-b, q, n);
        return __builtin_elementwise_fma(r, y, q);
    }
    return mk2(n.x / b.x, n.y / b.y);
}

/* t_near, t_far of the slab test (shader.wgsl:244-250) */
__device__ __forceinline__ void box_near_far(float tminx, float tminy, float tminz, float tmaxx, float tmaxy, float tmaxz,
                                             float &t_near, float &t_far) {
    float t1x = fmin_(tminx, tmaxx), t1y = fmin_(tminy, tmaxy), t1z = fmin_(tminz, tmaxz);
    float t2x = fmax_(tminx, tmaxx), t2y = fmax_(tminy, tmaxy), t2z = fmax_(tminz, tmaxz);
    t_near = fmax_(fmax_(t1x, t1y), t1z);
    t_far = fmin_(fmin_(t2x, t2y), t2z);
}
/* Conditions as lane masks in scalar registers: the compare writes the mask, the logic on masks runs on the scalar
 * unit, and in_mask() hands a mask back as a per-lane condition without a vector instruction.  (Written as `bool`
 * selects, the compiler turns such logic into 0/1 values in vector registers.) */
typedef unsigned long long lanemask;
__device__ __forceinline__ lanemask lanes_where(bool c) { return __builtin_amdgcn_ballot_w64(c); }
__device__ __forceinline__ bool in_mask(lanemask m) { return __builtin_amdgcn_inverse_ballot_w64(m); }

/* did_hit of ray_box, shader.wgsl:252 */
__device__ __forceinline__ bool box_hit(float t_near, float t_far) { return (t_near <= t_far) & (t_far >= 0.0f); }
__device__ __forceinline__ float box_result(float tminx, float tminy, float tminz, float tmaxx, float tmaxy, float tmaxz) {
    float t_near, t_far;
    box_near_far(tminx, tminy, tminz, tmaxx, tmaxy, tmaxz, t_near, t_far);
    return box_hit(t_near, t_far) ? t_near : PT_BOX_MISS;
}

/* The ray as the box test wants it: x and y ride together in packed instructions, z is
 * paired with itself (min plane and max plane of one box). */
struct RayPk { f2 oxy, ozz, dxy, dzz, yxy, yzz, ylxy, ylzz; };      /* y = 1/d, yl = pt_reciprocal_low(d, y) */

/* q0 = (A.min.x, A.min.y, A.max.x, A.max.y)  q1 = (B.min.x, B.min.y, B.max.x, B.max.y)
 * q2 = (A.min.z, A.max.z, B.min.z, B.max.z) */
struct Box2 { float near_a, far_a, near_b, far_b; };      /* t_near / t_far of the two children's slab tests */
template <bool FAST>
__device__ __forceinline__ Box2 ray_box2_nf(const RayPk &R, float4 q0, float4 q1, float4 q2) {
    f2 a_min = div2<FAST>(mk2(q0.x, q0.y) - R.oxy, R.dxy, R.yxy, R.ylxy);
    f2 a_max = div2<FAST>(mk2(q0.z, q0.w) - R.oxy, R.dxy, R.yxy, R.ylxy);
    f2 b_min = div2<FAST>(mk2(q1.x, q1.y) - R.oxy, R.dxy, R.yxy, R.ylxy);
    f2 b_max = div2<FAST>(mk2(q1.z, q1.w) - R.oxy, R.dxy, R.yxy, R.ylxy);
    f2 a_z = div2<FAST>(mk2(q2.x, q2.y) - R.ozz, R.dzz, R.yzz, R.ylzz);
    f2 b_z = div2<FAST>(mk2(q2.z, q2.w) - R.ozz, R.dzz, R.yzz, R.ylzz);
    Box2 r;
    box_near_far(a_min.x, a_min.y, a_z.x, a_max.x, a_max.y, a_z.y, r.near_a, r.far_a);
    box_near_far(b_min.x, b_min.y, b_z.x, b_max.x, b_max.y, b_z.y, r.near_b, r.far_b);
    return r;
}
template <bool FAST>
__device__ __forceinline__ void ray_box2(const RayPk &R, float4 q0, float4 q1, float4 q2, float &dst_a, float &dst_b) {
    const Box2 b = ray_box2_nf<FAST>(R, q0, q1, q2);
    dst_a = box_hit(b.near_a, b.far_a) ? b.near_a : PT_BOX_MISS;
    dst_b = box_hit(b.near_b, b.far_b) ? b.near_b : PT_BOX_MISS;
}

/* may this lane's ray use the reciprocal form?  (scene-side condition: P.fast_div) */
__device__ __forceinline__ bool ray_div_safe(V3 o, V3 d) {
    return pt_div_safe_denominator(d.x) && pt_div_safe_denominator(d.y) && pt_div_safe_denominator(d.z) &&
           pt_div_tame_coordinate(o.x) && pt_div_tame_coordinate(o.y) && pt_div_tame_coordinate(o.z);
}

/* All three quads of a triangle record are requested before the facing test: left alone the compiler
 * fetches the normal, waits, and only then fetches the rest - two memory round trips per triangle. */
#define PT_KEEP_LOADS_TOGETHER(a, b, c) asm volatile("" : "+v"(a.x), "+v"(b.x), "+v"(c.x))

/* ---- ray_triangle distance part, shader.wgsl:258-280 (edges and normal precomputed) ---- */
__device__ __forceinline__ bool ray_triangle_q(V3 o, V3 d, float4 q0, float4 q1, float4 q2, float &dst_out);
__device__ __forceinline__ bool ray_triangle(V3 o, V3 d, const PTTriRecord *t, float &dst_out) {
    float4 q0 = t->q[0], q1 = t->q[1], q2 = t->q[2];
    return ray_triangle_q(o, d, q0, q1, q2, dst_out);
}
__device__ __forceinline__ bool ray_triangle_q(V3 o, V3 d, float4 q0, float4 q1, float4 q2, float &dst_out) {
    PT_KEEP_LOADS_TOGETHER(q0, q1, q2);
    V3 a = mk(q0.x, q0.y, q0.z), edge_ab = mk(q0.w, q1.x, q1.y), edge_ac = mk(q1.z, q1.w, q2.x);
    V3 normal_vector = mk(q2.y, q2.z, q2.w);
    float determinant = -dot3(d, normal_vector);
    if (!(determinant >= 0.0001f)) return false;      /* did_hit needs it (shader.wgsl:279); nothing else is kept on a miss */
    V3 ao = o - a;
    V3 dao = cross3(ao, d);
    float inv_det = 1.0f / determinant;
    float dst = dot3(ao, normal_vector) * inv_det;
    float u = dot3(edge_ac, dao) * inv_det;
    float v = -dot3(edge_ab, dao) * inv_det;
    float w = 1.0f - u - v;
    dst_out = dst;
    return dst >= 0.0f && u >= 0.0f && v >= 0.0f && w >= 0.0f;
}

/* ---- ray_sphere distance part, shader.wgsl:292-315 ---- */
__device__ __forceinline__ bool ray_sphere(V3 o, V3 d, const float *s, float &dst_out) {
    V3 oc = o - ld3(s);
    float radius = s[3];
    float a = dot3(d, d);
    float b = 2.0f * dot3(oc, d);
    float c = dot3(oc, oc) - radius * radius;
    float discriminant = b * b - 4.0f * a * c;
    if (discriminant >= 0.0f) {
        float distance = (-b - sqrtf(discriminant)) / (2.0f * a);
        if (distance >= 0.0f) { dst_out = distance; return true; }
    }
    return false;
}

/*
 * ray_triangle_bvh, shader.wgsl:169-235, on the repacked records, as a resumable
 * state machine: trav_step() visits exactly one node.  Visit order, near/far
 * rule (`dst_a < dst_b` else b is near), strict `<` pruning against the running
 * best and "no re-test when a pushed node is popped" are the shader's.  The near
 * child is kept in `cur` instead of being pushed and popped again; that is the
 * same order.  stk points at this lane's column of the LDS stack (entries 64
 * dwords apart, so a wave's accesses never bank-conflict).
 */

struct Trav {
    uint32_t cur;      /* node to visit next, or PT_REF_NONE: pop one */
    uint32_t sp;       /* stack pointer, see stack_push */
    float best_t;
    int32_t best_id;
    RayPk R;           /* origin, direction and 1/direction in the layout ray_box2 wants */
};

/* the low parts of the reciprocals (pt_reciprocal_low), from d and y */
__device__ __forceinline__ void raypk_low(RayPk &R) {
    R.ylxy = __builtin_elementwise_fma(-R.dxy, R.yxy, mk2(1.0f, 1.0f)) * R.yxy;
    R.ylzz = __builtin_elementwise_fma(-R.dzz, R.yzz, mk2(1.0f, 1.0f)) * R.yzz;
}
__device__ __forceinline__ void raypk_set(RayPk &R, V3 o, V3 d) {
    R.oxy = mk2(o.x, o.y); R.ozz = mk2(o.z, o.z);
    R.dxy = mk2(d.x, d.y); R.dzz = mk2(d.z, d.z);
    float iz = 1.0f / d.z;
    R.yxy = mk2(1.0f / d.x, 1.0f / d.y); R.yzz = mk2(iz, iz);
    raypk_low(R);
}

/* origin and direction only (a ray that is shaded, not cast) */
__device__ __forceinline__ void raypk_set_od(RayPk &R, V3 o, V3 d) {
    R.oxy = mk2(o.x, o.y); R.ozz = mk2(o.z, o.z);
    R.dxy = mk2(d.x, d.y); R.dzz = mk2(d.z, d.z);
}

__device__ __forceinline__ V3 ray_origin(const RayPk &R) { return mk(R.oxy.x, R.oxy.y, R.ozz.x); }
__device__ __forceinline__ V3 ray_dir(const RayPk &R) { return mk(R.dxy.x, R.dxy.y, R.dzz.x); }

/* T.R must hold the ray */
__device__ __forceinline__ void trav_begin(const PTKernelParams &P, uint32_t stack_bottom_offset, Trav &T) {
    T.cur = P.root_ref;
    T.sp = stack_bottom_offset;
    T.best_t = P.far_distance;
    T.best_id = PT_NONE;
}



/*
 * Dynamic LDS of both traversal kernels: one traversal stack per lane, stack[wave][entry][lane]
 * (entries 64 dwords apart, so a wave's pushes and pops never bank-conflict).
 *
 * Measured on MI355X and rejected: (1) a copy of the top ~1000 records of the tree in LDS (40 % of
 * all visits) - no gain; (2) quad-cooperative record fetches through LDS-DMA - 20 % slower (longer
 * dependent chain per step); (3) prefetching both children while the box tests run - 25 % slower.
 * Probes that added work without changing results (a second box test, four more loads per visit) put the cost of the four 16-byte
 * loads of a visit at about 27 % of the kernel and that of the two box tests at about 16 %; the
 * loads cost the same whether they hit or miss and whether they fetch 8 or 16 bytes, so it is the
 * number of gather instructions that counts.
 */
extern __shared__ uint32_t pt_lds[];
__device__ __forceinline__ uint32_t lane_id();

struct WaveMem {
    uint32_t xchg;          /* byte offset in pt_lds of this wave's record exchange area (quad fetch only; wave-uniform) */
    uint32_t base;          /* byte offset in pt_lds of this wave's stack rows (wave-uniform) */
    uint32_t limit;         /* base + 256 * entries kept in LDS (wave-uniform); deeper entries go to `spill` */
    uint32_t *spill;        /* overflow entries in global memory (SPILL variants only; wave-uniform) */
    uint32_t spill_lane;    /* this lane's column there */
    uint32_t spill_stride;
};

/* The stack pointer T.sp is the LDS byte offset of the next free entry of this lane's column:
 * base + 4 * lane + 256 * depth.  Empty <=> T.sp < base + 256, a compare against a scalar. */
__device__ __forceinline__ uint32_t stack_bottom(const WaveMem &M) { return M.base + 4u * lane_id(); }
__device__ __forceinline__ bool stack_empty(const WaveMem &M, const Trav &T) { return T.sp < M.base + 256u; }
__device__ __forceinline__ uint32_t *lds_word(uint32_t byte_offset) {
    return (uint32_t *)((char *)pt_lds + byte_offset);
}

/* SPILL = false: the whole stack is in LDS (plain ds_write / ds_read, no address-space choice). */
template <bool SPILL>
__device__ __forceinline__ void stack_push(const WaveMem &M, Trav &T, uint32_t v) {
    if (!SPILL || T.sp < M.limit) *lds_word(T.sp) = v;
    else M.spill[((T.sp - M.limit) >> 8) * M.spill_stride + M.spill_lane] = v;
    T.sp += 256u;
}

/* after a visit: continue with the near child if one was kept, else pop; true = stack empty */
template <bool SPILL>
__device__ __forceinline__ bool trav_advance(const WaveMem &M, Trav &T) {
    if (T.cur != PT_REF_NONE) return false;
    if (stack_empty(M, T)) return true;
    T.sp -= 256u;
    if (!SPILL || T.sp < M.limit) T.cur = *lds_word(T.sp);
    else T.cur = M.spill[((T.sp - M.limit) >> 8) * M.spill_stride + M.spill_lane];
    return false;
}

/*
 * Quad fetch of the inner records.  A 64-lane gather of 16 bytes per lane costs the CU's address unit 16 cycles
 * plus about 0.6 for every cache line beyond one per quad of lanes, and a visit needs four of them: with every
 * lane fetching its own record that is 4 x 44 cycles for 64 unrelated records, with the four lanes of a quad
 * fetching each other's records it is 4 x 16.  Gather k reads the record of the quad's lane k, one 16-byte piece
 * per lane (four lanes, one line), straight into LDS: buffer_load_dwordx4 ... lds puts lane l's piece at M0 + 16 l,
 * so the record of quad g's lane k is the 64 bytes at area k + 64 g, and every lane reads its own record back with
 * ds_read_b128.  The four areas are 1040 bytes apart, so that the four lanes of a quad read from different banks.
 * Lanes without an inner node pass an offset beyond the array: the buffer descriptor returns zeros for those
 * without touching memory.  All 64 lanes of the wave must call quad_fetch together.
 */
#define PT_XCHG_AREA 1040u
#define PT_XCHG_BYTES (4u * PT_XCHG_AREA)
#define PT_NO_RECORD_OFFSET 0xffffffc0u     /* past the end of any inner-record array (the upload refuses 2^26 records) */
#define PT_DPP_QUAD_LANE(k) ((k) * 0x55)

/* Where the calling lane finds its own record: area (lane & 3), quad (lane >> 2) = xchg + 16 lane + 1024 (lane & 3).
 * Worked out from the lane number where it is needed (five instructions): kept in a register across the kernel it
 * ends up in scratch memory, one more vector-memory instruction per step. */
__device__ __forceinline__ uint32_t quad_fetch_own(const WaveMem &M) {
    uint32_t lane;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
    const uint32_t t = lane * 16u + M.xchg;            /* xchg is a multiple of 64 */
    return ((t & 0x30u) << 6) + t;
}

__device__ __forceinline__ void quad_fetch(const PTKernelParams &P, const WaveMem &M, uint32_t own, uint32_t record_offset) {
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)P.inner, 0, (int)P.inner_bytes, 0x00020000);
    typedef __attribute__((address_space(3))) void *lds_ptr;
    char *area = (char *)pt_lds + M.xchg;
    const uint32_t piece = own & 0x30u;         /* 16 x the lane's place in its quad: the 16 bytes of a record this lane fetches */
    const uint32_t a0 = (uint32_t)__builtin_amdgcn_mov_dpp((int)record_offset, PT_DPP_QUAD_LANE(0), 0xf, 0xf, true) + piece;
    const uint32_t a1 = (uint32_t)__builtin_amdgcn_mov_dpp((int)record_offset, PT_DPP_QUAD_LANE(1), 0xf, 0xf, true) + piece;
    const uint32_t a2 = (uint32_t)__builtin_amdgcn_mov_dpp((int)record_offset, PT_DPP_QUAD_LANE(2), 0xf, 0xf, true) + piece;
    const uint32_t a3 = (uint32_t)__builtin_amdgcn_mov_dpp((int)record_offset, PT_DPP_QUAD_LANE(3), 0xf, 0xf, true) + piece;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr)(area), 16, a0, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr)(area + PT_XCHG_AREA), 16, a1, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr)(area + 2u * PT_XCHG_AREA), 16, a2, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr)(area + 3u * PT_XCHG_AREA), 16, a3, 0, 0, 0);
}

/* the calling lane's record, once the gathers of quad_fetch have landed */
__device__ __forceinline__ void quad_fetch_read(uint32_t own_offset, float4 &q0, float4 &q1, float4 &q2, uint2 &refs) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const char *own = (const char *)pt_lds + own_offset;
    q0 = *(const float4 *)(own);
    q1 = *(const float4 *)(own + 16);
    q2 = *(const float4 *)(own + 32);
    refs = *(const uint2 *)(own + 48);
}

template <bool FAST, bool COUNT, bool SPILL, bool COOP = false>
__device__ __forceinline__ void inner_visit(const PTKernelParams &P, const WaveMem &M, Trav &T, Counters &cn, uint32_t own = 0) {
    float4 q0, q1, q2;
    uint2 refs;
    if (COOP) {
        quad_fetch_read(own, q0, q1, q2, refs);       /* the wave has called quad_fetch */
    } else {
        /* buffer loads: scalar base + 32-bit offset (no 64-bit address arithmetic per lane), and a record index
         * outside the array reads zeros instead of faulting */
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)P.inner, 0, (int)P.inner_bytes, 0x00020000);
        const uint32_t off = T.cur * 64u;
        const u4_t w0 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
        const u4_t w1 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + 16u, 0, 0);
        const u4_t w2 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + 32u, 0, 0);
        const u2_t w3 = __builtin_amdgcn_raw_buffer_load_b64(rsrc, off + 48u, 0, 0);
        q0 = make_float4(__uint_as_float(w0.x), __uint_as_float(w0.y), __uint_as_float(w0.z), __uint_as_float(w0.w));
        q1 = make_float4(__uint_as_float(w1.x), __uint_as_float(w1.y), __uint_as_float(w1.z), __uint_as_float(w1.w));
        q2 = make_float4(__uint_as_float(w2.x), __uint_as_float(w2.y), __uint_as_float(w2.z), __uint_as_float(w2.w));
        refs = make_uint2(w3.x, w3.y);
    }
    const Box2 bx = ray_box2_nf<FAST>(T.R, q0, q1, q2);
    if (COUNT) cn.boxes += 2;
    const uint32_t ref_a = refs.x, ref_b = refs.y;
    /* did_hit of ray_box, shader.wgsl:252, as lane masks */
    const lanemask m_hit_a = lanes_where(bx.near_a <= bx.far_a) & lanes_where(bx.far_a >= 0.0f);
    const lanemask m_hit_b = lanes_where(bx.near_b <= bx.far_b) & lanes_where(bx.far_b >= 0.0f);
    /* shader.wgsl:205-230 with ray_box's result (`did_hit ? t_near : 999999999.0`) never materialised.  A missed box
     * is never visited: its 999999999.0 is not below a best distance that starts at far_distance <= 999999999.0
     * (pt_create refuses larger values) and only shrinks.  So "may be visited" is hit & (t_near < best), and which
     * of the two boxes is "near" matters only when both may be visited; then both were hit and `dst_a < dst_b` is
     * t_near_a < t_near_b.  The conditions stay lane masks (scalar logic); the vector ALU does seven compares. */
    const lanemask m_ok_a = m_hit_a & lanes_where(bx.near_a < T.best_t), m_ok_b = m_hit_b & lanes_where(bx.near_b < T.best_t);
    const lanemask m_a_near = m_hit_a & (~m_hit_b | lanes_where(bx.near_a < bx.near_b));
    const bool a_near = in_mask(m_a_near);
    const bool go_near = in_mask((m_a_near & m_ok_a) | (~m_a_near & m_ok_b));
    const bool push_far = in_mask((m_a_near & m_ok_b) | (~m_a_near & m_ok_a));
    const uint32_t ref_near = a_near ? ref_a : ref_b, ref_far = a_near ? ref_b : ref_a;
    if (push_far) stack_push<SPILL>(M, T, ref_far);
    T.cur = go_near ? ref_near : PT_REF_NONE;
}

/* compare-exchange of two (distance, reference) pairs: afterwards da <= db */
__device__ __forceinline__ void order2(float &da, uint32_t &ra, float &db, uint32_t &rb) {
    const bool swap = db < da;
    const float dl = swap ? db : da, dh = swap ? da : db;
    const uint32_t rl = swap ? rb : ra, rh = swap ? ra : rb;
    da = dl; db = dh; ra = rl; rb = rh;
}

/*
 * One visit of a four-wide node (PT_FLAG_WIDE_BVH; no counterpart in the shader, whose tree is binary,
 * shader.wgsl:169-235).  The same ray_box arithmetic on the same box values, up to four boxes per visit;
 * children are visited nearest first (five compare-exchanges), the others are pushed farthest first, each
 * only if its entry distance is below the running best, as in the binary walk.  What differs from the
 * binary walk: the boxes of the inner nodes that were folded away are never tested, and among equal
 * distances the order is this function's, so a closest hit can differ where the shader's result hangs on
 * an exact tie or on a box test that rejects a triangle the ray does hit.
 */
template <bool FAST, bool SPILL>
__device__ __forceinline__ void wide_visit(const PTKernelParams &P, const WaveMem &M, Trav &T) {
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)P.wide, 0, (int)P.wide_bytes, 0x00020000);
    const uint32_t off = T.cur * 128u;
    u4_t w[7];
#pragma unroll
    for (int k = 0; k < 7; k++) w[k] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + 16u * k, 0, 0);
    float4 q[6];
#pragma unroll
    for (int k = 0; k < 6; k++)
        q[k] = make_float4(__uint_as_float(w[k].x), __uint_as_float(w[k].y), __uint_as_float(w[k].z), __uint_as_float(w[k].w));
    float d0, d1, d2, d3;
    ray_box2<FAST>(T.R, q[0], q[1], q[2], d0, d1);
    ray_box2<FAST>(T.R, q[3], q[4], q[5], d2, d3);
    uint32_t r0 = w[6].x, r1 = w[6].y, r2 = w[6].z, r3 = w[6].w;
    if (r2 == PT_REF_NONE) d2 = PT_BOX_MISS;         /* a node has two to four children */
    if (r3 == PT_REF_NONE) d3 = PT_BOX_MISS;
    order2(d0, r0, d1, r1);
    order2(d2, r2, d3, r3);
    order2(d0, r0, d2, r2);
    order2(d1, r1, d3, r3);
    order2(d1, r1, d2, r2);
    if (d3 < T.best_t) stack_push<SPILL>(M, T, r3);
    if (d2 < T.best_t) stack_push<SPILL>(M, T, r2);
    if (d1 < T.best_t) stack_push<SPILL>(M, T, r1);
    T.cur = (d0 < T.best_t) ? r0 : PT_REF_NONE;
}

template <bool COUNT>
__device__ __forceinline__ void leaf_visit(const PTKernelParams &P, Trav &T, Counters &cn) {
    const V3 o = ray_origin(T.R), d = ray_dir(T.R);
    uint32_t first = T.cur & 0x00ffffffu;
    uint32_t cnt = (T.cur >> 24) & 0x7fu;
    if (cnt == PT_REF_COUNT_ESCAPE) cnt = P.big_leaf_count[first];
    for (uint32_t i = first; i < first + cnt; i++) {
        float dst;
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)P.tris, 0, (int)P.tris_bytes, 0x00020000);
        const u4_t w0 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, i * 48u, 0, 0);
        const u4_t w1 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, i * 48u + 16u, 0, 0);
        const u4_t w2 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, i * 48u + 32u, 0, 0);
        bool hit = ray_triangle_q(o, d, make_float4(__uint_as_float(w0.x), __uint_as_float(w0.y), __uint_as_float(w0.z), __uint_as_float(w0.w)),
                                  make_float4(__uint_as_float(w1.x), __uint_as_float(w1.y), __uint_as_float(w1.z), __uint_as_float(w1.w)),
                                  make_float4(__uint_as_float(w2.x), __uint_as_float(w2.y), __uint_as_float(w2.z), __uint_as_float(w2.w)), dst);
        if (COUNT) cn.tris++;
        if (hit && dst < T.best_t) { T.best_t = dst; T.best_id = (int32_t)i; }
    }
    T.cur = PT_REF_NONE;
}

struct WaveStats { unsigned long long steps, inner_lanes, leaf_lanes, leaf_runs; };

/*
 * One step of a whole wave.
 * Lanes with `traversing` set hold a valid T.cur.  Lanes at an inner node always advance; lanes
 * at a leaf advance only in some steps and otherwise wait with their state untouched.  With
 * leaf_turn < 0 they advance when at least P.leaf_min lanes of the wave are at a leaf, or they are
 * at least 1/2^P.leaf_shift of the traversing lanes, or no lane has inner work; with leaf_turn >= 0
 * the caller decides (1 = this step takes the leaves).  Waiting changes nothing a
 * lane computes (its own visit order is the shader's); it keeps the triangle code from running for
 * one or two lanes in every step.  Returns true for lanes whose traversal completed in this step.
 */
template <bool FAST, bool COUNT, bool SPILL, bool WIDE = false, bool COOP = false>
__device__ __forceinline__ bool trav_wave_step(const PTKernelParams &P, const WaveMem &M, bool traversing,
                                               Trav &T, Counters &cn, WaveStats *ws = nullptr, int leaf_turn = -1) {
    const bool is_leaf = traversing && (T.cur & PT_REF_LEAF) != 0u;
    bool is_inner = traversing && !is_leaf;
    const unsigned long long inner_mask = __ballot(is_inner);
    const unsigned long long leaf_mask = __ballot(is_leaf);
    const uint32_t n_leaf = (uint32_t)__popcll(leaf_mask), n_inner = (uint32_t)__popcll(inner_mask);
    bool run_leaves;
    if (leaf_turn < 0) run_leaves = !inner_mask || n_leaf >= P.leaf_min || (n_leaf << P.leaf_shift) >= n_leaf + n_inner;
    else run_leaves = leaf_turn > 0 && leaf_mask;       /* the path kernel gives leaves fixed turns */
    bool done = false;
    /* In a leaf turn the leaves go first: a lane that pops an inner node after its leaf visits that node in this
     * same step, together with the lanes that stood at an inner node anyway (one step saved per leaf visit). */
    if (run_leaves) {
        if (is_leaf) {
            leaf_visit<COUNT>(P, T, cn);
            done = trav_advance<SPILL>(M, T);
            is_inner = !done && (T.cur & PT_REF_LEAF) == 0u;
        }
    }
    if (COUNT && ws) {
        ws->steps++;
        ws->inner_lanes += (unsigned)__popcll(__ballot(is_inner));
        if (run_leaves) { ws->leaf_runs++; ws->leaf_lanes += (unsigned)__popcll(leaf_mask); }
    }
    if (COOP && !WIDE) {
        if (__ballot(is_inner)) {             /* wave-uniform: every lane takes part in the fetch */
            const uint32_t own = quad_fetch_own(M);
            quad_fetch(P, M, own, is_inner ? T.cur * 64u : PT_NO_RECORD_OFFSET);
            if (is_inner) {
                inner_visit<FAST, COUNT, SPILL, true>(P, M, T, cn, own);
                done = trav_advance<SPILL>(M, T);
            }
        }
    } else if (is_inner) {
        if (WIDE) wide_visit<FAST, SPILL>(P, M, T);
        else inner_visit<FAST, COUNT, SPILL>(P, M, T, cn);
        done = trav_advance<SPILL>(M, T);
    }
    return done;
}

/* the sphere loop of calculate_ray_collision, shader.wgsl:144-154 */
__device__ __forceinline__ void spheres_closest(const PTKernelParams &P, V3 o, V3 d, float &best, int32_t &id) {
    best = P.far_distance;
    id = PT_NONE;
    for (uint32_t k = 0; k < P.sphere_count; k++) {
        float dst;
        if (ray_sphere(o, d, P.spheres + 12 * k, dst) && dst < best) { best = dst; id = -(int32_t)(1 + k); }
    }
}

/* calculate_ray_collision, shader.wgsl:139-163, one whole cast for this lane: spheres in
 * index order, then the BVH (which does not see the sphere distance), strict `<`
 * everywhere.  `active` lanes take part; the division form is chosen per wave. */
template <bool COUNT, bool WIDE>
__device__ __forceinline__ void closest_hit(const PTKernelParams &P, const WaveMem &M, bool active, V3 o, V3 d,
                                            float &t_out, int32_t &id_out, Counters &cn) {
    float best = P.far_distance;
    int32_t id = PT_NONE;
    const bool fast = P.fast_div && __all(!active || ray_div_safe(o, d));
    if (active) {
        spheres_closest(P, o, d, best, id);
    }
    Trav T;
    raypk_set(T.R, o, d);
    trav_begin(P, stack_bottom(M), T);
    bool traversing = active;
    while (__ballot(traversing)) {
        bool done = fast ? trav_wave_step<true, COUNT, false, WIDE>(P, M, traversing, T, cn)
                         : trav_wave_step<false, COUNT, false, WIDE>(P, M, traversing, T, cn);
        if (done) traversing = false;
    }
    if (active) {
        if (T.best_id != PT_NONE && T.best_t < best) { best = T.best_t; id = T.best_id; }
        if (COUNT) cn.casts++;
    }
    t_out = best;
    id_out = id;
}

struct Surface { V3 position, normal, color, emission_color; float emission_strength, smoothness; };

/* the HitInfo fields of the winning primitive (shader.wgsl:280-286, 313-320) */
__device__ __forceinline__ Surface surface_of(const PTKernelParams &P, V3 o, V3 d, float t, int32_t id) {
    Surface s;
    s.position = o + d * t;
    if (id >= 0) {
        const PTTriRecord *tr = P.tris + id;
        float4 q2 = tr->q[2];
        s.normal = normalize3(mk(q2.y, q2.z, q2.w));
        s.color = ld3(P.tri_color);
        s.emission_color = ld3(P.tri_emission_color);
        s.emission_strength = P.tri_emission_strength;
        s.smoothness = P.tri_smoothness;
    } else {
        const float *sp = P.spheres + 12 * (uint32_t)(-(id + 1));
        s.normal = normalize3(s.position - ld3(sp));
        s.color = ld3(sp + 4);
        s.emission_color = ld3(sp + 7);
        s.emission_strength = sp[10];
        s.smoothness = sp[11];
    }
    return s;
}

__device__ __forceinline__ float smoothstep_(float e0, float e1, float x) {
    float t = (x - e0) / (e1 - e0);
    t = fmin_(fmax_(t, 0.0f), 1.0f);
    return t * t * (3.0f - 2.0f * t);
}

/* get_environment_light, shader.wgsl:391-404 (neg_sun = -normalize(sun_light_direction)) */
template <bool EXACT>
__device__ __forceinline__ V3 environment_light(const PTKernelParams &P, V3 dir) {
    float sky_gradient_t = t_pow<EXACT>(smoothstep_(0.0f, 0.4f, dir.y), 0.35f);
    V3 sky_gradient = lerp3(ld3(P.sky_horizon), ld3(P.sky_zenith), sky_gradient_t);
    float sun = t_pow<EXACT>(fmax_(0.0f, dot3(dir, ld3(P.neg_sun))), P.sun_focus) * P.sun_intensity;
    float ground_to_sky_t = smoothstep_(-0.01f, 0.0f, dir.y);
    float sun_mask = (ground_to_sky_t >= 1.0f) ? 1.0f : 0.0f;
    V3 r = lerp3(ld3(P.ground), sky_gradient, ground_to_sky_t);
    float ss = sun * sun_mask;
    return mk(r.x + ss, r.y + ss, r.z + ss);
}

/* one "hit" iteration of trace, shader.wgsl:119-127.  `seed` already contains the
 * frame term: seed = pixel_index + sample * 248135 + frame_count * 393939123 (u32 wrap). */
template <bool EXACT>
__device__ __forceinline__ void bounce(const PTKernelParams &P, const Surface &s, uint32_t seed, uint32_t i,
                                       V3 &ro, V3 &rd, V3 &light, V3 &ray_color) {
    ro = s.position;
    uint32_t rs = seed + i * 12345u;
    V3 diffuse_dir = normalize3(s.normal + random_direction<EXACT>(rs));
    float dn2 = 2.0f * dot3(s.normal, rd);
    V3 specular_dir = rd - s.normal * dn2;
    rd = lerp3(diffuse_dir, specular_dir, s.smoothness);
    V3 emitted_light = s.emission_color * s.emission_strength;
    light = light + emitted_light * ray_color;
    ray_color = ray_color * s.color;
}

/* the tail of fs_main, shader.wgsl:102-106, for one frame: `acc` is the pixel's accumulation
 * value, `sum` the sum of the frame's samples in sample order */
__device__ __forceinline__ V3 fold_frame(const PTKernelParams &P, V3 acc, V3 sum, uint32_t frame_count) {
    float n = (float)P.rays_per_pixel;
    V3 pixel_color = mk(sum.x / n, sum.y / n, sum.z / n);
    if (P.accum_mode == 0) {
        float weight = 1.0f / (float)(frame_count + 1u);
        return acc * (1.0f - weight) + pixel_color * weight;
    }
    return acc + pixel_color;             /* frame-sharded renders keep a sum */
}

__device__ __forceinline__ uint32_t fast_div(uint32_t n, PTFastDiv f) {
    uint32_t t = __umulhi(f.mul, n);
    return (t + ((n - t) >> f.sh1)) >> f.sh2;
}

/* Where sample `fs` (= frame of the batch * rays_per_pixel + sample) of hit pixel `hit` lives:
 * 64 hit pixels are interleaved, so that the resolve kernel, one lane per hit pixel, reads
 * consecutive 16-byte values across a wave. */
__device__ __forceinline__ uint32_t sample_slot(uint32_t hit, uint32_t fs, uint32_t per_hit) {
    return ((hit >> 6) * per_hit + fs) * 64u + (hit & 63u);
}

__device__ __forceinline__ uint32_t lane_id() {
    return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}
__device__ __forceinline__ uint32_t prefix_count(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

/* counters[0..3] = casts, box tests, triangle tests, paths as the reference would execute them;
 * counters[4..6] = the share of the first three done by the path kernel */
__device__ __forceinline__ void flush_counters(const PTKernelParams &P, const Counters &cn, unsigned long long scale,
                                               bool path_kernel) {
    unsigned long long c = wave_sum(cn.casts), b = wave_sum(cn.boxes), t = wave_sum(cn.tris), p = wave_sum(cn.paths);
    if (lane_id() == 0) {
        atomicAdd(&P.counters[0], c * scale);
        atomicAdd(&P.counters[1], b * scale);
        atomicAdd(&P.counters[2], t * scale);
        atomicAdd(&P.counters[3], p * scale);
        if (path_kernel) {
            atomicAdd(&P.counters[4], c);
            atomicAdd(&P.counters[5], b);
            atomicAdd(&P.counters[6], t);
        }
    }
}

/* ======================= primary rays ======================= */
template <bool EXACT, bool COUNT, bool AOV, bool WIDE = false>
__global__ __launch_bounds__(PT_BLOCK) void primary_kernel(PTKernelParams P) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    WaveMem M;                       /* the primary kernel keeps the whole stack in LDS */
    M.base = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave) * P.stack_entries * 256u;
    M.limit = M.base + P.stack_entries * 256u; M.spill = nullptr; M.spill_lane = 0; M.spill_stride = 0; M.xchg = 0;
    uint32_t px, py;
    if (AOV || P.shard_world <= 1u) {
        /* block = 16x16 pixels, wave = 8x8 tile (the ids/distance output always covers the whole image) */
        px = blockIdx.x * 16u + (wave & 1u) * 8u + (lane & 7u);
        py = blockIdx.y * 16u + (wave >> 1) * 8u + (lane >> 3);
    } else {
        /* pixel sharding: this rank owns the 8x8 tiles with (tile_x + tile_y) % world == rank.  The grid
         * holds owned tiles only: row of tiles = blockIdx.y, the k-th owned tile of that row per wave. */
        const uint32_t ty = blockIdx.y, k = blockIdx.x * PT_WAVES_PER_BLOCK + wave;
        const uint32_t first = (P.shard_rank + P.shard_world - ty % P.shard_world) % P.shard_world;
        px = (first + k * P.shard_world) * 8u + (lane & 7u);
        py = ty * 8u + (lane >> 3);
    }
    const bool valid = px < P.width && py < P.height;
    Counters cn = {0, 0, 0, 0};

    V3 ro = ld3(P.cam_pos), rd = mk(0.0f, 0.0f, -1.0f);
    uint32_t pixel_index = 0;
    float t = P.far_distance;
    int32_t id = PT_NONE;
    if (valid) primary_ray(P, px, py, rd, pixel_index);
    closest_hit<COUNT, WIDE>(P, M, valid, ro, rd, t, id, cn);
    const uint32_t pixel = py * P.width + px;
    if (AOV) {
        if (valid) { P.out_ids[pixel] = id; P.out_dist[pixel] = t; }
        return;
    }
    const bool hit = valid && id != PT_NONE;
    /* Compact the hit pixels into the global list: ONE atomic per block (every block of the grid makes one on the
     * same word, and atomics on one word are served one after the other: with one per wave they were most of this
     * kernel's time).  The waves' counts meet in the first words of the LDS, which the traversal stacks no longer
     * need. */
    unsigned long long mask = __ballot(hit);
    uint32_t hit_index = PT_PIXEL_MISS;
    __syncthreads();                     /* every wave's traversal is over */
    if (lane == 0) pt_lds[wave] = (uint32_t)__popcll(mask);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t n = 0;
        for (uint32_t w = 0; w < PT_WAVES_PER_BLOCK; w++) n += pt_lds[w];
        pt_lds[PT_WAVES_PER_BLOCK] = n ? atomicAdd(&P.ctrl->hit_count, n) : 0u;
    }
    __syncthreads();
    if (hit) {
        uint32_t base = pt_lds[PT_WAVES_PER_BLOCK];
        for (uint32_t w = 0; w < wave; w++) base += pt_lds[w];
        hit_index = base + prefix_count(mask);
        PTHitRecord r;
        r.pixel = pixel; r.prim_id = id; r.distance = t; r.pad = 0;
        P.hits[hit_index] = r;
    }
    if (valid) {
        P.pixel_hit[pixel] = hit_index;
        if (!hit) {
            /* every sample of every frame is the same environment lookup (trace breaks at its
             * first cast, shader.wgsl:128-132): store the sum over the samples of one frame */
            V3 e = mk(0.0f, 0.0f, 0.0f);
            if (P.use_env) e = environment_light<EXACT>(P, rd) * mk(1.0f, 1.0f, 1.0f);
            V3 incoming = mk(0.0f, 0.0f, 0.0f) + e;
            V3 sum = mk(0.0f, 0.0f, 0.0f);
            for (uint32_t s = 0; s < P.rays_per_pixel; s++) sum = sum + incoming;
            P.pixel_const[pixel] = make_float4(sum.x, sum.y, sum.z, 0.0f);
        }
    }
    if (COUNT) {
        if (valid) cn.paths = 1;
        /* the reference repeats the primary cast for every sample of every frame */
        flush_counters(P, cn, (unsigned long long)P.rays_per_pixel * P.batch_frames, false);
    }
}

/* ======================= bounces ======================= */
/*
 * The persistent path kernel.  A lane owns one path at a time; the
 * BVH traversal of its current ray is resumable (Trav), so a wave does not have
 * to wait for its slowest lane: as soon as P.refill_min lanes have finished
 * their traversal the wave leaves the traversal loop, shades those lanes (which
 * either produces their next ray or ends the path and pulls a new one from the
 * global queue) and re-enters the loop with the other lanes' traversals intact.
 */
/* Six waves per SIMD: told so, the register allocator fits the kernel in 79 VGPRs without scratch
 * (left alone it takes 102 and runs at four).  Measured: 4 waves 3.68 ms per frame, 5 waves 3.51,
 * 6 waves 3.10 (with the other changes made since).  The whole-stack-in-LDS variant reaches six
 * resident blocks per CU for trees of up to 26 levels (6 x 26 KiB), five up to 30. */
#ifndef PT_PATH_WAVES_PER_EU
#define PT_PATH_WAVES_PER_EU 6
#endif
#ifndef PT_STEPS_PER_CHECK
#define PT_STEPS_PER_CHECK 12
#endif
#ifndef PT_LEAF_PERIOD
#define PT_LEAF_PERIOD 3
#endif
/* Wave priority (s_setprio) while a wave walks the tree and while it shades.  A walking wave that has its records is let
 * ahead of waves in the middle of a shading round (a long stretch of arithmetic without a fetch), so its next gathers are
 * issued sooner and the address units idle less: 3 / 0 against 0 / 0 on one box, four runs: cfg 4 +1.0, +0.5, +0.45,
 * +0.55 %, cfg 5 +0.7 %, cfg 3 +0.1 to +0.3 %; 1 / 0 and 3 / 1 the same or less; raising it only while a step's loads are
 * issued gains nothing. */
#ifndef PT_PRIO_WALK
#define PT_PRIO_WALK 3
#endif
#ifndef PT_PRIO_ROUND
#define PT_PRIO_ROUND 0
#endif
#define PT_PATH_OCCUPANCY __attribute__((amdgpu_waves_per_eu(PT_PATH_WAVES_PER_EU, PT_PATH_WAVES_PER_EU)))
template <bool EXACT, bool COUNT, bool SPILL, bool WIDE = false, bool COOP = false>
__global__ __launch_bounds__(PT_PATH_BLOCK) PT_PATH_OCCUPANCY void path_kernel(PTKernelParams P, uint32_t lds_entries) {
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    WaveMem M;                       /* lds_entries stack entries per lane in LDS, deeper ones in global memory */
    M.xchg = PT_PATH_WAVES_PER_BLOCK * lds_entries * 256u + wave * PT_XCHG_BYTES;     /* exchange areas follow the stacks */
    M.base = wave * lds_entries * 256u;
    M.limit = M.base + lds_entries * 256u;
    M.spill = SPILL ? P.spill : nullptr;
    M.spill_lane = blockIdx.x * PT_PATH_BLOCK + threadIdx.x;
    M.spill_stride = P.spill_stride;
    const uint32_t rpp = P.rays_per_pixel;
    const uint32_t per_hit = P.batch_frames * rpp;
    const uint32_t total = P.ctrl->hit_count * per_hit;
    /* The queue is cut into PT_QUEUE_PARTS equal, contiguous parts of whole pixels, each with a head of its own;
     * PT_QUEUE_PARTS / 8 consecutive parts belong to one XCD (consecutive paths belong to neighbouring pixels, so an
     * XCD's parts are a compact piece of the image and the BVH nodes its rays visit stay in that XCD's 4 MiB L2). */
    const uint32_t part = ((P.ctrl->hit_count + PT_QUEUE_PARTS - 1u) / PT_QUEUE_PARTS) * per_hit;
    /* XCC_ID: bits 3:0 of hardware register 20 */
    const uint32_t xcd = P.xcd_queues ? (__builtin_amdgcn_s_getreg(20 | (0 << 6) | ((4 - 1) << 11)) & 7u) : (blockIdx.x & 7u);
    const uint32_t wave_in_xcd = (blockIdx.x >> 3) * PT_PATH_WAVES_PER_BLOCK + wave;      /* blocks go round the XCDs */
    uint32_t queue = xcd * (PT_QUEUE_PARTS / 8u) + wave_in_xcd % (PT_QUEUE_PARTS / 8u);
    bool part_done = false;      /* the part in hand has no paths left */
    Counters cn = {0, 0, 0, 0};

    /* the lane's phase, one flag each (the compiler keeps them as lane masks in scalar registers) */
    bool idle = true;        /* no path: wants one from the queue */
    bool has_ray = false;    /* has a ray that has not started its cast */
    bool trav = false;       /* BVH traversal in progress */
    bool hit = false;        /* cast complete, waiting to be shaded */
    bool dead = false;       /* queue drained and path finished */
    bool drained = total == 0;
    uint32_t slot = 0, seed = 0, i = 0;
    V3 light = mk(0, 0, 0), ray_color = mk(1, 1, 1);
    float sph_t = P.far_distance;
    int32_t sph_id = PT_NONE;
    Trav T;      /* T.R is the lane's current ray: the only copy kept across the traversal loop */
    T.cur = PT_REF_NONE; T.sp = 0; T.best_t = P.far_distance; T.best_id = PT_NONE;
    raypk_set(T.R, mk(0, 0, 0), mk(0, 0, -1));
    bool safe = false;
    WaveStats ws = {0, 0, 0, 0};
    unsigned long long t_start = 0, t_trav = 0, outer_iters = 0, shade_lanes = 0, t_drained = 0;
    unsigned long long rt_start = 0, rt_drained = 0;      /* constant-rate device clock (100 MHz), comparable across waves */
    if (COUNT) { t_start = __builtin_amdgcn_s_memtime(); rt_start = wall_clock64(); }

    for (;;) {
        __builtin_amdgcn_s_setprio(PT_PRIO_ROUND);
        if (COUNT) { outer_iters++; shade_lanes += (unsigned)__popcll(__ballot(hit)); }
        /* (A) lanes whose cast is complete pick their closest hit; a miss ends the path with the
         * sky term (shader.wgsl:128-132) */
        float hit_t = P.far_distance;
        int32_t hit_id = PT_NONE;
        if (hit) {
            hit_t = sph_t; hit_id = sph_id;
            if (T.best_id != PT_NONE && T.best_t < hit_t) { hit_t = T.best_t; hit_id = T.best_id; }
            if (hit_id == PT_NONE) {
                if (P.use_env) light = light + environment_light<EXACT>(P, ray_dir(T.R)) * ray_color;
                P.samples[slot] = make_float4(light.x, light.y, light.z, 0.0f);
                hit = false;
                idle = true;
            }
        }
        /* (B) idle lanes pull new paths: one atomic per wave and round, on the head of the part the wave works on
         * (a part of its own XCD while one has paths left, then it helps the others).  The paths a round takes are
         * consecutive samples of one pixel or of two neighbours: they start their first cast in step.  (Runs of
         * 64 to 1024 paths per atomic, handed out over several rounds, were measured too: a round then often mixes
         * the end of one run with the start of another, far away in the image, and 35 to 37.5 lanes instead of 39 do
         * an inner visit per step.)  A new path starts as "cast complete" at its primary hit, iteration i = 0 of
         * trace (shader.wgsl:117-127), and is shaded below together with the others. */
        unsigned long long need = __ballot(idle);
        if (need) {
            while (need && !drained) {
                if (part_done) {
                    /* The part in hand is used up.  One coherent load shows where every head stands: the wave
                     * goes on with a part of its own XCD that has paths left, else with any such part (the waves
                     * spread over the candidates by their number).  The picture may be stale: a part found empty
                     * after all sends the wave back here. */
                    part_done = false;
                    bool has = false;
                    if (lane_id() < PT_QUEUE_PARTS) {
                        const uint32_t ps = lane_id() * part < total ? lane_id() * part : total;
                        const uint32_t pl = (total - ps < part) ? total - ps : part;
                        has = __hip_atomic_load(&P.ctrl->queue_head[lane_id() * PT_HEAD_STRIDE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < pl;
                    }
                    const unsigned long long any = __ballot(has);
                    const unsigned long long own = any & (((1ull << (PT_QUEUE_PARTS / 8u)) - 1ull) << (xcd * (PT_QUEUE_PARTS / 8u)));
                    unsigned long long pick = own ? own : any;
                    if (!pick) { drained = true; break; }
                    for (uint32_t skip = wave_in_xcd % (uint32_t)__popcll(pick); skip; skip--) pick &= pick - 1ull;
                    queue = (uint32_t)__builtin_ctzll(pick);
                }
                /* as many consecutive paths as the wave has idle lanes, from the head of the part: one atomic */
                const uint32_t q_start = queue * part < total ? queue * part : total;
                const uint32_t q_len = (total - q_start < part) ? total - q_start : part;
                const uint32_t n_need = (uint32_t)__popcll(need);
                uint32_t base = 0;
                if (lane_id() == 0) base = atomicAdd(&P.ctrl->queue_head[queue * PT_HEAD_STRIDE], n_need);
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                part_done = base + n_need >= q_len;
                const uint32_t local = base + prefix_count(need);
                if (idle && local < q_len) {
                    const uint32_t item = q_start + local;
                    /* queue index -> (hit pixel, frame of the batch, sample) */
                    uint32_t hf = fast_div(item, P.div_rpp);
                    uint32_t s = item - hf * rpp;
                    uint32_t hi = fast_div(hf, P.div_batch);
                    slot = sample_slot(hi, item - hi * per_hit, per_hit);
                    uint32_t frame_count = P.frame_first + (hf - hi * P.batch_frames) * P.frame_stride;
                    PTHitRecord h = P.hits[hi];
                    uint32_t py = fast_div(h.pixel, P.div_width);
                    uint32_t px = h.pixel - py * P.width;
                    uint32_t pixel_index;
                    V3 rd;
                    primary_ray(P, px, py, rd, pixel_index);
                    raypk_set_od(T.R, ld3(P.cam_pos), rd);
                    /* shader.wgsl:100 and :121, folded: the frame term is the same for every bounce */
                    seed = pixel_index + s * 248135u + frame_count * 393939123u;
                    light = mk(0.0f, 0.0f, 0.0f);
                    ray_color = mk(1.0f, 1.0f, 1.0f);
                    i = 0;
                    hit_t = h.distance; hit_id = h.prim_id;
                    idle = false;
                    hit = true;
                }
                need = __ballot(idle);          /* lanes the part had nothing left for: served from another part */
            }
            if (drained && idle) { idle = false; dead = true; }
            if (COUNT && drained && !t_drained) { t_drained = __builtin_amdgcn_s_memtime(); rt_drained = wall_clock64(); }
        }
        /* (A, continued) one "hit" iteration of trace for old and new paths alike, shader.wgsl:119-127 */
        if (hit) {
            V3 ro = ray_origin(T.R), rd = ray_dir(T.R);
            Surface sf = surface_of(P, ro, rd, hit_t, hit_id);
            bounce<EXACT>(P, sf, seed, i, ro, rd, light, ray_color);
            i++;
            hit = false;
            if (i > P.max_bounce_count) {                 /* loop bound reached: no sky term */
                P.samples[slot] = make_float4(light.x, light.y, light.z, 0.0f);
                idle = true;                              /* refilled in the next round */
            } else {
                raypk_set(T.R, ro, rd);
                has_ray = true;
            }
        }
        /* (C) new rays: spheres first (shader.wgsl:144-154), then set up the traversal */
        if (has_ray) {
            V3 ro = ray_origin(T.R), rd = ray_dir(T.R);
            spheres_closest(P, ro, rd, sph_t, sph_id);
            trav_begin(P, stack_bottom(M), T);
            safe = ray_div_safe(ro, rd);
            if (COUNT) cn.casts++;
            has_ray = false;
            trav = true;
        }
        if (!__ballot(!dead)) break;
        /* (D) traverse until enough lanes wait for shading (or, at the tail, all are done) */
        const bool fast = P.fast_div && __all(!trav || safe);
        unsigned long long t0 = 0;
        if (COUNT) t0 = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_s_setprio(PT_PRIO_WALK);
        for (;;) {
            const unsigned long long trav_mask = __ballot(trav);
            if (!trav_mask) break;
            const uint32_t n_wait = (uint32_t)__popcll(__ballot(hit));
            if (n_wait >= P.refill_min) break;
            /* once the queue is empty the wave never reaches refill_min again: let the waiting lanes go
             * on as soon as they are as many as the lanes still traversing */
            if (drained && n_wait >= (uint32_t)__popcll(trav_mask)) break;
#pragma unroll
            /* Several (twelve) steps per pass: the two tests above cost a fifth of a step.  Every PT_LEAF_PERIOD-th
             * step also runs the triangle code for the lanes that stand at a leaf (they wait for it): measured
             * faster than deciding from lane counts in every step (2.98 -> 2.90 ms per frame). */
            for (int rep = 0; rep < PT_STEPS_PER_CHECK; rep++) {
                const int turn = (rep % PT_LEAF_PERIOD) == PT_LEAF_PERIOD - 1 ? 1 : 0;
                bool done = fast ? trav_wave_step<true, COUNT, SPILL, WIDE, COOP>(P, M, trav, T, cn, &ws, turn)
                                 : trav_wave_step<false, COUNT, SPILL, WIDE, COOP>(P, M, trav, T, cn, &ws, turn);
                if (done) { trav = false; hit = true; }
            }
        }
        if (COUNT) t_trav += __builtin_amdgcn_s_memtime() - t0;
    }
    if (COUNT) {
        flush_counters(P, cn, 1ull, true);
        /* diagnostics: where a wave's time and lanes go (pt_get_debug_stats) */
        if (lane_id() == 0) {
            atomicAdd(&P.counters[8], __builtin_amdgcn_s_memtime() - t_start);
            atomicAdd(&P.counters[9], t_trav);
            atomicAdd(&P.counters[10], ws.steps);
            atomicAdd(&P.counters[11], ws.inner_lanes);
            atomicAdd(&P.counters[12], ws.leaf_lanes);
            atomicAdd(&P.counters[13], ws.leaf_runs);
            atomicAdd(&P.counters[14], outer_iters);
            atomicAdd(&P.counters[15], shade_lanes);
            /* the tail: time this wave went on after it found the queue empty (timestamps of
             * different XCDs are not comparable, so only differences within a wave are kept) */
            const unsigned long long t_end = __builtin_amdgcn_s_memtime();
            if (!t_drained) t_drained = t_end;
            atomicMax(&P.counters[16], t_end - t_start);
            atomicMax(&P.counters[17], t_end - t_drained);
            atomicAdd(&P.counters[20], t_end - t_drained);
            /* wall-clock timeline of the launch: earliest wave start, earliest "queue empty", latest end
             * (minima kept as maxima of 2^62 - t) */
            const unsigned long long rt_end = wall_clock64();
            if (!rt_drained) rt_drained = rt_end;
            atomicMax(&P.counters[21], (1ull << 62) - rt_start);
            atomicMax(&P.counters[22], (1ull << 62) - rt_drained);
            atomicMax(&P.counters[23], rt_end);
        }
    }
}

/* ======================= bounces, two paths per lane ======================= */
/*
 * The same work as path_kernel under another schedule.  A lane owns TWO paths.  Their state (ray origin and
 * direction, random seed, radiance so far, the sphere hit of the current ray, the result of the last cast) lives in
 * a per-lane buffer pair in global memory; one path at a time is "walked": its ray sits in the lane's registers and
 * runs through the tree.  Every PT_MB_SERVICE steps a lane whose cast is complete writes the hit back (8 bytes; that
 * buffer now waits for shading) and takes up its other path if that one has a ray ready - the ready ray of the
 * parked path waits in registers too (nine values), because a load from the buffers at every service point stalls
 * the wave for a trip to the Infinity Cache.  The wave shades when P.refill_min lanes have a waiting buffer: one
 * buffer per lane and round, through the same code as path_kernel's round (closest hit, sky or bounce, new path
 * from the queue, sphere pass of the new ray).  So a round still runs with most lanes busy, but a lane does not sit
 * idle from the end of its cast to the next round: 53 of 64 lanes do an inner-node visit per step instead of 41
 * (tools/sim/block_sim.c, sim_multibuf, on recorded paths of the benchmark scene).
 *
 * Buffer of one path: five 16-byte quads, quad q of buffer b of lane l of wave w at
 * pathbuf[((w * 2 + b) * 5 + q) * 64 + l], so that a wave's accesses are whole 1-KiB rows:
 *   q0 = o.x o.y o.z i     q1 = d.x d.y d.z seed      q2 = light.xyz slot
 *   q3 = ray_color.xyz sph_t                          q4 = sph_id best_t best_id -
 */
#ifndef PT_MB_SERVICE
#define PT_MB_SERVICE 3
#endif
#define PT_MB_QUADS 5

template <bool EXACT, bool COUNT, bool SPILL, bool COOP>
__global__ __launch_bounds__(PT_PATH_BLOCK) PT_PATH_OCCUPANCY void path_kernel_mb(PTKernelParams P, uint32_t lds_entries) {
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    WaveMem M;
    M.xchg = PT_PATH_WAVES_PER_BLOCK * lds_entries * 256u + wave * PT_XCHG_BYTES;
    M.base = wave * lds_entries * 256u;
    M.limit = M.base + lds_entries * 256u;
    M.spill = SPILL ? P.spill : nullptr;
    M.spill_lane = blockIdx.x * PT_PATH_BLOCK + threadIdx.x;
    M.spill_stride = P.spill_stride;
    const uint32_t rpp = P.rays_per_pixel;
    const uint32_t per_hit = P.batch_frames * rpp;
    const uint32_t total = P.ctrl->hit_count * per_hit;
    const uint32_t part = ((P.ctrl->hit_count + 7u) / 8u) * per_hit;     /* whole pixels per part */
    uint32_t queue = P.xcd_queues ? (__builtin_amdgcn_s_getreg(20 | (0 << 6) | ((4 - 1) << 11)) & 7u) : (blockIdx.x & 7u);
    uint32_t queues_done = 0;
    Counters cn = {0, 0, 0, 0};
    /* this lane's two buffers: quad q of buffer b at byte mb_off + (b * 6 + q) * 1024 of pathbuf (buffer
     * instructions: descriptor in scalar registers, one 32-bit offset per lane) */
    const __amdgpu_buffer_rsrc_t mb = __builtin_amdgcn_make_buffer_rsrc((void *)P.pathbuf, 0, (int)P.pathbuf_bytes, 0x00020000);
    const uint32_t mb_off = (blockIdx.x * PT_PATH_WAVES_PER_BLOCK + wave) * (2u * PT_MB_QUADS * 1024u) + lane_id() * 16u;
#define MB_LOAD(off, q) u4_as_f4(__builtin_amdgcn_raw_buffer_load_b128(mb, (off), (q) * 1024, 0))
#define MB_STORE(off, q, v) __builtin_amdgcn_raw_buffer_store_b128(f4_as_u4(v), mb, (off), (q) * 1024, 0)

    bool drained = total == 0;
    bool walking = false;       /* a cast is in progress in the lane's registers */
    bool finished = false;      /* ... and is complete: to be written back at the next service point */
    bool act1 = false;          /* which buffer is being walked */
    /* per buffer: it holds a ray that has not been cast / it waits for a shading round / it holds no path yet */
    bool ready0 = false, ready1 = false, wait0 = true, wait1 = true, fresh0 = true, fresh1 = true;
    V3 park_o = mk(0, 0, 0), park_d = mk(0, 0, -1), park_y = mk(0, 0, -1);     /* the ready ray of the path that is not walked */
    bool park_safe = false;
    Trav T;
    T.cur = PT_REF_NONE; T.sp = 0; T.best_t = P.far_distance; T.best_id = PT_NONE;
    raypk_set(T.R, mk(0, 0, 0), mk(0, 0, -1));
    bool safe = false;
    WaveStats ws = {0, 0, 0, 0};
    unsigned long long t_start = 0, t_trav = 0, outer_iters = 0, shade_lanes = 0, t_drained = 0;
    unsigned long long rt_start = 0, rt_drained = 0;
    if (COUNT) { t_start = __builtin_amdgcn_s_memtime(); rt_start = wall_clock64(); }

    for (;;) {
        /* ---- service point: write finished casts back, take up ready rays ---- */
        if (__ballot(finished)) {
            if (finished) {
                u2_t r; r.x = __float_as_uint(T.best_t); r.y = (uint32_t)T.best_id;
                __builtin_amdgcn_raw_buffer_store_b64(r, mb, mb_off + (act1 ? PT_MB_QUADS * 1024u : 0u), 4 * 1024 + 4, 0);
                if (act1) wait1 = true; else wait0 = true;
                finished = false;
            }
        }
        const bool pick = !walking && (ready0 || ready1);
        if (__ballot(pick)) {
            if (pick) {
                act1 = !ready0;
                T.R.oxy = mk2(park_o.x, park_o.y); T.R.ozz = mk2(park_o.z, park_o.z);
                T.R.dxy = mk2(park_d.x, park_d.y); T.R.dzz = mk2(park_d.z, park_d.z);
                T.R.yxy = mk2(park_y.x, park_y.y); T.R.yzz = mk2(park_y.z, park_y.z);
                raypk_low(T.R);
                safe = park_safe;
                trav_begin(P, stack_bottom(M), T);
                if (act1) ready1 = false; else ready0 = false;
                walking = true;
                if (COUNT) cn.casts++;
            }
        }
        /* ---- shade? ---- */
        const bool waiting = wait0 || wait1;
        const uint32_t n_wait = (uint32_t)__popcll(__ballot(waiting)), n_walk = (uint32_t)__popcll(__ballot(walking));
        if (!n_wait && !n_walk) break;           /* every buffer of every lane is empty (ready rays were just taken up) */
        const uint32_t n_starve = (uint32_t)__popcll(__ballot(waiting && !walking));
        if (n_wait && (n_wait >= P.refill_min || n_starve >= P.starve_min || !n_walk || (drained && n_wait >= n_walk))) {
            if (COUNT) { outer_iters++; shade_lanes += n_wait; }
            const bool sb1 = !wait0;             /* the buffer this lane shades in this round */
            const uint32_t b = mb_off + (sb1 ? PT_MB_QUADS * 1024u : 0u);
            bool hit = false, idle = false, has_ray = false, emptied = false;
            V3 ro = mk(0, 0, 0), rd = mk(0, 0, -1), light = mk(0, 0, 0), ray_color = mk(1, 1, 1);
            uint32_t seed = 0, i = 0, slot = 0;
            float hit_t = P.far_distance;
            int32_t hit_id = PT_NONE;
            if (waiting) {
                if (sb1 ? fresh1 : fresh0) idle = true;
                else {
                    const float4 q0 = MB_LOAD(b, 0), q1 = MB_LOAD(b, 1), q3 = MB_LOAD(b, 2), q4 = MB_LOAD(b, 3), q5 = MB_LOAD(b, 4);
                    ro = mk(q0.x, q0.y, q0.z); rd = mk(q1.x, q1.y, q1.z);
                    seed = __float_as_uint(q1.w); i = __float_as_uint(q0.w); slot = __float_as_uint(q3.w);
                    light = mk(q3.x, q3.y, q3.z); ray_color = mk(q4.x, q4.y, q4.z);
                    /* (A) the closest hit of the finished cast: spheres win ties (shader.wgsl:144-160); a miss ends
                     * the path with the sky term (shader.wgsl:128-132) */
                    hit_t = q4.w; hit_id = __float_as_int(q5.x);
                    const float best_t = q5.y;
                    const int32_t best_id = __float_as_int(q5.z);
                    if (best_id != PT_NONE && best_t < hit_t) { hit_t = best_t; hit_id = best_id; }
                    if (hit_id == PT_NONE) {
                        if (P.use_env) light = light + environment_light<EXACT>(P, rd) * ray_color;
                        P.samples[slot] = make_float4(light.x, light.y, light.z, 0.0f);
                        idle = true;
                    } else hit = true;
                }
            }
            /* (B) buffers without a path take one from the queue (see path_kernel) */
            unsigned long long need = __ballot(idle);
            if (need) {
                while (!drained) {
                    const uint32_t q_start = queue * part < total ? queue * part : total;
                    const uint32_t q_len = (total - q_start < part) ? total - q_start : part;
                    uint32_t base = 0;
                    if (lane_id() == 0) base = atomicAdd(&P.ctrl->queue_head[queue * PT_HEAD_STRIDE], (uint32_t)__popcll(need));
                    base = __shfl(base, 0, 64);
                    const bool exhausted = base + (uint32_t)__popcll(need) >= q_len;
                    if (base < q_len && idle) {
                        const uint32_t local = base + prefix_count(need);
                        if (local < q_len) {
                            const uint32_t item = q_start + local;
                            uint32_t hf = fast_div(item, P.div_rpp);
                            uint32_t s = item - hf * rpp;
                            uint32_t hi = fast_div(hf, P.div_batch);
                            slot = sample_slot(hi, item - hi * per_hit, per_hit);
                            uint32_t frame_count = P.frame_first + (hf - hi * P.batch_frames) * P.frame_stride;
                            PTHitRecord h = P.hits[hi];
                            uint32_t py = fast_div(h.pixel, P.div_width);
                            uint32_t px = h.pixel - py * P.width;
                            uint32_t pixel_index;
                            primary_ray(P, px, py, rd, pixel_index);
                            ro = ld3(P.cam_pos);
                            seed = pixel_index + s * 248135u + frame_count * 393939123u;
                            light = mk(0.0f, 0.0f, 0.0f);
                            ray_color = mk(1.0f, 1.0f, 1.0f);
                            i = 0;
                            hit_t = h.distance; hit_id = h.prim_id;
                            idle = false;
                            hit = true;
                        }
                    }
                    if (exhausted) {
                        queue = (queue + 1u) & 7u;
                        if (++queues_done == 8u) drained = true;
                    }
                    if (base < q_len) break;
                }
                if (drained && idle) { idle = false; emptied = true; }
                if (COUNT && drained && !t_drained) { t_drained = __builtin_amdgcn_s_memtime(); rt_drained = wall_clock64(); }
            }
            /* one "hit" iteration of trace for old and new paths alike, shader.wgsl:119-127 */
            if (hit) {
                Surface sf = surface_of(P, ro, rd, hit_t, hit_id);
                bounce<EXACT>(P, sf, seed, i, ro, rd, light, ray_color);
                i++;
                if (i > P.max_bounce_count) {                 /* loop bound reached: no sky term */
                    P.samples[slot] = make_float4(light.x, light.y, light.z, 0.0f);
                    idle = true;                              /* takes a new path in the next round */
                } else has_ray = true;
            }
            /* the new ray: spheres first (shader.wgsl:144-154), reciprocals for the box test */
            if (has_ray) {
                float sph_t; int32_t sph_id;
                spheres_closest(P, ro, rd, sph_t, sph_id);
                park_safe = ray_div_safe(ro, rd);
                park_o = ro; park_d = rd;
                park_y = mk(1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z);
                MB_STORE(b, 0, make_float4(ro.x, ro.y, ro.z, __uint_as_float(i)));
                MB_STORE(b, 1, make_float4(rd.x, rd.y, rd.z, __uint_as_float(seed)));
                MB_STORE(b, 2, make_float4(light.x, light.y, light.z, __uint_as_float(slot)));
                MB_STORE(b, 3, make_float4(ray_color.x, ray_color.y, ray_color.z, sph_t));
                __builtin_amdgcn_raw_buffer_store_b32((uint32_t)sph_id, mb, b, 4 * 1024, 0);
            }
            if (waiting) {
                /* has_ray: ready.  idle: no path (stays waiting, takes one in the next round).  emptied: queue dry. */
                if (sb1) { ready1 = has_ray; wait1 = idle; fresh1 = !has_ray; }
                else     { ready0 = has_ray; wait0 = idle; fresh0 = !has_ray; }
            }
            (void)emptied;
            continue;
        }
        /* ---- traverse ---- */
        const bool fast = P.fast_div && __all(!walking || safe);
        unsigned long long t0 = 0;
        if (COUNT) t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int rep = 0; rep < PT_MB_SERVICE; rep++) {
            const int turn = (rep % PT_LEAF_PERIOD) == PT_LEAF_PERIOD - 1 ? 1 : 0;
            bool done = fast ? trav_wave_step<true, COUNT, SPILL, false, COOP>(P, M, walking, T, cn, &ws, turn)
                             : trav_wave_step<false, COUNT, SPILL, false, COOP>(P, M, walking, T, cn, &ws, turn);
            if (done) { walking = false; finished = true; }
        }
        if (COUNT) t_trav += __builtin_amdgcn_s_memtime() - t0;
    }
    if (COUNT) {
        flush_counters(P, cn, 1ull, true);
        if (lane_id() == 0) {
            atomicAdd(&P.counters[8], __builtin_amdgcn_s_memtime() - t_start);
            atomicAdd(&P.counters[9], t_trav);
            atomicAdd(&P.counters[10], ws.steps);
            atomicAdd(&P.counters[11], ws.inner_lanes);
            atomicAdd(&P.counters[12], ws.leaf_lanes);
            atomicAdd(&P.counters[13], ws.leaf_runs);
            atomicAdd(&P.counters[14], outer_iters);
            atomicAdd(&P.counters[15], shade_lanes);
            const unsigned long long t_end = __builtin_amdgcn_s_memtime();
            if (!t_drained) t_drained = t_end;
            atomicMax(&P.counters[16], t_end - t_start);
            atomicMax(&P.counters[17], t_end - t_drained);
            atomicAdd(&P.counters[20], t_end - t_drained);
            const unsigned long long rt_end = wall_clock64();
            if (!rt_drained) rt_drained = rt_end;
            atomicMax(&P.counters[21], (1ull << 62) - rt_start);
            atomicMax(&P.counters[22], (1ull << 62) - rt_drained);
            atomicMax(&P.counters[23], rt_end);
        }
    }
#undef MB_LOAD
#undef MB_STORE
}

/* ======================= per-pixel resolve ======================= */
/* Folds the frames of the batch into the accumulation buffer in frame order (shader.wgsl:99-106):
 * per frame the samples are added in sample order, divided by rays_per_pixel and blended with
 * weight 1/(frame_count+1).  One read-modify-write of a pixel per batch.  Thread t finishes hit
 * pixel number t of the hit list (its samples are interleaved with those of the 63 other hit
 * pixels of its wave, see sample_slot) and, independently, pixel t of the image if that is a miss
 * pixel; every pixel is written by exactly one thread.
 * The kernel may share the GPU with the path kernel of the NEXT batch (enqueued on another stream), whose waves
 * would leave it one issue slot in seven: it raises its own priority, because the batch after that one waits for it.
 * It also leaves the queue heads at zero for the next path kernel that uses these control words.  Beside a FULL grid of
 * the path kernel (six waves of 80 registers per SIMD) no wave of any other kernel becomes resident, not even one of 8
 * registers (tools/corun_probe.py): this kernel then runs when that grid begins to drain, which is when the batch that waits
 * for it can start anyway.  Beside the smaller grids that short batches get it runs at once. */
/* A pixel's folded value goes to the accumulation buffer and, for the last batch of a call that has a second output
 * (pt_next_render_output), there as well - or there INSTEAD, the accumulation buffer being left at zero for the next sum. */
__device__ __forceinline__ void resolve_store(const PTKernelParams &P, uint32_t pixel, V3 acc) {
    const float4 v = make_float4(acc.x, acc.y, acc.z, 0.0f);
    if (P.mirror) {
        P.mirror[pixel] = v;
        P.accum[pixel] = P.mirror_move ? make_float4(0.0f, 0.0f, 0.0f, 0.0f) : v;
    } else {
        P.accum[pixel] = v;
    }
}

__global__ __launch_bounds__(256) void resolve_kernel(PTKernelParams P) {
    __builtin_amdgcn_s_setprio(3);
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t rpp = P.rays_per_pixel, per_hit = P.batch_frames * rpp;
    if (t < PT_QUEUE_PARTS) P.ctrl->queue_head[t * PT_HEAD_STRIDE] = 0;
    if (t < P.ctrl->hit_count) {
        const uint32_t pixel = P.hits[t].pixel;
        float4 a4 = P.accum[pixel];
        V3 acc = mk(a4.x, a4.y, a4.z);
        const float4 *s = P.samples + (size_t)sample_slot(t, 0u, per_hit);
        for (uint32_t b = 0; b < P.batch_frames; b++) {
            V3 sum = mk(0.0f, 0.0f, 0.0f);
#pragma unroll 2
            for (uint32_t k = 0; k < rpp; k++) {          /* shader.wgsl:99-101, in sample order */
                float4 v = s[(size_t)(b * rpp + k) * 64u];
                sum = sum + mk(v.x, v.y, v.z);
            }
            acc = fold_frame(P, acc, sum, P.frame_first + b * P.frame_stride);
        }
        resolve_store(P, pixel, acc);
    }
    if (t < P.width * P.height && P.pixel_hit[t] == PT_PIXEL_MISS) {
        float4 a4 = P.accum[t];
        V3 acc = mk(a4.x, a4.y, a4.z);
        float4 c = P.pixel_const[t];
        for (uint32_t b = 0; b < P.batch_frames; b++)
            acc = fold_frame(P, acc, mk(c.x, c.y, c.z), P.frame_first + b * P.frame_stride);
        resolve_store(P, t, acc);
    }
    /* pixels of other ranks' shards: nothing to fold, but a second output must not keep what an earlier use left there */
    if (P.mirror && t < P.width * P.height && P.pixel_hit[t] == PT_PIXEL_NOT_MINE) P.mirror[t] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}

__global__ __launch_bounds__(256) void scale_kernel(float4 *accum, uint32_t n, float s) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 v = accum[i];
    accum[i] = make_float4(v.x * s, v.y * s, v.z * s, v.w);
}

/* ======================= function probe ======================= */
/* Evaluates the small shader functions on the device for given inputs, so that tests can compare
 * them one by one with the CPU restatement.  out[i] = 16 floats: [0] random_value(seed_i),
 * [1] random_normal, [2..4] random_direction, [5..7] random_hemisphere_direction(normal, seed_i),
 * [8..10] get_environment_light(dir_i), [11..13] rotate_vector(dir_i) with the context's camera
 * rotation, [14..15] zero. */
template <bool EXACT>
__global__ __launch_bounds__(256) void probe_kernel(PTKernelParams P, const uint32_t *seeds, const float *dirs,
                                                    float nx, float ny, float nz, uint32_t n, float *out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t seed = seeds[i];
    const V3 dir = ld3(dirs + 3 * i);
    float *o = out + 16 * (size_t)i;
    o[0] = random_value(seed);
    o[1] = random_normal<EXACT>(seed);
    V3 rd = random_direction<EXACT>(seed);
    o[2] = rd.x; o[3] = rd.y; o[4] = rd.z;
    V3 rh = random_hemisphere_direction<EXACT>(mk(nx, ny, nz), seed);
    o[5] = rh.x; o[6] = rh.y; o[7] = rh.z;
    V3 e = environment_light<EXACT>(P, dir);
    o[8] = e.x; o[9] = e.y; o[10] = e.z;
    V3 rv = rotate_vector(P, dir);
    o[11] = rv.x; o[12] = rv.y; o[13] = rv.z;
    o[14] = 0.0f; o[15] = 0.0f;
}

/* Intersection functions for given inputs, one thread per input: ray (origin, dir), box (min, max),
 * triangle (a, b, c), sphere (centre, radius).  out[i] = 8 floats: [0] ray_box with the plain
 * divider, [1] ray_box with the reciprocal form (NaN when ray or box fail its predicates),
 * [2] ray_triangle did_hit (0/1), [3] its distance, [4] ray_sphere did_hit, [5] its distance
 * (distances are meaningful where did_hit is set), [6..7] zero. */
__global__ __launch_bounds__(256) void probe_intersections_kernel(const float *rays, const float *boxes, const float *tris,
                                                                  const float *spheres, uint32_t n, float *out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const V3 o = ld3(rays + 6 * i), d = ld3(rays + 6 * i + 3);
    const float *b = boxes + 6 * i;
    float *r = out + 8 * (size_t)i;
    RayPk R;
    raypk_set(R, o, d);
    const float4 q0 = make_float4(b[0], b[1], b[3], b[4]), q2 = make_float4(b[2], b[5], b[2], b[5]);
    float da, db;
    ray_box2<false>(R, q0, q0, q2, da, db);
    r[0] = da;
    bool tame = ray_div_safe(o, d);
    for (int k = 0; k < 6; k++) tame = tame && pt_div_tame_coordinate(b[k]);
    float fa = __builtin_nanf(""), fb;
    if (tame) ray_box2<true>(R, q0, q0, q2, fa, fb);
    r[1] = fa;
    /* the triangle record as the host builds it (build_tri_records): same f32 operations */
    const float *t = tris + 9 * i;
    PTTriRecord rec;
    const float abx = t[3] - t[0], aby = t[4] - t[1], abz = t[5] - t[2];
    const float acx = t[6] - t[0], acy = t[7] - t[1], acz = t[8] - t[2];
    rec.q[0] = make_float4(t[0], t[1], t[2], abx);
    rec.q[1] = make_float4(aby, abz, acx, acy);
    rec.q[2] = make_float4(acz, aby * acz - abz * acy, abz * acx - abx * acz, abx * acy - aby * acx);
    float dst = 0.0f;
    const bool th = ray_triangle(o, d, &rec, dst);
    r[2] = th ? 1.0f : 0.0f; r[3] = dst;
    float sd = 0.0f;
    const bool sh = ray_sphere(o, d, spheres + 4 * i, sd);
    r[4] = sh ? 1.0f : 0.0f; r[5] = sd;
    r[6] = 0.0f; r[7] = 0.0f;
}

hipError_t pt_launch_probe_intersections(const float *rays, const float *boxes, const float *tris, const float *spheres,
                                         uint32_t n, float *out, hipStream_t st) {
    hipLaunchKernelGGL(probe_intersections_kernel, dim3((n + 255) / 256), dim3(256), 0, st, rays, boxes, tris, spheres, n, out);
    return hipGetLastError();
}

hipError_t pt_launch_probe(const PTKernelParams &kp, bool exact, const uint32_t *seeds, const float *dirs,
                           const float normal[3], uint32_t n, float *out, hipStream_t st) {
    dim3 grid((n + 255) / 256), block(256);
    if (exact) hipLaunchKernelGGL(probe_kernel<true>, grid, block, 0, st, kp, seeds, dirs, normal[0], normal[1], normal[2], n, out);
    else       hipLaunchKernelGGL(probe_kernel<false>, grid, block, 0, st, kp, seeds, dirs, normal[0], normal[1], normal[2], n, out);
    return hipGetLastError();
}

/* ======================= launchers ======================= */
int pt_max_stack_entries(void) { return 64; }       /* 64 entries x 4 waves x 256 B = 64 KiB, the default dynamic-LDS limit */
/* the persistent grid never exceeds 8 blocks per CU */
uint32_t pt_lane_stride(int num_cus) { return (uint32_t)num_cus * 32u * 64u; }      /* at most 32 waves per CU */
/* two buffers of PT_MB_QUADS quads per lane of the largest grid pt_launch_paths uses */
size_t pt_pathbuf_bytes(int num_cus) { return (size_t)pt_lane_stride(num_cus) * 2u * PT_MB_QUADS * sizeof(float4); }

#define PT_PATH_LDS_STACK 16      /* stack entries per lane the path kernel keeps in LDS when PT_FORCE_SPILL asks for a spilling stack */
#define PT_PATH_LDS_ALL 30        /* deepest tree whose whole stack the path kernel keeps in LDS (own-record fetch) */
#define PT_PATH_LDS_MIN 8         /* the fewest stack entries per lane the path kernel ever keeps in LDS */

static size_t block_lds_bytes(uint32_t entries) {
    return (size_t)PT_WAVES_PER_BLOCK * (size_t)entries * 64 * sizeof(uint32_t);
}

/* stack entries per lane the path kernel spills to global memory, for sizing PTKernelParams::spill */
uint32_t pt_spill_entries(uint32_t stack_entries) {
    return stack_entries > PT_PATH_LDS_MIN ? stack_entries - PT_PATH_LDS_MIN : 0;
}

hipError_t pt_launch_primary(const PTKernelParams &kp, bool exact, bool count, bool aov, hipStream_t st) {
    if (kp.stack_entries < 1 || kp.stack_entries > (uint32_t)pt_max_stack_entries()) return hipErrorInvalidValue;
    dim3 grid((kp.width + 15) / 16, (kp.height + 15) / 16), block(PT_BLOCK);
    if (!aov && kp.shard_world > 1u) {        /* owned tiles only: see primary_kernel */
        const uint32_t tiles_x = (kp.width + 7) / 8, tiles_y = (kp.height + 7) / 8;
        const uint32_t owned_per_row = (tiles_x + kp.shard_world - 1) / kp.shard_world;
        grid = dim3((owned_per_row + PT_WAVES_PER_BLOCK - 1) / PT_WAVES_PER_BLOCK, tiles_y);
    }
    const size_t lds = block_lds_bytes(kp.stack_entries);
    if (kp.wide) {            /* four-wide nodes: no work counters in this mode */
        if (count) return hipErrorInvalidValue;
        if (aov)        hipLaunchKernelGGL((primary_kernel<true, false, true, true>), grid, block, lds, st, kp);
        else if (exact) hipLaunchKernelGGL((primary_kernel<true, false, false, true>), grid, block, lds, st, kp);
        else            hipLaunchKernelGGL((primary_kernel<false, false, false, true>), grid, block, lds, st, kp);
        return hipGetLastError();
    }
    if (aov) {
        hipLaunchKernelGGL((primary_kernel<true, false, true>), grid, block, lds, st, kp);
    } else if (count) {
        if (exact) hipLaunchKernelGGL((primary_kernel<true, true, false>), grid, block, lds, st, kp);
        else       hipLaunchKernelGGL((primary_kernel<false, true, false>), grid, block, lds, st, kp);
    } else {
        if (exact) hipLaunchKernelGGL((primary_kernel<true, false, false>), grid, block, lds, st, kp);
        else       hipLaunchKernelGGL((primary_kernel<false, false, false>), grid, block, lds, st, kp);
    }
    return hipGetLastError();
}

template <bool SPILL, bool COOP>
static void launch_paths_variant(const PTKernelParams &kp, bool exact, bool count, dim3 grid, size_t lds, uint32_t lds_entries,
                                 hipStream_t st) {
    dim3 block(PT_PATH_BLOCK);
    if (kp.wide) {          /* the four-wide walk fetches its own records */
        if (exact) hipLaunchKernelGGL((path_kernel<true, false, SPILL, true>), grid, block, lds, st, kp, lds_entries);
        else       hipLaunchKernelGGL((path_kernel<false, false, SPILL, true>), grid, block, lds, st, kp, lds_entries);
        return;
    }
    if (kp.two_paths) {
        if (count) {
            if (exact) hipLaunchKernelGGL((path_kernel_mb<true, true, SPILL, COOP>), grid, block, lds, st, kp, lds_entries);
            else       hipLaunchKernelGGL((path_kernel_mb<false, true, SPILL, COOP>), grid, block, lds, st, kp, lds_entries);
        } else {
            if (exact) hipLaunchKernelGGL((path_kernel_mb<true, false, SPILL, COOP>), grid, block, lds, st, kp, lds_entries);
            else       hipLaunchKernelGGL((path_kernel_mb<false, false, SPILL, COOP>), grid, block, lds, st, kp, lds_entries);
        }
        return;
    }
    if (count) {
        if (exact) hipLaunchKernelGGL((path_kernel<true, true, SPILL, false, COOP>), grid, block, lds, st, kp, lds_entries);
        else       hipLaunchKernelGGL((path_kernel<false, true, SPILL, false, COOP>), grid, block, lds, st, kp, lds_entries);
    } else {
        if (exact) hipLaunchKernelGGL((path_kernel<true, false, SPILL, false, COOP>), grid, block, lds, st, kp, lds_entries);
        else       hipLaunchKernelGGL((path_kernel<false, false, SPILL, false, COOP>), grid, block, lds, st, kp, lds_entries);
    }
}

int pt_path_blocks_per_cu_max(void) { return PT_PATH_WAVES_PER_EU * 4 / PT_PATH_WAVES_PER_BLOCK; }

hipError_t pt_launch_paths(const PTKernelParams &kp, bool exact, bool count, int num_cus, int blocks_per_cu_cap, hipStream_t st) {
    if (kp.stack_entries < 1 || kp.stack_entries > (uint32_t)pt_max_stack_entries()) return hipErrorInvalidValue;
    if (kp.wide && count) return hipErrorInvalidValue;
    /* Resident blocks per CU are set by the registers (PT_PATH_WAVES_PER_EU waves per SIMD); each block's share of the
     * CU's 160 KiB of LDS holds its waves' stacks and, with the quad fetch, their exchange areas.  A tree whose
     * stack fits keeps it in LDS as a whole; a deeper one keeps the entries that fit (at least PT_PATH_LDS_MIN)
     * there and the rest in global memory. */
    const bool coop = !kp.wide && kp.quad_fetch;
    const int by_registers = PT_PATH_WAVES_PER_EU * 4 / PT_PATH_WAVES_PER_BLOCK;     /* 4 SIMDs x waves per SIMD / waves per block */
    const size_t wave_share = (160 * 1024 - 512) / (size_t)by_registers / PT_PATH_WAVES_PER_BLOCK;
    const size_t wave_fixed = coop ? PT_XCHG_BYTES : 0;
    uint32_t fit = wave_share > wave_fixed + 256 ? (uint32_t)((wave_share - wave_fixed) / 256) : 1u;
    if (!coop && fit > PT_PATH_LDS_ALL) fit = PT_PATH_LDS_ALL;
    if (fit < PT_PATH_LDS_MIN) fit = PT_PATH_LDS_MIN;
    if (const char *e = getenv("PT_LDS_STACK_ENTRIES")) { int v = atoi(e); if (v >= (int)PT_PATH_LDS_MIN && v <= 64) fit = (uint32_t)v; }
    bool spill = kp.stack_entries > fit;
    if (getenv("PT_FORCE_SPILL") && kp.stack_entries > PT_PATH_LDS_MIN) {
        spill = true;
        if (fit >= kp.stack_entries) fit = PT_PATH_LDS_STACK < kp.stack_entries ? PT_PATH_LDS_STACK : PT_PATH_LDS_MIN;
    }
    uint32_t lds_entries = spill ? fit : kp.stack_entries;
    const size_t lds = (size_t)PT_PATH_WAVES_PER_BLOCK * ((size_t)lds_entries * 256 + wave_fixed);
    /* persistent grid: as many blocks per CU as stay resident */
    int per_cu = (int)((160 * 1024 - 512) / lds);
    if (per_cu > by_registers) per_cu = by_registers;
    if (per_cu < 1) per_cu = 1;
    if (blocks_per_cu_cap >= 1 && blocks_per_cu_cap < per_cu) per_cu = blocks_per_cu_cap;
    if (const char *e = getenv("PT_BLOCKS_PER_CU")) { int v = atoi(e); if (v >= 1 && v < per_cu) per_cu = v; }
    dim3 grid(num_cus * per_cu);
    if (coop) {
        if (spill) launch_paths_variant<true, true>(kp, exact, count, grid, lds, lds_entries, st);
        else       launch_paths_variant<false, true>(kp, exact, count, grid, lds, lds_entries, st);
    } else {
        if (spill) launch_paths_variant<true, false>(kp, exact, count, grid, lds, lds_entries, st);
        else       launch_paths_variant<false, false>(kp, exact, count, grid, lds, lds_entries, st);
    }
    return hipGetLastError();
}

hipError_t pt_launch_resolve(const PTKernelParams &kp, hipStream_t st) {
    uint32_t n = kp.width * kp.height;
    hipLaunchKernelGGL(resolve_kernel, dim3((n + 255) / 256), dim3(256), 0, st, kp);
    return hipGetLastError();
}

hipError_t pt_launch_scale(float4 *accum, uint32_t n_pixels, float s, hipStream_t st) {
    hipLaunchKernelGGL(scale_kernel, dim3((n_pixels + 255) / 256), dim3(256), 0, st, accum, n_pixels, s);
    return hipGetLastError();
}
