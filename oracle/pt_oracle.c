/*
 * CPU oracle: restatement of src/shaders/shader.wgsl (reference) in C.
 * TEST INFRASTRUCTURE ONLY - see pt_oracle.h for the rules and for how the
 * implementation-defined WGSL built-ins are fixed.  Build with
 *   gcc -O2 -std=c11 -ffp-contract=off -fno-fast-math -fopenmp
 */
#include "pt_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifdef ORC_SHARED_MATH
/* log / cos / pow keep their own arithmetic in every build (the "contract" variant fuses the shader's expressions only) */
#pragma GCC push_options
#pragma GCC optimize("fp-contract=off")
#include "../path-tracing_amd/csrc/pt_portable_math.h"
#pragma GCC pop_options
#define ORC_LOG(x) pm_logf(x)
#define ORC_COS(x) pm_cosf(x)
#define ORC_POW(x, y) pm_powf((x), (y))
int orc_uses_shared_math(void) { return 1; }
#else
#define ORC_LOG(x) logf(x)
#define ORC_COS(x) cosf(x)
#define ORC_POW(x, y) powf((x), (y))
int orc_uses_shared_math(void) { return 0; }
#endif

#define ORC_PI 3.14159f            /* shader.wgsl:36,332-334,365 */
#define ORC_FAR 1000000.0f         /* shader.wgsl:142,177 */
#define ORC_BOX_MISS 999999999.0f  /* shader.wgsl:254 */
#define ORC_MAX_STACK 4096
/* lets an analysis tool that includes this file watch the node visits of a traversal */
#ifndef ORC_VISIT_HOOK
#define ORC_VISIT_HOOK(is_leaf) ((void)0)
#endif

/* ---- vector helpers with the evaluation order fixed in pt_oracle.h ---- */
static inline orc_v3 v3(float x, float y, float z) { orc_v3 r = {x, y, z}; return r; }
static inline orc_v3 vadd(orc_v3 a, orc_v3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline orc_v3 vsub(orc_v3 a, orc_v3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline orc_v3 vmul(orc_v3 a, orc_v3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline orc_v3 vscale(orc_v3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
/* Built-ins whose rounding WGSL leaves to the implementation are pinned as pt_oracle.h says.  The ORC_VAR_* builds
 * make one other choice each that a WGSL implementation may legally make; they exist to measure how much of the
 * pinned results hangs on the choice (tests/test_builtin_variants.py), and nothing else uses them. */
#ifdef ORC_VAR_DOT_FMA        /* dot() as a chain of fused multiply-adds */
static inline float vdot(orc_v3 a, orc_v3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
#else
static inline float vdot(orc_v3 a, orc_v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
#endif
static inline orc_v3 vcross(orc_v3 a, orc_v3 b) {
    return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
#ifdef ORC_VAR_NORMALIZE_DIV  /* normalize(v) = v / length(v), one division per component */
static inline orc_v3 vnormalize(orc_v3 a) {
    float len = sqrtf(vdot(a, a));
    return v3(a.x / len, a.y / len, a.z / len);
}
#else
static inline orc_v3 vnormalize(orc_v3 a) {
    float inv = 1.0f / sqrtf(vdot(a, a));
    return vscale(a, inv);
}
#endif
#ifdef ORC_VAR_MINMAX_NAN     /* min / max (and the clamp of smoothstep) hand a NaN operand on */
static inline float minnum(float a, float b) { return (a != a || b != b) ? NAN : (a < b ? a : b); }
static inline float maxnum(float a, float b) { return (a != a || b != b) ? NAN : (a > b ? a : b); }
#else
static inline float minnum(float a, float b) { return (b != b) ? a : ((a != a) ? b : (a < b ? a : b)); }
static inline float maxnum(float a, float b) { return (b != b) ? a : ((a != a) ? b : (a > b ? a : b)); }
#endif
#ifdef ORC_VAR_CAMERA_F32     /* the camera's sin / cos / tan in binary32 (libm) instead of double, rounded once */
#define ORC_CAM_SIN(x) sinf(x)
#define ORC_CAM_COS(x) cosf(x)
#define ORC_CAM_TAN(x) tanf(x)
#else
#define ORC_CAM_SIN(x) ((float)sin((double)(x)))
#define ORC_CAM_COS(x) ((float)cos((double)(x)))
#define ORC_CAM_TAN(x) ((float)tan((double)(x)))
#endif
static inline orc_v3 from3(const float *p) { return v3(p[0], p[1], p[2]); }

/* shader.wgsl:385-388 */
static inline orc_v3 lerp3(orc_v3 a, orc_v3 b, float t) {
    return vadd(vscale(a, 1.0f - t), vscale(b, t));
}

static inline float smoothstepf(float e0, float e1, float x) {
    float t = (x - e0) / (e1 - e0);
    t = minnum(maxnum(t, 0.0f), 1.0f);
    return t * t * (3.0f - 2.0f * t);
}

void orc_default_params(orc_params *p) {
    memset(p, 0, sizeof(*p));
    p->width = 1200; p->height = 600;            /* shader.wgsl:35 */
    p->rays_per_pixel = 20;                      /* shader.wgsl:34 */
    p->max_bounce_count = 10;                    /* shader.wgsl:33 */
    p->stack_entries = 17;                       /* shader.wgsl:30,170 */
    p->fov_degrees = 60.0f;                      /* shader.wgsl:36 */
    p->tri_color[0] = 0.65f; p->tri_color[1] = 0.65f; p->tri_color[2] = 1.0f; /* :283 */
    p->tri_emission_strength = 0.0f;             /* :285 */
    p->tri_smoothness = 0.9f;                    /* :286 */
    p->sky_horizon[0] = 0.5f; p->sky_horizon[1] = 0.7f; p->sky_horizon[2] = 1.0f;   /* :48 */
    p->sky_zenith[0] = 0.1f; p->sky_zenith[1] = 0.25f; p->sky_zenith[2] = 1.0f;     /* :49 */
    p->ground[0] = 0.2f; p->ground[1] = 0.2f; p->ground[2] = 0.2f;                  /* :50 */
    p->sun_dir[0] = 0.0f; p->sun_dir[1] = -0.4f; p->sun_dir[2] = 0.5f;              /* :51 */
    p->sun_intensity = 3.0f; p->sun_focus = 200.0f;                                 /* :52-53 */
    p->use_environment_lighting = 1;                                                /* :54 */
    p->far_distance = ORC_FAR;
}

/* ---- RNG: shader.wgsl:351-375 ---- */
float orc_random_value(uint32_t seed) {
    uint32_t state = seed;
    state = state ^ (state << 13);
    state = state ^ (state >> 17);
    state = state ^ (state << 5);
    return (float)state / 4294967296.0f;   /* f32(u32) rounds to nearest: can be exactly 1.0 */
}

float orc_random_normal(uint32_t seed) {
    float u1 = orc_random_value(seed);
    float u2 = orc_random_value(seed * 7462u);
    return sqrtf(-2.0f * ORC_LOG(u1)) * ORC_COS(2.0f * ORC_PI * u2);
}

static orc_v3 random_direction(uint32_t seed) {
    float x = orc_random_normal(seed);
    float y = orc_random_normal(seed * 379u);
    float z = orc_random_normal(seed * 123u);
    return vnormalize(v3(x, y, z));
}

/* shader.wgsl:378-382 (never called by the shader).  WGSL sign(): 1, 0 or -1 */
void orc_random_hemisphere_direction(const float normal[3], uint32_t seed, float out[3]) {
    orc_v3 dir = random_direction(seed);
    float d = vdot(from3(normal), dir);
    float sg = (d > 0.0f) ? 1.0f : ((d < 0.0f) ? -1.0f : 0.0f);
    out[0] = dir.x * sg; out[1] = dir.y * sg; out[2] = dir.z * sg;
}

void orc_random_direction(uint32_t seed, float out[3]) {
    orc_v3 d = random_direction(seed);
    out[0] = d.x; out[1] = d.y; out[2] = d.z;
}

/* ---- camera: shader.wgsl:327-348.  The six sin/cos values are evaluated in
 * double and rounded once, so they do not depend on a libm's float routines. */
static orc_v3 rotate_vector(orc_v3 ray, orc_v3 angles) {
    float x = ray.x, y = ray.y, z = ray.z;
    float a = angles.x * ORC_PI / 180.0f;
    float b = angles.y * ORC_PI / 180.0f;
    float c = angles.z * ORC_PI / 180.0f;
    float cos_a = ORC_CAM_COS(a), sin_a = ORC_CAM_SIN(a);
    float cos_b = ORC_CAM_COS(b), sin_b = ORC_CAM_SIN(b);
    float cos_c = ORC_CAM_COS(c), sin_c = ORC_CAM_SIN(c);
    float x_rot = x * cos_c * cos_b + y * (cos_c * sin_b * sin_a - sin_c * cos_a) + z * (cos_c * sin_b * cos_a + sin_c * sin_a);
    float y_rot = x * sin_c * cos_b + y * (sin_c * sin_b * sin_a + cos_c * cos_a) + z * (sin_c * sin_b * cos_a - cos_c * sin_a);
    float z_rot = -x * sin_b + y * cos_b * sin_a + z * cos_b * cos_a;
    return v3(x_rot, y_rot, z_rot);
}

void orc_rotate_vector(const float v[3], const float angles_deg[3], float out[3]) {
    orc_v3 r = rotate_vector(from3(v), from3(angles_deg));
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}

/* shader.wgsl:36-37,68-69 (vs_main) */
void orc_screen_size(const orc_params *p, float *screen_width, float *screen_height) {
    /* `const fov: f32 = 60.0 * 3.14159 / 180.0` is a WGSL const-expression of
     * abstract floats: evaluated in f64, rounded to f32 once (shader.wgsl:36) */
    float fov = (float)((double)p->fov_degrees * 3.14159 / 180.0);
    float aspect = (float)p->width / (float)p->height;
    float sw = ORC_CAM_TAN(fov * 0.5f) * 2.0f;
    *screen_width = sw;
    *screen_height = sw / aspect;
}

/* shader.wgsl:80-95: in.pos is the pixel centre (x+0.5, y+0.5), y down */
void orc_primary_ray(const orc_params *p, const float cam_pos[3], const float cam_rot_deg[3],
                     uint32_t px, uint32_t py, float origin[3], float dir[3], uint32_t *pixel_index) {
    float sw, sh;
    orc_screen_size(p, &sw, &sh);
    float W = (float)p->width, H = (float)p->height;
    float pos_x = (float)px + 0.5f, pos_y = (float)py + 0.5f;
    float u = (2.0f * pos_x / W - 1.0f) * sw / 2.0f;
    float v = (1.0f - 2.0f * pos_y / H) * sh / 2.0f;
    if (pixel_index) *pixel_index = (uint32_t)(pos_x + pos_y * W);
    orc_v3 d = vnormalize(v3(u, v, -1.0f));
    d = rotate_vector(d, from3(cam_rot_deg));
    origin[0] = cam_pos[0]; origin[1] = cam_pos[1]; origin[2] = cam_pos[2];
    dir[0] = d.x; dir[1] = d.y; dir[2] = d.z;
}

/* ---- intersections ---- */
/* shader.wgsl:237-256 */
static float ray_box(orc_v3 o, orc_v3 d, const float *b) {
    float tminx = (b[0] - o.x) / d.x, tminy = (b[1] - o.y) / d.y, tminz = (b[2] - o.z) / d.z;
    float tmaxx = (b[3] - o.x) / d.x, tmaxy = (b[4] - o.y) / d.y, tmaxz = (b[5] - o.z) / d.z;
    float t1x = minnum(tminx, tmaxx), t1y = minnum(tminy, tmaxy), t1z = minnum(tminz, tmaxz);
    float t2x = maxnum(tminx, tmaxx), t2y = maxnum(tminy, tmaxy), t2z = maxnum(tminz, tmaxz);
    float t_near = maxnum(maxnum(t1x, t1y), t1z);
    float t_far = minnum(minnum(t2x, t2y), t2z);
    int did_hit = (t_near <= t_far) && (t_far >= 0.0f);
    return did_hit ? t_near : ORC_BOX_MISS;
}

float orc_ray_box(const float origin[3], const float dir[3], const float box[6]) {
    return ray_box(from3(origin), from3(dir), box);
}

static void hit_clear(orc_hit *h) {
    memset(h, 0, sizeof(*h));       /* WGSL `var x: HitInfo;` is zero-initialised */
    h->prim_id = INT32_MIN;
}

/* shader.wgsl:258-290 */
static void ray_triangle(const orc_params *p, orc_v3 o, orc_v3 d, const float *t, orc_hit *h) {
    hit_clear(h);
    orc_v3 A = from3(t), B = from3(t + 3), C = from3(t + 6);
    orc_v3 edge_ab = vsub(B, A);
    orc_v3 edge_ac = vsub(C, A);
    orc_v3 normal_vector = vcross(edge_ab, edge_ac);
    orc_v3 ao = vsub(o, A);
    orc_v3 dao = vcross(ao, d);
    float determinant = -vdot(d, normal_vector);
    float inv_det = 1.0f / determinant;
    float dst = vdot(ao, normal_vector) * inv_det;
    float u = vdot(edge_ac, dao) * inv_det;
    float v = -vdot(edge_ab, dao) * inv_det;
    float w = 1.0f - u - v;
    if (dst >= 0.0f && u >= 0.0f && v >= 0.0f && w >= 0.0f) {
        h->did_hit = determinant >= 0.0001f;   /* the repeated >= 0 tests of :279 are already true here */
        h->distance = dst;
        h->position = vadd(o, vscale(d, dst));
        h->normal = vnormalize(normal_vector);
        h->color = from3(p->tri_color);
        h->emission_color = from3(p->tri_emission_color);
        h->emission_strength = p->tri_emission_strength;
        h->smoothness = p->tri_smoothness;
    }
}

void orc_ray_triangle(const orc_params *p, const float origin[3], const float dir[3],
                      const float tri[9], orc_hit *out) {
    ray_triangle(p, from3(origin), from3(dir), tri, out);
}

/* shader.wgsl:292-325 */
static void ray_sphere(orc_v3 o, orc_v3 d, const float *s, orc_hit *h) {
    hit_clear(h);
    orc_v3 center = from3(s);
    float radius = s[3];
    orc_v3 oc = vsub(o, center);
    float a = vdot(d, d);
    float b = 2.0f * vdot(oc, d);
    float c = vdot(oc, oc) - radius * radius;
    float discriminant = b * b - 4.0f * a * c;
    if (discriminant >= 0.0f) {
        float distance = (-b - sqrtf(discriminant)) / (2.0f * a);
        if (distance >= 0.0f) {
            h->did_hit = 1;
            h->distance = distance;
            h->position = vadd(o, vscale(d, distance));
            h->normal = vnormalize(vsub(h->position, center));
            h->color = from3(s + 4);
            h->emission_color = from3(s + 7);
            h->emission_strength = s[10];
            h->smoothness = s[11];
        }
    }
}

void orc_ray_sphere(const float origin[3], const float dir[3], const float sphere[12], orc_hit *out) {
    ray_sphere(from3(origin), from3(dir), sphere, out);
}

/* shader.wgsl:169-235.  The stack is sized generously here; the occupancy the
 * shader's array<i32, bvh_max_depth+1> would need is reported in c->max_stack. */
void orc_ray_triangle_bvh(const orc_params *p, const orc_scene *s, const float origin[3],
                          const float dir[3], orc_hit *result, orc_counters *c) {
    orc_v3 o = from3(origin), d = from3(dir);
    int32_t node_stack[ORC_MAX_STACK];
    int stack_index = 0;
    node_stack[stack_index++] = 0;
    hit_clear(result);
    result->distance = p->far_distance;
    uint64_t boxes = 0, tris = 0, popped = 0, max_stack = 1;

    while (stack_index > 0) {
        stack_index--;
        int32_t node_index = node_stack[stack_index];
        const float *node = s->bvh + 9 * (size_t)node_index;   /* get_node, :165-167 */
        popped++;
        ORC_VISIT_HOOK(node[8] == 0.0f);
        if (node[8] == 0.0f) {
            uint32_t first = (uint32_t)node[6];
            uint32_t end = (uint32_t)(node[6] + node[7]);
            for (uint32_t i = first; i < end; i++) {
                orc_hit th;
                ray_triangle(p, o, d, s->triangles + 9 * (size_t)i, &th);
                tris++;
                if (th.did_hit && th.distance < result->distance) {
                    *result = th;
                    result->prim_id = (int32_t)i;
                }
            }
        } else {
            int32_t child_index_a = (int32_t)node[8];
            int32_t child_index_b = (int32_t)(node[8] + 1.0f);
            const float *child_a = s->bvh + 9 * (size_t)child_index_a;
            const float *child_b = s->bvh + 9 * (size_t)child_index_b;
            float dst_a = ray_box(o, d, child_a);
            float dst_b = ray_box(o, d, child_b);
            boxes += 2;
            float dst_near, dst_far;
            int32_t child_index_near, child_index_far;
            if (dst_a < dst_b) {
                dst_near = dst_a; dst_far = dst_b;
                child_index_near = child_index_a; child_index_far = child_index_b;
            } else {
                dst_near = dst_b; dst_far = dst_a;
                child_index_near = child_index_b; child_index_far = child_index_a;
            }
            if (dst_far < result->distance) node_stack[stack_index++] = child_index_far;
            if (dst_near < result->distance) node_stack[stack_index++] = child_index_near;
            if ((uint64_t)stack_index > max_stack) max_stack = (uint64_t)stack_index;
            if (stack_index > ORC_MAX_STACK - 2) break;   /* malformed (cyclic) file; never reached on valid input */
        }
    }
    if (c) {
        c->box_tests += boxes; c->tri_tests += tris; c->nodes_popped += popped;
        if (max_stack > c->max_stack) c->max_stack = max_stack;
    }
}

void orc_ray_triangles_brute(const orc_params *p, const orc_scene *s, const float origin[3],
                             const float dir[3], orc_hit *result) {
    orc_v3 o = from3(origin), d = from3(dir);
    hit_clear(result);
    result->distance = p->far_distance;
    for (uint32_t i = 0; i < s->triangle_count; i++) {
        orc_hit th;
        ray_triangle(p, o, d, s->triangles + 9 * (size_t)i, &th);
        if (th.did_hit && th.distance < result->distance) {
            *result = th;
            result->prim_id = (int32_t)i;
        }
    }
}

/* shader.wgsl:139-163 */
void orc_calculate_ray_collision(const orc_params *p, const orc_scene *s, const float origin[3],
                                 const float dir[3], orc_hit *closest, orc_counters *c) {
    orc_v3 o = from3(origin), d = from3(dir);
    hit_clear(closest);
    closest->distance = p->far_distance;
    for (uint32_t i = 0; i < s->sphere_count; i++) {
        orc_hit h;
        ray_sphere(o, d, s->spheres + 12 * (size_t)i, &h);
        if (h.did_hit && h.distance < closest->distance) {
            *closest = h;
            closest->prim_id = -(int32_t)(1 + i);
        }
    }
    orc_hit h;
    orc_ray_triangle_bvh(p, s, origin, dir, &h, c);
    if (h.did_hit && h.distance < closest->distance) *closest = h;
    if (c) c->ray_casts++;
}

/* shader.wgsl:391-404 */
static orc_v3 environment_light(const orc_params *p, orc_v3 dir) {
    float sky_gradient_t = ORC_POW(smoothstepf(0.0f, 0.4f, dir.y), 0.35f);
    orc_v3 sky_gradient = lerp3(from3(p->sky_horizon), from3(p->sky_zenith), sky_gradient_t);
    orc_v3 nsun = vnormalize(from3(p->sun_dir));
    nsun = v3(-nsun.x, -nsun.y, -nsun.z);
    float sun = ORC_POW(maxnum(0.0f, vdot(dir, nsun)), p->sun_focus) * p->sun_intensity;
    float ground_to_sky_t = smoothstepf(-0.01f, 0.0f, dir.y);
    float sun_mask = (ground_to_sky_t >= 1.0f) ? 1.0f : 0.0f;
    orc_v3 r = lerp3(from3(p->ground), sky_gradient, ground_to_sky_t);
    float ss = sun * sun_mask;
    return v3(r.x + ss, r.y + ss, r.z + ss);
}

void orc_environment_light(const orc_params *p, const float dir[3], float out[3]) {
    orc_v3 r = environment_light(p, from3(dir));
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}

/* shader.wgsl:111-137 */
void orc_trace(const orc_params *p, const orc_scene *s, const float origin[3], const float dir[3],
               uint32_t seed, uint32_t frame_count, float out[3], orc_counters *c) {
    orc_v3 ro = from3(origin), rd = from3(dir);
    orc_v3 incoming_light = v3(0, 0, 0);
    orc_v3 ray_color = v3(1, 1, 1);
    if (c) c->paths++;
    for (uint32_t i = 0; i <= p->max_bounce_count; i++) {
        orc_hit hit;
        float o3[3] = {ro.x, ro.y, ro.z}, d3[3] = {rd.x, rd.y, rd.z};
        orc_calculate_ray_collision(p, s, o3, d3, &hit, c);
        if (hit.did_hit) {
            ro = hit.position;
            uint32_t rs = seed + i * 12345u + frame_count * 393939123u;
            orc_v3 diffuse_dir = vnormalize(vadd(hit.normal, random_direction(rs)));
            float dn2 = 2.0f * vdot(hit.normal, rd);
            orc_v3 specular_dir = vsub(rd, vscale(hit.normal, dn2));       /* reflect */
            rd = lerp3(diffuse_dir, specular_dir, hit.smoothness);        /* not re-normalised */
            orc_v3 emitted_light = vscale(hit.emission_color, hit.emission_strength);
            incoming_light = vadd(incoming_light, vmul(emitted_light, ray_color));
            ray_color = vmul(ray_color, hit.color);
        } else {
            if (p->use_environment_lighting) {
                incoming_light = vadd(incoming_light, vmul(environment_light(p, rd), ray_color));
            }
            break;
        }
    }
    out[0] = incoming_light.x; out[1] = incoming_light.y; out[2] = incoming_light.z;
}

static int check_args(const orc_params *p, const orc_scene *s) {
    if (!p || !s || p->width == 0 || p->height == 0 || p->rays_per_pixel == 0) return -1;
    if (s->node_count == 0 || !s->bvh) return -1;
    if (s->triangle_count && !s->triangles) return -1;
    if (s->sphere_count && !s->spheres) return -1;
    return 0;
}

static void counters_add(orc_counters *dst, const orc_counters *src) {
    dst->ray_casts += src->ray_casts; dst->box_tests += src->box_tests;
    dst->tri_tests += src->tri_tests; dst->nodes_popped += src->nodes_popped;
    dst->paths += src->paths;
    if (src->max_stack > dst->max_stack) dst->max_stack = src->max_stack;
}

/* mode 0: running mean (shader.wgsl:104-106); mode 1: add the frame mean to the buffer */
static int render_frame_impl(const orc_params *p, const orc_scene *s, const float cam_pos[3],
                             const float cam_rot_deg[3], uint32_t frame_count, float *accum,
                             uint32_t x0, uint32_t x1, uint32_t sx, uint32_t y0, uint32_t y1, uint32_t sy,
                             int nthreads, orc_counters *c, int mode) {
    if (check_args(p, s) || !accum || sx == 0 || sy == 0) return -1;
    if (x1 > p->width) x1 = p->width;
    if (y1 > p->height) y1 = p->height;
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#else
    nthreads = 1;
#endif
    orc_counters total; memset(&total, 0, sizeof(total));
    long nrows = (y1 > y0) ? (long)((y1 - y0 + sy - 1) / sy) : 0;
#pragma omp parallel num_threads(nthreads)
    {
        orc_counters local; memset(&local, 0, sizeof(local));
#pragma omp for schedule(dynamic, 1)
        for (long r = 0; r < nrows; r++) {
            uint32_t y = y0 + (uint32_t)r * sy;
            for (uint32_t x = x0; x < x1; x += sx) {
                float o[3], d[3]; uint32_t pixel_index;
                orc_primary_ray(p, cam_pos, cam_rot_deg, x, y, o, d, &pixel_index);
                orc_v3 pixel_color = v3(0, 0, 0);
                for (uint32_t i = 0; i < p->rays_per_pixel; i++) {
                    float t[3];
                    orc_trace(p, s, o, d, pixel_index + i * 248135u, frame_count, t, c ? &local : NULL);
                    pixel_color = vadd(pixel_color, from3(t));
                }
                float n = (float)p->rays_per_pixel;
                pixel_color = v3(pixel_color.x / n, pixel_color.y / n, pixel_color.z / n);
                float *px = accum + 4 * ((size_t)y * p->width + x);
                if (mode == 0) {
                    float weight = 1.0f / (float)(frame_count + 1u);
                    orc_v3 old = from3(px);
                    orc_v3 wa = vadd(vscale(old, 1.0f - weight), vscale(pixel_color, weight));
                    px[0] = wa.x; px[1] = wa.y; px[2] = wa.z;
                } else {
                    px[0] += pixel_color.x; px[1] += pixel_color.y; px[2] += pixel_color.z;
                }
            }
        }
#pragma omp critical
        counters_add(&total, &local);
    }
    if (c) counters_add(c, &total);
    return 0;
}

int orc_render_frame(const orc_params *p, const orc_scene *s, const float cam_pos[3],
                     const float cam_rot_deg[3], uint32_t frame_count, float *accum,
                     uint32_t x0, uint32_t x1, uint32_t sx, uint32_t y0, uint32_t y1, uint32_t sy,
                     int nthreads, orc_counters *c) {
    return render_frame_impl(p, s, cam_pos, cam_rot_deg, frame_count, accum, x0, x1, sx, y0, y1, sy, nthreads, c, 0);
}

int orc_render_frames(const orc_params *p, const orc_scene *s, const float cam_pos[3],
                      const float cam_rot_deg[3], uint32_t first_frame, uint32_t n_frames,
                      float *accum, int nthreads, orc_counters *c) {
    for (uint32_t f = 0; f < n_frames; f++) {
        int rc = render_frame_impl(p, s, cam_pos, cam_rot_deg, first_frame + f, accum,
                                   0, p->width, 1, 0, p->height, 1, nthreads, c, 0);
        if (rc) return rc;
    }
    return 0;
}

int orc_render_frames_sum(const orc_params *p, const orc_scene *s, const float cam_pos[3],
                          const float cam_rot_deg[3], uint32_t first_frame, uint32_t n_frames,
                          uint32_t frame_stride, float *sum, int nthreads, orc_counters *c) {
    for (uint32_t f = 0; f < n_frames; f++) {
        int rc = render_frame_impl(p, s, cam_pos, cam_rot_deg, first_frame + f * frame_stride, sum,
                                   0, p->width, 1, 0, p->height, 1, nthreads, c, 1);
        if (rc) return rc;
    }
    return 0;
}

int orc_primary_hits(const orc_params *p, const orc_scene *s, const float cam_pos[3],
                     const float cam_rot_deg[3], int32_t *prim_id, float *distance, int nthreads) {
    if (check_args(p, s) || !prim_id || !distance) return -1;
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#else
    nthreads = 1;
#endif
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
    for (long y = 0; y < (long)p->height; y++) {
        for (uint32_t x = 0; x < p->width; x++) {
            float o[3], d[3];
            orc_primary_ray(p, cam_pos, cam_rot_deg, x, (uint32_t)y, o, d, NULL);
            orc_hit h;
            orc_calculate_ray_collision(p, s, o, d, &h, NULL);
            size_t i = (size_t)y * p->width + x;
            prim_id[i] = h.did_hit ? h.prim_id : INT32_MIN;
            distance[i] = h.distance;
        }
    }
    return 0;
}

int orc_bvh_depth(const float *bvh, uint32_t node_count) {
    if (!bvh || node_count == 0) return -1;
    /* iterative DFS carrying depth; a node file is a tree when no walk exceeds the cap */
    typedef struct { uint32_t node; int depth; } item;
    size_t cap = 8192; size_t n = 0;
    item *st = (item *)malloc(cap * sizeof(item));
    if (!st) return -1;
    st[n].node = 0; st[n].depth = 0; n++;
    int max_depth = 0;
    uint64_t visited = 0;
    while (n) {
        item it = st[--n];
        if (++visited > (uint64_t)node_count * 4u + 16u) { free(st); return -1; }
        if (it.depth > max_depth) max_depth = it.depth;
        if (it.depth > 4096) { free(st); return -1; }
        const float *node = bvh + 9 * (size_t)it.node;
        if (node[8] != 0.0f) {
            float cf = node[8];
            if (!(cf >= 1.0f) || cf + 1.0f >= (float)node_count + 0.5f) { free(st); return -1; }
            uint32_t ca = (uint32_t)cf;
            if (n + 2 > cap) { cap *= 2; st = (item *)realloc(st, cap * sizeof(item)); if (!st) return -1; }
            st[n].node = ca; st[n].depth = it.depth + 1; n++;
            st[n].node = ca + 1; st[n].depth = it.depth + 1; n++;
        }
    }
    free(st);
    return max_depth;
}
