/* Scheduling model of the path kernel with deferred leaves (analysis tool, not part of the product or the tests).
 *
 * A lane that reaches a leaf may put the leaf aside and walk on speculatively (with the best hit it had) until
 * the wave's next leaf turn; if the leaf's triangles then improve the best hit the speculative steps are thrown
 * away and the lane resumes right after the leaf.  The model replays the real visit sequences of the oracle
 * (node kind, stack depth and best distance at every visit) under that rule and reports wave steps and time.
 *
 * build: gcc -O2 -shared -fPIC -fopenmp -o tools/sim/libspec_sim.so tools/sim/spec_sim.c -lm
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static void sim_visit(int is_leaf, int depth, float best);
#define ORC_VISIT_HOOK(is_leaf) sim_visit(is_leaf, stack_index, result->distance)
#include "../../oracle/pt_oracle.c"

typedef struct { uint8_t leaf; uint8_t depth; uint8_t improved; } visit_t;     /* improved: this leaf lowered the best distance */
typedef struct { uint32_t first_visit, n_visits; } cast_t;
typedef struct { uint32_t first_cast, n_casts, pixel; } path_t;

static visit_t *g_visits; static size_t g_nvis, g_capvis;
static cast_t *g_casts; static size_t g_ncast, g_capcast;
static path_t *g_paths; static size_t g_npath, g_cappath;
static int g_recording;
static float g_prev_best; static long g_prev_leaf = -1;

static void sim_visit(int is_leaf, int depth, float best) {
    if (!g_recording) return;
    if (g_prev_leaf >= 0 && best != g_prev_best) g_visits[g_prev_leaf].improved = 1;
    if (g_nvis == g_capvis) { g_capvis = g_capvis ? g_capvis * 2 : (1u << 20); g_visits = realloc(g_visits, g_capvis * sizeof(visit_t)); }
    g_visits[g_nvis].leaf = (uint8_t)is_leaf; g_visits[g_nvis].depth = (uint8_t)depth; g_visits[g_nvis].improved = 0;
    g_prev_leaf = is_leaf ? (long)g_nvis : -1; g_prev_best = best;
    g_nvis++;
}

int sim_profile(const orc_params *p, const orc_scene *s, const float cam[3], const float rot[3],
                uint32_t x0, uint32_t y0, uint32_t w, uint32_t h, uint32_t frame) {
    g_nvis = g_ncast = g_npath = 0;
    for (uint32_t ty = 0; ty < h; ty += 8) for (uint32_t tx = 0; tx < w; tx += 8)
    for (uint32_t l = 0; l < 64; l++) {
        uint32_t x = x0 + tx + (l & 7), y = y0 + ty + (l >> 3);
        float o[3], d[3]; uint32_t pixel_index;
        orc_primary_ray(p, cam, rot, x, y, o, d, &pixel_index);
        orc_hit first;
        g_recording = 0;
        orc_calculate_ray_collision(p, s, o, d, &first, NULL);
        if (!first.did_hit) continue;
        for (uint32_t smp = 0; smp < p->rays_per_pixel; smp++) {
            uint32_t seed = pixel_index + smp * 248135u;
            orc_v3 ro = from3(o), rd = from3(d);
            if (g_npath == g_cappath) { g_cappath = g_cappath ? g_cappath * 2 : 4096; g_paths = realloc(g_paths, g_cappath * sizeof(path_t)); }
            path_t *P = &g_paths[g_npath++]; P->first_cast = (uint32_t)g_ncast; P->n_casts = 0; P->pixel = y * p->width + x;
            for (uint32_t i = 0; i <= p->max_bounce_count; i++) {
                orc_hit hit;
                float o3[3] = {ro.x, ro.y, ro.z}, d3[3] = {rd.x, rd.y, rd.z};
                g_recording = i > 0; g_prev_leaf = -1;
                size_t v0 = g_nvis;
                orc_calculate_ray_collision(p, s, o3, d3, &hit, NULL);
                g_recording = 0;
                if (i > 0) {
                    if (g_ncast == g_capcast) { g_capcast = g_capcast ? g_capcast * 2 : 65536; g_casts = realloc(g_casts, g_capcast * sizeof(cast_t)); }
                    g_casts[g_ncast].first_visit = (uint32_t)v0; g_casts[g_ncast].n_visits = (uint32_t)(g_nvis - v0);
                    g_ncast++; g_paths[g_npath - 1].n_casts++;
                }
                if (!hit.did_hit) break;
                ro = hit.position;
                uint32_t rs = seed + i * 12345u + frame * 393939123u;
                orc_v3 diffuse_dir = vnormalize(vadd(hit.normal, random_direction(rs)));
                float dn2 = 2.0f * vdot(hit.normal, rd);
                orc_v3 specular_dir = vsub(rd, vscale(hit.normal, dn2));
                rd = lerp3(diffuse_dir, specular_dir, hit.smoothness);
            }
        }
    }
    return (int)g_npath;
}

/* per path: pixel, bounce casts, node visits of all its casts */
void sim_paths(uint32_t *pixel, uint32_t *casts, uint32_t *visits) {
    for (size_t i = 0; i < g_npath; i++) {
        pixel[i] = g_paths[i].pixel; casts[i] = g_paths[i].n_casts;
        uint32_t v = 0;
        for (uint32_t c = 0; c < g_paths[i].n_casts; c++) v += g_casts[g_paths[i].first_cast + c].n_visits;
        visits[i] = v;
    }
}

void sim_sizes(uint64_t out[4]) {
    out[0] = g_npath; out[1] = g_ncast; out[2] = g_nvis; out[3] = 0;
    for (size_t i = 0; i < g_nvis; i++) out[3] += g_visits[i].leaf && g_visits[i].improved;
}

typedef struct {
    int waves, refill_min, leaf_period, steps_per_check;
    int spec;             /* 0: a lane at a leaf waits for the leaf turn; 1: it puts the leaf aside and walks on */
    int defer_free;       /* 1: putting a leaf aside takes no step of its own (done when the leaf is popped) */
    int chain;            /* leaf turn: up to this many further leaves in a row are taken in the same turn (0 = one leaf per turn) */
    double cost_inner, cost_leaf, cost_shade, cost_chain;
} sim_policy;

typedef struct { double time, steps, inner_lane_steps, wasted_lane_steps, leaf_runs, leaf_lane_steps, shade_rounds, shade_lanes, stalls_leaf, stalls_floor, rollbacks; } sim_result;

typedef struct {
    int path, cast, pos;      /* pos == n_visits: cast complete */
    int pending;              /* position of the leaf put aside, or -1 */
    int floor;                /* stack depth below which the speculative walk may not pop */
    int garbage;              /* the leaf put aside will improve the best hit: whatever the lane does until the turn is thrown away */
} lane_t;

static const cast_t *cast_of(const lane_t *l) { return &g_casts[g_paths[l->path].first_cast + l->cast]; }

int sim_run(const sim_policy *pol, sim_result *out) {
    memset(out, 0, sizeof(*out));
    const int W = pol->waves;
    lane_t *L = calloc((size_t)W * 64, sizeof(lane_t));
    double *wave_time = calloc((size_t)W, sizeof(double));
    char *wave_done = calloc((size_t)W, 1);
    size_t head = 0;
    for (int i = 0; i < W * 64; i++) { L[i].path = -1; L[i].pending = -1; }
    int alive = W;
    while (alive) {
        int w = -1;
        for (int k = 0; k < W; k++) if (!wave_done[k] && (w < 0 || wave_time[k] < wave_time[w])) w = k;
        lane_t *l = L + (size_t)w * 64;
        int shade_lanes = 0;
        for (int i = 0; i < 64; i++) {
            if (l[i].path >= 0 && l[i].pending < 0 && l[i].pos == (int)cast_of(&l[i])->n_visits) {
                shade_lanes++;
                if (l[i].cast + 1 < (int)g_paths[l[i].path].n_casts) l[i].cast++; else l[i].path = -1;
                l[i].pos = 0;
            }
            while (l[i].path < 0) {
                if (head >= g_npath) break;
                int p = (int)head++;
                if (g_paths[p].n_casts) { l[i].path = p; l[i].cast = 0; l[i].pos = 0; l[i].pending = -1; }
            }
        }
        out->shade_rounds++; out->shade_lanes += shade_lanes;
        wave_time[w] += pol->cost_shade;
        for (int step_no = 0;; step_no++) {
            int trav = 0, waiting = 0;
            for (int i = 0; i < 64; i++) {
                if (l[i].path < 0) continue;
                if (l[i].pending >= 0 || l[i].pos < (int)cast_of(&l[i])->n_visits) trav++; else waiting++;
            }
            const int check = pol->steps_per_check <= 1 || step_no % pol->steps_per_check == 0;
            if (check && (!trav || waiting >= pol->refill_min)) break;
            if (check && head >= g_npath && waiting >= trav) break;
            const int leaf_turn = (step_no % pol->leaf_period) == pol->leaf_period - 1;
            int n_inner = 0, n_leaf = 0, n_chain = 0;
            for (int i = 0; i < 64; i++) {
                if (l[i].path < 0) continue;
                const cast_t *c = cast_of(&l[i]);
                const visit_t *v = g_visits + c->first_visit;
                /* inner part of the step */
                if (l[i].pending >= 0 && l[i].garbage) { n_inner++; out->wasted_lane_steps++; }      /* walks on, for nothing */
                else if (l[i].pos < (int)c->n_visits) {
                    if (!v[l[i].pos].leaf) {
                        if (l[i].pending >= 0 && v[l[i].pos].depth < l[i].floor) out->stalls_floor++;
                        else {
                            n_inner++; l[i].pos++;
                            if (pol->spec && pol->defer_free && l[i].pending < 0 && l[i].pos < (int)c->n_visits && v[l[i].pos].leaf) {
                                l[i].pending = l[i].pos; l[i].garbage = v[l[i].pos].improved;
                                l[i].pos++;
                                l[i].floor = l[i].pos < (int)c->n_visits ? v[l[i].pos].depth : 255;
                            }
                        }
                    } else if (pol->spec && l[i].pending < 0) {
                        /* put the leaf aside, pop the next node and go on with it in the next step */
                        l[i].pending = l[i].pos; l[i].garbage = v[l[i].pos].improved;
                        l[i].pos++;
                        l[i].floor = l[i].pos < (int)c->n_visits ? v[l[i].pos].depth : 255;
                    } else if (l[i].pending >= 0) out->stalls_leaf++;
                }
                /* leaf part */
                if (leaf_turn) {
                    if (l[i].pending >= 0) {
                        n_leaf++;
                        if (l[i].garbage) { l[i].pos = l[i].pending + 1; out->rollbacks++; }
                        l[i].pending = -1; l[i].garbage = 0;
                    } else if (l[i].pos < (int)c->n_visits && v[l[i].pos].leaf && !pol->spec) {
                        n_leaf++; l[i].pos++;
                        for (int k = 0; k < pol->chain && l[i].pos < (int)c->n_visits && v[l[i].pos].leaf; k++) { l[i].pos++; if (k + 1 > n_chain) n_chain = k + 1; }
                    } else if (l[i].pos < (int)c->n_visits && v[l[i].pos].leaf && pol->spec) {
                        /* reached in this very step or stalled at it with nothing pending any more: taken directly */
                        n_leaf++; l[i].pos++;
                    }
                }
            }
            out->steps++; out->inner_lane_steps += n_inner;
            wave_time[w] += n_inner ? pol->cost_inner : 0.3 * pol->cost_inner;
            if (leaf_turn && n_leaf) { out->leaf_runs++; out->leaf_lane_steps += n_leaf; wave_time[w] += pol->cost_leaf + n_chain * pol->cost_chain; }
        }
        int any = 0;
        for (int i = 0; i < 64; i++) if (l[i].path >= 0) any = 1;
        if (!any && head >= g_npath) { wave_done[w] = 1; alive--; }
    }
    double sum = 0; for (int k = 0; k < W; k++) sum += wave_time[k];
    out->time = sum / W;
    free(L); free(wave_time); free(wave_done);
    return 0;
}

/* How far do all rays that leave one surface point walk the tree together?  For the primary hit point o of every pixel
 * of a region: the number of levels, from the root, at which o lies strictly inside one child box and strictly outside
 * the other on some axis (then every ray with finite, non-zero direction takes the first child as the near one and only
 * the other box needs a test of its own).  hist[k] = pixels whose walk in common is k levels long (k < 64);
 * out[0] = hit pixels, out[1] = sum of k, out[2] = sum of the depth at which the containing chain ends in a leaf or fails. */
void sim_prefix(const orc_params *p, const orc_scene *s, const float cam[3], const float rot[3],
                uint32_t x0, uint32_t y0, uint32_t w, uint32_t h, uint64_t hist[64], uint64_t out[4]) {
    memset(hist, 0, 64 * sizeof(uint64_t)); memset(out, 0, 4 * sizeof(uint64_t));
    for (uint32_t y = y0; y < y0 + h; y++) for (uint32_t x = x0; x < x0 + w; x++) {
        float o[3], d[3]; uint32_t pixel_index;
        orc_primary_ray(p, cam, rot, x, y, o, d, &pixel_index);
        orc_hit first;
        orc_calculate_ray_collision(p, s, o, d, &first, NULL);
        if (!first.did_hit) continue;
        const float q[3] = {first.position.x, first.position.y, first.position.z};
        uint32_t node = 0, k = 0;
        int ended_in_leaf = 0;
        for (;;) {
            const float *nd = s->bvh + 9 * (size_t)node;
            if (nd[8] == 0.0f) { ended_in_leaf = 1; break; }
            const uint32_t ca = (uint32_t)nd[8], cb = ca + 1;
            const float *a = s->bvh + 9 * (size_t)ca, *b = s->bvh + 9 * (size_t)cb;
            int in_a = 1, in_b = 1, out_a = 0, out_b = 0;
            for (int ax = 0; ax < 3; ax++) {
                if (!(a[ax] < q[ax] && q[ax] < a[3 + ax])) in_a = 0;
                if (!(b[ax] < q[ax] && q[ax] < b[3 + ax])) in_b = 0;
                if (q[ax] < a[ax] || q[ax] > a[3 + ax]) out_a = 1;
                if (q[ax] < b[ax] || q[ax] > b[3 + ax]) out_b = 1;
            }
            if (in_a && out_b) node = ca; else if (in_b && out_a) node = cb; else break;
            k++;
        }
        out[0]++; out[1] += k; out[2] += ended_in_leaf;
        hist[k < 63 ? k : 63]++;
    }
}
