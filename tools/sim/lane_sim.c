/* Scheduling model of the path kernel (analysis tool, not part of the product or the tests).
 *
 * sim_profile() traces real paths of a scene with the CPU oracle and records, for every bounce
 * cast of every path, the sequence of node visits (inner / leaf).  sim_run() then replays those
 * sequences on a model of W waves x 64 lanes under a lane-scheduling policy and reports how many
 * wave steps and shading rounds the policy needs.  Costs are in units of one inner-node step.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static void sim_visit(int is_leaf);
#define ORC_VISIT_HOOK(is_leaf) sim_visit(is_leaf)
#include "../../oracle/pt_oracle.c"

typedef struct { uint32_t first_visit, n_visits; } cast_t;
typedef struct { uint32_t first_cast, n_casts; } path_t;

static uint8_t *g_visits; static size_t g_nvis, g_capvis;
static cast_t *g_casts; static size_t g_ncast, g_capcast;
static path_t *g_paths; static size_t g_npath, g_cappath;
static int g_recording;

static void sim_visit(int is_leaf) {
    if (!g_recording) return;
    if (g_nvis == g_capvis) { g_capvis = g_capvis ? g_capvis * 2 : (1u << 20); g_visits = realloc(g_visits, g_capvis); }
    g_visits[g_nvis++] = (uint8_t)is_leaf;
}

/* traces the paths of pixels [x0,x0+w) x [y0,y0+h), 8x8 tiles in row-major order, rpp samples each */
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
        if (!first.did_hit) continue;                    /* miss pixels never reach the path kernel */
        for (uint32_t smp = 0; smp < p->rays_per_pixel; smp++) {
            /* trace(), shader.wgsl:111-137, with a record of every cast after the primary one */
            uint32_t seed = pixel_index + smp * 248135u;
            orc_v3 ro = from3(o), rd = from3(d);
            if (g_npath == g_cappath) { g_cappath = g_cappath ? g_cappath * 2 : 4096; g_paths = realloc(g_paths, g_cappath * sizeof(path_t)); }
            path_t *P = &g_paths[g_npath++]; P->first_cast = (uint32_t)g_ncast; P->n_casts = 0;
            for (uint32_t i = 0; i <= p->max_bounce_count; i++) {
                orc_hit hit;
                float o3[3] = {ro.x, ro.y, ro.z}, d3[3] = {rd.x, rd.y, rd.z};
                g_recording = i > 0;
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

/* How many inner-node visits are "stale": the node was pushed as a far child, and by the time it is
 * popped the best hit is already nearer than the node's entry distance.  out = {inner visits, stale
 * inner visits, leaf visits, stale leaf visits} over the bounce casts of the same paths as
 * sim_profile. */
static void count_stale(const orc_params *p, const orc_scene *s, orc_v3 o, orc_v3 d, uint64_t out[4]) {
    int32_t node_stack[ORC_MAX_STACK]; float dst_stack[ORC_MAX_STACK];
    int sp = 0;
    node_stack[sp] = 0; dst_stack[sp++] = -1.0f;
    float best = ORC_FAR;
    while (sp > 0) {
        sp--;
        int32_t ni = node_stack[sp]; float nd = dst_stack[sp];
        const float *node = s->bvh + 9 * (size_t)ni;
        int stale = nd >= best;
        if (node[8] == 0.0f) {
            out[2]++; out[3] += stale;
            uint32_t first = (uint32_t)node[6], end = (uint32_t)(node[6] + node[7]);
            for (uint32_t i = first; i < end; i++) {
                orc_hit th;
                ray_triangle(p, o, d, s->triangles + 9 * (size_t)i, &th);
                if (th.did_hit && th.distance < best) best = th.distance;
            }
        } else {
            out[0]++; out[1] += stale;
            int32_t ca = (int32_t)node[8], cb = ca + 1;
            float da = ray_box(o, d, s->bvh + 9 * (size_t)ca), db = ray_box(o, d, s->bvh + 9 * (size_t)cb);
            float dn, df; int32_t cn, cf;
            if (da < db) { dn = da; df = db; cn = ca; cf = cb; } else { dn = db; df = da; cn = cb; cf = ca; }
            if (df < best) { node_stack[sp] = cf; dst_stack[sp++] = df; }
            if (dn < best) { node_stack[sp] = cn; dst_stack[sp++] = dn; }
        }
    }
}

int sim_stale(const orc_params *p, const orc_scene *s, const float cam[3], const float rot[3],
              uint32_t x0, uint32_t y0, uint32_t w, uint32_t h, uint32_t frame, uint64_t out[4]) {
    memset(out, 0, 4 * sizeof(uint64_t));
    g_recording = 0;
    for (uint32_t y = y0; y < y0 + h; y++) for (uint32_t x = x0; x < x0 + w; x++) {
        float o[3], d[3]; uint32_t pixel_index;
        orc_primary_ray(p, cam, rot, x, y, o, d, &pixel_index);
        for (uint32_t smp = 0; smp < p->rays_per_pixel; smp++) {
            uint32_t seed = pixel_index + smp * 248135u;
            orc_v3 ro = from3(o), rd = from3(d);
            for (uint32_t i = 0; i <= p->max_bounce_count; i++) {
                orc_hit hit;
                float o3[3] = {ro.x, ro.y, ro.z}, d3[3] = {rd.x, rd.y, rd.z};
                if (i > 0) count_stale(p, s, ro, rd, out);
                orc_calculate_ray_collision(p, s, o3, d3, &hit, NULL);
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
    return 0;
}

/* histogram of bounce casts per path (bin 15 = 15 or more) and of node visits per cast (bins of 16 visits) */
void sim_histograms(uint64_t casts_per_path[16], uint64_t visits_per_cast[32]) {
    memset(casts_per_path, 0, 16 * sizeof(uint64_t));
    memset(visits_per_cast, 0, 32 * sizeof(uint64_t));
    for (size_t i = 0; i < g_npath; i++) casts_per_path[g_paths[i].n_casts < 15 ? g_paths[i].n_casts : 15]++;
    for (size_t i = 0; i < g_ncast; i++) { uint32_t b = g_casts[i].n_visits / 16; visits_per_cast[b < 31 ? b : 31]++; }
}

void sim_sizes(uint64_t out[3]) { out[0] = g_npath; out[1] = g_ncast; out[2] = g_nvis; }

typedef struct {
    int waves;            /* waves sharing the path queue */
    int slots;            /* 1: one path per lane; 2: a second, parked path per lane */
    int refill_min;       /* 1 slot: lanes waiting for shading before the wave leaves the traversal loop
                             2 slots: parked paths waiting for shading before a shading round runs */
    int swap_min;         /* 2 slots: idle lanes (with a ready parked path) before a swap round runs */
    int leaf_min, leaf_shift;
    int leaf_period;      /* > 0: leaves run in every leaf_period-th step only (leaf_min/leaf_shift unused) */
    int steps_per_check;  /* steps between two looks at the swap / refill conditions */
    double cost_inner, cost_leaf, cost_shade, cost_swap;
} sim_policy;

typedef struct { double time, steps, inner_lane_steps, leaf_runs, leaf_lane_steps, shade_rounds, shade_lanes, swap_rounds, swap_lanes; } sim_result;

/* per lane: the path in the traversal registers (a) and, with 2 slots, the parked one (b) */
typedef struct {
    int a_path, a_cast, a_pos;     /* a_path < 0: none.  a_pos == n_visits: cast complete (waiting for shading) */
    int b_path, b_cast, b_ready;   /* b_path < 0: none.  b_ready: has a ray that has not been cast; else waits for shading */
} lane_t;

static int next_path(size_t *head) { if (*head >= g_npath) return -1; return (int)(*head)++; }

/* shading of one finished cast: returns 1 if the path continues with another cast */
static int advance_cast(int path, int *cast) {
    if (*cast + 1 < (int)g_paths[path].n_casts) { (*cast)++; return 1; }
    return 0;
}

int sim_run(const sim_policy *pol, sim_result *out) {
    memset(out, 0, sizeof(*out));
    const int W = pol->waves;
    lane_t *L = calloc((size_t)W * 64, sizeof(lane_t));
    double *wave_time = calloc((size_t)W, sizeof(double));
    char *wave_done = calloc((size_t)W, 1);
    size_t head = 0;
    for (int i = 0; i < W * 64; i++) { L[i].a_path = -1; L[i].b_path = -1; }
    int alive = W;
    /* waves advance in order of their local time, so the shared queue is drained as on the device */
    while (alive) {
        int w = -1;
        for (int k = 0; k < W; k++) if (!wave_done[k] && (w < 0 || wave_time[k] < wave_time[w])) w = k;
        lane_t *l = L + (size_t)w * 64;
        /* ---- shading / refill round ---- */
        int shade_lanes = 0;
        for (int i = 0; i < 64; i++) {
            /* a paths with a complete cast (one path per lane only: with a parked path the lane swaps instead) */
            if (pol->slots == 1 && l[i].a_path >= 0 && l[i].a_pos == (int)g_casts[g_paths[l[i].a_path].first_cast + l[i].a_cast].n_visits) {
                shade_lanes++;
                if (!advance_cast(l[i].a_path, &l[i].a_cast)) l[i].a_path = -1;
                l[i].a_pos = 0;
            }
            if (pol->slots == 2 && l[i].b_path >= 0 && !l[i].b_ready) {
                shade_lanes++;
                if (!advance_cast(l[i].b_path, &l[i].b_cast)) l[i].b_path = -1; else l[i].b_ready = 1;
            }
            /* paths without any bounce cast are finished at refill time */
            if (pol->slots == 1) while (l[i].a_path < 0) { int p = next_path(&head); if (p < 0) break; if (g_paths[p].n_casts) { l[i].a_path = p; l[i].a_cast = 0; l[i].a_pos = 0; } }
            if (pol->slots == 2) while (l[i].b_path < 0) { int p = next_path(&head); if (p < 0) break; if (g_paths[p].n_casts) { l[i].b_path = p; l[i].b_cast = 0; l[i].b_ready = 1; } }
        }
        out->shade_rounds++; out->shade_lanes += shade_lanes;
        wave_time[w] += pol->cost_shade;
        /* ---- traversal loop ---- */
        for (int step_no = 0;; step_no++) {
            int trav = 0, waiting_a = 0, waiting_b = 0, swappable = 0, n_inner = 0, n_leaf = 0;
            for (int i = 0; i < 64; i++) {
                int a_done = 1;
                if (l[i].a_path >= 0) {
                    const cast_t *c = &g_casts[g_paths[l[i].a_path].first_cast + l[i].a_cast];
                    if (l[i].a_pos < (int)c->n_visits) { a_done = 0; trav++; if (g_visits[c->first_visit + l[i].a_pos]) n_leaf++; else n_inner++; }
                    else waiting_a++;
                }
                if (pol->slots == 2) {
                    if (l[i].b_path >= 0) { if (!l[i].b_ready) waiting_b++; else if (a_done) swappable++; }
                    else if (a_done && l[i].a_path >= 0) swappable++;      /* nothing parked: the finished cast moves over */
                }
            }
            const int check = pol->steps_per_check <= 1 || step_no % pol->steps_per_check == 0;
            if (check) {
                if (pol->slots == 1) {
                    if (!trav || waiting_a >= pol->refill_min) break;
                } else {
                    if (swappable >= pol->swap_min || (!trav && swappable)) {
                        /* swap round: idle lanes exchange their finished (or absent) path with the ready parked one */
                        for (int i = 0; i < 64; i++) {
                            if (l[i].b_path >= 0 && !l[i].b_ready) continue;
                            if (l[i].b_path < 0 && l[i].a_path < 0) continue;
                            int a_done = 1;
                            if (l[i].a_path >= 0) {
                                const cast_t *c = &g_casts[g_paths[l[i].a_path].first_cast + l[i].a_cast];
                                if (l[i].a_pos < (int)c->n_visits) a_done = 0;
                            }
                            if (!a_done) continue;
                            int ap = l[i].a_path, ac = l[i].a_cast;
                            l[i].a_path = l[i].b_path; l[i].a_cast = l[i].b_cast; l[i].a_pos = 0;
                            l[i].b_path = ap; l[i].b_cast = ac; l[i].b_ready = 0;      /* parked, waits for shading (or empty) */
                            out->swap_lanes++;
                        }
                        out->swap_rounds++;
                        wave_time[w] += pol->cost_swap;
                        step_no = -1;
                        continue;
                    }
                    /* lanes that cannot proceed: parked path waits for shading, or nothing parked and the queue has work */
                    int stuck = 0;
                    for (int i = 0; i < 64; i++) {
                        int b_wait = l[i].b_path >= 0 ? !l[i].b_ready : (head < g_npath);
                        if (b_wait) stuck++;
                    }
                    if (!trav || stuck >= pol->refill_min) break;
                }
            } else if (!trav) {
                wave_time[w] += 0.3 * pol->cost_inner;     /* an empty step until the next check */
                continue;
            }
            /* one wave step */
            int run_leaves;
            if (pol->leaf_period > 0) run_leaves = (step_no % pol->leaf_period) == pol->leaf_period - 1;
            else run_leaves = !n_inner || n_leaf >= pol->leaf_min || (n_leaf << pol->leaf_shift) >= n_leaf + n_inner;
            for (int i = 0; i < 64; i++) {
                if (l[i].a_path < 0) continue;
                const cast_t *c = &g_casts[g_paths[l[i].a_path].first_cast + l[i].a_cast];
                if (l[i].a_pos >= (int)c->n_visits) continue;
                int leaf = g_visits[c->first_visit + l[i].a_pos];
                if (!leaf || run_leaves) l[i].a_pos++;
            }
            out->steps++; out->inner_lane_steps += n_inner;
            wave_time[w] += n_inner ? pol->cost_inner : 0.3 * pol->cost_inner;
            if (run_leaves && n_leaf) { out->leaf_runs++; out->leaf_lane_steps += n_leaf; wave_time[w] += pol->cost_leaf; }
        }
        /* finished? */
        int any = 0;
        for (int i = 0; i < 64; i++) if (l[i].a_path >= 0 || (pol->slots == 2 && l[i].b_path >= 0)) any = 1;
        if (!any && head >= g_npath) { wave_done[w] = 1; alive--; }
    }
    for (int k = 0; k < W; k++) if (wave_time[k] > out->time) out->time = wave_time[k];
    double sum = 0; for (int k = 0; k < W; k++) sum += wave_time[k];
    out->time = sum / W;       /* mean wave time: the tail of a short simulated queue is not representative */
    free(L); free(wave_time); free(wave_done);
    return 0;
}
