/* Scheduling model of a block-level schedule of the path kernel (analysis tool, not part of the product or
 * the tests).  Same recorded visit sequences as lane_sim.c; two more models:
 *
 * sim_async()   traversal waves whose lanes only walk the tree.  A lane owns `buffers` paths; while one is being
 *               traversed the others are with a shading wave of the same block, which turns a finished cast into
 *               the path's next ray (or into a new path) `latency` steps after it was handed over.  Lanes hand
 *               over / pick up at service points, every `service` steps.  Lanes are independent of each other in
 *               this model (leaf turns and service points come at fixed steps), so the result is the share of
 *               lane-steps that do an inner visit.
 * sim_packed()  the block keeps `slots` paths for `waves` x 64 lanes; in every step the paths that stand at an
 *               inner node are packed into as few waves as possible, column by column (a path stays in its
 *               column = lane index, so its stack column in LDS needs no move): wave k of the block runs lane l
 *               if at least k+1 of the paths of column l have inner work.  A wave runs a step only if at least
 *               `min_lanes` of its lanes have work.  Finished casts are shaded `latency` steps later.
 */
#include "lane_sim.c"

typedef struct {
    int buffers, latency, service, leaf_period, leaf_first;
    int slots_per_column, waves, min_lanes;
} block_policy;

typedef struct { double lane_steps, inner_lane_steps, leaf_lane_steps, idle_no_ray, wave_steps, casts; } block_result;

typedef struct { int path, cast, pos; long ready_at; } buf_t;      /* path < 0: none.  ready_at: step at which its next ray is ready */

/* next ray of buffer b: the path's next cast, or the first cast of a new path; 0 = queue empty and path over */
static int next_ray(buf_t *b, size_t *head) {
    if (b->path >= 0 && advance_cast(b->path, &b->cast)) { b->pos = 0; return 1; }
    for (;;) {
        int p = next_path(head);
        if (p < 0) { b->path = -1; return 0; }
        if (g_paths[p].n_casts) { b->path = p; b->cast = 0; b->pos = 0; return 1; }
    }
}

int sim_async(const block_policy *pol, int lanes, block_result *out) {
    memset(out, 0, sizeof(*out));
    size_t head = 0;
    const int B = pol->buffers;
    buf_t *buf = calloc((size_t)lanes * B, sizeof(buf_t));
    int *active = calloc((size_t)lanes, sizeof(int));        /* buffer being traversed, or -1 */
    for (int i = 0; i < lanes; i++) {
        active[i] = -1;
        for (int k = 0; k < B; k++) { buf_t *b = &buf[i * B + k]; b->path = -1; next_ray(b, &head); b->ready_at = 0; }
    }
    /* the measurement stops when the queue runs dry: the tail is another matter */
    for (long t = 0; head < g_npath; t++) {
        const int service = (t % pol->service) == 0;
        const int leaf_turn = (t % pol->leaf_period) == pol->leaf_period - 1;
        for (int i = 0; i < lanes; i++) {
            buf_t *bb = &buf[i * B];
            if (active[i] >= 0) {
                buf_t *b = &bb[active[i]];
                const cast_t *c = &g_casts[g_paths[b->path].first_cast + b->cast];
                if (b->pos >= (int)c->n_visits) {
                    if (service) {           /* hand the finished cast over */
                        out->casts++;
                        if (next_ray(b, &head)) b->ready_at = t + pol->latency; else b->ready_at = -1;
                        active[i] = -1;
                    }
                }
            }
            if (active[i] < 0 && service) {
                for (int k = 0; k < B; k++) if (bb[k].path >= 0 && bb[k].ready_at >= 0 && bb[k].ready_at <= t) { active[i] = k; break; }
            }
            out->lane_steps++;
            if (active[i] < 0) { out->idle_no_ray++; continue; }
            buf_t *b = &bb[active[i]];
            const cast_t *c = &g_casts[g_paths[b->path].first_cast + b->cast];
            if (b->pos >= (int)c->n_visits) continue;      /* waits for the service point */
            int leaf = g_visits[c->first_visit + b->pos];
            if (leaf) {
                if (!leaf_turn) continue;
                b->pos++; out->leaf_lane_steps++;
                if (!pol->leaf_first) continue;
                /* leaves first: the node popped after the leaf is visited in the same step if it is an inner one */
                if (b->pos >= (int)c->n_visits || g_visits[c->first_visit + b->pos]) continue;
            }
            b->pos++; out->inner_lane_steps++;
        }
    }
    free(buf); free(active);
    return 0;
}

/* paths of a block, column-wise: slot s of column l */
int sim_packed(const block_policy *pol, int blocks, block_result *out) {
    memset(out, 0, sizeof(*out));
    size_t head = 0;
    const int S = pol->slots_per_column, W = pol->waves;
    const int n = blocks * 64 * S;
    buf_t *buf = calloc((size_t)n, sizeof(buf_t));
    for (int i = 0; i < n; i++) { buf[i].path = -1; next_ray(&buf[i], &head); buf[i].ready_at = 0; }
    int *rank = calloc(64, sizeof(int));
    for (long t = 0; head < g_npath; t++) {
        const int leaf_turn = (t % pol->leaf_period) == pol->leaf_period - 1;
        for (int bk = 0; bk < blocks; bk++) {
            buf_t *bb = &buf[(size_t)bk * 64 * S];
            /* lanes with inner work per wave: wave k gets column l if the column has more than k paths at an inner node */
            int per_wave[16] = {0};
            for (int l = 0; l < 64; l++) {
                int cnt = 0;
                for (int s = 0; s < S; s++) {
                    buf_t *b = &bb[l * S + s];
                    if (b->path < 0 || b->ready_at < 0 || b->ready_at > t) continue;
                    const cast_t *c = &g_casts[g_paths[b->path].first_cast + b->cast];
                    if (b->pos < (int)c->n_visits && !g_visits[c->first_visit + b->pos]) cnt++;
                }
                rank[l] = cnt;
                for (int k = 0; k < W && k < cnt; k++) per_wave[k]++;
            }
            int run_waves = 0;
            for (int k = 0; k < W; k++) if (per_wave[k] >= pol->min_lanes || (k == 0 && per_wave[0] > 0)) run_waves = k + 1;
            out->wave_steps += run_waves;
            out->lane_steps += 64.0 * run_waves;
            for (int l = 0; l < 64; l++) {
                int served = 0;
                for (int s = 0; s < S; s++) {
                    buf_t *b = &bb[l * S + s];
                    if (b->path < 0 || b->ready_at < 0 || b->ready_at > t) continue;
                    const cast_t *c = &g_casts[g_paths[b->path].first_cast + b->cast];
                    if (b->pos >= (int)c->n_visits) {           /* finished: to the shading wave */
                        out->casts++;
                        if (next_ray(b, &head)) b->ready_at = t + pol->latency; else b->ready_at = -1;
                        continue;
                    }
                    if (g_visits[c->first_visit + b->pos]) {    /* leaves are taken by other waves, in leaf turns */
                        if (leaf_turn) { b->pos++; out->leaf_lane_steps++; }
                        continue;
                    }
                    if (served < run_waves) { b->pos++; served++; out->inner_lane_steps++; }
                }
            }
        }
    }
    free(buf); free(rank);
    return 0;
}

/*
 * sim_multibuf(): every lane of a wave owns `buffers` paths kept in memory; one of them at a time is walked
 * through the tree in the lane's registers.  Every `service` steps a lane whose cast is complete writes the hit
 * back (the buffer now waits for shading) and takes up one of its buffers that hold a ready ray.  The wave shades
 * when at least `shade_min` lanes have a buffer that waits (or when `starve_min` lanes have nothing to walk and
 * something to shade): one buffer per lane per round, so a round runs with up to 64 lanes busy.  Waves take new
 * paths from one shared queue.  Costs in units of one traversal step.
 */
typedef struct {
    int buffers, service, shade_min, starve_min, leaf_period, leaf_first, waves;
    double cost_step, cost_shade, cost_service;
} multibuf_policy;
typedef struct { double time, steps, inner_lane_steps, shade_rounds, shade_lanes, services, idle_lane_steps; } multibuf_result;

typedef struct { int path, cast, pos, state; } mb_t;      /* state: 0 empty, 1 ready, 2 walking, 3 waits for shading */

int sim_multibuf(const multibuf_policy *pol, multibuf_result *out) {
    memset(out, 0, sizeof(*out));
    const int W = pol->waves, B = pol->buffers;
    mb_t *buf = calloc((size_t)W * 64 * B, sizeof(mb_t));
    int *active = calloc((size_t)W * 64, sizeof(int));
    double *wave_time = calloc((size_t)W, sizeof(double));
    long *wave_step = calloc((size_t)W, sizeof(long));
    size_t head = 0;
    for (int i = 0; i < W * 64; i++) active[i] = -1;
    for (int i = 0; i < W * 64 * B; i++) { buf[i].path = -1; buf[i].state = 3; }       /* everything starts by asking for a path */
    double measured_time = 0; int measuring = 1;
    while (head < g_npath) {
        int w = 0;
        for (int k = 1; k < W; k++) if (wave_time[k] < wave_time[w]) w = k;
        mb_t *wb = buf + (size_t)w * 64 * B;
        int *act = active + (size_t)w * 64;
        /* service point */
        int swaps = 0, waiting_lanes = 0, starving = 0;
        for (int l = 0; l < 64; l++) {
            mb_t *bb = wb + l * B;
            if (act[l] >= 0) {
                mb_t *b = &bb[act[l]];
                const cast_t *c = &g_casts[g_paths[b->path].first_cast + b->cast];
                if (b->pos >= (int)c->n_visits) { b->state = 3; act[l] = -1; swaps++; }
            }
            if (act[l] < 0) for (int k = 0; k < B; k++) if (bb[k].state == 1) { bb[k].state = 2; act[l] = k; swaps++; break; }
            int waits = 0;
            for (int k = 0; k < B; k++) if (bb[k].state == 3) waits = 1;
            waiting_lanes += waits;
            if (act[l] < 0 && waits) starving++;
        }
        if (swaps) { out->services++; wave_time[w] += pol->cost_service; }
        if (waiting_lanes >= pol->shade_min || starving >= pol->starve_min) {
            int n = 0;
            for (int l = 0; l < 64; l++) {
                mb_t *bb = wb + l * B;
                for (int k = 0; k < B; k++) if (bb[k].state == 3) {
                    mb_t *b = &bb[k];
                    n++;
                    if (b->path >= 0 && advance_cast(b->path, &b->cast)) { b->pos = 0; b->state = 1; }
                    else {
                        b->state = 0; b->path = -1;
                        for (;;) { int p = next_path(&head); if (p < 0) break; if (g_paths[p].n_casts) { b->path = p; b->cast = 0; b->pos = 0; b->state = 1; break; } }
                    }
                    break;
                }
            }
            out->shade_rounds++; out->shade_lanes += n;
            wave_time[w] += pol->cost_shade;
            continue;          /* straight to another service point: the fresh rays are taken up */
        }
        for (int s = 0; s < pol->service; s++) {
            const long t = wave_step[w]++;
            const int leaf_turn = (t % pol->leaf_period) == pol->leaf_period - 1;
            int n_inner = 0;
            for (int l = 0; l < 64; l++) {
                if (act[l] < 0) { out->idle_lane_steps++; continue; }
                mb_t *b = &wb[l * B + act[l]];
                const cast_t *c = &g_casts[g_paths[b->path].first_cast + b->cast];
                if (b->pos >= (int)c->n_visits) continue;
                if (g_visits[c->first_visit + b->pos]) {
                    if (!leaf_turn) continue;
                    b->pos++;
                    if (!pol->leaf_first || b->pos >= (int)c->n_visits || g_visits[c->first_visit + b->pos]) continue;
                }
                b->pos++; n_inner++;
            }
            out->steps++; out->inner_lane_steps += n_inner;
            wave_time[w] += pol->cost_step * (leaf_turn ? 1.3 : 1.0);
        }
    }
    (void)measured_time; (void)measuring;
    double sum = 0; for (int k = 0; k < W; k++) sum += wave_time[k];
    out->time = sum / W;
    free(buf); free(active); free(wave_time); free(wave_step);
    return 0;
}
