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
