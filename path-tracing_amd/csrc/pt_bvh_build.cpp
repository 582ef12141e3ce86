/*
 * Host-side BVH builder: produces the .bin / _bvh.bin pair the renderer loads.
 *
 * Behaviour follows the reference tool prepare_data/src/main.rs:14-136 exactly
 * (the output must be bit-identical, node order and triangle order included):
 *   - root bounds over all triangles (:16-20);
 *   - per node, 3 axes x 10 candidate planes at t = (i+1)/11 (:33-57), triangle
 *     on side A iff centre[axis] < pos with centre = (a+b+c)/3 (:67,
 *     triangle.rs:19), cost = half-area x count per side (:28-31, :76-77),
 *     strict `<` keeps the first best;
 *   - stop at max_depth (:88) or when cost >= parent cost (:96);
 *   - in-place partition with the reference's swap sequence (:103-122), which
 *     fixes the triangle order inside every leaf;
 *   - children appended as a consecutive pair, A subtree numbered before B
 *     (:124-135).
 *
 * Unlike the reference, which re-scans a node's triangles once per candidate
 * plane (30 scans), the candidates of one axis are evaluated from a single
 * binning pass: plane positions are non-decreasing in i, so a triangle whose
 * centre first falls below plane j is on side A for every plane >= j.  Bounds
 * are merged by prefix/suffix min/max and counts by prefix sums, which are the
 * same values the per-plane scans produce (min/max and integer adds are exact).
 */
#include "../../include/pt_api.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <new>
#include <vector>

namespace {

struct Box { float mn[3], mx[3]; };
struct Tri { float a[3], b[3], c[3], center[3], mn[3], mx[3]; };
struct Node { Box bounds; int32_t triangle_index, triangle_count, child_index; };

inline float fmin_nan(float a, float b) { return (b != b) ? a : ((a != a) ? b : (a < b ? a : b)); }
inline float fmax_nan(float a, float b) { return (b != b) ? a : ((a != a) ? b : (a > b ? a : b)); }

inline void box_reset(Box &b) {
    const float inf = std::numeric_limits<float>::infinity();
    for (int k = 0; k < 3; k++) { b.mn[k] = inf; b.mx[k] = -inf; }
}
inline void box_grow(Box &b, const Tri &t) {            /* bounding_box.rs:34-44 */
    for (int k = 0; k < 3; k++) {
        if (t.mn[k] < b.mn[k]) b.mn[k] = t.mn[k];
        if (t.mx[k] > b.mx[k]) b.mx[k] = t.mx[k];
    }
}
inline void box_merge(Box &b, const Box &o) {
    for (int k = 0; k < 3; k++) {
        if (o.mn[k] < b.mn[k]) b.mn[k] = o.mn[k];
        if (o.mx[k] > b.mx[k]) b.mx[k] = o.mx[k];
    }
}
inline float node_cost(const Box &b, float n) {          /* main.rs:28-31 */
    float sx = b.mx[0] - b.mn[0], sy = b.mx[1] - b.mn[1], sz = b.mx[2] - b.mn[2];
    float half_area = sx * (sy + sz) + sy * sz;
    return half_area * n;
}

struct Builder {
    std::vector<Node> nodes;
    std::vector<Tri> tris;
    int max_depth;

    void choose_split(const Node &node, int &best_axis, float &best_pos, float &best_cost) const {
        constexpr int N = 10;                              /* num_tests_per_axis */
        best_cost = std::numeric_limits<float>::infinity();
        best_pos = 0.0f;
        best_axis = 0;
        const int32_t first = node.triangle_index, last = first + node.triangle_count;
        for (int axis = 0; axis < 3; axis++) {
            const float start = node.bounds.mn[axis], end = node.bounds.mx[axis];
            float pos[N];
            bool monotone = true;
            for (int i = 0; i < N; i++) {
                float split_t = (float)(i + 1) / (float)(N + 1);
                pos[i] = start + (end - start) * split_t;
                if (i && !(pos[i] >= pos[i - 1])) monotone = false;
            }
            float cost[N];
            if (monotone) {
                Box bb[N + 1]; int32_t cnt[N + 1];
                for (int j = 0; j <= N; j++) { box_reset(bb[j]); cnt[j] = 0; }
                for (int32_t t = first; t < last; t++) {
                    const Tri &tr = tris[t];
                    const float c = tr.center[axis];
                    int j = 0;
                    while (j < N && !(c < pos[j])) j++;   /* first plane the centre is below */
                    box_grow(bb[j], tr);
                    cnt[j]++;
                }
                Box pre[N]; int32_t npre[N];               /* side A of plane i = buckets 0..i */
                Box run; box_reset(run); int32_t nrun = 0;
                for (int i = 0; i < N; i++) { box_merge(run, bb[i]); nrun += cnt[i]; pre[i] = run; npre[i] = nrun; }
                box_reset(run); nrun = 0;                  /* side B of plane i = buckets i+1..N */
                for (int i = N - 1; i >= 0; i--) {
                    box_merge(run, bb[i + 1]); nrun += cnt[i + 1];
                    cost[i] = node_cost(pre[i], (float)npre[i]) + node_cost(run, (float)nrun);
                }
            } else {
                /* NaN / inverted bounds: fall back to one scan per plane, like the reference */
                for (int i = 0; i < N; i++) {
                    Box ba, bb2; box_reset(ba); box_reset(bb2); int32_t na = 0, nb = 0;
                    for (int32_t t = first; t < last; t++) {
                        if (tris[t].center[axis] < pos[i]) { box_grow(ba, tris[t]); na++; }
                        else { box_grow(bb2, tris[t]); nb++; }
                    }
                    cost[i] = node_cost(ba, (float)na) + node_cost(bb2, (float)nb);
                }
            }
            for (int i = 0; i < N; i++) {
                if (cost[i] < best_cost) { best_cost = cost[i]; best_pos = pos[i]; best_axis = axis; }
            }
        }
    }

    void split(Node parent, size_t parent_index, int depth) {
        if (depth == max_depth) return;
        int axis; float pos, cost;
        choose_split(parent, axis, pos, cost);
        if (cost >= node_cost(parent.bounds, (float)parent.triangle_count)) return;

        Node a, b;
        box_reset(a.bounds); box_reset(b.bounds);
        a.triangle_index = b.triangle_index = parent.triangle_index;
        a.triangle_count = b.triangle_count = 0;
        a.child_index = b.child_index = 0;
        for (int32_t i = parent.triangle_index; i < parent.triangle_index + parent.triangle_count; i++) {
            if (tris[i].center[axis] < pos) {
                box_grow(a.bounds, tris[i]);
                a.triangle_count++;
                int32_t swap = a.triangle_index + a.triangle_count - 1;
                if (swap != i) { Tri tmp = tris[i]; tris[i] = tris[swap]; tris[swap] = tmp; }
                b.triangle_index++;
            } else {
                box_grow(b.bounds, tris[i]);
                b.triangle_count++;
            }
        }
        if (a.triangle_count > 0 && b.triangle_count > 0) {
            nodes.push_back(a);
            nodes.push_back(b);
            size_t ia = nodes.size() - 2, ib = nodes.size() - 1;
            nodes[parent_index].child_index = (int32_t)ia;
            split(a, ia, depth + 1);
            split(b, ib, depth + 1);
        }
    }
};

}  // namespace

extern "C" int pt_bvh_build(const float *triangles_in, uint32_t triangle_count, int max_depth,
                            float *triangles_out, float **nodes_out, uint32_t *node_count) {
    if (!nodes_out || !node_count || max_depth < 0 || max_depth > 1024) return PT_EINVAL;
    if (triangle_count && (!triangles_in || !triangles_out)) return PT_EINVAL;
    if (triangle_count > (1u << 24)) return PT_EINVAL;     /* indices are stored as f32 in the node file */
    try {
        Builder B;
        B.max_depth = max_depth;
        B.tris.resize(triangle_count);
        for (uint32_t i = 0; i < triangle_count; i++) {     /* triangle.rs:14-31 */
            Tri &t = B.tris[i];
            const float *p = triangles_in + 9 * (size_t)i;
            for (int k = 0; k < 3; k++) {
                t.a[k] = p[k]; t.b[k] = p[3 + k]; t.c[k] = p[6 + k];
                t.center[k] = ((t.a[k] + t.b[k]) + t.c[k]) / 3.0f;
                t.mn[k] = fmin_nan(fmin_nan(t.a[k], t.b[k]), t.c[k]);
                t.mx[k] = fmax_nan(fmax_nan(t.a[k], t.b[k]), t.c[k]);
            }
        }
        Node root;
        box_reset(root.bounds);
        for (uint32_t i = 0; i < triangle_count; i++) box_grow(root.bounds, B.tris[i]);
        root.triangle_index = 0; root.triangle_count = (int32_t)triangle_count; root.child_index = 0;
        B.nodes.reserve(2 * (size_t)triangle_count + 1);
        B.nodes.push_back(root);
        B.split(root, 0, 0);

        float *out = (float *)malloc(B.nodes.size() * 9 * sizeof(float));
        if (!out) return PT_ENOMEM;
        for (size_t i = 0; i < B.nodes.size(); i++) {        /* main.rs:181-192 */
            const Node &n = B.nodes[i];
            float *o = out + 9 * i;
            o[0] = n.bounds.mn[0]; o[1] = n.bounds.mn[1]; o[2] = n.bounds.mn[2];
            o[3] = n.bounds.mx[0]; o[4] = n.bounds.mx[1]; o[5] = n.bounds.mx[2];
            o[6] = (float)n.triangle_index; o[7] = (float)n.triangle_count; o[8] = (float)n.child_index;
        }
        for (uint32_t i = 0; i < triangle_count; i++) {      /* main.rs:194-205 */
            float *o = triangles_out + 9 * (size_t)i;
            memcpy(o, B.tris[i].a, 12); memcpy(o + 3, B.tris[i].b, 12); memcpy(o + 6, B.tris[i].c, 12);
        }
        *nodes_out = out;
        *node_count = (uint32_t)B.nodes.size();
        return PT_OK;
    } catch (const std::bad_alloc &) {
        return PT_ENOMEM;
    }
}

extern "C" void pt_free(void *p) { free(p); }
