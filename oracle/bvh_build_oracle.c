/*
 * CPU oracle for the offline BVH builder.  TEST INFRASTRUCTURE ONLY.
 *
 * Restatement in C of the reference builder prepare_data/src/main.rs:14-136
 * (with node.rs, bounding_box.rs:17-44, triangle.rs:13-31, vector.rs) in
 * strict IEEE binary32.  PINNED: running it on the reference's own
 * objects/teapot.bin with max_depth 16 reproduces objects/teapot.bin and
 * objects/teapot_bvh.bin bit for bit (tests/test_bvh_builder.py).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float v[3]; } bv3;
typedef struct { bv3 min, max; } bbox;                 /* bounding_box.rs:5-9 (center is never read) */
typedef struct { float a[3], b[3], c[3]; bv3 center, min, max; } btri;   /* triangle.rs:4-11 */
typedef struct { bbox bounds; int32_t triangle_index, triangle_count, child_index; } bnode; /* node.rs:6-11 */

typedef struct {
    bnode *nodes; size_t n_nodes, cap_nodes;
    btri *tris; size_t n_tris;
    int max_depth;
} builder;

/* f32::min / f32::max of Rust: a NaN operand is ignored (triangle.rs:20-29) */
static float rmin(float a, float b) { return (b != b) ? a : ((a != a) ? b : (a < b ? a : b)); }
static float rmax(float a, float b) { return (b != b) ? a : ((a != a) ? b : (a > b ? a : b)); }

static void bbox_new(bbox *b) {                        /* bounding_box.rs:18-26 */
    for (int k = 0; k < 3; k++) { b->min.v[k] = INFINITY; b->max.v[k] = -INFINITY; }
}

static void grow_to_include(bbox *b, const btri *t) {  /* bounding_box.rs:34-44 */
    for (int k = 0; k < 3; k++) {
        b->min.v[k] = (t->min.v[k] < b->min.v[k]) ? t->min.v[k] : b->min.v[k];
        b->max.v[k] = (t->max.v[k] > b->max.v[k]) ? t->max.v[k] : b->max.v[k];
    }
}

static void tri_new(btri *t, const float *p) {         /* triangle.rs:14-31 */
    memcpy(t->a, p, 12); memcpy(t->b, p + 3, 12); memcpy(t->c, p + 6, 12);
    for (int k = 0; k < 3; k++) {
        t->center.v[k] = ((t->a[k] + t->b[k]) + t->c[k]) / 3.0f;
        t->min.v[k] = rmin(rmin(t->a[k], t->b[k]), t->c[k]);
        t->max.v[k] = rmax(rmax(t->a[k], t->b[k]), t->c[k]);
    }
}

static float node_cost(const bbox *b, float num_triangles) {   /* main.rs:28-31, size = max - min */
    float sx = b->max.v[0] - b->min.v[0], sy = b->max.v[1] - b->min.v[1], sz = b->max.v[2] - b->min.v[2];
    float half_area = sx * (sy + sz) + sy * sz;
    return half_area * num_triangles;
}

static float evaluate_split(const builder *B, const bnode *node, int axis, float pos) { /* main.rs:59-78 */
    bbox ba, bb; bbox_new(&ba); bbox_new(&bb);
    int32_t na = 0, nb = 0;
    for (int32_t i = node->triangle_index; i < node->triangle_index + node->triangle_count; i++) {
        const btri *t = &B->tris[i];
        if (t->center.v[axis] < pos) { grow_to_include(&ba, t); na++; }
        else { grow_to_include(&bb, t); nb++; }
    }
    return node_cost(&ba, (float)na) + node_cost(&bb, (float)nb);
}

static void choose_split(const builder *B, const bnode *node, int *axis_out, float *pos_out, float *cost_out) { /* main.rs:33-57 */
    const int num_tests_per_axis = 10;
    float best_cost = INFINITY, best_pos = 0.0f; int best_axis = 0;
    for (int axis = 0; axis < 3; axis++) {
        float bounds_start = node->bounds.min.v[axis];
        float bounds_end = node->bounds.max.v[axis];
        for (int i = 0; i < num_tests_per_axis; i++) {
            float split_t = (float)(i + 1) / (float)(num_tests_per_axis + 1);
            float pos = bounds_start + (bounds_end - bounds_start) * split_t;
            float cost = evaluate_split(B, node, axis, pos);
            if (cost < best_cost) { best_cost = cost; best_pos = pos; best_axis = axis; }
        }
    }
    *axis_out = best_axis; *pos_out = best_pos; *cost_out = best_cost;
}

static size_t push_node(builder *B, const bnode *n) {
    if (B->n_nodes == B->cap_nodes) {
        B->cap_nodes = B->cap_nodes ? B->cap_nodes * 2 : 1024;
        B->nodes = (bnode *)realloc(B->nodes, B->cap_nodes * sizeof(bnode));
    }
    B->nodes[B->n_nodes] = *n;
    return B->n_nodes++;
}

/* main.rs:80-136.  `parent` is a by-value copy, like the reference's `&mut Node`
 * argument that aliases a local copy rather than the vector element. */
static void split(builder *B, bnode parent, size_t parent_index, int depth) {
    if (depth == B->max_depth) return;
    int split_axis; float split_pos, cost;
    choose_split(B, &parent, &split_axis, &split_pos, &cost);
    if (cost >= node_cost(&parent.bounds, (float)parent.triangle_count)) return;

    bnode child_a, child_b;
    bbox_new(&child_a.bounds); bbox_new(&child_b.bounds);
    child_a.triangle_index = parent.triangle_index; child_a.triangle_count = 0; child_a.child_index = 0;
    child_b.triangle_index = parent.triangle_index; child_b.triangle_count = 0; child_b.child_index = 0;

    for (int32_t i = parent.triangle_index; i < parent.triangle_index + parent.triangle_count; i++) {
        int is_side_a = B->tris[i].center.v[split_axis] < split_pos;
        if (is_side_a) {
            grow_to_include(&child_a.bounds, &B->tris[i]);
            child_a.triangle_count += 1;
            int32_t swap = child_a.triangle_index + child_a.triangle_count - 1;
            btri tmp = B->tris[i]; B->tris[i] = B->tris[swap]; B->tris[swap] = tmp;
            child_b.triangle_index += 1;
        } else {
            grow_to_include(&child_b.bounds, &B->tris[i]);
            child_b.triangle_count += 1;
        }
    }

    if (child_a.triangle_count > 0 && child_b.triangle_count > 0) {
        size_t child_a_index = push_node(B, &child_a);
        size_t child_b_index = push_node(B, &child_b);
        B->nodes[parent_index].child_index = (int32_t)child_a_index;
        split(B, child_a, child_a_index, depth + 1);
        split(B, child_b, child_b_index, depth + 1);
    }
}

/*
 * triangles_in : n_tris*9 floats (the .bin contents)
 * triangles_out: n_tris*9 floats, reordered as the builder leaves them (main.rs:194-205)
 * nodes_out    : malloc'ed n_nodes*9 floats in _bvh.bin layout (main.rs:181-192); free with orc_bvh_free
 * returns the node count, or -1 on allocation failure.
 */
long orc_bvh_build(const float *triangles_in, uint32_t n_tris, int max_depth,
                   float *triangles_out, float **nodes_out) {
    builder B; memset(&B, 0, sizeof(B));
    B.max_depth = max_depth;
    B.n_tris = n_tris;
    B.tris = (btri *)malloc((n_tris ? n_tris : 1) * sizeof(btri));
    if (!B.tris) return -1;
    for (uint32_t i = 0; i < n_tris; i++) tri_new(&B.tris[i], triangles_in + 9 * (size_t)i);

    bnode root; bbox_new(&root.bounds);                 /* main.rs:14-26 */
    for (uint32_t i = 0; i < n_tris; i++) grow_to_include(&root.bounds, &B.tris[i]);
    root.triangle_index = 0; root.triangle_count = (int32_t)n_tris; root.child_index = 0;
    push_node(&B, &root);
    split(&B, root, 0, 0);

    float *out = (float *)malloc(B.n_nodes * 9 * sizeof(float));
    if (!out) { free(B.tris); free(B.nodes); return -1; }
    for (size_t i = 0; i < B.n_nodes; i++) {
        const bnode *n = &B.nodes[i];
        float *o = out + 9 * i;
        o[0] = n->bounds.min.v[0]; o[1] = n->bounds.min.v[1]; o[2] = n->bounds.min.v[2];
        o[3] = n->bounds.max.v[0]; o[4] = n->bounds.max.v[1]; o[5] = n->bounds.max.v[2];
        o[6] = (float)n->triangle_index; o[7] = (float)n->triangle_count; o[8] = (float)n->child_index;
    }
    for (uint32_t i = 0; i < n_tris; i++) {
        float *o = triangles_out + 9 * (size_t)i;
        memcpy(o, B.tris[i].a, 12); memcpy(o + 3, B.tris[i].b, 12); memcpy(o + 6, B.tris[i].c, 12);
    }
    long n = (long)B.n_nodes;
    *nodes_out = out;
    free(B.tris); free(B.nodes);
    return n;
}

void orc_bvh_free(float *nodes) { free(nodes); }
