/*
 * BVH builder on the GPU: pt_bvh_build_gpu produces the same bytes as pt_bvh_build (pt_bvh_build.cpp),
 * which in turn reproduces the reference tool prepare_data/src/main.rs:14-136 + 179-220 bit for bit.
 *
 * The reference recurses depth first and scans a node's triangles 30 times.  Here the tree grows one
 * LEVEL per pass (all nodes of a depth at once), every pass a handful of launches over the whole
 * triangle array:
 *
 *   bin       every triangle of a node that may still split drops its bounds and a count into one of
 *             11 buckets per axis (bucket = first candidate plane its centre lies below, main.rs:33-57,
 *             :67).  Side A of plane i is buckets 0..i, so prefix / suffix merges of the buckets are the
 *             sums the reference's per-plane scans produce: min, max and integer adds are exact in any
 *             order.  Floats are merged with integer atomics on an order-preserving key.
 *   decide    per node: the 30 costs (half-area x count per side, main.rs:28-31, :76-77), the first
 *             minimum (strict <, :51), the stop rule cost >= parent cost (:96); the child boxes are the
 *             merged buckets either side of the chosen plane.
 *   permute   the reference partitions in place (main.rs:103-122): a triangle of side A is swapped with
 *             the first triangle of side B seen so far.  That keeps side A in its order and ROTATES
 *             side B.  The final place of every triangle is computed directly: the k-th triangle of
 *             side A ends at start + k; a triangle of side B sitting at position q is thrown out when
 *             the (q - start)-th A triangle arrives and lands where that one stood, again and again
 *             until it stands behind the A block (`chase`).
 *   children  pairs are appended per level; when the tree is complete the nodes are renumbered into
 *             the reference's order (a pair is appended when its parent is split, subtree A before
 *             subtree B, main.rs:124-135) from per-subtree counts of split nodes.
 *
 * Only an order of triangles (`order`) moves between passes; the 9 floats of a triangle are gathered
 * once at the end.  Inputs with a non-finite coordinate or a negative zero are refused (PT_EINVAL): with
 * those the reference's result depends on the order of its scans (min / max keep the first of two equal
 * zeros, NaN poisons comparisons), which atomics do not reproduce; pt_bvh_build handles them.
 */
#include "../../include/pt_api.h"
#include "pt_scene_build.h"
#include "pt_exact_div.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <mutex>
#include <string>
#include <vector>

#pragma clang fp contract(off)

namespace {

constexpr int NP = 10;                    /* candidate planes per axis (num_tests_per_axis, main.rs:34) */
constexpr int NB = NP + 1;                /* buckets per axis */
constexpr int BIN_WORDS = 3 * NB * 7;     /* per node: axis x bucket x (min xyz, max xyz, count) */
constexpr uint32_t INACTIVE = 0xffffffffu;
constexpr uint32_t SMALL = 8;             /* nodes of at most this many triangles skip the buckets: one thread scans them plane by plane */
constexpr int SLOTS = 32;                 /* nodes a block of the bin kernel merges in LDS before it touches global memory */

thread_local std::string g_error;

struct GNode {                            /* nodes in the order they are created (level by level) */
    float mn[3], mx[3];
    uint32_t start, count;
    uint32_t child;                       /* creation index of child A (B follows), 0: none */
    uint32_t axis; float pos; uint32_t n_a;
    uint32_t permuted;                    /* this pass partitions the node's triangles */
};

struct SplitT { float t[NP]; };           /* (i + 1) / 11 as the reference rounds it (main.rs:37) */

/* order-preserving map float -> uint32 (no NaN, no -0 in the input) */
__device__ __forceinline__ uint32_t fkey(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float kfloat(uint32_t k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}
#define KEY_POS_INF 0xff800000u           /* fkey(+inf) */
#define KEY_NEG_INF 0x007fffffu           /* fkey(-inf) */

__device__ __forceinline__ float fmin2(float a, float b) { return a < b ? a : b; }
__device__ __forceinline__ float fmax2(float a, float b) { return a > b ? a : b; }

/* main.rs:28-31 */
__device__ __forceinline__ float node_cost(const float mn[3], const float mx[3], float n) {
    const float sx = mx[0] - mn[0], sy = mx[1] - mn[1], sz = mx[2] - mn[2];
    const float half_area = sx * (sy + sz) + sy * sz;
    return half_area * n;
}

/* ---- per triangle: centre and bounds (triangle.rs:14-31), root bounds ---- */
__global__ __launch_bounds__(256) void prep_kernel(const float *tri9, uint32_t n, float *ctr, float *tmn, float *tmx,
                                                   uint32_t *order, uint32_t *pos_node, uint32_t *root_keys) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    __shared__ uint32_t s_keys[6];
    if (threadIdx.x < 6) s_keys[threadIdx.x] = threadIdx.x < 3 ? KEY_POS_INF : KEY_NEG_INF;
    __syncthreads();
    if (i < n) {
        const float *p = tri9 + 9 * (size_t)i;
        for (int k = 0; k < 3; k++) {
            const float a = p[k], b = p[3 + k], c = p[6 + k];
            ctr[3 * (size_t)i + k] = ((a + b) + c) / 3.0f;
            const float lo = fmin2(fmin2(a, b), c), hi = fmax2(fmax2(a, b), c);
            tmn[3 * (size_t)i + k] = lo;
            tmx[3 * (size_t)i + k] = hi;
            atomicMin(&s_keys[k], fkey(lo));
            atomicMax(&s_keys[3 + k], fkey(hi));
        }
        order[i] = i;
        pos_node[i] = 0;
    }
    __syncthreads();
    if (threadIdx.x < 3) atomicMin(&root_keys[threadIdx.x], s_keys[threadIdx.x]);
    else if (threadIdx.x < 6) atomicMax(&root_keys[threadIdx.x], s_keys[threadIdx.x]);
}

__global__ void root_kernel(const uint32_t *root_keys, uint32_t n, GNode *nodes) {
    GNode r;
    for (int k = 0; k < 3; k++) { r.mn[k] = kfloat(root_keys[k]); r.mx[k] = kfloat(root_keys[3 + k]); }
    r.start = 0; r.count = n; r.child = 0; r.axis = 0; r.pos = 0.0f; r.n_a = 0; r.permuted = 0;
    nodes[0] = r;
}

__global__ __launch_bounds__(256) void bins_init_kernel(uint32_t *bins, size_t words) {
    const size_t w = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= words) return;
    const uint32_t f = (uint32_t)(w % 7);
    bins[w] = f < 3 ? KEY_POS_INF : (f < 6 ? KEY_NEG_INF : 0u);
}

__device__ __forceinline__ void plane_positions(const GNode &nd, int axis, const SplitT &st, float pos[NP]) {
    const float start = nd.mn[axis], end = nd.mx[axis];
    for (int i = 0; i < NP; i++) pos[i] = start + (end - start) * st.t[i];
}

/* One thread per position of the triangle order.  Nodes own contiguous ranges of positions and, within a
 * level, their numbers rise with the positions, so the (at most SLOTS) nodes a block of 256 positions meets
 * are merged in LDS first and only the touched bucket entries go to global memory as atomics; a node beyond
 * the LDS slots is merged in global memory directly.  Nodes of at most SMALL triangles are left to
 * small_decide_kernel. */
__global__ __launch_bounds__(256) void bin_kernel(const GNode *nodes, uint32_t level_begin, const uint32_t *pos_node,
                                                  const uint32_t *order, const float *ctr, const float *tmn, const float *tmx,
                                                  uint32_t n, SplitT st, uint32_t *bins) {
    __shared__ uint32_t s_bins[SLOTS * BIN_WORDS];
    __shared__ uint32_t s_base, s_top;
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (threadIdx.x == 0) { s_base = INACTIVE; s_top = 0; }
    __syncthreads();
    uint32_t nid = p < n ? pos_node[p] : INACTIVE;
    GNode nd;
    if (nid != INACTIVE) {
        nd = nodes[nid];
        if (nd.count <= SMALL) nid = INACTIVE;
    }
    if (nid != INACTIVE) atomicMin(&s_base, nid);
    __syncthreads();
    const uint32_t base = s_base;
    if (base == INACTIVE) return;                       /* nothing to merge in this block */
    if (nid != INACTIVE && nid - base < (uint32_t)SLOTS) atomicMax(&s_top, nid - base + 1u);
    __syncthreads();
    const uint32_t used = s_top * BIN_WORDS;
    for (uint32_t w = threadIdx.x; w < used; w += blockDim.x) {
        const uint32_t f = w % 7;
        s_bins[w] = f < 3 ? KEY_POS_INF : (f < 6 ? KEY_NEG_INF : 0u);
    }
    __syncthreads();
    if (nid != INACTIVE) {
        const uint32_t t = order[p];
        uint32_t klo[3], khi[3];
        for (int k = 0; k < 3; k++) { klo[k] = fkey(tmn[3 * (size_t)t + k]); khi[k] = fkey(tmx[3 * (size_t)t + k]); }
        uint32_t *dst = nid - base < (uint32_t)SLOTS ? s_bins + (nid - base) * BIN_WORDS : bins + (size_t)(nid - level_begin) * BIN_WORDS;
        for (int axis = 0; axis < 3; axis++) {
            float pos[NP];
            plane_positions(nd, axis, st, pos);
            const float c = ctr[3 * (size_t)t + axis];
            int j = 0;
            while (j < NP && !(c < pos[j])) j++;               /* first plane the centre lies below */
            uint32_t *b = dst + (axis * NB + j) * 7;
            for (int k = 0; k < 3; k++) { atomicMin(&b[k], klo[k]); atomicMax(&b[3 + k], khi[k]); }
            atomicAdd(&b[6], 1u);
        }
    }
    __syncthreads();
    uint32_t *g = bins + (size_t)(base - level_begin) * BIN_WORDS;
    for (uint32_t w = threadIdx.x; w < used; w += blockDim.x) {
        const uint32_t f = w % 7, v = s_bins[w];
        if (f < 3) { if (v != KEY_POS_INF) atomicMin(&g[w], v); }
        else if (f < 6) { if (v != KEY_NEG_INF) atomicMax(&g[w], v); }
        else if (v) atomicAdd(&g[w], v);
    }
}

struct Side { float mn[3], mx[3]; uint32_t n; };
__device__ __forceinline__ void side_reset(Side &s) {
    for (int k = 0; k < 3; k++) { s.mn[k] = INFINITY; s.mx[k] = -INFINITY; }
    s.n = 0;
}
__device__ __forceinline__ void side_merge(Side &s, const uint32_t *b) {
    for (int k = 0; k < 3; k++) { s.mn[k] = fmin2(kfloat(b[k]), s.mn[k]); s.mx[k] = fmax2(kfloat(b[3 + k]), s.mx[k]); }
    s.n += b[6];
}

/* the stop rule and the split that follows from the best candidate, main.rs:88-96 and :124 */
__device__ __forceinline__ void finish_decision(GNode &nd, float best_cost, int best_axis, int best_i, float best_pos,
                                                const Side &sa, const Side &sb, float *cb, uint32_t *has_children, uint32_t *error_flag) {
    const float parent_cost = node_cost(nd.mn, nd.mx, (float)nd.count);
    nd.child = 0; nd.permuted = 0; nd.n_a = 0; nd.axis = 0; nd.pos = 0.0f;
    uint32_t both = 0;
    if (!(best_cost >= parent_cost)) {
        if (best_i < 0) {
            /* no candidate was better than +inf and the parent cost is NaN: the reference then partitions at
             * axis 0, position 0, which is none of the candidate planes */
            atomicOr(error_flag, 1u);
        } else {
            nd.permuted = 1; nd.axis = (uint32_t)best_axis; nd.pos = best_pos; nd.n_a = sa.n;
            both = (sa.n > 0 && sb.n > 0) ? 1u : 0u;           /* main.rs:124 */
            for (int k = 0; k < 3; k++) { cb[k] = sa.mn[k]; cb[3 + k] = sa.mx[k]; cb[6 + k] = sb.mn[k]; cb[9 + k] = sb.mx[k]; }
        }
    }
    *has_children = both;
}

/* choose_split + the stop rule, main.rs:33-57 and :88-96, from the buckets.  One thread per node of the level. */
__global__ __launch_bounds__(64) void decide_kernel(GNode *nodes, uint32_t level_begin, uint32_t level_count, const uint32_t *bins,
                                                    SplitT st, float *child_boxes, uint32_t *has_children, uint32_t *error_flag) {
    const uint32_t a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= level_count) return;
    GNode nd = nodes[level_begin + a];
    if (nd.count <= SMALL) return;                      /* small_decide_kernel's */
    const uint32_t *nb = bins + (size_t)a * BIN_WORDS;
    float best_cost = INFINITY;
    int best_axis = 0, best_i = -1;
    float best_pos = 0.0f;
    for (int axis = 0; axis < 3; axis++) {
        float pos[NP], cost[NP];
        plane_positions(nd, axis, st, pos);
        const uint32_t *ab = nb + axis * NB * 7;
        float cost_a[NP];
        Side run;
        side_reset(run);
        for (int i = 0; i < NP; i++) { side_merge(run, ab + i * 7); cost_a[i] = node_cost(run.mn, run.mx, (float)run.n); }
        side_reset(run);
        for (int i = NP - 1; i >= 0; i--) { side_merge(run, ab + (i + 1) * 7); cost[i] = cost_a[i] + node_cost(run.mn, run.mx, (float)run.n); }
        for (int i = 0; i < NP; i++)
            if (cost[i] < best_cost) { best_cost = cost[i]; best_pos = pos[i]; best_axis = axis; best_i = i; }
    }
    Side sa, sb;
    side_reset(sa); side_reset(sb);
    if (best_i >= 0) {
        const uint32_t *ab = nb + best_axis * NB * 7;
        for (int i = 0; i <= best_i; i++) side_merge(sa, ab + i * 7);
        for (int i = NP; i > best_i; i--) side_merge(sb, ab + i * 7);
    }
    finish_decision(nd, best_cost, best_axis, best_i, best_pos, sa, sb, child_boxes + 12 * (size_t)a, has_children + a, error_flag);
    nodes[level_begin + a] = nd;
}

/* The same decision for nodes of at most SMALL triangles, the way the reference takes it: one scan of the
 * node's triangles per candidate plane (main.rs:59-78), one thread per node, triangles held in registers. */
__global__ __launch_bounds__(64) void small_decide_kernel(GNode *nodes, uint32_t level_begin, uint32_t level_count, const uint32_t *order,
                                                          const float *ctr, const float *tmn, const float *tmx, SplitT st,
                                                          float *child_boxes, uint32_t *has_children, uint32_t *error_flag) {
    const uint32_t a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= level_count) return;
    GNode nd = nodes[level_begin + a];
    if (nd.count > SMALL) return;
    float c[SMALL][3], lo[SMALL][3], hi[SMALL][3];
#pragma unroll
    for (uint32_t k = 0; k < SMALL; k++) {
        if (k < nd.count) {
            const uint32_t t = order[nd.start + k];
            for (int x = 0; x < 3; x++) { c[k][x] = ctr[3 * (size_t)t + x]; lo[k][x] = tmn[3 * (size_t)t + x]; hi[k][x] = tmx[3 * (size_t)t + x]; }
        } else {
            for (int x = 0; x < 3; x++) { c[k][x] = 0.0f; lo[k][x] = INFINITY; hi[k][x] = -INFINITY; }
        }
    }
    float best_cost = INFINITY;
    int best_axis = 0, best_i = -1;
    float best_pos = 0.0f;
    Side best_a, best_b;
    side_reset(best_a); side_reset(best_b);
#pragma unroll
    for (int axis = 0; axis < 3; axis++) {
        float pos[NP];
        plane_positions(nd, axis, st, pos);
        for (int i = 0; i < NP; i++) {
            Side sa, sb;
            side_reset(sa); side_reset(sb);
#pragma unroll
            for (uint32_t k = 0; k < SMALL; k++) {
                if (k < nd.count) {
                    const bool in_a = c[k][axis] < pos[i];
                    for (int x = 0; x < 3; x++) {
                        if (in_a) { sa.mn[x] = fmin2(lo[k][x], sa.mn[x]); sa.mx[x] = fmax2(hi[k][x], sa.mx[x]); }
                        else      { sb.mn[x] = fmin2(lo[k][x], sb.mn[x]); sb.mx[x] = fmax2(hi[k][x], sb.mx[x]); }
                    }
                    if (in_a) sa.n++; else sb.n++;
                }
            }
            const float cost = node_cost(sa.mn, sa.mx, (float)sa.n) + node_cost(sb.mn, sb.mx, (float)sb.n);
            if (cost < best_cost) { best_cost = cost; best_pos = pos[i]; best_axis = axis; best_i = i; best_a = sa; best_b = sb; }
        }
    }
    finish_decision(nd, best_cost, best_axis, best_i, best_pos, best_a, best_b, child_boxes + 12 * (size_t)a, has_children + a, error_flag);
    nodes[level_begin + a] = nd;
}

/* 1 for the triangles that go to side A of their node in this pass (main.rs:106) */
__global__ __launch_bounds__(256) void flag_kernel(const GNode *nodes, const uint32_t *pos_node, const uint32_t *order, const float *ctr,
                                                   uint32_t n, uint32_t *flag) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    uint32_t f = 0;
    const uint32_t nid = pos_node[p];
    if (nid != INACTIVE && nodes[nid].permuted)
        f = ctr[3 * (size_t)order[p] + nodes[nid].axis] < nodes[nid].pos ? 1u : 0u;
    flag[p] = f;
}

/* where the k-th A triangle of each node stood */
__global__ __launch_bounds__(256) void a_index_kernel(const GNode *nodes, const uint32_t *pos_node, const uint32_t *flag, const uint32_t *scan,
                                                      uint32_t n, uint32_t *a_index) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n || !flag[p]) return;
    const uint32_t start = nodes[pos_node[p]].start;
    a_index[start + (scan[p] - scan[start])] = p;
}

/* the reference's in-place partition (main.rs:103-122) as a direct computation of every triangle's final place.
 * A side-B triangle follows a chain of positions (it is thrown out by the A triangle that takes its place, lands where
 * that one stood, and so on until it stands behind the A block).  Chains are a few links long on ordinary meshes, but
 * an order like "one far triangle, then everything else sorted" makes one chain as long as the node: a thread gives up
 * after CHASE_CAP links and leaves the triangle to jump_kernel + chase_rest_kernel.  rest[0] = number of such triangles, then
 * (position, where the chase stopped) pairs. */
#define CHASE_CAP 1024
__global__ __launch_bounds__(256) void permute_kernel(const GNode *nodes, const uint32_t *pos_node, const uint32_t *flag, const uint32_t *scan,
                                                      const uint32_t *a_index, const uint32_t *order, uint32_t n, uint32_t *order_next,
                                                      uint32_t *rest) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const uint32_t nid = pos_node[p];
    uint32_t dest = p;
    if (nid != INACTIVE && nodes[nid].permuted) {
        const uint32_t start = nodes[nid].start, n_a = nodes[nid].n_a;
        if (flag[p]) dest = start + (scan[p] - scan[start]);
        else {
            int links = 0;
            while (dest - start < n_a && links < CHASE_CAP) { dest = a_index[dest]; links++; }      /* thrown out by the (dest - start)-th A triangle */
            if (dest - start < n_a) {
                const uint32_t k = atomicAdd(&rest[0], 1u);
                rest[2 + 2 * (size_t)k] = p; rest[3 + 2 * (size_t)k] = dest;
                return;
            }
        }
    }
    order_next[dest] = order[p];
}

/* The triangles permute_kernel gave up on (usually none: then both kernels below look at a counter and return).
 * Pointer jumping over the chains, in place and without rounds: every A slot's entry is replaced by its successor's
 * until it points behind the A block.  A thread that meets a half-updated entry still gets a position further down
 * the same chain, so no synchronisation is needed, and with all threads at it a chain of n links collapses in about
 * log2(n) replacements per slot.  Afterwards every left-over triangle is a link or two from its place. */
__global__ __launch_bounds__(256) void jump_kernel(const GNode *nodes, const uint32_t *pos_node, uint32_t *a_index, uint32_t n,
                                                   const uint32_t *rest) {
    if (rest[0] == 0) return;
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= n) return;
    const uint32_t nid = pos_node[x];
    if (nid == INACTIVE || !nodes[nid].permuted) return;
    const uint32_t start = nodes[nid].start, n_a = nodes[nid].n_a;
    if (x - start >= n_a) return;
    volatile uint32_t *a = a_index;
    for (int it = 0; it < 64; it++) {
        const uint32_t j = a[x];
        if (j - start >= n_a) break;
        a[x] = a[j];
    }
}

__global__ __launch_bounds__(256) void chase_rest_kernel(const GNode *nodes, const uint32_t *pos_node, const uint32_t *a_index,
                                                         const uint32_t *order, uint32_t *order_next, const uint32_t *rest) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= rest[0]) return;
    const uint32_t p = rest[2 + 2 * (size_t)k];
    uint32_t dest = rest[3 + 2 * (size_t)k];
    const uint32_t nid = pos_node[p];
    const uint32_t start = nodes[nid].start, n_a = nodes[nid].n_a;
    while (dest - start < n_a) dest = a_index[dest];
    order_next[dest] = order[p];
}

__global__ __launch_bounds__(64) void children_kernel(GNode *nodes, uint32_t level_begin, uint32_t level_count, const uint32_t *has_children,
                                                      const uint32_t *child_rank, const float *child_boxes, uint32_t next_begin) {
    const uint32_t a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= level_count || !has_children[a]) return;
    GNode &nd = nodes[level_begin + a];
    const uint32_t ca = next_begin + 2u * child_rank[a];
    const float *cb = child_boxes + 12 * (size_t)a;
    GNode x;
    x.child = 0; x.axis = 0; x.pos = 0.0f; x.n_a = 0; x.permuted = 0;
    for (int k = 0; k < 3; k++) { x.mn[k] = cb[k]; x.mx[k] = cb[3 + k]; }
    x.start = nd.start; x.count = nd.n_a;
    nodes[ca] = x;
    for (int k = 0; k < 3; k++) { x.mn[k] = cb[6 + k]; x.mx[k] = cb[9 + k]; }
    x.start = nd.start + nd.n_a; x.count = nd.count - nd.n_a;
    nodes[ca + 1] = x;
    nd.child = ca;
}

/* positions follow their triangles into the child nodes; positions of finished nodes drop out */
__global__ __launch_bounds__(256) void assign_kernel(const GNode *nodes, uint32_t n, uint32_t *pos_node) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const uint32_t nid = pos_node[p];
    if (nid == INACTIVE) return;
    const uint32_t child = nodes[nid].child;
    pos_node[p] = child ? (p < nodes[nid].start + nodes[nid].n_a ? child : child + 1u) : INACTIVE;
}

/* ---- exclusive prefix sum of 32-bit counts: block scan, scan of the block sums, add ---- */
constexpr int SCAN_ITEMS = 4, SCAN_THREADS = 256, SCAN_BLOCK = SCAN_ITEMS * SCAN_THREADS;

__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t *total) {
    __shared__ uint32_t s_wave[SCAN_THREADS / 64];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t inc = v;
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_up(inc, off, 64);
        if (lane >= (uint32_t)off) inc += o;
    }
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    uint32_t base = 0, sum = 0;
    for (uint32_t w = 0; w < SCAN_THREADS / 64; w++) { if (w < wave) base += s_wave[w]; sum += s_wave[w]; }
    __syncthreads();
    *total = sum;
    return base + inc - v;
}

__global__ __launch_bounds__(SCAN_THREADS) void scan_blocks_kernel(const uint32_t *in, uint32_t n, uint32_t *out, uint32_t *block_sums) {
    const uint32_t base = blockIdx.x * SCAN_BLOCK + threadIdx.x * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS], s = 0;
    for (int k = 0; k < SCAN_ITEMS; k++) { v[k] = base + k < n ? in[base + k] : 0u; s += v[k]; }
    uint32_t total;
    uint32_t ex = block_exclusive_scan(s, &total);
    for (int k = 0; k < SCAN_ITEMS; k++) { if (base + k < n) out[base + k] = ex; ex += v[k]; }
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

/* one block: exclusive scan of the block sums in place, grand total to *total */
__global__ __launch_bounds__(SCAN_THREADS) void scan_sums_kernel(uint32_t *block_sums, uint32_t blocks, uint32_t *total) {
    uint32_t carry = 0;
    for (uint32_t first = 0; first < blocks; first += SCAN_THREADS) {
        const uint32_t i = first + threadIdx.x;
        const uint32_t v = i < blocks ? block_sums[i] : 0u;
        uint32_t chunk;
        const uint32_t ex = block_exclusive_scan(v, &chunk);
        if (i < blocks) block_sums[i] = carry + ex;
        carry += chunk;
    }
    if (threadIdx.x == 0) *total = carry;
}

__global__ __launch_bounds__(SCAN_THREADS) void scan_add_kernel(uint32_t *out, uint32_t n, const uint32_t *block_sums) {
    const uint32_t base = blockIdx.x * SCAN_BLOCK + threadIdx.x * SCAN_ITEMS;
    const uint32_t add = block_sums[blockIdx.x];
    for (int k = 0; k < SCAN_ITEMS; k++) if (base + k < n) out[base + k] += add;
}

/* ---- renumbering into the reference's node order ---- */
/* split nodes in the subtree of every node, one level per launch, deepest level first */
__global__ __launch_bounds__(256) void subtree_kernel(const GNode *nodes, uint32_t begin, uint32_t count, uint32_t *splits) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const uint32_t c = nodes[begin + i].child;
    splits[begin + i] = c ? 1u + splits[c] + splits[c + 1] : 0u;
}

/* rank of a split node among the split nodes in the reference's recursion order -> its pair sits at
 * 1 + 2 * rank (main.rs:124-135); one level per launch, root first */
__global__ __launch_bounds__(256) void number_kernel(const GNode *nodes, uint32_t begin, uint32_t count, const uint32_t *splits,
                                                     uint32_t *rank, uint32_t *file_index) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const uint32_t c = nodes[begin + i].child;
    if (!c) return;
    const uint32_t r = rank[begin + i], pair = 1u + 2u * r;
    file_index[c] = pair; file_index[c + 1] = pair + 1u;
    rank[c] = r + 1u;
    rank[c + 1] = r + 1u + splits[c];
}

/* main.rs:181-192 */
__global__ __launch_bounds__(256) void write_nodes_kernel(const GNode *nodes, uint32_t count, const uint32_t *file_index, float *out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const GNode nd = nodes[i];
    float *o = out + 9 * (size_t)file_index[i];
    for (int k = 0; k < 3; k++) { o[k] = nd.mn[k]; o[3 + k] = nd.mx[k]; }
    o[6] = (float)(int32_t)nd.start; o[7] = (float)(int32_t)nd.count;
    o[8] = (float)(int32_t)(nd.child ? file_index[nd.child] : 0u);
}

/* main.rs:194-205 */
__global__ __launch_bounds__(256) void write_tris_kernel(const float *tri9, const uint32_t *order, uint32_t n, float *out) {
    const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= 9u * n) return;
    out[w] = tri9[9 * (size_t)order[w / 9u] + w % 9u];
}

/* Scratch memory of the builds.  hipMalloc and hipFree of some twenty buffers cost more than the build itself
 * (about 15 ms against 9 ms for 784 k triangles), so freed scratch blocks are kept per device and handed out again
 * to a later build that asks for about the same size: a program that rebuilds every frame allocates once.  At most
 * CACHE_LIMIT bytes are kept; pt_bvh_build_gpu_release_cache() returns them to the driver. */
/* The builder works on the device it is asked for and leaves the calling thread on the device it came with (a
 * caller that shares the thread with another runtime user, e.g. torch, must not find its device changed). */
struct DeviceRestore {
    int prev = -1;
    DeviceRestore() { if (hipGetDevice(&prev) != hipSuccess) { (void)hipGetLastError(); prev = -1; } }
    ~DeviceRestore() { if (prev >= 0) (void)hipSetDevice(prev); }
};

struct CachedBlock { int device; size_t bytes; void *ptr; };
std::mutex g_cache_mutex;
std::vector<CachedBlock> g_cache;
size_t g_cache_bytes = 0;
constexpr size_t CACHE_LIMIT = (size_t)1 << 30;       /* five builds of the 784 k-triangle benchmark mesh */

hipError_t scratch_alloc(int device, size_t bytes, void **out, size_t *got) {
    {
        std::lock_guard<std::mutex> lock(g_cache_mutex);
        size_t best = g_cache.size();
        for (size_t i = 0; i < g_cache.size(); i++)
            if (g_cache[i].device == device && g_cache[i].bytes >= bytes && g_cache[i].bytes <= bytes + bytes / 2 + 4096 &&
                (best == g_cache.size() || g_cache[i].bytes < g_cache[best].bytes)) best = i;
        if (best != g_cache.size()) {
            *out = g_cache[best].ptr; *got = g_cache[best].bytes;
            g_cache_bytes -= g_cache[best].bytes;
            g_cache.erase(g_cache.begin() + (long)best);
            return hipSuccess;
        }
    }
    *got = bytes;
    return hipMalloc(out, bytes);
}

void scratch_free(int device, void *p, size_t bytes) {
    {
        std::lock_guard<std::mutex> lock(g_cache_mutex);
        if (g_cache_bytes + bytes <= CACHE_LIMIT && g_cache.size() < 256) {
            g_cache.push_back(CachedBlock{device, bytes, p});
            g_cache_bytes += bytes;
            return;
        }
    }
    (void)hipFree(p);
}

struct DeviceBuffers {
    int device = 0;
    struct Held { void *ptr; size_t bytes; };
    std::vector<Held> held;
    ~DeviceBuffers() { for (const Held &h : held) scratch_free(device, h.ptr, h.bytes); }
    template <typename T> hipError_t alloc(T *&p, size_t count) {
        void *v = nullptr;
        size_t got = 0;
        hipError_t e = scratch_alloc(device, (count ? count : 1) * sizeof(T), &v, &got);
        if (e == hipSuccess) { held.push_back(Held{v, got}); p = (T *)v; }
        return e;
    }
    void release(void *p) {
        for (size_t i = 0; i < held.size(); i++)
            if (held[i].ptr == p) { scratch_free(device, p, held[i].bytes); held.erase(held.begin() + (long)i); return; }
    }
};

int fail(int code, const std::string &msg) { g_error = msg; return code; }

#define GB_TRY(call)                                                                                          \
    do {                                                                                                      \
        hipError_t e_ = (call);                                                                               \
        if (e_ != hipSuccess) return fail(PT_EHIP, std::string(#call) + ": " + hipGetErrorString(e_));       \
    } while (0)

inline dim3 grid_for(size_t n, unsigned block) { return dim3((unsigned)((n + block - 1) / block)); }

/* out[i] = sum of in[0..i), *d_total = sum of all; d_sums holds ceil(n / SCAN_BLOCK) words */
void exclusive_scan(const uint32_t *in, uint32_t n, uint32_t *out, uint32_t *d_sums, uint32_t *d_total, hipStream_t st) {
    const uint32_t blocks = (n + SCAN_BLOCK - 1) / SCAN_BLOCK;
    hipLaunchKernelGGL(scan_blocks_kernel, dim3(blocks), dim3(SCAN_THREADS), 0, st, in, n, out, d_sums);
    hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(SCAN_THREADS), 0, st, d_sums, blocks, d_total);
    hipLaunchKernelGGL(scan_add_kernel, dim3(blocks), dim3(SCAN_THREADS), 0, st, out, n, d_sums);
}

}  // namespace


extern "C" const char *pt_bvh_build_gpu_error(void) { return g_error.c_str(); }

extern "C" void pt_bvh_build_gpu_release_cache(void) {
    DeviceRestore keep;
    std::lock_guard<std::mutex> lock(g_cache_mutex);
    for (const CachedBlock &b : g_cache) {
        if (hipSetDevice(b.device) == hipSuccess) (void)hipFree(b.ptr);
    }
    g_cache.clear();
    g_cache_bytes = 0;
}

namespace {

int check_triangles(const float *triangles_in, uint32_t n) {
    if (n > (1u << 24)) return fail(PT_EINVAL, "more than 2^24 triangles (indices are stored as f32 in the node file)");
    for (size_t i = 0; i < 9 * (size_t)n; i++) {
        uint32_t u;
        memcpy(&u, triangles_in + i, 4);
        if ((u & 0x7f800000u) == 0x7f800000u || u == 0x80000000u)
            return fail(PT_EINVAL, "a coordinate is not finite or is a negative zero: the reference's result then depends on the order "
                                   "of its scans; use pt_bvh_build");
    }
    return PT_OK;
}

int select_device(int device_ordinal) {
    int dev_count = 0;
    GB_TRY(hipGetDeviceCount(&dev_count));
    if (device_ordinal < 0 || device_ordinal >= dev_count) return fail(PT_EINVAL, "no such device");
    GB_TRY(hipSetDevice(device_ordinal));
    hipDeviceProp_t prop;
    GB_TRY(hipGetDeviceProperties(&prop, device_ordinal));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(PT_EHIP, std::string("this library is built for gfx950 (MI355X); the device is ") + prop.gcnArchName);
    return PT_OK;
}

/* what grow_tree leaves in device memory (owned by the DeviceBuffers it was given) */
struct Tree {
    float *d_tri9 = nullptr;               /* the triangles as given */
    GNode *d_nodes = nullptr;              /* nodes in creation order */
    uint32_t *d_order = nullptr;           /* order[p] = input index of the triangle at position p of the output */
    uint32_t *d_splits = nullptr;          /* split nodes in each node's subtree */
    uint32_t *d_rank = nullptr;            /* of a split node: its number among the split nodes in the reference's recursion order */
    uint32_t *d_file_index = nullptr;      /* the node's index in the node file */
    std::vector<uint32_t> level_begin, level_count;
    uint32_t total_nodes = 0;
};

/* With the device current and n >= 1: uploads the triangles, records ev0, grows the tree level by level and
 * numbers the nodes.  Everything is enqueued on `stream`; the last level's read-back has synchronised it. */
int grow_tree(const float *triangles_in, uint32_t n, int max_depth, hipStream_t stream, hipEvent_t ev0, DeviceBuffers &db, Tree &t) {
    SplitT st;
    for (int i = 0; i < NP; i++) st.t[i] = (float)(i + 1) / (float)(NP + 1);

    const size_t max_nodes = 2 * (size_t)n + 1;
    float *d_tri9 = nullptr, *d_ctr = nullptr, *d_tmn = nullptr, *d_tmx = nullptr, *d_child_boxes = nullptr;
    uint32_t *d_order = nullptr, *d_order_next = nullptr, *d_pos_node = nullptr, *d_flag = nullptr, *d_scan = nullptr, *d_a_index = nullptr;
    uint32_t *d_sums = nullptr, *d_small = nullptr, *d_has = nullptr, *d_rank_in_level = nullptr, *d_bins = nullptr;
    GNode *d_nodes = nullptr;
    GB_TRY(db.alloc(d_tri9, 9 * (size_t)n));
    GB_TRY(db.alloc(d_ctr, 3 * (size_t)n));
    GB_TRY(db.alloc(d_tmn, 3 * (size_t)n));
    GB_TRY(db.alloc(d_tmx, 3 * (size_t)n));
    GB_TRY(db.alloc(d_order, n));
    GB_TRY(db.alloc(d_order_next, n));
    GB_TRY(db.alloc(d_pos_node, n));
    GB_TRY(db.alloc(d_flag, n));
    GB_TRY(db.alloc(d_scan, n));
    GB_TRY(db.alloc(d_a_index, n));
    uint32_t *d_rest = nullptr;            /* triangles whose chase permute_kernel gave up: a counter and (position, stop) pairs */
    GB_TRY(db.alloc(d_rest, 2 * (size_t)n + 2));
    GB_TRY(hipMemsetAsync(d_rest, 0, 2 * sizeof(uint32_t), stream));
    GB_TRY(db.alloc(d_sums, (max_nodes + SCAN_BLOCK - 1) / SCAN_BLOCK + 1));
    GB_TRY(db.alloc(d_small, 16));          /* [0..5] root keys, [6] scan total, [7] error flag */
    GB_TRY(db.alloc(d_nodes, max_nodes));
    /* a level has at most n nodes (every node holds a triangle) */
    GB_TRY(db.alloc(d_has, n));
    GB_TRY(db.alloc(d_rank_in_level, n));
    GB_TRY(db.alloc(d_child_boxes, 12 * (size_t)n));
    size_t bins_nodes = 0;

    GB_TRY(hipMemcpyAsync(d_tri9, triangles_in, 9 * (size_t)n * sizeof(float), hipMemcpyHostToDevice, stream));
    GB_TRY(hipEventRecord(ev0, stream));

    const uint32_t small_init[8] = {KEY_POS_INF, KEY_POS_INF, KEY_POS_INF, KEY_NEG_INF, KEY_NEG_INF, KEY_NEG_INF, 0u, 0u};
    GB_TRY(hipMemcpyAsync(d_small, small_init, sizeof(small_init), hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(prep_kernel, grid_for(n, 256), dim3(256), 0, stream, d_tri9, n, d_ctr, d_tmn, d_tmx, d_order, d_pos_node, d_small);
    hipLaunchKernelGGL(root_kernel, dim3(1), dim3(1), 0, stream, d_small, n, d_nodes);

    std::vector<uint32_t> &level_begin = t.level_begin, &level_count = t.level_count;
    level_begin.assign(1, 0); level_count.assign(1, 1);
    uint32_t total_nodes = 1;
    for (int depth = 0; depth < max_depth; depth++) {           /* main.rs:88: a node at max_depth is not split */
        const uint32_t lb = level_begin.back(), lc = level_count.back();
        if (lc == 0) break;
        if (lc > bins_nodes) {
            if (d_bins) db.release(d_bins);
            d_bins = nullptr;
            bins_nodes = (size_t)lc + lc / 2;
            if (bins_nodes > n) bins_nodes = n;
            GB_TRY(db.alloc(d_bins, bins_nodes * BIN_WORDS));
        }
        const size_t words = (size_t)lc * BIN_WORDS;
        hipLaunchKernelGGL(bins_init_kernel, grid_for(words, 256), dim3(256), 0, stream, d_bins, words);
        hipLaunchKernelGGL(bin_kernel, grid_for(n, 256), dim3(256), 0, stream, d_nodes, lb, d_pos_node, d_order, d_ctr, d_tmn, d_tmx, n, st, d_bins);
        hipLaunchKernelGGL(decide_kernel, grid_for(lc, 64), dim3(64), 0, stream, d_nodes, lb, lc, d_bins, st, d_child_boxes, d_has, d_small + 7);
        hipLaunchKernelGGL(small_decide_kernel, grid_for(lc, 64), dim3(64), 0, stream, d_nodes, lb, lc, d_order, d_ctr, d_tmn, d_tmx, st, d_child_boxes,
                           d_has, d_small + 7);
        hipLaunchKernelGGL(flag_kernel, grid_for(n, 256), dim3(256), 0, stream, d_nodes, d_pos_node, d_order, d_ctr, n, d_flag);
        exclusive_scan(d_flag, n, d_scan, d_sums, d_small + 6, stream);
        hipLaunchKernelGGL(a_index_kernel, grid_for(n, 256), dim3(256), 0, stream, d_nodes, d_pos_node, d_flag, d_scan, n, d_a_index);
        hipLaunchKernelGGL(permute_kernel, grid_for(n, 256), dim3(256), 0, stream, d_nodes, d_pos_node, d_flag, d_scan, d_a_index, d_order, n, d_order_next, d_rest);
        hipLaunchKernelGGL(jump_kernel, grid_for(n, 256), dim3(256), 0, stream, d_nodes, d_pos_node, d_a_index, n, d_rest);
        hipLaunchKernelGGL(chase_rest_kernel, grid_for(n, 256), dim3(256), 0, stream, d_nodes, d_pos_node, d_a_index, d_order, d_order_next, d_rest);
        GB_TRY(hipMemsetAsync(d_rest, 0, sizeof(uint32_t), stream));
        exclusive_scan(d_has, lc, d_rank_in_level, d_sums, d_small + 6, stream);
        hipLaunchKernelGGL(children_kernel, grid_for(lc, 64), dim3(64), 0, stream, d_nodes, lb, lc, d_has, d_rank_in_level, d_child_boxes, lb + lc);
        hipLaunchKernelGGL(assign_kernel, grid_for(n, 256), dim3(256), 0, stream, d_nodes, n, d_pos_node);
        std::swap(d_order, d_order_next);
        uint32_t tail[2] = {0, 0};          /* pairs created, error flag */
        GB_TRY(hipMemcpyAsync(tail, d_small + 6, sizeof(tail), hipMemcpyDeviceToHost, stream));
        GB_TRY(hipStreamSynchronize(stream));
        if (tail[1]) return fail(PT_EINVAL, "a node's cost is not a number (coordinates overflow f32 arithmetic); use pt_bvh_build");
        level_begin.push_back(lb + lc);
        level_count.push_back(2 * tail[0]);
        total_nodes += 2 * tail[0];
    }
    if (level_count.back() == 0) { level_begin.pop_back(); level_count.pop_back(); }

    /* reference order of the nodes */
    uint32_t *d_splits = nullptr, *d_rank = nullptr, *d_file_index = nullptr;
    GB_TRY(db.alloc(d_splits, total_nodes));
    GB_TRY(db.alloc(d_rank, total_nodes));
    GB_TRY(db.alloc(d_file_index, total_nodes));
    GB_TRY(hipMemsetAsync(d_rank, 0, sizeof(uint32_t), stream));
    GB_TRY(hipMemsetAsync(d_file_index, 0, sizeof(uint32_t), stream));
    for (size_t l = level_begin.size(); l-- > 0;)
        hipLaunchKernelGGL(subtree_kernel, grid_for(level_count[l], 256), dim3(256), 0, stream, d_nodes, level_begin[l], level_count[l], d_splits);
    for (size_t l = 0; l < level_begin.size(); l++)
        hipLaunchKernelGGL(number_kernel, grid_for(level_count[l], 256), dim3(256), 0, stream, d_nodes, level_begin[l], level_count[l], d_splits, d_rank, d_file_index);
    GB_TRY(hipGetLastError());
    t.d_tri9 = d_tri9; t.d_nodes = d_nodes; t.d_order = d_order;
    t.d_splits = d_splits; t.d_rank = d_rank; t.d_file_index = d_file_index; t.total_nodes = total_nodes;
    return PT_OK;
}

struct EventPairGuard {
    hipEvent_t a = nullptr, b = nullptr;
    hipError_t create() { hipError_t e = hipEventCreate(&a); return e != hipSuccess ? e : hipEventCreate(&b); }
    ~EventPairGuard() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
};

/* ---- the renderer's records straight from the tree (pt_gpu_build_resident) ---- */
__device__ __forceinline__ uint32_t leaf_ref_of(const GNode &nd, uint32_t *big) {
    if (nd.count >= PT_REF_COUNT_ESCAPE) {
        big[nd.start] = nd.count;
        return PT_REF_LEAF | (PT_REF_COUNT_ESCAPE << 24) | nd.start;
    }
    return PT_REF_LEAF | (nd.count << 24) | nd.start;
}

/* one inner record per split node, at the node's rank: the order in which the upload numbers the inner nodes of the
 * file (depth first, child A before child B) */
__global__ __launch_bounds__(256) void records_kernel(const GNode *nodes, uint32_t total, const uint32_t *rank, PTInnerRecord *inner,
                                                      uint32_t *big, uint32_t *not_tame) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const uint32_t c = nodes[i].child;
    if (!c) return;
    const GNode a = nodes[c], b = nodes[c + 1];
    const uint32_t ra = a.child ? rank[c] : leaf_ref_of(a, big);
    const uint32_t rb = b.child ? rank[c + 1] : leaf_ref_of(b, big);
    PTInnerRecord r;
    r.q[0] = make_float4(a.mn[0], a.mn[1], a.mx[0], a.mx[1]);
    r.q[1] = make_float4(b.mn[0], b.mn[1], b.mx[0], b.mx[1]);
    r.q[2] = make_float4(a.mn[2], a.mx[2], b.mn[2], b.mx[2]);
    r.q[3] = make_float4(__uint_as_float(ra), __uint_as_float(rb), 0.0f, 0.0f);
    inner[rank[i]] = r;
    bool tame = true;
    for (int k = 0; k < 3; k++)
        tame = tame && pt_div_tame_coordinate(a.mn[k]) && pt_div_tame_coordinate(a.mx[k]) &&
               pt_div_tame_coordinate(b.mn[k]) && pt_div_tame_coordinate(b.mx[k]);
    if (!tame) atomicOr(not_tame, 1u);
}

/* The refit walks the tree bottom up and only needs children before parents: the split nodes' records level by
 * level, deepest level first.  A split node's place among the split nodes of its level is half the distance of its
 * child pair from the start of the next level (pairs are appended in the order of their parents). */
__global__ __launch_bounds__(256) void refit_order_kernel(const GNode *nodes, uint32_t begin, uint32_t count, uint32_t next_begin,
                                                          const uint32_t *rank, uint32_t out_base, uint32_t *order) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const uint32_t c = nodes[begin + i].child;
    if (c) order[out_base + (c - next_begin) / 2u] = rank[begin + i];
}

/* triangle records in the tree's triangle order: the f32 operations of the upload (shader.wgsl:262-264) */
__global__ __launch_bounds__(256) void tri_records_kernel(const float *tri9, const uint32_t *order, uint32_t n, PTTriRecord *out) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const float *t = tri9 + 9 * (size_t)order[p];
    const float abx = t[3] - t[0], aby = t[4] - t[1], abz = t[5] - t[2];
    const float acx = t[6] - t[0], acy = t[7] - t[1], acz = t[8] - t[2];
    const float nx = aby * acz - abz * acy;
    const float ny = abz * acx - abx * acz;
    const float nz = abx * acy - aby * acx;
    out[p].q[0] = make_float4(t[0], t[1], t[2], abx);
    out[p].q[1] = make_float4(aby, abz, acx, acy);
    out[p].q[2] = make_float4(acz, nx, ny, nz);
}

}  // namespace

void pt_scratch_give_back(int device, void *p, size_t bytes) {
    if (p) scratch_free(device, p, bytes);
}

void pt_resident_tree_free(PTResidentTree &t) {
    if (t.d_inner) (void)hipFree(t.d_inner);
    if (t.d_tris) (void)hipFree(t.d_tris);
    if (t.d_big) (void)hipFree(t.d_big);
    if (t.d_refit_order) (void)hipFree(t.d_refit_order);
    t.d_inner = nullptr; t.d_tris = nullptr; t.d_big = nullptr; t.d_refit_order = nullptr;
}

int pt_gpu_build_resident(int device_ordinal, const float *triangles, uint32_t n, int max_depth, PTResidentTree &out, std::string &err) {
    g_error.clear();
    auto done = [&](int rc) { if (rc) { err = g_error; pt_resident_tree_free(out); } return rc; };
    if (!triangles || n == 0 || max_depth < 0 || max_depth > 1024) return done(fail(PT_EINVAL, "bad argument"));
    int rc = check_triangles(triangles, n);
    if (rc) return done(rc);
    DeviceRestore keep;
    if ((rc = select_device(device_ordinal))) return done(rc);
    DeviceBuffers db;
    db.device = device_ordinal;
    Tree t;
    hipStream_t stream = nullptr;
    EventPairGuard ev;
    hipError_t e = ev.create();
    if (e != hipSuccess) return done(fail(PT_EHIP, std::string("hipEventCreate: ") + hipGetErrorString(e)));
    if ((rc = grow_tree(triangles, n, max_depth, stream, ev.a, db, t))) return done(rc);

    auto hip = [&](hipError_t he, const char *what) { return he == hipSuccess ? PT_OK : fail(PT_EHIP, std::string(what) + ": " + hipGetErrorString(he)); };
    uint32_t *d_flag = nullptr;            /* "a bound outside the reciprocal slab test's range" */
    if ((rc = hip(db.alloc(d_flag, 1), "hipMalloc"))) return done(rc);
    const uint32_t depth = (uint32_t)t.level_begin.size() - 1u;        /* levels below the root */
    const uint32_t inner_count = (t.total_nodes - 1u) / 2u;             /* every split node added one pair */

    out.inner_count = inner_count; out.node_count = t.total_nodes; out.leaf_count = t.total_nodes - inner_count; out.depth = depth;
    void *v = nullptr;
    if ((rc = hip(scratch_alloc(device_ordinal, (inner_count ? inner_count : 1) * sizeof(PTInnerRecord), &v, &out.bytes_inner), "hipMalloc(inner records)"))) return done(rc);
    out.d_inner = (PTInnerRecord *)v;
    if ((rc = hip(scratch_alloc(device_ordinal, (size_t)n * sizeof(PTTriRecord), &v, &out.bytes_tris), "hipMalloc(triangle records)"))) return done(rc);
    out.d_tris = (PTTriRecord *)v;
    if ((rc = hip(scratch_alloc(device_ordinal, (size_t)n * sizeof(uint32_t), &v, &out.bytes_big), "hipMalloc(leaf counts)"))) return done(rc);
    out.d_big = (uint32_t *)v;
    if ((rc = hip(scratch_alloc(device_ordinal, (inner_count ? inner_count : 1) * sizeof(uint32_t), &v, &out.bytes_refit_order), "hipMalloc(refit order)"))) return done(rc);
    out.d_refit_order = (uint32_t *)v;
    if ((rc = hip(hipMemsetAsync(out.d_big, 0, (size_t)n * sizeof(uint32_t), stream), "hipMemset"))) return done(rc);
    if ((rc = hip(hipMemsetAsync(d_flag, 0, sizeof(uint32_t), stream), "hipMemset"))) return done(rc);
    hipLaunchKernelGGL(records_kernel, grid_for(t.total_nodes, 256), dim3(256), 0, stream, t.d_nodes, t.total_nodes, t.d_rank, out.d_inner, out.d_big, d_flag);
    hipLaunchKernelGGL(tri_records_kernel, grid_for(n, 256), dim3(256), 0, stream, t.d_tri9, t.d_order, n, out.d_tris);
    /* stage h of a refit (1 = first) is level depth - h; the table has depth + 2 entries like the upload's table of heights */
    out.refit_level_begin.assign((size_t)depth + 2, 0);
    for (uint32_t h = 1; h <= depth; h++) {
        const uint32_t level = depth - h, splits = t.level_count[level + 1] / 2u;
        out.refit_level_begin[h + 1] = out.refit_level_begin[h] + splits;
        hipLaunchKernelGGL(refit_order_kernel, grid_for(t.level_count[level], 256), dim3(256), 0, stream, t.d_nodes, t.level_begin[level],
                           t.level_count[level], t.level_begin[level + 1], t.d_rank, out.refit_level_begin[h], out.d_refit_order);
    }
    uint32_t not_tame = 0;
    GNode root;
    if ((rc = hip(hipMemcpyAsync(&not_tame, d_flag, 4, hipMemcpyDeviceToHost, stream), "hipMemcpy"))) return done(rc);
    if ((rc = hip(hipMemcpyAsync(&root, t.d_nodes, sizeof(GNode), hipMemcpyDeviceToHost, stream), "hipMemcpy"))) return done(rc);
    if ((rc = hip(hipEventRecord(ev.b, stream), "hipEventRecord"))) return done(rc);
    if ((rc = hip(hipGetLastError(), "kernel launch"))) return done(rc);
    if ((rc = hip(hipStreamSynchronize(stream), "hipStreamSynchronize"))) return done(rc);
    if ((rc = hip(hipEventElapsedTime(&out.build_ms, ev.a, ev.b), "hipEventElapsedTime"))) return done(rc);
    out.tame = not_tame == 0;
    if (inner_count) out.root_ref = 0;                      /* the root is split node number 0 */
    else out.root_ref = root.count >= PT_REF_COUNT_ESCAPE ? (PT_REF_LEAF | (PT_REF_COUNT_ESCAPE << 24)) : (PT_REF_LEAF | (root.count << 24));
    if (!inner_count && root.count >= PT_REF_COUNT_ESCAPE) {
        if ((rc = hip(hipMemcpy(out.d_big, &root.count, 4, hipMemcpyHostToDevice), "hipMemcpy"))) return done(rc);
    }
    return PT_OK;
}

extern "C" int pt_bvh_build_gpu(int device_ordinal, const float *triangles_in, uint32_t triangle_count, int max_depth,
                                float *triangles_out, float **nodes_out, uint32_t *node_count, float *build_ms) {
    g_error.clear();
    if (!nodes_out || !node_count || max_depth < 0 || max_depth > 1024) return fail(PT_EINVAL, "bad argument");
    if (triangle_count && (!triangles_in || !triangles_out)) return fail(PT_EINVAL, "bad argument");
    const uint32_t n = triangle_count;
    int rc = check_triangles(triangles_in, n);
    if (rc) return rc;
    if (build_ms) *build_ms = 0.0f;
    if (n == 0) {
        /* the reference writes the empty root: bounds (+inf, -inf), no triangles (main.rs:16-20) */
        float *out = (float *)malloc(9 * sizeof(float));
        if (!out) return fail(PT_ENOMEM, "out of memory");
        const float inf = std::numeric_limits<float>::infinity();
        out[0] = out[1] = out[2] = inf; out[3] = out[4] = out[5] = -inf; out[6] = out[7] = out[8] = 0.0f;
        *nodes_out = out; *node_count = 1;
        return PT_OK;
    }
    DeviceRestore keep;
    if ((rc = select_device(device_ordinal))) return rc;
    DeviceBuffers db;
    db.device = device_ordinal;
    Tree t;
    hipStream_t stream = nullptr;          /* the default stream: the call is synchronous */
    EventPairGuard ev;
    GB_TRY(ev.create());
    if ((rc = grow_tree(triangles_in, n, max_depth, stream, ev.a, db, t))) return rc;
    const uint32_t total_nodes = t.total_nodes;
    float *d_nodes_out = nullptr, *d_tris_out = nullptr;
    GB_TRY(db.alloc(d_nodes_out, 9 * (size_t)total_nodes));
    GB_TRY(db.alloc(d_tris_out, 9 * (size_t)n));
    hipLaunchKernelGGL(write_nodes_kernel, grid_for(total_nodes, 256), dim3(256), 0, stream, t.d_nodes, total_nodes, t.d_file_index, d_nodes_out);
    hipLaunchKernelGGL(write_tris_kernel, grid_for(9 * (size_t)n, 256), dim3(256), 0, stream, t.d_tri9, t.d_order, n, d_tris_out);
    GB_TRY(hipEventRecord(ev.b, stream));
    GB_TRY(hipGetLastError());
    GB_TRY(hipStreamSynchronize(stream));
    if (build_ms) GB_TRY(hipEventElapsedTime(build_ms, ev.a, ev.b));

    float *out = (float *)malloc(9 * (size_t)total_nodes * sizeof(float));
    if (!out) return fail(PT_ENOMEM, "out of memory");
    hipError_t e = hipMemcpy(out, d_nodes_out, 9 * (size_t)total_nodes * sizeof(float), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(triangles_out, d_tris_out, 9 * (size_t)n * sizeof(float), hipMemcpyDeviceToHost);
    if (e != hipSuccess) { free(out); return fail(PT_EHIP, std::string("copy of the result: ") + hipGetErrorString(e)); }
    *nodes_out = out;
    *node_count = total_nodes;
    return PT_OK;
}
