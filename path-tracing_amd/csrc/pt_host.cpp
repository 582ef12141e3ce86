/*
 * Host side of libpt_hip.so: the C ABI of include/pt_api.h.
 *
 * Mirrors what State::new / State::render do around the shader in the
 * reference (src/main.rs:286-409 buffer construction, :444-523 per-frame
 * protocol), with the wgpu buffers replaced by HIP device memory and the two
 * draws by three kernel launches per frame.
 */
#include "../../include/pt_api.h"
#include "pt_internal.h"
#include "pt_scene_build.h"
#include "pt_exact_div.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <mutex>

/*
 * Device memory for pt_render_frames_static, the entry point that allocates nothing: the hit list,
 * the control words, the per-pixel tables and the sample buffer of one batch of frames live in this
 * array instead of in buffers the context owns.  Every GPU the library runs on holds its own copy
 * (at its own address) until the process ends.  In host code the array's name is only a handle for
 * hipGetSymbolAddress: kernels, memsets and copies get the per-device address kept in
 * g_static_scratch[], looked up once per device in pt_create.
 */
#define PT_STATIC_SCRATCH_BYTES (64u << 20)
#define PT_MAX_DEVICES 64
static __device__ __align__(256) unsigned char scratch[PT_STATIC_SCRATCH_BYTES];

namespace {

std::mutex g_static_scratch_mutex;
unsigned char *g_static_scratch[PT_MAX_DEVICES];     /* device address of `scratch` on each GPU, or null */
/* Guard against two static renders that overlap on one GPU (they would share the array): the context that last
 * enqueued one (by serial number, 0 = none) and an event recorded behind its kernels.  The event is created here,
 * at set-up, so that the render call itself creates nothing. */
hipEvent_t g_static_done[PT_MAX_DEVICES];
unsigned long long g_static_owner[PT_MAX_DEVICES];
unsigned long long g_next_serial = 1;

/* with `device` current */
hipError_t static_scratch_lookup(int device) {
    if (device < 0 || device >= PT_MAX_DEVICES) return hipSuccess;      /* such a device has no static path */
    std::lock_guard<std::mutex> lock(g_static_scratch_mutex);
    if (g_static_scratch[device]) return hipSuccess;
    void *ptr = nullptr;
    hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
    if (e != hipSuccess) return e;
    e = hipEventCreateWithFlags(&g_static_done[device], hipEventDisableTiming);
    if (e != hipSuccess) return e;
    g_static_scratch[device] = (unsigned char *)ptr;
    return hipSuccess;
}

thread_local std::string g_create_error;

struct EventPair { hipEvent_t start, stop; int kind; };

}  // namespace

/*
 * What one batch of frames in flight owns: a stream, the hit list and per-pixel tables of the primary pass, the
 * control words, the sample buffer, the overflow area of the traversal stacks.  A context has two.  The path kernel
 * is a persistent grid whose last waves wait for single long paths while most of the GPU idles; the kernels of the
 * next batch, enqueued on the OTHER slot's stream, become resident as those blocks retire.  Only the resolve kernels
 * (which fold the frames into the accumulation buffer, in frame order) are chained across the two streams.
 * Slot 0 is made by pt_create; slot 1 by the first batch that is pipelined.
 */
struct BatchSlot {
    hipStream_t stream = nullptr;
    PTHitRecord *d_hits = nullptr;
    PTControl *d_ctrl = nullptr;
    uint32_t *d_pixel_hit = nullptr;
    float4 *d_pixel_const = nullptr;
    float4 *d_samples = nullptr;
    size_t sample_slots = 0;           /* float4 slots d_samples holds (allocated by the render calls, grow-only) */
    uint32_t *d_spill = nullptr;
    size_t spill_words = 0;
    uint32_t *h_hit_count = nullptr;   /* pinned host word the hit count of the primary pass is copied to */
    hipEvent_t accum_done = nullptr;   /* recorded behind this slot's last operation on the accumulation buffer */
    /* The primary pass depends on camera, scene, parameters and pixel shard only: its results are kept and reused by
     * later batches (of this and later calls) until one of those changes (pt_ctx::view_epoch). */
    unsigned long long primary_epoch = 0;      /* view_epoch the tables were made for; 0 = none */
    bool queue_clean = false;                  /* the queue heads in d_ctrl are zero (the resolve kernel leaves them so) */
    bool hit_count_pending = false;            /* the copy to h_hit_count is enqueued, hit_count not yet read */
    uint32_t hit_count = 0;
    bool ready = false;
};

struct pt_ctx {
    unsigned long long serial = 0;     /* distinguishes contexts for the static-scratch guard (addresses are reused) */
    int device = 0;
    int num_cus = 256;
    hipStream_t stream = nullptr;      /* = slot[0].stream: everything that is not a batch of frames runs here */
    BatchSlot slot[2];
    int last_slot = 0;                 /* slot of the batch enqueued last */
    unsigned long long batch_seq = 0;  /* batches enqueued so far */
    hipEvent_t batch_done[2] = {nullptr, nullptr};     /* recorded behind batch number n in batch_done[n & 1] (pt_wait_in_flight) */
    int accum_last = -1;               /* slot whose stream touched the accumulation buffer last (-1: none yet) */
    hipEvent_t ext_done = nullptr;     /* pt_accum_acquire: recorded on the caller's stream behind its work on the buffer */
    bool ext_pending = false;          /* ... and nothing of the context has waited for it yet */
    float4 *next_out = nullptr;        /* pt_next_render_output: second output of the next render call (one call only) */
    bool next_move = false;
    unsigned long long view_epoch = 1; /* bumped by whatever changes the primary pass's results */
    bool pipeline = true;              /* PT_PIPELINE=0: every batch on slot 0 */
    bool adapt_grid = true;            /* PT_ADAPT_GRID=0: the path kernel always gets its full grid */
    bool timing_events = true;         /* PT_NO_TIMING=1: no HIP events around the kernels (pt_get_timing then reports no kernel time) */
    size_t pipeline_bytes = (size_t)16 << 30;  /* largest sample buffer a pipelined batch may need (PT_PIPELINE_MB) */
    size_t spill_words = 0;            /* size the scene asks of BatchSlot::d_spill */
    unsigned long long known_hits_epoch = 0;   /* view_epoch for which known_hits is the primary pass's hit count */
    uint32_t known_hits = 0;
    pt_params params{};
    float cam_pos[3] = {200.0f, 0.0f, 200.0f};      /* src/main.rs:382 */
    float cam_rot[3] = {0.0f, 90.0f, 0.0f};         /* src/main.rs:383 */

    PTInnerRecord *d_inner = nullptr;
    PTTriRecord *d_tris = nullptr;
    float *d_spheres = nullptr;
    uint32_t *d_big = nullptr;
    PTWideRecord *d_wide = nullptr;     /* PT_FLAG_WIDE_BVH: the four-wide tree */
    uint32_t wide_count = 0, wide_root_ref = 0, wide_depth = 0;
    int wide_stack_entries = 0;
    float4 *d_accum_own = nullptr;
    float4 *d_accum = nullptr;
    size_t sample_budget = 0;          /* most bytes one slot's sample buffer may take */
    /* Batches whose worst case (every pixel a hit pixel) stays below this allocate for the worst case and
     * stay asynchronous; larger ones wait for the primary pass and size the buffer by the hit count.
     * 1 GiB; PT_SMALL_SAMPLE_MB overrides (0: always by hit count). */
    size_t small_sample_bytes = (size_t)1 << 30;
    uint32_t batch_cap = 1;            /* most frames one batch may hold (32-bit slot numbers, at most 64) */
    unsigned long long *d_counters = nullptr;
    int32_t *d_ids = nullptr;
    float *d_dist = nullptr;

    /* refit (pt_refit_scene): inner records ordered by height, where each height starts, a staging buffer for
     * the new triangles (allocated by the first refit), a flag word */
    uint32_t *d_refit_order = nullptr;
    std::vector<uint32_t> refit_level_begin;
    float *d_tri9 = nullptr;
    uint32_t *d_refit_flag = nullptr;
    bool refit_ok = false;
    std::string refit_why;
    size_t bytes_inner = 0, bytes_tris = 0, bytes_big = 0, bytes_refit_order = 0;     /* sizes of the scene allocations */

    bool scene_loaded = false;
    uint32_t sphere_count = 0, tri_count = 0, node_count = 0;
    uint32_t root_ref = 0, depth = 0, inner_count = 0, leaf_count = 0;
    int stack_entries = 0;
    bool fast_div = false;
    uint32_t shard_rank = 0, shard_world = 1;
    uint32_t refill_min = 48;
    uint32_t quad_fetch = 0;           /* PT_QUAD_FETCH: see PTKernelParams::quad_fetch */
    uint32_t two_paths = 0;            /* PT_TWO_PATHS: see PTKernelParams::two_paths */
    uint32_t starve_min = 16;
    float4 *d_pathbuf = nullptr;
    uint32_t leaf_min = 8;
    uint32_t leaf_shift = 2;

    std::vector<EventPair> pending;    /* recorded, not yet read */
    std::vector<EventPair> free_events;
    pt_timing timing{};

    std::string err;
};

namespace {

int fail(pt_ctx *c, int code, const std::string &msg) {
    if (c) c->err = msg; else g_create_error = msg;
    return code;
}

int fail_hip(pt_ctx *c, hipError_t e, const char *what) {
    return fail(c, PT_EHIP, std::string(what) + ": " + hipGetErrorString(e));
}

#define HIP_TRY(ctx, call)                                              \
    do {                                                                \
        hipError_t e_ = (call);                                         \
        if (e_ != hipSuccess) return fail_hip((ctx), e_, #call);        \
    } while (0)

template <typename T> void dev_free(T *&p) {
    if (p) { (void)hipFree(p); p = nullptr; }
}

/* ---- node file -> inner records --------------------------------------- */
struct Repacked {
    std::vector<PTInnerRecord> inner;
    std::vector<uint32_t> big;     /* per-triangle count table, only when a leaf has >= 127 triangles */
    uint32_t root_ref = 0, depth = 0, leaves = 0;
    bool tame_bounds = true;       /* see pt_exact_div.h */
    std::vector<uint32_t> height;  /* per inner record: 1 + the larger height of its inner children (leaf children count 0) */
    bool refit_ok = true;          /* every reachable inner node's triangle range is the ranges of its children, A then B */
    std::string refit_why;
};

bool finite_nonneg_int(float f) { return f >= 0.0f && f <= 16777216.0f && f == std::floor(f); }

/* leaf reference for node `n` of the file; returns false on a malformed range */
bool leaf_ref(const float *node, uint32_t tri_count, Repacked &out, uint32_t &ref, std::string &why) {
    if (!finite_nonneg_int(node[6]) || !(node[7] >= 0.0f) || !(node[6] + node[7] <= 16777216.0f)) {
        why = "leaf with a negative, fractional or non-finite triangle range";
        return false;
    }
    /* the shader loops i = u32(node[6]) .. u32(node[6] + node[7]) (shader.wgsl:186) */
    uint32_t first = (uint32_t)node[6];
    uint32_t end = (uint32_t)(node[6] + node[7]);
    uint32_t count = end > first ? end - first : 0;
    if ((uint64_t)first + count > tri_count) { why = "leaf triangle range leaves the triangle array"; return false; }
    if (first >= (1u << 24)) { why = "triangle index does not fit 24 bits"; return false; }
    if (count >= PT_REF_COUNT_ESCAPE) {
        if (out.big.empty()) out.big.assign(tri_count ? tri_count : 1, 0);
        /* the side table is keyed by the first triangle: two big leaves that start at the same triangle
         * with different lengths (a node file with overlapping ranges) cannot both be represented */
        if (out.big[first] != 0 && out.big[first] != count) { why = "two leaves of 127 or more triangles start at the same triangle with different counts"; return false; }
        out.big[first] = count;
        ref = PT_REF_LEAF | (PT_REF_COUNT_ESCAPE << 24) | first;
    } else {
        ref = PT_REF_LEAF | (count << 24) | first;
    }
    return true;
}

/*
 * Walks the file from node 0 exactly as the shader would (leaf iff
 * node[8] == 0, children i32(node[8]) and i32(node[8] + 1), shader.wgsl:184,
 * 197-198), numbering reachable inner nodes in depth-first pre-order.  A node
 * reached twice (the legacy dragon_8k file has such links) is converted once.
 * Cycles and links outside the array are errors.
 */
bool repack_nodes(const float *bvh, uint32_t node_count, uint32_t tri_count, Repacked &out, std::string &why) {
    const uint32_t UNSEEN = 0xffffffffu, OPEN = 0xfffffffeu, LEAF_SEEN = 0xfffffffdu;
    std::vector<uint32_t> rec_of(node_count, UNSEEN);   /* inner record index per file node */
    std::vector<uint32_t> height(node_count, 0);
    std::vector<uint32_t> leafref(node_count, 0);
    struct Frame { uint32_t node, a, b, rec; int stage; };
    std::vector<Frame> st;

    const float *root = bvh;
    if (root[8] == 0.0f) {
        if (!leaf_ref(root, tri_count, out, out.root_ref, why)) return false;
        out.depth = 0; out.leaves = 1;
        return true;
    }
    auto open = [&](uint32_t n) -> bool {
        const float *node = bvh + 9 * (size_t)n;
        float cf = node[8];
        if (!(cf >= 1.0f) || !(cf < 2147483000.0f)) { why = "inner node with a child link outside the node array"; return false; }
        int32_t ia = (int32_t)cf, ib = (int32_t)(cf + 1.0f);
        if (ia <= 0 || ib <= 0 || (uint32_t)ia >= node_count || (uint32_t)ib >= node_count) {
            why = "inner node with a child link outside the node array";
            return false;
        }
        rec_of[n] = OPEN;
        Frame f = {n, (uint32_t)ia, (uint32_t)ib, (uint32_t)out.inner.size(), 0};
        out.inner.emplace_back();
        st.push_back(f);
        return true;
    };
    if (!open(0)) return false;
    while (!st.empty()) {
        const size_t top = st.size() - 1;
        if (st[top].stage < 2) {
            const uint32_t c = st[top].stage == 0 ? st[top].a : st[top].b;
            st[top].stage++;
            const float *cn = bvh + 9 * (size_t)c;
            if (cn[8] == 0.0f) {
                if (rec_of[c] == UNSEEN) {
                    if (!leaf_ref(cn, tri_count, out, leafref[c], why)) return false;
                    rec_of[c] = LEAF_SEEN;
                    out.leaves++;
                }
            } else if (rec_of[c] == UNSEEN) {
                if (st.size() > 4096) { why = "node file deeper than 4096 levels"; return false; }
                if (!open(c)) return false;
            } else if (rec_of[c] == OPEN) {
                why = "node file contains a cycle";
                return false;
            }
            continue;
        }
        /* both children converted: fill this node's record */
        const Frame f = st[top];
        const float *na = bvh + 9 * (size_t)f.a, *nb = bvh + 9 * (size_t)f.b;
        const bool leaf_a = na[8] == 0.0f, leaf_b = nb[8] == 0.0f;
        uint32_t ra = leaf_a ? leafref[f.a] : rec_of[f.a];
        uint32_t rb = leaf_b ? leafref[f.b] : rec_of[f.b];
        uint32_t ha = leaf_a ? 0 : height[f.a];
        uint32_t hb = leaf_b ? 0 : height[f.b];
        PTInnerRecord &r = out.inner[f.rec];
        r.q[0] = make_float4(na[0], na[1], na[3], na[4]);
        r.q[1] = make_float4(nb[0], nb[1], nb[3], nb[4]);
        r.q[2] = make_float4(na[2], na[5], nb[2], nb[5]);
        for (int k = 0; k < 6; k++)
            if (!pt_div_tame_coordinate(na[k]) || !pt_div_tame_coordinate(nb[k])) out.tame_bounds = false;
        float fa, fb;
        memcpy(&fa, &ra, 4); memcpy(&fb, &rb, 4);
        r.q[3] = make_float4(fa, fb, 0.0f, 0.0f);
        height[f.node] = 1 + (ha > hb ? ha : hb);
        if (out.height.size() < out.inner.size()) out.height.resize(out.inner.size(), 0);
        out.height[f.rec] = height[f.node];
        if (out.refit_ok) {
            /* what a refit relies on: the node's range is child A's followed by child B's, none of them empty */
            const float *nn = bvh + 9 * (size_t)f.node;
            const bool ints = finite_nonneg_int(nn[6]) && finite_nonneg_int(nn[7]) && finite_nonneg_int(na[6]) && finite_nonneg_int(na[7]) &&
                              finite_nonneg_int(nb[6]) && finite_nonneg_int(nb[7]);
            if (!ints || na[7] < 1.0f || nb[7] < 1.0f || na[6] != nn[6] || nb[6] != na[6] + na[7] || na[7] + nb[7] != nn[7] ||
                (double)nn[6] + (double)nn[7] > (double)tri_count) {
                out.refit_ok = false;
                out.refit_why = "node " + std::to_string(f.node) + " of the file does not cover exactly the triangle ranges of its children";
            }
        }
        rec_of[f.node] = f.rec;
        st.pop_back();
    }
    /* record offsets (index * 64) are 32-bit in the kernels */
    if (out.inner.size() >= (1u << 26)) { why = "more than 2^26 reachable inner nodes"; return false; }
    out.root_ref = rec_of[0];
    out.depth = height[0];
    return true;
}

void build_tri_records(const float *tris, uint32_t n, std::vector<PTTriRecord> &out) {
    out.resize(n ? n : 1);
    for (uint32_t i = 0; i < n; i++) {
        const float *t = tris + 9 * (size_t)i;
        /* shader.wgsl:262-264, same operations in the same order */
        float abx = t[3] - t[0], aby = t[4] - t[1], abz = t[5] - t[2];
        float acx = t[6] - t[0], acy = t[7] - t[1], acz = t[8] - t[2];
        float nx = aby * acz - abz * acy;
        float ny = abz * acx - abx * acz;
        float nz = abx * acy - aby * acx;
        out[i].q[0] = make_float4(t[0], t[1], t[2], abx);
        out[i].q[1] = make_float4(aby, abz, acx, acy);
        out[i].q[2] = make_float4(acz, nx, ny, nz);
    }
}

/* ---- binary records -> four-wide records (PT_FLAG_WIDE_BVH) ------------- */
struct WideTree {
    std::vector<PTWideRecord> nodes;
    uint32_t root_ref = 0, depth = 0;
};

uint32_t float_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
float bits_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* Folds the binary tree: a wide node is an inner node of the file whose inner children are replaced by
 * their own two children (two to four children in all).  Boxes are copied, never merged or re-derived.
 * A binary node reached over several links (legacy files have such links) is folded once. */
void build_wide(const Repacked &rp, WideTree &out) {
    if (rp.root_ref & PT_REF_LEAF) { out.root_ref = rp.root_ref; out.depth = 0; return; }
    struct Child { float lo[3], hi[3]; uint32_t ref; };
    auto children_of = [&](uint32_t rec, Child c[2]) {
        const PTInnerRecord &r = rp.inner[rec];
        c[0] = Child{{r.q[0].x, r.q[0].y, r.q[2].x}, {r.q[0].z, r.q[0].w, r.q[2].y}, float_bits(r.q[3].x)};
        c[1] = Child{{r.q[1].x, r.q[1].y, r.q[2].z}, {r.q[1].z, r.q[1].w, r.q[2].w}, float_bits(r.q[3].y)};
    };
    std::vector<uint32_t> wide_of(rp.inner.size(), PT_REF_NONE), depth_of(rp.inner.size(), 0);
    struct Frame { uint32_t rec, wide; Child ch[4]; int n, next; };
    std::vector<Frame> st;
    auto open = [&](uint32_t rec) {
        Frame f;
        f.rec = rec; f.wide = (uint32_t)out.nodes.size(); f.n = 0; f.next = 0;
        Child two[2];
        children_of(rec, two);
        for (int k = 0; k < 2; k++) {
            if (two[k].ref & PT_REF_LEAF) f.ch[f.n++] = two[k];
            else { Child g[2]; children_of(two[k].ref, g); f.ch[f.n++] = g[0]; f.ch[f.n++] = g[1]; }
        }
        wide_of[rec] = f.wide;
        out.nodes.emplace_back();
        st.push_back(f);
    };
    open(rp.root_ref);
    while (!st.empty()) {
        Frame &f = st.back();
        if (f.next < f.n) {
            const uint32_t ref = f.ch[f.next++].ref;
            if (!(ref & PT_REF_LEAF) && wide_of[ref] == PT_REF_NONE) open(ref);       /* invalidates f: re-read st.back() next round */
            continue;
        }
        uint32_t d = 0;
        PTWideRecord &w = out.nodes[f.wide];
        uint32_t refs[4];
        Child c[4];
        for (int k = 0; k < 4; k++) {
            c[k] = k < f.n ? f.ch[k] : f.ch[0];                /* an absent child repeats a real box and carries no reference */
            refs[k] = PT_REF_NONE;
            if (k < f.n) {
                refs[k] = (c[k].ref & PT_REF_LEAF) ? c[k].ref : wide_of[c[k].ref];
                if (!(c[k].ref & PT_REF_LEAF) && depth_of[c[k].ref] > d) d = depth_of[c[k].ref];
            }
        }
        for (int pair = 0; pair < 2; pair++) {
            const Child &a = c[2 * pair], &b = c[2 * pair + 1];
            w.q[3 * pair + 0] = make_float4(a.lo[0], a.lo[1], a.hi[0], a.hi[1]);
            w.q[3 * pair + 1] = make_float4(b.lo[0], b.lo[1], b.hi[0], b.hi[1]);
            w.q[3 * pair + 2] = make_float4(a.lo[2], a.hi[2], b.lo[2], b.hi[2]);
        }
        w.q[6] = make_float4(bits_float(refs[0]), bits_float(refs[1]), bits_float(refs[2]), bits_float(refs[3]));
        w.q[7] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        depth_of[f.rec] = d + 1;
        st.pop_back();
    }
    out.root_ref = wide_of[rp.root_ref];
    out.depth = depth_of[rp.root_ref];
}

int validate_params(const pt_params *p, std::string &why) {
    if (!p) { why = "params is NULL"; return PT_EINVAL; }
    if (p->width == 0 || p->height == 0 || p->width > 16384 || p->height > 16384) { why = "width/height must be in 1..16384"; return PT_EINVAL; }
    /* pixel_index is computed in f32 (shader.wgsl:83) and must stay exact */
    if ((uint64_t)p->width * p->height + p->width > (1ull << 23)) { why = "width*height too large for the shader's f32 pixel index (2^23)"; return PT_EINVAL; }
    if (p->rays_per_pixel == 0 || p->rays_per_pixel > 1024) { why = "rays_per_pixel must be in 1..1024"; return PT_EINVAL; }
    if (p->max_bounce_count > 1000000) { why = "max_bounce_count too large"; return PT_EINVAL; }
    if (!(p->fov_degrees > 0.0f && p->fov_degrees < 180.0f)) { why = "fov_degrees must be in (0,180)"; return PT_EINVAL; }
    /* a missed box reports 999999999.0 (shader.wgsl:254) and must never pass `dst < far` */
    if (!(p->far_distance > 0.0f && p->far_distance <= 999999999.0f)) { why = "far_distance must be in (0, 999999999]"; return PT_EINVAL; }
    if (p->bvh_max_depth > 4096) { why = "bvh_max_depth too large"; return PT_EINVAL; }
    if ((p->flags & PT_FLAG_WIDE_BVH) && (p->flags & PT_FLAG_COUNT_WORK)) { why = "PT_FLAG_WIDE_BVH and PT_FLAG_COUNT_WORK cannot be combined"; return PT_EINVAL; }
    return PT_OK;
}

void fill_kernel_params(const pt_ctx *c, const BatchSlot &bs, uint32_t frame_first, uint32_t frame_stride, uint32_t batch_frames,
                        uint32_t accum_mode, PTKernelParams &k) {
    memset(&k, 0, sizeof(k));
    const pt_params &p = c->params;
    k.width = p.width; k.height = p.height;
    k.rays_per_pixel = p.rays_per_pixel; k.max_bounce_count = p.max_bounce_count;
    k.frame_first = frame_first; k.frame_stride = frame_stride; k.batch_frames = batch_frames;
    k.div_width = pt_fastdiv_make(p.width);
    k.div_rpp = pt_fastdiv_make(p.rays_per_pixel);
    k.div_batch = pt_fastdiv_make(batch_frames);
    k.accum_mode = accum_mode;
    k.mirror = nullptr; k.mirror_move = 0;
    k.shard_rank = c->shard_rank; k.shard_world = c->shard_world;
    k.width_f = (float)p.width; k.height_f = (float)p.height;
    /* vs_main, shader.wgsl:36-37,68-69.  fov is a const-expression of abstract
     * floats (f64, rounded once); tan is taken in double and rounded once. */
    float fov = (float)((double)p.fov_degrees * 3.14159 / 180.0);
    float aspect = (float)p.width / (float)p.height;
    k.screen_w = (float)std::tan((double)(fov * 0.5f)) * 2.0f;
    k.screen_h = k.screen_w / aspect;
    for (int i = 0; i < 3; i++) k.cam_pos[i] = c->cam_pos[i];
    /* rotate_vector, shader.wgsl:332-341: angle * 3.14159 / 180 in f32 */
    float a = c->cam_rot[0] * 3.14159f / 180.0f;
    float b = c->cam_rot[1] * 3.14159f / 180.0f;
    float cc = c->cam_rot[2] * 3.14159f / 180.0f;
    k.cos_a = (float)std::cos((double)a); k.sin_a = (float)std::sin((double)a);
    k.cos_b = (float)std::cos((double)b); k.sin_b = (float)std::sin((double)b);
    k.cos_c = (float)std::cos((double)cc); k.sin_c = (float)std::sin((double)cc);
    for (int i = 0; i < 3; i++) {
        k.tri_color[i] = p.tri_color[i];
        k.tri_emission_color[i] = p.tri_emission_color[i];
        k.sky_horizon[i] = p.sky_horizon[i];
        k.sky_zenith[i] = p.sky_zenith[i];
        k.ground[i] = p.ground[i];
    }
    k.tri_emission_strength = p.tri_emission_strength;
    k.tri_smoothness = p.tri_smoothness;
    /* -normalize(sun_light_direction), shader.wgsl:395 */
    float sx = p.sun_dir[0], sy = p.sun_dir[1], sz = p.sun_dir[2];
    float inv = 1.0f / std::sqrt(sx * sx + sy * sy + sz * sz);
    k.neg_sun[0] = -(sx * inv); k.neg_sun[1] = -(sy * inv); k.neg_sun[2] = -(sz * inv);
    k.sun_intensity = p.sun_intensity; k.sun_focus = p.sun_focus;
    k.use_env = p.use_environment_lighting ? 1u : 0u;
    k.far_distance = p.far_distance;
    k.sphere_count = c->sphere_count;
    k.root_ref = c->root_ref;
    k.stack_entries = (uint32_t)c->stack_entries;
    if (c->d_wide) {
        k.wide = c->d_wide; k.wide_bytes = c->wide_count * (uint32_t)sizeof(PTWideRecord);
        k.root_ref = c->wide_root_ref;
        k.stack_entries = (uint32_t)c->wide_stack_entries;
    }
    k.fast_div = c->fast_div ? 1u : 0u;
    k.refill_min = c->refill_min;
    k.quad_fetch = c->quad_fetch;
    k.two_paths = (c->two_paths && c->d_pathbuf && !c->d_wide) ? 1u : 0u;
    k.starve_min = c->starve_min;
    k.pathbuf = c->d_pathbuf;
    k.pathbuf_bytes = (uint32_t)pt_pathbuf_bytes(c->num_cus);
    k.leaf_min = c->leaf_min;
    k.leaf_shift = c->leaf_shift;
    k.xcd_queues = getenv("PT_NO_XCD_QUEUES") ? 0u : 1u;
    k.inner = c->d_inner; k.inner_bytes = (c->inner_count ? c->inner_count : 1u) * (uint32_t)sizeof(PTInnerRecord);
    k.tris_bytes = (c->tri_count ? c->tri_count : 1u) * (uint32_t)sizeof(PTTriRecord); k.tris = c->d_tris; k.spheres = c->d_spheres; k.big_leaf_count = c->d_big;
    k.accum = c->d_accum; k.hits = bs.d_hits; k.ctrl = bs.d_ctrl; k.samples = bs.d_samples;
    k.pixel_hit = bs.d_pixel_hit; k.pixel_const = bs.d_pixel_const;
    k.spill = bs.d_spill; k.spill_stride = pt_lane_stride(c->num_cus);
    k.out_ids = c->d_ids; k.out_dist = c->d_dist; k.counters = c->d_counters;
}

int get_event_pair(pt_ctx *c, EventPair &ep) {
    if (!c->free_events.empty()) { ep = c->free_events.back(); c->free_events.pop_back(); return PT_OK; }
    HIP_TRY(c, hipEventCreate(&ep.start));
    HIP_TRY(c, hipEventCreate(&ep.stop));
    return PT_OK;
}

int drain_events(pt_ctx *c) {
    for (EventPair &ep : c->pending) {
        float ms = 0.0f;
        HIP_TRY(c, hipEventElapsedTime(&ms, ep.start, ep.stop));
        if (ep.kind == 0) c->timing.primary_ms += ms;
        else if (ep.kind == 1) c->timing.path_ms += ms;
        else c->timing.resolve_ms += ms;
        c->free_events.push_back(ep);
    }
    c->pending.clear();
    return PT_OK;
}

struct Timed {
    pt_ctx *c; hipStream_t st; EventPair ep; bool ok;
    Timed(pt_ctx *ctx, hipStream_t stream, int kind) : c(ctx), st(stream), ok(false) {
        if (!c->timing_events) return;
        if (get_event_pair(c, ep) == PT_OK) { ep.kind = kind; ok = hipEventRecord(ep.start, st) == hipSuccess; }
    }
    void done() { if (ok) { (void)hipEventRecord(ep.stop, st); c->pending.push_back(ep); } }
};

/* waits for everything the context has enqueued, on both slots' streams */
hipError_t sync_all(pt_ctx *c) {
    hipError_t e = hipSuccess;
    for (int s = 0; s < 2; s++)
        if (c->slot[s].stream) { hipError_t es = hipStreamSynchronize(c->slot[s].stream); if (e == hipSuccess) e = es; }
    /* ... and for what the caller's stream did to the accumulation buffer before it handed it back */
    if (c->ext_pending) { hipError_t es = hipEventSynchronize(c->ext_done); if (e == hipSuccess) e = es; c->ext_pending = false; }
    return e;
}

/* The accumulation buffer is written in a fixed order (a running mean is folded frame by frame).  Whatever touches
 * it on slot s's stream first waits for the last such operation of the other stream, and leaves its own mark. */
hipError_t accum_wait(pt_ctx *c, int s) {
    if (c->ext_pending) {       /* the buffer was lent to a stream of the caller's (pt_accum_acquire): its work comes first */
        hipError_t e = hipStreamWaitEvent(c->slot[s].stream, c->ext_done, 0);
        if (e != hipSuccess) return e;
        c->ext_pending = false;
    }
    if (c->accum_last < 0 || c->accum_last == s) return hipSuccess;
    return hipStreamWaitEvent(c->slot[s].stream, c->slot[c->accum_last].accum_done, 0);
}
hipError_t accum_mark(pt_ctx *c, int s) {
    c->accum_last = s;
    return hipEventRecord(c->slot[s].accum_done, c->slot[s].stream);
}

/* stream, event, pinned word and the per-pixel buffers of a slot (not its sample buffer and stack overflow area:
 * the render calls size those) */
hipError_t slot_prepare(pt_ctx *c, int s) {
    BatchSlot &b = c->slot[s];
    if (b.ready) return hipSuccess;
    const size_t npix = (size_t)c->params.width * c->params.height;
    hipError_t e;
    if (!b.stream && (e = hipStreamCreateWithFlags(&b.stream, hipStreamNonBlocking)) != hipSuccess) return e;
    if (!b.accum_done && (e = hipEventCreateWithFlags(&b.accum_done, hipEventDisableTiming)) != hipSuccess) return e;
    if (!b.h_hit_count && (e = hipHostMalloc((void **)&b.h_hit_count, sizeof(uint32_t), hipHostMallocDefault)) != hipSuccess) return e;
    if (!b.d_hits && (e = hipMalloc(&b.d_hits, npix * sizeof(PTHitRecord))) != hipSuccess) return e;
    if (!b.d_ctrl && (e = hipMalloc(&b.d_ctrl, sizeof(PTControl))) != hipSuccess) return e;
    if (!b.d_pixel_hit && (e = hipMalloc(&b.d_pixel_hit, npix * sizeof(uint32_t))) != hipSuccess) return e;
    if (!b.d_pixel_const && (e = hipMalloc(&b.d_pixel_const, npix * sizeof(float4))) != hipSuccess) return e;
    b.ready = true;
    return hipSuccess;
}

void slot_release(BatchSlot &b) {
    dev_free(b.d_hits); dev_free(b.d_ctrl); dev_free(b.d_pixel_hit); dev_free(b.d_pixel_const); dev_free(b.d_samples); dev_free(b.d_spill);
    if (b.h_hit_count) { (void)hipHostFree(b.h_hit_count); b.h_hit_count = nullptr; }
    if (b.accum_done) { (void)hipEventDestroy(b.accum_done); b.accum_done = nullptr; }
    if (b.stream) { (void)hipStreamDestroy(b.stream); b.stream = nullptr; }
    b.ready = false; b.sample_slots = 0; b.spill_words = 0; b.primary_epoch = 0; b.queue_clean = false;
}

/* the overflow area of the traversal stacks, sized for the scene (only trees deeper than the LDS part use it) */
hipError_t ensure_spill(pt_ctx *c, BatchSlot &b) {
    const size_t words = c->spill_words ? c->spill_words : 1;
    if (b.d_spill && b.spill_words >= words) return hipSuccess;
    hipError_t e = hipStreamSynchronize(b.stream);
    if (e != hipSuccess) return e;
    dev_free(b.d_spill); b.spill_words = 0;
    e = hipMalloc(&b.d_spill, words * sizeof(uint32_t));
    if (e != hipSuccess) { (void)hipGetLastError(); b.d_spill = nullptr; return e; }
    b.spill_words = words;
    return hipSuccess;
}

/* where a render keeps its per-batch state when that is not the context's own buffers */
struct WorkBuffers {
    PTControl *ctrl; uint32_t *pixel_hit; float4 *pixel_const; PTHitRecord *hits; float4 *samples;
    uint32_t max_batch;
};

size_t round256(size_t n) { return (n + 255u) & ~(size_t)255u; }

/* Cuts the static scratch of this context's GPU into the buffers of one batch.  Fails when not even
 * one frame fits. */
int static_work_buffers(pt_ctx *c, uint32_t n_frames, WorkBuffers &wb) {
    unsigned char *base = (c->device >= 0 && c->device < PT_MAX_DEVICES) ? g_static_scratch[c->device] : nullptr;
    if (!base) return fail(c, PT_ESTATE, "no static scratch on this device");
    const size_t npix = (size_t)c->params.width * c->params.height;
    const size_t slots_per_frame = ((npix + 63) / 64 * 64) * c->params.rays_per_pixel;
    const size_t per_frame = slots_per_frame * sizeof(float4);
    size_t off = 0;
    const size_t o_ctrl = off; off += round256(sizeof(PTControl));
    const size_t o_hit = off; off += round256(npix * sizeof(uint32_t));
    const size_t o_const = off; off += round256(npix * sizeof(float4));
    const size_t o_hits = off; off += round256(npix * sizeof(PTHitRecord));
    const size_t o_samples = off;
    if (off + per_frame > PT_STATIC_SCRATCH_BYTES)
        return fail(c, PT_EINVAL, "pt_render_frames_static: one frame of " + std::to_string(c->params.width) + "x" +
                    std::to_string(c->params.height) + " at " + std::to_string(c->params.rays_per_pixel) + " rays per pixel needs " +
                    std::to_string(off + per_frame) + " bytes of scratch, the static array holds " +
                    std::to_string((size_t)PT_STATIC_SCRATCH_BYTES) + " (use pt_render_frames)");
    size_t nb = (PT_STATIC_SCRATCH_BYTES - off) / per_frame;
    if (nb > 0xffffffffull / (slots_per_frame + 512u * c->params.rays_per_pixel))       /* the queue arithmetic rounds the hit pixels up to whole parts (fewer than 512 pixels more) */
        nb = 0xffffffffull / (slots_per_frame + 512u * c->params.rays_per_pixel);
    if (nb > 64) nb = 64;
    if (nb > n_frames) nb = n_frames ? n_frames : 1;
    wb.ctrl = (PTControl *)(base + o_ctrl);
    wb.pixel_hit = (uint32_t *)(base + o_hit);
    wb.pixel_const = (float4 *)(base + o_const);
    wb.hits = (PTHitRecord *)(base + o_hits);
    wb.samples = (float4 *)(base + o_samples);
    wb.max_batch = (uint32_t)nb;
    return PT_OK;
}

/* Makes the slot's d_samples hold at least `slots` float4 values.  The old contents are not kept (a batch writes
 * every sample it reads).  hipErrorOutOfMemory is returned as such so that the caller can try a smaller batch. */
hipError_t ensure_sample_slots(BatchSlot &b, size_t slots) {
    if (slots <= b.sample_slots) return hipSuccess;
    hipError_t e = hipStreamSynchronize(b.stream);        /* earlier batches may still read the old buffer */
    if (e != hipSuccess) return e;
    dev_free(b.d_samples);
    b.sample_slots = 0;
    e = hipMalloc(&b.d_samples, slots * sizeof(float4));
    if (e != hipSuccess) { (void)hipGetLastError(); b.d_samples = nullptr; return e; }
    b.sample_slots = slots;
    return hipSuccess;
}


/* The hit count of the slot's primary pass, once its copy has arrived; `wait` blocks for it. */
hipError_t slot_hit_count(pt_ctx *c, BatchSlot &b, bool wait, bool &known) {
    known = false;
    if (b.primary_epoch != c->view_epoch) return hipSuccess;
    if (b.hit_count_pending) {
        if (wait) {
            hipError_t e = hipStreamSynchronize(b.stream);
            if (e != hipSuccess) return e;
        } else {
            hipError_t q = hipStreamQuery(b.stream);
            if (q == hipErrorNotReady) { (void)hipGetLastError(); return hipSuccess; }
            if (q != hipSuccess) return q;
        }
        b.hit_count = *b.h_hit_count;
        b.hit_count_pending = false;
        c->known_hits = b.hit_count;
        c->known_hits_epoch = c->view_epoch;
    }
    known = true;
    return hipSuccess;
}

/* wb == nullptr: the context's own buffers, kernels timed with events.  wb != nullptr (the static
 * path): no events either, so that the call creates no runtime object at all. */
int render_impl(pt_ctx *c, uint32_t first, uint32_t n_frames, uint32_t stride, uint32_t mode, const WorkBuffers *wb = nullptr) {
    if (!c) return PT_EINVAL;
    if (!c->scene_loaded) return fail(c, PT_ESTATE, "no scene uploaded");
    HIP_TRY(c, hipSetDevice(c->device));
    const bool exact = (c->params.flags & PT_FLAG_EXACT_MATH) != 0;
    const bool count = (c->params.flags & PT_FLAG_COUNT_WORK) != 0;
    const size_t npix = (size_t)c->params.width * c->params.height;
    const size_t rpp = c->params.rays_per_pixel;
    const size_t worst_per_frame = ((npix + 63) / 64 * 64) * rpp;      /* slots when every pixel is a hit pixel */
    uint32_t max_batch = wb ? wb->max_batch : c->batch_cap;
    if (!wb) {      /* the budget holds for the worst case */
        size_t by_budget = c->sample_budget / (worst_per_frame * sizeof(float4));
        if (by_budget < 1) by_budget = 1;
        if (by_budget < max_batch) max_batch = (uint32_t)by_budget;
    }
    /* counting contexts run their primary pass exactly once per batch and share one set of counters; the two-paths
     * schedule shares one set of path buffers: neither is pipelined */
    bool may_pipeline = !wb && c->pipeline && !count && !(c->two_paths && c->d_pathbuf);
    /* the second output asked for with pt_next_render_output belongs to this call, however it ends */
    float4 *const second_out = c->next_out;
    const bool second_move = c->next_move;
    c->next_out = nullptr; c->next_move = false;
    /* Frames are rendered in batches: camera and scene are constant within the call, so the
     * primary rays (identical for every sample of every frame, shader.wgsl:80-101) are cast once
     * per batch; the path kernel then runs the bounces of ALL frames of the batch from one queue
     * and the resolve kernel folds the frames into the accumulation buffer in frame order. */
    for (uint32_t done = 0; done < n_frames;) {
        uint32_t batch = n_frames - done < max_batch ? n_frames - done : max_batch;
        /* The slot: the one the last batch did NOT use, so that this batch's kernels can start while that one's path
         * kernel drains - if the sample buffer the batch needs is small (a long batch gains nothing). */
        int s = 0;
        if (!wb && !count) {
            for (int q = 0; q < 2; q++) {       /* pick up hit counts that have arrived, without waiting */
                bool known;
                if (c->slot[q].ready) HIP_TRY(c, slot_hit_count(c, c->slot[q], false, known));
            }
        }
        if (may_pipeline) {
            size_t per_frame = worst_per_frame;
            if (c->known_hits_epoch == c->view_epoch) per_frame = (((size_t)c->known_hits + 63) / 64 * 64) * rpp;
            if (per_frame * batch * sizeof(float4) <= c->pipeline_bytes) s = c->last_slot ^ 1;
        }
        if (!c->slot[s].ready) {
            hipError_t e = slot_prepare(c, s);
            if (e == hipErrorOutOfMemory && s == 1) { (void)hipGetLastError(); (void)hipStreamSynchronize(c->slot[1].stream); if (c->accum_last == 1) c->accum_last = -1; slot_release(c->slot[1]); c->pipeline = false; may_pipeline = false; c->last_slot = 0; continue; }
            if (e != hipSuccess) return fail_hip(c, e, "batch slot set-up");
        }
        BatchSlot &b = c->slot[s];
        hipStream_t st = b.stream;
        if (!wb) {
            hipError_t e = ensure_spill(c, b);
            if (e != hipSuccess) return fail_hip(c, e, "hipMalloc(stack overflow area)");
        }
        /* the primary pass of an earlier batch for the same camera, scene and shard is still good */
        const bool cached = !wb && !count && b.primary_epoch == c->view_epoch;
        bool by_hit_count = false;
        if (!wb) {
            const size_t worst = worst_per_frame * batch;
            if (worst > b.sample_slots) {
                /* counting contexts always take the worst case: their primary pass must run exactly once per batch */
                if (worst * sizeof(float4) <= c->small_sample_bytes || count) {
                    hipError_t e = ensure_sample_slots(b, worst);
                    if (e == hipErrorOutOfMemory && s == 1) { c->pipeline = false; may_pipeline = false; continue; }      /* no room for a second slot: go on with one */
                    if (e == hipErrorOutOfMemory && batch > 1) { max_batch = batch / 2; continue; }
                    if (e != hipSuccess) return fail_hip(c, e, "hipMalloc(samples)");
                } else {
                    by_hit_count = true;
                }
            }
        }
        PTKernelParams k;
        fill_kernel_params(c, b, first + done * stride, stride, batch, mode, k);
        if (done + batch == n_frames) { k.mirror = second_out; k.mirror_move = second_move ? 1u : 0u; }     /* the batch that ends the call */
        if (wb) { k.ctrl = wb->ctrl; k.pixel_hit = wb->pixel_hit; k.pixel_const = wb->pixel_const; k.hits = wb->hits; k.samples = wb->samples; }
        if (cached) {
            /* only the queue heads start again (the last resolve kernel has seen to that, unless a call failed half
             * way); hit list, hit count and per-pixel tables stay */
            if (!b.queue_clean) HIP_TRY(c, hipMemsetAsync(&k.ctrl->queue_head[0], 0, sizeof(k.ctrl->queue_head), st));
        } else {
            /* the control words must start at zero; everything else a batch reads it has written itself
             * (the hit list up to hit_count, pixel_hit of every pixel, pixel_const of miss pixels, one sample
             * per path), so stale contents of an earlier call are never seen */
            HIP_TRY(c, hipMemsetAsync(k.ctrl, 0, sizeof(PTControl), st));
            if (c->shard_world > 1)     /* pixels of other ranks: marked "not mine" (0xfefefefe), the primary kernel visits owned tiles only */
                HIP_TRY(c, hipMemsetAsync(k.pixel_hit, 0xfe, npix * sizeof(uint32_t), st));
        }
        if (wb) {
            HIP_TRY(c, pt_launch_primary(k, exact, count, false, st));
            HIP_TRY(c, pt_launch_paths(k, exact, count, c->num_cus, 0, st));
            HIP_TRY(c, accum_wait(c, s));
            HIP_TRY(c, pt_launch_resolve(k, st));
            HIP_TRY(c, accum_mark(c, s));
            HIP_TRY(c, hipEventRecord(c->batch_done[c->batch_seq & 1], st));
            c->batch_seq++;
            done += batch;
            continue;
        }
        bool copy_hit_count = false;
        if (!cached) {
            b.primary_epoch = 0;
            { Timed t(c, st, 0); HIP_TRY(c, pt_launch_primary(k, exact, count, false, st)); t.done(); }
            b.hit_count_pending = false;
            copy_hit_count = !count;        /* the tables count as kept once the copy of their hit count is enqueued */
        }
        if (by_hit_count) {
            /* a large render: size the buffer by the pixels the primary pass hit (waiting for that pass if its
             * count is not known yet) */
            if (copy_hit_count) {
                HIP_TRY(c, hipMemcpyAsync(b.h_hit_count, &b.d_ctrl->hit_count, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
                b.hit_count_pending = true; b.primary_epoch = c->view_epoch;
                copy_hit_count = false;
            }
            bool known = false;
            HIP_TRY(c, slot_hit_count(c, b, true, known));
            if (!known) return fail(c, PT_ESTATE, "internal: the hit count of the primary pass is not available");
            const size_t hits = b.hit_count;
            size_t need = ((hits + 63) / 64 * 64) * rpp * batch;
            if (need > b.sample_slots) {
                size_t want = need + need / 8;              /* headroom: a camera move changes the hit count */
                const size_t worst = worst_per_frame * batch;
                if (want > worst) want = worst;
                hipError_t e = ensure_sample_slots(b, want);
                if (e == hipErrorOutOfMemory && want > need) e = ensure_sample_slots(b, need);
                if (e == hipErrorOutOfMemory && s == 1) { c->pipeline = false; may_pipeline = false; c->last_slot = s; continue; }
                if (e == hipErrorOutOfMemory && batch > 1) {
                    /* start this batch again with half the frames; the primary pass that has just run is kept */
                    max_batch = batch / 2;
                    c->last_slot = s;
                    continue;
                }
                if (e != hipSuccess) return fail_hip(c, e, "hipMalloc(samples)");
            }
            k.samples = b.d_samples;
        }
        /* The size of the persistent grid, when the number of paths is known (a kept primary pass).  A render of
         * few paths gets a grid that gives every lane about eight of them: filling 1536 blocks costs more than
         * they are worth then.  A batch that will share the GPU with the batch before it (that one is still
         * running on the other stream) takes fewer blocks per CU the shorter it is: two grids side by side pass
         * the CUs to each other smoothly, and the shorter a kernel, the larger the part of its life its drain is.
         * Measured, 1080p, two renders in flight, ms per render with 6 / 4 / 3 / 2 blocks per CU: 16 M paths 2.33 / 2.30 /
         * 2.22 / 2.54, 33 M paths 9.39 / 9.10 / 9.22 / -; 271 M paths (6 / 5 / 4 / 3) 43.05 / 43.21 / 43.82 / 44.15. */
        int grid_cap = 0;
        if (c->adapt_grid && c->known_hits_epoch == c->view_epoch) {
            const unsigned long long paths = (unsigned long long)c->known_hits * rpp * batch;
            const unsigned long long lanes_per_block_row = (unsigned long long)c->num_cus * 256u;     /* one block on every CU */
            const unsigned long long full = lanes_per_block_row * (unsigned)pt_path_blocks_per_cu_max();
            if (paths < 8u * full) {
                grid_cap = (int)((paths + 8u * lanes_per_block_row - 1) / (8u * lanes_per_block_row));
                if (grid_cap < 1) grid_cap = 1;
            } else if (s != c->last_slot && c->slot[c->last_slot].stream && hipStreamQuery(c->slot[c->last_slot].stream) == hipErrorNotReady) {
                (void)hipGetLastError();
                grid_cap = paths < 64u * full ? 3 : paths < 256u * full ? 4 : 0;
            }
        }
        b.queue_clean = false;
        { Timed t(c, st, 1); HIP_TRY(c, pt_launch_paths(k, exact, count, c->num_cus, grid_cap, st)); t.done(); }
        HIP_TRY(c, accum_wait(c, s));
        { Timed t(c, st, 2); HIP_TRY(c, pt_launch_resolve(k, st)); t.done(); }
        b.queue_clean = true;
        HIP_TRY(c, accum_mark(c, s));
        if (copy_hit_count) {       /* behind the batch, where nothing waits for it */
            HIP_TRY(c, hipMemcpyAsync(b.h_hit_count, &b.d_ctrl->hit_count, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
            b.hit_count_pending = true; b.primary_epoch = c->view_epoch;
        }
        HIP_TRY(c, hipEventRecord(c->batch_done[c->batch_seq & 1], st));
        c->batch_seq++;
        c->last_slot = s;
        c->timing.frames += batch;
        c->timing.batches++;
        done += batch;
        if (c->pending.size() > 3000) {       /* bound the number of live events on long renders */
            HIP_TRY(c, sync_all(c));
            int rc = drain_events(c);
            if (rc) return rc;
        }
    }
    return PT_OK;
}

bool read_file(const char *path, std::vector<float> &out) {
    FILE *f = fopen(path, "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    if (n < 0 || n % 4) { fclose(f); return false; }
    out.resize((size_t)n / 4);
    size_t got = n ? fread(out.data(), 1, (size_t)n, f) : 0;
    fclose(f);
    return got == (size_t)n;
}

}  // namespace

extern "C" {

int pt_default_params(pt_params *p) {
    if (!p) return PT_EINVAL;
    memset(p, 0, sizeof(*p));
    p->width = 1200; p->height = 600;
    p->rays_per_pixel = 20;
    p->max_bounce_count = 10;
    p->fov_degrees = 60.0f;
    p->tri_color[0] = 0.65f; p->tri_color[1] = 0.65f; p->tri_color[2] = 1.0f;
    p->tri_emission_strength = 0.0f;
    p->tri_smoothness = 0.9f;
    p->sky_horizon[0] = 0.5f; p->sky_horizon[1] = 0.7f; p->sky_horizon[2] = 1.0f;
    p->sky_zenith[0] = 0.1f; p->sky_zenith[1] = 0.25f; p->sky_zenith[2] = 1.0f;
    p->ground[0] = p->ground[1] = p->ground[2] = 0.2f;
    p->sun_dir[0] = 0.0f; p->sun_dir[1] = -0.4f; p->sun_dir[2] = 0.5f;
    p->sun_intensity = 3.0f; p->sun_focus = 200.0f;
    p->use_environment_lighting = 1;
    p->far_distance = 1000000.0f;
    p->bvh_max_depth = 0;
    p->flags = 0;
    return PT_OK;
}

int pt_create(int device_ordinal, const pt_params *params, pt_ctx **out) {
    if (!out) return fail(nullptr, PT_EINVAL, "out is NULL");
    *out = nullptr;
    std::string why;
    int rc = validate_params(params, why);
    if (rc) return fail(nullptr, rc, why);
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess) return fail_hip(nullptr, e, "hipGetDeviceCount");
    if (device_ordinal < 0 || device_ordinal >= ndev) return fail(nullptr, PT_EINVAL, "device ordinal out of range");
    pt_ctx *c = new (std::nothrow) pt_ctx();
    if (!c) return fail(nullptr, PT_ENOMEM, "out of host memory");
    c->device = device_ordinal;
    c->params = *params;
    { std::lock_guard<std::mutex> lock(g_static_scratch_mutex); c->serial = g_next_serial++; }
    if (const char *e = getenv("PT_REFILL_MIN")) { int v = atoi(e); if (v >= 1 && v <= 64) c->refill_min = (uint32_t)v; }
    if (const char *e = getenv("PT_QUAD_FETCH")) c->quad_fetch = atoi(e) != 0;
    if (const char *e = getenv("PT_TWO_PATHS")) c->two_paths = atoi(e) != 0;
    if (c->two_paths && !getenv("PT_REFILL_MIN")) c->refill_min = 60;       /* that schedule shades at 60 waiting lanes */
    if (const char *e = getenv("PT_STARVE_MIN")) { int v = atoi(e); if (v >= 1 && v <= 64) c->starve_min = (uint32_t)v; }
    if (const char *e = getenv("PT_PIPELINE")) c->pipeline = atoi(e) != 0;
    if (const char *e = getenv("PT_ADAPT_GRID")) c->adapt_grid = atoi(e) != 0;
    if (const char *e = getenv("PT_NO_TIMING")) c->timing_events = atoi(e) == 0;
    if (const char *e = getenv("PT_PIPELINE_MB")) { long v = atol(e); if (v >= 0) c->pipeline_bytes = (size_t)v << 20; }
    if (const char *e = getenv("PT_LEAF_SHIFT")) { int v = atoi(e); if (v >= 0 && v <= 6) c->leaf_shift = (uint32_t)v; }
    if (const char *e = getenv("PT_LEAF_MIN")) { int v = atoi(e); if (v >= 1 && v <= 64) c->leaf_min = (uint32_t)v; }
    auto bail = [&](hipError_t he, const char *what) {
        int r = fail_hip(nullptr, he, what);
        pt_destroy(c);
        return r;
    };
    if ((e = hipSetDevice(device_ordinal)) != hipSuccess) return bail(e, "hipSetDevice");
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device_ordinal)) != hipSuccess) return bail(e, "hipGetDeviceProperties");
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        { fail(nullptr, PT_EHIP, std::string("unsupported GPU architecture ") + prop.gcnArchName + " (built for gfx950)"); pt_destroy(c); return PT_EHIP; }
    c->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if ((e = static_scratch_lookup(device_ordinal)) != hipSuccess) return bail(e, "hipGetSymbolAddress(scratch)");
    if ((e = slot_prepare(c, 0)) != hipSuccess) return bail(e, "batch slot set-up (stream, hit list, per-pixel tables)");
    c->stream = c->slot[0].stream;
    for (hipEvent_t &ev : c->batch_done)
        if ((e = hipEventCreateWithFlags(&ev, hipEventDisableTiming)) != hipSuccess) return bail(e, "hipEventCreate");
    if ((e = hipEventCreateWithFlags(&c->ext_done, hipEventDisableTiming)) != hipSuccess) return bail(e, "hipEventCreate");
    const size_t npix = (size_t)params->width * params->height;
    if ((e = hipMalloc(&c->d_accum_own, npix * sizeof(float4))) != hipSuccess) return bail(e, "hipMalloc(accum)");
    if ((e = hipMemsetAsync(c->d_accum_own, 0, npix * sizeof(float4), c->stream)) != hipSuccess) return bail(e, "hipMemset(accum)");
    c->d_accum = c->d_accum_own;
    /* The sample buffer (one float4 per path of a batch of frames) is allocated by the render calls, for the
     * frames actually asked for; here only its limits are fixed: at most PT_SAMPLE_BUDGET_MB (default 64 GiB
     * of the 288 GB) and 64 frames per batch, and slot numbers must fit 32 bits. */
    {
        size_t budget = (size_t)64 << 30;
        if (const char *ev = getenv("PT_SAMPLE_BUDGET_MB")) { long v = atol(ev); if (v > 0) budget = (size_t)v << 20; }
        c->sample_budget = budget;
        if (const char *ev = getenv("PT_SMALL_SAMPLE_MB")) { long v = atol(ev); if (v >= 0) c->small_sample_bytes = (size_t)v << 20; }
        /* hit pixels are stored in interleaved groups of 64; the path kernel's queue arithmetic (PT_QUEUE_PARTS equal
         * parts of whole pixels) stays below (hit pixels + 512) * frames * rays_per_pixel */
        const size_t slots_per_frame = ((npix + 63) / 64 * 64 + 512) * params->rays_per_pixel;
        size_t nb = 0xffffffffull / slots_per_frame;
        if (nb < 1) { fail(nullptr, PT_EINVAL, "width * height * rays_per_pixel does not fit the 32-bit sample numbering"); pt_destroy(c); return PT_EINVAL; }
        if (nb > 64) nb = 64;
        if (const char *ev = getenv("PT_MAX_BATCH")) { long v = atol(ev); if (v >= 1 && (size_t)v < nb) nb = (size_t)v; }
        c->batch_cap = (uint32_t)nb;
    }
    if (c->two_paths && (e = hipMalloc(&c->d_pathbuf, pt_pathbuf_bytes(c->num_cus))) != hipSuccess) return bail(e, "hipMalloc(pathbuf)");
    if ((e = hipMalloc(&c->d_counters, 24 * sizeof(unsigned long long))) != hipSuccess) return bail(e, "hipMalloc(counters)");
    if ((e = hipMemsetAsync(c->d_counters, 0, 24 * sizeof(unsigned long long), c->stream)) != hipSuccess) return bail(e, "hipMemset(counters)");
    if ((e = sync_all(c)) != hipSuccess) return bail(e, "hipStreamSynchronize");
    *out = c;
    return PT_OK;
}

int pt_upload_scene(pt_ctx *c, const float *spheres, uint32_t sphere_count,
                    const float *triangles, uint32_t triangle_count,
                    const float *bvh, uint32_t node_count) {
    if (!c) return PT_EINVAL;
    if (sphere_count && !spheres) return fail(c, PT_EINVAL, "spheres is NULL");
    if (triangle_count && !triangles) return fail(c, PT_EINVAL, "triangles is NULL");
    if (!bvh || node_count == 0) return fail(c, PT_EINVAL, "a BVH with at least one node is required");
    if (triangle_count > (1u << 24)) return fail(c, PT_EINVAL, "more than 2^24 triangles (indices are f32 in the node file)");
    if (sphere_count > 65536) return fail(c, PT_EINVAL, "too many spheres");
    Repacked rp;
    std::string why;
    if (!repack_nodes(bvh, node_count, triangle_count, rp, why)) return fail(c, PT_EINVAL, "malformed BVH file: " + why);
    /* The shader needs array<i32, bvh_max_depth + 1> (shader.wgsl:170).  Here the near child is
     * never pushed, so at most one entry per level is live: depth entries (at least 1). */
    int need = rp.depth > 0 ? (int)rp.depth : 1;
    if (c->params.bvh_max_depth != 0 && rp.depth > c->params.bvh_max_depth)
        return fail(c, PT_EINVAL, "node file is " + std::to_string(rp.depth) + " levels deep, pt_params.bvh_max_depth is " +
                    std::to_string(c->params.bvh_max_depth) + " (the shader's stack of bvh_max_depth + 1 entries would overflow)");
    if (need > pt_max_stack_entries())
        return fail(c, PT_EINVAL, "BVH depth " + std::to_string(rp.depth) + " exceeds the supported traversal stack");
    std::vector<PTTriRecord> trec;
    build_tri_records(triangles, triangle_count, trec);
    WideTree wt;
    int wide_need = 0;
    const bool wide = (c->params.flags & PT_FLAG_WIDE_BVH) != 0;
    if (wide) {
        build_wide(rp, wt);
        /* a visit keeps the nearest child and pushes up to three: three live entries per level */
        wide_need = wt.depth > 0 ? 3 * (int)wt.depth : 1;
        if (wide_need > pt_max_stack_entries())
            return fail(c, PT_EINVAL, "four-wide tree of depth " + std::to_string(wt.depth) + " exceeds the supported traversal stack");
    }

    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, sync_all(c));
    c->scene_loaded = false;
    c->view_epoch++;
    dev_free(c->d_inner); dev_free(c->d_tris); dev_free(c->d_spheres); dev_free(c->d_big);
    for (BatchSlot &b : c->slot) { dev_free(b.d_spill); b.spill_words = 0; }
    dev_free(c->d_wide); dev_free(c->d_refit_order); dev_free(c->d_tri9);
    c->refit_ok = false; c->refit_level_begin.clear();
    c->wide_count = 0;
    if (wide && !wt.nodes.empty()) {
        HIP_TRY(c, hipMalloc(&c->d_wide, wt.nodes.size() * sizeof(PTWideRecord)));
        HIP_TRY(c, hipMemcpy(c->d_wide, wt.nodes.data(), wt.nodes.size() * sizeof(PTWideRecord), hipMemcpyHostToDevice));
        c->wide_count = (uint32_t)wt.nodes.size();
    }
    c->wide_root_ref = wt.root_ref; c->wide_depth = wt.depth; c->wide_stack_entries = wide_need;
    size_t n_inner = rp.inner.size() ? rp.inner.size() : 1;
    HIP_TRY(c, hipMalloc(&c->d_inner, n_inner * sizeof(PTInnerRecord)));
    if (!rp.inner.empty())
        HIP_TRY(c, hipMemcpy(c->d_inner, rp.inner.data(), rp.inner.size() * sizeof(PTInnerRecord), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMalloc(&c->d_tris, trec.size() * sizeof(PTTriRecord)));
    HIP_TRY(c, hipMemcpy(c->d_tris, trec.data(), trec.size() * sizeof(PTTriRecord), hipMemcpyHostToDevice));
    size_t n_sph = sphere_count ? sphere_count : 1;
    HIP_TRY(c, hipMalloc(&c->d_spheres, n_sph * 12 * sizeof(float)));
    if (sphere_count)
        HIP_TRY(c, hipMemcpy(c->d_spheres, spheres, (size_t)sphere_count * 12 * sizeof(float), hipMemcpyHostToDevice));
    {   /* overflow area of the path kernel's traversal stacks (only trees deeper than its LDS part); the second
         * slot's is made by the first batch that uses that slot */
        c->spill_words = (size_t)pt_spill_entries((uint32_t)(wide_need > need ? wide_need : need)) * pt_lane_stride(c->num_cus);
        hipError_t es = ensure_spill(c, c->slot[0]);
        if (es != hipSuccess) return fail_hip(c, es, "hipMalloc(stack overflow area)");
    }
    if (rp.big.empty()) rp.big.assign(1, 0);
    HIP_TRY(c, hipMalloc(&c->d_big, rp.big.size() * sizeof(uint32_t)));
    HIP_TRY(c, hipMemcpy(c->d_big, rp.big.data(), rp.big.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    c->sphere_count = sphere_count; c->tri_count = triangle_count; c->node_count = node_count;
    c->root_ref = rp.root_ref; c->depth = rp.depth;
    c->inner_count = (uint32_t)rp.inner.size(); c->leaf_count = rp.leaves;
    c->stack_entries = need;
    c->fast_div = rp.tame_bounds && getenv("PT_NO_FAST_DIV") == nullptr;
    c->refit_ok = rp.refit_ok; c->refit_why = rp.refit_why;
    c->bytes_inner = n_inner * sizeof(PTInnerRecord); c->bytes_tris = trec.size() * sizeof(PTTriRecord);
    c->bytes_big = rp.big.size() * sizeof(uint32_t); c->bytes_refit_order = 0;
    if (c->refit_ok && !rp.inner.empty()) {
        /* counting sort of the records by height: a refit walks the heights upwards */
        uint32_t top = 0;
        for (uint32_t h : rp.height) if (h > top) top = h;
        std::vector<uint32_t> begin(top + 2, 0), order(rp.inner.size());
        for (uint32_t h : rp.height) begin[h + 1]++;
        for (uint32_t h = 1; h < top + 2; h++) begin[h] += begin[h - 1];
        std::vector<uint32_t> fill(begin.begin(), begin.end() - 1);
        for (uint32_t r = 0; r < rp.inner.size(); r++) order[fill[rp.height[r]]++] = r;
        HIP_TRY(c, hipMalloc(&c->d_refit_order, order.size() * sizeof(uint32_t)));
        HIP_TRY(c, hipMemcpy(c->d_refit_order, order.data(), order.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        c->bytes_refit_order = order.size() * sizeof(uint32_t);
        c->refit_level_begin = begin;
    }
    c->scene_loaded = true;
    return PT_OK;
}

int pt_rebuild_scene(pt_ctx *c, const float *triangles, uint32_t triangle_count, int max_depth, float *build_ms) {
    if (!c) return PT_EINVAL;
    if (build_ms) *build_ms = 0.0f;
    if (!triangles || triangle_count == 0) return fail(c, PT_EINVAL, "pt_rebuild_scene needs at least one triangle");
    if (c->params.flags & PT_FLAG_WIDE_BVH) return fail(c, PT_ESTATE, "pt_rebuild_scene does not build the four-wide tree (PT_FLAG_WIDE_BVH)");
    PTResidentTree t;
    std::string why;
    int rc = pt_gpu_build_resident(c->device, triangles, triangle_count, max_depth, t, why);
    if (rc) return fail(c, rc, "pt_rebuild_scene: " + why);
    const int need = t.depth > 0 ? (int)t.depth : 1;
    if (c->params.bvh_max_depth != 0 && t.depth > c->params.bvh_max_depth) {
        pt_resident_tree_free(t);
        return fail(c, PT_EINVAL, "the tree is " + std::to_string(t.depth) + " levels deep, pt_params.bvh_max_depth is " + std::to_string(c->params.bvh_max_depth));
    }
    if (need > pt_max_stack_entries()) {
        pt_resident_tree_free(t);
        return fail(c, PT_EINVAL, "BVH depth " + std::to_string(t.depth) + " exceeds the supported traversal stack");
    }
    /* the builder has put the caller's device back: what follows allocates on the context's */
    hipError_t e = hipSetDevice(c->device);
    if (e == hipSuccess) e = sync_all(c);
    uint32_t *spill = nullptr;
    float *sph = c->d_spheres;
    const size_t spill_words = (size_t)pt_spill_entries((uint32_t)need) * pt_lane_stride(c->num_cus);
    if (e == hipSuccess) e = hipMalloc(&spill, (spill_words ? spill_words : 1) * sizeof(uint32_t));
    if (e == hipSuccess && !sph) e = hipMalloc(&sph, 12 * sizeof(float));      /* no scene before: no spheres */
    if (e != hipSuccess) {
        pt_resident_tree_free(t);
        if (spill) (void)hipFree(spill);
        return fail_hip(c, e, "pt_rebuild_scene");
    }
    /* the old scene's arrays go to the builder's cache: the next rebuild of a scene of this size allocates nothing */
    pt_scratch_give_back(c->device, c->d_inner, c->bytes_inner); pt_scratch_give_back(c->device, c->d_tris, c->bytes_tris);
    pt_scratch_give_back(c->device, c->d_big, c->bytes_big); pt_scratch_give_back(c->device, c->d_refit_order, c->bytes_refit_order);
    c->d_inner = nullptr; c->d_tris = nullptr; c->d_big = nullptr; c->d_refit_order = nullptr;
    for (BatchSlot &b : c->slot) { dev_free(b.d_spill); b.spill_words = 0; }
    dev_free(c->d_tri9);
    c->spill_words = spill_words; c->slot[0].d_spill = spill; c->slot[0].spill_words = spill_words ? spill_words : 1;
    c->view_epoch++;
    c->bytes_inner = t.bytes_inner; c->bytes_tris = t.bytes_tris; c->bytes_big = t.bytes_big; c->bytes_refit_order = t.bytes_refit_order;
    c->d_inner = t.d_inner; c->d_tris = t.d_tris; c->d_big = t.d_big; c->d_refit_order = t.d_refit_order; c->d_spheres = sph;
    c->refit_level_begin = t.refit_level_begin;
    c->refit_ok = true; c->refit_why.clear();
    c->tri_count = triangle_count; c->node_count = t.node_count;
    c->root_ref = t.root_ref; c->depth = t.depth; c->inner_count = t.inner_count; c->leaf_count = t.leaf_count;
    c->stack_entries = need;
    c->fast_div = t.tame && getenv("PT_NO_FAST_DIV") == nullptr;
    c->scene_loaded = true;
    if (build_ms) *build_ms = t.build_ms;
    return PT_OK;
}

int pt_refit_scene(pt_ctx *c, const float *triangles) {
    if (!c) return PT_EINVAL;
    if (!c->scene_loaded) return fail(c, PT_ESTATE, "no scene uploaded");
    if (c->tri_count && !triangles) return fail(c, PT_EINVAL, "triangles is NULL");
    if (c->d_wide) return fail(c, PT_ESTATE, "pt_refit_scene does not update the four-wide tree (PT_FLAG_WIDE_BVH)");
    if (!c->refit_ok) return fail(c, PT_EINVAL, "this node file cannot be refitted: " + c->refit_why);
    const size_t n_floats = 9 * (size_t)c->tri_count;
    for (size_t i = 0; i < n_floats; i++) {
        uint32_t u;
        memcpy(&u, triangles + i, 4);
        if ((u & 0x7f800000u) == 0x7f800000u) return fail(c, PT_EINVAL, "a triangle coordinate is not finite");
    }
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, sync_all(c));            /* batches in flight on either stream read the records this call rewrites */
    c->view_epoch++;
    if (!c->d_tri9) HIP_TRY(c, hipMalloc(&c->d_tri9, (n_floats ? n_floats : 1) * sizeof(float)));
    if (!c->d_refit_flag) HIP_TRY(c, hipMalloc(&c->d_refit_flag, sizeof(uint32_t)));
    HIP_TRY(c, hipMemcpyAsync(c->d_tri9, triangles, n_floats * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemsetAsync(c->d_refit_flag, 0, sizeof(uint32_t), c->stream));
    HIP_TRY(c, pt_launch_refit_tris(c->d_tri9, c->tri_count, c->d_tris, c->stream));
    for (size_t h = 1; h + 1 < c->refit_level_begin.size(); h++)
        HIP_TRY(c, pt_launch_refit_level(c->d_inner, c->d_refit_order, c->refit_level_begin[h],
                                         c->refit_level_begin[h + 1] - c->refit_level_begin[h], c->d_tri9, c->d_big, c->d_refit_flag, c->stream));
    uint32_t not_tame = 0;
    HIP_TRY(c, hipMemcpyAsync(&not_tame, c->d_refit_flag, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, sync_all(c));
    c->fast_div = !not_tame && getenv("PT_NO_FAST_DIV") == nullptr;
    return PT_OK;
}

int pt_load_scene_files(pt_ctx *c, const char *triangles_bin, const char *bvh_bin,
                        const float *spheres, uint32_t sphere_count) {
    if (!c) return PT_EINVAL;
    if (!triangles_bin || !bvh_bin) return fail(c, PT_EINVAL, "file name is NULL");
    std::vector<float> tri, bvh;
    if (!read_file(triangles_bin, tri)) return fail(c, PT_EIO, std::string("cannot read ") + triangles_bin);
    if (!read_file(bvh_bin, bvh)) return fail(c, PT_EIO, std::string("cannot read ") + bvh_bin);
    if (tri.size() % 9) return fail(c, PT_EINVAL, "triangle file is not a multiple of 36 bytes");
    if (bvh.size() % 9 || bvh.empty()) return fail(c, PT_EINVAL, "BVH file is not a non-empty multiple of 36 bytes");
    return pt_upload_scene(c, spheres, sphere_count, tri.data(), (uint32_t)(tri.size() / 9),
                           bvh.data(), (uint32_t)(bvh.size() / 9));
}

int pt_set_camera(pt_ctx *c, const float position[3], const float rotation_degrees[3]) {
    if (!c) return PT_EINVAL;
    if (!position || !rotation_degrees) return fail(c, PT_EINVAL, "camera pointer is NULL");
    bool moved = false;      /* compared as bits: a NaN or a change of a zero's sign counts as a move */
    for (int i = 0; i < 3; i++)
        moved = moved || memcmp(&c->cam_pos[i], &position[i], sizeof(float)) != 0 || memcmp(&c->cam_rot[i], &rotation_degrees[i], sizeof(float)) != 0;
    for (int i = 0; i < 3; i++) { c->cam_pos[i] = position[i]; c->cam_rot[i] = rotation_degrees[i]; }
    if (moved) c->view_epoch++;       /* the primary pass of the next batch runs again */
    return PT_OK;
}

int pt_render_frames(pt_ctx *c, uint32_t first_frame_count, uint32_t n_frames) {
    return render_impl(c, first_frame_count, n_frames, 1, 0);
}

int pt_render_frames_static(pt_ctx *c, uint32_t first_frame_count, uint32_t n_frames) {
    if (!c) return PT_EINVAL;
    if (!c->scene_loaded) return fail(c, PT_ESTATE, "no scene uploaded");
    WorkBuffers wb;
    int rc = static_work_buffers(c, n_frames, wb);
    if (rc) return rc;
    /* All contexts of a GPU share the array.  Calls of ONE context are ordered by its stream; a call that finds
     * another context's static render still running on this GPU is refused instead of racing with it. */
    std::lock_guard<std::mutex> lock(g_static_scratch_mutex);
    HIP_TRY(c, hipSetDevice(c->device));
    if (g_static_owner[c->device] != 0 && g_static_owner[c->device] != c->serial) {
        hipError_t q = hipEventQuery(g_static_done[c->device]);
        if (q == hipErrorNotReady) {
            (void)hipGetLastError();
            return fail(c, PT_ESTATE, "pt_render_frames_static: another context's static render is still running on this GPU "
                                      "(they share one scratch array; synchronise that context first)");
        }
        if (q != hipSuccess) return fail_hip(c, q, "hipEventQuery(static scratch guard)");
    }
    rc = render_impl(c, first_frame_count, n_frames, 1, 0, &wb);
    /* whatever was enqueued (even by a call that failed half way) uses the array until this point of the stream */
    g_static_owner[c->device] = c->serial;
    hipError_t e = hipEventRecord(g_static_done[c->device], c->stream);
    if (rc == PT_OK && e != hipSuccess) return fail_hip(c, e, "hipEventRecord(static scratch guard)");
    return rc;
}

size_t pt_static_scratch_bytes(void) { return (size_t)PT_STATIC_SCRATCH_BYTES; }

int pt_render_frames_sum(pt_ctx *c, uint32_t first_frame_count, uint32_t n_frames, uint32_t frame_stride) {
    if (c && frame_stride == 0) return fail(c, PT_EINVAL, "frame_stride must be >= 1");
    return render_impl(c, first_frame_count, n_frames, frame_stride, 1);
}

int pt_set_pixel_shard(pt_ctx *c, uint32_t rank, uint32_t world_size) {
    if (!c) return PT_EINVAL;
    if (world_size == 0 || rank >= world_size || world_size > 4096) return fail(c, PT_EINVAL, "need 0 <= rank < world_size <= 4096");
    if (c->shard_rank != rank || c->shard_world != world_size) c->view_epoch++;
    c->shard_rank = rank; c->shard_world = world_size;
    return PT_OK;
}

int pt_scale_accum(pt_ctx *c, float s) {
    if (!c) return PT_EINVAL;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, accum_wait(c, 0));
    HIP_TRY(c, pt_launch_scale(c->d_accum, c->params.width * c->params.height, s, c->stream));
    HIP_TRY(c, accum_mark(c, 0));
    return PT_OK;
}

int pt_read_accum(pt_ctx *c, float *out, uint32_t stride_floats) {
    if (!c) return PT_EINVAL;
    if (!out || (stride_floats != 3 && stride_floats != 4)) return fail(c, PT_EINVAL, "out is NULL or stride_floats is not 3 or 4");
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t npix = (size_t)c->params.width * c->params.height;
    HIP_TRY(c, sync_all(c));
    if (stride_floats == 4) {
        HIP_TRY(c, hipMemcpy(out, c->d_accum, npix * sizeof(float4), hipMemcpyDeviceToHost));
    } else {
        std::vector<float> tmp(npix * 4);
        HIP_TRY(c, hipMemcpy(tmp.data(), c->d_accum, npix * sizeof(float4), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < npix; i++) { out[3 * i] = tmp[4 * i]; out[3 * i + 1] = tmp[4 * i + 1]; out[3 * i + 2] = tmp[4 * i + 2]; }
    }
    return PT_OK;
}

int pt_write_accum(pt_ctx *c, const float *in, uint32_t stride_floats) {
    if (!c) return PT_EINVAL;
    if (!in || (stride_floats != 3 && stride_floats != 4)) return fail(c, PT_EINVAL, "in is NULL or stride_floats is not 3 or 4");
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t npix = (size_t)c->params.width * c->params.height;
    HIP_TRY(c, sync_all(c));
    if (stride_floats == 4) {
        HIP_TRY(c, hipMemcpy(c->d_accum, in, npix * sizeof(float4), hipMemcpyHostToDevice));
    } else {
        std::vector<float> tmp(npix * 4, 0.0f);
        for (size_t i = 0; i < npix; i++) { tmp[4 * i] = in[3 * i]; tmp[4 * i + 1] = in[3 * i + 1]; tmp[4 * i + 2] = in[3 * i + 2]; }
        HIP_TRY(c, hipMemcpy(c->d_accum, tmp.data(), npix * sizeof(float4), hipMemcpyHostToDevice));
    }
    return PT_OK;
}

int pt_reset_accum(pt_ctx *c) {
    if (!c) return PT_EINVAL;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, accum_wait(c, 0));
    HIP_TRY(c, hipMemsetAsync(c->d_accum, 0, (size_t)c->params.width * c->params.height * sizeof(float4), c->stream));
    HIP_TRY(c, accum_mark(c, 0));
    return PT_OK;
}

/* Hands what a stream holds to the GPU without waiting for it.  A wait or a record alone may stay in the runtime's
 * batch of the stream until the next launch or query on it; a host that then blocks on something other than a HIP call
 * (a collective's completion, say) would wait for work the GPU has never been given. */
static void flush_stream(hipStream_t s) {
    if (hipStreamQuery(s) != hipSuccess) (void)hipGetLastError();
}

/* an image buffer of the caller's: 16-byte aligned device memory of the context's GPU */
static int check_image_pointer(pt_ctx *c, void *device_ptr) {
    if (((uintptr_t)device_ptr & 15u) != 0) return fail(c, PT_EINVAL, "image buffer must be 16-byte aligned");
    hipPointerAttribute_t attr;
    hipError_t e = hipPointerGetAttributes(&attr, device_ptr);
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(c, PT_EINVAL, "pointer is not known to the HIP runtime"); }
    if (attr.type != hipMemoryTypeDevice || attr.device != c->device)
        return fail(c, PT_EINVAL, "pointer is not device memory of this context's GPU");
    return PT_OK;
}

int pt_bind_accum(pt_ctx *c, void *device_ptr) {
    if (!c) return PT_EINVAL;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, sync_all(c));
    if (!device_ptr) { c->d_accum = c->d_accum_own; return PT_OK; }
    int rc = check_image_pointer(c, device_ptr);
    if (rc) return rc;
    c->d_accum = (float4 *)device_ptr;
    return PT_OK;
}

int pt_next_render_output(pt_ctx *c, void *device_ptr, uint32_t move) {
    if (!c) return PT_EINVAL;
    if (!device_ptr) { c->next_out = nullptr; c->next_move = false; return PT_OK; }
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = check_image_pointer(c, device_ptr);
    if (rc) return rc;
    if (device_ptr == (void *)c->d_accum) return fail(c, PT_EINVAL, "the second output is the accumulation buffer itself");
    c->next_out = (float4 *)device_ptr;
    c->next_move = move != 0;
    return PT_OK;
}

void *pt_accum_device_ptr(pt_ctx *c) { return c ? (void *)c->d_accum : nullptr; }

int pt_accum_release(pt_ctx *c, void *hip_stream) {
    if (!c) return PT_EINVAL;
    HIP_TRY(c, hipSetDevice(c->device));
    /* every operation on the buffer waits for the one before it, so the last mark stands for all of them */
    if (c->accum_last >= 0) {
        HIP_TRY(c, hipStreamWaitEvent((hipStream_t)hip_stream, c->slot[c->accum_last].accum_done, 0));
        flush_stream(c->slot[c->accum_last].stream);
        flush_stream((hipStream_t)hip_stream);
    }
    return PT_OK;
}

int pt_accum_acquire(pt_ctx *c, void *hip_stream) {
    if (!c) return PT_EINVAL;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipEventRecord(c->ext_done, (hipStream_t)hip_stream));
    flush_stream((hipStream_t)hip_stream);
    c->ext_pending = true;
    return PT_OK;
}

int pt_primary_hits(pt_ctx *c, int32_t *prim_id, float *distance) {
    if (!c) return PT_EINVAL;
    if (!prim_id || !distance) return fail(c, PT_EINVAL, "output pointer is NULL");
    if (!c->scene_loaded) return fail(c, PT_ESTATE, "no scene uploaded");
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t npix = (size_t)c->params.width * c->params.height;
    if (!c->d_ids) HIP_TRY(c, hipMalloc(&c->d_ids, npix * sizeof(int32_t)));
    if (!c->d_dist) HIP_TRY(c, hipMalloc(&c->d_dist, npix * sizeof(float)));
    PTKernelParams k;
    fill_kernel_params(c, c->slot[0], 0, 1, 1, 0, k);
    HIP_TRY(c, pt_launch_primary(k, true, false, true, c->stream));
    HIP_TRY(c, sync_all(c));
    HIP_TRY(c, hipMemcpy(prim_id, c->d_ids, npix * sizeof(int32_t), hipMemcpyDeviceToHost));
    HIP_TRY(c, hipMemcpy(distance, c->d_dist, npix * sizeof(float), hipMemcpyDeviceToHost));
    return PT_OK;
}

int pt_probe_functions(pt_ctx *c, const uint32_t *seeds, const float *dirs, const float normal[3], uint32_t n, float *out) {
    if (!c) return PT_EINVAL;
    if (!seeds || !dirs || !normal || !out) return fail(c, PT_EINVAL, "argument is NULL");
    if (n == 0) return PT_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    uint32_t *d_seeds = nullptr; float *d_dirs = nullptr, *d_out = nullptr;
    int rc = PT_OK;
    hipError_t e;
    if ((e = hipMalloc(&d_seeds, (size_t)n * 4)) != hipSuccess || (e = hipMalloc(&d_dirs, (size_t)n * 12)) != hipSuccess ||
        (e = hipMalloc(&d_out, (size_t)n * 64)) != hipSuccess ||
        (e = hipMemcpy(d_seeds, seeds, (size_t)n * 4, hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(d_dirs, dirs, (size_t)n * 12, hipMemcpyHostToDevice)) != hipSuccess) {
        rc = fail(c, PT_EHIP, hipGetErrorString(e));
    } else {
        PTKernelParams k;
        fill_kernel_params(c, c->slot[0], 0, 1, 1, 0, k);
        if ((e = pt_launch_probe(k, (c->params.flags & PT_FLAG_EXACT_MATH) != 0, d_seeds, d_dirs, normal, n, d_out, c->stream)) != hipSuccess ||
            (e = sync_all(c)) != hipSuccess ||
            (e = hipMemcpy(out, d_out, (size_t)n * 64, hipMemcpyDeviceToHost)) != hipSuccess)
            rc = fail(c, PT_EHIP, hipGetErrorString(e));
    }
    dev_free(d_seeds); dev_free(d_dirs); dev_free(d_out);
    return rc;
}

int pt_probe_intersections(pt_ctx *c, const float *rays, const float *boxes, const float *triangles, const float *spheres,
                           uint32_t n, float *out) {
    if (!c) return PT_EINVAL;
    if (!rays || !boxes || !triangles || !spheres || !out) return fail(c, PT_EINVAL, "argument is NULL");
    if (n == 0) return PT_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t sizes[5] = {(size_t)n * 24, (size_t)n * 24, (size_t)n * 36, (size_t)n * 16, (size_t)n * 32};
    const float *src[4] = {rays, boxes, triangles, spheres};
    float *d[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    int rc = PT_OK;
    hipError_t e = hipSuccess;
    for (int k = 0; k < 5 && e == hipSuccess; k++) e = hipMalloc(&d[k], sizes[k]);
    for (int k = 0; k < 4 && e == hipSuccess; k++) e = hipMemcpy(d[k], src[k], sizes[k], hipMemcpyHostToDevice);
    if (e == hipSuccess) e = pt_launch_probe_intersections(d[0], d[1], d[2], d[3], n, d[4], c->stream);
    if (e == hipSuccess) e = sync_all(c);
    if (e == hipSuccess) e = hipMemcpy(out, d[4], sizes[4], hipMemcpyDeviceToHost);
    if (e != hipSuccess) rc = fail(c, PT_EHIP, hipGetErrorString(e));
    for (int k = 0; k < 5; k++) dev_free(d[k]);
    return rc;
}

int pt_synchronize(pt_ctx *c) {
    if (!c) return PT_EINVAL;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, sync_all(c));
    return PT_OK;
}

int pt_wait_in_flight(pt_ctx *c, uint32_t max_in_flight) {
    if (!c) return PT_EINVAL;
    HIP_TRY(c, hipSetDevice(c->device));
    if (max_in_flight == 0) { HIP_TRY(c, sync_all(c)); return PT_OK; }
    if (max_in_flight >= 2) return PT_OK;
    /* the batch before the last one (and with it, by the order of the resolve kernels, every batch before that) */
    if (c->batch_seq >= 2) HIP_TRY(c, hipEventSynchronize(c->batch_done[c->batch_seq & 1]));
    return PT_OK;
}

int pt_get_counters(pt_ctx *c, pt_counters *out) {
    if (!c) return PT_EINVAL;
    if (!out) return fail(c, PT_EINVAL, "out is NULL");
    if (!(c->params.flags & PT_FLAG_COUNT_WORK)) return fail(c, PT_ESTATE, "context was created without PT_FLAG_COUNT_WORK");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, sync_all(c));
    unsigned long long h[8];
    HIP_TRY(c, hipMemcpy(h, c->d_counters, sizeof(h), hipMemcpyDeviceToHost));
    out->ray_casts = h[0]; out->box_tests = h[1]; out->tri_tests = h[2]; out->paths = h[3];
    out->path_kernel_ray_casts = h[4]; out->path_kernel_box_tests = h[5]; out->path_kernel_tri_tests = h[6];
    return PT_OK;
}

int pt_get_debug_stats(pt_ctx *c, uint64_t out[16]) {
    if (!c) return PT_EINVAL;
    if (!out) return fail(c, PT_EINVAL, "out is NULL");
    if (!(c->params.flags & PT_FLAG_COUNT_WORK)) return fail(c, PT_ESTATE, "context was created without PT_FLAG_COUNT_WORK");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, sync_all(c));
    unsigned long long h[24];
    HIP_TRY(c, hipMemcpy(h, c->d_counters, sizeof(h), hipMemcpyDeviceToHost));
    for (int i = 0; i < 8; i++) out[i] = h[8 + i];
    out[8] = h[16];                 /* longest wave lifetime */
    out[9] = h[17];                 /* longest time a wave went on after it found the queue empty */
    out[10] = h[20];                /* that time summed over waves */
    {   /* device real-time clock (100 MHz): from the earliest wave start to the first "queue empty" and to the last wave end */
        const unsigned long long t0 = (1ull << 62) - h[21];
        out[11] = h[21] ? (1ull << 62) - h[22] - t0 : 0;
        out[12] = h[21] ? h[23] - t0 : 0;
    }
    for (int i = 13; i < 16; i++) out[i] = 0;
    return PT_OK;
}

int pt_reset_counters(pt_ctx *c) {
    if (!c) return PT_EINVAL;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemsetAsync(c->d_counters, 0, 24 * sizeof(unsigned long long), c->stream));
    return PT_OK;
}

int pt_get_timing(pt_ctx *c, pt_timing *out) {
    if (!c) return PT_EINVAL;
    if (!out) return fail(c, PT_EINVAL, "out is NULL");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, sync_all(c));
    int rc = drain_events(c);
    if (rc) return rc;
    *out = c->timing;
    return PT_OK;
}

int pt_reset_timing(pt_ctx *c) {
    if (!c) return PT_EINVAL;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, sync_all(c));
    int rc = drain_events(c);
    if (rc) return rc;
    c->timing = pt_timing{};
    return PT_OK;
}

int pt_scene_info(pt_ctx *c, uint32_t *bvh_depth, uint32_t *inner_nodes, uint32_t *leaves) {
    if (!c) return PT_EINVAL;
    if (!c->scene_loaded) return fail(c, PT_ESTATE, "no scene uploaded");
    if (bvh_depth) *bvh_depth = c->depth;
    if (inner_nodes) *inner_nodes = c->inner_count;
    if (leaves) *leaves = c->leaf_count;
    return PT_OK;
}

void pt_destroy(pt_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)sync_all(c);
    for (EventPair &ep : c->pending) { (void)hipEventDestroy(ep.start); (void)hipEventDestroy(ep.stop); }
    for (EventPair &ep : c->free_events) { (void)hipEventDestroy(ep.start); (void)hipEventDestroy(ep.stop); }
    dev_free(c->d_inner); dev_free(c->d_tris); dev_free(c->d_spheres); dev_free(c->d_big); dev_free(c->d_wide);
    dev_free(c->d_accum_own);
    dev_free(c->d_counters); dev_free(c->d_ids); dev_free(c->d_dist);
    dev_free(c->d_pathbuf);
    dev_free(c->d_refit_order); dev_free(c->d_tri9); dev_free(c->d_refit_flag);
    for (BatchSlot &b : c->slot) slot_release(b);
    for (hipEvent_t &ev : c->batch_done) if (ev) { (void)hipEventDestroy(ev); ev = nullptr; }
    if (c->ext_done) { (void)hipEventDestroy(c->ext_done); c->ext_done = nullptr; }
    c->stream = nullptr;
    delete c;
}

const char *pt_last_error(pt_ctx *c) {
    return c ? c->err.c_str() : g_create_error.c_str();
}

}  // extern "C"
