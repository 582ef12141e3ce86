/*
 * Refit on the GPU: the triangles of an uploaded scene moved, the tree keeps its shape.
 *
 * The reference has no such step (its tree is built offline, prepare_data/src/main.rs); what a refit must
 * produce is fixed by the node file's own rule: a node's box is the tight bounds of the triangles in its
 * range (bounding_box.rs:34-44 applied to the node's triangles, main.rs:16-20, :108-117).  The kernels
 * rewrite, in device memory and without a host copy of the tree,
 *   - the triangle records (a, ab, ac, n: the same f32 operations as at upload, shader.wgsl:262-264),
 *   - the two child boxes of every inner record, bottom up: a leaf child's box from its triangles, an inner
 *     child's box as the merge of the two boxes in the child's own record (min and max are exact, so the
 *     merge equals the bounds over the child's whole range when the children partition it, which the
 *     upload checked),
 * and report whether every new bound still satisfies the scene-side condition of the reciprocal slab test
 * (pt_exact_div.h).
 */
#include "pt_internal.h"
#include "pt_exact_div.h"

#pragma clang fp contract(off)

namespace {

__device__ __forceinline__ float lo2(float a, float b) { return a < b ? a : b; }
__device__ __forceinline__ float hi2(float a, float b) { return a > b ? a : b; }

struct Box3 { float mn[3], mx[3]; };

/* triangle.rs:14-31 over the leaf's range */
__device__ __forceinline__ Box3 leaf_box(const float *tri9, uint32_t first, uint32_t count) {
    Box3 b;
    for (int k = 0; k < 3; k++) { b.mn[k] = INFINITY; b.mx[k] = -INFINITY; }
    for (uint32_t t = first; t < first + count; t++) {
        const float *p = tri9 + 9 * (size_t)t;
        for (int k = 0; k < 3; k++) {
            const float a = p[k], bb = p[3 + k], c = p[6 + k];
            b.mn[k] = lo2(lo2(lo2(a, bb), c), b.mn[k]);
            b.mx[k] = hi2(hi2(hi2(a, bb), c), b.mx[k]);
        }
    }
    return b;
}

__device__ __forceinline__ Box3 child_box(const PTInnerRecord *inner, const float *tri9, const uint32_t *big, uint32_t ref) {
    if (ref & PT_REF_LEAF) {
        const uint32_t first = ref & 0x00ffffffu;
        uint32_t count = (ref >> 24) & 0x7fu;
        if (count == PT_REF_COUNT_ESCAPE) count = big[first];
        return leaf_box(tri9, first, count);
    }
    const PTInnerRecord r = inner[ref];
    Box3 b;
    b.mn[0] = lo2(r.q[1].x, r.q[0].x); b.mn[1] = lo2(r.q[1].y, r.q[0].y); b.mn[2] = lo2(r.q[2].z, r.q[2].x);
    b.mx[0] = hi2(r.q[1].z, r.q[0].z); b.mx[1] = hi2(r.q[1].w, r.q[0].w); b.mx[2] = hi2(r.q[2].w, r.q[2].y);
    return b;
}

/* the records of one height (children strictly lower), one thread per record */
__global__ __launch_bounds__(256) void refit_level_kernel(PTInnerRecord *inner, const uint32_t *order, uint32_t begin, uint32_t count,
                                                          const float *tri9, const uint32_t *big, uint32_t *not_tame) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const uint32_t rec = order[begin + i];
    const float4 refs = inner[rec].q[3];
    const Box3 a = child_box(inner, tri9, big, __float_as_uint(refs.x));
    const Box3 b = child_box(inner, tri9, big, __float_as_uint(refs.y));
    inner[rec].q[0] = make_float4(a.mn[0], a.mn[1], a.mx[0], a.mx[1]);
    inner[rec].q[1] = make_float4(b.mn[0], b.mn[1], b.mx[0], b.mx[1]);
    inner[rec].q[2] = make_float4(a.mn[2], a.mx[2], b.mn[2], b.mx[2]);
    bool tame = true;
    for (int k = 0; k < 3; k++)
        tame = tame && pt_div_tame_coordinate(a.mn[k]) && pt_div_tame_coordinate(a.mx[k]) &&
               pt_div_tame_coordinate(b.mn[k]) && pt_div_tame_coordinate(b.mx[k]);
    if (!tame) atomicOr(not_tame, 1u);
}

/* the same f32 operations as the upload (shader.wgsl:262-264) */
__global__ __launch_bounds__(256) void refit_tris_kernel(const float *tri9, uint32_t n, PTTriRecord *out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *t = tri9 + 9 * (size_t)i;
    const float abx = t[3] - t[0], aby = t[4] - t[1], abz = t[5] - t[2];
    const float acx = t[6] - t[0], acy = t[7] - t[1], acz = t[8] - t[2];
    const float nx = aby * acz - abz * acy;
    const float ny = abz * acx - abx * acz;
    const float nz = abx * acy - aby * acx;
    out[i].q[0] = make_float4(t[0], t[1], t[2], abx);
    out[i].q[1] = make_float4(aby, abz, acx, acy);
    out[i].q[2] = make_float4(acz, nx, ny, nz);
}

}  // namespace

hipError_t pt_launch_refit_tris(const float *tri9, uint32_t n, PTTriRecord *out, hipStream_t st) {
    if (n) hipLaunchKernelGGL(refit_tris_kernel, dim3((n + 255) / 256), dim3(256), 0, st, tri9, n, out);
    return hipGetLastError();
}

hipError_t pt_launch_refit_level(PTInnerRecord *inner, const uint32_t *order, uint32_t begin, uint32_t count, const float *tri9,
                                 const uint32_t *big, uint32_t *not_tame, hipStream_t st) {
    if (count) hipLaunchKernelGGL(refit_level_kernel, dim3((count + 255) / 256), dim3(256), 0, st, inner, order, begin, count, tri9, big, not_tame);
    return hipGetLastError();
}
