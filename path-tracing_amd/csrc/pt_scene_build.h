/* Internal interface between the GPU builder (pt_bvh_build_gpu.hip) and the host side (pt_host.cpp): the renderer's
 * scene records built on the GPU from plain triangles.  Not part of the public ABI. */
#ifndef PT_SCENE_BUILD_H
#define PT_SCENE_BUILD_H

#include "pt_internal.h"

#include <string>
#include <vector>

/* What pt_upload_scene makes on the host from the triangle file and node file of one build, made on the device
 * instead.  The four arrays are hipMalloc'ed on the device of the call and belong to the caller afterwards
 * (pt_resident_tree_free, or adoption by a context). */
struct PTResidentTree {
    PTInnerRecord *d_inner = nullptr;          /* inner_count records, depth-first order */
    PTTriRecord *d_tris = nullptr;             /* triangle_count records in the tree's triangle order */
    uint32_t *d_big = nullptr;                 /* triangle_count words: count of a leaf of >= 127 triangles, at its first triangle */
    uint32_t *d_refit_order = nullptr;         /* inner records sorted by height */
    size_t bytes_inner = 0, bytes_tris = 0, bytes_big = 0, bytes_refit_order = 0;     /* sizes of the four allocations */
    std::vector<uint32_t> refit_level_begin;   /* where each height starts in d_refit_order (depth + 2 entries) */
    uint32_t inner_count = 0, leaf_count = 0, node_count = 0, depth = 0, root_ref = 0;
    bool tame = false;                         /* every bound satisfies pt_div_tame_coordinate */
    float build_ms = 0.0f;
};
int pt_gpu_build_resident(int device_ordinal, const float *triangles, uint32_t triangle_count, int max_depth,
                          PTResidentTree &out, std::string &err);
void pt_resident_tree_free(PTResidentTree &t);
/* hands a device allocation of `bytes` bytes to the builder's scratch cache (it is reused by a later build or freed) */
void pt_scratch_give_back(int device, void *p, size_t bytes);

#endif
