/*
 * pt_prepare - the reference's two data-preparation steps in one host tool, over the C ABI:
 *   1. prepare_data/data_formatter.js: ASCII STL -> .bin (9 f32 per triangle, scaled)
 *   2. prepare_data/src/main.rs: .bin -> BVH; rewrites the .bin in BVH order and writes _bvh.bin
 * usage: pt_prepare model.stl model.bin model_bvh.bin [--scale S] [--max-depth D] [--gpu [DEVICE]]
 *        pt_prepare --bin model.bin model_bvh.bin [--max-depth D] [--gpu [DEVICE]]   (step 2 only, like the Rust tool)
 * Defaults follow the reference sources: scale 1 (data_formatter.js:5), max depth 16
 * (prepare_data/src/main.rs:176).  Needs no GPU unless --gpu is given: then the tree is built by
 * pt_bvh_build_gpu on that MI355X (same output bytes; fails if there is none).
 */
#include "../../include/pt_api.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

static bool write_floats(const char *path, const float *p, size_t n) {
    FILE *f = fopen(path, "wb");
    if (!f) return false;
    size_t w = n ? fwrite(p, sizeof(float), n, f) : 0;
    fclose(f);
    return w == n;
}

static bool read_floats(const char *path, std::vector<float> &out) {
    FILE *f = fopen(path, "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    if (sz < 0 || sz % 4) { fclose(f); return false; }
    out.resize((size_t)sz / 4);
    size_t r = out.empty() ? 0 : fread(out.data(), 4, out.size(), f);
    fclose(f);
    return r == out.size();
}

int main(int argc, char **argv) {
    std::vector<std::string> pos;
    double scale = 1.0;
    int max_depth = 16;
    bool bin_only = false;
    int gpu = -1;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        if (a == "--scale" && i + 1 < argc) scale = atof(argv[++i]);
        else if (a == "--max-depth" && i + 1 < argc) max_depth = atoi(argv[++i]);
        else if (a == "--bin") bin_only = true;
        else if (a == "--gpu") { gpu = 0; if (i + 1 < argc && argv[i + 1][0] >= '0' && argv[i + 1][0] <= '9') gpu = atoi(argv[++i]); }
        else if (a.rfind("--", 0) == 0) { fprintf(stderr, "unknown option %s\n", a.c_str()); return 2; }
        else pos.push_back(a);
    }
    if (pos.size() != (bin_only ? 2u : 3u)) {
        fprintf(stderr, "usage: pt_prepare model.stl model.bin model_bvh.bin [--scale S] [--max-depth D] [--gpu [DEVICE]]\n"
                        "       pt_prepare --bin model.bin model_bvh.bin [--max-depth D] [--gpu [DEVICE]]\n");
        return 2;
    }
    const std::string bin = bin_only ? pos[0] : pos[1], bvh = bin_only ? pos[1] : pos[2];
    std::vector<float> tris;
    if (bin_only) {
        if (!read_floats(bin.c_str(), tris)) { fprintf(stderr, "cannot read %s\n", bin.c_str()); return 1; }
    } else {
        uint64_t n = 0;
        if (pt_stl_format_file(pos[0].c_str(), bin.c_str(), scale, &n)) { fprintf(stderr, "%s\n", pt_stl_format_error()); return 1; }
        if (!read_floats(bin.c_str(), tris)) { fprintf(stderr, "cannot read back %s\n", bin.c_str()); return 1; }
    }
    if (tris.empty() || tris.size() % 9) { fprintf(stderr, "%s: %zu floats is not a whole number of triangles\n", bin.c_str(), tris.size()); return 1; }
    const uint32_t n_tris = (uint32_t)(tris.size() / 9);
    std::vector<float> ordered(tris.size());
    float *nodes = nullptr;
    uint32_t n_nodes = 0;
    float build_ms = 0.0f;
    if (gpu >= 0) {
        if (pt_bvh_build_gpu(gpu, tris.data(), n_tris, max_depth, ordered.data(), &nodes, &n_nodes, &build_ms)) {
            fprintf(stderr, "pt_bvh_build_gpu: %s\n", pt_bvh_build_gpu_error());
            return 1;
        }
    } else if (pt_bvh_build(tris.data(), n_tris, max_depth, ordered.data(), &nodes, &n_nodes)) { fprintf(stderr, "pt_bvh_build failed\n"); return 1; }
    bool ok = write_floats(bin.c_str(), ordered.data(), ordered.size()) && write_floats(bvh.c_str(), nodes, (size_t)n_nodes * 9);
    pt_free(nodes);
    if (!ok) { fprintf(stderr, "cannot write %s / %s\n", bin.c_str(), bvh.c_str()); return 1; }
    printf("%u triangles, %u nodes -> %s, %s\n", n_tris, n_nodes, bin.c_str(), bvh.c_str());
    if (gpu >= 0) printf("built on GPU %d in %.2f ms\n", gpu, build_ms);
    return 0;
}
