/*
 * pt_render - headless C++ host driver over the C ABI (include/pt_api.h).
 *
 * Plays the role of the reference's `run()` / `State::render` loop
 * (src/main.rs:444-523, 531-739) without a window: loads a .bin/_bvh.bin pair,
 * sets spheres and camera, renders N frames and writes the accumulation buffer
 * as a binary PPM.  The reference presents the linear running mean to an sRGB
 * surface (src/main.rs:173-179, shader.wgsl:108), i.e. the display hardware
 * applies the sRGB transfer function on write; the PPM does the same.
 *
 * usage: pt_render --tris a.bin --bvh a_bvh.bin [--sphere x,y,z,r,cr,cg,cb,er,eg,eb,es,sm]...
 *                  [--cam x,y,z] [--rot a,b,c] [--size WxH] [--rpp N] [--frames N]
 *                  [--bounces N] [--device D] [--exact] [--out image.ppm] [--walk SCRIPT]
 *
 * --walk plays the reference's timer loop (src/main.rs:674-676: update_camera, then render) for
 * a scripted key sequence instead of --frames: SCRIPT is a comma-separated list of KEYS*TICKS,
 * KEYS being letters of "WSDA^vIKLJOU" (^ = Space, v = ShiftLeft) held together, or "-" for no
 * key.  Example: "W*10,WL*20,-*16" walks forward, walks while turning, then stands still for 16
 * ticks, during which the image accumulates (a tick with a key held restarts it, main.rs:47-49).
 */
#include "../../include/pt_api.h"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

static bool parse_floats(const char *s, float *out, int n) {
    for (int i = 0; i < n; i++) {
        char *end = nullptr;
        out[i] = strtof(s, &end);
        if (end == s) return false;
        s = end;
        if (i + 1 < n) { if (*s != ',') return false; s++; }
    }
    return *s == 0;
}

struct WalkStep { uint8_t keys[12]; unsigned ticks; };

static bool parse_walk(const char *s, std::vector<WalkStep> &out) {
    static const char codes[] = "WSDA^vIKLJOU";         /* order of State::keys_pressed, src/main.rs:39 */
    while (*s) {
        WalkStep st;
        memset(&st, 0, sizeof(st));
        bool any = false;
        for (; *s && *s != '*'; s++, any = true) {
            if (*s == '-') continue;
            const char *c = strchr(codes, *s);
            if (!c) return false;
            st.keys[c - codes] = 1;
        }
        if (!any || *s != '*') return false;
        char *end = nullptr;
        long n = strtol(s + 1, &end, 10);
        if (end == s + 1 || n < 0) return false;
        st.ticks = (unsigned)n;
        out.push_back(st);
        s = end;
        if (*s == ',') s++; else if (*s) return false;
    }
    return true;
}

static unsigned char to_srgb8(float v) {
    if (!(v > 0.0f)) v = 0.0f;           /* also maps NaN to 0 */
    if (v > 1.0f) v = 1.0f;
    float e = v <= 0.0031308f ? 12.92f * v : 1.055f * std::pow(v, 1.0f / 2.4f) - 0.055f;
    return (unsigned char)std::lround(e * 255.0f);
}

int main(int argc, char **argv) {
    std::string tris, bvh, out = "out.ppm";
    std::vector<float> spheres;
    float cam[3] = {200.0f, 0.0f, 200.0f}, rot[3] = {0.0f, 90.0f, 0.0f};   /* src/main.rs:382-383 */
    pt_params p;
    pt_default_params(&p);
    unsigned frames = 1;
    int device = 0;
    std::vector<WalkStep> walk;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto need = [&](const char *what) -> const char * {
            if (i + 1 >= argc) { fprintf(stderr, "%s needs a value\n", what); exit(2); }
            return argv[++i];
        };
        if (a == "--tris") tris = need("--tris");
        else if (a == "--bvh") bvh = need("--bvh");
        else if (a == "--out") out = need("--out");
        else if (a == "--sphere") { float s[12]; if (!parse_floats(need("--sphere"), s, 12)) { fprintf(stderr, "bad --sphere\n"); return 2; } spheres.insert(spheres.end(), s, s + 12); }
        else if (a == "--cam") { if (!parse_floats(need("--cam"), cam, 3)) { fprintf(stderr, "bad --cam\n"); return 2; } }
        else if (a == "--rot") { if (!parse_floats(need("--rot"), rot, 3)) { fprintf(stderr, "bad --rot\n"); return 2; } }
        else if (a == "--size") { if (sscanf(need("--size"), "%ux%u", &p.width, &p.height) != 2) { fprintf(stderr, "bad --size\n"); return 2; } }
        else if (a == "--rpp") p.rays_per_pixel = (unsigned)atoi(need("--rpp"));
        else if (a == "--frames") frames = (unsigned)atoi(need("--frames"));
        else if (a == "--bounces") p.max_bounce_count = (unsigned)atoi(need("--bounces"));
        else if (a == "--device") device = atoi(need("--device"));
        else if (a == "--exact") p.flags |= PT_FLAG_EXACT_MATH;
        else if (a == "--walk") { if (!parse_walk(need("--walk"), walk)) { fprintf(stderr, "bad --walk\n"); return 2; } }
        else { fprintf(stderr, "unknown argument %s\n", a.c_str()); return 2; }
    }
    if (tris.empty() || bvh.empty()) { fprintf(stderr, "--tris and --bvh are required\n"); return 2; }

    pt_ctx *ctx = nullptr;
    if (pt_create(device, &p, &ctx)) { fprintf(stderr, "pt_create: %s\n", pt_last_error(nullptr)); return 1; }
    if (pt_load_scene_files(ctx, tris.c_str(), bvh.c_str(), spheres.data(), (uint32_t)(spheres.size() / 12))) {
        fprintf(stderr, "pt_load_scene_files: %s\n", pt_last_error(ctx)); pt_destroy(ctx); return 1;
    }
    pt_set_camera(ctx, cam, rot);
    auto t0 = std::chrono::steady_clock::now();
    if (walk.empty()) {
        /* State::render once per frame, frame_count 0,1,2,... (src/main.rs:448-452,510) */
        if (pt_render_frames(ctx, 0, frames) || pt_synchronize(ctx)) {
            fprintf(stderr, "render: %s\n", pt_last_error(ctx)); pt_destroy(ctx); return 1;
        }
    } else {
        /* the timer loop: update_camera, then render (src/main.rs:674-676) */
        uint32_t frame_count = 0;
        frames = 0;
        for (const WalkStep &st : walk) {
            for (unsigned t = 0; t < st.ticks; t++) {
                int any_key = 0;
                pt_update_camera(st.keys, 1.0f, 1.0f, cam, rot, &any_key);     /* CAMERA_SPEED, CAMERA_ROT_SPEED */
                if (any_key) { frame_count = 0; pt_set_camera(ctx, cam, rot); }
                if (pt_render_frames(ctx, frame_count, 1)) {
                    fprintf(stderr, "render: %s\n", pt_last_error(ctx)); pt_destroy(ctx); return 1;
                }
                frame_count++;
                frames++;
            }
        }
        if (pt_synchronize(ctx)) { fprintf(stderr, "render: %s\n", pt_last_error(ctx)); pt_destroy(ctx); return 1; }
        printf("camera after the walk: position %.9g,%.9g,%.9g rotation %.9g,%.9g,%.9g, %u accumulated frame(s)\n",
               cam[0], cam[1], cam[2], rot[0], rot[1], rot[2], frame_count);
    }
    double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::vector<float> img((size_t)p.width * p.height * 3);
    if (pt_read_accum(ctx, img.data(), 3)) { fprintf(stderr, "read: %s\n", pt_last_error(ctx)); pt_destroy(ctx); return 1; }
    pt_destroy(ctx);

    FILE *f = fopen(out.c_str(), "wb");
    if (!f) { fprintf(stderr, "cannot write %s\n", out.c_str()); return 1; }
    fprintf(f, "P6\n%u %u\n255\n", p.width, p.height);
    std::vector<unsigned char> row((size_t)p.width * 3);
    for (unsigned y = 0; y < p.height; y++) {
        for (size_t i = 0; i < row.size(); i++) row[i] = to_srgb8(img[(size_t)y * p.width * 3 + i]);
        fwrite(row.data(), 1, row.size(), f);
    }
    fclose(f);
    double samples = (double)p.width * p.height * p.rays_per_pixel * frames;
    printf("%ux%u, %u frames x %u rays/pixel, %.3f s, %.1f Msamples/s -> %s\n",
           p.width, p.height, frames, p.rays_per_pixel, sec, samples / sec / 1e6, out.c_str());
    return 0;
}
