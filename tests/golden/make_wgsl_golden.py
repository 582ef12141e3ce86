#!/usr/bin/env python3
"""Writes tests/golden/wgsl_golden.npz: what the REFERENCE SHADER ITSELF returns, function by function and for
whole frames, when its source text (/root/reference/src/shaders/shader.wgsl, read here and never copied) is
executed by tests/golden/wgsl_interp.py in strict binary32 arithmetic with the portable log / cos / pow
(tests/native/pm_shim.c).  Only numbers are stored: inputs and the shader's outputs.

Scene constants the reference expects its user to edit in the source (sphere_count, triangle_count,
bvh_node_count, bvh_max_depth, max_bounce_count, rays_per_pixel, screen_size) are overridden per case; everything
else, including the material and sky constants and the 1000000.0 / 999999999.0 distances, is the source's.

usage: python3 tests/golden/make_wgsl_golden.py [--shader PATH] [--quick]
"""
import argparse
import os
import subprocess
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import wgsl_interp as W  # noqa: E402

F32, U32 = W.F32, W.U32


def build_shim():
    out = os.path.join(ROOT, "tests", "native", "pm_shim.so")
    src = os.path.join(ROOT, "tests", "native", "pm_shim.c")
    hdr = os.path.join(ROOT, "path-tracing_amd", "csrc", "pt_portable_math.h")
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.run(["gcc", "-O2", "-std=gnu11", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-shared", "-o", out, src], check=True)
    return out


def fvec(v):
    return np.array([float(x) for x in v.c], dtype=np.float32)


def ray_struct(I, o, d):
    return I.struct("Ray", origin=W.vec3(*o), dir=W.vec3(*d))


def hit_row(h):
    """did_hit, distance, position, normal, color, emission_color, emission_strength, smoothness as 16 floats"""
    f = h.f
    return np.array([1.0 if f["did_hit"] else 0.0, float(f["distance"])] + list(fvec(f["position"])) + list(fvec(f["normal"])) +
                    list(fvec(f["color"])) + list(fvec(f["emission_color"])) + [float(f["emission_strength"]), float(f["smoothness"])],
                    dtype=np.float32)


def make_interpreter(src, shim, **overrides):
    return W.Interpreter(src, W.Builtins(W.SharedMath(shim)), overrides)


def bind_scene(I, spheres, tris, bvh, cam_pos, cam_rot, width, height):
    I.bind("sphere_data", W.NpArr(np.ascontiguousarray(spheres, dtype=np.float32).reshape(-1, 12)))
    I.bind("triangle_data", W.NpArr(np.ascontiguousarray(tris, dtype=np.float32).reshape(-1)))
    I.bind("bvh_data", W.NpArr(np.ascontiguousarray(bvh, dtype=np.float32).reshape(-1)))
    I.bind("camera_position", W.vec3(*cam_pos))
    I.bind("camera_rotation", W.vec3(*cam_rot))
    I.bind("frame_count", U32(0))
    frame = np.zeros((height, width, 3), dtype=np.float32)
    I.bind("frame_data", W.NpArr(frame, vec_last=True))
    return frame


def run_frame(I, frame_count, width, height):
    """one draw: vs_main once (the flat varyings), fs_main for every pixel centre"""
    I.bind("frame_count", U32(frame_count))
    vo = I.call("vs_main", [U32(0)])
    for y in range(height):
        for x in range(width):
            vin = I.struct("VertexOutput", pos=W.Vec((F32(x + 0.5), F32(y + 0.5), F32(0.0), F32(1.0))),
                           screen_width=vo.f["screen_width"], screen_height=vo.f["screen_height"])
            I.call("fs_main", [vin])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shader", default="/root/reference/src/shaders/shader.wgsl")
    ap.add_argument("--out", default=os.path.join(HERE, "wgsl_golden.npz"))
    ap.add_argument("--quick", action="store_true", help="small cases only (what the test re-runs when the reference is present)")
    args = ap.parse_args()
    src = open(args.shader).read()
    shim = build_shim()
    scenes = import_module_scenes()
    out = generate(src, shim, scenes, quick=args.quick)
    np.savez_compressed(args.out, **out)
    print("wrote %s: %d arrays" % (args.out, len(out)))


def import_module_scenes():
    import importlib.util
    p = os.path.join(ROOT, "path-tracing_amd", "scenes.py")
    spec = importlib.util.spec_from_file_location("pt_scenes_for_golden", p)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def tree_depth(nodes):
    """levels below the root (leaf iff child_index == 0, children at child_index and child_index + 1)"""
    best, stack = 0, [(0, 0)]
    while stack:
        n, d = stack.pop()
        best = max(best, d)
        c = int(nodes[n, 8])
        if c != 0:
            stack.append((c, d + 1))
            stack.append((c + 1, d + 1))
    return best


def generate(src, shim, scenes, quick=False):
    rng = np.random.default_rng(20241111)
    out = {}
    t0 = time.time()
    I = make_interpreter(src, shim)

    # ---- RNG (shader.wgsl random_value .. random_hemisphere_direction)
    seeds = np.array([0, 1, 2, 3, 12345, 248135, 393939123, 0x7FFFFFFF, 0x80000000, 0xFFFFFFFF, 4000000000] +
                     [int(x) for x in rng.integers(0, 2**32, 8 if quick else 40)], dtype=np.uint32)
    out["rng_seeds"] = seeds
    out["rng_value"] = np.array([I.call("random_value", [U32(int(s))]) for s in seeds], dtype=np.float32)
    out["rng_normal"] = np.array([I.call("random_normal", [U32(int(s))]) for s in seeds], dtype=np.float32)
    out["rng_direction"] = np.stack([fvec(I.call("random_direction", [U32(int(s))])) for s in seeds])
    hemi_n = np.array([0.0, 0.6, -0.8], dtype=np.float32)
    out["rng_hemisphere_normal"] = hemi_n
    out["rng_hemisphere"] = np.stack([fvec(I.call("random_hemisphere_direction", [W.vec3(*hemi_n), U32(int(s))])) for s in seeds])

    # ---- rotate_vector, get_environment_light, lerp
    n = 6 if quick else 24
    vecs = rng.normal(size=(n, 3)).astype(np.float32)
    angles = np.concatenate([np.array([[0, 0, 0], [0, 90, 0], [30, -45, 10]], dtype=np.float32),
                             rng.uniform(-180, 180, size=(n - 3, 3)).astype(np.float32)])
    out["rot_vectors"], out["rot_angles"] = vecs, angles
    out["rot_result"] = np.stack([fvec(I.call("rotate_vector", [W.vec3(*v), W.vec3(*a)])) for v, a in zip(vecs, angles)])
    dirs = rng.normal(size=(8 if quick else 48, 3)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True).astype(np.float32)
    dirs[0] = [0.0, 1.0, 0.0]
    dirs[1] = [1.0, 0.0, 0.0]
    dirs[2] = [0.0, -0.005, 1.0]
    sun = np.array([0.0, 0.4, -0.5], dtype=np.float32)
    dirs[3] = sun / np.float32(np.sqrt(np.float32(0.41)))
    out["env_dirs"] = dirs
    out["env_light"] = np.stack([fvec(I.call("get_environment_light", [ray_struct(I, (0, 0, 0), d)])) for d in dirs])

    # ---- ray_box, ray_triangle, ray_sphere
    n = 24 if quick else 240
    o = rng.uniform(-60, 60, size=(n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d[::7, 0] = 0.0                      # axis-parallel rays: division by zero inside ray_box
    d[3::11, 1] = 0.0
    lo = rng.uniform(-50, 40, size=(n, 3)).astype(np.float32)
    hi = lo + rng.uniform(0, 30, size=(n, 3)).astype(np.float32)
    hi[5::13, 2] = lo[5::13, 2]          # flat boxes
    o[9::17] = (lo[9::17] + hi[9::17]) * np.float32(0.5)      # origins inside the box
    o[4::19, 0] = lo[4::19, 0]           # origin on a plane with a ray parallel to it: 0 / 0
    d[4::19, 0] = 0.0
    boxes = np.concatenate([lo, hi], axis=1)
    out["box_origins"], out["box_dirs"], out["box_bounds"] = o, d, boxes
    out["box_result"] = np.array([I.call("ray_box", [ray_struct(I, oo, dd), W.Arr(F32(x) for x in bb)])
                                  for oo, dd, bb in zip(o, d, boxes)], dtype=np.float32)
    tri = rng.uniform(-20, 20, size=(n, 9)).astype(np.float32)
    to = rng.uniform(-40, 40, size=(n, 3)).astype(np.float32)
    centre = (tri[:, 0:3] + tri[:, 3:6] + tri[:, 6:9]) / np.float32(3.0)
    td = (centre - to + rng.normal(scale=4.0, size=(n, 3))).astype(np.float32)      # aimed near the triangle
    td[::9] = -td[::9]                   # pointing away
    out["tri_origins"], out["tri_dirs"], out["tri_vertices"] = to, td, tri
    out["tri_hits"] = np.stack([hit_row(I.call("ray_triangle", [ray_struct(I, oo, dd), W.Arr(W.vec3(*t[3 * k:3 * k + 3]) for k in range(3))]))
                                for oo, dd, t in zip(to, td, tri)])
    sph = np.zeros((n, 12), dtype=np.float32)
    sph[:, 0:3] = rng.uniform(-30, 30, size=(n, 3))
    sph[:, 3] = rng.uniform(1, 15, size=n)
    sph[:, 4:12] = rng.uniform(0, 1, size=(n, 8))
    so = rng.uniform(-60, 60, size=(n, 3)).astype(np.float32)
    sd = (sph[:, 0:3] - so + rng.normal(scale=6.0, size=(n, 3))).astype(np.float32)
    so[::8] = sph[::8, 0:3]              # origin at the centre: inside the sphere
    out["sph_origins"], out["sph_dirs"], out["sph_records"] = so, sd, sph
    out["sph_hits"] = np.stack([hit_row(I.call("ray_sphere", [ray_struct(I, oo, dd), W.Arr(F32(x) for x in s)]))
                                for oo, dd, s in zip(so, sd, sph)])
    print("functions: %.1f s" % (time.time() - t0))

    # ---- the teapot scene with the three spheres of configuration 1: calculate_ray_collision, trace, whole frames
    cfg = scenes.config(1)
    tris, bvh = scenes.mesh(cfg["mesh"])
    spheres = np.asarray(scenes.spheres_for(cfg["spheres"]), dtype=np.float32).reshape(-1, 12)
    depth = tree_depth(np.asarray(bvh, dtype=np.float32).reshape(-1, 9))
    W_, H_ = (12, 6) if quick else (32, 18)
    rpp, bounces, frames = (1, 2, 2) if quick else (2, 3, 3)
    I = make_interpreter(src, shim, sphere_count=len(spheres), triangle_count=len(tris), bvh_node_count=len(bvh),
                         bvh_max_depth=depth, max_bounce_count=bounces, rays_per_pixel=rpp, screen_size=(float(W_), float(H_)))
    cam_pos, cam_rot = cfg["camera_position"], cfg["camera_rotation"]
    frame = bind_scene(I, spheres, tris, bvh, cam_pos, cam_rot, W_, H_)
    out["scene_width_height_rpp_bounces_frames_depth"] = np.array([W_, H_, rpp, bounces, frames, depth], dtype=np.int64)
    out["scene_camera"] = np.array(list(cam_pos) + list(cam_rot), dtype=np.float32)

    t0 = time.time()
    n = 8 if quick else 48
    ro = np.tile(np.array(cam_pos, dtype=np.float32), (n, 1))
    rd = rng.normal(size=(n, 3)).astype(np.float32)
    rd[:, 2] = -np.abs(rd[:, 2]) * 3            # roughly down the camera's view (cfg 1 looks along -z)
    ro[n // 2:] = rng.uniform(-60, 60, size=(n - n // 2, 3)).astype(np.float32)      # and rays from anywhere
    out["cast_origins"], out["cast_dirs"] = ro, rd
    out["cast_hits"] = np.stack([hit_row(I.call("calculate_ray_collision", [ray_struct(I, oo, dd)])) for oo, dd in zip(ro, rd)])
    print("casts: %.1f s" % (time.time() - t0))

    t0 = time.time()
    n = 4 if quick else 24
    tseeds = rng.integers(0, 2**32, n).astype(np.uint32)
    tframes = np.array([0, 1, 5, 11, 12, 1000][:min(6, n)] + [0] * max(0, n - 6), dtype=np.uint32)[:n]
    res = []
    for k in range(n):
        I.bind("frame_count", U32(int(tframes[k])))
        res.append(fvec(I.call("trace", [ray_struct(I, ro[k], rd[k]), U32(int(tseeds[k]))])))
    out["trace_seeds"], out["trace_frame_counts"], out["trace_result"] = tseeds, tframes, np.stack(res)
    print("traces: %.1f s" % (time.time() - t0))

    t0 = time.time()
    snaps = []
    for f in range(frames):
        run_frame(I, f, W_, H_)
        snaps.append(frame.copy())
        print("frame %d: %.1f s" % (f, time.time() - t0))
    out["frames_accum"] = np.stack(snaps)          # frame_data after each draw: (frames, H, W, 3)
    return out


if __name__ == "__main__":
    main()
