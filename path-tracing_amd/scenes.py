"""The five benchmark / parity configurations and their synthetic inputs.

Scene files are the reference's own data contract (`.bin` = 9 f32 per triangle,
`_bvh.bin` = 9 f32 per node, prepare_data/src/main.rs:179-205); sphere records
are the 12-float layout of src/main.rs:286.  The reference ships no usable
camera pose (its default, main.rs:382-383, does not see the mesh), so the poses
below are this repository's own.

cfg 4 needs `dragon_800k`, which the reference repository does not contain
(neither the mesh nor its BVH).  `dragon_800k_synth()` builds a stand-in from
dragon_80k.bin by splitting every triangle into 9 (edge trisection, file order)
and running the BVH builder on the result; every number quoted on it must say
"synthetic stand-in".
"""
import os

import numpy as np

ASSETS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "assets", "objects")

# BVH depth cap for regenerated trees (the reference tool uses 16,
# prepare_data/src/main.rs:176, which would leave ~12 triangles per leaf at 784k triangles)
REGEN_MAX_DEPTH = 32


def load_bin(name):
    return np.fromfile(os.path.join(ASSETS, name), dtype="<f4").astype(np.float32).reshape(-1, 9)


def split_1_to_9(triangles):
    """Every triangle -> 9 coplanar triangles with the same winding (f32 arithmetic)."""
    t = np.asarray(triangles, dtype=np.float32).reshape(-1, 3, 3)
    a, b, c = t[:, 0], t[:, 1], t[:, 2]
    ab, ac = (b - a).astype(np.float32), (c - a).astype(np.float32)

    def p(i, j):
        return (a + ab * np.float32(i / 3.0) + ac * np.float32(j / 3.0)).astype(np.float32)

    out = []
    for i in range(3):
        for j in range(3 - i):
            out.append(np.stack([p(i, j), p(i + 1, j), p(i, j + 1)], axis=1))       # upright
            if i + j <= 1:
                out.append(np.stack([p(i + 1, j), p(i + 1, j + 1), p(i, j + 1)], axis=1))  # inverted
    res = np.stack(out, axis=1)   # (n, 9, 3, 3)
    return np.ascontiguousarray(res.reshape(-1, 9).astype(np.float32))


def _lcg(seed):
    state = seed & 0xFFFFFFFF
    while True:
        state = (state * 1664525 + 1013904223) & 0xFFFFFFFF
        yield (state >> 8) / float(1 << 24)


def knight_spheres(seed=20240531):
    """16 spheres on a ring around the knight (cfg 5): radius 10-30, smoothness from
    {0, .25, .5, .75, .9, 1}, emission strength 0-10; fixed-seed LCG."""
    rnd = _lcg(seed)
    smooth = [0.0, 0.25, 0.5, 0.75, 0.9, 1.0]
    out = []
    for k in range(16):
        ang = 2.0 * np.pi * k / 16.0
        ring = 170.0 + 60.0 * next(rnd)
        x, y = ring * np.cos(ang), ring * np.sin(ang)
        z = 20.0 + 280.0 * next(rnd)
        radius = 10.0 + 20.0 * next(rnd)
        col = [0.2 + 0.8 * next(rnd) for _ in range(3)]
        emissive = next(rnd) < 0.4
        ecol = [next(rnd), next(rnd), next(rnd)]
        strength = 10.0 * next(rnd) if emissive else 0.0
        sm = smooth[int(next(rnd) * 6) % 6]
        out.append([x, y, z, radius, *col, *ecol, strength, sm])
    return np.asarray(out, dtype=np.float32)


# sphere lists of src/main.rs:288-302
SPHERES_CFG1 = np.asarray([
    [-40.0, 0.0, 0.0, 10.0, 0.25, 0.25, 1.0, 0.0, 0.0, 0.0, 0.0, 0.9],
    [40.0, 0.0, 0.0, 10.0, 1.0, 1.0, 1.0, 0.0, 0.0, 0.0, 0.0, 1.0],
    [0.0, -20.0, 0.0, 10.0, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0],
], dtype=np.float32)
RED_LIGHT = np.asarray([[40.0, 40.0, -20.0, 10.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 10.0, 0.0]], dtype=np.float32)
WHITE_LIGHT = np.asarray([[40.0, 40.0, -20.0, 10.0, 0.0, 0.0, 0.0, 1.0, 1.0, 1.0, 10.0, 0.0]], dtype=np.float32)

CONFIGS = {
    1: dict(name="teapot_640x360_1spp_4b", mesh="teapot", width=640, height=360, rays_per_pixel=1,
            n_frames=1, max_bounce_count=4, spheres="cfg1",
            camera_position=(5.0, 35.0, 150.0), camera_rotation=(0.0, 0.0, 0.0)),
    2: dict(name="dragon_8k_1080p_16spp_8b", mesh="dragon_8k", width=1920, height=1080, rays_per_pixel=16,
            n_frames=1, max_bounce_count=8, spheres="white_light",
            camera_position=(125.0, -23.0, 3.0), camera_rotation=(0.0, 90.0, 0.0)),
    3: dict(name="dragon_80k_1080p_64spp_8b", mesh="dragon_80k", width=1920, height=1080, rays_per_pixel=16,
            n_frames=4, max_bounce_count=8, spheres="red_light",
            camera_position=(125.0, -23.0, 3.0), camera_rotation=(0.0, 90.0, 0.0)),
    4: dict(name="dragon_800k_synth_1080p_256spp_8b", mesh="dragon_800k_synth", width=1920, height=1080,
            rays_per_pixel=16, n_frames=16, max_bounce_count=8, spheres="red_light",
            camera_position=(125.0, -23.0, 3.0), camera_rotation=(0.0, 90.0, 0.0)),
    5: dict(name="knight_16spheres_4k_1024spp_12b", mesh="knight", width=3840, height=2160, rays_per_pixel=16,
            n_frames=64, max_bounce_count=12, spheres="knight16",
            camera_position=(0.0, -540.0, 168.0), camera_rotation=(90.0, 0.0, 0.0)),
}


def spheres_for(kind):
    return {"cfg1": SPHERES_CFG1, "red_light": RED_LIGHT, "white_light": WHITE_LIGHT,
            "knight16": knight_spheres(), "none": np.zeros((0, 12), np.float32)}[kind]


_mesh_cache = {}


def mesh(name, builder=None):
    """Returns (triangles[n,9], bvh[m,9]).  Committed pairs are loaded as they are;
    dragon_80k / dragon_800k_synth get their BVH from `builder(triangles, max_depth)`
    (default: the library's host builder, pt_bvh_build)."""
    key = (name, id(builder))
    if key in _mesh_cache:
        return _mesh_cache[key]
    if name in ("teapot", "knight", "dragon_8k"):
        res = (load_bin(name + ".bin"), load_bin(name + "_bvh.bin"))
    else:
        if builder is None:
            from . import bvh_build as builder
        tris = load_bin("dragon_80k.bin")
        if name == "dragon_800k_synth":
            tris = split_1_to_9(tris)
        elif name != "dragon_80k":
            raise KeyError(name)
        res = builder(tris, REGEN_MAX_DEPTH)
    _mesh_cache[key] = res
    return res


def config(n, **overrides):
    c = dict(CONFIGS[n])
    c.update(overrides)
    return c
