#!/usr/bin/env python3
"""Generates the committed golden fixtures from the CPU oracle.

The reference holds no golden vectors for its shader and cannot be run in this environment, so
these are SELF-CONSISTENCY pins of the oracle (they catch accidental changes of the restatement,
of compiler flags or of libm), not outputs of the reference.  Re-run only when the oracle is
changed on purpose:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

SEEDS = [0, 1, 2, 3, 12345, 248135, 393939123, 2072640, 0x7FFFFFFF, 0x80000000, 0xFFFFFFFE, 0xFFFFFFFF,
         960 + 540 * 1920, 1920 + 2159 * 3840 + 3839]


def main():
    entry.load_package()
    from path_tracing_amd import scenes
    om = entry.load_oracle_module()
    o, os_ = om.Oracle(False), om.Oracle(True)
    out = {}
    out["seeds"] = np.array(SEEDS, dtype=np.uint64)
    out["random_value"] = np.array([o.random_value(s) for s in SEEDS], np.float32)
    out["random_normal_libm"] = np.array([o.random_normal(s) for s in SEEDS], np.float32)
    out["random_normal_portable"] = np.array([os_.random_normal(s) for s in SEEDS], np.float32)
    out["random_direction_portable"] = np.array([os_.random_direction(s) for s in SEEDS], np.float32)

    cfg = scenes.config(1)
    tris, bvh = scenes.mesh("teapot")
    sph = scenes.spheres_for("cfg1")
    cam, rot = cfg["camera_position"], cfg["camera_rotation"]
    p = o.default_params(width=64, height=36, rays_per_pixel=1, max_bounce_count=4)
    sc = o.scene(sph, tris, bvh)
    ids, dist = o.primary_hits(p, sc, cam, rot)
    out["teapot_64x36_prim_id"] = ids
    out["teapot_64x36_distance"] = dist
    p2 = o.default_params(width=16, height=9, rays_per_pixel=4, max_bounce_count=4)
    out["teapot_16x9_4spp_2frames_libm"] = o.render_frames(p2, sc, cam, rot, 0, 2)
    out["teapot_16x9_4spp_2frames_portable"] = os_.render_frames(os_.default_params(
        width=16, height=9, rays_per_pixel=4, max_bounce_count=4), os_.scene(sph, tris, bvh), cam, rot, 0, 2)

    cfg2 = scenes.config(2)
    t8, b8 = scenes.mesh("dragon_8k")
    p3 = o.default_params(width=48, height=27, rays_per_pixel=1, max_bounce_count=8)
    ids8, dist8 = o.primary_hits(p3, o.scene(scenes.spheres_for("white_light"), t8, b8),
                                 cfg2["camera_position"], cfg2["camera_rotation"])
    out["dragon8k_48x27_prim_id"] = ids8
    out["dragon8k_48x27_distance"] = dist8

    # scalar spot cases
    boxes = np.array([[1, -1, -1, 2, 1, 1]], np.float32)
    rays = np.array([[0, 0, 0, 1, 0, 0], [1.5, 0, 0, 1, 0, 0], [0, -1, 0, 1, 0, 0], [0.3, 0.2, -4, 0.1, 0.05, 1]], np.float32)
    out["ray_box_cases"] = np.array([o.ray_box(r[:3], r[3:], boxes[0]) for r in rays], np.float32)
    out["ray_box_rays"] = rays
    np.savez_compressed(os.path.join(HERE, "oracle_golden.npz"), **out)
    print("wrote", os.path.join(HERE, "oracle_golden.npz"), {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
