#!/usr/bin/env python3
"""Writes tests/golden/wgsl_golden_scenes.npz: whole frames of the REFERENCE SHADER (its source text executed by
wgsl_interp.py, see make_wgsl_golden.py) on the other benchmark scenes: the legacy dragon_8k node file (21 levels, a
corrupt link, unreachable nodes) with the emissive sphere of configuration 2, the 87 k-triangle dragon with the tree of
this repository's builder (configuration 3), the 784 k-triangle stand-in of configuration 4, and the knight with 16 spheres and 12 bounces (configuration 5).  Each
case: the accumulation buffer after draws with frame_count 0, 1 and 13 (13 * 393939123 wraps around 2^32).
Only numbers are stored.

usage: python3 tests/golden/make_wgsl_golden_scenes.py [--shader PATH] [--quick]
"""
import argparse
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_wgsl_golden as mk  # noqa: E402

CASES = {      # configuration: width, height, rays per pixel
    2: (48, 27, 2),
    3: (40, 22, 2),
    4: (24, 14, 2),
    5: (48, 27, 2),
}
FRAME_COUNTS = (0, 1, 13)


def generate(src, shim, scenes, quick=False):
    out = {"frame_counts": np.array(FRAME_COUNTS, dtype=np.uint32)}
    for n, (w, h, rpp) in CASES.items():
        if quick:
            w, h, rpp = 8, 4, 1
        cfg = scenes.config(n)
        tris, bvh = scenes.mesh(cfg["mesh"])
        spheres = np.asarray(scenes.spheres_for(cfg["spheres"]), dtype=np.float32).reshape(-1, 12)
        depth = mk.tree_depth(np.asarray(bvh, dtype=np.float32).reshape(-1, 9)) if n != 2 else 32    # the legacy file has a corrupt link: walk it like the shader, with room
        bounces = cfg["max_bounce_count"]
        I = mk.make_interpreter(src, shim, sphere_count=len(spheres), triangle_count=len(tris), bvh_node_count=len(bvh),
                                bvh_max_depth=depth, max_bounce_count=bounces, rays_per_pixel=rpp, screen_size=(float(w), float(h)))
        frame = mk.bind_scene(I, spheres, tris, bvh, cfg["camera_position"], cfg["camera_rotation"], w, h)
        t0 = time.time()
        snaps = []
        for fc in FRAME_COUNTS:
            mk.run_frame(I, fc, w, h)
            snaps.append(frame.copy())
            print("configuration %d, frame_count %d: %.1f s" % (n, fc, time.time() - t0), flush=True)
        out["cfg%d_size_rpp_bounces" % n] = np.array([w, h, rpp, bounces], dtype=np.int64)
        out["cfg%d_frames_accum" % n] = np.stack(snaps)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shader", default="/root/reference/src/shaders/shader.wgsl")
    ap.add_argument("--out", default=os.path.join(HERE, "wgsl_golden_scenes.npz"))
    ap.add_argument("--quick", action="store_true")
    args = ap.parse_args()
    src = open(args.shader).read()
    sys.path.insert(0, mk.ROOT)
    import __graft_entry__ as entry
    entry.load_package()                    # the regenerated trees come from the library's host builder
    from path_tracing_amd import scenes
    out = generate(src, mk.build_shim(), scenes, quick=args.quick)
    np.savez_compressed(args.out, **out)
    print("wrote %s: %d arrays" % (args.out, len(out)))


if __name__ == "__main__":
    main()
