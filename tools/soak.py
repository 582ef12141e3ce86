#!/usr/bin/env python3
"""Determinism soak: renders the same frames many times and checks that every image has the same
bits (the persistent kernel's queue, lane refills and hit-list order vary from run to run; the
result must not).  usage: tools/soak.py [config] [repetitions]"""
import os, sys, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
pkg = entry.load_package()
from path_tracing_amd import scenes
cfg = scenes.config(int(sys.argv[1]) if len(sys.argv) > 1 else 4)
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
tris, bvh = scenes.mesh(cfg["mesh"])
p = pkg.default_params(width=cfg["width"], height=cfg["height"], rays_per_pixel=cfg["rays_per_pixel"],
                       max_bounce_count=cfg["max_bounce_count"])
with pkg.Renderer(p) as r:
    r.upload_scene(scenes.spheres_for(cfg["spheres"]), tris, bvh)
    r.set_camera(cfg["camera_position"], cfg["camera_rotation"])
    sums = set()
    for k in range(reps):
        r.render_frames(0, 2)
        sums.add(zlib.crc32(r.read_accum(4).tobytes()))
    print("%s: %d renders of 2 frames, %d distinct image checksum(s)" % (cfg["name"], reps, len(sums)))
    sys.exit(0 if len(sums) == 1 else 1)
