#!/usr/bin/env python3
"""Determinism soak: renders the same frames many times and checks that every image has the same
bits (the persistent kernel's queue, lane refills and hit-list order vary from run to run, batches of
consecutive calls overlap on two streams; the result must not depend on any of it).  Each repetition
enqueues, without waiting in between: a running mean over calls of 1, 2, 1, 3 frames, a camera move and
back (which restarts it), the same again.  usage: tools/soak.py [config] [repetitions]"""
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
pos, rot = list(cfg["camera_position"]), list(cfg["camera_rotation"])
away = [pos[0] + 2.0, pos[1], pos[2]]
with pkg.Renderer(p) as r:
    r.upload_scene(scenes.spheres_for(cfg["spheres"]), tris, bvh)
    sums = set()
    for k in range(reps):
        r.set_camera(pos, rot)
        f = 0
        for n in (1, 2, 1, 3):
            r.render_frames(f, n); f += n
        r.set_camera(away, rot)
        r.render_frames(0, 1)
        r.set_camera(pos, rot)
        f = 0
        for n in (1, 2, 1, 3):
            r.render_frames(f, n); f += n
            if k % 3 == 0:
                r.wait_in_flight(1)
        sums.add(zlib.crc32(r.read_accum(4).tobytes()))
    print("%s: %d repetitions of 7 frames in 4 calls, %d distinct image checksum(s)" % (cfg["name"], reps, len(sums)))
    sys.exit(0 if len(sums) == 1 else 1)
