#!/usr/bin/env python3
"""Random sequences of API calls (renders, sums, renders with a second output handed to a stream of the caller's, camera
moves, uploads, refits, shards, ...) applied to two contexts, one with the two batch slots and the kept primary pass, one
with a single stream (PT_PIPELINE=0) and the full grid (PT_ADAPT_GRID=0); after every read the two accumulation buffers
(and the two second outputs) must hold the same bits (needs a GPU).  usage: tools/fuzz_calls.py [seed] [operations]"""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import __graft_entry__ as entry
pkg = entry.load_package()
from path_tracing_amd import scenes

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n_ops = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
rng = random.Random(seed)
W, H, RPP = 96, 54, 3
cfg = scenes.config(1, width=W, height=H, rays_per_pixel=RPP)
meshes = [scenes.mesh("teapot"), scenes.mesh("dragon_8k")]
spheres = scenes.spheres_for(cfg["spheres"])
p = pkg.default_params(width=W, height=H, rays_per_pixel=RPP, max_bounce_count=4, flags=pkg.PT_FLAG_EXACT_MATH)

def make(env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return pkg.Renderer(p)
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v

a = make({})
b = make({"PT_PIPELINE": "0", "PT_ADAPT_GRID": "0"})
both = (a, b)
mesh = 0
for r in both:
    r.upload_scene(spheres, *meshes[mesh])
poses = [(list(cfg["camera_position"]), list(cfg["camera_rotation"])),
         ([cfg["camera_position"][0] + 3.0, cfg["camera_position"][1], cfg["camera_position"][2]], list(cfg["camera_rotation"])),
         (list(cfg["camera_position"]), [cfg["camera_rotation"][0], cfg["camera_rotation"][1] + 7.0, cfg["camera_rotation"][2]])]
frame = 0
reads = 0
log = []
# second outputs (pt_next_render_output), one per context, and a stream of the caller's that takes part in the hand-over
outs = [torch.zeros(H * W * 4, dtype=torch.float32, device="cuda") for _ in both]
side = torch.cuda.Stream()
torch.cuda.synchronize()
for k in range(n_ops):
    x = rng.random()
    if x < 0.40:
        n = rng.randint(1, 3)
        if rng.random() < 0.15:
            frame = 0
        op = ("render", frame, n); frame += n
        for r in both: r.render_frames(op[1], n)
    elif x < 0.50:
        i = rng.randrange(3); op = ("camera", i); frame = 0
        for r in both: r.set_camera(*poses[i])
    elif x < 0.56:
        n = rng.randint(1, 2); op = ("sum", frame, n)
        for r in both: r.render_frames_sum(frame, n, 1)
        frame += n
    elif x < 0.585:
        n = rng.randint(1, 3); move = rng.random() < 0.5; op = ("render+output", frame, n, move)
        for r, out in zip(both, outs):
            r.accum_acquire(side.cuda_stream)
            r.next_render_output(out.data_ptr(), move=move)
            if move:
                r.render_frames_sum(frame, n, 1)
            else:
                r.render_frames(frame, n)
            r.accum_release(side.cuda_stream)
            with torch.cuda.stream(side):
                out.mul_(1.0)                   # the caller's stream reads and writes the image it was handed
        frame += n
    elif x < 0.60:
        op = ("reset",); frame = 0
        for r in both: r.reset_accum()
    elif x < 0.64:
        op = ("scale",)
        for r in both: r.scale_accum(0.5)
    elif x < 0.72:
        m = rng.randint(0, 1); op = ("wait", m)
        a.wait_in_flight(m); b.wait_in_flight(m)
    elif x < 0.76:
        n = rng.randint(1, 2); op = ("static", frame, n)
        for r in both:
            r.synchronize()                 # the static array is shared by the contexts of a GPU
            r.render_frames_static(frame, n)
            r.synchronize()
        frame += n
    elif x < 0.79:
        mesh ^= 1; op = ("upload", mesh); frame = 0
        for r in both: r.upload_scene(spheres, *meshes[mesh])
    elif x < 0.82 and mesh == 0:
        s = 1.0 + 0.1 * rng.random(); op = ("refit", s); frame = 0
        t = meshes[0][0].copy(); t[:, 1::3] *= np.float32(s)
        for r in both: r.refit_scene(t)
    elif x < 0.85:
        w = rng.choice((1, 2, 3)); rk = rng.randrange(w); op = ("shard", rk, w); frame = 0
        for r in both:
            r.set_pixel_shard(rk, w); r.reset_accum()
    else:
        op = ("read",)
        ia, ib = a.read_accum(4), b.read_accum(4)
        reads += 1
        side.synchronize()
        oa, ob = outs[0].cpu().numpy(), outs[1].cpu().numpy()
        if not np.array_equal(ia.view(np.uint32), ib.view(np.uint32)) or not np.array_equal(oa.view(np.uint32), ob.view(np.uint32)):
            print("MISMATCH after operation %d; last operations: %s" % (k, log[-12:]))
            sys.exit(1)
    log.append(op)
ia, ib = a.read_accum(4), b.read_accum(4)
ok = np.array_equal(ia.view(np.uint32), ib.view(np.uint32))
for r in both: r.close()
print("seed %d: %d operations, %d reads compared: %s" % (seed, n_ops, reads + 1, "all equal" if ok else "MISMATCH at the end"))
sys.exit(0 if ok else 1)
