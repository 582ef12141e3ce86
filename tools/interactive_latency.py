#!/usr/bin/env python3
"""Frame time of the reference's interactive loop (one State::render per tick, src/main.rs:674-676) at
the reference's own constants: 1200x600, 20 rays per pixel, 10 bounces, teapot + its sphere list.  Two
pacings: the host waits for every frame before it builds the next one, or - like the reference, whose
render() returns after queue.submit and whose surface allows a frame in flight - for the frame before."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
pkg = entry.load_package()
from path_tracing_amd import scenes
mesh = sys.argv[1] if len(sys.argv) > 1 else "teapot"
tris, bvh = scenes.mesh(mesh)
p = pkg.default_params()
with pkg.Renderer(p) as r:
    r.upload_scene(scenes.spheres_for("cfg1"), tris, bvh)
    cfg = scenes.config(1)
    r.set_camera(cfg["camera_position"], cfg["camera_rotation"])
    for pacing in ("wait for every frame", "one frame in flight"):
        for mode in ("still", "moving"):
            r.render(3); r.synchronize(); r.reset_timing()
            n = 100
            t0 = time.perf_counter()
            for k in range(n):
                if mode == "moving":
                    r.update_camera(["L"] if k % 2 else ["J"])
                r.render(1)
                if pacing == "wait for every frame":
                    r.synchronize()           # a displayed frame
                else:
                    r.wait_in_flight(1)       # the frame before this one is displayed
            r.synchronize()
            dt = (time.perf_counter() - t0) / n
            t = r.timing()
            print("%s %s, %s: %.3f ms per frame (%.0f fps), kernels primary %.3f path %.3f resolve %.3f ms, %.0f Msamples/s" % (
                mesh, mode, pacing, dt * 1e3, 1 / dt, t["primary_ms"] / n, t["path_ms"] / n, t["resolve_ms"] / n,
                p.width * p.height * p.rays_per_pixel / dt / 1e6))
