"""How much of "bit-exact with the reference" hangs on the built-ins WGSL leaves to the implementation.

No output of the reference shader under a real WGSL compiler and driver exists here, so oracle, kernels and the WGSL
interpreter of tests/golden pin one evaluation of `normalize`, `dot`, `min` / `max` on NaN and the camera's trigonometry
(oracle/pt_oracle.h).  oracle/variants/ holds builds of the oracle with ONE of those choices made another legal way,
plus one that lets the C compiler fuse every a * b + c as GPU shader compilers do.  This file
  - checks, where the reference shader is on the machine, that each variant build computes what the shader's text
    computes under the same choice (interpreter with `choices`), bit for bit, on a few casts and paths;
  - counts what changes against the pinned oracle on the five benchmark scenes (tests/golden/builtin_variants.json is
    the committed table) and bounds it: the primary id moves on at most one pixel in 57 600,
    at most one pixel in a thousand moves by more than 2e-3 relative.
"""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHADER = "/root/reference/src/shaders/shader.wgsl"
TABLE = os.path.join(ROOT, "tests", "golden", "builtin_variants.json")
VARIANTS = ("normalize_div", "dot_fma", "minmax_nan", "camera_f32", "contract")
CHOICES = {"normalize_div": {"normalize": "div"}, "dot_fma": {"dot": "fma"}, "minmax_nan": {"minmax": "nan"},
           "camera_f32": {"camera": "f32"}}
SIZES = {1: (320, 180, 2), 2: (320, 180, 2), 3: (320, 180, 2), 4: (320, 180, 2), 5: (320, 180, 2)}      # width, height, rays per pixel; one frame


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def _has_fma():
    try:
        return " fma " in open("/proc/cpuinfo").read().replace("\n", " ")
    except OSError:
        return False


def _render(o, scenes, n):
    w, h, rpp = SIZES[n]
    cfg = scenes.config(n, width=w, height=h, rays_per_pixel=rpp)
    tris, bvh = scenes.mesh(cfg["mesh"])
    sp = scenes.spheres_for(cfg["spheres"])
    p = o.default_params(width=w, height=h, rays_per_pixel=rpp, max_bounce_count=cfg["max_bounce_count"])
    sc = o.scene(sp, tris, bvh)
    ids, dist = o.primary_hits(p, sc, cfg["camera_position"], cfg["camera_rotation"])
    img = o.render_frames(p, sc, cfg["camera_position"], cfg["camera_rotation"], 0, 1)
    return ids, dist, img[..., :3]


def measure(orc_mod, scenes):
    """The table: per variant and configuration, what differs from the pinned oracle."""
    base = orc_mod.Oracle(shared_math=True)
    out = {}
    for n in SIZES:
        ref = _render(base, scenes, n)
        hit = ref[0] != -(2 ** 31)
        for v in VARIANTS:
            if v == "contract" and not _has_fma():
                continue
            ids, dist, img = _render(orc_mod.Oracle(shared_math=True, variant=v), scenes, n)
            rel = np.abs(img - ref[2]) / (np.abs(ref[2]) + 1e-4)
            out["%s/cfg%d" % (v, n)] = {
                "pixels": int(hit.size), "hit_pixels": int(hit.sum()),
                "primary_ids_differ": int((ids != ref[0]).sum()),
                "primary_distance_bits_differ": int((bits(dist) != bits(ref[1]))[hit].sum()),
                "pixels_with_any_bit_different": int((bits(img) != bits(ref[2])).any(-1).sum()),
                "pixels_beyond_2e-3_relative": int((rel.max(-1) > 2e-3).sum()),
                "rmse_over_mean": float(np.sqrt(((img.astype(np.float64) - ref[2]) ** 2).mean()) / ref[2].mean()),
            }
    return out


def test_variant_table_is_current_and_small(orc_mod, scenes):
    got = measure(orc_mod, scenes)
    if os.environ.get("PT_WRITE_VARIANT_TABLE") == "1":
        with open(TABLE, "w") as f:
            json.dump(got, f, indent=1, sort_keys=True)
    want = json.load(open(TABLE))
    for k, row in got.items():
        assert k in want, k
        for field, val in row.items():
            if field == "rmse_over_mean":
                assert abs(val - want[k][field]) <= 1e-9 + 1e-6 * abs(want[k][field]), (k, field)
            else:
                assert val == want[k][field], (k, field, val, want[k][field])
        # the error bar itself: the primary id moves on one pixel in 57 600 at most, distances and radiance move in their
        # last bits on up to half of the pixels, and a pixel in a few thousand (paths that meet the mirror spheres of
        # configuration 5 several times) ends on another path: RMSE / mean up to 5e-3 for one frame of two samples
        assert row["primary_ids_differ"] <= 1, k
        assert row["pixels_beyond_2e-3_relative"] <= row["pixels"] // 1000, k
        assert row["rmse_over_mean"] < 1e-2, k


@pytest.mark.skipif(not os.path.exists(SHADER), reason="the reference is not on this machine")
@pytest.mark.parametrize("variant", sorted(CHOICES))
def test_variant_builds_compute_what_the_shader_text_computes_under_the_same_choice(variant, orc_mod, scenes):
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_wgsl_golden as mk
    import wgsl_interp as W
    G = dict(np.load(os.path.join(ROOT, "tests", "golden", "wgsl_golden.npz")))
    src = open(SHADER).read()
    cfg = scenes.config(1)
    tris, bvh = scenes.mesh(cfg["mesh"])
    spheres = np.asarray(scenes.spheres_for(cfg["spheres"]), dtype=np.float32).reshape(-1, 12)
    w, h, rpp, bounces, frames, depth = [int(x) for x in G["scene_width_height_rpp_bounces_frames_depth"]]
    over = dict(sphere_count=len(spheres), triangle_count=len(tris), bvh_node_count=len(bvh), bvh_max_depth=depth,
                max_bounce_count=bounces, rays_per_pixel=rpp, screen_size=(float(w), float(h)))
    I = W.Interpreter(src, W.Builtins(W.SharedMath(mk.build_shim()), CHOICES[variant]), over)
    mk.bind_scene(I, spheres, tris, bvh, cfg["camera_position"], cfg["camera_rotation"], w, h)
    o = orc_mod.Oracle(shared_math=True, variant=variant)
    p = o.default_params(width=w, height=h, rays_per_pixel=rpp, max_bounce_count=bounces)
    sc = o.scene(spheres, tris, bvh)
    k = slice(0, 10)
    for a, d, s, f in zip(G["cast_origins"][k], G["cast_dirs"][k], G["trace_seeds"][k], G["trace_frame_counts"][k]):
        I.bind("frame_count", W.U32(int(f)))
        want = mk.fvec(I.call("trace", [mk.ray_struct(I, a, d), W.U32(int(s))]))
        got = np.asarray(o.trace(p, sc, a, d, int(s), int(f)), dtype=np.float32)
        assert np.array_equal(bits(got), bits(want)) or (np.isnan(got) == np.isnan(want)).all() and np.array_equal(bits(got)[~np.isnan(got)], bits(want)[~np.isnan(want)])
    # the camera ray and the first hit of a few pixels through vs_main / fs_main's own arithmetic
    if variant == "camera_f32":
        for v, ang in zip(G["rot_vectors"][:6], G["rot_angles"][:6]):
            want = mk.fvec(I.call("rotate_vector", [W.vec3(*v), W.vec3(*ang)]))
            assert np.array_equal(bits(o.rotate_vector(v, ang)), bits(want))
