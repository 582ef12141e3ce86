"""Benchmark configurations and data-contract facts.  CPU only."""
import os

import numpy as np


def test_committed_scene_files_have_the_reference_layout(scenes):
    sizes = {"teapot": (15704, 30661), "knight": (456, 423), "dragon_8k": (8712, 17147)}
    for name, (nt, nn) in sizes.items():
        assert os.path.getsize(os.path.join(scenes.ASSETS, name + ".bin")) == nt * 36
        assert os.path.getsize(os.path.join(scenes.ASSETS, name + "_bvh.bin")) == nn * 36
        t, b = scenes.mesh(name)
        assert t.shape == (nt, 9) and b.shape == (nn, 9)
        assert b[0, 6] == 0 and b[0, 8] != 0                       # root is node 0 and is an inner node
    assert scenes.load_bin("dragon_80k.bin").shape == (87130, 9)
    assert scenes.load_bin("small_knight.bin").shape == (44, 9)


def test_configs_follow_baseline(scenes):
    c = scenes.CONFIGS
    assert (c[1]["width"], c[1]["height"], c[1]["rays_per_pixel"] * c[1]["n_frames"], c[1]["max_bounce_count"]) == (640, 360, 1, 4)
    assert (c[2]["rays_per_pixel"] * c[2]["n_frames"], c[2]["max_bounce_count"]) == (16, 8)
    assert (c[3]["rays_per_pixel"] * c[3]["n_frames"], c[3]["max_bounce_count"]) == (64, 8)
    assert (c[4]["width"], c[4]["height"], c[4]["rays_per_pixel"] * c[4]["n_frames"], c[4]["max_bounce_count"]) == (1920, 1080, 256, 8)
    assert (c[5]["width"], c[5]["height"], c[5]["rays_per_pixel"] * c[5]["n_frames"], c[5]["max_bounce_count"]) == (3840, 2160, 1024, 12)
    for n in c:
        assert c[n]["rays_per_pixel"] == min(16, c[n]["rays_per_pixel"] * c[n]["n_frames"])


def test_sphere_lists(scenes):
    s = scenes.knight_spheres()
    assert s.shape == (16, 12) and s.dtype == np.float32
    assert np.array_equal(s, scenes.knight_spheres())               # fixed seed
    assert ((s[:, 3] >= 10) & (s[:, 3] <= 30)).all()
    assert {round(float(x), 2) for x in s[:, 11]} <= {0.0, 0.25, 0.5, 0.75, 0.9, 1.0}
    assert (s[:, 10] > 0).any() and (s[:, 10] == 0).any()
    # the red light of src/main.rs:300-302
    assert list(scenes.RED_LIGHT[0]) == [40, 40, -20, 10, 0, 0, 0, 1, 0, 0, 10, 0]


def test_every_config_camera_sees_its_scene(scenes, oracle):
    for n in (1, 2, 5):
        cfg = scenes.config(n)
        tris, bvh = scenes.mesh(cfg["mesh"])
        p = oracle.default_params(width=96, height=54, rays_per_pixel=1)
        sc = oracle.scene(scenes.spheres_for(cfg["spheres"]), tris, bvh)
        ids, _ = oracle.primary_hits(p, sc, cfg["camera_position"], cfg["camera_rotation"])
        assert (ids != -2 ** 31).mean() > 0.1, n
