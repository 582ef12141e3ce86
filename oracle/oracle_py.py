"""ctypes binding of the CPU oracle (oracle/libpt_oracle*.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The shipped package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


class OrcHit(C.Structure):
    _fields_ = [
        ("did_hit", C.c_int32), ("distance", C.c_float),
        ("position", C.c_float * 3), ("normal", C.c_float * 3), ("color", C.c_float * 3),
        ("emission_color", C.c_float * 3), ("emission_strength", C.c_float),
        ("smoothness", C.c_float), ("prim_id", C.c_int32),
    ]


class OrcParams(C.Structure):
    _fields_ = [
        ("width", C.c_uint32), ("height", C.c_uint32), ("rays_per_pixel", C.c_uint32),
        ("max_bounce_count", C.c_uint32), ("stack_entries", C.c_uint32),
        ("fov_degrees", C.c_float),
        ("tri_color", C.c_float * 3), ("tri_emission_color", C.c_float * 3),
        ("tri_emission_strength", C.c_float), ("tri_smoothness", C.c_float),
        ("sky_horizon", C.c_float * 3), ("sky_zenith", C.c_float * 3), ("ground", C.c_float * 3),
        ("sun_dir", C.c_float * 3), ("sun_intensity", C.c_float), ("sun_focus", C.c_float),
        ("use_environment_lighting", C.c_uint32), ("far_distance", C.c_float),
    ]


class OrcScene(C.Structure):
    _fields_ = [
        ("spheres", C.POINTER(C.c_float)), ("sphere_count", C.c_uint32),
        ("triangles", C.POINTER(C.c_float)), ("triangle_count", C.c_uint32),
        ("bvh", C.POINTER(C.c_float)), ("node_count", C.c_uint32),
    ]


class OrcCounters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in
                ("ray_casts", "box_tests", "tri_tests", "nodes_popped", "max_stack", "paths")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


def build():
    """Compile both oracle libraries (gcc, a few seconds)."""
    subprocess.run(["make", "-s", "-C", _HERE], check=True)


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


class Oracle:
    """One loaded oracle library. shared_math=True selects the build whose
    log/cos/pow are the portable polynomial versions (bit-exact with the HIP
    kernels' exact-math mode)."""

    def __init__(self, shared_math=False, variant=None):
        """variant: one of the oracle/variants builds (shared math with ONE built-in evaluated another way a WGSL
        implementation may choose: normalize_div, dot_fma, minmax_nan, camera_f32, contract); None = the pinned choices."""
        name = "libpt_oracle_shared.so" if shared_math else "libpt_oracle.so"
        if variant:
            assert shared_math, "the variants are builds of the shared-math oracle"
            name = os.path.join("variants", "libpt_oracle_%s.so" % variant)
        path = os.path.join(_HERE, name)
        if not os.path.exists(path):
            build()
        self.lib = L = C.CDLL(path)
        assert bool(L.orc_uses_shared_math()) == bool(shared_math)
        L.orc_random_value.restype = C.c_float
        L.orc_random_value.argtypes = [C.c_uint32]
        L.orc_random_normal.restype = C.c_float
        L.orc_random_normal.argtypes = [C.c_uint32]
        L.orc_random_direction.argtypes = [C.c_uint32, C.POINTER(C.c_float)]
        L.orc_random_hemisphere_direction.argtypes = [C.POINTER(C.c_float), C.c_uint32, C.POINTER(C.c_float)]
        L.orc_ray_box.restype = C.c_float
        L.orc_bvh_build.restype = C.c_long
        L.orc_bvh_build.argtypes = [C.POINTER(C.c_float), C.c_uint32, C.c_int,
                                    C.POINTER(C.c_float), C.POINTER(C.POINTER(C.c_float))]
        L.orc_bvh_free.argtypes = [C.POINTER(C.c_float)]
        L.orc_bvh_depth.argtypes = [C.POINTER(C.c_float), C.c_uint32]

    # ---- parameter / scene helpers -------------------------------------
    def default_params(self, **kw):
        p = OrcParams()
        self.lib.orc_default_params(C.byref(p))
        for k, v in kw.items():
            cur = getattr(p, k)
            if hasattr(cur, "__len__"):
                setattr(p, k, (C.c_float * 3)(*v))
            else:
                setattr(p, k, v)
        return p

    @staticmethod
    def scene(spheres, triangles, bvh):
        """Arrays are kept alive on the returned object."""
        sp = np.ascontiguousarray(np.asarray(spheres, dtype=np.float32).reshape(-1, 12))
        tr = np.ascontiguousarray(np.asarray(triangles, dtype=np.float32).reshape(-1, 9))
        bv = np.ascontiguousarray(np.asarray(bvh, dtype=np.float32).reshape(-1, 9))
        s = OrcScene(_fp(sp), sp.shape[0], _fp(tr), tr.shape[0], _fp(bv), bv.shape[0])
        s._keep = (sp, tr, bv)
        return s

    # ---- scalar pieces -------------------------------------------------
    def random_value(self, seed):
        return float(self.lib.orc_random_value(C.c_uint32(seed & 0xFFFFFFFF)))

    def random_normal(self, seed):
        return float(self.lib.orc_random_normal(C.c_uint32(seed & 0xFFFFFFFF)))

    def random_direction(self, seed):
        out = (C.c_float * 3)()
        self.lib.orc_random_direction(C.c_uint32(seed & 0xFFFFFFFF), out)
        return np.array(out[:], dtype=np.float32)

    def random_hemisphere_direction(self, normal, seed):
        out = (C.c_float * 3)()
        self.lib.orc_random_hemisphere_direction(_f3(normal), C.c_uint32(seed & 0xFFFFFFFF), out)
        return np.array(out[:], dtype=np.float32)

    def rotate_vector(self, v, angles_deg):
        out = (C.c_float * 3)()
        self.lib.orc_rotate_vector(_f3(v), _f3(angles_deg), out)
        return np.array(out[:], dtype=np.float32)

    def ray_box(self, origin, direction, box):
        b = (C.c_float * 6)(*[float(x) for x in box])
        return float(self.lib.orc_ray_box(_f3(origin), _f3(direction), b))

    def ray_triangle(self, params, origin, direction, tri):
        h = OrcHit()
        t = (C.c_float * 9)(*[float(x) for x in np.asarray(tri).reshape(-1)])
        self.lib.orc_ray_triangle(C.byref(params), _f3(origin), _f3(direction), t, C.byref(h))
        return h

    def ray_sphere(self, origin, direction, sphere):
        h = OrcHit()
        s = (C.c_float * 12)(*[float(x) for x in sphere])
        self.lib.orc_ray_sphere(_f3(origin), _f3(direction), s, C.byref(h))
        return h

    def environment_light(self, params, direction):
        out = (C.c_float * 3)()
        self.lib.orc_environment_light(C.byref(params), _f3(direction), out)
        return np.array(out[:], dtype=np.float32)

    def screen_size(self, params):
        w, h = C.c_float(), C.c_float()
        self.lib.orc_screen_size(C.byref(params), C.byref(w), C.byref(h))
        return w.value, h.value

    def primary_ray(self, params, cam_pos, cam_rot, px, py):
        o, d, idx = (C.c_float * 3)(), (C.c_float * 3)(), C.c_uint32()
        self.lib.orc_primary_ray(C.byref(params), _f3(cam_pos), _f3(cam_rot), C.c_uint32(px),
                                 C.c_uint32(py), o, d, C.byref(idx))
        return np.array(o[:], np.float32), np.array(d[:], np.float32), idx.value

    def closest_hit(self, params, scene, origin, direction, counters=None, brute=False, bvh_only=False):
        h = OrcHit()
        cref = C.byref(counters) if counters is not None else None
        if brute:
            self.lib.orc_ray_triangles_brute(C.byref(params), C.byref(scene), _f3(origin), _f3(direction), C.byref(h))
        elif bvh_only:
            self.lib.orc_ray_triangle_bvh(C.byref(params), C.byref(scene), _f3(origin), _f3(direction), C.byref(h), cref)
        else:
            self.lib.orc_calculate_ray_collision(C.byref(params), C.byref(scene), _f3(origin), _f3(direction), C.byref(h), cref)
        return h

    def trace(self, params, scene, origin, direction, seed, frame_count, counters=None):
        out = (C.c_float * 3)()
        cref = C.byref(counters) if counters is not None else None
        self.lib.orc_trace(C.byref(params), C.byref(scene), _f3(origin), _f3(direction),
                           C.c_uint32(seed & 0xFFFFFFFF), C.c_uint32(frame_count), out, cref)
        return np.array(out[:], dtype=np.float32)

    # ---- frames --------------------------------------------------------
    def render_frames(self, params, scene, cam_pos, cam_rot, first_frame, n_frames,
                      accum=None, nthreads=0, counters=None):
        """Returns the (H, W, 4) accumulation buffer (16-byte pixel stride)."""
        if accum is None:
            accum = np.zeros((params.height, params.width, 4), dtype=np.float32)
        cref = C.byref(counters) if counters is not None else None
        rc = self.lib.orc_render_frames(C.byref(params), C.byref(scene), _f3(cam_pos), _f3(cam_rot),
                                        C.c_uint32(first_frame), C.c_uint32(n_frames), _fp(accum),
                                        C.c_int(nthreads), cref)
        if rc:
            raise ValueError("orc_render_frames: bad arguments")
        return accum

    def render_frames_sum(self, params, scene, cam_pos, cam_rot, first_frame, n_frames, frame_stride,
                          accum=None, nthreads=0, counters=None):
        if accum is None:
            accum = np.zeros((params.height, params.width, 4), dtype=np.float32)
        cref = C.byref(counters) if counters is not None else None
        rc = self.lib.orc_render_frames_sum(C.byref(params), C.byref(scene), _f3(cam_pos), _f3(cam_rot),
                                            C.c_uint32(first_frame), C.c_uint32(n_frames),
                                            C.c_uint32(frame_stride), _fp(accum), C.c_int(nthreads), cref)
        if rc:
            raise ValueError("orc_render_frames_sum: bad arguments")
        return accum

    def render_frame_region(self, params, scene, cam_pos, cam_rot, frame_count, accum,
                            x0, x1, sx, y0, y1, sy, nthreads=0, counters=None):
        cref = C.byref(counters) if counters is not None else None
        rc = self.lib.orc_render_frame(C.byref(params), C.byref(scene), _f3(cam_pos), _f3(cam_rot),
                                       C.c_uint32(frame_count), _fp(accum),
                                       C.c_uint32(x0), C.c_uint32(x1), C.c_uint32(sx),
                                       C.c_uint32(y0), C.c_uint32(y1), C.c_uint32(sy),
                                       C.c_int(nthreads), cref)
        if rc:
            raise ValueError("orc_render_frame: bad arguments")
        return accum

    def primary_hits(self, params, scene, cam_pos, cam_rot, nthreads=0):
        ids = np.empty((params.height, params.width), dtype=np.int32)
        dist = np.empty((params.height, params.width), dtype=np.float32)
        rc = self.lib.orc_primary_hits(C.byref(params), C.byref(scene), _f3(cam_pos), _f3(cam_rot),
                                       ids.ctypes.data_as(C.POINTER(C.c_int32)), _fp(dist), C.c_int(nthreads))
        if rc:
            raise ValueError("orc_primary_hits: bad arguments")
        return ids, dist

    # ---- builder -------------------------------------------------------
    def bvh_build(self, triangles, max_depth):
        tr = np.ascontiguousarray(np.asarray(triangles, dtype=np.float32).reshape(-1, 9))
        out_tr = np.empty_like(tr)
        nodes_p = C.POINTER(C.c_float)()
        n = self.lib.orc_bvh_build(_fp(tr), tr.shape[0], int(max_depth), _fp(out_tr), C.byref(nodes_p))
        if n < 0:
            raise MemoryError("orc_bvh_build")
        nodes = np.ctypeslib.as_array(nodes_p, shape=(n, 9)).copy()
        self.lib.orc_bvh_free(nodes_p)
        return out_tr, nodes

    def bvh_depth(self, bvh):
        bv = np.ascontiguousarray(np.asarray(bvh, dtype=np.float32).reshape(-1, 9))
        return int(self.lib.orc_bvh_depth(_fp(bv), bv.shape[0]))
