"""Python host side of the MI355X path-tracing backend.

A thin ctypes layer over libpt_hip.so (the C ABI in include/pt_api.h) plus
`Renderer`, which mirrors the reference host's `State` (src/main.rs:17-41,
144-433, 444-523): it owns the scene buffers, camera and frame counter and
exposes `render()` with the reference's per-frame protocol.

The directory name contains a hyphen, so load it with `load_package()` from
`__graft_entry__` (or importlib) under the module name `path_tracing_amd`.

There is no CPU fallback: if libpt_hip.so is missing or no gfx950 GPU is
present, construction fails with an exception.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpt_hip.so")

PT_OK = 0
PT_EINVAL, PT_EHIP, PT_ENOMEM, PT_EIO, PT_ESTATE = -1, -2, -3, -4, -5      # include/pt_api.h
PT_FLAG_EXACT_MATH = 1
PT_FLAG_COUNT_WORK = 2
PT_FLAG_WIDE_BVH = 4

MISS_ID = -(2 ** 31)


class PtParams(C.Structure):
    _fields_ = [
        ("width", C.c_uint32), ("height", C.c_uint32), ("rays_per_pixel", C.c_uint32),
        ("max_bounce_count", C.c_uint32), ("fov_degrees", C.c_float),
        ("tri_color", C.c_float * 3), ("tri_emission_color", C.c_float * 3),
        ("tri_emission_strength", C.c_float), ("tri_smoothness", C.c_float),
        ("sky_horizon", C.c_float * 3), ("sky_zenith", C.c_float * 3), ("ground", C.c_float * 3),
        ("sun_dir", C.c_float * 3), ("sun_intensity", C.c_float), ("sun_focus", C.c_float),
        ("use_environment_lighting", C.c_uint32), ("far_distance", C.c_float), ("bvh_max_depth", C.c_uint32),
        ("flags", C.c_uint32),
    ]


class PtCounters(C.Structure):
    _fields_ = [("ray_casts", C.c_uint64), ("box_tests", C.c_uint64),
                ("tri_tests", C.c_uint64), ("paths", C.c_uint64),
                ("path_kernel_ray_casts", C.c_uint64), ("path_kernel_box_tests", C.c_uint64),
                ("path_kernel_tri_tests", C.c_uint64)]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class PtTiming(C.Structure):
    _fields_ = [("frames", C.c_uint64), ("batches", C.c_uint64), ("primary_ms", C.c_double),
                ("path_ms", C.c_double), ("resolve_ms", C.c_double)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


# every symbol include/pt_api.h declares
API_SYMBOLS = [
    "pt_default_params", "pt_create", "pt_upload_scene", "pt_load_scene_files", "pt_set_camera",
    "pt_render_frames", "pt_render_frames_sum", "pt_scale_accum", "pt_read_accum", "pt_reset_accum",
    "pt_bind_accum", "pt_accum_device_ptr", "pt_primary_hits", "pt_synchronize", "pt_get_counters",
    "pt_reset_counters", "pt_get_debug_stats", "pt_get_timing", "pt_reset_timing", "pt_scene_info", "pt_destroy",
    "pt_last_error", "pt_bvh_build", "pt_free", "pt_bvh_build_gpu", "pt_bvh_build_gpu_error", "pt_bvh_build_gpu_release_cache", "pt_refit_scene", "pt_rebuild_scene",
    "pt_stl_format", "pt_stl_format_file", "pt_stl_format_error",
    "pt_update_camera", "pt_host_rotate_vector", "pt_set_pixel_shard", "pt_probe_functions", "pt_probe_intersections", "pt_write_accum",
    "pt_render_frames_static", "pt_static_scratch_bytes", "pt_wait_in_flight", "pt_accum_release", "pt_accum_acquire",
    "pt_next_render_output",
]

# key order of State::keys_pressed (main.rs:39)
KEYS = ("W", "S", "D", "A", "Space", "Shift", "I", "K", "L", "J", "O", "U")
CAMERA_SPEED = 1.0        # main.rs:17
CAMERA_ROT_SPEED = 1.0    # main.rs:18


class PtError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"pt error {code}: {message}")
        self.code = code


_lib = None


def load_library():
    """Loads libpt_hip.so; raises if the HIP extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FileNotFoundError(
            f"{LIB_PATH} is missing: build it with `make -C {_HERE}` (there is no CPU fallback)")
    # PyTorch wheels bundle their own libamdhip64.so.7.  Two HIP runtimes in one
    # process cannot see each other's allocations (pt_bind_accum on a torch tensor
    # would fail), so torch is loaded first when it is installed: the dynamic linker
    # then resolves this library's libamdhip64.so.7 dependency to the copy that is
    # already mapped.  Without torch the system ROCm runtime is used.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    fp = C.POINTER(C.c_float)
    L.pt_default_params.argtypes = [C.POINTER(PtParams)]
    L.pt_create.argtypes = [C.c_int, C.POINTER(PtParams), C.POINTER(C.c_void_p)]
    L.pt_upload_scene.argtypes = [C.c_void_p, fp, C.c_uint32, fp, C.c_uint32, fp, C.c_uint32]
    L.pt_refit_scene.argtypes = [C.c_void_p, fp]
    L.pt_rebuild_scene.argtypes = [C.c_void_p, fp, C.c_uint32, C.c_int, fp]
    L.pt_load_scene_files.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, fp, C.c_uint32]
    L.pt_set_camera.argtypes = [C.c_void_p, fp, fp]
    L.pt_render_frames.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
    L.pt_render_frames_sum.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]
    L.pt_render_frames_static.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
    L.pt_static_scratch_bytes.argtypes = []
    L.pt_static_scratch_bytes.restype = C.c_size_t
    L.pt_scale_accum.argtypes = [C.c_void_p, C.c_float]
    L.pt_set_pixel_shard.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
    L.pt_probe_intersections.argtypes = [C.c_void_p, fp, fp, fp, fp, C.c_uint32, fp]
    L.pt_probe_functions.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), fp, fp, C.c_uint32, fp]
    L.pt_read_accum.argtypes = [C.c_void_p, fp, C.c_uint32]
    L.pt_write_accum.argtypes = [C.c_void_p, fp, C.c_uint32]
    L.pt_reset_accum.argtypes = [C.c_void_p]
    L.pt_bind_accum.argtypes = [C.c_void_p, C.c_void_p]
    L.pt_accum_device_ptr.argtypes = [C.c_void_p]
    L.pt_accum_device_ptr.restype = C.c_void_p
    L.pt_primary_hits.argtypes = [C.c_void_p, C.POINTER(C.c_int32), fp]
    L.pt_synchronize.argtypes = [C.c_void_p]
    L.pt_wait_in_flight.argtypes = [C.c_void_p, C.c_uint32]
    L.pt_accum_release.argtypes = [C.c_void_p, C.c_void_p]
    L.pt_accum_acquire.argtypes = [C.c_void_p, C.c_void_p]
    L.pt_next_render_output.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
    L.pt_get_counters.argtypes = [C.c_void_p, C.POINTER(PtCounters)]
    L.pt_reset_counters.argtypes = [C.c_void_p]
    L.pt_get_debug_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    L.pt_get_timing.argtypes = [C.c_void_p, C.POINTER(PtTiming)]
    L.pt_reset_timing.argtypes = [C.c_void_p]
    L.pt_scene_info.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.pt_destroy.argtypes = [C.c_void_p]
    L.pt_destroy.restype = None
    L.pt_last_error.argtypes = [C.c_void_p]
    L.pt_last_error.restype = C.c_char_p
    L.pt_bvh_build.argtypes = [fp, C.c_uint32, C.c_int, fp, C.POINTER(fp), C.POINTER(C.c_uint32)]
    L.pt_bvh_build_gpu.argtypes = [C.c_int, fp, C.c_uint32, C.c_int, fp, C.POINTER(fp), C.POINTER(C.c_uint32), fp]
    L.pt_bvh_build_gpu_error.argtypes = []
    L.pt_bvh_build_gpu_release_cache.argtypes = []
    L.pt_bvh_build_gpu_release_cache.restype = None
    L.pt_bvh_build_gpu_error.restype = C.c_char_p
    L.pt_free.argtypes = [C.c_void_p]
    L.pt_free.restype = None
    L.pt_stl_format.argtypes = [C.c_char_p, C.c_size_t, C.c_double, C.POINTER(fp), C.POINTER(C.c_uint64)]
    L.pt_stl_format_file.argtypes = [C.c_char_p, C.c_char_p, C.c_double, C.POINTER(C.c_uint64)]
    L.pt_stl_format_error.argtypes = []
    L.pt_stl_format_error.restype = C.c_char_p
    L.pt_update_camera.argtypes = [C.POINTER(C.c_uint8), C.c_float, C.c_float, fp, fp, C.POINTER(C.c_int)]
    L.pt_host_rotate_vector.argtypes = [fp, fp, fp]
    _lib = L
    return L


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


def default_params(**overrides):
    """pt_params initialised to the committed reference constants
    (shader.wgsl:24-54, 283-286), with keyword overrides."""
    p = PtParams()
    rc = load_library().pt_default_params(C.byref(p))
    if rc:
        raise PtError(rc, "pt_default_params")
    for k, v in overrides.items():
        cur = getattr(p, k)
        if hasattr(cur, "__len__"):
            setattr(p, k, (C.c_float * 3)(*[float(x) for x in v]))
        else:
            setattr(p, k, v)
    return p


def bvh_build(triangles, max_depth):
    """Host BVH builder (prepare_data/src/main.rs). Returns (reordered_triangles[n,9], nodes[m,9])."""
    L = load_library()
    tr = np.ascontiguousarray(np.asarray(triangles, dtype=np.float32).reshape(-1, 9))
    out_tr = np.empty_like(tr)
    nodes_p = C.POINTER(C.c_float)()
    n = C.c_uint32()
    rc = L.pt_bvh_build(_fp(tr), tr.shape[0], int(max_depth), _fp(out_tr), C.byref(nodes_p), C.byref(n))
    if rc:
        raise PtError(rc, "pt_bvh_build failed")
    nodes = np.ctypeslib.as_array(nodes_p, shape=(n.value, 9)).copy()
    L.pt_free(nodes_p)
    return out_tr, nodes


def bvh_build_gpu(triangles, max_depth, device=0, return_ms=False):
    """The same builder on the GPU (pt_bvh_build_gpu): same outputs as bvh_build, bit for bit.
    With return_ms also the GPU time of the build in milliseconds (transfers excluded)."""
    L = load_library()
    tr = np.ascontiguousarray(np.asarray(triangles, dtype=np.float32).reshape(-1, 9))
    out_tr = np.empty_like(tr)
    nodes_p = C.POINTER(C.c_float)()
    n = C.c_uint32()
    ms = C.c_float()
    rc = L.pt_bvh_build_gpu(int(device), _fp(tr), tr.shape[0], int(max_depth), _fp(out_tr), C.byref(nodes_p), C.byref(n), C.byref(ms))
    if rc:
        raise PtError(rc, (L.pt_bvh_build_gpu_error() or b"pt_bvh_build_gpu failed").decode())
    nodes = np.ctypeslib.as_array(nodes_p, shape=(n.value, 9)).copy()
    L.pt_free(nodes_p)
    return (out_tr, nodes, float(ms.value)) if return_ms else (out_tr, nodes)


def _key_array(keys):
    """keys: iterable of key names from KEYS (or a 12-element sequence of truth values)."""
    keys = list(keys)
    if len(keys) == 12 and not any(isinstance(k, str) for k in keys):
        flags = [1 if k else 0 for k in keys]
    else:
        flags = [0] * 12
        for k in keys:
            flags[KEYS.index(k)] = 1
    return (C.c_uint8 * 12)(*flags)


def update_camera(keys, position, rotation_degrees, speed=CAMERA_SPEED, rot_speed=CAMERA_ROT_SPEED):
    """One tick of State::update_camera (main.rs:45-115) on plain values.
    Returns (position, rotation_degrees, any_key) with f32-exact list values."""
    L = load_library()
    pos, rot, any_key = _f3(position), _f3(rotation_degrees), C.c_int()
    rc = L.pt_update_camera(_key_array(keys), float(speed), float(rot_speed), pos, rot, C.byref(any_key))
    if rc:
        raise PtError(rc, "pt_update_camera")
    return [pos[0], pos[1], pos[2]], [rot[0], rot[1], rot[2]], bool(any_key.value)


def host_rotate_vector(v, rotation_degrees):
    """State::rotate_vector (main.rs:117-142)."""
    out = (C.c_float * 3)()
    rc = load_library().pt_host_rotate_vector(_f3(v), _f3(rotation_degrees), out)
    if rc:
        raise PtError(rc, "pt_host_rotate_vector")
    return [out[0], out[1], out[2]]


def stl_format(stl_text, scale=1.0):
    """ASCII STL text (bytes or str) -> flat f32 array, 9 values per triangle
    (prepare_data/data_formatter.js:9-18)."""
    L = load_library()
    data = stl_text.encode("utf-8") if isinstance(stl_text, str) else bytes(stl_text)
    out_p = C.POINTER(C.c_float)()
    n = C.c_uint64()
    rc = L.pt_stl_format(data, len(data), float(scale), C.byref(out_p), C.byref(n))
    if rc:
        raise PtError(rc, L.pt_stl_format_error().decode())
    out = np.ctypeslib.as_array(out_p, shape=(n.value,)).copy() if n.value else np.zeros(0, np.float32)
    L.pt_free(out_p)
    return out


def stl_format_file(stl_path, bin_path, scale=1.0):
    """Writes bin_path from stl_path; returns the number of f32 values written."""
    L = load_library()
    n = C.c_uint64()
    rc = L.pt_stl_format_file(os.fsencode(stl_path), os.fsencode(bin_path), float(scale), C.byref(n))
    if rc:
        raise PtError(rc, L.pt_stl_format_error().decode())
    return n.value


class Renderer:
    """Mirror of the reference's `State` (src/main.rs:17-41).

    render()            -> State::render (main.rs:444-523): one frame, frame_count += 1
    set_camera()/reset  -> update_camera's "any input resets frame_count" rule (main.rs:47-49)
    """

    def __init__(self, params, device=0):
        self._L = load_library()
        self.params = params
        self._ctx = C.c_void_p()
        rc = self._L.pt_create(int(device), C.byref(params), C.byref(self._ctx))
        if rc:
            raise PtError(rc, self._L.pt_last_error(None).decode())
        self.frame_count = 0                       # main.rs:420
        self.camera_position = [200.0, 0.0, 200.0]  # main.rs:382
        self.camera_rotation = [0.0, 90.0, 0.0]     # main.rs:383
        self.width, self.height = params.width, params.height

    # -- plumbing --------------------------------------------------------
    def _check(self, rc):
        if rc:
            raise PtError(rc, self._L.pt_last_error(self._ctx).decode())

    def close(self):
        if self._ctx:
            self._L.pt_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- scene (main.rs:286-355) ----------------------------------------
    def upload_scene(self, spheres, triangles, bvh):
        sp = np.ascontiguousarray(np.asarray(spheres, dtype=np.float32).reshape(-1, 12))
        tr = np.ascontiguousarray(np.asarray(triangles, dtype=np.float32).reshape(-1, 9))
        bv = np.ascontiguousarray(np.asarray(bvh, dtype=np.float32).reshape(-1, 9))
        self._check(self._L.pt_upload_scene(self._ctx, _fp(sp), sp.shape[0], _fp(tr), tr.shape[0],
                                            _fp(bv), bv.shape[0]))
        self.triangle_count = tr.shape[0]

    def load_scene_files(self, triangles_bin, bvh_bin, spheres=()):
        sp = np.ascontiguousarray(np.asarray(spheres, dtype=np.float32).reshape(-1, 12))
        self._check(self._L.pt_load_scene_files(self._ctx, os.fsencode(triangles_bin), os.fsencode(bvh_bin),
                                                _fp(sp), sp.shape[0]))
        self.triangle_count = os.path.getsize(triangles_bin) // 36

    def scene_info(self):
        d, i, l = C.c_uint32(), C.c_uint32(), C.c_uint32()
        self._check(self._L.pt_scene_info(self._ctx, C.byref(d), C.byref(i), C.byref(l)))
        return {"bvh_depth": d.value, "inner_nodes": i.value, "leaves": l.value}

    def rebuild_scene(self, triangles, max_depth, reset=True):
        """pt_rebuild_scene: a new tree for new triangles, built and packed on the GPU.  Returns the GPU time in ms."""
        tr = np.ascontiguousarray(np.asarray(triangles, dtype=np.float32).reshape(-1, 9))
        ms = C.c_float()
        self._check(self._L.pt_rebuild_scene(self._ctx, _fp(tr), tr.shape[0], int(max_depth), C.byref(ms)))
        self.triangle_count = tr.shape[0]
        if reset:
            self.frame_count = 0
        return float(ms.value)

    def refit_scene(self, triangles, reset=True):
        """pt_refit_scene: the triangles moved, the tree keeps its shape; the node boxes and triangle records
        are recomputed on the GPU.  Restarts the accumulation like a camera move unless reset is False."""
        tr = np.ascontiguousarray(np.asarray(triangles, dtype=np.float32).reshape(-1, 9))
        if tr.shape[0] != getattr(self, "triangle_count", -1):
            raise ValueError("refit_scene needs the %d triangles of the uploaded scene, got %d" % (getattr(self, "triangle_count", 0), tr.shape[0]))
        self._check(self._L.pt_refit_scene(self._ctx, _fp(tr)))
        if reset:
            self.frame_count = 0

    # -- camera (main.rs:45-115, 455-464) ---------------------------------
    def set_camera(self, position, rotation_degrees, reset=True):
        self.camera_position = [float(x) for x in position]
        self.camera_rotation = [float(x) for x in rotation_degrees]
        self._check(self._L.pt_set_camera(self._ctx, _f3(position), _f3(rotation_degrees)))
        if reset:
            self.frame_count = 0          # main.rs:47-49

    def update_camera(self, keys):
        """One timer tick with `keys` held (main.rs:674 -> :45-115): moves the camera and, if any
        key is held, restarts the accumulation.  Returns True when the pose changed."""
        pos, rot, any_key = update_camera(keys, self.camera_position, self.camera_rotation)
        if any_key:
            self.set_camera(pos, rot, reset=True)
        return any_key

    # -- frames ------------------------------------------------------------
    def render(self, n_frames=1):
        """State::render n times (main.rs:444-523)."""
        self._check(self._L.pt_render_frames(self._ctx, self.frame_count, int(n_frames)))
        self.frame_count += int(n_frames)   # main.rs:510

    def render_frames(self, first_frame_count, n_frames):
        self._check(self._L.pt_render_frames(self._ctx, int(first_frame_count), int(n_frames)))

    def render_frames_static(self, first_frame_count, n_frames):
        """pt_render_frames without any device allocation: working state in the library's static scratch."""
        self._check(self._L.pt_render_frames_static(self._ctx, int(first_frame_count), int(n_frames)))

    def render_frames_sum(self, first_frame_count, n_frames, frame_stride=1):
        self._check(self._L.pt_render_frames_sum(self._ctx, int(first_frame_count), int(n_frames), int(frame_stride)))

    def probe_functions(self, seeds, dirs, normal):
        """Device values of the small shader functions for each (seed, dir): dict of arrays."""
        seeds = np.ascontiguousarray(np.asarray(seeds, dtype=np.uint32))
        dirs = np.ascontiguousarray(np.asarray(dirs, dtype=np.float32).reshape(-1, 3))
        assert len(seeds) == len(dirs)
        out = np.zeros((len(seeds), 16), np.float32)
        self._check(self._L.pt_probe_functions(self._ctx, seeds.ctypes.data_as(C.POINTER(C.c_uint32)), _fp(dirs),
                                               _f3(normal), len(seeds), _fp(out)))
        return {"random_value": out[:, 0], "random_normal": out[:, 1], "random_direction": out[:, 2:5],
                "random_hemisphere_direction": out[:, 5:8], "environment_light": out[:, 8:11],
                "rotate_vector": out[:, 11:14]}

    def probe_intersections(self, rays, boxes, triangles, spheres):
        """Device values of ray_box (plain and reciprocal division), ray_triangle and ray_sphere."""
        a = [np.ascontiguousarray(np.asarray(x, dtype=np.float32).reshape(-1, w))
             for x, w in ((rays, 6), (boxes, 6), (triangles, 9), (spheres, 4))]
        n = len(a[0])
        assert all(len(x) == n for x in a)
        out = np.zeros((n, 8), np.float32)
        self._check(self._L.pt_probe_intersections(self._ctx, _fp(a[0]), _fp(a[1]), _fp(a[2]), _fp(a[3]), n, _fp(out)))
        return {"box": out[:, 0], "box_reciprocal": out[:, 1], "tri_hit": out[:, 2] != 0, "tri_distance": out[:, 3],
                "sphere_hit": out[:, 4] != 0, "sphere_distance": out[:, 5]}

    def set_pixel_shard(self, rank, world_size):
        """Render only the 8x8 tiles with (tile_x + tile_y) % world_size == rank (multi-GPU)."""
        self._check(self._L.pt_set_pixel_shard(self._ctx, int(rank), int(world_size)))

    def scale_accum(self, s):
        self._check(self._L.pt_scale_accum(self._ctx, float(s)))

    def reset_accum(self):
        self._check(self._L.pt_reset_accum(self._ctx))

    def read_accum(self, stride_floats=4):
        out = np.empty((self.height, self.width, stride_floats), dtype=np.float32)
        self._check(self._L.pt_read_accum(self._ctx, _fp(out), stride_floats))
        return out

    def write_accum(self, image, frames_done=None):
        """Restores a saved accumulation buffer ((H, W, 3) or (H, W, 4) float32); with frames_done
        the next render() continues the running mean from there."""
        a = np.ascontiguousarray(np.asarray(image, dtype=np.float32))
        if a.shape[:2] != (self.height, self.width) or a.shape[2] not in (3, 4):
            raise ValueError("image must be (height, width, 3 or 4)")
        self._check(self._L.pt_write_accum(self._ctx, _fp(a), a.shape[2]))
        if frames_done is not None:
            self.frame_count = int(frames_done)

    def bind_accum(self, device_ptr):
        self._check(self._L.pt_bind_accum(self._ctx, C.c_void_p(device_ptr)))

    def accum_release(self, hip_stream):
        """pt_accum_release: work enqueued on `hip_stream` (a hipStream_t handle as an integer, e.g.
        torch.cuda.Stream.cuda_stream; 0 = the default stream) from now on runs after everything enqueued so far
        that touches the accumulation buffer.  No wait on the host."""
        self._check(self._L.pt_accum_release(self._ctx, C.c_void_p(int(hip_stream))))

    def accum_acquire(self, hip_stream):
        """pt_accum_acquire: later work of this context on the accumulation buffer runs after what `hip_stream` holds
        now; path kernels of later renders start without waiting."""
        self._check(self._L.pt_accum_acquire(self._ctx, C.c_void_p(int(hip_stream))))

    def next_render_output(self, device_ptr, move=False):
        """pt_next_render_output: the next render call also writes its image to `device_ptr` (an integer device
        address, or None to withdraw); move=True leaves the accumulation buffer at zero instead."""
        self._check(self._L.pt_next_render_output(self._ctx, C.c_void_p(device_ptr), 1 if move else 0))

    def accum_device_ptr(self):
        return self._L.pt_accum_device_ptr(self._ctx)

    def primary_hits(self):
        ids = np.empty((self.height, self.width), dtype=np.int32)
        dist = np.empty((self.height, self.width), dtype=np.float32)
        self._check(self._L.pt_primary_hits(self._ctx, ids.ctypes.data_as(C.POINTER(C.c_int32)), _fp(dist)))
        return ids, dist

    def synchronize(self):
        self._check(self._L.pt_synchronize(self._ctx))

    def wait_in_flight(self, max_in_flight=1):
        """pt_wait_in_flight: returns when at most `max_in_flight` batches of frames are still running."""
        self._check(self._L.pt_wait_in_flight(self._ctx, int(max_in_flight)))

    def counters(self):
        c = PtCounters()
        self._check(self._L.pt_get_counters(self._ctx, C.byref(c)))
        return c.as_dict()

    def debug_stats(self):
        out = (C.c_uint64 * 16)()
        self._check(self._L.pt_get_debug_stats(self._ctx, out))
        names = ["wave_clocks", "trav_clocks", "trav_steps", "inner_lanes", "leaf_lanes", "leaf_runs",
                 "outer_iters", "shade_lanes", "longest_wave", "longest_tail", "tail_sum", "first_drained_ticks", "last_end_ticks"]
        return dict(zip(names, [int(x) for x in out]))

    def reset_counters(self):
        self._check(self._L.pt_reset_counters(self._ctx))

    def timing(self):
        t = PtTiming()
        self._check(self._L.pt_get_timing(self._ctx, C.byref(t)))
        return t.as_dict()

    def reset_timing(self):
        self._check(self._L.pt_reset_timing(self._ctx))
