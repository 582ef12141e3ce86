"""Multi-GPU rendering: one process per GPU, the image or the frames sharded across ranks.

Two ways to split a render, both with the scene replicated on every GPU and ONE collective per
render (RCCL over xGMI when the process group's backend is "nccl"):

pixels (default)  rank r renders the 8x8-pixel tiles with (tile_x + tile_y) % N == r for ALL
                  frames, keeping the reference's running mean per pixel (pt_set_pixel_shard +
                  pt_render_frames); the other pixels of its zero-initialised buffer stay 0, so a
                  sum all-reduce assembles the image.  x + 0 is exact: the result equals the
                  single-GPU image bit for bit.  The primary-ray pass is sharded too.
frames            described below; kept for renders whose frames outnumber their pixels.

Frame sharding:

The reference is single-GPU.  Its unit of accumulation is the frame
(src/main.rs:448-452,510; shader.wgsl:104-106), and every (pixel, frame, sample)
is independent given the read-only scene, so the scene is replicated on every
GPU and rank r of N renders frames f with f % N == r into a SUM buffer
(pt_render_frames_sum).  One float32 sum all-reduce of the W*H*4 buffer per
render (RCCL over xGMI when the process group's backend is "nccl") followed by a
multiplication by 1/n_frames gives the running mean of shader.wgsl:104-106 up to
f32 rounding.  There is no per-frame communication.
"""


def shard_frames(first_frame, n_frames, rank, world_size):
    """Frames rank `rank` renders: returns (first, count, stride)."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad rank/world_size")
    count = (n_frames - rank + world_size - 1) // world_size if n_frames > rank else 0
    return first_frame + rank, count, world_size


def render_sharded(render_sum, all_reduce, scale, first_frame, n_frames, rank, world_size):
    """Generic driver, independent of where the buffer lives.

    render_sum(first, count, stride): adds the per-frame means of the rank's frames to its buffer
    all_reduce():                     sums the buffers of all ranks in place
    scale(s):                         multiplies the buffer by s
    """
    first, count, stride = shard_frames(first_frame, n_frames, rank, world_size)
    if count:
        render_sum(first, count, stride)
    all_reduce()
    scale(1.0 / float(n_frames))


TILE = 8      # pixel sharding works on the primary kernel's 8x8 wave tiles


def tile_owner(x, y, world_size):
    """Rank that renders pixel (x, y) under pixel sharding (same rule as pt_set_pixel_shard)."""
    return ((x // TILE) + (y // TILE)) % world_size


def owner_mask(width, height, rank, world_size):
    """(height, width) boolean array: True where `rank` owns the pixel."""
    import numpy as np
    tx = np.arange(width) // TILE
    ty = np.arange(height) // TILE
    return (tx[None, :] + ty[:, None]) % world_size == rank


def render_pixel_sharded(render_owned, all_reduce, first_frame, n_frames):
    """Generic driver of pixel sharding, independent of where the buffer lives.

    render_owned(first, count): running mean of the frames over the rank's own pixels, other
                                pixels of the (zeroed) buffer untouched
    all_reduce():               sums the buffers of all ranks in place
    """
    render_owned(first_frame, n_frames)
    all_reduce()


class StepsInFlight:
    """Generic driver of renders whose collectives are still running when the next render is enqueued, independent of
    where the buffers live: the image of call k goes to staging buffer k mod 2, its collective is started and NOT waited
    for, and is settled (waited for, then scaled) right before that buffer is written again two calls later, or by
    finish().  Every rank makes the same calls in the same order, so the collectives match up.

    render_into(j):       enqueues the render whose image ends in staging buffer j
    start_collective(j):  starts the sum of buffer j over the ranks; returns a handle for wait_collective, or None
    wait_collective(h):   waits for that collective (on a stream or on the host, as the backend has it)
    scale(j, s):          multiplies buffer j by s, ordered after the collective"""

    def __init__(self, render_into, start_collective, wait_collective, scale):
        self.render_into, self.start_collective = render_into, start_collective
        self.wait_collective, self.scale = wait_collective, scale
        self.pending = [None, None]         # per staging buffer: (handle or None, scale factor or None)
        self.calls = 0
        self.last = None                    # staging buffer of the last call

    def settle(self, j):
        if self.pending[j] is None:
            return
        handle, factor = self.pending[j]
        if handle is not None:
            self.wait_collective(handle)
        if factor is not None:
            self.scale(j, factor)
        self.pending[j] = None

    def submit(self, factor=None):
        """One render and its collective; returns the staging buffer it uses."""
        j = self.calls & 1
        self.calls += 1
        self.settle(j)                      # the collective that used this buffer two calls ago
        self.render_into(j)
        self.pending[j] = (self.start_collective(j), factor)
        self.last = j
        return j

    def finish(self):
        """Settles everything; returns the staging buffer that holds the last call's image (None: no call yet)."""
        if self.last is None:
            return None
        for j in (self.last ^ 1, self.last):        # the older one first: the same order on every rank
            self.settle(j)
        return self.last


def combine(buffer, world_size, group=None, collective="all_reduce", force_collective=False):
    """Sums `buffer` (a torch tensor, CPU or GPU) over the `world_size` ranks that share a render.

    A renderer of ONE rank issues no collective, whatever process groups the job has: its image is complete, and a
    collective on the job's default group would add unrelated images (or hang, if the other ranks are not rendering).
    `force_collective` runs the collective on a one-rank group all the same (a rehearsal of the RCCL calls on one GPU).
    The group the collective runs on must have exactly `world_size` ranks."""
    import torch.distributed as dist

    if world_size == 1 and not force_collective:
        return False
    if not dist.is_initialized():
        raise RuntimeError("a render sharded over %d ranks needs an initialised torch.distributed process group" % world_size)
    n = dist.get_world_size(group)
    if n != world_size:
        raise ValueError("the render is sharded over %d ranks but the process group has %d" % (world_size, n))
    if collective == "reduce":
        dist.reduce(buffer, dst=dist.get_global_rank(group, 0) if group is not None else 0, op=dist.ReduceOp.SUM, group=group)
    else:
        dist.all_reduce(buffer, op=dist.ReduceOp.SUM, group=group)
    return True


class ShardedRenderer:
    """A Renderer whose accumulation buffer is a torch CUDA tensor so that
    torch.distributed (backend "nccl" = RCCL) can reduce it.

    pixels mode keeps TWO tensors: `owned`, bound to the library, holds the running mean of this
    rank's own tiles and zeros elsewhere, and lives across calls, so a progressive render continues
    (render(0, 4) then render(4, 4) == render(0, 8), like Renderer.render); `buffer` receives a copy
    per call and is what the collective sums.  frames mode sums frame means and divides by their
    number, which is the running mean only when the render starts at frame 0."""

    def __init__(self, renderer, rank, world_size, group=None, mode="pixels", collective="all_reduce", force_collective=False):
        """collective: "all_reduce" leaves the image on every rank, "reduce" on rank 0 of the group only (what a
        host that displays or stores the image needs; half the traffic per link on a ring).  `group` must have
        exactly `world_size` ranks.  With world_size 1 no collective runs unless `force_collective` asks for it."""
        import torch

        if mode not in ("pixels", "frames"):
            raise ValueError("mode must be 'pixels' or 'frames'")
        if collective not in ("all_reduce", "reduce"):
            raise ValueError("collective must be 'all_reduce' or 'reduce'")
        self.collective = collective
        self.force_collective = bool(force_collective)
        self.r = renderer
        self.rank, self.world_size, self.group, self.mode = rank, world_size, group, mode
        renderer.set_pixel_shard(*((rank, world_size) if mode == "pixels" else (0, 1)))
        dev = torch.device("cuda", torch.cuda.current_device())
        n = renderer.height * renderer.width * 4
        self.buffer = torch.zeros(n, dtype=torch.float32, device=dev)
        self.owned = torch.zeros(n, dtype=torch.float32, device=dev) if mode == "pixels" else self.buffer
        torch.cuda.synchronize()
        renderer.bind_accum(self.owned.data_ptr())
        self.last_timing = {}           # seconds of the last render(): kernels, collective
        self._side = None               # render_async(): side stream, staging tensors and collectives in flight (_flight)
        self._steps = None
        self._result = None             # ... and the tensor that holds the last result after finish()

    def render(self, first_frame, n_frames):
        import time

        import torch

        self.finish()                 # renders of render_async() still in flight come first
        self._result = None
        if self.mode == "frames":
            if first_frame != 0:
                raise ValueError("frame sharding sums whole renders: first_frame must be 0 (use mode='pixels' to continue a render)")
            self.r.reset_accum()      # zeroes the bound buffer on the library's stream, ordered before the render
            self._bound_is_clean = False      # it will hold this render's image
        t0 = time.perf_counter()

        def all_reduce():
            self.r.synchronize()                  # kernels run on the library's own stream
            t1 = time.perf_counter()
            if self.mode == "pixels":
                self.buffer.copy_(self.owned)     # the collective must not touch the running means kept in `owned`
            combine(self.buffer, self.world_size, self.group, self.collective, self.force_collective)
            torch.cuda.synchronize()
            self.last_timing = {"render_s": t1 - t0, "collective_s": time.perf_counter() - t1}

        if self.mode == "pixels":
            render_pixel_sharded(lambda f, c: self.r.render_frames(f, c), all_reduce, first_frame, n_frames)
        else:
            render_sharded(lambda f, c, s: self.r.render_frames_sum(f, c, s), all_reduce,
                           lambda s: self.r.scale_accum(s), first_frame, n_frames, self.rank, self.world_size)
            self.r.synchronize()

    # ---- renders in flight: no wait on the host between a render, its collective and the next render ----
    #
    # render() above ends every step with a wait: kernels, then the collective, then the host goes on.  At N GPUs a
    # rank's share of a render is short, the persistent path kernel's drain (about a millisecond in which most of the
    # GPU idles) is a large part of it, and the collective adds to it.  render_async() only enqueues.  The render's
    # last frame fold writes the image to one of two staging tensors as well (pt_next_render_output; in frames mode
    # it MOVES it there, so the next sum starts from zeros): no copy kernel, no clearing.  A side stream waits for
    # that fold (pt_accum_release) and starts the collective on the staging tensor as an asynchronous operation; the
    # fold that will write this staging tensor again, two calls later, waits for the side stream (pt_accum_acquire),
    # which has by then waited for the collective.  The next render's path kernel, which touches neither buffer,
    # starts while this one drains and while the collective runs.  Every rank issues the same collectives in the same
    # order.  Results equal render()'s bit for bit (the same sum over ranks, then the same scale).
    # (A copy of the bound buffer by a kernel or memcpy on the side stream was measured first: a third kernel that must
    # find room beside two persistent grids, on the critical path between two frame folds, cost 0.3 to 1.2 ms per
    # step, more than the steps in flight gain.)

    def _flight(self):
        import torch
        import torch.distributed as dist

        if self._side is not None:
            return self._steps
        self._side = side = torch.cuda.Stream()
        self._staging = [torch.zeros_like(self.buffer), torch.zeros_like(self.buffer)]
        self._bound_is_clean = False        # frames mode: the bound buffer holds zeros
        torch.cuda.synchronize()            # the zero fills ran on torch's stream; the library writes from its own
        handle = side.cuda_stream

        def render_into(j):
            first_frame, n_frames = self._call
            self.r.accum_acquire(handle)                  # the collective that used this tensor must be over before the last fold writes it
            rendered = True
            if self.mode == "frames":
                if not self._bound_is_clean:
                    self.r.reset_accum()
                    self._bound_is_clean = True
                first, count, stride = shard_frames(first_frame, n_frames, self.rank, self.world_size)
                if count:
                    self.r.next_render_output(self._staging[j].data_ptr(), move=True)
                    self.r.render_frames_sum(first, count, stride)
                rendered = count > 0
            else:
                self.r.next_render_output(self._staging[j].data_ptr(), move=False)
                self.r.render_frames(first_frame, n_frames)
            self.r.accum_release(handle)                  # the side stream waits for the frame folds of this render
            if not rendered:
                with torch.cuda.stream(side):
                    self._staging[j].zero_()              # more ranks than frames: this rank adds nothing

        def start_collective(j):
            if not (self.world_size > 1 or self.force_collective):
                return None
            if not dist.is_initialized():
                raise RuntimeError("a render sharded over %d ranks needs an initialised torch.distributed process group" % self.world_size)
            if dist.get_world_size(self.group) != self.world_size:
                raise ValueError("the render is sharded over %d ranks but the process group has %d"
                                 % (self.world_size, dist.get_world_size(self.group)))
            # RCCL runs the collective on the GPU, ordered by streams: started here, waited for (by a stream) when it is
            # settled.  A backend that works on the host (gloo: the rehearsal of this path with two ranks on one GPU)
            # copies the tensor on streams of its own from a thread of its own and has its caller wait on the HOST; beside
            # renders that the main thread keeps enqueuing that hung both ranks in work.wait() one run in five, and
            # stalled for half a minute in others.  Such a backend gets the finished image and is waited for at once,
            # as in render().
            on_gpu = dist.get_backend(self.group) == "nccl"
            with torch.cuda.stream(side):
                if not on_gpu:
                    side.synchronize()
                dst = dist.get_global_rank(self.group, 0) if self.group is not None else 0
                if self.collective == "reduce":
                    work = dist.reduce(self._staging[j], dst=dst, op=dist.ReduceOp.SUM, group=self.group, async_op=on_gpu)
                else:
                    work = dist.all_reduce(self._staging[j], op=dist.ReduceOp.SUM, group=self.group, async_op=on_gpu)
            return work if on_gpu else None

        def wait_collective(work):
            with torch.cuda.stream(side):
                work.wait()                               # RCCL: the side stream waits, not the host

        def scale(j, factor):
            with torch.cuda.stream(side):
                self._staging[j].mul_(factor)

        self._steps = StepsInFlight(render_into, start_collective, wait_collective, scale)
        return self._steps

    def render_async(self, first_frame, n_frames):
        """render() without a wait on the host: enqueues the render and the collective, and returns.  finish() waits
        for everything; image() then holds the result of the LAST call."""
        if n_frames < 1:
            raise ValueError("a render has at least one frame")
        if self.mode == "frames" and first_frame != 0:
            raise ValueError("frame sharding sums whole renders: first_frame must be 0 (use mode='pixels' to continue a render)")
        steps = self._flight()
        self._call = (first_frame, n_frames)
        steps.submit(1.0 / float(n_frames) if self.mode == "frames" else None)

    def finish(self):
        """Waits for every render_async() so far; afterwards image() is the result of the last one."""
        if self._side is None:
            return
        j = self._steps.finish()
        if j is None:
            return
        self._side.synchronize()
        self.r.synchronize()
        self._result = self._staging[j]

    def image(self):
        """The image of the last render() or, after finish(), of the last render_async() (on rank 0 of the group only
        when the collective is "reduce")."""
        t = self._result if self._result is not None else self.buffer
        return t.view(self.r.height, self.r.width, 4)
