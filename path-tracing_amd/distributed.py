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

    def render(self, first_frame, n_frames):
        import time

        import torch

        if self.mode == "frames":
            if first_frame != 0:
                raise ValueError("frame sharding sums whole renders: first_frame must be 0 (use mode='pixels' to continue a render)")
            self.r.reset_accum()      # zeroes the bound buffer on the library's stream, ordered before the render
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

    def image(self):
        return self.buffer.view(self.r.height, self.r.width, 4)
