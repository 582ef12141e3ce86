"""Multi-GPU rendering: one process per GPU, frames sharded across ranks.

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


class ShardedRenderer:
    """A Renderer whose accumulation buffer is a torch CUDA tensor so that
    torch.distributed (backend "nccl" = RCCL) can reduce it in place."""

    def __init__(self, renderer, rank, world_size, group=None):
        import torch

        self.r = renderer
        self.rank, self.world_size, self.group = rank, world_size, group
        dev = torch.device("cuda", torch.cuda.current_device())
        self.buffer = torch.zeros(renderer.height * renderer.width * 4, dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        renderer.bind_accum(self.buffer.data_ptr())

    def render(self, first_frame, n_frames):
        import torch
        import torch.distributed as dist

        self.buffer.zero_()
        torch.cuda.synchronize()

        def all_reduce():
            self.r.synchronize()                  # kernels run on the library's own stream
            if self.world_size > 1:
                dist.all_reduce(self.buffer, op=dist.ReduceOp.SUM, group=self.group)
                torch.cuda.synchronize()

        render_sharded(lambda f, c, s: self.r.render_frames_sum(f, c, s), all_reduce,
                       lambda s: self.r.scale_accum(s), first_frame, n_frames, self.rank, self.world_size)
        self.r.synchronize()

    def image(self):
        return self.buffer.view(self.r.height, self.r.width, 4)
