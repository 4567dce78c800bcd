"""One process per GPU: tile-block gather to rank 0 over torch.distributed (backend "nccl" is RCCL on ROCm; "gloo"
on CPU for tests).  The only exchange step of the render path: every rank renders its interleaved 32x32 tiles
into one contiguous tile-major block, rank 0 gathers the blocks and re-orders them into the row-major image."""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def env_rank_world() -> tuple[int, int, int]:
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def init_process_group(backend: str | None = None) -> tuple[int, int, int]:
    """Reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* from the environment (torchrun's contract)."""
    rank, world, local = env_rank_world()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kwargs = {}
        if backend == "nccl":
            torch.cuda.set_device(local)
            kwargs["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kwargs)
    return rank, world, local


def gather_blocks(block: torch.Tensor, dst: int = 0) -> torch.Tensor | None:
    """block: this rank's flat uint8 tile-major bytes (equal length on all ranks).  Returns on `dst` the
    rank-major concatenation [world * len(block)], elsewhere None.  world == 1: returns block itself."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return block
    world, rank = dist.get_world_size(), dist.get_rank()
    if rank == dst:
        out = torch.empty(world * block.numel(), dtype=block.dtype, device=block.device)
        dist.gather(block, list(out.chunk(world)), dst=dst)
        return out
    dist.gather(block, None, dst=dst)
    return None


def gather_blocks_async(block: torch.Tensor, out: torch.Tensor | None, dst: int = 0):
    """Starts the gather of `block` into the preallocated rank-major `out` (only on `dst`; None elsewhere) and
    returns the work handle: `handle.wait()` makes the CURRENT stream wait for the transfer (it does not block the
    host), so a frame's gather can run beside the next frame's kernels."""
    world = dist.get_world_size()
    return dist.gather(block, list(out.chunk(world)) if dist.get_rank() == dst else None, dst=dst, async_op=True)


def broadcast_bytes(data: bytes | None, src: int = 0, device: torch.device | None = None) -> bytes:
    """Scene text / flags travel from rank 0 once per frame set (the scene and BVH broadcast of the design)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        assert data is not None
        return data
    device = device or torch.device("cpu")
    n = torch.tensor([len(data) if data is not None else 0], dtype=torch.int64, device=device)
    dist.broadcast(n, src)
    if dist.get_rank() == src:
        buf = torch.frombuffer(bytearray(data), dtype=torch.uint8).to(device)
    else:
        buf = torch.empty(int(n.item()), dtype=torch.uint8, device=device)
    dist.broadcast(buf, src)
    return bytes(buf.cpu().numpy().tobytes())


def broadcast_array(arr, src: int = 0, device: torch.device | None = None):
    """Broadcast one uint8 numpy array from `src` (other ranks pass None); returns the array on every rank.
    With the nccl backend the bytes travel GPU to GPU over xGMI (`device` = this rank's GPU)."""
    import numpy as np
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return arr
    device = device or torch.device("cpu")
    n = torch.tensor([arr.nbytes if dist.get_rank() == src else 0], dtype=torch.int64, device=device)
    dist.broadcast(n, src)
    if dist.get_rank() == src:
        buf = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.uint8)).to(device)
    else:
        buf = torch.empty(int(n.item()), dtype=torch.uint8, device=device)
    dist.broadcast(buf, src)
    return arr if dist.get_rank() == src else buf.cpu().numpy()


def broadcast_scene(loaded, with_bvh: bool, src: int = 0, device: torch.device | None = None):
    """The scene broadcast of the multi-GPU design: rank `src` has parsed the scene and built the BVH on its host
    (`loaded`, an api.LoadedScene; other ranks pass None); every other rank receives the flat arrays once and
    rebuilds the C-ABI view from them.  Returns an object usable with api.Renderer.upload on every rank."""
    from . import api
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return loaded
    blob = api.pack_scene(loaded, with_bvh) if dist.get_rank() == src else None
    blob = broadcast_array(blob, src, device)
    return loaded if dist.get_rank() == src else api.PackedScene(blob)


class _DeviceBytes:
    """A (device pointer, bytes) pair as something torch.as_tensor can wrap without copying."""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def broadcast_device_scene(renderer, src: int = 0, device: torch.device | None = None):
    """The scene broadcast of the multi-GPU design, device to device: rank `src` has uploaded the scene and the tree into
    `renderer` (parse, host BVH build, upload and on-GPU construction happen there once); every other rank's `renderer`
    adopts the layout - it allocates, nothing else - and receives every device array straight into place with one
    torch.distributed.broadcast each (RCCL over xGMI with the nccl backend).  No rank but `src` touches host scene data."""
    import numpy as np
    if not dist.is_initialized():
        return
    rank = dist.get_rank()
    layout = broadcast_bytes(renderer.export_layout() if rank == src else None, src, device)
    if rank != src:
        renderer.adopt_layout(layout)
    for ptr, nbytes in renderer.scene_arrays():
        if nbytes:
            t = torch.as_tensor(_DeviceBytes(ptr, nbytes), device=device)
            dist.broadcast(t, src)
    torch.cuda.synchronize(device)


class FramePipeline:
    """Frames of a multi-rank run, pipelined: frame k is rendered on `render_stream` into one of two blocks; its gather
    runs beside the kernels of frame k + 1 (on the communicator's stream) and rank 0 re-orders it into the row-major
    image on a third stream.  Buffers are reused every other frame behind the gather handle and an event.
    `render_stream` must be the stream the renderer enqueues on (torch.cuda.ExternalStream(renderer.stream_handle()), or
    a torch stream handed to renderer.set_stream beforehand).  `start_gather(block, out_or_None)` returns a handle whose wait() makes the current stream wait for the transfer
    (gather_blocks_async over RCCL; tests substitute device copies).
    assemble_inline: rank 0 re-orders on `render_stream` itself, behind the gather, instead of on a third stream.  That
    stalls nothing when several pipelines take turns (the stream's next frame is a few frames away) and saves one of
    the process's few hardware queues."""

    def __init__(self, renderer, cfg, rank: int, world: int, device, render_stream, start_gather=gather_blocks_async, asm_stream=None,
                 assemble_inline: bool = False):
        self.r, self.cfg, self.rank, self.world = renderer, cfg, rank, world
        self.stream = render_stream
        self.start_gather = start_gather
        n = renderer.out_bytes()
        self.blocks = [torch.empty(n, dtype=torch.uint8, device=device) for _ in range(2)]
        self.gathered = [torch.empty(world * n, dtype=torch.uint8, device=device) if rank == 0 else None for _ in range(2)]
        self.image = torch.empty(renderer.width * renderer.height * 3, dtype=torch.uint8, device=device) if rank == 0 else None
        # several pipelines of one process can share the re-ordering stream (every stream takes one of few hardware queues)
        self.inline = assemble_inline
        self.asm_stream = (asm_stream or torch.cuda.Stream(device)) if (rank == 0 and not assemble_inline) else None
        self.pending = [None, None]
        self.asm_done = [None, None]
        self.frame = 0

    def step(self):
        b = self.frame % 2
        self.frame += 1
        with torch.cuda.stream(self.stream):
            if self.pending[b] is not None:
                self.pending[b].wait()                          # the gather that last read blocks[b], two frames ago
            if self.asm_done[b] is not None:
                self.stream.wait_event(self.asm_done[b])        # the re-ordering that last read gathered[b]
            self.r.render_device(self.cfg, self.blocks[b].data_ptr())             # on render_stream, the context's stream
            self.pending[b] = self.start_gather(self.blocks[b], self.gathered[b])
            if self.rank == 0 and self.inline:
                self.pending[b].wait()                          # stream order also protects gathered[b] and the image
                self.r.assemble(self.gathered[b].data_ptr(), self.world, self.image.data_ptr())
        if self.rank == 0 and not self.inline:
            with torch.cuda.stream(self.asm_stream):
                self.pending[b].wait()
                self.r.set_stream(self.asm_stream.cuda_stream)
                self.r.assemble(self.gathered[b].data_ptr(), self.world, self.image.data_ptr())
                self.r.set_stream(self.stream.cuda_stream)
                ev = torch.cuda.Event(); ev.record(self.asm_stream)
                self.asm_done[b] = ev
