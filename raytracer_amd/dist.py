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
