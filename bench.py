#!/usr/bin/env python3
"""bench.py --gpus N --steps K --warmup W

One step = one frame of the render loop (generate -> quantised RGB8 image resident on rank 0's GPU) of the
reference's scene.json at 1920x1080 with  -r -rd 4 -ss -sss 16  (reflection/refraction depth 4 + 16-sample
soft shadows): the configuration the headline metric is quoted on.  Scene, BVH and textures are uploaded before
the timed region (they are broadcast once per frame set); with N > 1 every rank renders its interleaved 32x32
tiles and the tile blocks are gathered to rank 0 over RCCL inside the timed region.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))

import numpy as np
import torch
import torch.distributed as dist

import cases
from raytracer_amd import api
from raytracer_amd import dist as rdist

WORKLOADS = {
    # the headline configuration: reflection/refraction depth 4 + 16-sample soft shadows (soft shadows are -ss, config.cpp:15)
    "config2": dict(scene="ref", res=(1920, 1080), flags=["-r", "-rd", "4", "-ss", "-sss", "16"]),
    "config1": dict(scene="ref", res=(1920, 1080), flags=[]),
    # large-scene configuration, synthetic part: 1M random spheres/cubes over a ground plane (seed 42), BVH on
    "config3": dict(scene="random", n=1_000_000, res=(1920, 1080), flags=[]),
    "config3u": dict(scene="random", n=1_000_000, res=(1920, 1080), flags=["-u"]),
    # large-scene configuration, the reference's own part: its 18-object Blender scene bvh_vs_unaccel_5, BVH on / off
    "bvh5": dict(scene="blend:bvh_vs_unaccel_5", res=(1920, 1080), flags=[]),
    "bvh5u": dict(scene="blend:bvh_vs_unaccel_5", res=(1920, 1080), flags=["-u"]),
    # the reference's motion-blur scene with the multi-sample flag set at 1080p (64 samples per pixel)
    "cube_mb": dict(scene="blend:cube_mb", res=(1920, 1080), flags=["-aa", "-aas", "4", "-dof", "-dofs", "4", "-m", "-mbs", "4", "-r", "-gr", "-grs", "4"]),
    "config4": dict(scene="ref", res=(3840, 2160), flags=["-aa", "-aas", "16", "-gr", "-grs", "16", "-dof", "-dofs", "16", "-m", "-mbs", "16"]),
}
HBM_PEAK_GBS = 8000.0


# Algorithmic bytes per unit of work: what a wavefront pipeline with explicit ray and hit
# records has to move per ray, independent of how this implementation lays the records out.
BYTES_PER_EXTEND_RAY = 120      # ray record 48 B written + read, hit record 12 B written + read
BYTES_PER_SHADOW_RAY = 72       # shadow-ray record 32 B written + read, result 4 B written + read
BYTES_PER_SHADED_HIT = 32 + 64 + 48   # hit-tree node 16 B written + read, one primitive record, one material


def copy_ceiling_GBps(dev) -> float:
    """Device-to-device copy rate of this GPU (bytes read + written per second): the practical HBM ceiling next to
    the nominal 8 TB/s."""
    n = 1 << 30
    a = torch.empty(n, dtype=torch.uint8, device=dev); b = torch.empty_like(a)
    a.zero_(); b.copy_(a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        b.copy_(a)
    e1.record(); torch.cuda.synchronize(dev)
    return 5 * 2 * n / 1e9 / (e0.elapsed_time(e1) / 1e3)


def roofline(st: api.Stats, workload: str, small_scene: bool) -> dict:
    """Per-kernel achieved rate from the per-stage HIP-event times of one profiled frame."""
    kernels = {
        "k_extend<true>": dict(units=st.primary_rays, bytes_per_unit=BYTES_PER_EXTEND_RAY, ms=st.ms_extend_primary, launches=st.launches_extend_primary),
        "k_extend<false>": dict(units=st.reflection_rays + st.refraction_rays, bytes_per_unit=BYTES_PER_EXTEND_RAY,
                                ms=st.ms_extend_secondary, launches=st.launches_extend_secondary),
        "k_shadow": dict(units=st.shadow_rays, bytes_per_unit=BYTES_PER_SHADOW_RAY, ms=st.ms_shadow, launches=st.launches_shadow),
        "k_shade": dict(units=st.nodes_shaded, bytes_per_unit=BYTES_PER_SHADED_HIT, ms=st.ms_shade, launches=st.launches_shade),
    }
    for k in kernels.values():
        k["avg_launch_ms"] = k["ms"] / max(1, k["launches"])
        k["bytes_per_launch"] = k["units"] * k["bytes_per_unit"] / max(1, k["launches"])
        k["GBps"] = (k["units"] * k["bytes_per_unit"] / 1e9) / (k["ms"] / 1e3) if k["ms"] > 0 else 0.0
        k["Grays_or_Ghits_per_s"] = (k["units"] / 1e9) / (k["ms"] / 1e3) if k["ms"] > 0 else 0.0
    dom = max(kernels, key=lambda n: kernels[n]["ms"])
    d = kernels[dom]
    traffic = None
    tf = ROOT / "profiles" / "traffic_latest.json"
    if tf.exists():                    # HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (tools/profile.sh)
        t = json.loads(tf.read_text())
        if t.get("workload") == workload and dom in t.get("kernels", {}):
            traffic = t["kernels"][dom]["hbm_bytes_per_launch"]
    return {"bound": "hbm", "achieved": d["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": d["GBps"] / HBM_PEAK_GBS,
            "traffic": traffic, "kernel": dom, "avg_launch_ms": d["avg_launch_ms"], "bytes_per_launch": d["bytes_per_launch"],
            "by_kernel": kernels,
            "note": ("scene.json has 3 primitives: BVH nodes and primitive records are served from LDS/L2, HBM carries only the "
                     "ray queues and the hit tree, and the kernels are bound by VALU work (IEEE divides, double-precision "
                     "normalisation kept for bit parity), so the HBM fraction is small by construction") if small_scene else
                    ("the algorithmic bytes count ray and hit records only; on this scene the time goes to the dependent "
                     "node/primitive fetches of the second-level hierarchy walk (L2/HBM latency), not to streaming")}


def _oracle_band(job):
    """Worker of the all-cores leg: the oracle port on one horizontal band of the sample film."""
    text, flags, tex, win = job
    import oracle
    _, st = oracle.render(text, flags, tex, window=win)
    return st.total_rays()


def port_all_cores(text: str, flags: list, w: int, h: int) -> dict:
    """The oracle port on every host core of this GPU's share: the film cut into bands, one forked worker per core
    (the port is scalar and keeps global state, so it scales by processes, not threads)."""
    import multiprocessing as mp
    cores = max(1, min(int(os.environ.get("OMP_NUM_THREADS", "0")) or (os.cpu_count() or 1), os.cpu_count() or 1, 16))
    bands = min(h, cores * 16)
    jobs = [(text, flags, str(cases.TEXTURES), (0, (h * b) // bands, w, (h * (b + 1)) // bands)) for b in range(bands)]
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(cores) as pool:
        rays = sum(pool.map(_oracle_band, jobs, chunksize=1))
    secs = time.perf_counter() - t0
    return {"port_all_cores_value": rays / max(secs, 1e-9), "port_all_cores_seconds": secs, "port_all_cores": cores}


def cpu_baseline(workload: dict) -> dict:
    """The reference's CPU path on a bounded sample of the same workload (same scene and flags, smaller film)."""
    sys.path.insert(0, str(ROOT / "oracle"))
    import oracle
    import ref_runner
    flags = workload["flags"]
    if workload["scene"] == "random":
        return {"unit": "rays/s", "value": None, "cores": 1, "kind": "port",
                "sample": "not run: the reference needs minutes per thousand pixels on this scene (every leaf of its depth-8 tree "
                          "holds thousands of primitives); its rate on its own scenes is in the default workload's line"}
    cfg, _ = api.parse_args(flags)
    spp = max(1, cfg.samples_per_pixel())
    scale = max(1, int(round(spp ** 0.5)))                       # keep the sample near 360 000 primary samples
    w, h = max(32, 800 // scale), max(18, 450 // scale)
    text = cases.reference_scene(w, h) if workload["scene"] == "ref" else cases.blend_scene(workload["scene"][6:], w, h)
    _, ost = oracle.render(text, flags, str(cases.TEXTURES))          # counts the rays the reference's code path traces
    rays = ost.total_rays()
    out = {"unit": "rays/s", "cores": 1,
           "sample": f"{'scene.json' if workload['scene'] == 'ref' else workload['scene'][6:]} at {w}x{h} ({w*h} of {workload['res'][0] * workload['res'][1]} pixels, same camera and flags), flags {' '.join(flags)}, {rays} rays "
                     f"(the reference re-traces its 16 identical reflection rays per hit; all are counted)"}
    if ref_runner.have_ref():
        tex = {p.name: p for p in cases.TEXTURES.glob("*.ppm")}
        _, secs, _ = ref_runner.run_ref(text, flags, tex)
        out.update(kind="reference", value=rays / max(secs, 1e-9), seconds=secs, port_seconds=ost.seconds,
                   port_value=rays / max(ost.seconds, 1e-9))
    else:
        out.update(kind="port", value=rays / max(ost.seconds, 1e-9), seconds=ost.seconds)
    try:
        out.update(port_all_cores(text, flags, w, h))
    except Exception as e:                      # a reported extra, never a reason to lose the line
        out["port_all_cores_error"] = repr(e)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)        # a frame takes a fraction of a millisecond: time enough of them
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--workload", default="config2", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--frames-in-flight", type=int, default=4,
                    help="contexts rendering consecutive frames concurrently on their own streams (1 = one frame at a time)")
    ap.add_argument("--backend", default=None, help="torch.distributed backend (default: nccl = RCCL)")
    ap.add_argument("--single-device", action="store_true",
                    help="rehearsal on a one-GPU machine: every rank uses GPU 0 (use with --backend gloo)")
    args = ap.parse_args()

    if args.single_device:
        os.environ["LOCAL_RANK"] = "0"
    rank, world, local = rdist.init_process_group(args.backend)
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run", file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py needs a GPU: the render path has no CPU fallback", file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    wl = WORKLOADS[args.workload]
    W, H = wl["res"]
    # Rank 0 parses the scene and builds the BVH on its host; flags and the flat scene/BVH arrays are broadcast once
    # per frame set (over RCCL with the nccl backend), every rank uploads its copy.
    on_dev = dev if (world > 1 and args.backend != "gloo") else None
    flags = json.loads(rdist.broadcast_bytes(json.dumps(wl["flags"]).encode() if rank == 0 else None, 0, on_dev).decode())
    cfg, _ = api.parse_args(flags)
    ls = None
    if rank == 0:
        if wl["scene"] == "ref":
            ls = api.LoadedScene(cases.reference_scene(W, H), cfg, str(cases.TEXTURES))
        elif wl["scene"].startswith("blend:"):
            ls = api.LoadedScene(cases.blend_scene(wl["scene"][6:], W, H), cfg, str(cases.TEXTURES))
        else:
            from raytracer_amd import scenegen as sg
            ls = api.ArrayScene(sg.random_prims(wl["n"], 42), (W, H))
    ls = rdist.broadcast_scene(ls, bool(cfg.bvh), 0, on_dev)
    # Frames in flight: consecutive frames are independent, so P contexts (each with its own work buffers and stream)
    # render them concurrently; a frame's small launches (deep recursion levels) then fill the other frames' gaps.
    # Four by default: one per hardware queue of the process (more share queues and lose again).
    # Not used for frames of many batches, which keep the GPU busy on their own.  With three or more frames in flight
    # the contexts do not fork their shadow launches onto second streams: the frames already overlap each other, and
    # a process has only four hardware queues for all of its streams.
    P = max(1, args.frames_in_flight) if cfg.samples_per_pixel() <= 64 else 1
    if world > 1:
        P = min(P, 3)                                 # the communicator's stream takes the fourth hardware queue

    def new_renderer():
        rr = api.Renderer(local, workspace_bytes=16 << 30)
        if P >= 3 or world > 1:
            rr.set_option("overlap", 0)
        for kv in filter(None, os.environ.get("RT_BENCH_OPTS", "").split(",")):      # development: name=value library options
            rr.set_option(kv.split("=")[0], int(kv.split("=")[1]))
        return rr

    r = new_renderer()
    r.upload(ls, with_bvh=bool(cfg.bvh))
    r.set_shard(rank, world)
    # Everything of a step runs on the context's own stream, which torch sees as an external stream (the RCCL gather
    # is ordered against it).  No further streams are created for rendering: a process has few hardware queues, and
    # two render streams that share one do not overlap.
    stream = torch.cuda.ExternalStream(r.stream_handle(), device=dev)
    torch.cuda.set_stream(stream)
    assert stream.cuda_stream != 0

    block = torch.empty(r.out_bytes(), dtype=torch.uint8, device=dev)
    image = torch.empty(H * W * 3, dtype=torch.uint8, device=dev) if rank == 0 else None

    ctxs = [(r, stream, block)]
    for _ in range(P - 1):
        r2 = new_renderer()
        r2.upload(ls, with_bvh=bool(cfg.bvh)); r2.set_shard(rank, world)
        s2 = torch.cuda.ExternalStream(r2.stream_handle(), device=dev)
        ctxs.append((r2, s2, torch.empty_like(block)))

    staged = args.backend == "gloo"      # rehearsal path: gloo moves host tensors

    # N > 1 over RCCL: frames are pipelined.  Frame k is rendered on the render stream into one of two blocks; its
    # gather runs on the communicator's stream and rank 0 re-orders it on a third stream, both beside the kernels of
    # frame k + 1.  Buffers are reused every other frame behind events; the timed region ends with a device-wide
    # synchronisation, so every frame's image has reached rank 0 inside it.
    pipelined = world > 1 and not staged
    # With several pipelines taking turns, rank 0 re-orders each frame on that frame's own render stream (no fifth stream)
    inline = P >= 2
    asm_stream = torch.cuda.Stream(dev) if (pipelined and rank == 0 and not inline) else None
    pipes = [rdist.FramePipeline(rr, cfg, rank, world, dev, ss, asm_stream=asm_stream, assemble_inline=inline) for rr, ss, _ in ctxs] if pipelined else None
    frame_no = [0]

    def step():
        k = frame_no[0] % P; frame_no[0] += 1
        if pipelined:
            pipes[k].step()
            return None
        if world == 1:
            rr, _, bb = ctxs[k]
            rr.render_device(cfg, bb.data_ptr())
            return None
        r.render_device(cfg, block.data_ptr())
        if world > 1:
            g = rdist.gather_blocks(block.cpu(), 0)
            gathered = g.to(dev) if rank == 0 else None
            if rank == 0:
                r.assemble(gathered.data_ptr(), world, image.data_ptr())
                keep.append(gathered)            # the assemble kernel is only enqueued: keep its input alive
                del keep[:-2]
        return None

    keep = []

    # Ray counts (identical every frame: the generator is counter based) and per-kernel device times.  The profiled
    # frame is enqueued behind four ordinary ones so that the GPU, not the host, paces it.
    st = r.render_device(cfg, block.data_ptr(), want_stats=True)     # ray counts (the counting kernel variants)
    r.set_option("skip_unlit", 0)                                    # how many shadow rays the N.L <= 0 rule leaves untraced
    shadow_all = r.render_device(cfg, block.data_ptr(), want_stats=True).shadow_rays
    r.set_option("skip_unlit", 1)
    for _ in range(4):
        r.render_device(cfg, block.data_ptr())
    r.set_option("count_rays", 0)                                    # stage times with the kernels the timed frames run
    tm = r.render_device(cfg, block.data_ptr(), want_stats=True)
    r.set_option("count_rays", 1)
    for name in ("ms_extend_primary", "ms_extend_secondary", "ms_shadow", "ms_shade", "ms_other", "ms_total",
                 "launches_extend_primary", "launches_extend_secondary", "launches_shadow", "launches_shade"):
        setattr(st, name, getattr(tm, name))
    rays_local = torch.tensor([st.primary_rays, st.shadow_rays, st.reflection_rays, st.refraction_rays,
                               st.shadow_rays_ref + st.reflection_rays_ref + st.refraction_rays_ref, st.nodes_shaded],
                              dtype=torch.float64, device=dev)
    if world > 1:
        rays_local = rays_local.cpu() if staged else rays_local
        dist.all_reduce(rays_local)
    tot = [int(x) for x in rays_local.tolist()]
    rays_total = tot[0] + tot[1] + tot[2] + tot[3]

    # latency of one frame alone (enqueue -> done), next to the throughput the timed loop measures
    torch.cuda.synchronize(dev)
    t_lat = time.perf_counter()
    for _ in range(10):
        r.render_device(cfg, block.data_ptr()); r.synchronize()
    frame_latency_ms = (time.perf_counter() - t_lat) * 100.0
    # every context records a frame graph on its second frame with the same output buffer; a pipeline alternates
    # between two blocks, so it needs four frames before both graphs exist
    warm_steps = max(args.warmup, (4 if pipelined else 2) * P)
    for _ in range(warm_steps):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    t_enq = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if staged else dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    if rank == 0:
        ms_per_step = 1000.0 * elapsed / args.steps
        # rank-0 sanity: the frame is not black and (N=1) is the row-major image itself
        last = (frame_no[0] - 1) % P
        final = ((pipes[last].image if pipelined else image) if world > 1 else ctxs[last][2]).cpu().numpy()
        lit = float((final.reshape(-1, 3).sum(axis=1) > 0).mean())
        line = {
            "metric": "rays/sec (all ray types), 1080p scene.json" if args.workload == "config2" else f"rays/sec (all ray types), {args.workload}",
            "value": rays_total / (elapsed / args.steps), "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": warm_steps,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic: the reference's own ASCII/scene.json (committed as a fixture), no dataset" if wl["scene"] == "ref"
                    else f"the reference's Blend/{wl['scene'][6:]}.blend exported to scene.json text (committed as a fixture)" if wl["scene"].startswith("blend:")
                    else f"synthetic: {wl['n']} random spheres/cubes (seed 42) over a ground plane, generated in-process",
            "config": {"workload": f"{args.workload}: {'scene.json' if wl['scene'] == 'ref' else wl['scene'][6:] if wl['scene'].startswith('blend:') else str(wl['n']) + ' random primitives'} {W}x{H}, flags {' '.join(wl['flags']) or '(none)'}",
                       "spp": cfg.samples_per_pixel(), "tiles": "32x32 interleaved, rank = (tx+ty) % N",
                       "frames_in_flight": P, "frame_ms_alone": frame_latency_ms, "host_enqueue_ms_per_step": 1000.0 * t_enq / args.steps,
                       "timing": "ms_per_step = timed wall clock / steps with frames_in_flight independent frames overlapping on the GPU; "
                                 "frame_ms_alone = one frame enqueued and waited for by itself",
                       "rays_per_frame": {"primary": tot[0], "shadow": tot[1], "reflection": tot[2], "refraction": tot[3],
                                          "hits_shaded": tot[5], "as_the_reference_would_trace_them": tot[0] + tot[4],
                                          "shadow_not_traced_because_the_surface_faces_away_from_the_light": (shadow_all - st.shadow_rays) if world == 1 else None},
                       "lit_fraction": lit},
            "roofline": roofline(st, args.workload, wl["scene"] != "random"),
        }
        ceiling = copy_ceiling_GBps(dev)
        line["roofline"]["measured_copy_GBps"] = ceiling
        line["roofline"]["frac_of_measured_copy"] = line["roofline"]["achieved"] / ceiling
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(wl)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    for rr, _, _ in ctxs:
        rr.close()
    ls.close()


if __name__ == "__main__":
    main()
