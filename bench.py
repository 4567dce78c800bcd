#!/usr/bin/env python3
"""bench.py --gpus N --steps K --warmup W

One step = one frame of the render loop (generate -> quantised RGB8 image resident on rank 0's GPU) of the
reference's scene.json at 1920x1080 with  -r -rd 4 -ss -sss 16  (reflection/refraction depth 4 + 16-sample
soft shadows): the configuration the headline metric is quoted on.  Scene, BVH and textures are uploaded before
the timed region (they are broadcast once per frame set); with N > 1 every rank renders its interleaved 32x32
tiles and the tile blocks are gathered to rank 0 over RCCL inside the timed region.

`python bench.py --gpus N` with N > 1 and no torchrun environment starts the N ranks itself (as a child
`python -m torch.distributed.run`, before this process has touched a GPU) and relays rank 0's line.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))

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
    # the 4K multi-sample configuration at 64 instead of 4096 samples per pixel: the frame the multi-GPU split is meant for
    "config4s": dict(scene="ref", res=(3840, 2160), flags=["-aa", "-aas", "4", "-gr", "-grs", "16", "-dof", "-dofs", "4", "-m", "-mbs", "4"]),
    # the all-effects configuration, synthetic part: 10M random primitives (seed 43), 4K, every effect on at a reduced sample count
    "config5s": dict(scene="random", n=10_000_000, seed=43, res=(3840, 2160),
                     flags=["-aa", "-aas", "2", "-dof", "-dofs", "2", "-m", "-mbs", "2", "-t", "-ss", "-sss", "4", "-r", "-gr", "-grs", "2"]),
}
HBM_PEAK_GBS = 8000.0
# FP32 vector peak of the chip: 1024 SIMDs x one 64-lane instruction per 2 cycles x 2.4 GHz x 2 flop (an FMA) per lane
VALU_PEAK_TFLOPS = 157.3
FLOP_PER_VALU_INST = 128.0

# Algorithmic bytes per unit of work: what a wavefront pipeline with explicit ray and hit
# records has to move per ray, independent of how this implementation lays the records out.
BYTES_PER_EXTEND_RAY = 120      # ray record 48 B written + read, hit record 12 B written + read
BYTES_PER_SHADOW_RAY = 72       # shadow-ray record 32 B written + read, result 4 B written + read
BYTES_PER_SHADED_HIT = 32 + 64 + 48   # hit-tree node 16 B written + read, one primitive record, one material


def describe(name: str) -> str:
    wl = WORKLOADS[name]
    what = "scene.json" if wl["scene"] == "ref" else wl["scene"][6:] if wl["scene"].startswith("blend:") else f"{wl['n']} random primitives"
    return f"{name}: {what} {wl['res'][0]}x{wl['res'][1]}, flags {' '.join(wl['flags']) or '(none)'}"


# ---- CPU baseline (runs before this process makes any GPU call) -----------------------------------------------------

def _oracle_band(job):
    """Worker of the all-cores leg: the oracle port on one horizontal band of the sample film."""
    text, flags, tex, win = job
    import oracle
    _, st = oracle.render(text, flags, tex, window=win)
    return st.total_rays()


def port_all_cores(text: str, flags: list, w: int, h: int, textures: str) -> dict:
    """The oracle port on every host core of this GPU's share: the film cut into bands, one forked worker per core
    (the port is scalar and keeps global state, so it scales by processes, not threads)."""
    import multiprocessing as mp
    cores = max(1, min(int(os.environ.get("OMP_NUM_THREADS", "0")) or (os.cpu_count() or 1), os.cpu_count() or 1, 16))
    bands = min(h, cores * 16)
    jobs = [(text, flags, textures, (0, (h * b) // bands, w, (h * (b + 1)) // bands)) for b in range(bands)]
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(cores) as pool:
        rays = sum(pool.map_async(_oracle_band, jobs, chunksize=1).get(timeout=240))
    secs = time.perf_counter() - t0
    return {"port_all_cores_value": rays / max(secs, 1e-9), "port_all_cores_seconds": secs, "port_all_cores": cores}


def cpu_baseline(workload: dict) -> dict:
    """The reference's CPU path on a bounded sample of the same workload (same scene and flags, smaller film).
    Must be called before torch.cuda / HIP is initialised in this process: it forks workers and runs a child."""
    sys.path.insert(0, str(ROOT / "oracle"))
    import cases
    import oracle
    import ref_runner
    from raytracer_amd import api
    flags = workload["flags"]
    if workload["scene"] == "random":
        return {"unit": "rays/s", "value": None, "cores": 1, "kind": "port",
                "sample": "not run: the reference needs minutes per thousand pixels on this scene (every leaf of its depth-8 tree "
                          "holds thousands of primitives); its rate on its own scenes is in the default workload's line"}
    textures = str(cases.TEXTURES)

    def scene_at(w, h):
        return cases.reference_scene(w, h) if workload["scene"] == "ref" else cases.blend_scene(workload["scene"][6:], w, h)

    # Bounded sample: a pilot render of the port on a 96x54 film prices a pixel of this flag set (the reference re-traces
    # its gr_samples identical reflection rays at every level, so deep recursion and high sample counts cost far more
    # per pixel); the sample film is then sized for about 20 s of one core, at most 800x450.
    _, pilot = oracle.render(scene_at(96, 54), flags, textures)
    per_pixel = max(pilot.seconds, 1e-4) / (96 * 54)
    # (the compiled reference is about five times slower per ray than the port: it allocates per vector operation)
    budget_s = 4.0 if ref_runner.have_ref() else 15.0
    pixels = min(800 * 450, max(48 * 27, int(budget_s / per_pixel)))
    h = max(27, int((pixels * 9 / 16) ** 0.5)); w = h * 16 // 9
    text = scene_at(w, h)
    sample_flags = list(flags)
    _, ost = oracle.render(text, sample_flags, textures)              # counts the rays the reference's code path traces
    rays = ost.total_rays()
    out = {"unit": "rays/s", "cores": 1,
           "sample": f"{'scene.json' if workload['scene'] == 'ref' else workload['scene'][6:]} at {w}x{h} ({w*h} of "
                     f"{workload['res'][0] * workload['res'][1]} pixels, same camera and flags: {' '.join(sample_flags) or '(none)'}), "
                     f"{rays} rays, all counted as the reference traces them (it re-traces its gr_samples identical reflection rays per hit and level)"}
    if ref_runner.have_ref():
        tex = {p.name: p for p in cases.TEXTURES.glob("*.ppm")}
        _, secs, _ = ref_runner.run_ref(text, sample_flags, tex)
        out.update(kind="reference", value=rays / max(secs, 1e-9), seconds=secs, port_seconds=ost.seconds,
                   port_value=rays / max(ost.seconds, 1e-9))
    else:
        out.update(kind="port", value=rays / max(ost.seconds, 1e-9), seconds=ost.seconds)
    # the same sample without the BVH (-u, raytracer.cpp:202-256), the reference's other code path
    if "-u" not in sample_flags:
        try:
            u_flags = sample_flags + ["-u"]
            _, ust = oracle.render(text, u_flags, textures)
            un = {"flags": " ".join(u_flags), "rays": ust.total_rays(), "cores": 1}
            if ref_runner.have_ref():
                _, usecs, _ = ref_runner.run_ref(text, u_flags, {p.name: p for p in cases.TEXTURES.glob("*.ppm")})
                un.update(kind="reference", value=ust.total_rays() / max(usecs, 1e-9), seconds=usecs)
            else:
                un.update(kind="port", value=ust.total_rays() / max(ust.seconds, 1e-9), seconds=ust.seconds)
            out["unaccelerated"] = un
        except Exception as e:
            out["unaccelerated"] = {"error": repr(e)}
    try:
        out.update(port_all_cores(text, sample_flags, w, h, textures))
    except Exception as e:                      # a reported extra, never a reason to lose the line
        out["port_all_cores_error"] = repr(e)
    return out


# ---- launch ---------------------------------------------------------------------------------------------------------

def self_launch(args) -> int:
    """N > 1 without a torchrun environment: start the ranks as a child (this process has not touched a GPU and does
    not exec: it waits and passes the child's exit status on)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    env = dict(os.environ); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0"); env.setdefault("OMP_NUM_THREADS", "4")
    # stdout carries the result line only: whatever else the ranks or the communication library print there goes to stderr
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, bufsize=1)
    for line in child.stdout:
        is_result = False
        if line.lstrip().startswith("{"):
            try:
                json.loads(line); is_result = True
            except ValueError:
                pass
        (sys.stdout if is_result else sys.stderr).write(line)
        (sys.stdout if is_result else sys.stderr).flush()
    return child.wait()


KERNEL_SOURCES = ["rt_kernels.hip", "rt_context.hip", "rt_build.hip", "rt_types.h", "rt_build.h", "ref_libm.h", "Makefile"]


def kernel_sources_sha() -> str:
    """Identifies the kernel build a set of hardware counters belongs to: SHA-256 over the device sources and the build
    flags (tools/profile.sh stamps it into the counters it collects; the bench line compares it with the tree it runs from)."""
    import hashlib
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        h.update(name.encode()); h.update((ROOT / "raytracer_amd" / "csrc" / name).read_bytes())
    return h.hexdigest()[:16]


def load_counters(workload: str) -> tuple[dict, dict]:
    """Per-kernel, per-launch hardware counters of this workload collected by tools/profile.sh (rocprofv3 --pmc passes),
    committed as profiles/counters_latest.json, and where they come from: the stamp of the kernel sources they were
    collected on, and whether that is the tree this process runs from."""
    f = ROOT / "profiles" / "counters_latest.json"
    if not f.exists():
        return {}, {"file": None}
    rec = json.loads(f.read_text()).get(workload, {})
    stamp = rec.get("kernel_sources_sha")
    here = kernel_sources_sha()
    origin = {"file": "profiles/counters_latest.json", "kernel_sources_sha": stamp, "this_tree": here, "collected": rec.get("collected"),
              "stale": bool(rec) and stamp != here}
    return rec.get("kernels", {}), origin


def roofline(st, workload: str, small_scene: bool) -> dict:
    """The dominant kernel (by device time in a profiled frame) against the limit that bounds it.  Launch durations
    are measured live with HIP events on the render stream; instruction counts and HBM/L2 traffic per launch come from
    the committed rocprofv3 counter passes of the same workload."""
    kernels = {
        "k_extend<true>": dict(units=st.primary_rays, bytes_per_unit=BYTES_PER_EXTEND_RAY, ms=st.ms_extend_primary, launches=st.launches_extend_primary),
        "k_extend<false>": dict(units=st.reflection_rays + st.refraction_rays, bytes_per_unit=BYTES_PER_EXTEND_RAY,
                                ms=st.ms_extend_secondary, launches=st.launches_extend_secondary),
        "k_shadow": dict(units=st.shadow_rays, bytes_per_unit=BYTES_PER_SHADOW_RAY, ms=st.ms_shadow, launches=st.launches_shadow),
        "k_shade": dict(units=st.nodes_shaded, bytes_per_unit=BYTES_PER_SHADED_HIT, ms=st.ms_shade, launches=st.launches_shade),
    }
    if st.launches_shadow == 0 and st.launches_shade == 0 and st.launches_extend_primary > 0:
        # frames without recursion and with hard shadows run as one kernel: primary rays, their shadow rays and the shading
        units = st.primary_rays + st.shadow_rays
        total = st.primary_rays * BYTES_PER_EXTEND_RAY + st.shadow_rays * BYTES_PER_SHADOW_RAY + st.nodes_shaded * BYTES_PER_SHADED_HIT
        kernels = {"k_primary_shaded": dict(units=units, bytes_per_unit=total / max(1, units), ms=st.ms_extend_primary, launches=st.launches_extend_primary)}
    ctr, ctr_origin = load_counters(workload)
    if ctr_origin.get("stale"):
        ctr = {}                 # counters of another kernel build say nothing about this one: only live timings are reported
    for name, k in kernels.items():
        n = max(1, k["launches"])
        k["avg_launch_ms"] = k["ms"] / n
        k["bytes_per_launch"] = k["units"] * k["bytes_per_unit"] / n
        k["GBps"] = (k["units"] * k["bytes_per_unit"] / 1e9) / (k["ms"] / 1e3) if k["ms"] > 0 else 0.0
        k["Grays_or_Ghits_per_s"] = (k["units"] / 1e9) / (k["ms"] / 1e3) if k["ms"] > 0 else 0.0
        c = ctr.get(name)
        if c and k["ms"] > 0:
            secs = k["avg_launch_ms"] / 1e3
            if c.get("SQ_INSTS_VALU"):
                k["valu_insts_per_launch"] = c["SQ_INSTS_VALU"]
                k["valu_TFLOPs_equiv"] = c["SQ_INSTS_VALU"] * FLOP_PER_VALU_INST / secs / 1e12
                k["valu_issue_frac"] = k["valu_TFLOPs_equiv"] / VALU_PEAK_TFLOPS
            if c.get("SQ_ACTIVE_INST_VALU"):
                # time the vector ALUs spend executing (the counter ticks once per four cycles of an instruction in flight),
                # as a share of the launch: double-precision and transcendental instructions hold a SIMD longer than the two
                # cycles the issue-slot figure above prices every instruction at
                k["valu_busy_frac"] = c["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * secs * 2.4e9)
            if c.get("hbm_bytes_per_launch") is not None:
                k["hbm_bytes_per_launch"] = c["hbm_bytes_per_launch"]
                k["hbm_GBps"] = c["hbm_bytes_per_launch"] / 1e9 / secs
            if c.get("TCC_HIT_sum") is not None and (c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0)) > 0:
                k["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
            for extra in ("SQ_WAIT_ANY", "SQ_WAVE_CYCLES", "SQ_INSTS_VMEM_RD"):
                if c.get(extra):
                    k[extra + "_per_launch"] = c[extra]
    # the whole frame against the vector ALUs: every kernel's executing time (SQ_ACTIVE_INST_VALU, committed counter pass)
    # x its launches, as a share of one frame alone and - filled in by the caller - of a frame's share of the timed region
    frame_busy_s = sum(ctr[n]["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * 2.4e9) * max(1, k["launches"])
                       for n, k in kernels.items() if n in ctr and ctr[n].get("SQ_ACTIVE_INST_VALU"))
    dom = max(kernels, key=lambda n: kernels[n]["ms"])
    d = kernels[dom]
    alg = {"achieved_GBps": d["GBps"], "peak_GBps": HBM_PEAK_GBS, "frac": d["GBps"] / HBM_PEAK_GBS, "bytes_per_launch": d["bytes_per_launch"],
           "note": "prescribed per-ray record bytes x rays of the launch / launch duration; this implementation moves fewer bytes "
                   "(primary rays are generated in registers, shadow rays are never stored)"}
    traffic = d.get("hbm_bytes_per_launch")
    out = {"kernel": dom, "avg_launch_ms": d["avg_launch_ms"], "traffic": traffic, "counters_from": ctr_origin, "stale": bool(ctr_origin.get("stale")), "frame_valu_busy_ms": frame_busy_s * 1e3 if frame_busy_s else None,
           "traffic_ratio": (traffic / d["bytes_per_launch"]) if (traffic and d["bytes_per_launch"]) else None,
           "hbm_algorithmic": alg, "l2_hit_rate": d.get("l2_hit_rate"), "by_kernel": kernels}
    if small_scene and dom == "k_shade" and not ctr_origin.get("stale"):
        # the pass over the hit tree: records, counts and colours streamed once, a handful of dependent loads per hit, little arithmetic
        out.update(bound="hbm", achieved=d["GBps"], peak=HBM_PEAK_GBS, unit="GB/s", frac=d["GBps"] / HBM_PEAK_GBS,
                   hbm_measured_GBps=d.get("hbm_GBps"), valu_busy_frac=d.get("valu_busy_frac"), valu_issue_frac=d.get("valu_issue_frac"),
                   note="achieved = the prescribed algorithmic bytes of the launch (144 B per shaded hit) / launch duration measured live; "
                        "traffic = HBM bytes per launch from the committed counter pass (FETCH_SIZE x2 + WRITE_SIZE; fewer than the "
                        "prescribed bytes: records are 48 B and stay in the memory-side cache between the kernels), hbm_measured_GBps = "
                        "traffic / duration.  The pass waits for its dependent loads (hit record -> link -> child colour): "
                        "valu_busy_frac says how little of it is arithmetic")
    elif small_scene and "valu_issue_frac" in d:
        out.update(bound="valu", achieved=d["valu_TFLOPs_equiv"], peak=VALU_PEAK_TFLOPS, unit="TFLOP/s", frac=d["valu_issue_frac"],
                   valu_busy_frac=d.get("valu_busy_frac"),
                   note="a handful of primitives: nodes, primitive records and materials are served from LDS, HBM carries only the hit tree "
                        "and the ray queues, and the kernels are bound by vector-instruction issue (IEEE divides, double-precision "
                        "normalisation kept for bit parity).  achieved = vector instructions of the launch (rocprofv3 SQ_INSTS_VALU, "
                        "committed counter pass) x 128 flop (64 lanes x one FMA) / launch duration measured live: the FP32 vector rate "
                        "the used issue slots stand for; frac = share of the chip's vector issue slots in use; valu_busy_frac = share of the "
                        "launch during which the vector ALUs execute (SQ_ACTIVE_INST_VALU), which prices slow instructions at what they cost")
    elif not small_scene and d.get("hbm_GBps"):
        out.update(bound="hbm", achieved=d["hbm_GBps"], peak=HBM_PEAK_GBS, unit="GB/s", frac=d["hbm_GBps"] / HBM_PEAK_GBS,
                   valu_busy_frac=d.get("valu_busy_frac"), valu_issue_frac=d.get("valu_issue_frac"),
                   note="achieved = HBM bytes of the launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, committed counter pass) / launch duration "
                        "measured live.  The walk of the second-level hierarchies is bound by the latency of dependent node and "
                        "primitive fetches (L2 / Infinity Cache), not by streaming bandwidth; l2_hit_rate is from the same passes")
    elif ctr_origin.get("stale"):
        out.update(bound="valu" if small_scene else "hbm", achieved=None, peak=VALU_PEAK_TFLOPS if small_scene else HBM_PEAK_GBS,
                   unit="TFLOP/s" if small_scene else "GB/s", frac=None,
                   note="the committed counter pass was collected on other kernel sources (counters_from): no counter-based fraction; "
                        "hbm_algorithmic is the prescribed record bytes / launch duration measured live")
    else:
        out.update(bound="hbm", achieved=d["GBps"], peak=HBM_PEAK_GBS, unit="GB/s", frac=d["GBps"] / HBM_PEAK_GBS,
                   note="no committed counter pass for this workload: algorithmic record bytes / launch duration")
    return out


def copy_ceiling_GBps(torch, dev) -> float:
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


def run_workload(name: str, args, rank: int, world: int, local: int, dev, steps: int, warmup: int, frames_in_flight: int, detail: bool) -> dict | None:
    """Uploads the workload on every rank, times `steps` frames, returns rank 0's result dictionary."""
    import numpy as np
    import torch
    import torch.distributed as dist
    import cases
    from raytracer_amd import api
    from raytracer_amd import dist as rdist

    wl = WORKLOADS[name]
    W, H = wl["res"]
    # multi: the frame goes through the process group (gather to rank 0, re-ordering).  True for one rank too in the
    # RCCL rehearsal on a one-GPU machine, which runs the communicator calls of the N > 1 path with a group of one.
    multi = world > 1 or args.rehearse_rccl
    # Rank 0 parses the scene and builds the BVH on its host; flags and the flat scene/BVH arrays are broadcast once
    # per frame set (over RCCL with the nccl backend), every rank uploads its copy.
    on_dev = dev if (world > 1 and args.backend != "gloo") else None
    flags = json.loads(rdist.broadcast_bytes(json.dumps(wl["flags"]).encode() if rank == 0 else None, 0, on_dev).decode())
    cfg, _ = api.parse_args(flags)
    ls = None
    t_load = time.perf_counter()
    if rank == 0:
        if wl["scene"] == "ref":
            ls = api.LoadedScene(cases.reference_scene(W, H), cfg, str(cases.TEXTURES))
        elif wl["scene"].startswith("blend:"):
            ls = api.LoadedScene(cases.blend_scene(wl["scene"][6:], W, H), cfg, str(cases.TEXTURES))
        else:
            from raytracer_amd import scenegen as sg
            moving = bool(cfg.motion_blur) and os.environ.get("RT_BENCH_STATIC", "") != "1"
            ls = api.ArrayScene(sg.random_prims(wl["n"], wl.get("seed", 42), moving=moving), (W, H), moving=moving)
    # Over RCCL the scene travels device to device: rank 0 uploads (and builds the second-level hierarchies on its GPU) once,
    # the other ranks adopt the layout and receive every device array by broadcast; no rank but 0 touches host scene data.
    # (gloo rehearsals move the flat host arrays instead and every rank uploads its copy.)
    device_scene = multi and args.backend != "gloo"
    if not device_scene:
        ls = rdist.broadcast_scene(ls, bool(cfg.bvh), 0, on_dev)
    # Frames in flight: consecutive frames are independent, so P contexts (each with its own work buffers and stream)
    # render them concurrently; a frame's small launches (deep recursion levels) then fill the other frames' gaps.
    # Four by default: one per hardware queue of the process (more share queues and lose again).  Not used for frames
    # of many samples per pixel, which keep the GPU busy on their own.
    spp = cfg.samples_per_pixel()
    P = max(1, frames_in_flight) if (spp <= 64 and W * H <= 1920 * 1080) else 1
    if multi:
        P = min(P, 3)                                 # the communicator's stream takes the fourth hardware queue

    def set_mode(rr, in_flight: bool):
        """The library's defaults serve one frame at a time (shadow rays of the primary hits on a forked stream, one
        workgroup per chunk or two in the soft-shadow launches so that no workgroup holds the frame up).  Frames in
        flight fill each other's gaps instead: no forked streams (hardware queues are few) and the smaller grid, which
        costs fewer workgroup start-ups."""
        rr.set_option("overlap", 0 if (in_flight and (P >= 3 or multi)) else 1)
        rr.set_option("grid_shadow", 1536 if (in_flight and P >= 3 and soft_small) else 0)
        rr.set_option("grid_primary", 1024 if (in_flight and P >= 3 and one_kernel_small) else 0)
        for kv in filter(None, os.environ.get("RT_BENCH_OPTS", "").split(",")):      # development: name=value library options
            rr.set_option(kv.split("=")[0], int(kv.split("=")[1]))

    soft_small = bool(cfg.soft_shadows) and wl["scene"] != "random"
    one_kernel_small = (not cfg.soft_shadows) and cfg.reflect_refract_depth <= 0 and wl["scene"] != "random"     # frames rendered by one kernel

    def new_renderer():
        rr = api.Renderer(local, workspace_bytes=16 << 30)
        set_mode(rr, True)
        return rr

    r = new_renderer()
    if device_scene:
        if rank == 0:
            r.upload(ls, with_bvh=bool(cfg.bvh))
        rdist.broadcast_device_scene(r, 0, dev)
    else:
        r.upload(ls, with_bvh=bool(cfg.bvh))
    r.set_shard(rank, world)
    upload_s = time.perf_counter() - t_load
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
        if os.environ.get("RT_BENCH_SHARE", "1") != "0":
            r2.share_from(r)                          # the rank's other contexts render from the first one's arrays: one copy of the scene in the caches
        elif device_scene:
            r2.adopt_from(r)
        else:
            r2.upload(ls, with_bvh=bool(cfg.bvh))
        r2.set_shard(rank, world)
        s2 = torch.cuda.ExternalStream(r2.stream_handle(), device=dev)
        ctxs.append((r2, s2, torch.empty_like(block)))

    staged = args.backend == "gloo"      # rehearsal path: gloo moves host tensors

    # N > 1 over RCCL: frames are pipelined.  Frame k is rendered on the render stream into one of two blocks; its
    # gather runs on the communicator's stream and rank 0 re-orders it, both beside the kernels of frame k + 1.
    # Buffers are reused every other frame behind events; the timed region ends with a device-wide synchronisation,
    # so every frame's image has reached rank 0 inside it.
    pipelined = multi and not staged
    # With several pipelines taking turns, rank 0 re-orders each frame on that frame's own render stream (no fifth stream)
    inline = P >= 2
    asm_stream = torch.cuda.Stream(dev) if (pipelined and rank == 0 and not inline) else None
    pipes = [rdist.FramePipeline(rr, cfg, rank, world, dev, ss, asm_stream=asm_stream, assemble_inline=inline) for rr, ss, _ in ctxs] if pipelined else None
    frame_no = [0]
    keep = []

    def step():
        k = frame_no[0] % P; frame_no[0] += 1
        if pipelined:
            pipes[k].step()
            return
        if world == 1:
            rr, _, bb = ctxs[k]
            rr.render_device(cfg, bb.data_ptr())
            return
        r.render_device(cfg, block.data_ptr())
        g = rdist.gather_blocks(block.cpu(), 0)
        if rank == 0:
            gathered = g.to(dev)
            r.assemble(gathered.data_ptr(), world, image.data_ptr())
            keep.append(gathered)            # the assemble kernel is only enqueued: keep its input alive
            del keep[:-2]

    # Ray counts (identical every frame: the generator is counter based) and per-kernel device times.  The profiled
    # frame is enqueued behind four ordinary ones so that the GPU, not the host, paces it.
    def note(msg):
        if rank == 0:
            print(f"[bench {name}] {msg}", file=sys.stderr, flush=True)

    note(f"scene uploaded in {upload_s:.2f} s")
    set_mode(r, False)       # ray counts, stage times and the latency of one frame: the library as a one-frame-at-a-time caller gets it
    st = r.render_device(cfg, block.data_ptr(), want_stats=True)     # ray counts (the counting kernel variants)
    heavy = st.ms_total > 150.0                                      # a frame of hundreds of ms: no extra frames beyond the timed ones
    note(f"first frame {st.ms_total:.1f} ms, {st.total_rays()} rays, {st.batches} batches")
    shadow_all = shadow_no_cone = shadow_no_umbra = None
    if heavy:
        steps = min(steps, 3); warmup = min(warmup, 1)
    if detail and heavy:
        r.set_option("count_rays", 0)
        tm = r.render_device(cfg, block.data_ptr(), want_stats=True)
        r.set_option("count_rays", 1)
        for field in ("ms_extend_primary", "ms_extend_secondary", "ms_shadow", "ms_shade", "ms_other", "ms_total",
                      "launches_extend_primary", "launches_extend_secondary", "launches_shadow", "launches_shade"):
            setattr(st, field, getattr(tm, field))
        note(f"stage times: {tm.ms_total:.1f} ms")
    elif detail and args.no_variants:
        for _ in range(4):
            r.render_device(cfg, block.data_ptr())
        r.set_option("count_rays", 0)
        tm = r.render_device(cfg, block.data_ptr(), want_stats=True)
        r.set_option("count_rays", 1)
        for field in ("ms_extend_primary", "ms_extend_secondary", "ms_shadow", "ms_shade", "ms_other", "ms_total",
                      "launches_extend_primary", "launches_extend_secondary", "launches_shadow", "launches_shade"):
            setattr(st, field, getattr(tm, field))
    elif detail:
        r.set_option("skip_unlit", 0); r.set_option("cull_cones", 0)  # how many shadow rays the N.L <= 0 rule leaves untraced ...
        shadow_all = r.render_device(cfg, block.data_ptr(), want_stats=True).shadow_rays
        r.set_option("skip_unlit", 1)                                # ... and the cone test of soft shadows (tiny scenes)
        shadow_no_cone = r.render_device(cfg, block.data_ptr(), want_stats=True).shadow_rays
        r.set_option("cull_cones", 1); r.set_option("cull_umbra", 0)  # ... and the pairs in full shadow of one sphere or cube
        shadow_no_umbra = r.render_device(cfg, block.data_ptr(), want_stats=True).shadow_rays
        r.set_option("cull_umbra", 1)
        for _ in range(4):
            r.render_device(cfg, block.data_ptr())
        r.set_option("count_rays", 0)                                # stage times with the kernels the timed frames run
        tm = r.render_device(cfg, block.data_ptr(), want_stats=True)
        r.set_option("count_rays", 1)
        for field in ("ms_extend_primary", "ms_extend_secondary", "ms_shadow", "ms_shade", "ms_other", "ms_total",
                      "launches_extend_primary", "launches_extend_secondary", "launches_shadow", "launches_shade"):
            setattr(st, field, getattr(tm, field))
    rays_local = torch.tensor([st.primary_rays, st.shadow_rays, st.reflection_rays, st.refraction_rays,
                               st.shadow_rays_ref + st.reflection_rays_ref + st.refraction_rays_ref, st.nodes_shaded],
                              dtype=torch.float64, device=dev)
    if multi:
        rays_local = rays_local.cpu() if staged else rays_local
        dist.all_reduce(rays_local)
    tot = [int(x) for x in rays_local.tolist()]
    rays_total = tot[0] + tot[1] + tot[2] + tot[3]

    # latency of one frame alone (enqueue -> done), next to the throughput the timed loop measures
    torch.cuda.synchronize(dev)
    n_lat = 1 if heavy else (20 if spp < 16 else 2)
    for _ in range(1 if heavy else 3):                   # (the third request replays the frame's recorded graph)
        r.render_device(cfg, block.data_ptr()); r.synchronize()
    t_lat = time.perf_counter()
    for _ in range(n_lat):
        r.render_device(cfg, block.data_ptr()); r.synchronize()
    frame_latency_ms = (time.perf_counter() - t_lat) * 1000.0 / n_lat
    # the same frame through rt_render (waits, copies the image to the host): into page-locked memory from
    # rt_alloc_image and into an ordinary array.  Reported, never `value`.
    to_host = None
    if detail and world == 1 and not heavy:
        to_host = {}
        for kind, buf in (("page_locked", api.image_buffer(r.out_bytes())), ("pageable", np.zeros(r.out_bytes(), dtype=np.uint8))):
            r.render(cfg, out=buf, want_stats=False)
            each = []
            for _ in range(n_lat):
                t_h = time.perf_counter()
                r.render(cfg, out=buf, want_stats=False)
                each.append((time.perf_counter() - t_h) * 1000.0)
            to_host[kind] = sorted(each)[len(each) // 2]             # median: a frame to the host is a host-timed quantity
        del buf
    # every context records a frame graph on its second frame with the same output buffer; a pipeline alternates
    # between two blocks, so it needs four frames before both graphs exist
    # Part of the set-up, not of the warm-up: these frames make every context record its graph(s).  Then exactly `warmup`
    # untimed steps and `steps` timed ones.
    set_mode(r, True)
    prime_steps = 0 if heavy else (4 if pipelined else 2) * P
    note(f"one frame alone {frame_latency_ms:.2f} ms; {prime_steps} frames to record the frame graphs, warming up ({warmup}) and timing {steps} steps")
    for _ in range(prime_steps):
        step()
    torch.cuda.synchronize(dev)
    for _ in range(warmup):
        step()
    if multi:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    t_enq = time.perf_counter() - t0
    if multi:
        dist.barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if staged else dev)
    if multi:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    res = None
    if rank == 0:
        # rank-0 sanity: the frame is not black and (N=1) is the row-major image itself
        last = (frame_no[0] - 1) % P
        final = ((pipes[last].image if pipelined else image) if multi else ctxs[last][2]).cpu().numpy()
        lit = float((final.reshape(-1, 3).sum(axis=1) > 0).mean())
        import zlib
        crc = zlib.crc32(final.tobytes())                # of the row-major RGB bytes of the last timed frame
        res = {"workload": describe(name), "rays_total": rays_total, "value": rays_total / (elapsed / steps), "ms_per_step": 1000.0 * elapsed / steps,
               "steps": steps, "warmup": warmup, "graph_recording_frames": prime_steps, "spp": spp, "frames_in_flight": P, "frame_ms_alone": frame_latency_ms,
               "frame_ms_to_host": to_host, "host_enqueue_ms_per_step": 1000.0 * t_enq / steps, "upload_seconds": upload_s,
               "rays_per_frame": {"primary": tot[0], "shadow": tot[1], "reflection": tot[2], "refraction": tot[3],
                                  "hits_shaded": tot[5], "as_the_reference_would_trace_them": tot[0] + tot[4],
                                  "shadow_not_traced_because_the_surface_faces_away_from_the_light":
                                      (shadow_all - shadow_no_cone) if (world == 1 and shadow_all is not None) else None,
                                  "shadow_not_traced_because_no_sample_ray_of_the_pair_can_reach_a_primitive":
                                      (shadow_no_cone - shadow_no_umbra) if (world == 1 and shadow_no_cone is not None) else None,
                                  "shadow_not_traced_because_every_sample_ray_of_the_pair_is_certain_to_be_blocked":
                                      (shadow_no_umbra - st.shadow_rays) if (world == 1 and shadow_no_umbra is not None) else None},
               "lit_fraction": lit, "image_crc32": crc, "stats": st}
    torch.cuda.synchronize(dev)
    torch.cuda.set_stream(torch.cuda.default_stream(dev))      # the contexts' streams die with them
    for rr, _, _ in reversed(ctxs):                  # (contexts that share the first one's scene go first)
        rr.close()
    if ls is not None:
        ls.close()
    del pipes, ctxs, block, image, keep
    torch.cuda.synchronize(dev)
    torch.cuda.empty_cache()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)        # a frame takes milliseconds: time enough of them
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--workload", default="config2", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-scaling-4k", action="store_true", help="skip the second, 4K multi-sample measurement (scaling_4k)")
    ap.add_argument("--frames-in-flight", type=int, default=4,
                    help="contexts rendering consecutive frames concurrently on their own streams (1 = one frame at a time)")
    ap.add_argument("--backend", default=None, help="torch.distributed backend (default: nccl = RCCL)")
    ap.add_argument("--single-device", action="store_true",
                    help="rehearsal on a one-GPU machine: every rank uses GPU 0 (use with --backend gloo)")
    ap.add_argument("--no-variants", action="store_true",
                    help="profiling runs: no frames with other library options (the counts of untraced shadow rays are then not reported)")
    ap.add_argument("--rehearse-rccl", action="store_true",
                    help="one-GPU machine: one rank in an RCCL process group of one, frames through the N > 1 path (pipelined gather, re-ordering)")
    ap.add_argument("--launch-check", action="store_true",
                    help="start the ranks, initialise the process group, reduce one number, print it and stop (needs no GPU)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))

    from raytracer_amd import dist as rdist
    rank, world, local = rdist.env_rank_world()
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)

    # The CPU leg first: it forks workers and runs the reference binary as a child, which must not happen in a process
    # that has initialised the GPU runtime.
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.launch_check:
        try:
            cpu = cpu_baseline(WORKLOADS[args.workload])
        except Exception as e:
            cpu = {"unit": "rays/s", "value": None, "cores": 1, "kind": "port", "sample": f"failed: {e!r}"}

    import torch
    import torch.distributed as dist
    if args.single_device:
        os.environ["LOCAL_RANK"] = "0"; local = 0
    if args.launch_check:
        rdist.init_process_group(args.backend or "gloo")
        x = torch.tensor([float(rank + 1)])
        if world > 1:
            dist.all_reduce(x); dist.barrier()
        if rank == 0:
            print(json.dumps({"launch_check": "ok", "world": world, "sum_of_ranks_plus_one": float(x.item())}), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return
    rank, world, local = rdist.init_process_group(args.backend)
    if args.rehearse_rccl and world == 1 and torch.cuda.is_available():
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29533", rank=0, world_size=1, device_id=torch.device("cuda", local))
    if not torch.cuda.is_available():
        print("bench.py needs a GPU: the render path has no CPU fallback", file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    head = run_workload(args.workload, args, rank, world, local, dev, args.steps, args.warmup, args.frames_in_flight, detail=True)
    big = None
    if not args.no_scaling_4k and args.workload == "config2":
        # the frame the tile split is designed for: 4K, 64 samples per pixel (config 2 alone is a sub-millisecond frame per shard)
        big = run_workload("config4s", args, rank, world, local, dev, steps=max(3, min(10, args.steps)), warmup=2, frames_in_flight=1, detail=False)

    if rank == 0:
        wl = WORKLOADS[args.workload]
        st = head.pop("stats")
        line = {
            "metric": "rays/sec (all ray types), 1080p scene.json" if args.workload == "config2" else f"rays/sec (all ray types), {args.workload}",
            "value": head["value"], "unit": "rays/s", "n_gpus": world, "steps": head["steps"], "warmup": head["warmup"],
            "ms_per_step": head["ms_per_step"], "frame_ms_alone": head["frame_ms_alone"],
            "value_one_frame_at_a_time": head["rays_total"] / (head["frame_ms_alone"] / 1e3),
            "frames_per_s": 1e3 / head["ms_per_step"],
            # the rays the reference traces for the same frames (raytracer.cpp:400-432, 534-571 count every sample ray and every
            # identical reflection ray): what a caller that switches libraries gets per second, not what this library traces
            "reference_rays_per_s": head["rays_per_frame"]["as_the_reference_would_trace_them"] / (head["ms_per_step"] / 1e3),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic: the reference's own ASCII/scene.json (committed as a fixture), no dataset" if wl["scene"] == "ref"
                    else f"the reference's Blend/{wl['scene'][6:]}.blend exported to scene.json text (committed as a fixture)" if wl["scene"].startswith("blend:")
                    else f"synthetic: {wl['n']} random spheres/cubes (seed {wl.get('seed', 42)}) over a ground plane, generated in-process",
            "config": {"workload": head["workload"], "spp": head["spp"], "tiles": "32x32 interleaved, rank = (tx+ty) % N",
                       "frames_in_flight": head["frames_in_flight"], "frame_ms_alone": head["frame_ms_alone"],
                       "host_enqueue_ms_per_step": head["host_enqueue_ms_per_step"],
                       "graph_recording_frames_before_warmup": head["graph_recording_frames"],
                       "frame_ms_to_host": head.get("frame_ms_to_host"),
                       "timing": "ms_per_step = timed wall clock / steps with frames_in_flight independent frames overlapping on the GPU; "
                                 "frame_ms_alone = one frame enqueued and waited for by itself",
                       "soft_shadow_sampling": "the first reflection of every path is recorded gr_samples times (each copy draws its own shadow "
                                               "samples, as the reference's identical rays do; one traced ray per bundle); deeper reflections once",
                       "value_counts": "traced rays only: shadow rays whose outcome is certain without tracing (surface facing away from the light; "
                                       "no sample ray of the pair can reach a primitive; every sample ray of the pair is certain to be blocked) and "
                                       "the identical copies of a reflection ray are not counted - a rule that leaves more rays untraced lowers "
                                       "`value` while it shortens the frame: compare frames (ms_per_step, frame_ms_alone) across builds, not rays",
                       "rays_per_frame": head["rays_per_frame"], "lit_fraction": head["lit_fraction"], "image_crc32": head["image_crc32"]},
            "roofline": roofline(st, args.workload, wl["scene"] != "random"),
        }
        ceiling = copy_ceiling_GBps(torch, dev)
        line["roofline"]["measured_copy_GBps"] = ceiling
        busy_ms = line["roofline"].get("frame_valu_busy_ms")
        if busy_ms:
            # vector ALUs executing, whole frame: alone, and with the frames in flight of the timed region
            line["roofline"]["frame_valu_busy_frac_alone"] = busy_ms / head["frame_ms_alone"]
            line["roofline"]["frame_valu_busy_frac_in_flight"] = busy_ms / head["ms_per_step"]
        if big is not None:
            big.pop("stats")
            line["scaling_4k"] = {"metric": "rays/sec (all ray types), 4K multi-sample frame", "unit": "rays/s", "n_gpus": world,
                                  **{k: big[k] for k in ("workload", "value", "ms_per_step", "steps", "warmup", "spp", "frames_in_flight", "frame_ms_alone", "rays_per_frame")}}
        if cpu is not None:
            line["cpu_baseline"] = cpu
        print(json.dumps(line), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
