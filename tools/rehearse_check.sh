#!/bin/bash
# Development aid: the last frame of bench.py without and with the one-rank RCCL group against a plain context's frame.
cd "$(dirname "$0")/.." || exit 1
A="--steps 7 --warmup 2 --no-cpu-baseline --no-scaling-4k"
timeout -k 10 200 python bench.py $A 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('plain   ', d['config']['image_crc32'], d['config']['frames_in_flight'])" || exit 1
timeout -k 10 200 python bench.py $A --rehearse-rccl 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rehearse', d['config']['image_crc32'], d['config']['frames_in_flight'])" || exit 1
timeout -k 10 200 python bench.py $A --rehearse-rccl --frames-in-flight 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rehearse P=1', d['config']['image_crc32'], d['config']['frames_in_flight'])" || exit 1
timeout -k 10 200 python - <<'PY'
import sys, zlib
sys.path.insert(0, "tests")
import numpy as np, bench, cases
from raytracer_amd import api
wl = bench.WORKLOADS["config2"]
cfg, _ = api.parse_args(wl["flags"])
print("seed", cfg.seed)
ls = api.LoadedScene(cases.reference_scene(*wl["res"]), cfg, str(cases.TEXTURES))
r = api.Renderer(0); r.upload(ls, with_bvh=bool(cfg.bvh))
img, _ = r.render(cfg)
print("direct  ", zlib.crc32(np.ascontiguousarray(img).tobytes()), img.shape, img.dtype)
r.set_shard(0, 1)
img2, _ = r.render(cfg)
print("direct shard(0,1)", zlib.crc32(np.ascontiguousarray(img2).tobytes()))
PY
