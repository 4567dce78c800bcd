"""Development aid: the sampling noise behind the seed-mean soft-shadow cases (tests/cases.SEED_MEAN) - PSNR between the
GPU's mean over seeds, the restatement's mean over seeds and the reference's high-sample golden."""
import sys, copy, numpy as np
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests")); sys.path.insert(0, str(ROOT / "oracle"))
import cases
from raytracer_amd import api
scene, flags, gold_flags = cases.SEED_MEAN[1]
cfg, _ = api.parse_args(flags)
ls = api.LoadedScene(cases.SCENES[scene](), cfg, str(cases.TEXTURES))
out = {}
for name, opts in (("default", {}),):
    r = api.Renderer(0)
    for k, v in opts.items(): r.set_option(k, v)
    r.upload(ls, with_bvh=True)
    imgs = []
    for s in range(1, 65):
        c = copy.copy(cfg); c.seed = s
        imgs.append(r.render(c)[0])
    r.close()
    x = np.stack(imgs).astype(np.float64)
    out[name + "_mean16"] = x[:16].mean(0); out[name + "_mean64"] = x.mean(0); out[name + "_var64"] = x.var(0, ddof=1)
np.savez_compressed(ROOT / "gpurun_out" / "seedmean.npz", **out)
print("saved")
