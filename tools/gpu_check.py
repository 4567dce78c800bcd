"""Ad-hoc GPU parity sweep: renders every deterministic case on the GPU and with the CPU restatement (checker)
and prints the per-case pixel differences. Development aid; the real tests are tests/test_gpu_parity.py."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "oracle")); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import oracle
from raytracer_amd import api
import cases

def main():
    worst = 0
    for scene, flags in cases.DETERMINISTIC:
        txt = cases.SCENES[scene]()
        t0 = time.time()
        img, st = api.render_scene_text(txt, flags, str(cases.TEXTURES))
        t1 = time.time() - t0
        ref, ost = oracle.render(txt, flags, str(cases.TEXTURES))
        d = np.abs(img.astype(np.int32) - ref)
        npx = int((d.max(axis=2) > 0).sum()); mx = int(d.max())
        worst = max(worst, mx)
        print(f"{cases.case_id(scene, flags):40s} maxdiff {mx:3d} px!=0 {npx:6d} gt1 {(d.max(axis=2) > 1).sum():5d} "
              f"rays(ref-equiv) gpu {st.total_rays_ref()} orc {ost.total_rays()} traced {st.total_rays()}  gpu {st.ms_total:.2f} ms wall {t1:.2f}s", flush=True)
    print("worst", worst)

if __name__ == "__main__":
    main()
