"""Exports the reference's .blend scenes to scene.json text with raytracer_amd.blend_export (what Blend/Export.py writes
when run inside Blender).  Runs only where /root/reference is mounted; the outputs are data fixtures:
  tests/golden/scenes/<name>.json  for every Blend/<name>.blend
The one pair the reference ships, Blend/dof.blend -> ASCII/scene.json, must come out byte-identical."""
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent.parent))
from raytracer_amd import blend_export  # noqa: E402

REF = Path("/root/reference")


def main():
    out = HERE / "scenes"
    out.mkdir(exist_ok=True)
    for blend in sorted((REF / "Blend").glob("*.blend")):
        text = blend_export.export_scene_json(str(blend))
        (out / (blend.stem + ".json")).write_text(text)
        print(f"{blend.name:28s} {len(text):7d} bytes")
    assert (out / "dof.json").read_text() == (REF / "ASCII" / "scene.json").read_text(), "dof.blend no longer reproduces ASCII/scene.json"


if __name__ == "__main__":
    main()
