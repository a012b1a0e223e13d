#!/usr/bin/env python3
"""Regenerate tests/golden/scenes/ from the reference's data files.

These are DATA fixtures (scene descriptions / meshes the reference ships under
/root/reference/data), not source code.  JSON files are minified (same numbers:
Python's float repr round-trips every f64) so the 2 MB scene_500.json travels as
~420 KB; the YAML twin of scene_10 and the two OBJ meshes are kept byte-for-byte
because the product's YAML / OBJ readers are tested on them.
Run here (needs /root/reference); the GPU box only ever sees the committed output.
"""
import json, os, shutil, sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/data"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "scenes")
os.makedirs(OUT, exist_ok=True)
for name in ("scene_10", "scene_200_no_bvh", "scene_500", "test"):
    with open(os.path.join(REF, name + ".json")) as f:
        doc = json.load(f)
    with open(os.path.join(OUT, name + ".json"), "w") as f:
        json.dump(doc, f, separators=(",", ":"))
shutil.copyfile(os.path.join(REF, "scene_10.yaml"), os.path.join(OUT, "scene_10.yaml"))
for name in ("cube.obj", "bun315.obj"):
    shutil.copyfile(os.path.join(REF, "mesh", name), os.path.join(OUT, name))
for n in sorted(os.listdir(OUT)):
    print(n, os.path.getsize(os.path.join(OUT, n)))
