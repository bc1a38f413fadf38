"""Builds library variants for A/B runs: python3 tools/build_ab.py name="-DFOO=1 -DBAR=2" name2="" ...
-> tools/_ab/<name>.so (objects in tools/_ab/obj_<name>)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "grid-vision_amd"))
from gvamd import build as b
os.makedirs(os.path.join(ROOT, "tools/_ab"), exist_ok=True)
for arg in sys.argv[1:]:
    name, _, flags = arg.partition("=")
    print(b.build(lib=os.path.join(ROOT, f"tools/_ab/{name}.so"), extra=flags.split(), obj_dir=os.path.join(ROOT, f"tools/_ab/obj_{name}")))
