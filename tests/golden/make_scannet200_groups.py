"""Extracts the ScanNet200 head / common / tail category-name lists (benchmark metadata the reference averages
over, utils_instance_seg_3d_eval.py:247-281) into segdino3d_amd/data/scannet200_groups.json.  Build container only."""
import ast
import json
import os

src = open("/root/reference/evaluation/utils_instance_seg_3d_eval.py").read()
tree = ast.parse(src)
out = {}
for node in ast.walk(tree):
    if isinstance(node, ast.Assign) and len(node.targets) == 1 and isinstance(node.targets[0], ast.Name):
        name = node.targets[0].id
        if name.endswith("_cats_scannet_200"):
            out[name.split("_")[0]] = ast.literal_eval(node.value)
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
path = os.path.join(root, "segdino3d_amd", "data", "scannet200_groups.json")
with open(path, "w") as f:
    json.dump(out, f, indent=0)
print({k: len(v) for k, v in out.items()}, "->", path)
