#!/usr/bin/env python
"""TEST INFRASTRUCTURE — mints tests/golden/opensora_image_sizes.json from the reference's literal tables
(/root/reference/videosys/pipelines/open_sora/data_process.py:40-505).  The module cannot be imported here (torchvision,
requests), so the dict literals are read with ``ast``; nothing is copied into the product.
    python oracle/make_golden_geometry.py"""
import ast
import json
import os

REF = "/root/reference/videosys/pipelines/open_sora/data_process.py"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def reference_tables(path=REF):
    mod = ast.parse(open(path).read())
    lit, ratios_src = {}, None
    for n in mod.body:
        if isinstance(n, ast.Assign) and isinstance(n.targets[0], ast.Name):
            name = n.targets[0].id
            if name == "ASPECT_RATIOS":
                ratios_src = n.value
            elif name.startswith("ASPECT_RATIO") or name == "NUM_FRAMES_MAP":
                lit[name] = ast.literal_eval(n.value)
    res_tab = {k.value: v.elts[1].id for k, v in zip(ratios_src.keys, ratios_src.values)}
    return lit, res_tab


def main():
    lit, res_tab = reference_tables()
    sizes = {}
    for res, tab_name in res_tab.items():
        for ar, key in lit["ASPECT_RATIO_MAP"].items():
            tab = lit[tab_name]
            sizes[f"{res}|{ar}"] = list(tab[key]) if key in tab else None   # None: the reference's get_image_size asserts
    out = dict(image_sizes=sizes, ratio_keys=lit["ASPECT_RATIO_MAP"], num_frames=lit["NUM_FRAMES_MAP"])
    with open(os.path.join(ROOT, "tests", "golden", "opensora_image_sizes.json"), "w") as fh:
        json.dump(out, fh, indent=0, sort_keys=True)
    print(len(sizes), "pairs,", sum(v is None for v in sizes.values()), "assert in the reference")


if __name__ == "__main__":
    main()
