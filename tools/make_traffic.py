#!/usr/bin/env python3
"""profiles/pmc_traffic.json from the committed rocprofv3 summaries (tools/prof.sh output copied to profiles/).
HBM bytes per launch = FETCH_SIZE (KiB) x 1024 x 2 (gfx950 correction, MI355X_MICROARCH.md HBM section) + WRITE_SIZE (KiB) x 1024."""
import json
import os
import re
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r02"
# the kernel source the profiled library was built from: bench.py reports `traffic` only while it still matches
sys.path.insert(0, root)
import importlib.util  # noqa: E402
_spec = importlib.util.spec_from_file_location("bench", os.path.join(root, "bench.py"))
_bench = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_bench)
out = {"kernel_source_sha": _bench.kernel_source_sha()}  # the same hash bench.py checks
# keys as bench.py looks them up: <dtype>_<interp>[_<homography>]; files as tools/prof.sh tags them
for key, tag in (("u8_linear", "u8_linear"), ("f32_linear", "f32_linear"), ("u8_nearest", "u8_nearest"), ("u8_linear_brno", "u8_brno"),
                 ("f32_linear_brno", "f32_brno"), ("u8_nearest_brno", "u8_nearest_brno")):
    path = os.path.join(root, "profiles", "%s_%s_rocprofv3_summary.txt" % (rnd, tag))
    if not os.path.exists(path):
        continue
    txt = open(path).read()
    fetch = float(re.search(r"FETCH_SIZE\s+([\d.]+)", txt).group(1))
    write = float(re.search(r"WRITE_SIZE\s+([\d.]+)", txt).group(1))
    avg = float(re.search(r"warp_rows<.*?avg\s+([\d.]+) ns", txt).group(1))
    sq = {m.group(1): float(m.group(2)) for m in re.finditer(r"^\s+(SQ_\w+|GRBM_GUI_ACTIVE)\s+([\d.]+)", txt, re.M)}
    out[key] = {"hbm_bytes_per_launch": int(fetch * 1024 * 2 + write * 1024), "fetch_size_kib_raw": fetch, "write_size_kib": write,
                "fetch_correction": "x2 (gfx950, MI355X_MICROARCH.md HBM section)", "kernel_avg_ns_profiled": avg, "sq": sq,
                "source": os.path.relpath(path, root)}
with open(os.path.join(root, "profiles", "pmc_traffic.json"), "w") as f:
    json.dump(out, f, indent=1)
print(json.dumps(out, indent=1))
