#!/usr/bin/env python3
"""Print the numbers of a bench.py JSON line that DESIGN.md / README.md quote:  python tools/bench_digest.py profiles/r03_bench_default.json"""
import json
import sys

d = json.load(open(sys.argv[1]))
r = d["roofline"]
print("headline %s Mpix/s  ms/step %s  kernel %s ms  frac %s  ceiling %s GB/s  of measured %s  sclk %s  traffic %s  bound %s" % (
    d["value"], d["ms_per_step"], r["kernel_ms_mean"], r["frac"], r.get("measured_ceiling_gbs"), r.get("frac_of_measured"), r.get("sclk_mhz"), r.get("traffic"), r["bound"]))
print("probe", r.get("measured_ceiling"))
for v in d.get("variants", []):
    rr = v["roofline"]
    rv = rr.get("roofline_valu") or {}
    print("%-18s %-8s %-9s kernel %.1f us  frac %.3f  traffic %s (x%.3f)  bound %s  valu %s busy %s  hbm share %s" % (
        v["dtype"], v["interp"], v["homography"], 1e3 * rr["kernel_ms_mean"], rr["frac"], rr.get("traffic"),
        (rr["traffic"] / rr["algorithmic_bytes_per_launch"]) if rr.get("traffic") else float("nan"), rr["bound"], rv.get("insts_per_launch"), rv.get("busy_frac"),
        rr.get("hbm_share_of_streaming_ceiling")))
c = d.get("configs", {})
if c:
    g = c["configs[0]"]["gpu_resident"]
    print("configs[0] %.2f us per call, host %.2f; pcie serial %.3f ms; oracle %.2f / %.2f ms" % (g["us_median"], g["host_us_per_call"], c["configs[0]"]["gpu_pcie_inclusive_serial"]["ms_median"],
                                                                                               c["configs[0]"]["cpu_oracle_1_thread"]["ms_median"], c["configs[0]"]["cpu_oracle_all_cores"]["ms_median"]))
    print("configs[2] f32 %.1f us  f64 %.1f us" % (c["configs[2]"]["f32"]["us_mean"], c["configs[2]"]["f64"]["us_mean"]))
    print("configs[3]", {k: (round(1e3 * v["roofline"]["kernel_ms_mean"], 1), v["roofline"]["frac"]) for k, v in c["configs[3]"].items() if isinstance(v, dict)})
    print("configs[4]", {k: v for k, v in c["configs[4]"].items() if k != "workload"})
    print("composite", {k: v for k, v in c["f3_composite"].items() if k.endswith("us")})
    print("pcie", {k: v for k, v in c["pcie_pipeline"].items() if k not in ("workload", "what")})
cb = d["cpu_baseline"]
print("cpu %s Mpix/s on %s cores, single %s" % (cb["value"], cb["cores"], cb.get("single_thread_value")))
