#!/usr/bin/env python3
"""Copy the artefacts of a GPU session into profiles/ (run HERE after gpurun has merged gpurun_out/):
  python scripts/refresh_profiles.py <session dir under gpurun_out, e.g. s44> [tag]
Expects <session>/bench.json (python bench.py), optionally <session>/benchB.log (bench.py --batch 512/256/128) and
<session>/dense.log (scripts/bench_configs.py dense), and gpurun_out/prof_<tag>/ from scripts/collect_profile.sh."""
import glob, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ses = os.path.join(ROOT, "gpurun_out", sys.argv[1])
tag = sys.argv[2] if len(sys.argv) > 2 else "r03"
prof = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
P = os.path.join(ROOT, "profiles")
shutil.copy(os.path.join(prof, "summary.json"), os.path.join(P, f"{tag}_pmc_summary.json"))
shutil.copy(os.path.join(prof, "summary.txt"), os.path.join(P, f"{tag}_profile_summary.txt"))
stats = sorted(glob.glob(os.path.join(prof, "trace", "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)
shutil.copy(stats[-1], os.path.join(P, f"{tag}_kernel_stats.csv"))
line = [l for l in open(os.path.join(ses, "bench.json")) if l.startswith("{")][-1]
open(os.path.join(P, f"{tag}_bench.json"), "w").write(line)
rows = []
d = json.loads(line)
rows.append((1024, d["value"], d["ms_per_step"], d["roofline"]["phase_ms"]["lq"]))
bb = os.path.join(ses, "benchB.log")
if os.path.exists(bb):
    for l in open(bb):
        if l.startswith("{"):
            d = json.loads(l)
            rows.append((d["config"]["batch_per_gpu"], d["value"], d["ms_per_step"], d["roofline"]["phase_ms"]["lq"]))
    t = {b: ms for b, _, ms, _ in rows}
    path = os.path.join(P, f"{tag}_strong_scaling_proxy.txt")
    s = open(path).read()
    a, b = s.index("    B     accepted steps/s"), s.index("    predicted speed-up")
    s = s[:a] + "    B     accepted steps/s   ms per step   LQ ms\n" + "".join(
        f"    {bb_:<5d} {v:>14.0f}   {ms:>10.3f}   {lq:>6.3f}\n" for bb_, v, ms, lq in rows) + s[b:]
    import re
    s = re.sub(r"t\(1024\) / t\(128\) = [0-9.]+", f"t(1024) / t(128) = {t[1024] / t[128]:.2f}", s, count=1)
    open(path, "w").write(s)
dl = os.path.join(ses, "dense.log")
if os.path.exists(dl):
    new = [l.rstrip("\n") for l in open(dl) if l.startswith("{")]
    names = {json.loads(l)["config"] + json.loads(l)["controller"] for l in new}
    path = os.path.join(P, f"{tag}_configs.jsonl")
    cur = [l for l in (open(path).read().rstrip("\n").split("\n") if os.path.exists(path) else []) if l.strip() and not ("config" in json.loads(l) and json.loads(l)["config"] + json.loads(l).get("controller", "") in names)]
    open(path, "w").write("\n".join(cur + new) + "\n")
print("profiles refreshed:", [(b, round(v), round(ms, 3)) for b, v, ms, _ in rows])
