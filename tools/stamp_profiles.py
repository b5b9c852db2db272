#!/usr/bin/env python
"""tools/stamp_profiles.py <tag> -- copy the judged artefacts of `tools/collect_profiles.sh <tag>` from gpurun_out/ (scratch) to
profiles/ (tracked).  The GPU box has no .git, so the JSON files that record a `git_head` get it stamped here: the snapshot a
gpurun call ships is the working tree, i.e. HEAD when `git status` is clean (asserted)."""
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
head = sys.argv[2] if len(sys.argv) > 2 else None
if head is None:
    dirty = subprocess.run(["git", "status", "--porcelain", "--untracked-files=no"], cwd=ROOT, stdout=subprocess.PIPE, text=True).stdout.strip()
    assert not dirty, "working tree differs from HEAD: pass the commit the snapshot was taken at"
    head = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, stdout=subprocess.PIPE, text=True).stdout.strip()
src, dst = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
names = [n for n in sorted(os.listdir(src)) if n.startswith(tag + "_") and os.path.isfile(os.path.join(src, n))
         and not n.endswith((".err", ".log"))]
for n in names:
    if n.endswith(".json"):
        d = json.load(open(os.path.join(src, n)))
        if isinstance(d, dict) and d.get("git_head", None) in ("working tree", ""):
            d["git_head"] = head
        if isinstance(d, dict) and "roofline" in d and isinstance(d["roofline"].get("traffic_source"), str):
            d["roofline"]["traffic_source"] = d["roofline"]["traffic_source"].replace("collected on working tree", f"collected on {head}")
        if isinstance(d, dict) and "metric" in d:
            d["collected_on_commit"] = head
        json.dump(d, open(os.path.join(dst, n), "w"), indent=None if "metric" in d else 1)
        if "metric" in d:
            open(os.path.join(dst, n), "a").write("\n")
    else:
        shutil.copyfile(os.path.join(src, n), os.path.join(dst, n))
    print(n)
