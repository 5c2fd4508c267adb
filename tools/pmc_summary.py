#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output for one kernel-name substring.

usage: pmc_summary.py <dir> <kernel-substring> [min-grid-size]
Prints JSON: average duration (kernel trace) and per-dispatch average of every PMC counter
found under <dir> (recursively) for dispatches whose kernel name contains the substring.
"""
import csv
import glob
import json
import os
import sys


def main():
  root, needle = sys.argv[1], sys.argv[2]
  min_grid = int(sys.argv[3]) if len(sys.argv) > 3 else 0
  out = {"kernel": needle, "min_grid": min_grid}

  def keep(r):
    grid = r.get("Grid_Size") or r.get("Grid_Size_X") or min_grid
    return needle in r.get("Kernel_Name", "") and int(grid) >= min_grid

  for path in sorted(glob.glob(os.path.join(root, "**", "*.csv"), recursive=True)):
    with open(path, newline="") as f:
      rows = list(csv.DictReader(f))
    if not rows:
      continue
    cols = rows[0].keys()
    if "Start_Timestamp" in cols and "Kernel_Name" in cols and "Counter_Name" not in cols:
      d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows if keep(r)]
      if d:
        entry = {"calls": len(d), "avg_ns": sum(d) / len(d), "min_ns": min(d), "max_ns": max(d)}
        split = float(os.environ.get("PMC_SUMMARY_SPLIT_NS", "0"))
        if split > 0:  # launches of two shapes under one name (bench.py --steps 20: clock warm-up vs timed)
          for name, sel in (("shorter", [x for x in d if x < split]), ("longer", [x for x in d if x >= split])):
            if sel:
              entry[name + f"_than_{int(split)}ns"] = {"calls": len(sel), "avg_ns": sum(sel) / len(sel), "median_ns": sorted(sel)[len(sel) // 2], "min_ns": min(sel), "max_ns": max(sel)}
        out.setdefault("trace", {})[os.path.relpath(path, root)] = entry
    if "Counter_Name" in cols and "Counter_Value" in cols:
      acc = {}
      for r in rows:
        if keep(r):
          acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
      for k, v in acc.items():
        entry = {"dispatches": len(v), "avg": sum(v) / len(v)}
        # launches of two shapes under one name (bench.py --steps 20: T=128 clock warm-up vs the
        # timed 20-step launches): counters of the smaller shape are below this fraction of the maximum
        frac = float(os.environ.get("PMC_SUMMARY_SPLIT_FRAC", "0"))
        if frac > 0:
          cut = frac * max(v)
          for name, sel in (("smaller", [x for x in v if x < cut]), ("larger", [x for x in v if x >= cut])):
            if sel:
              srt = sorted(sel)  # the median is the figure of the group's usual shape (the 20-step launches;
              # the one 5-step --warmup launch among them moves the mean, not the median)
              entry[name] = {"dispatches": len(sel), "avg": sum(sel) / len(sel), "median": srt[len(srt) // 2], "min": srt[0], "max": srt[-1]}
        out.setdefault("pmc", {})[k] = entry
  print(json.dumps(out, indent=1))


if __name__ == "__main__":
  main()
