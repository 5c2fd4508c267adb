#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output for one kernel-name substring.

usage: pmc_summary.py <dir> <kernel-substring>
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
  out = {"kernel": needle}
  for path in sorted(glob.glob(os.path.join(root, "**", "*.csv"), recursive=True)):
    with open(path, newline="") as f:
      rows = list(csv.DictReader(f))
    if not rows:
      continue
    cols = rows[0].keys()
    if "Start_Timestamp" in cols and "Kernel_Name" in cols and "Counter_Name" not in cols:
      d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows if needle in r["Kernel_Name"]]
      if d:
        out.setdefault("trace", {})[os.path.relpath(path, root)] = {
          "calls": len(d), "avg_ns": sum(d) / len(d), "min_ns": min(d), "max_ns": max(d)}
    if "Counter_Name" in cols and "Counter_Value" in cols:
      acc = {}
      for r in rows:
        if needle in r.get("Kernel_Name", ""):
          acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
      for k, v in acc.items():
        out.setdefault("pmc", {})[k] = {"dispatches": len(v), "avg": sum(v) / len(v)}
  print(json.dumps(out, indent=1))


if __name__ == "__main__":
  main()
