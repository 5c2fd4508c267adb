#!/usr/bin/env python3
"""From a rocprofv3 kernel trace of tools/mp_profile.py: per multi-player send, the histogram
kernel's duration, the gap to the fused kernel behind it (the kernel boundary a one-launch version
would trade for a grid-wide wait) and the fused kernel's duration. usage: mp_gap.py <trace dir>"""
import csv
import glob
import json
import os
import sys


def main():
  root = sys.argv[1]
  rows = []
  for path in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
    with open(path, newline="") as f:
      rows += [r for r in csv.DictReader(f) if "mp_" in r.get("Kernel_Name", "")]
  rows.sort(key=lambda r: int(r["Start_Timestamp"]))
  cnt, gap, fus = [], [], []
  for a, b in zip(rows, rows[1:]):
    if "mp_count_rows" in a["Kernel_Name"] and "mp_step_fused" in b["Kernel_Name"]:
      cnt.append(int(a["End_Timestamp"]) - int(a["Start_Timestamp"]))
      gap.append(int(b["Start_Timestamp"]) - int(a["End_Timestamp"]))
      fus.append(int(b["End_Timestamp"]) - int(b["Start_Timestamp"]))
  med = lambda v: sorted(v)[len(v) // 2] / 1e3 if v else None  # noqa: E731
  print(json.dumps({"sends": len(cnt), "mp_count_rows_us_median": med(cnt), "gap_count_to_fused_us_median": med(gap),
                    "mp_step_fused_us_median": med(fus),
                    "from_histogram_start_to_fused_end_us_median": med([c + g + f for c, g, f in zip(cnt, gap, fus)])}, indent=1))


if __name__ == "__main__":
  main()
