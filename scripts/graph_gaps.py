#!/usr/bin/env python3
"""Inter-kernel gaps of the captured forward: reads a rocprofv3 --kernel-trace CSV, takes the dispatches of the last graph replay
(the last run of N kernels starting at the stem kernel) and prints busy time, span and the gap histogram."""
import csv
import glob
import sys

root = sys.argv[1]
rows = []
for f in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
starts = [i for i, r in enumerate(rows) if "conv_stem_kernel" in r[2]]
if len(starts) < 3:
    sys.exit("no forward found")
a, b = starts[-2], starts[-1]          # the second-to-last forward: complete by construction
fw = rows[a:b]
busy = sum(e - s for s, e, _ in fw)
span = fw[-1][1] - fw[0][0]
gaps = [fw[i + 1][0] - fw[i][1] for i in range(len(fw) - 1)]
print(f"kernels {len(fw)}  busy {busy / 1e3:.1f} us  span {span / 1e3:.1f} us  gaps total {sum(gaps) / 1e3:.1f} us  "
      f"mean {sum(gaps) / len(gaps) / 1e3:.2f} us  max {max(gaps) / 1e3:.2f} us  next forward starts {(rows[b][0] - fw[-1][1]) / 1e3:.2f} us later")
worst = sorted(range(len(gaps)), key=lambda i: -gaps[i])[:8]
for i in worst:
    print(f"  gap {gaps[i] / 1e3:6.2f} us after {fw[i][2][:60]:60s} ({(fw[i][1] - fw[i][0]) / 1e3:.1f} us)")
