"""Where a launch-chain-bound step spends its time: kernel durations against the gaps between dependent launches.

Reads the kernel trace of `rocprofv3 --kernel-trace --output-format csv -- python3 bench.py --batch 32 ...` and, over the
last --steps steps of the trace (steady state; a step is --launches launches), prints the busy time, the idle time between
the end of a kernel and the start of the next one on the device, and both per kernel name.

   python tools/launch_gaps.py gpurun_out/<tag>/**/p_kernel_trace.csv --launches 130 --steps 100
"""
import argparse
import collections
import csv

ap = argparse.ArgumentParser()
ap.add_argument("csv")
ap.add_argument("--launches", type=int, required=True, help="launches per step")
ap.add_argument("--steps", type=int, default=100)
ap.add_argument("--top", type=int, default=25)
a = ap.parse_args()

rows = list(csv.DictReader(open(a.csv)))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda e: e[0])
ev = ev[-a.launches * a.steps:]
busy = sum(e - s for s, e, _ in ev)
wall = ev[-1][1] - ev[0][0]
gap_after = collections.defaultdict(lambda: [0, 0, 0])   # name -> [launches, kernel ns, gap to the next launch ns]
prev_end = None
gaps = []
for i, (s, e, n) in enumerate(ev):
    n = n.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0][:60]
    g = gap_after[n]
    g[0] += 1
    g[1] += e - s
    if i + 1 < len(ev):
        gap = max(0, ev[i + 1][0] - e)
        g[2] += gap
        gaps.append(gap)
n_steps = len(ev) / a.launches
print(f"{len(ev)} launches = {n_steps:.1f} steps: wall {wall / n_steps / 1e3:.1f} us/step, kernels {busy / n_steps / 1e3:.1f} us/step, "
      f"idle between launches {sum(gaps) / n_steps / 1e3:.1f} us/step ({sum(gaps) / len(gaps) / 1e3:.2f} us per boundary, "
      f"median {sorted(gaps)[len(gaps) // 2] / 1e3:.2f})")
print(f"{'kernel':60s} {'x/step':>7s} {'avg us':>8s} {'gap after us':>13s} {'us/step (kernel + gap)':>24s}")
for n, (c, k, g) in sorted(gap_after.items(), key=lambda kv: -(kv[1][1] + kv[1][2]))[:a.top]:
    print(f"{n:60s} {c / n_steps:7.1f} {k / c / 1e3:8.2f} {g / c / 1e3:13.2f} {(k + g) / n_steps / 1e3:24.1f}")
