"""Per-step table of a rocprofv3 kernel_stats.csv of `bench.py --steps K --warmup W` (3 timed blocks: 3K + W steps in all).
   python tools/kstats.py profiles/r04a_bench_bs512_kernel_stats.csv --steps 33 [--top 40]"""
import argparse, csv
ap = argparse.ArgumentParser()
ap.add_argument("csv"); ap.add_argument("--steps", type=int, required=True); ap.add_argument("--top", type=int, default=40)
a = ap.parse_args()
rows = list(csv.DictReader(open(a.csv)))
tot = sum(int(r['TotalDurationNs']) for r in rows)
print(f"sum of kernel time {tot / a.steps / 1e6:.3f} ms/step, {sum(int(r['Calls']) for r in rows) / a.steps:.0f} launches/step")
for r in rows[:a.top]:
    n = r['Name'].replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '')
    n = n.split('(')[0][:64]
    print(f"{n:64s} x{int(r['Calls']) / a.steps:5.1f}  avg {float(r['AverageNs']) / 1e3:8.1f} us  {int(r['TotalDurationNs']) / a.steps / 1e6:6.3f} ms/step  {100 * int(r['TotalDurationNs']) / tot:5.1f}%")
