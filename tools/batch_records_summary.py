"""Developer tool: what the records of a compute-trajectories-batch run say about its solves -- which ADMM pipeline ran the
QPs, how many persistent launches gave up, and the distribution of the per-solve wall times (stragglers).
usage: python tools/batch_records_summary.py RESULTS_DIR [WORLD_SIZE]"""
import collections
import glob
import json
import os
import sys

import numpy as np


def main():
    d = sys.argv[1]
    path = sorted(glob.glob(os.path.join(d, "scp_benchmark_*.json")), key=os.path.getmtime)[-1]
    recs = json.load(open(path))
    if isinstance(recs, dict):
        recs = recs.get("runs", [])
    ok = [r for r in recs if r.get("status") == "success"]
    t = np.array([r["time_sec"] for r in ok]) * 1e3
    pipes = collections.Counter(p for r in ok for p in r.get("qp_pipeline", []))
    print(f"{path}: {len(recs)} records, {len(ok)} ok; solve wall ms: p10 {np.percentile(t, 10):.2f} p50 {np.percentile(t, 50):.2f} "
          f"p90 {np.percentile(t, 90):.2f} p99 {np.percentile(t, 99):.2f} max {t.max():.2f}")
    print(f"  pipelines: {dict(pipes)}; persistent launches that gave up: {sum(r.get('persist_gave_up', 0) for r in ok)}")
    world = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    for rk in range(world):
        mine = [r for r in ok if r.get("trial_index", 0) % world == rk]
        tt = np.array([r["time_sec"] for r in mine]) * 1e3
        long_qps = sum(1 for r in mine for q in r.get("qp_iterations", []) if q >= 1000)
        print(f"  rank {rk}: {len(mine)} solves, wall ms p50 {np.percentile(tt, 50):.2f} p90 {np.percentile(tt, 90):.2f} max {tt.max():.2f}, "
              f"sum {tt.sum():.0f} ms, QPs with >= 1000 ADMM steps: {long_qps}")
    slow = sorted(ok, key=lambda r: -r["time_sec"])[:5]
    for r in slow:
        print(f"  slow: {r['time_sec'] * 1e3:.1f} ms seed {r.get('seed')} pipelines {r.get('qp_pipeline')} gave_up {r.get('persist_gave_up')} "
              f"qp iterations {r.get('qp_iterations')}")


if __name__ == "__main__":
    main()
