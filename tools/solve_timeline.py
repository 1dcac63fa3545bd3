"""Developer tool: the launches of ONE complete solve, in order, from a rocprofv3 kernel trace of tools/solve128.py (the sixth
solve: the solver object is warm): start, duration and the gap to the previous launch.
usage: python tools/solve_timeline.py KERNEL_TRACE_CSV [SOLVE_INDEX]"""
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    which = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    starts = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("bounds_time_major")]
    a = starts[which]
    b = starts[which + 1] if which + 1 < len(starts) else len(rows)
    t0 = int(rows[a]["Start_Timestamp"])
    prev = t0
    n = 0
    for r in rows[a:b]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if s - prev > 5_000_000:  # (the next solve's first copy: outside this solve)
            break
        print(f"{(s - t0) / 1e3:9.1f} us  {(e - s) / 1e3:7.1f} us  gap {(s - prev) / 1e3:6.1f}  {r['Kernel_Name'][:90]}")
        prev = e
        n += 1
    print(f"{n} launches, {(prev - t0) / 1e3:.1f} us from the first launch's start to the last one's end; "
          f"{len(rows) / max(len(starts), 1):.1f} launches per solve over all {len(starts)} solves of the trace (cold ones included)")


if __name__ == "__main__":
    main()
