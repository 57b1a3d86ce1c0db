"""Turns a rocprofv3 (--kernel-trace --stats) rocpd database into a small per-kernel text summary for profiles/.
Usage: python scripts/rocprof_summary.py gpurun_out/prof_x/x_results.db profiles/r01_x_kernel_stats.txt "<command>" """
import sqlite3
import sys


def main():
    db, out, cmd = sys.argv[1], sys.argv[2], (sys.argv[3] if len(sys.argv) > 3 else "")
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, count(*), avg(end-start), sum(end-start), min(end-start), max(end-start), "
                       "max(vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size), max(grid_x), max(workgroup_x) "
                       "from kernels group by name order by 4 desc").fetchall()
    tot = sum(r[3] for r in rows)
    with open(out, "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --stats  -- {cmd}\n")
        f.write("# per-kernel summary (durations in microseconds)\n")
        f.write("%-92s %6s %10s %10s %10s %6s %5s %5s %7s %8s %9s\n" % ("kernel", "calls", "avg_us", "min_us", "max_us", "pct", "vgpr", "sgpr", "lds_B", "scratch", "grid"))
        for r in rows:
            f.write("%-92s %6d %10.1f %10.1f %10.1f %6.2f %5d %5d %7d %8d %9d\n" % (
                r[0][:92], r[1], r[2] / 1e3, r[4] / 1e3, r[5] / 1e3, 100.0 * r[3] / tot, r[6] or 0, r[7] or 0, r[8] or 0,
                r[9] or 0, r[10] or 0))
    print(open(out).read()[:1800])


if __name__ == "__main__":
    main()
