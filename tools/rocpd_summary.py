"""Per-kernel totals of a rocprofv3 --kernel-trace run that wrote a rocpd database (`*_results.db`, the default output format of this
ROCm): name, launches, total and mean duration, share -- the table rocprofv3 --stats prints, from the file.

    python tools/rocpd_summary.py gpurun_out/f32prof/gen_results.db [top=12]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    rows = db.execute("select name, count(*), sum(end - start) / 1e3, avg(end - start) / 1e3 from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    print(f"{sys.argv[1]}: {len(rows)} kernels, {tot / 1e3:.2f} ms of kernel time")
    for r in rows[:top]:
        print(f"  {r[0][:96]:96s} n={r[1]:5d} total {r[2] / 1e3:9.2f} ms ({100 * r[2] / tot:4.1f} %) mean {r[3]:9.1f} us")


if __name__ == "__main__":
    main()
