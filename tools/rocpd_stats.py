"""Summarise a rocprofv3 rocpd SQLite database (kernel-trace) into a per-kernel stats table
(the same numbers `--stats` prints): calls, total / average / min / max duration, share."""
import re
import sqlite3
import subprocess
import sys


def main(path, out=None):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tables = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    disp = [t for t in tables if t.startswith("rocpd_kernel_dispatch")][0]
    sym = [t for t in tables if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in cur.execute("pragma table_info(%s)" % disp)]
    scol = [r[1] for r in cur.execute("pragma table_info(%s)" % sym)]
    name_col = "display_name" if "display_name" in scol else ("kernel_name" if "kernel_name" in scol else scol[-1])
    q = "select s.%s, count(*), sum(d.end - d.start), min(d.end - d.start), max(d.end - d.start) from %s d join %s s on d.kernel_id = s.id group by s.%s order by 3 desc" % (
        name_col, disp, sym, name_col)
    rows = list(cur.execute(q))
    total = sum(r[2] for r in rows)
    lines = ["%-8s %-12s %-10s %-10s %-10s %-7s %s" % ("calls", "total_us", "avg_us", "min_us", "max_us", "pct", "kernel")]
    for name, n, tot, mn, mx in rows:
        short = re.sub(r"\(anonymous namespace\)::", "", name)
        short = re.sub(r"\[clone .*\]", "", short)[:150]
        lines.append("%-8d %-12.1f %-10.2f %-10.2f %-10.2f %-7.2f %s" % (n, tot / 1e3, tot / n / 1e3, mn / 1e3, mx / 1e3, 100.0 * tot / total, short))
    lines.append("TOTAL kernel time %.1f us over %d dispatches" % (total / 1e3, sum(r[1] for r in rows)))
    text = "\n".join(lines)
    print(text)
    if out:
        open(out, "w").write(text + "\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
