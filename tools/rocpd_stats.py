"""Summarise a rocprofv3 rocpd SQLite result (kernel-trace) into a per-kernel stats table (markdown).
usage: python tools/rocpd_stats.py results.db [out.md]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"void ", "", name)
    m = re.match(r"([\w:]+(<[^(]*>)?)", name)
    return (m.group(1) if m else name)[:110]


def main(db, out=None):
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name "
                     "order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows)
    lines = ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---:|---:|---:|---:|---:|---:|"]
    for name, n, tot, avg, mn, mx in rows[:45]:
        lines.append(f"| `{short(name)}` | {n} | {tot / 1e6:.2f} | {avg / 1e3:.1f} | {mn / 1e3:.1f} | {mx / 1e3:.1f} | {100 * tot / total:.1f} |")
    lines.append(f"\ntotal kernel time {total / 1e6:.1f} ms over {sum(r[1] for r in rows)} dispatches")
    text = "\n".join(lines)
    print(text)
    if out:
        open(out, "w").write(text + "\n")


if __name__ == "__main__":
    main(*sys.argv[1:3])
