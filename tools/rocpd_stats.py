"""Summarise a rocprofv3 rocpd SQLite database (kernel-trace) into a per-kernel
stats table (calls, total/avg/min/max us) and, with --timeline N, the last N
dispatches with start offsets -- the text that gets committed under profiles/."""
import re
import sqlite3
import sys


def main():
    db = sys.argv[1]
    timeline = int(sys.argv[sys.argv.index("--timeline") + 1]) if "--timeline" in sys.argv else 0
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "kernel_symbol" in t][0]
    rows = c.execute(f"select s.kernel_name, d.start, d.end, d.grid_size_x, d.workgroup_size_x, "
                     f"s.arch_vgpr_count, s.accum_vgpr_count, d.group_segment_size from `{kd}` d "
                     f"join `{ks}` s on d.kernel_id = s.id order by d.start").fetchall()
    short = lambda n: re.sub(r"\(.*", "", n)[:90]
    agg = {}
    for name, st, en, gx, wx, vg, ag, lds in rows:
        a = agg.setdefault(short(name), [0, 0.0, 1e30, 0.0, gx, wx, vg, ag, lds])
        d = (en - st) / 1e3
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    print(f"{'kernel':<92}{'calls':>7}{'total_us':>12}{'avg_us':>10}{'min_us':>10}{'max_us':>10}{'%':>7}"
          f"{'grid':>9}{'wg':>5}{'vgpr':>6}{'agpr':>6}{'lds':>8}")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{k:<92}{a[0]:>7}{a[1]:>12.1f}{a[1]/a[0]:>10.2f}{a[2]:>10.2f}{a[3]:>10.2f}{100*a[1]/tot:>7.1f}"
              f"{a[4]:>9}{a[5]:>5}{a[6]:>6}{a[7]:>6}{a[8]:>8}")
    if timeline:
        print("\nlast dispatches (start offset us, duration us, gap to previous end us):")
        sel = rows[-timeline:]
        t0 = sel[0][1]
        prev = None
        for name, st, en, *_ in sel:
            gap = (st - prev) / 1e3 if prev else 0.0
            print(f"  {((st - t0) / 1e3):>10.2f} {((en - st) / 1e3):>9.2f} {gap:>8.2f}  {short(name)}")
            prev = en


if __name__ == "__main__":
    main()
