"""Summarise a rocprofv3 (ROCm 7.2 rocpd sqlite) kernel trace: per-kernel count / total / avg / share, and
optionally the per-dispatch timeline of the last step.  Dev tool; writes the text kept under profiles/."""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'void ', '', name)
    m = re.match(r'([\w:<>, ]+?)\(', name)
    return (m.group(1) if m else name)[:90]


def main(path, timeline=False):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    rows = cur.execute("select name, start, end, grid_x, grid_y, grid_z, workgroup_x from kernels order by start").fetchall() \
        if 'grid_x' in cols else cur.execute("select name, start, end from kernels order by start").fetchall()
    agg = {}
    for r in rows:
        a = agg.setdefault(short(r[0]), [0, 0.0])
        a[0] += 1
        a[1] += (r[2] - r[1]) * 1e-3
    tot = sum(a[1] for a in agg.values())
    print('%-92s %7s %12s %10s %6s' % ('kernel', 'calls', 'total_us', 'avg_us', '%'))
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print('%-92s %7d %12.1f %10.2f %6.2f' % (k, c, t, t / c, 100 * t / tot))
    print('TOTAL kernel time %.1f us over %d dispatches; wall span %.1f us' % (tot, len(rows), (rows[-1][2] - rows[0][1]) * 1e-3))
    if timeline:
        # last Adam kernel marks a step boundary: print the dispatches of the last full step
        idx = [i for i, r in enumerate(rows) if 'adam_kernel' in r[0]]
        if len(idx) >= 2:
            seg = rows[idx[-2] + 1: idx[-1] + 1]
            t0 = seg[0][1]
            print('\nlast step timeline (%d dispatches, %.1f us):' % (len(seg), (seg[-1][2] - t0) * 1e-3))
            prev_end = t0
            for r in seg:
                print('  +%9.1f us  dur %9.1f  gap %7.1f  grid %s  %s' % ((r[1] - t0) * 1e-3, (r[2] - r[1]) * 1e-3,
                      (r[1] - prev_end) * 1e-3, tuple(r[3:6]) if len(r) > 3 else '', short(r[0])))
                prev_end = r[2]


if __name__ == '__main__':
    main(sys.argv[1], '--timeline' in sys.argv)
