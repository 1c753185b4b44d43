"""Per-kernel PMC counter means from a rocprofv3 rocpd sqlite db (counters summed over XCDs/SEs per dispatch)."""
import re, sqlite3, sys
from collections import defaultdict


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'void ', '', name)
    m = re.match(r'([\w:<>, ]+?)\(', name)
    return (m.group(1) if m else name)[:70]


def main(path, filt='igemm'):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    # expected columns: dispatch_id, kernel name, counter_name, value ...
    namecol = 'kernel_name' if 'kernel_name' in cols else ('name' if 'name' in cols else None)
    q = "select dispatch_id, %s, counter_name, sum(value), max(end - start) from counters_collection group by dispatch_id, counter_name" % namecol
    per = defaultdict(lambda: defaultdict(list))
    dur = defaultdict(list)
    for did, kn, cn, val, d in cur.execute(q):
        k = short(kn)
        if filt and filt not in k:
            continue
        per[k][cn].append(val)
        if cn == list(per[k].keys())[0]:
            dur[k].append(d)
    for k in sorted(per):
        print(k, ' dispatches=%d  avg_dur_us=%.1f' % (len(dur[k]), sum(dur[k]) / max(len(dur[k]), 1) * 1e-3))
        for cn, vals in sorted(per[k].items()):
            print('    %-34s %16.1f' % (cn, sum(vals) / len(vals)))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else 'igemm')
