"""Join a rocprofv3 kernel trace (ROCm 7.2 rocpd sqlite) with bench.py's --shape-table: one row per (conv-engine kernel
instance, layer shape) with the rocprof launch count, the rocprof average duration, the algorithmic GFLOP per launch
and the TF/s that follows -- so every `roofline` figure of a workload can be recomputed from profiles/ alone.

usage: python tools/rocpd_rows.py <results.db> <shape_table.json> [--steps K]

The trace names only the template instance (igemm_lean_kernel<2,128,128>); the join key is (instance, workgroups of the
launch): bench.py records the workgroup count of every conv call from the C ABI (contrad_conv2d_grid_blocks).  Shapes
that share a key are merged into one row (their FLOPs and launch counts add up).  Dispatches of an instance whose grid
matches no row of the table are listed as 'unmatched' (e.g. the generator forward's transposed convs)."""
import json
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'void ', '', name)
    m = re.match(r'([\w:<>, ]+?)\(', name)
    return (m.group(1) if m else name)[:90]


def norm(k):
    return re.sub(r'\s+', '', k)


def main(db_path, table_path, steps=None):
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    if 'grid_x' not in cols:
        raise SystemExit('kernel trace without grid sizes: %s' % cols)
    rows = cur.execute("select name, start, end, grid_x, grid_y, grid_z, workgroup_x, workgroup_y, workgroup_z "
                       "from kernels order by start").fetchall()
    disp = {}
    for name, t0, t1, gx, gy, gz, wx, wy, wz in rows:
        k = norm(short(name))
        if 'igemm' not in k:
            continue
        threads = gx * max(gy, 1) * max(gz, 1)
        wg = max(wx, 1) * max(wy, 1) * max(wz, 1)
        blocks = threads // wg if threads % wg == 0 and threads >= wg else threads     # (grid in work-items)
        a = disp.setdefault((k, blocks), [0, 0.0])
        a[0] += 1
        a[1] += (t1 - t0) * 1e-3
    table = json.load(open(table_path))
    print('# %s: rocprofv3 kernel trace joined with the shape table of `bench.py --config %s` (per-GPU batch %d)' % (
        db_path.split('/')[-1], table['config'], table['per_gpu_batch']))
    print('# TF/s = GFLOP per launch / rocprof average duration of the igemm dispatch alone (a WGRAD call\'s '
          'wgrad_reduce_kernel is a separate trace row); FLOP rule: %s' % table['flop_rule'])
    used = set()
    for sec, d in table['sections'].items():
        merged = {}
        for r in d['rows']:
            key = (norm(r['kernel']), r['grid_blocks'])
            m = merged.setdefault(key, {'shapes': [], 'lps': 0.0, 'gflop_step': 0.0, 'bracket_us_step': 0.0})
            m['shapes'].append(r['shape'])
            m['lps'] += r['launches_per_step']
            m['gflop_step'] += r['gflop_per_launch'] * r['launches_per_step']
            m['bracket_us_step'] += r['bracket_us'] * r['launches_per_step']
        print('\n== section %s (%d eager step(s) sampled for the table) ==' % (sec, d['steps_sampled']))
        print('%-34s %8s %7s %9s %10s %10s %8s %8s  %s' % ('kernel', 'blocks', 'n/step', 'calls', 'avg_us', 'GFLOP/call',
                                                          'TF/s', 'frac', 'shape(s) N,H,W,C,K,KH,KW,s,p'))
        tot_f = tot_t = 0.0
        for (k, blocks), m in sorted(merged.items(), key=lambda kv: -kv[1]['gflop_step']):
            dsp = disp.get((k, blocks))
            gpl = m['gflop_step'] / m['lps']
            if dsp is None:
                print('%-34s %8d %7.1f %9s %10s %10.3f %8s %8s  %s' % (k, blocks, m['lps'], '-', '-', gpl, '-', '-', m['shapes']))
                continue
            used.add((k, blocks))
            avg = dsp[1] / dsp[0]
            tf = gpl / avg * 1e-3
            tot_f += m['gflop_step']
            tot_t += avg * m['lps']
            print('%-34s %8d %7.1f %9d %10.2f %10.3f %8.1f %8.3f  %s' % (k, blocks, m['lps'], dsp[0], avg, gpl, tf, tf / 157.3,
                                                                        ' '.join(','.join(map(str, s)) for s in m['shapes'])))
        if tot_t > 0:
            print('-- conv engine, this section: %.1f GFLOP per step in %.1f us of igemm dispatches -> %.1f TF/s (%.3f of 157.3)'
                  % (tot_f, tot_t, tot_f / tot_t * 1e-3, tot_f / tot_t * 1e-3 / 157.3))
    rest = [(k, v) for k, v in disp.items() if k not in used]
    if rest:
        print('\n== igemm dispatches not in the table (generator forward, first cold step, ...) ==')
        for (k, blocks), (c, t) in sorted(rest, key=lambda kv: -kv[1][1]):
            print('%-34s %8d calls %6d  avg_us %10.2f' % (k, blocks, c, t / c))


if __name__ == '__main__':
    st = None
    if '--steps' in sys.argv:
        st = int(sys.argv[sys.argv.index('--steps') + 1])
    main(sys.argv[1], sys.argv[2], st)
