"""Join a rocprofv3 kernel trace (ROCm 7.2 rocpd sqlite) with bench.py's --shape-table: one row per (conv-engine kernel
instance, layer shape) with the rocprof launch count, the rocprof average duration, the algorithmic GFLOP per launch
and the TF/s that follows -- so every `roofline` figure of a workload can be recomputed from profiles/ alone.

usage: python tools/rocpd_rows.py <results.db> <shape_table.json>

The trace names only the template instance (igemm_lean_kernel<2,128,128>).  bench.py records, for one eager step per
section (plain step / lazy-R1 step), the ORDER of its conv-engine calls with kernel instance, layer shape and workgroup
count (C ABI: contrad_conv2d_grid_blocks).  The trace is cut into steps at the optimizer kernel; a step whose igemm
dispatches agree with a section's sequence in number, instance and workgroup count, one by one, is attributed dispatch
by dispatch (eager steps and hipGraph replays alike -- a replay preserves the stream order).  Steps that agree with no
section (the cold first step of a process, whose plans may differ) are reported as unmatched."""
import collections
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
    k = re.sub(r'\s+', '', k)
    k = re.sub(r'wino(22|23|44n?)?::', '', k)    # (csrc/wino.h / wino22.h live in namespaces; the engine names its kernels without them)
    # igemm_lean_kernel<MODE,BM,BN,false|true>: the 4th argument is the balanced block order of the strided data gradient
    # (round 4); the shape table names the instance by (MODE, BM, BN) only
    k = re.sub(r'^wino23_kernel<\d+>', 'wino23_kernel', k)           # (<raw pieces per mover>)
    k = re.sub(r'^(wino(?:22|23|44n?)_kernel<\d+),\d+>', r'\1>', k)     # (<MODE, raw pieces per mover | raw box width>: named by MODE)
    return re.sub(r'^(igemm_lean_kernel<\d+,\d+,\d+),(?:false|true)>', r'\1>', k)


# kernels the conv engine's C-ABI entry points launch as their MAIN dispatch (contrad_conv2d_path), i.e. the rows of
# bench.py's shape table; their reduce kernels are separate trace rows
CONV_KERNELS = ('igemm', 'wgrad_c32_kernel', 'fwd_k1_kernel', 'conv_c32_kernel', 'wino_kernel', 'wino_wgrad_kernel', 'wino22_kernel', 'wino22_wgrad_kernel', 'wino44_kernel', 'wino44n_kernel', 'wino23_kernel')


def main(db_path, table_path):
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    if 'grid_x' not in cols:
        raise SystemExit('kernel trace without grid sizes: %s' % cols)
    rows = cur.execute("select name, start, end, grid_x, grid_y, grid_z, workgroup_x, workgroup_y, workgroup_z "
                       "from kernels order by start").fetchall()
    # cut into steps at the Adam launch that ends a D-step
    steps, cur_step = [], []
    for name, t0, t1, gx, gy, gz, wx, wy, wz in rows:
        k = norm(short(name))
        if any(n in k for n in CONV_KERNELS):
            threads = gx * max(gy, 1) * max(gz, 1)
            wg = max(wx, 1) * max(wy, 1) * max(wz, 1)
            blocks = threads // wg if threads % wg == 0 and threads >= wg else threads     # (grid in work-items)
            cur_step.append((k, blocks, (t1 - t0) * 1e-3))
        elif k.startswith('adam_kernel') or k.startswith('adam_dev_kernel') or 'adam' in k:
            if cur_step:
                steps.append(cur_step)
            cur_step = []
    table = json.load(open(table_path))
    print('# %s: rocprofv3 kernel trace joined with the shape table of `bench.py --config %s` (per-GPU batch %d)' % (
        db_path.split('/')[-1], table['config'], table['per_gpu_batch']))
    print('# GFLOP/call = the layer\'s NOMINAL (dense) count, FLOP rule: %s; exec = share of it the kernel issues: pixel-major '
          'tiles (contrad_conv2d_path 3, small maps) skip the tap-positions that read zero padding, which the rule counts'
          % table['flop_rule'])
    print('# TF/s = GFLOP/call * exec / rocprof average duration of the igemm dispatch alone (a WGRAD call\'s '
          'wgrad_reduce_kernel is a separate trace row) = the rate the matrix pipe ran at; frac = TF/s / 157.3 (fp32 MFMA peak '
          'at 2.4 GHz); nominalTF/s = GFLOP/call / duration: a speed-up over the dense algorithm, not a utilisation')
    print('# %d steps in the trace (cut at the optimizer launch)' % len(steps))
    matched_steps = set()
    reasons = {}
    reordered = 0
    approx = set()
    for sec, d in table['sections'].items():
        seq = d.get('sequence') or []
        keyseq = [(norm(q[0]), q[2]) for q in seq]
        agg = {}
        nmatch = 0
        # launch orders this section may show: the eager step the table was written from, and (hipGraph replay) the
        # order of the captured step (bench.py records it during the capture)
        orders = [seq] + ([d['graph_sequence']] if d.get('graph_sequence') else [])
        key_orders = [[(norm(q[0]), q[2]) for q in o] for o in orders]
        want = collections.Counter(keyseq)
        slots = collections.defaultdict(list)       # (instance, workgroups) -> positions in the section's sequence, in order
        for pos, key in enumerate(keyseq):
            slots[key].append(pos)
        shapes_of = collections.defaultdict(set)
        for q in seq:
            shapes_of[(norm(q[0]), q[2])].add(tuple(q[1]))
        for si, st in enumerate(steps):
            got_seq = [(a[0], a[1]) for a in st]
            exact = next((oi for oi, ko in enumerate(key_orders) if ko == got_seq), None)
            if exact is not None:                   # dispatch by dispatch, in a recorded order
                nmatch += 1
                matched_steps.add(si)
                for (k, blocks, us), q in zip(st, orders[exact]):
                    a = agg.setdefault((norm(q[0]), tuple(q[1]), q[2]), [0, 0.0, q[3], q[4] if len(q) > 4 else 1.0])
                    a[0] += 1
                    a[1] += us
                continue
            # Same MULTISET of (instance, workgroups) in an order nobody recorded: the i-th dispatch of a key goes to the i-th
            # table entry of that key -- exact where a key belongs to one shape, approximate ('~' in the table) where
            # several shapes share instance AND workgroup count (every split-K WGRAD has ~1008 blocks)
            got = collections.Counter(got_seq)
            if got != want:
                if si not in matched_steps:
                    diff = sorted(((got - want) + (want - got)).items(), key=lambda kv: -kv[1])[:4]
                    reasons.setdefault(si, []).append('%s: %d vs %d launches, differing keys %s' % (
                        sec, len(st), len(keyseq), ['%s x%d: trace %d / table %d' % (k[0], k[1], got[k], want[k]) for k, _ in diff]))
                continue
            nmatch += 1
            matched_steps.add(si)
            reordered += 1
            taken = collections.Counter()
            for (k, blocks, us) in st:
                q = seq[slots[(k, blocks)][taken[(k, blocks)]]]
                taken[(k, blocks)] += 1
                a = agg.setdefault((norm(q[0]), tuple(q[1]), q[2]), [0, 0.0, q[3], q[4] if len(q) > 4 else 1.0])
                a[0] += 1
                a[1] += us
                if len(shapes_of[(k, blocks)]) > 1:
                    approx.add((norm(q[0]), tuple(q[1]), q[2]))
        print('\n== section %s: %d conv-engine launches per step, %d trace step(s) matched ==' % (sec, len(seq), nmatch))
        if not nmatch:
            continue
        # TF/s and frac are on the multiply-adds ISSUED (GFLOP/call * exec); the last column prices the same duration on
        # the layer's nominal (dense) count and is labelled as such -- it is not a utilisation and may exceed the peak
        print('%-30s %-34s %7s %7s %8s %10s %11s %5s %7s %6s %12s' % ('kernel', 'shape N,H,W,C,K,KH,KW,s,p', 'blocks', 'n/step',
                                                                   'calls', 'avg_us', 'GFLOP/call', 'exec', 'TF/s', 'frac',
                                                                   'nominalTF/s'))
        tot_f = tot_t = tot_x = 0.0
        per_kernel = {}
        for (k, shape, blocks), (c, us, gf, ex) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            avg = us / c
            tf = gf / avg * 1e3 if avg > 0 else 0.0
            tot_f += gf * c / nmatch
            tot_x += gf * ex * c / nmatch
            tot_t += us / nmatch
            pk = per_kernel.setdefault(k, [0.0, 0.0, 0, 0.0])
            pk[0] += gf * c / nmatch; pk[1] += us / nmatch; pk[2] += c / nmatch; pk[3] += gf * ex * c / nmatch
            xf_ = tf * ex / 157.3
            # an issued fraction above 1 is impossible: it only appears on '~' rows whose durations were swapped among
            # shapes sharing instance and workgroup count -- printed as 'ambig' instead of a number
            print('%-30s %-34s %7d %7.1f %8d %10.2f %11.4f %5.2f %7.1f %6s %12.1f%s' % (
                k, ','.join(map(str, shape)), blocks, c / nmatch, c, avg, gf, ex, tf * ex,
                ('%.3f' % xf_) if xf_ <= 1.0 else 'ambig', tf,
                ' ~' if (k, shape, blocks) in approx else ''))
        print('-- per kernel instance (this section, per step):')
        for k, (gf, us, n, xf) in sorted(per_kernel.items(), key=lambda kv: -kv[1][1]):
            print('   %-30s %5.1f launches %9.1f us   issued %10.2f GFLOP -> %6.1f TF/s (%.3f of 157.3)   nominal %10.2f GFLOP -> %6.1f TF/s (nominal %.3f)' % (
                k, n, us, xf, xf / us * 1e3 if us else 0, xf / us * 1e3 / 157.3 if us else 0,
                gf, gf / us * 1e3 if us else 0, gf / us * 1e3 / 157.3 if us else 0))
        print('-- conv engine, %s: issued %.1f GFLOP per step in %.1f us of igemm dispatches -> %.1f TF/s (%.3f of 157.3); nominal '
              '%.1f GFLOP -> %.1f TF/s (nominal %.3f)' % (sec, tot_x, tot_t, tot_x / tot_t * 1e3, tot_x / tot_t * 1e3 / 157.3,
                                                        tot_f, tot_f / tot_t * 1e3, tot_f / tot_t * 1e3 / 157.3))
    if reordered:
        print('\n(%d matched step(s) launched their kernels in an order that was not recorded; rows marked ~ share instance and '
              'workgroup count with another shape, their durations may be swapped among those shapes)' % reordered)
    un = [i for i in range(len(steps)) if i not in matched_steps]
    if un:
        print('\n== %d trace step(s) matched no section: ==' % len(un))
        for i in un[:8]:
            print('   step %d (%d conv-engine launches): %s' % (i, len(steps[i]), ' | '.join(reasons.get(i, ['-']))))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
