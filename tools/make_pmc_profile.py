"""Turn the rocprofv3 --pmc passes of tools/prof_pmc.sh into profiles/<prefix>_pmc.{txt,json}.
usage: python tools/make_pmc_profile.py <pass dir> <out prefix, e.g. r02_bench_n1> "<profiled command>"
Passes expected in <pass dir>: fetch, write, sq, misc [, tcc: TCC_HIT_sum / TCC_MISS_sum -> l2_hit_rate] (one rocprofv3 run each; FETCH_SIZE and WRITE_SIZE never share a
pass, no sys/hip/memory-copy trace domains beside the counters).
"""
import io, json, os, sys
from contextlib import redirect_stdout
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import rocpd_pmc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def table(db):
    buf = io.StringIO()
    with redirect_stdout(buf):
        rocpd_pmc.main(db, '')
    return buf.getvalue()


def parse(text):
    out, cur = {}, None
    for line in text.splitlines():
        if not line.startswith(' '):
            if '  dispatches=' not in line:
                cur = None
                continue
            name, rest = line.split('  dispatches=')
            n, dur = rest.split('  avg_dur_us=')
            cur = out.setdefault(name.strip(), {'dispatches': int(n), 'avg_dur_us': float(dur)})
        elif cur is not None:
            k, v = line.split()
            cur[k] = float(v)
    return out


def main(d, prefix, command):
    txt = ['# rocprofv3 --pmc passes on: %s  (1x MI355X)' % command,
           '# per-dispatch means; counters summed over XCDs/SEs; FETCH_SIZE/WRITE_SIZE in KiB (FETCH_SIZE under-reports '
           'wide coalesced reads by 2x on gfx950, MI355X_MICROARCH.md HBM section)']
    parsed = {}
    passes = ['fetch', 'write', 'sq', 'misc'] + (['tcc'] if os.path.exists(os.path.join(d, 'tcc_results.db')) else [])
    for p in passes:
        t = table(os.path.join(d, p + '_results.db'))
        t = '\n'.join(l for l in t.splitlines() if not l.startswith('at::') or True)
        txt.append('== pass: ' + p)
        txt.append(t.rstrip())
        parsed[p] = parse(t)
    open(os.path.join(ROOT, 'profiles', prefix + '_pmc.txt'), 'w').write('\n'.join(txt) + '\n')

    def entry(kernel):
        f, w, sq, m = (parsed[p][kernel] for p in ('fetch', 'write', 'sq', 'misc'))
        cyc = m['GRBM_GUI_ACTIVE'] / 8.0                       # summed over the 8 XCDs
        traffic = (2 * f['FETCH_SIZE'] + w['WRITE_SIZE']) * 1024
        tcc = parsed.get('tcc', {}).get(kernel, {})
        hit, miss = tcc.get('TCC_HIT_sum'), tcc.get('TCC_MISS_sum')
        return {
            "l2_hit_rate": (hit / max(hit + miss, 1.0)) if hit is not None and miss is not None else None,
            "TCC_HIT_sum": hit, "TCC_MISS_sum": miss,
            "kernel": kernel, "dispatches": f['dispatches'],
            "FETCH_SIZE_KiB_per_launch": f['FETCH_SIZE'], "WRITE_SIZE_KiB_per_launch": w['WRITE_SIZE'],
            "traffic_bytes_per_launch": traffic,
            "hbm_GBps": traffic / (sq['avg_dur_us'] * 1e-6) / 1e9,
            "hbm_fraction_of_8TBps": traffic / (sq['avg_dur_us'] * 1e-6) / 8e12,
            "SQ_VALU_MFMA_BUSY_CYCLES": sq.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0),
            "GRBM_GUI_ACTIVE_sum_over_8_xcd": m['GRBM_GUI_ACTIVE'],
            "avg_dur_us": sq['avg_dur_us'], "avg_dur_us_misc_pass": m['avg_dur_us'],
            "mfma_busy_fraction": sq.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / 1024.0 / max(cyc, 1.0),   # 1024 SIMDs
            "effective_clock_GHz": cyc / m['avg_dur_us'] * 1e-3,
            "lds_bank_conflict_fraction": m.get('SQ_LDS_BANK_CONFLICT', 0.0) / max(m.get('SQ_LDS_IDX_ACTIVE', 0.0), 1.0),
        }
    names = [k for k in parsed['fetch'] if all(k in parsed[p] for p in ('write', 'sq', 'misc')) and
             parsed['fetch'][k]['avg_dur_us'] > 10.0 and not k.startswith('at::')]
    ents = {k: entry(k) for k in names}
    dom = max(ents.values(), key=lambda e: e['avg_dur_us'] * e['dispatches'] if ('igemm' in e['kernel'] or 'wino' in e['kernel']) else 0)['kernel']
    sys.path.insert(0, ROOT)
    from bench import _csrc_fingerprint
    js = {
        "csrc_fingerprint": _csrc_fingerprint(),        # bench.py quotes `traffic` only for the sources it was collected on
        "command": command + " (under rocprofv3 --kernel-trace --pmc <counter set>, one pass per counter set)",
        "correction": "gfx950 FETCH_SIZE reports 1/2 of wide coalesced reads (MI355X_MICROARCH.md, HBM): bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024",
        "kernel": dom,
        "kernels": ents,
    }
    js.update(ents[dom])
    json.dump(js, open(os.path.join(ROOT, 'profiles', prefix + '_pmc.json'), 'w'), indent=1)
    for k, v in sorted(ents.items(), key=lambda kv: -kv[1]['avg_dur_us'] * kv[1]['dispatches'])[:40]:
        print('%-46s %8.1f us  %8.1f MB  %5.2f TB/s  mfma %4.2f  clk %4.2f  L2 hit %s' % (
            k[:46], v['avg_dur_us'], v['traffic_bytes_per_launch'] / 1e6, v['hbm_GBps'] / 1e3, v['mfma_busy_fraction'],
            v['effective_clock_GHz'], '%.2f' % v['l2_hit_rate'] if v['l2_hit_rate'] is not None else '-'))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2], sys.argv[3])
