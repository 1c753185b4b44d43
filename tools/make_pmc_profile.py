"""Turn the rocprofv3 --pmc passes of tools/prof_pmc.sh into profiles/r01_bench_n1_pmc.{txt,json}.
usage: python tools/make_pmc_profile.py gpurun_out/pmc_bench "igemm_lean_kernel<2, 128, 128>"
"""
import io, json, os, sys
from contextlib import redirect_stdout
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import rocpd_pmc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def table(db):
    buf = io.StringIO()
    with redirect_stdout(buf):
        rocpd_pmc.main(db, 'igemm')
    return buf.getvalue()


def parse(text):
    out, cur = {}, None
    for line in text.splitlines():
        if not line.startswith(' '):
            name, rest = line.split('  dispatches=')
            n, dur = rest.split('  avg_dur_us=')
            cur = out.setdefault(name.strip(), {'dispatches': int(n), 'avg_dur_us': float(dur)})
        elif cur is not None:
            k, v = line.split()
            cur[k] = float(v)
    return out


def main(d, kernel):
    txt = ['# rocprofv3 --pmc passes on: python bench.py --steps 3 --warmup 2 --no-cpu-baseline  (1x MI355X, round 1)',
           '# per-dispatch means; counters summed over XCDs/SEs; FETCH_SIZE/WRITE_SIZE in KiB (FETCH_SIZE under-reports '
           'wide coalesced reads by 2x on gfx950, MI355X_MICROARCH.md HBM section)']
    parsed = {}
    for p in ('fetch', 'write', 'sq', 'misc'):
        t = table(os.path.join(d, p + '_results.db'))
        txt.append('== pass: ' + p)
        txt.append(t.rstrip())
        parsed[p] = parse(t)
    open(os.path.join(ROOT, 'profiles', 'r01_bench_n1_pmc.txt'), 'w').write('\n'.join(txt) + '\n')
    def entry(kernel):
        f, w, sq, m = (parsed[p][kernel] for p in ('fetch', 'write', 'sq', 'misc'))
        cyc = m['GRBM_GUI_ACTIVE'] / 8.0                       # summed over the 8 XCDs
        return {
            "kernel": kernel, "dispatches": f['dispatches'],
            "FETCH_SIZE_KiB_per_launch": f['FETCH_SIZE'], "WRITE_SIZE_KiB_per_launch": w['WRITE_SIZE'],
            "traffic_bytes_per_launch": (2 * f['FETCH_SIZE'] + w['WRITE_SIZE']) * 1024,
            "SQ_VALU_MFMA_BUSY_CYCLES": sq['SQ_VALU_MFMA_BUSY_CYCLES'],
            "GRBM_GUI_ACTIVE_sum_over_8_xcd": m['GRBM_GUI_ACTIVE'],
            "avg_dur_us": sq['avg_dur_us'], "avg_dur_us_misc_pass": m['avg_dur_us'],
            "mfma_busy_fraction": sq['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024.0 / cyc,   # 1024 SIMDs
            "effective_clock_GHz": cyc / m['avg_dur_us'] * 1e-3,
            "lds_bank_conflict_fraction": m['SQ_LDS_BANK_CONFLICT'] / max(m['SQ_LDS_IDX_ACTIVE'], 1.0),
        }
    names = [k for k in parsed['fetch'] if all(k in parsed[p] for p in ('write', 'sq', 'misc')) and
             parsed['fetch'][k]['avg_dur_us'] > 100.0]
    js = {
        "command": "python bench.py --steps 3 --warmup 2 --no-cpu-baseline (under rocprofv3 --kernel-trace --pmc <counter>, one pass per counter set)",
        "correction": "gfx950 FETCH_SIZE reports 1/2 of wide coalesced reads (MI355X_MICROARCH.md, HBM): bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024",
        "kernel": kernel,
        "kernels": {k: entry(k) for k in names},
    }
    js.update(entry(kernel))
    json.dump(js, open(os.path.join(ROOT, 'profiles', 'r01_bench_n1_pmc.json'), 'w'), indent=1)
    print(json.dumps({k: (round(v['traffic_bytes_per_launch'] / 1e6, 1), round(v['mfma_busy_fraction'], 3),
                          round(v['effective_clock_GHz'], 2)) for k, v in js['kernels'].items()}, indent=1))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
