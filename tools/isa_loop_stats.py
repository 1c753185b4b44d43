#!/usr/bin/env python3
"""Static instruction mix of the main loop of every MFMA kernel in libcontrad_hip.so (runs without a GPU).

    python tools/isa_loop_stats.py [path/to/libcontrad_hip.so] > profiles/rNN_isa_loop_stats.txt

Extracts the gfx950 code objects from the shared library (llvm-objdump --offloading), disassembles them, and for each
kernel that holds v_mfma instructions finds its main loop = the backward branch whose span holds the most MFMAs.  For
that span it prints the instruction classes and the issue budget: a v_mfma_f32_32x32x2_f32 occupies its SIMD's matrix
pipe for 64 cycles (MI355X_MICROARCH.md, constants table) and an issue slot is ~4 cycles, so a wave has 16 slots per
MFMA of its own and, at W waves per SIMD (each waiting for the pipe while the other W-1 use it), 16*W slots of wall
time per MFMA it issues.  `other/mfma` well under 16 means the loop is not bound by instruction issue; what is left is
waiting (s_waitcnt / s_barrier, SQ_WAIT_ANY in the PMC summaries) and the clock.

Inner scalar loops (the tap / row advance `while`s of the gather address walk) are counted once: they run 0 or 1 times
per K-tile on every layer of the benchmarks.
"""
import os
import re
import subprocess
import sys
import tempfile
from collections import Counter

OBJDUMP = '/opt/rocm/lib/llvm/bin/llvm-objdump'
CXXFILT = 'c++filt'

INS = re.compile(r'^\t(\S+)\s*(.*?)\s*// ([0-9A-F]{12}):')
SYM = re.compile(r'^[0-9a-f]+ <(\S+)>:')


def classify(op):
    if op.startswith('v_mfma'):
        return 'mfma'
    if op.startswith('ds_read') or op.startswith('ds_load'):
        return 'lds_read'
    if op.startswith('ds_write') or op.startswith('ds_store'):
        return 'lds_write'
    if op.startswith('ds_'):
        return 'lds_other'
    if op.startswith(('buffer_load', 'global_load', 'flat_load', 'scratch_load')):
        return 'vmem_load'
    if op.startswith(('buffer_store', 'global_store', 'flat_store', 'scratch_store', 'buffer_atomic', 'global_atomic')):
        return 'vmem_store'
    if op == 's_waitcnt':
        return 'waitcnt'
    if op == 's_barrier':
        return 'barrier'
    if op == 's_nop':
        return 'nop'
    if op.startswith(('s_cbranch', 's_branch')):
        return 'branch'
    if op.startswith('s_load') or op.startswith('s_buffer_load'):
        return 'smem'
    if op.startswith('s_'):
        return 'salu'
    if op.startswith('v_accvgpr'):
        return 'acc_move'
    if op.startswith('v_'):
        return 'valu'
    return 'other'


def kernels_of(obj):
    text = subprocess.run([OBJDUMP, '-d', obj], capture_output=True, text=True, check=True).stdout
    cur, out = None, {}
    for line in text.splitlines():
        m = SYM.match(line)
        if m:
            cur = m.group(1)
            out[cur] = []
            continue
        m = INS.match(line)
        if m and cur is not None:
            out[cur].append((int(m.group(3), 16), m.group(1), m.group(2)))
    return out


def main_loop(ins):
    """(first, last) instruction indices of the backward-branch span with the most MFMAs, or None."""
    addr = {a: i for i, (a, _, _) in enumerate(ins)}
    mf = [0]
    for _, op, _ in ins:
        mf.append(mf[-1] + (1 if op.startswith('v_mfma') else 0))
    best = None
    for i, (a, op, arg) in enumerate(ins):
        if not op.startswith(('s_cbranch', 's_branch')):
            continue
        off = int(arg.split()[0])
        if off < 32768:
            continue
        tgt = a + 4 + 4 * (off - 65536)
        j = addr.get(tgt)
        if j is None:
            continue
        n = mf[i + 1] - mf[j]
        # the innermost span holding the maximum: prefer more MFMAs, then the shorter span
        if n and (best is None or n > best[0] or (n == best[0] and i - j < best[2] - best[1])):
            best = (n, j, i)
    return None if best is None else best[1:]


def demangle(names):
    try:
        out = subprocess.run([CXXFILT], input='\n'.join(names), capture_output=True, text=True, check=True).stdout
        return dict(zip(names, out.splitlines()))
    except Exception:
        return {n: n for n in names}


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(
        os.path.dirname(os.path.abspath(__file__)), '..', 'contrad_amd', 'csrc', 'libcontrad_hip.so')
    lib = os.path.abspath(lib)
    with tempfile.TemporaryDirectory() as tmp:
        link = os.path.join(tmp, 'lib.so')
        os.symlink(lib, link)
        subprocess.run([OBJDUMP, '--offloading', link], capture_output=True, text=True, check=True, cwd=tmp)
        objs = sorted(f for f in os.listdir(tmp) if 'gfx950' in f)
        rows = []
        for o in objs:
            for name, ins in kernels_of(os.path.join(tmp, o)).items():
                if not any(op.startswith('v_mfma') for _, op, _ in ins):
                    continue
                span = main_loop(ins)
                if span is None:
                    continue
                c = Counter(classify(op) for _, op, _ in ins[span[0]:span[1] + 1])
                kinds = Counter(op for _, op, _ in ins[span[0]:span[1] + 1] if op.startswith('v_mfma'))
                rows.append((name, len(ins), c, kinds))
    dm = demangle([r[0] for r in rows])
    cols = ['mfma', 'lds_read', 'lds_write', 'vmem_load', 'vmem_store', 'valu', 'acc_move', 'salu', 'smem', 'waitcnt',
            'barrier', 'branch', 'nop']
    print('# static instruction mix of the main loop (backward-branch span with the most MFMAs) per MFMA kernel of '
          + os.path.basename(lib))
    print('# other/mfma = every non-MFMA instruction of the span per MFMA; a 64-cycle v_mfma_f32_32x32x2_f32 leaves '
          '~16 issue slots of its own wave (x waves per SIMD of wall time)')
    print('%-40s %6s ' % ('kernel', 'insns') + ' '.join('%9s' % c for c in cols) + ' %10s' % 'other/mfma')
    for name, n, c, kinds in sorted(rows, key=lambda r: dm[r[0]]):
        short = dm[name].replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '')
        other = sum(v for k, v in c.items() if k != 'mfma')
        print('%-40s %6d ' % (short, n) + ' '.join('%9d' % c.get(k, 0) for k in cols)
              + ' %10.2f' % (other / max(1, c['mfma'])) + '   ' + ','.join('%s x%d' % kv for kv in kinds.items()))


if __name__ == '__main__':
    main()
