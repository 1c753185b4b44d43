"""Instruction / register budget of the lean implicit-GEMM instances, checked on the generated gfx950 ISA (no GPU needed).

On gfx950 every VALU / SALU instruction a wave issues between its MFMAs takes matrix-pipe time ("The
loop, instruction by instruction"), and the 128 x 128 instances sit at the 128-VGPR limit of 4 waves per SIMD: a harmless
looking source change can make the allocator spill accumulators into the K loop (seen in round 4: 25 ... 215 spilled
registers from unrolling those instances) or put index arithmetic back into it.  This test compiles igemm.hip once for the
device (about 20 s) and pins, per instance,
  * registers: <= 128 VGPRs, spills within the few the prologue -> epilogue values cost today;
  * the K loop (first ... last v_mfma of the kernel, extended to the enclosing labels / branch): no scratch traffic and
    no more non-MFMA instructions per MFMA than today's count plus a margin.
"""
import collections
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'contrad_amd', 'csrc', 'igemm.hip')
HIPCC = '/opt/rocm/bin/hipcc'

# (MODE, BM, BN, BAL) -> max non-MFMA instructions per MFMA inside the K loop (today's value in the comment)
LOOP_BUDGET = {
    (0, 128, 128, 0): 2.15,   # 1.84
    (1, 128, 128, 0): 1.85,   # 1.56
    (2, 128, 128, 0): 3.30,   # 2.86
    (1, 128, 128, 1): 2.00,   # 1.69
    (0, 64, 128, 0): 4.3, (1, 64, 128, 0): 4.0, (0, 128, 64, 0): 4.3, (1, 128, 64, 0): 4.1, (2, 128, 64, 0): 5.6,
    (0, 64, 64, 0): 5.5, (1, 64, 64, 0): 5.2, (2, 64, 64, 0): 6.1,
    (1, 128, 64, 1): 4.0, (1, 64, 128, 1): 4.0, (1, 64, 64, 1): 6.4,
}
MAX_VGPR_SPILL = 8        # today: 0 everywhere except 3 (WGRAD 128 x 128) and 5 (balanced DGRAD 128 x 128), outside the loop


@pytest.fixture(scope='module')
def isa(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip('hipcc not installed')
    d = tmp_path_factory.mktemp('isa')
    out = os.path.join(str(d), 'igemm.s')
    r = subprocess.run([HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', SRC, '--cuda-device-only', '-S', '-o', out,
                        '-Rpass-analysis=kernel-resource-usage'], capture_output=True, text=True, cwd=str(d))
    assert r.returncode == 0, r.stderr[-2000:]
    text = open(out).read()
    shutil.rmtree(str(d), ignore_errors=True)
    return text, r.stderr


def _instances(text):
    """{(MODE, BM, BN, BAL): [instruction lines of the kernel, labels kept]}"""
    out, cur = {}, None
    for line in text.split('\n'):
        m = re.match(r'^_ZN\d+_GLOBAL__N_1\d+igemm_lean_kernelILi(\d)ELi(\d+)ELi(\d+)ELb([01])E\w*:', line)
        if m:
            cur = out.setdefault(tuple(int(g) for g in m.groups()), [])
            continue
        if cur is not None and line.startswith('.Lfunc_end'):
            cur = None
            continue
        if cur is not None:
            t = line.split(';')[0].rstrip()
            if t.strip() and (re.match(r'^\.LBB', t) or not t.strip().startswith('.')):
                cur.append(t)
    return out


def _loop(lines):
    idx = [i for i, l in enumerate(lines) if 'v_mfma' in l]
    lo, hi = idx[0], idx[-1]
    while lo > 0 and not lines[lo].startswith('.LBB'):
        lo -= 1
    while hi < len(lines) - 1 and 's_cbranch' not in lines[hi]:
        hi += 1
    return [l.strip() for l in lines[lo:hi + 1] if not l.startswith('.LBB')]


def test_lean_instances_fit_the_register_file(isa):
    _, remarks = isa
    cur, seen = None, 0
    for line in remarks.split('\n'):
        m = re.search(r'Function Name: (\S+)', line)
        if m:
            cur = m.group(1) if 'igemm_lean_kernel' in m.group(1) else None
            seen += cur is not None
            continue
        if cur is None:
            continue
        m = re.search(r'remark:\s+VGPRs: (\d+)', line)
        if m:
            assert int(m.group(1)) <= 128, '%s: %s VGPRs -> fewer than 4 waves per SIMD' % (cur, m.group(1))
        m = re.search(r'VGPRs Spill: (\d+)', line)
        if m:
            assert int(m.group(1)) <= MAX_VGPR_SPILL, '%s spills %s vector registers' % (cur, m.group(1))
    assert seen >= 19          # 3 modes x 5 tiles + the balanced DGRAD instances


def test_k_loops_stay_lean(isa):
    text, _ = isa
    inst = _instances(text)
    assert set(LOOP_BUDGET) <= set(inst), sorted(set(LOOP_BUDGET) - set(inst))
    report = []
    for key, lines in sorted(inst.items()):
        loop = _loop(lines)
        ops = collections.Counter('mfma' if 'mfma' in l.split()[0] else l.split()[0] for l in loop)
        n_mfma = ops['mfma']
        assert n_mfma >= 8, (key, n_mfma)
        assert not any(l.startswith('scratch_') for l in loop), '%s: register spills inside the K loop' % (key,)
        ratio = (len(loop) - n_mfma) / float(n_mfma)
        report.append((key, n_mfma, round(ratio, 2)))
        if key in LOOP_BUDGET:
            assert ratio <= LOOP_BUDGET[key], '%s: %.2f non-MFMA instructions per MFMA in the K loop (budget %.2f)' % (
                key, ratio, LOOP_BUDGET[key])
    print(report)
