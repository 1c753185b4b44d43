"""CPU test of the profile-reproducibility tooling: tools/rocpd_rows.py must attribute the igemm dispatches of a
rocprofv3 kernel trace (ROCm 7.2 rocpd sqlite) to the (kernel, layer shape) rows of bench.py's --shape-table by their
ORDER within a step, keep the lazy-R1 step in its own section, and report steps it cannot match."""
import io
import json
import os
import sqlite3
import sys
from contextlib import redirect_stdout

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))


def _db(path, steps):
    db = sqlite3.connect(path)
    db.execute("create table kernels (name text, start integer, end integer, grid_x integer, grid_y integer, "
               "grid_z integer, workgroup_x integer, workgroup_y integer, workgroup_z integer)")
    t = 0
    for launches in steps:
        for name, blocks, us in launches:
            db.execute("insert into kernels values (?,?,?,?,?,?,?,?,?)",
                       (name, t, t + int(us * 1000), blocks * 256, 1, 1, 256, 1, 1))
            t += int(us * 1000) + 500
    db.commit()
    db.close()


def test_rows_are_attributed_by_launch_order(tmp_path):
    import rocpd_rows
    lean = 'void (anonymous namespace)::igemm_lean_kernel<2, 128, 128>((anonymous namespace)::IgemmArgs)'
    fwd = 'void (anonymous namespace)::igemm_lean_kernel<0, 128, 128>((anonymous namespace)::IgemmArgs)'
    adam = 'void (anonymous namespace)::adam_kernel((anonymous namespace)::AdamArgs)'
    other = 'upfirdn4_u1d1_kernel(float const*)'
    # two shapes share instance AND workgroup count: only the order tells them apart
    # the engine's dedicated kernels (contrad_conv2d_path 4 / 5) are rows of the shape table as well
    c32 = 'void (anonymous namespace)::wgrad_c32_kernel((anonymous namespace)::WgradC32Args)'
    k1 = 'void (anonymous namespace)::fwd_k1_kernel(float const*, float const*, long long, int)'
    plain = [(fwd, 768, 100.0), (other, 10, 5.0), (lean, 1008, 200.0), (lean, 1008, 400.0), (k1, 1, 2.0), (c32, 512, 100.0),
             (adam, 64, 50.0)]
    r1 = [(fwd, 768, 100.0), (fwd, 256, 30.0), (lean, 1008, 200.0), (lean, 1008, 400.0), (lean, 512, 60.0), (adam, 64, 50.0)]
    cold = [(fwd, 999, 100.0), (adam, 64, 50.0)]
    dbp = str(tmp_path / 'kt.db')
    _db(dbp, [cold, r1, plain, plain, plain])
    table = {'config': 'unit', 'per_gpu_batch': 4, 'flop_rule': '2*N*Ho*Wo*K*C*KH*KW per launch',
             'sections': {
                 'r1_step': {'steps_sampled': 1, 'rows': [], 'sequence': [
                     ['igemm_lean_kernel<0, 128, 128>', [4, 8, 8, 16, 16, 3, 3, 1, 1], 768, 10.0],
                     ['igemm_lean_kernel<0, 128, 128>', [2, 8, 8, 16, 16, 3, 3, 1, 1], 256, 3.0],
                     ['igemm_lean_kernel<2, 128, 128>', [4, 8, 8, 16, 16, 3, 3, 1, 1], 1008, 20.0],
                     ['igemm_lean_kernel<2, 128, 128>', [4, 4, 4, 32, 32, 3, 3, 1, 1], 1008, 40.0],
                     ['igemm_lean_kernel<2, 128, 128>', [2, 8, 8, 16, 16, 3, 3, 1, 1], 512, 6.0]]},
                 'plain_step': {'steps_sampled': 2, 'rows': [], 'sequence': [
                     ['igemm_lean_kernel<0, 128, 128>', [4, 8, 8, 16, 16, 3, 3, 1, 1], 768, 10.0],
                     ['igemm_lean_kernel<2, 128, 128>', [4, 8, 8, 16, 16, 3, 3, 1, 1], 1008, 20.0],
                     ['igemm_lean_kernel<2, 128, 128>', [4, 4, 4, 32, 32, 3, 3, 1, 1], 1008, 40.0, 0.5],   # pixel-major: half issued
                     ['fwd_k1_kernel', [4, 1, 1, 16, 1, 1, 1, 1, 0], 1, 0.0],
                     ['wgrad_c32_kernel', [4, 8, 32, 32, 32, 3, 3, 1, 1], 512, 10.0]]}}}
    tp = str(tmp_path / 'shapes.json')
    json.dump(table, open(tp, 'w'))
    buf = io.StringIO()
    with redirect_stdout(buf):
        rocpd_rows.main(dbp, tp)
    out = buf.getvalue()
    assert '5 steps in the trace' in out
    assert 'section r1_step: 5 conv-engine launches per step, 1 trace step(s) matched' in out
    assert 'section plain_step: 5 conv-engine launches per step, 3 trace step(s) matched' in out
    assert '1 trace step(s) matched no section' in out and 'step 0 (1 conv-engine launches)' in out
    assert 'trace 1 / table 0' in out            # the mismatch reason is spelled out (which instance x workgroups differs)
    plain_part = out.split('section plain_step')[1]
    rows = [l.split() for l in plain_part.splitlines() if l.startswith('igemm_lean_kernel<2,128,128>')]
    by_shape = {r[1]: r for r in rows}
    # 20 GFLOP in 200 us = 100 TF/s; 40 GFLOP in 400 us = 100 TF/s: each shape got ITS dispatches (same instance + grid)
    assert abs(float(by_shape['4,8,8,16,16,3,3,1,1'][5]) - 200.0) < 1e-6 and abs(float(by_shape['4,4,4,32,32,3,3,1,1'][5]) - 400.0) < 1e-6
    # columns: kernel shape blocks n/step calls avg_us GFLOP/call exec TF/s(issued) frac(issued) nominalTF/s
    assert abs(float(by_shape['4,8,8,16,16,3,3,1,1'][10]) - 100.0) < 0.1 and abs(float(by_shape['4,4,4,32,32,3,3,1,1'][10]) - 100.0) < 0.1
    assert 'issued 60.0 GFLOP per step in 802.0 us of igemm dispatches' in plain_part and 'nominal 80.0 GFLOP' in plain_part
    assert abs(float(by_shape['4,4,4,32,32,3,3,1,1'][7]) - 0.5) < 1e-6 and abs(float(by_shape['4,4,4,32,32,3,3,1,1'][8]) - 50.0) < 0.1
    assert abs(float(by_shape['4,4,4,32,32,3,3,1,1'][9]) - 50.0 / 157.3) < 1e-3
    # no fraction above 1 is ever printed in the frac column (issued work cannot exceed the peak)
    for l in plain_part.splitlines():
        f = l.split()
        if len(f) >= 11 and l.startswith(('igemm', 'wgrad_c32', 'conv_c32', 'fwd_k1')) and f[9] != 'ambig':
            assert float(f[9]) <= 1.0, l
    assert any(l.startswith('wgrad_c32_kernel') and abs(float(l.split()[5]) - 100.0) < 1e-6 for l in plain_part.splitlines())


def test_isa_loop_stats_finds_the_main_loops_and_they_are_not_issue_bound():
    """tools/isa_loop_stats.py (profiles/r05_isa_loop_stats.txt): every 128x128 lean instance's main loop is found in
    the built library's gfx950 code object and carries far fewer than the ~16 issue slots a 64-cycle
    v_mfma_f32_32x32x2_f32 leaves its wave."""
    import subprocess
    import pytest
    lib = os.path.join(ROOT, 'contrad_amd', 'csrc', 'libcontrad_hip.so')
    import shutil
    if not (os.path.exists(lib) and os.path.exists('/opt/rocm/lib/llvm/bin/llvm-objdump') and shutil.which('c++filt')):
        pytest.skip('needs the built library, llvm-objdump and c++filt')
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'isa_loop_stats.py'), lib], capture_output=True,
                         text=True, check=True).stdout
    rows = {l[:40].strip(): l[40:].split() for l in out.splitlines() if l and not l.startswith(('#', 'kernel'))}
    for mode, mfma in ((0, 32), (1, 32), (2, 64)):
        r = rows['igemm_lean_kernel<%d, 128, 128, false>' % mode]
        assert int(r[1]) == mfma, r                  # K-tile of 32 MFMAs per wave (WGRAD: two K-tiles per trip)
        assert float(r[14]) < 4.0, r                 # non-MFMA instructions per MFMA
    assert int(rows['wgrad_c32_kernel'][1]) == 36 and float(rows['wgrad_c32_kernel'][14]) < 2.0
