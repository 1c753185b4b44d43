"""bench.py's own multi-rank code path, before 8-GPU hardware meets it: the self-launch under torch.distributed.run, one
rank per process, the barrier + synchronize brackets, MAX over ranks of the timed region, the single JSON line from rank
0 -- and the fallback order for a graph capture that never returns (eager result first, capture under a watchdog).
Two ranks share cuda:0 over gloo (``--dev-backend gloo``; RCCL refuses two ranks on one device)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_extra=None, timeout=600, config='c10_b512', exchange='overlap'):
    env = dict(os.environ)
    env.update(env_extra or {})
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--dev-backend', 'gloo', '--config', config,
           '--steps', '2', '--warmup', '1', '--no-cpu-baseline', '--exchange', exchange] + extra
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout, text=True)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    return r, lines


def test_two_gloo_ranks_eager_line():
    r, lines = _run(['--graph', 'off'])
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['steps'] == 2 and out['warmup'] == 1 and out['scaling'] == 'strong'
    assert out['config']['global_batch'] == 512 and out['config']['per_gpu_batch'] == 256
    assert out['config']['rccl_ranks'] == 2 and out['config']['backend'] == 'gloo' and out['config']['launch'] == 'eager'
    assert out['config']['losses_finite'] and out['value'] > 0
    assert abs(out['value'] - 512 * 2 / (out['ms_per_step'] * 2e-3)) < 1e-2 * out['value']
    assert out['roofline']['kernel'] and 'cpu_baseline' not in out


def test_two_gloo_ranks_pick_the_faster_gradient_exchange():
    """--exchange auto (the default with several ranks, round 6): every workload is timed eagerly with the exchange overlapped
    with the backward AND after it; every rank takes rank 0's decision; the line says which one it carries and shows both."""
    r, lines = _run(['--graph', 'off'], exchange='auto')
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    both = out['config']['eager_ms_per_step_by_exchange']
    assert set(both) == {'overlapped', 'after_the_backward'} and min(both.values()) > 0
    picked_serial = out['config']['grad_exchange'].startswith('after')
    assert abs(out['ms_per_step'] - (both['after_the_backward'] if picked_serial else both['overlapped'])) < 1e-6
    if picked_serial:
        assert both['after_the_backward'] < 0.97 * both['overlapped']
    assert out['config']['losses_finite'] and out['n_gpus'] == 2


def test_graph_capture_that_never_returns_reports_the_eager_result():
    r, lines = _run(['--graph', 'on', '--graph-timeout', '6'], {'CONTRAD_BENCH_FAKE_CAPTURE_HANG': '1'})
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['config']['launch'] == 'eager (graph capture timed out)' and out['config']['graph_hung'] is True
    assert out['config']['losses_finite'] and out['value'] > 0 and out['steps'] == 2
    assert 'did not get through graph capture' in r.stderr


def test_graph_capture_that_raises_keeps_every_workloads_eager_line():
    """A gloo collective cannot be captured: the capture of the FIRST workload raises and leaves the process in a state in
    which nothing else can be measured (stream "capture invalidated", device RNG in capture mode).  All three workloads
    were timed eagerly before any capture was attempted, so the line still carries every one of them."""
    r, lines = _run(['--graph', 'on', '--graph-timeout', '120'], config='all', timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out['config']['launch'].startswith('eager (graph capture failed') and out['value'] > 0
    assert set(out['other_configs']) == {'sg2_32', 'sg2_512'}
    for name, w in out['other_configs'].items():
        assert 'error' not in w and w['value'] > 0 and w['config']['launch'] == 'eager' and w['config']['losses_finite'], name
    assert out['other_configs']['sg2_512']['scaling'] == 'weak' and out['other_configs']['sg2_512']['config']['global_batch'] == 32
