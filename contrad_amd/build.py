"""Build csrc/*.hip into csrc/libcontrad_hip.so for gfx950 with hipcc (in-tree, no torch headers)."""
import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'csrc')
LIB = os.path.join(CSRC, 'libcontrad_hip.so')
# the same library with the development switches of the conv engine compiled in (common.h: CONTRAD_DEV_SWITCHES); only the
# A/B tools and the two parity tests that force a tile plan load it (CONTRAD_HIP_LIB=<this path>)
DEV_LIB = os.path.join(CSRC, 'libcontrad_hip_dev.so')
DEV_SOURCES = ('igemm',)                 # every getenv-style switch lives in igemm.hip and the headers it includes
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC'] + os.environ.get('CONTRAD_EXTRA_HIPCC_FLAGS', '').split()


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    srcs = sorted(glob.glob(os.path.join(CSRC, '*.hip')))
    hdrs = sorted(glob.glob(os.path.join(CSRC, '*.h'))) + \
        sorted(glob.glob(os.path.join(CSRC, '..', '..', 'include', '*.h')))
    objs, dev_objs = [], []
    jobs = []
    for s in srcs:
        o = s[:-4] + '.o'
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            jobs.append((s, o, []))
        if os.path.basename(s)[:-4] in DEV_SOURCES:
            od = s[:-4] + '_dev.o'
            dev_objs.append(od)
            if force or _stale(od, [s] + hdrs):
                jobs.append((s, od, ['-DCONTRAD_DEV_SWITCHES']))
        else:
            dev_objs.append(o)

    def cc(job):
        s, o, extra = job
        cmd = [HIPCC] + FLAGS + extra + ['-c', s, '-o', o]
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(cc, jobs))
    if force or jobs or _stale(LIB, objs):
        cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    if force or jobs or _stale(DEV_LIB, dev_objs):
        cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', DEV_LIB] + dev_objs
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
    print(LIB)
