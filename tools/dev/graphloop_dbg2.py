"""dev: train_gan main() with --graph, variants in separate processes."""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
VARIANTS = {'plain': [], 'warmup': ['--use_warmup'], 'eager_first': ['--use_warmup'], 'print1': ['--print_every', '1']}

if len(sys.argv) > 1:
    import faulthandler
    faulthandler.enable()
    sys.path.insert(0, ROOT)
    from contrad_amd.train_gan import main
    v = sys.argv[1]
    gin = os.path.join(ROOT, 'configs', 'gan', 'cifar10', 'c10_b64.gin')
    base = [gin, 'sndcgan', '--mode=contrad', '--aug=simclr', '--synthetic', '--max_steps', '6', '--evaluate_every', '6',
            '--seed', '5']
    d = tempfile.mkdtemp()
    if v == 'eager_first':
        main(base + VARIANTS[v] + ['--logdir', d + '/e'])
    main(base + VARIANTS[v] + ['--graph', '--logdir', d + '/g'])
    print(v, 'ok')
else:
    for v in VARIANTS:
        r = subprocess.run([sys.executable, __file__, v], capture_output=True, text=True, timeout=300)
        print(v, 'rc', r.returncode, r.stdout.strip()[-100:].replace('\n', ' | '), '||', r.stderr.strip()[:600].replace('\n', ' | '))
