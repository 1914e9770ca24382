import sys, os, subprocess
if len(sys.argv) > 1:
    rl, nr, ms, ml = map(int, sys.argv[1:5])
    sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
    import numpy as np, synth, kit4b_amd as k4
    names, chroms = synth.golden_genome()
    ix = k4.SfxIndex.open('tests/golden/g1.sfx'); ix.set_max_iter(5000)
    r, _ = synth.make_reads(chroms[:3], nr, rl, seed=rl, sub_lambda=max(1.0, rl / 80.0), n_prob=0.05, edge_frac=0.1)
    eg = ix.kalign_batch(r, max_subs=ms, max_ml=ml, pe_mode=1 if ml > 1 else 0)
    print('ok', rl, nr, ms, ml, np.bincount(eg['out']['nar'], minlength=6), ix.counters()['n_slow'], flush=True)
    sys.exit(0)
for rl, nr in ((15, 20), (129, 200), (257, 150), (400, 100), (513, 60), (900, 40), (2000, 20), (2000, 2), (1200, 5)):
    for ms, ml in ((2, 1), (5, 10)):
        p = subprocess.run(['timeout', '-k', '5', '60', sys.executable, sys.argv[0], str(rl), str(nr), str(ms), str(ml)], capture_output=True, text=True)
        print(rl, nr, ms, ml, 'rc', p.returncode, p.stdout.strip()[-200:], p.stderr.strip()[-300:].replace('\n', ' | ') if p.returncode else '', flush=True)
