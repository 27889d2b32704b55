import os, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools')
import bench_conv as b
L = b.wgrad_layers()
for rnd in range(3):
    for mode in ('0', '1'):
        os.environ['MCGEN_WGRAD_MODE'] = mode
        print(mode, ' '.join(f'{n}={f/ b.timeit(fn)/1e12:6.0f}' for n, (fn, f) in L.items()), flush=True)
