"""Rehearsal of bench.py's N>1 path on ONE card: two ranks under torch.distributed.run share cuda:0 and exchange
gradients through gloo (MCGEN_DIST_BACKEND=gloo; RCCL refuses two ranks on one device).  Checks the launch
contract (env rendezvous on 127.0.0.1, graph capture with a live process group, all-reduce between replays,
max-over-ranks timing, the collective-free roofline pass on rank 0 while the other rank waits at the final barrier,
one JSON line from rank 0) -- not performance."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('workload,batch,extra', [('mcpixelcnn', 16, []), ('cifar10', 16, []),
                                                  # BASELINE configs[2]'s per-GPU shard (global 512 over 8 GPUs = 64 / GPU), bf16 gradient wire
                                                  ('coil100', 64, ['--grad-bf16'])])
def test_bench_two_ranks_one_card(workload, batch, extra):
    env = dict(os.environ, MCGEN_DIST_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', '29533', os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--workload', workload, '--steps', '2',
           '--warmup', '1', '--batch', str(batch), '--sustain-steps', '0'] + extra
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['steps'] == 2 and out['scaling'] == 'weak'
    assert out['config']['global_batch'] == 2 * batch and out['value'] > 0
    if workload in ('cifar10', 'coil100'):
        # N > 1 line: the gradient exchange's own time per iteration and the part of it hidden under the backward pass
        assert out['comm'] is not None and out['comm']['comm_us'] > 0
        assert 0 <= out['overlap_us'] <= out['comm']['comm_us'] + 1e-6 and out['grad_wire_dtype'] == ('bf16' if extra else 'f32')


@pytest.mark.parametrize('workload', ['cifar10', 'coil100'])
def test_bench_default_path_replays_graphs(workload):
    """The single-GPU bench command at its real batch: the HIP-graph capture must succeed (an eager fall-back would still
    print a line -- `config.graph_replay` says which path was timed) and the line carries the contract's keys."""
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--workload', workload, '--steps', '3', '--warmup', '1',
           '--no-cpu-baseline', '--sustain-steps', '0']
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')][-1])
    assert out['config']['graph_replay'] is True, r.stderr[-3000:]
    assert out['n_gpus'] == 1 and out['value'] > 0 and out['mask_compaction'] is True
    rf = out['roofline']
    assert rf['unit'] in ('TFLOP/s', 'GB/s') and 0 < rf['frac'] < 1 and rf['bound'] in ('mfma', 'hbm')
