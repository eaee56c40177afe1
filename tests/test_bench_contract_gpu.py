"""bench.py prints ONE JSON line with the driver's contract keys (+ roofline, + cpu_baseline at N = 1)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_json_contract():
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '2', '--warmup', '1',
                          '--model', 'esm2_150m', '--tokens', '16384', '--seq-len', '512', '--cpu-sample-tokens', '512'],
                         capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith('{')]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
              'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in d, k
    assert d['n_gpus'] == 1 and d['steps'] == 2 and d['warmup'] == 1 and d['higher_is_better'] is True
    assert d['unit'] == 'residues/s' and d['scaling'] == 'weak' and d['dtype'] == 'bf16' and d['data'] == 'synthetic'
    assert d['vs_baseline'] is None and 'workload' in d['config'] and 'model' not in d['config']
    assert abs(d['value'] - 16384 / (d['ms_per_step'] * 1e-3)) / d['value'] < 1e-3
    r = d['roofline']
    assert r['bound'] == 'mfma' and r['unit'] == 'TFLOP/s' and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-3 and 'traffic' in r
    assert 'traffic_source' in r and (r['traffic'] is None) == (r['traffic_source'] is None)      # a static file, named; None off the headline
    assert 'instrumented' in d['kernel_ms_per_step_source'] and 'multi_gpu' not in d
    assert d['parity']['rows_vs_oracle_bf16'] == 512
    c = d['cpu_baseline']
    assert c['kind'] == 'port' and c['unit'] == 'residues/s' and c['cores'] >= 1 and c['value'] > 0 and c['sample']


def test_bench_self_launch_rccl_world1():
    """`--spawn` forces the N>1 launcher path on this 1-GPU box: bench.py re-executes itself under
    torch.distributed.run, initialises RCCL (backend nccl) with world size 1 and must say so."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--spawn', '--steps', '2', '--warmup', '1',
                          '--model', 'esm2_8m', '--tokens', '4096', '--seq-len', '256', '--no-cpu-baseline'],
                         capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith('{')]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d['n_gpus'] == 1 and d['config']['launcher'] == 'torch.distributed.run (self-launched)'
    m = d['multi_gpu']                                                           # launcher runs say what RCCL saw and time the gather apart
    assert m['world_size_seen'] == 1 and m['backend'].startswith('nccl') and m['gather_ms'] > 0
    assert m['gather_bytes_per_rank'] == 4096 * 33 * 2 and m['rank_ms_per_step']['min'] <= m['rank_ms_per_step']['max']
    assert abs(m['ms_per_step_incl_gather'] - (d['ms_per_step'] + m['gather_ms'])) < 2e-3
    assert 'esm2_8m' in d['metric'] and 'ESM2-650M' not in d['metric']          # label follows --model


def test_sharded_forward_real_model_rccl_world1():
    """esme.shard.sharded_forward around the REAL model on RCCL (world size 1 on this box): logits equal the
    plain single-process forward bit for bit, in input order (the plan permutes sequences longest-first)."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'shard_check.py')], capture_output=True, text=True,
                         timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    assert 'sharded == single: True' in out.stdout, out.stdout
