#!/usr/bin/env python
"""Headline benchmark: residues/s of the packed ESM-2 forward on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--model esm2_650m] [--tokens 50000]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one `model(tokens, (cu_lens, max_len))` forward (embedding -> L layers ->
final LN -> LM head -> (T, V) bf16 logits on device) over one synthetic packed batch
that is already resident in HBM.  Workload at N=1: BASELINE.json configs[2], the config
the metric is quoted on: ESM2-650M, 50 000 packed residues, uniform-500 (B=100 x S=500;
closed-form FLOPs).  With N>1 every rank runs its own 50 000-residue batch
(weak scaling, weights replicated, proteins never span GPUs) and the only collective is
the all-gather of the (T_r, V) logits over RCCL/xGMI at the end of the step.

Prints ONE JSON line (rank 0).  Extra objects:
  roofline     dominant kernel = the bf16 MFMA GEMM (94 % of the FLOPs); `achieved` is
               algorithmic FLOPs per launch / average launch duration of the FFN
               up-projection GEMM (M=T, N=4E, K=E, GELU epilogue), measured with HIP events
               on the launch stream in an instrumented pass; `all_gemms` aggregates every
               GEMM launch of a step the same way.
  cpu_baseline the CPU oracle (torch-CPU restatement of the reference, bf16 like the
               reference's default) timed on this box's host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, 'esm-efficient_amd')):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch

PEAK_BF16_TFLOPS = 2500.0      # dense bf16 MFMA peak, MI355X_MICROARCH.md (measured 2495)
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--model', default='esm2_650m')
    ap.add_argument('--tokens', type=int, default=50000)
    ap.add_argument('--seq-len', type=int, default=500)
    ap.add_argument('--batch', choices=['uniform', 'proteome'], default='uniform')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-sample-tokens', type=int, default=8000)
    ap.add_argument('--no-gather', action='store_true', help='skip the logits all-gather when N>1')
    ap.add_argument('--graph', action='store_true',
                    help='replay the forward from a hipGraph (esme/graph.py); matters for small models / batches')
    ap.add_argument('--quantization', choices=['none', '4bit', '8bit'], default='none',
                    help="'4bit': layer projections resident in the esme-q4 format (not the headline config)")
    return ap.parse_args()


def pmc_traffic():
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/r01_traffic.json: FETCH_SIZE doubled as the gfx950 guide prescribes, + WRITE_SIZE).
    Counters cannot be read from inside the process, so this is the last profiled value, valid
    for the default workload only; None otherwise."""
    path = os.path.join(ROOT, 'profiles', 'r01_traffic.json')
    try:
        with open(path) as f:
            return int(json.load(f)['traffic_bytes_per_launch'])
    except Exception:
        return None


def cpu_baseline(weights, heads, kind, L, E, seq_len, sample_tokens):
    """Oracle (port of the reference's CPU path) on a bounded sample of the same workload:
    `sample_tokens` residues in sequences of `seq_len`, all L layers + head, bf16."""
    from oracle import esm_oracle as O
    from esme import synthetic as syn
    cores = os.cpu_count() or 1
    threads = min(cores, 64)
    torch.set_num_threads(threads)
    tokens, cu, max_len, lengths = syn.uniform_batch(sample_tokens, seq_len, seed=0)
    t0 = time.time()
    with torch.no_grad():
        out = O.forward_logits(weights, heads, tokens, cu, max_len, torch.bfloat16)
    dt = time.time() - t0
    assert out.shape[0] == sample_tokens
    return {'value': round(sample_tokens / dt, 1), 'unit': 'residues/s', 'cores': threads, 'kind': 'port',
            'sample': f'{sample_tokens} residues ({len(lengths)} x {seq_len}) through all {L} layers + LM head, '
                      f'bf16 torch-CPU oracle, {dt:.1f} s'}


def main():
    args = parse()
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world != args.gpus and world > 1:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        torch.cuda.set_device(local_rank)
        dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
    else:
        torch.cuda.set_device(0)
    dev = torch.device('cuda', local_rank if world > 1 else 0)

    from esme import ESM, _hip, synthetic as syn
    _hip.load()
    kind, L, E, H = syn.MODEL_ZOO[args.model]

    # ---- synthetic checkpoint in the reference layout -> from_pretrained
    weights = syn.synthetic_state_dict(kind, L, E, seed=0)
    with tempfile.TemporaryDirectory() as td:
        from safetensors.torch import save_file
        path = os.path.join(td, f'{args.model}.safetensors')
        save_file(weights, path, metadata=syn.checkpoint_metadata(args.model, L, E, H))
        model = ESM.from_pretrained(path, quantization=None if args.quantization == 'none' else args.quantization,
                                    device=str(dev))

    # ---- this rank's packed batch (resident in HBM before the timed region)
    if args.batch == 'uniform':
        tokens, cu, max_len, lengths = syn.uniform_batch(args.tokens, args.seq_len, seed=rank)
    else:
        tokens, cu, max_len, lengths = syn.proteome_batch(args.tokens, seed=rank)
    tokens, cu = tokens.to(dev), cu.to(dev)
    T = tokens.numel()
    V = model.vocab_size
    gathered = torch.empty(world * T, V, dtype=torch.bfloat16, device=dev) if world > 1 else None

    def step():
        logits = model.graphed(tokens, (cu, max_len), 'forward', clone=False) if args.graph else model(tokens, (cu, max_len))
        if world > 1 and not args.no_gather:
            dist.all_gather_into_tensor(gathered, logits)
        return logits

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        # one-time setup, not part of the W warm-up steps: the first two forwards pack / fold the weights into
        # their GEMM layouts, load the kernel modules and grow the allocator pools (338 ms and ~145 ms vs 75 ms)
        for _ in range(2):
            step()
        fence()
        for i in range(args.warmup):
            if os.environ.get('BENCH_DEBUG'):
                torch.cuda.synchronize(); _t = time.perf_counter()
            step()
            if os.environ.get('BENCH_DEBUG'):
                torch.cuda.synchronize(); print(f'[debug] warmup step {i}: {1e3 * (time.perf_counter() - _t):.1f} ms', file=sys.stderr)
        fence()
        t0 = time.perf_counter()
        for i in range(args.steps):
            if os.environ.get('BENCH_DEBUG'):
                torch.cuda.synchronize(); _t = time.perf_counter()
            out = step()
            if os.environ.get('BENCH_DEBUG'):
                torch.cuda.synchronize(); print(f'[debug] timed step {i}: {1e3 * (time.perf_counter() - _t):.1f} ms', file=sys.stderr)
        fence()
        elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert torch.isfinite(out.float()).all()
    ms_per_step = 1e3 * elapsed / args.steps
    value = world * T * args.steps / elapsed
    flops_step = syn.algorithmic_flops(kind, L, E, lengths)

    result = {
        'metric': 'residues/sec ESM2-650M fwd, 50k packed tokens; % bf16 MFMA peak; 1/2/4/8 GPU',
        'value': round(value, 1), 'unit': 'residues/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 3), 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
        'config': {'workload': f'{args.model} packed forward -> logits, {T} residues/GPU, '
                               f'{args.batch} batch ({len(lengths)} seqs, max_len {max_len})',
                   'residues_per_gpu': T, 'sequences_per_gpu': len(lengths), 'max_len': max_len,
                   'parallelism': f'dp{world} (protein-sharded, logits all-gather)' if world > 1 else 'single GPU',
                   'launch': 'hipGraph replay' if args.graph else 'eager (one ctypes launch per kernel)',
                   'setup': '2 untimed forwards before the warm-up steps (weight packing / LN folding, module load)',
                   'weights': 'synthetic (numpy PCG64), reference checkpoint layout'
                              + ('' if args.quantization == 'none' else f', layer projections {args.quantization} (esme/quantization.py)')},
        'e2e': {'algorithmic_tflop_per_step': round(flops_step / 1e12, 3),
                'tflops_per_gpu': round(flops_step / (ms_per_step * 1e-3) / 1e12, 1),
                'frac_bf16_mfma_peak': round(flops_step / (ms_per_step * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4)},
    }

    if rank == 0:
        # ---- instrumented pass: HIP events around every launch on the launch stream
        with torch.no_grad():
            _hip.TRACE = []
            for _ in range(max(1, min(args.steps, 3))):
                step()
            torch.cuda.synchronize()
            trace, _hip.TRACE = _hip.TRACE, None
        per_op = {}
        for op, meta, s, e in trace:
            per_op.setdefault((op, meta), []).append(s.elapsed_time(e))
        key = ('gemm', (T, 4 * E if kind == 'esm2' else 2 * syn.swiglu_width(E), E,
                        _hip.EPI_GELU if kind == 'esm2' else _hip.EPI_SWIGLU))
        if key in per_op:
            ms = sum(per_op[key]) / len(per_op[key])
            fl = 2.0 * key[1][0] * key[1][1] * key[1][2]
            g_ms = g_fl = 0.0
            for (op, meta), v in per_op.items():
                if op == 'gemm':
                    g_ms += sum(v)
                    g_fl += 2.0 * meta[0] * meta[1] * meta[2] * len(v)
            result['roofline'] = {
                'bound': 'mfma', 'kernel': f'gemm_bf16_kernel M={key[1][0]} N={key[1][1]} K={key[1][2]} (FFN up, fused epilogue)',
                'achieved': round(fl / (ms * 1e-3) / 1e12, 1), 'peak': PEAK_BF16_TFLOPS, 'unit': 'TFLOP/s',
                'frac': round(fl / (ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4), 'traffic': pmc_traffic() if (args.model == 'esm2_650m' and T == 50000) else None,
                'avg_launch_ms': round(ms, 4), 'launches_timed': len(per_op[key]),
                'all_gemms': {'achieved': round(g_fl / (g_ms * 1e-3) / 1e12, 1),
                              'frac': round(g_fl / (g_ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4)},
            }
        nsteps = max(1, min(args.steps, 3))
        by_op = {}
        for (op, meta), v in per_op.items():
            by_op[op] = by_op.get(op, 0.0) + sum(v) / nsteps
        result['kernel_ms_per_step'] = {k: round(v, 3) for k, v in sorted(by_op.items())}
        # HBM-bound kernels: algorithmic bytes / measured time
        hbm = {}
        for (op, meta), v in per_op.items():
            if op == 'layernorm' and meta == (T, E):
                hbm['layernorm'] = round(4.0 * E * T / (sum(v) / len(v) * 1e-3) / 1e9, 1)
            if op == 'rotary':
                hbm['rotary'] = round(8.0 * E * T / (sum(v) / len(v) * 1e-3) / 1e9, 1)
            if op == 'dequant4':        # 0.5 B code + 2 B bf16 per weight (+ 1/16 B absmax)
                hbm.setdefault('dequant4_bytes', 0.0)
                hbm.setdefault('dequant4_ms', 0.0)
                hbm['dequant4_bytes'] += 2.5625 * meta[0] * meta[1] * len(v)
                hbm['dequant4_ms'] += sum(v)
        if 'dequant4_ms' in hbm:
            hbm['dequant4'] = round(hbm.pop('dequant4_bytes') / (hbm.pop('dequant4_ms') * 1e-3) / 1e9, 1)
        result['hbm_bound_GBps'] = hbm
        if not args.no_cpu_baseline and world == 1:          # CPU leg: rank 0 of the single-GPU run only
            result['cpu_baseline'] = cpu_baseline(weights, H, kind, L, E, args.seq_len,
                                                  min(args.cpu_sample_tokens, T))
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
