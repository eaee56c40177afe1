#!/usr/bin/env python
"""Headline benchmark: residues/s of the packed ESM-2 forward on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--model esm2_650m] [--tokens 50000]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Launched plainly with --gpus N > 1 (no RANK in the environment) it re-executes itself under
`torch.distributed.run` with N ranks on 127.0.0.1 (one process per GPU, RCCL), so both forms
give an N-rank job; a world size that differs from --gpus is an error, never a silent 1-GPU run.

One step = one `model(tokens, (cu_lens, max_len))` forward (embedding -> L layers ->
final LN -> LM head -> (T, V) bf16 logits on device) over one synthetic packed batch
that is already resident in HBM.  Workload at N=1: BASELINE.json configs[2], the config
the metric is quoted on: ESM2-650M, 50 000 packed residues, uniform-500 (B=100 x S=500;
closed-form FLOPs).  With N>1 every rank runs its own 50 000-residue batch
(weak scaling, weights replicated, proteins never span GPUs); the timed step is the forward with the logits
left on the device, and the only collective of the path -- the all-gather of the (T_r, V) logits over
RCCL/xGMI -- is timed separately right after (`multi_gpu.gather_ms`).

Prints ONE JSON line (rank 0).  Extra objects:
  roofline     dominant kernel = the bf16 MFMA GEMM (94 % of the FLOPs); `achieved` is
               algorithmic FLOPs per launch / average launch duration of the FFN
               up-projection GEMM (M=T, N=4E, K=E, GELU epilogue), measured with HIP events
               on the launch stream in an instrumented pass; `all_gemms` aggregates every
               GEMM launch of a step the same way.
  multi_gpu    (launcher runs) RCCL world size actually seen, per-rank ms_per_step min / max, gather_ms.
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

_t_import = time.perf_counter()
import torch
_IMPORT_S = time.perf_counter() - _t_import

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
    ap.add_argument('--layers', type=int, default=0,
                    help='run only the first N layers of --model (e.g. a 15B-width model whose 48 layers do not fit the synthesis time budget); the metric label '
                         'and config say so: never the headline')
    ap.add_argument('--batch', choices=['uniform', 'proteome'], default='uniform')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-sample-tokens', type=int, default=4000)
    ap.add_argument('--no-gather', action='store_true', help='skip the (separately timed) logits all-gather when N>1')
    ap.add_argument('--dry-run', action='store_true',
                    help='CPU rehearsal of the N-rank bookkeeping (gloo, NO kernel runs, logits are zeros): launcher, barriers, '
                         'max-over-ranks timing, separate gather timing, JSON fields.  The line says "dry_run": true; never a result.')
    ap.add_argument('--auto-graph', action='store_true',
                    help='replay from a hipGraph when the forward is launch-bound (T x L <= 400 000)')
    ap.add_argument('--high-precision', action='store_true',
                    help="model.set_precision('high'): fp32 residual stream (not the headline mode; see DESIGN.md section 4)")
    ap.add_argument('--precision', choices=['fast', 'high', 'half', 'exact'], default=None,
                    help="model.set_precision(...): 'fast' = the headline mode (bf16 storage at the reference's rounding points); 'high' = fp32 "
                         "residual stream; 'half' = fp16-pair residual stream + IEEE fp16 MFMA operands, fp32 logits within ~4e-4 of the fp32 forward at "
                         "~1.1x the time (the line then says dtype f16); 'exact' = split (hi, lo) bf16 operand pairs, fp32 logits: the "
                         "reference's fp32 forward to ~1e-5 at ~2.2x the time (DESIGN.md section 4).  Not the headline; the line says which mode ran.")
    ap.add_argument('--no-half', action='store_true',
                    help="skip the extra leg of the default single-GPU run that also times model.set_precision('half') on the same batch "
                         "(a few steps after the timed region; reported under 'precision_half', never in 'value')")
    ap.add_argument('--spawn', action='store_true',
                    help='go through the torch.distributed.run self-launch even for --gpus 1 (exercises the RCCL '
                         'init + launcher path on a single-GPU box)')
    ap.add_argument('--graph', action='store_true',
                    help='replay the forward from a hipGraph (esme/graph.py); matters for small models / batches')
    ap.add_argument('--quantization', choices=['none', '4bit', '8bit'], default='none',
                    help="'4bit': layer projections resident in the esme-q4 format (not the headline config)")
    args = ap.parse_args()
    if args.precision is None:
        args.precision = 'high' if args.high_precision else 'fast'
    args.high_precision = args.precision == 'high'
    return args


def pmc_traffic(suffix=''):
    """(bytes, source): HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (newest
    profiles/rNN_traffic.json: FETCH_SIZE doubled as the gfx950 guide prescribes, + WRITE_SIZE).  Counters cannot be read
    from inside the process, so this is the last PROFILED value -- a static file, not an observation of this run -- valid
    for the default workload only; (None, None) otherwise."""
    for name in ((f'r06_traffic{suffix}.json', f'r05_traffic{suffix}.json') if suffix else ('r06_traffic.json', 'r05_traffic.json', 'r04_traffic.json', 'r03_traffic.json', 'r02_traffic.json')):
        path = os.path.join(ROOT, 'profiles', name)
        try:
            with open(path) as f:
                return int(json.load(f)['traffic_bytes_per_launch']), f'profiles/{name} (separate rocprofv3 --pmc passes of this workload; static file, not observed in this run)'
        except Exception:
            continue
    return None, None


def cpu_baseline(weights, heads, kind, L, E, seq_len, sample_tokens, gpu_logits=None, other_modes=None):
    """Oracle (port of the reference's CPU path) on a bounded sample of the same workload:
    `sample_tokens` residues in sequences of `seq_len`, all L layers + head, bf16.  The sample is the
    first `sample_tokens` residues of rank 0's batch (same generator stream), so when `gpu_logits`
    (the timed forward's output rows for those residues) is given the two are compared as well: the
    full-depth, full-batch forward is value-checked in the same run, not only isfinite-checked.  `other_modes` ({name: logits rows of
    the same residues from another precision mode}) are compared with the fp32-math oracle as well."""
    from oracle import esm_oracle as O
    from esme import synthetic as syn
    cores = os.cpu_count() or 1
    threads = min(cores, 64)
    torch.set_num_threads(threads)
    tokens, cu, max_len, lengths = syn.uniform_batch(sample_tokens, seq_len, seed=0)
    if gpu_logits is not None:
        gpu_logits, gpu_tokens = gpu_logits
        # the sample must be the same residues the GPU rows belong to (whole sequences of the same generator stream)
        if sample_tokens % seq_len != 0 or not torch.equal(gpu_tokens.cpu(), tokens):
            gpu_logits = None
    t0 = time.time()
    with torch.no_grad():
        out = O.forward_logits(weights, heads, tokens, cu, max_len, torch.bfloat16)
    dt = time.time() - t0
    assert out.shape[0] == sample_tokens
    res = {'value': round(sample_tokens / dt, 1), 'unit': 'residues/s', 'cores': threads, 'kind': 'port',
           'sample': f'{sample_tokens} residues ({len(lengths)} x {seq_len}) through all {L} layers + LM head, '
                     f'bf16 torch-CPU oracle, {dt:.1f} s'}
    parity = None
    if gpu_logits is not None:
        got = gpu_logits.float().cpu()
        ref = out.float()
        n32 = min(sample_tokens, 2 * seq_len)                 # fp32-math oracle on the first two sequences
        tok32, cu32, ml32, _ = syn.uniform_batch(n32, seq_len, seed=0)
        with torch.no_grad():
            ref32 = O.forward_logits(weights, heads, tok32, cu32, ml32, torch.float32).float()
        rel = lambda a, b: float((a - b).norm() / b.norm())
        parity = {'rows_vs_oracle_bf16': sample_tokens, 'rel_fro_hip_vs_oracle_bf16': round(rel(got, ref), 5),
                  'rows_vs_oracle_fp32': n32, 'rel_fro_hip_vs_oracle_fp32': round(rel(got[:n32], ref32), 5),
                  'rel_fro_oracle_bf16_vs_fp32': round(rel(ref[:n32], ref32), 5),
                  'max_abs_hip_vs_oracle_fp32': round(float((got[:n32] - ref32).abs().max()), 4)}
        for name, rows in (other_modes or {}).items():
            parity[f'rel_fro_{name}_vs_oracle_fp32'] = round(rel(rows.float().cpu()[:n32], ref32), 6)
    return res, parity


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: re-execute under torch.distributed.run with N ranks on
    this node (one process per GPU; backend nccl = RCCL over xGMI)."""
    import subprocess
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env['ESME_BENCH_SPAWNED'] = '1'
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


def rank_timing(dist, elapsed, steps, dev):
    """Every rank's wall time of the K timed steps -> the job's time (MAX over ranks) and the per-rank spread."""
    mine = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    every = torch.empty(dist.get_world_size(), dtype=torch.float64, device=dev)
    dist.all_gather_into_tensor(every, mine)
    every = every.cpu()
    return {'world_size_seen': dist.get_world_size(), 'elapsed_max': float(every.max()),
            'rank_ms_per_step': {'min': round(1e3 * float(every.min()) / steps, 3), 'max': round(1e3 * float(every.max()) / steps, 3)}}


def time_gather(dist, gathered, logits, steps, sync, dev):
    """The path's one collective, timed on its own: K all-gathers of the (T, V) logits between two
    barrier + synchronize fences, max over ranks."""
    dist.all_gather_into_tensor(gathered, logits)               # warm-up (communicator set-up)
    dist.barrier(); sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        dist.all_gather_into_tensor(gathered, logits)
    dist.barrier(); sync()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return {'gather_ms': round(1e3 * float(t.item()) / steps, 4)}


def dry_run(args, launched, rank, world):
    """CPU rehearsal of the multi-rank bookkeeping (gloo): no HIP kernel runs and nothing is measured -- the "forward" is a
    zero tensor of the logits' shape.  It exists so that the first real N > 1 run cannot trip over the launcher, the fences,
    the max-over-ranks reduction, the separately timed gather or the JSON fields (tests/test_host_cpu.py)."""
    import torch.distributed as dist
    from esme import synthetic as syn
    assert launched, 'dry run goes through the launcher'
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    dist.init_process_group('gloo')
    dev = torch.device('cpu')
    kind, L, E, H = syn.MODEL_ZOO[args.model]
    tokens, cu, max_len, lengths = syn.uniform_batch(args.tokens, args.seq_len, seed=rank)
    T, V = tokens.numel(), 33
    step = lambda: torch.zeros(T, V, dtype=torch.bfloat16)
    for _ in range(args.warmup):
        step()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    dist.barrier()
    elapsed = time.perf_counter() - t0
    multi = rank_timing(dist, elapsed, args.steps, dev)
    elapsed = multi.pop('elapsed_max')
    multi['backend'] = dist.get_backend()
    if not args.no_gather:
        multi.update(time_gather(dist, torch.empty(world * T, V, dtype=torch.bfloat16), out, args.steps, lambda: None, dev))
        multi['gather_bytes_per_rank'] = T * V * 2
    ms = 1e3 * elapsed / args.steps
    if 'gather_ms' in multi:
        multi['ms_per_step_incl_gather'] = round(ms + multi['gather_ms'], 3)
    if rank == 0:
        print(json.dumps({'dry_run': True, 'metric': 'DRY RUN (CPU, gloo, no kernel ran): ' + metric_label(args.model, T, args.batch),
                          'value': None, 'unit': 'residues/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
                          'ms_per_step': round(ms, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
                          'dtype': 'bf16', 'data': 'synthetic', 'config': {'workload': 'bookkeeping rehearsal', 'residues_per_gpu': T},
                          'multi_gpu': multi}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def metric_label(model, T, batch):
    if model == 'esm2_650m' and T == 50000:
        return 'residues/sec ESM2-650M fwd, 50k packed tokens; % bf16 MFMA peak; 1/2/4/8 GPU'      # BASELINE.json's metric
    return f'residues/sec {model} fwd, {T} packed tokens per GPU ({batch}); % bf16 MFMA peak (not the headline config)'


def half_leg(model, tokens, cu, max_len, steps, T, E, kind, lengths, flops_step, rooflines=True):
    """`steps` forwards of `model` (already in precision 'half') on the batch: wall clock between synchronisations, per-step HIP events on
    the launch stream, and -- from one instrumented pass afterwards -- the mode's own roofline object (its FFN-up launch on fp16 operands,
    all GEMMs, attention).  Returns the JSON fields + 'rows' (the logits)."""
    from esme import _hip
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    with torch.no_grad():
        for _ in range(2):
            out_h = model(tokens, (cu, max_len))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            ev[i][0].record()
            out_h = model(tokens, (cu, max_len))
            ev[i][1].record()
        torch.cuda.synchronize()
        h_ms = 1e3 * (time.perf_counter() - t0) / steps
    assert torch.isfinite(out_h).all()
    ems = sorted(s.elapsed_time(e) for s, e in ev)
    plan = model.half_plan()
    # the run-time plan guard (model.check_plan): the device maxima of every forward of this leg against the plan the mode ran with
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        verdict = model.check_plan(update=False)
    guard = ({'verdict': 'plan holds on this batch', 'checked': 'largest |value| of every stream channel after every branch of every layer vs the median channel '
                                                                   f'(threshold {model.HALF_CHANNEL_RATIO}x outside the extension tile); score bound of every layer without q/k pairs '
                                                                   f'(threshold {model.HALF_SCORE_BOUND})'}
             if verdict is None else {'verdict': 'STALE', 'channels': verdict['channels'][:8], 'layers': verdict['layers'][:8]})
    if getattr(model, '_half_guard', None) is None:
        guard = {'verdict': 'guard off'}
    res = {'what': "model.set_precision('half'): IEEE fp16 MFMA operands (the bf16 checkpoint converts exactly; LayerNorm gains folded as "
                   "powers of two, the rest rides on the stream), residual stream as an fp16 pair, split-operand LM head, fp32 logits; "
                   "attention in the fixed-reference form where the calibrated plan allows it (plan.fixed_reference_attention); "
                   "same batch, after the timed region",
           'dtype': 'f16', 'steps': steps, 'ms_per_step': round(h_ms, 3),
           'step_ms_events': {'median': round(ems[len(ems) // 2], 3), 'min': round(ems[0], 3), 'max': round(ems[-1], 3)},
           'value': round(T / (h_ms * 1e-3), 1), 'unit': 'residues/s',
           'frac_bf16_mfma_peak': round(flops_step / (h_ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
           'plan': {'extension_channels': 0 if plan.ext_sel is None else int(plan.ext_sel.numel()), 'qk_pairs': bool(plan.qk_pair),
                    **{k: (round(v, 2) if isinstance(v, float) else v) for k, v in plan.info.items()}, 'guard': guard},
           'rows': out_h}
    if rooflines:
        with torch.no_grad():
            _hip.TRACE = []
            model(tokens, (cu, max_len))
            torch.cuda.synchronize()
            trace, _hip.TRACE = _hip.TRACE, None
        up = [s.elapsed_time(e) for op, meta, s, e in trace if op == 'gemm' and meta[1] == 4 * E and str(meta[3]).startswith('f16:')]
        g_ms = sum(s.elapsed_time(e) for op, meta, s, e in trace if op == 'gemm')
        g_fl = sum(2.0 * meta[0] * meta[1] * (E if meta[2] == E + 64 else meta[2]) for op, meta, s, e in trace if op == 'gemm')    # (an extension K-tile is overhead, not work)
        if up and kind != 'esmc':
            fl = 2.0 * T * 4 * E * E                     # algorithmic FLOPs of the FFN up-projection (an extension K-tile is overhead, not work)
            ms = sum(up) / len(up)
            res['roofline'] = {'bound': 'mfma', 'kernel': f'gemm_bf16_kernel<F16> M={T} N={4 * E} K={E} (FFN up, LN-folded, GELU epilogue, fp16 operands)',
                               'achieved': round(fl / (ms * 1e-3) / 1e12, 1), 'peak': PEAK_BF16_TFLOPS, 'unit': 'TFLOP/s',
                               'frac': round(fl / (ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4), 'traffic': None, 'traffic_source': None,
                               'avg_launch_ms': round(ms, 4), 'launches_timed': len(up),
                               'all_gemms': {'achieved': round(g_fl / (g_ms * 1e-3) / 1e12, 1), 'frac': round(g_fl / (g_ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4)}}
            if E == 1280 and T == 50000 and max_len == 500:          # the headline workload: the mode's own PMC passes (a static file, like the fast leg's)
                res['roofline']['traffic'], res['roofline']['traffic_source'] = pmc_traffic('_half')
        at = [(meta, s.elapsed_time(e)) for op, meta, s, e in trace if op in ('attn', 'attn_qkpair')]
        if at:
            ams = sum(v for _, v in at) / len(at)
            afl = 4.0 * E * sum(x * x for x in lengths)
            res['attention'] = {'achieved': round(afl / (ams * 1e-3) / 1e12, 1), 'unit': 'TFLOP/s (algorithmic: 4 S E per residue)',
                                'frac': round(afl / (ams * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4), 'avg_launch_ms': round(ams, 4)}
    return res


class PhaseClock:
    """Wall-clock seconds of the UNTIMED parts of a run (checkpoint synthesis, load, setup forwards, instrumented pass, the secondary legs,
    the CPU legs), reported as `untimed_s` so that the line says where a slow host spent the driver's time."""

    def __init__(self):
        self.t0 = self.last = time.perf_counter()
        self.phases = {}

    def mark(self, name):
        now = time.perf_counter()
        self.phases[name] = round(self.phases.get(name, 0.0) + now - self.last, 2)
        self.last = now

    def report(self, timed_s):
        self.mark('other')
        out = dict(self.phases)
        out['timed_region'] = round(timed_s, 2)
        out['total_since_main'] = round(time.perf_counter() - self.t0, 2)
        out['import_torch_before_main'] = round(_IMPORT_S, 2)
        return out


def main():
    args = parse()
    clock = PhaseClock()
    launched = 'RANK' in os.environ                 # under torch.distributed.run (driver's N>1 form, or self_launch)
    if not launched and (args.gpus > 1 or args.spawn or args.dry_run):
        self_launch(args)
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a mislabelled run')
    dist = None
    if args.dry_run:
        return dry_run(args, launched, rank, world)
    if launched:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if torch.cuda.device_count() < world:
            raise SystemExit(f'--gpus {args.gpus} but only {torch.cuda.device_count()} HIP devices are visible')
        torch.cuda.set_device(local_rank)
        dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f'process group has {dist.get_world_size()} ranks, --gpus {args.gpus}')
    else:
        torch.cuda.set_device(0)
    dev = torch.device('cuda', local_rank if launched else 0)

    from esme import ESM, _hip, synthetic as syn
    _hip.load()
    kind, L, E, H = syn.MODEL_ZOO[args.model]
    full_L = L
    if args.layers:
        L = min(L, args.layers)

    # ---- synthetic checkpoint in the reference layout -> from_pretrained
    weights = syn.synthetic_state_dict(kind, L, E, seed=0)
    clock.mark('synthesise_checkpoint')
    with tempfile.TemporaryDirectory() as td:
        from safetensors.torch import save_file
        path = os.path.join(td, f'{args.model}.safetensors')
        save_file(weights, path, metadata=syn.checkpoint_metadata(args.model, L, E, H))
        model = ESM.from_pretrained(path, quantization=None if args.quantization == 'none' else args.quantization,
                                    device=str(dev))
    if args.precision != 'fast':
        model.set_precision(args.precision)
    clock.mark('write_and_load_checkpoint')

    # ---- this rank's packed batch (resident in HBM before the timed region)
    if args.batch == 'uniform':
        tokens, cu, max_len, lengths = syn.uniform_batch(args.tokens, args.seq_len, seed=rank)
    else:
        tokens, cu, max_len, lengths = syn.proteome_batch(args.tokens, seed=rank)
    tokens, cu = tokens.to(dev), cu.to(dev)
    T = tokens.numel()
    V = model.vocab_size
    use_graph = args.graph or (args.auto_graph and T * L <= 400_000)

    def step():                                    # the timed step: logits materialised on the device
        return model.graphed(tokens, (cu, max_len), 'forward', clone=False) if use_graph else model(tokens, (cu, max_len))

    def fence():
        if launched:
            dist.barrier()
        torch.cuda.synchronize()

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    with torch.no_grad():
        # one-time setup, not part of the W warm-up steps: the first two forwards pack / fold the weights into
        # their GEMM layouts, load the kernel modules and grow the allocator pools (338 ms and ~145 ms vs 75 ms)
        for _ in range(2):
            step()
        fence()
        clock.mark('setup_forwards')
        for i in range(args.warmup):
            step()
        fence()
        t0 = time.perf_counter()
        for i in range(args.steps):
            ev[i][0].record()                     # HIP events on the launch stream: per-step device time
            out = step()
            ev[i][1].record()
        fence()
        elapsed = time.perf_counter() - t0
    clock.mark('warmup_and_timed_steps')
    timed_s = elapsed
    step_ms = sorted(s.elapsed_time(e) for s, e in ev)
    multi = None
    if launched:
        multi = rank_timing(dist, elapsed, args.steps, dev)
        elapsed = multi.pop('elapsed_max')
        multi['backend'] = f'{dist.get_backend()} (RCCL over xGMI)' if dist.get_backend() == 'nccl' else dist.get_backend()
        if not args.no_gather:
            gathered = torch.empty(world * T, V, dtype=out.dtype, device=dev)
            multi.update(time_gather(dist, gathered, out, args.steps, lambda: torch.cuda.synchronize(), dev))
            multi['gather_bytes_per_rank'] = T * V * out.element_size()
    assert torch.isfinite(out.float()).all()
    ms_per_step = 1e3 * elapsed / args.steps
    value = world * T * args.steps / elapsed
    flops_step = syn.algorithmic_flops(kind, L, E, lengths)

    result = {
        'metric': metric_label(args.model, T, args.batch) + (f' [first {L} of {full_L} layers only]' if L != full_L else ''),
        'value': round(value, 1), 'unit': 'residues/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 3), 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f16' if args.precision == 'half' else 'bf16', 'data': 'synthetic',
        'config': {'workload': f'{args.model} packed forward -> logits, {T} residues/GPU, '
                               f'{args.batch} batch ({len(lengths)} seqs, max_len {max_len})',
                   'residues_per_gpu': T, 'sequences_per_gpu': len(lengths), 'max_len': max_len,
                   'parallelism': f'dp{world} (protein-sharded, logits all-gather over RCCL)' if world > 1 else 'single GPU',
                   'launch': 'hipGraph replay' if use_graph else
                             ('eager (one C call for the layer stack: esme_hip_forward' + ('_half)' if args.precision == 'half' else ')')
                              if (model.c_forward and args.precision in ('fast', 'half') and model._c_forward_ok(args.precision))
                              else 'eager (one ctypes launch per kernel)'),
                   'launcher': 'torch.distributed.run (self-launched)' if os.environ.get('ESME_BENCH_SPAWNED') else
                               ('torch.distributed.run' if launched else 'plain python'),
                   'precision': {'fast': 'fast (bf16 residual stream)', 'high': 'high (fp32 residual stream)',
                                 'half': 'half (fp16-pair residual stream, IEEE fp16 MFMA operands converted exactly from the bf16 weights, '
                                         'split-operand LM head, fp32 logits)',
                                 'exact': 'exact (split (hi, lo) bf16 operand pairs, fp32 residual stream, fp32 logits; 2 MFMA passes per '
                                          'projection, 3 per attention product)'}[args.precision],
                   'setup': '2 untimed forwards before the warm-up steps (weight packing / LN folding, module load)',
                   'weights': 'synthetic (numpy PCG64), reference checkpoint layout'
                              + ('' if args.quantization == 'none' else f', layer projections {args.quantization} (esme/quantization.py)')},
        'step_ms_events': {'median': round(step_ms[len(step_ms) // 2], 3), 'min': round(step_ms[0], 3), 'max': round(step_ms[-1], 3)},
        'e2e': {'algorithmic_tflop_per_step': round(flops_step / 1e12, 3),
                'tflops_per_gpu': round(flops_step / (ms_per_step * 1e-3) / 1e12, 1),
                'frac_bf16_mfma_peak': round(flops_step / (ms_per_step * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4)},
    }

    if multi is not None:
        result['multi_gpu'] = multi
        if 'gather_ms' in multi:
            result['multi_gpu']['ms_per_step_incl_gather'] = round(ms_per_step + multi['gather_ms'], 3)
            # (rounds 1-2 timed the gather inside the step: the comparable whole-job rate under that definition)
            result['multi_gpu']['value_incl_gather'] = round(world * T / ((ms_per_step + multi['gather_ms']) * 1e-3), 1)
    if rank == 0:
        # ---- instrumented pass: HIP events around every launch on the launch stream
        with torch.no_grad():
            _hip.TRACE = []
            for _ in range(max(1, min(args.steps, 3))):
                model(tokens, (cu, max_len))
            torch.cuda.synchronize()
            trace, _hip.TRACE = _hip.TRACE, None
        per_op = {}
        for op, meta, s, e in trace:
            per_op.setdefault((op, meta), []).append(s.elapsed_time(e))
        key = ('gemm', (T, 4 * E if kind != 'esmc' else 2 * syn.swiglu_width(E), E,
                        _hip.EPI_GELU if kind != 'esmc' else _hip.EPI_SWIGLU))
        if args.precision == 'exact':              # the same launch in pair form: K doubled ([hi | lo]); MFMA FLOPs are counted as executed
            key = ('gemm', (T, 4 * E, 2 * E, f'split:{_hip.EPI_GELU}'))
        if args.precision == 'half':               # the same launch on fp16 operands
            key = ('gemm', (key[1][0], key[1][1], key[1][2], f'f16:{key[1][3]}'))
        traffic, traffic_src = pmc_traffic() if (args.model == 'esm2_650m' and T == 50000 and args.precision == 'fast'
                                                 and args.quantization == 'none') else (None, None)
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
                'frac': round(fl / (ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
                'traffic': traffic, 'traffic_source': traffic_src,
                'avg_launch_ms': round(ms, 4), 'launches_timed': len(per_op[key]),
                'all_gemms': {'achieved': round(g_fl / (g_ms * 1e-3) / 1e12, 1),
                              'frac': round(g_fl / (g_ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4)},
            }
        for (op, meta), v in per_op.items():
            if op in ('attn', 'attn_split'):     # 4*S*E flops per residue per launch (QK^T + PV); the split form executes 3x that on the MFMAs
                afl = 4.0 * (E // meta[1]) * meta[1] * sum(s * s for s in lengths) * (3.0 if op == 'attn_split' else 1.0)
                ams = sum(v) / len(v)
                result['attention'] = {'achieved': round(afl / (ams * 1e-3) / 1e12, 1), 'unit': 'TFLOP/s',
                                       'frac': round(afl / (ams * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
                                       'avg_launch_ms': round(ams, 4), 'head_dim': meta[2]}
        nsteps = max(1, min(args.steps, 3))
        by_op = {}
        for (op, meta), v in per_op.items():
            by_op[op] = by_op.get(op, 0.0) + sum(v) / nsteps
        result['kernel_ms_per_step'] = {k: round(v, 3) for k, v in sorted(by_op.items())}
        result['kernel_ms_per_step_source'] = ('instrumented module-by-module pass (HIP events around every launch, after the timed '
                                               'region); the timed steps go through one esme_hip_forward call, so the sum may exceed ms_per_step')
        # HBM-bound kernels: algorithmic bytes / measured time
        hbm = {}
        for (op, meta), v in per_op.items():
            if op == 'layernorm' and meta == (T, E):
                hbm['layernorm'] = round(4.0 * E * T / (sum(v) / len(v) * 1e-3) / 1e9, 1)
            if op == 'rotary':
                hbm['rotary'] = round(8.0 * E * T / (sum(v) / len(v) * 1e-3) / 1e9, 1)
            if op == 'qk_norm_rotary':
                hbm['qk_norm_rotary'] = round(8.0 * E * T / (sum(v) / len(v) * 1e-3) / 1e9, 1)
            if op == 'dequant4':        # 0.5 B code + 2 B bf16 per weight (+ 1/16 B absmax)
                hbm.setdefault('dequant4_bytes', 0.0)
                hbm.setdefault('dequant4_ms', 0.0)
                hbm['dequant4_bytes'] += 2.5625 * meta[0] * meta[1] * len(v)
                hbm['dequant4_ms'] += sum(v)
        if 'dequant4_ms' in hbm:
            hbm['dequant4'] = round(hbm.pop('dequant4_bytes') / (hbm.pop('dequant4_ms') * 1e-3) / 1e9, 1)
        result['hbm_bound_GBps'] = hbm
        clock.mark('instrumented_pass')
        # ---- the same batch through precision 'half' (fp16 MFMA operands, fp16-pair residual stream, fp32 logits): the mode that meets
        # north_star's 1e-3 in one pass.  After the timed region, its own fences, per-step HIP events and its own roofline object; reported
        # beside the headline, never as `value`.  Then the same on the ill-conditioned probe model (massive stream channels: the regime
        # trained checkpoints are known for), where the mode's calibration switches its robustness measures on -- time AND parity for both.
        half_rows = None
        outlier = None
        if args.precision == 'fast' and world == 1 and not args.no_half and not use_graph and args.quantization == 'none':
            try:
                model.set_precision('half')
                hs = max(1, min(args.steps, 5))
                h = half_leg(model, tokens, cu, max_len, hs, T, E, kind, lengths, flops_step)
                half_rows = h.pop('rows')
                h['vs_fast_mode'] = round(h['ms_per_step'] / ms_per_step, 4)
                result['precision_half'] = h
            except Exception as e:           # layouts the mode does not cover (4-bit weights) -- and nothing in this secondary leg may cost the headline line
                result['precision_half'] = {'skipped': f'{type(e).__name__}: {e}'[:300]}
            finally:
                model.set_precision('fast')
            clock.mark('half_leg')
            if half_rows is not None and kind == 'esm2' and not args.no_cpu_baseline:
                # the probe model is the SAME synthetic state dict with 1 + 2L small tensors patched (massive_channel_state_dict(base=...)): the
                # loaded model is patched in place and restored afterwards -- no second 650 M-parameter synthesis, no second checkpoint file
                patched = syn.massive_channel_patched_names(L)
                try:
                    w_out, _ = syn.massive_channel_state_dict(L, E, 50.0, seed=0, base=weights)
                    model.load_state_dict({n: w_out[n] for n in patched}, strict=False)
                    model.set_precision('half')
                    h2 = half_leg(model, tokens, cu, max_len, max(1, min(hs, 3)), T, E, kind, lengths, flops_step, rooflines=False)
                    outlier = (w_out, h2.pop('rows'))
                    h2['weights'] = ('synthetic + 4 massive stream channels (embedding columns and FFN-down biases x 50, two attention-LayerNorm '
                                     'gains x 10: esme.synthetic.massive_channel_state_dict), the probe model of tools/half_outlier_probe.py')
                    h2['vs_fast_mode'] = round(h2['ms_per_step'] / ms_per_step, 4)
                    result['precision_half']['outlier_model'] = h2
                except Exception as e:       # (a secondary leg: report, never fail the run)
                    outlier = None
                    result['precision_half']['outlier_model'] = {'skipped': f'{type(e).__name__}: {e}'[:300]}
                finally:
                    model.load_state_dict({n: weights[n] for n in patched}, strict=False)
                    model.set_precision('fast')
                clock.mark('half_leg_outlier_model')
        if not args.no_cpu_baseline and world == 1:          # CPU leg: rank 0 of the single-GPU run only
            n = min(args.cpu_sample_tokens, T)
            n = max(args.seq_len, n // args.seq_len * args.seq_len) if n >= args.seq_len else n      # whole sequences only
            same = args.batch == 'uniform' and args.quantization == 'none' and n <= T      # sample == first n residues of the batch
            result['cpu_baseline'], parity = cpu_baseline(weights, H, kind, L, E, args.seq_len, n,
                                                          (out[:n], tokens[:n]) if same else None,
                                                          {'half': half_rows[:n]} if (same and half_rows is not None) else None)
            if parity is not None:
                result['parity'] = parity
                if outlier is not None and same:            # precision 'half' on the massive-channel model vs ITS fp32-math oracle
                    try:
                        from oracle import esm_oracle as O
                        n32 = min(n, 2 * args.seq_len)
                        tok32, cu32, ml32, _ = syn.uniform_batch(n32, args.seq_len, seed=0)
                        with torch.no_grad():
                            ref32 = O.forward_logits(outlier[0], H, tok32, cu32, ml32, torch.float32).float()
                        got = outlier[1][:n32].float().cpu()
                        parity['rel_fro_half_outlier_model_vs_oracle_fp32'] = round(float((got - ref32).norm() / ref32.norm()), 6)
                    except Exception as e:
                        parity['rel_fro_half_outlier_model_vs_oracle_fp32'] = f'skipped: {type(e).__name__}'
            clock.mark('cpu_oracle_legs')
        result['untimed_s'] = clock.report(timed_s)
        print(json.dumps(result), flush=True)
    if launched:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
