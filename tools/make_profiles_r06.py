#!/usr/bin/env python
"""Turn gpurun_out/r06/* (tools/final_measure_r06.sh) into the committed profiles/r06_* files."""
import csv, glob, json, os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.join(ROOT, 'gpurun_out', 'r06', 'final')
P = os.path.join(ROOT, 'profiles')


def load(name):
    try:
        txt = open(os.path.join(O, name)).read()
        line = [l for l in txt.splitlines() if l.startswith('{')][-1]
        return json.loads(line)
    except Exception as e:
        print('missing', name, e)
        return None


def short(name):
    return name.replace('void esme::', '').replace('esme::', '').split('(')[0][:120]


def kernel_stats(sub='prof', out='r06_kernel_stats.md', flags=''):
    f = sorted(glob.glob(os.path.join(O, sub, '**', '*kernel_stats.csv'), recursive=True), key=os.path.getmtime, reverse=True)
    if not f:
        return                                   # (newest run first: gpurun_out/ accumulates the runs of a round)
    rows = list(csv.DictReader(open(f[0])))
    tot = sum(float(r['TotalDurationNs']) for r in rows)
    lines = [f'# rocprofv3 --kernel-trace --stats -- python bench.py{flags} --steps 3 --warmup 1 --no-cpu-baseline [--no-half] (round 6, headline workload)', '',
             f'source: `{os.path.relpath(f[0], ROOT)}`; durations in microseconds; 1 + 2 + 3 + instrumented 3 forwards = 9 forwards',
             '', '| kernel | calls | total us | avg us | % |', '|---|---:|---:|---:|---:|']
    for r in rows:
        pct = 100 * float(r['TotalDurationNs']) / tot
        if pct < 0.02:
            continue
        lines.append(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['TotalDurationNs']) / 1e3:.0f} | {float(r['AverageNs']) / 1e3:.1f} | {pct:.2f} |")
    open(os.path.join(P, out), 'w').write('\n'.join(lines) + '\n')
    print('\n'.join(lines[:16]))


def pmc(dirs, out, title):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in dirs:
        for f in sorted(glob.glob(os.path.join(O, d, '**', '*counter_collection.csv'), recursive=True), key=os.path.getmtime, reverse=True)[:1]:
            for row in csv.DictReader(open(f)):
                agg[short(row['Kernel_Name'])][row['Counter_Name']].append(float(row['Counter_Value']))
    counters = sorted({c for k in agg.values() for c in k})
    lines = [f'# {title}', '', 'rocprofv3 --pmc passes of `python bench.py --steps 2 --warmup 1 --no-cpu-baseline`, mean per dispatch; sources: '
             + ', '.join(f'gpurun_out/r06/final/{d}' for d in dirs), '', '| kernel | dispatches | ' + ' | '.join(counters) + ' |', '|---|---:|' + '---:|' * len(counters)]
    for k, cs in sorted(agg.items(), key=lambda kv: -sum(kv[1].get(counters[0], [0]))):
        if not k.startswith(('gemm', 'attn', 'layernorm', 'rotary', 'embed', 'softmax', 'row_sums')):
            continue
        n = max(len(v) for v in cs.values())
        lines.append(f'| `{k}` | {n} | ' + ' | '.join(f'{sum(cs[c]) / len(cs[c]):.4g}' if c in cs else '' for c in counters) + ' |')
    open(os.path.join(P, out), 'w').write('\n'.join(lines) + '\n')
    return agg


def main():
    b = load('bench.json')
    if b:
        json.dump(b, open(os.path.join(P, 'r06_bench.json'), 'w'), indent=1)
        print('headline', b['value'], b['ms_per_step'], b['e2e'], b['roofline']['frac'], b.get('attention'))
    kernel_stats()
    kernel_stats('prof_half', 'r06_kernel_stats_half.md', ' --precision half')
    for src, dst in (('bench_half.json', 'r06_bench_half.json'), ('bench_exact.json', 'r06_bench_exact.json')):
        d = load(src)
        if d:
            json.dump(d, open(os.path.join(P, dst), 'w'), indent=1)
    agg = pmc(['pmc_fetch', 'pmc_write'], 'r06_pmc_traffic.md', 'HBM / fabric traffic counters (FETCH_SIZE, WRITE_SIZE in KiB)')
    key = [k for k in agg if k.startswith('gemm_bf16_kernel<256, 256, 2, 4, 1, 0, true')]
    if key:
        f, w = agg[key[0]].get('FETCH_SIZE'), agg[key[0]].get('WRITE_SIZE')
        if f and w:
            fk, wk = sum(f) / len(f), sum(w) / len(w)
            traffic = int((2 * fk + wk) * 1024)
            json.dump({'_comment': 'HBM/fabric bytes per launch of the dominant kernel (FFN-up GEMM) from the round-5 rocprofv3 PMC passes (gpurun_out/r06/final/pmc_fetch, pmc_write: profiles/r06_pmc_traffic.md); FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half the bytes of wide coalesced reads); KiB units',
                       'kernel': key[0] + ' M=50000 N=5120 K=1280', 'fetch_size_kib_raw': fk, 'write_size_kib': wk,
                       'traffic_bytes_per_launch': traffic, 'algorithmic_bytes_per_launch': 653107200}, open(os.path.join(P, 'r06_traffic.json'), 'w'), indent=1)
            print('traffic', traffic)
    # precision 'half': its own passes (bench.py's half leg reads r06_traffic_half.json)
    aggh = pmc(['pmc_fetch_half', 'pmc_write_half'], 'r06_pmc_traffic_half.md', "HBM / fabric traffic counters in precision 'half' (FETCH_SIZE, WRITE_SIZE in KiB; `bench.py --precision half`)")
    keyh = [k for k in aggh if k.startswith('gemm_bf16_kernel<256, 256, 2, 4, 1, 0, true, false, true, false, false, true')]
    if keyh and aggh[keyh[0]].get('FETCH_SIZE') and aggh[keyh[0]].get('WRITE_SIZE'):
        f, w = aggh[keyh[0]]['FETCH_SIZE'], aggh[keyh[0]]['WRITE_SIZE']
        fk, wk = sum(f) / len(f), sum(w) / len(w)
        json.dump({'_comment': "HBM/fabric bytes per launch of precision 'half''s dominant kernel (FFN-up GEMM, fp16 operands, LN-folded, GELU) from the round-5 rocprofv3 PMC passes of "
                               "`bench.py --precision half`; FETCH_SIZE doubled per MI355X_MICROARCH.md; KiB units",
                   'kernel': keyh[0] + ' M=50000 N=5120 K=1280', 'fetch_size_kib_raw': fk, 'write_size_kib': wk,
                   'traffic_bytes_per_launch': int((2 * fk + wk) * 1024), 'algorithmic_bytes_per_launch': 653107200}, open(os.path.join(P, 'r06_traffic_half.json'), 'w'), indent=1)
    if glob.glob(os.path.join(O, 'pmc_sq_half', '**', '*counter_collection.csv'), recursive=True):
        sqh = pmc(['pmc_sq_half'], 'r06_pmc_counters_half.md', "SQ / MFMA counters per kernel in precision 'half' (round 6)")
        mean_ = lambda v: sum(v) / len(v)
        hl = ['', '## Derived (mean per dispatch)', '', '| kernel | launch cycles (GRBM_GUI_ACTIVE / 8) | matrix pipe busy (SQ_VALU_MFMA_BUSY_CYCLES / 1 024 / launch cycles) | SQ_WAIT_ANY / SQ_WAVE_CYCLES |', '|---|---:|---:|---:|']
        for k, cs in sorted(sqh.items(), key=lambda kv: -sum(kv[1].get('GRBM_GUI_ACTIVE', [0]))):
            if not k.startswith(('gemm', 'attn')) or any(c not in cs for c in ('GRBM_GUI_ACTIVE', 'SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_WAIT_ANY', 'SQ_WAVE_CYCLES')):
                continue
            cyc = mean_(cs['GRBM_GUI_ACTIVE']) / 8
            hl.append(f"| `{k}` | {cyc:,.0f} | {100 * mean_(cs['SQ_VALU_MFMA_BUSY_CYCLES']) / 1024 / cyc:.1f} % | {100 * mean_(cs['SQ_WAIT_ANY']) / mean_(cs['SQ_WAVE_CYCLES']):.1f} % |")
        open(os.path.join(P, 'r06_pmc_counters_half.md'), 'a').write('\n'.join(hl) + '\n')
    sq = pmc(['pmc_sq', 'pmc_sq2'], 'r06_pmc_counters.md', 'SQ / LDS / MFMA / L2 counters per kernel (round 6)')
    # derived ratios (1 024 SIMDs, 8 XCDs: GRBM_GUI_ACTIVE is summed over the XCDs)
    mean = lambda v: sum(v) / len(v)
    lines = ['', '## Derived (mean per dispatch)', '',
             '| kernel | launch cycles (GRBM_GUI_ACTIVE / 8) | matrix pipe busy (SQ_VALU_MFMA_BUSY_CYCLES / 1 024 / launch cycles) | SQ_WAIT_ANY / SQ_WAVE_CYCLES | '
             'LDS bank-conflict cycles / LDS cycles | L2 hit rate |', '|---|---:|---:|---:|---:|---:|']
    for k, cs in sorted(sq.items(), key=lambda kv: -sum(kv[1].get('GRBM_GUI_ACTIVE', [0]))):
        need = ('GRBM_GUI_ACTIVE', 'SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_WAIT_ANY', 'SQ_WAVE_CYCLES', 'SQ_LDS_BANK_CONFLICT', 'SQ_LDS_IDX_ACTIVE', 'TCC_HIT_sum', 'TCC_MISS_sum')
        if not k.startswith(('gemm', 'attn')) or any(c not in cs for c in need):
            continue
        cyc = mean(cs['GRBM_GUI_ACTIVE']) / 8
        lds = mean(cs['SQ_LDS_IDX_ACTIVE'])
        lines.append(f"| `{k}` | {cyc:,.0f} | {100 * mean(cs['SQ_VALU_MFMA_BUSY_CYCLES']) / 1024 / cyc:.1f} % | "
                     f"{100 * mean(cs['SQ_WAIT_ANY']) / mean(cs['SQ_WAVE_CYCLES']):.1f} % | "
                     f"{100 * mean(cs['SQ_LDS_BANK_CONFLICT']) / lds if lds else 0:.1f} % | "
                     f"{100 * mean(cs['TCC_HIT_sum']) / (mean(cs['TCC_HIT_sum']) + mean(cs['TCC_MISS_sum'])):.1f} % |")
    open(os.path.join(P, 'r06_pmc_counters.md'), 'a').write('\n'.join(lines) + '\n')
    rows = [('ESM2-650M, 50 000 residues, 100 x 500 (headline, configs[2])', 'bench.json'),
            ('same, self-launched through torch.distributed.run (--gpus 1 --spawn, RCCL world 1)', 'bench_spawn.json'),
            ('ESM2-650M, 50 000 residues, proteome-like lengths', 'bench_proteome.json'),
            ("ESM2-650M, 50 000 residues, precision 'half' (fp16 MFMA operands, power-of-two LayerNorm fold, fp16-pair residual stream, fp32 logits: 3.6e-4 vs the fp32 forward)", 'bench_half.json'),
            ("same, proteome-like lengths", 'bench_half_proteome.json'),
            ("ESM2-3B, 50 000 residues, precision 'half'", 'bench_half_3b.json'),
            ("ESMC-600M, 32 x 1 002 residues, precision 'half' (7.7e-4 vs the fp32 forward)", 'bench_half_esmc600m.json'),
            ("ESM2-150M, 8 192 residues, precision 'half'", 'bench_half_150m.json'),
            ("ESM2-650M, 50 000 residues, split-operand mode (precision 'exact': fp32 logits, 5.8e-6 vs the fp32 forward; one C call, rotary in the pair epilogue)", 'bench_exact.json'),
            ("ESMC-600M, 32 x 1 002 residues, split-operand mode", 'bench_exact_esmc600m.json'),
            ('ESM2-650M, 50 000 residues, 4-bit (esme-q4) layer weights', 'bench_650m_q4.json'),
            ('ESM-1b (learned positions), 50 000 residues', 'bench_esm1b.json'),
            ('ESM2-3B, 50 000 residues (configs[3] per-GPU share)', 'bench_3b.json'),
            ('ESMC-600M, 32 064 residues = 32 x 1 002 (configs[4] batch shape)', 'bench_esmc600m.json'),
            ('ESM2-150M, 8 192 residues = 16 x 512 (configs[1]), eager, one C call per forward (esme_hip_forward)', 'bench_150m.json'),
            ("15B-WIDTH model: ESM2-15B's geometry (E = 5 120, 40 heads of 128), first 4 of 48 layers, 50 000 residues (head dim 128: first-generation attention kernel)", 'bench_15b_width.json'),
            ("same, precision 'half'", 'bench_half_15b_width.json')]
    lines = ['# bench.py on other workloads (1 x MI355X, round 6; separate gpurun boxes differ by +-3 %)', '',
             '| workload | residues/s | ms/step | % of 2.5 PF bf16 peak (algorithmic FLOPs) | kernel ms per step |', '|---|---:|---:|---:|---|']
    for label, f in rows:
        d = load(f)
        if d:
            lines.append(f"| {label} | {d['value']:,.0f} | {d['ms_per_step']:.2f} | {100 * d['e2e']['frac_bf16_mfma_peak']:.1f} % | {d.get('kernel_ms_per_step')} |")
    open(os.path.join(P, 'r06_other_configs.md'), 'w').write('\n'.join(lines) + '\n')
    print('\n'.join(lines))
    att = ['# Attention kernel variants, head dim 64 (tools/attn_lab.py; variant 1 = round-1 kernel, 4 / 8 = ping-pong with 4 / 8 waves)', '']
    for f in ('attn_lab_qp_uniform.txt',):
        try:
            att += ['```', open(os.path.join(O, f)).read().strip(), '```', '']
        except Exception:
            pass
    open(os.path.join(P, 'r06_attention.md'), 'w').write('\n'.join(att))
    for src, dst in (('half_outlier_probe_33x1280.txt', None), ('half_outlier_probe_12x640.txt', None)):
        pass
    try:
        txt = ['# tools/half_outlier_probe.py (round 6): logits rel-Frobenius vs the fp32 oracle with MASSIVE stream channels (4 embedding columns and the matching',
               '# FFN-down biases x scale, two attention-LayerNorm gains x min(scale, 10)).  "half" = the calibrated form (esme.attention.HalfPlan), "plain form" = robust=False',
               '# (round 4 + the power-of-two LayerNorm fold).  First block 12 layers x 640, second 33 x 1280.']
        for f in ('half_outlier_probe_12x640.txt', 'half_outlier_probe_33x1280.txt'):
            txt += [l for l in open(os.path.join(O, f)).read().splitlines() if l.startswith('outlier scale')]
        open(os.path.join(P, 'r06_half_outlier_probe.txt'), 'w').write('\n'.join(txt) + '\n')
    except Exception as e:
        print('missing outlier probe', e)
    for src, dst in (('proteome_e2e.json', 'r06_proteome_e2e.json'), ('proteome_e2e_half.json', 'r06_proteome_e2e_half.json'),
                     ('half_robust_breakdown.json', 'r06_half_robust_breakdown.json'), ('bench_half_proteome.json', None)):
        try:
            if dst:
                open(os.path.join(P, dst), 'w').write(open(os.path.join(O, src)).read())
        except Exception as e:
            print('missing', src, e)
    for src, dst in (('qk_norm_bench.txt', 'r06_qk_norm_bench.txt'), ('tail_round_probe.txt', 'r06_gemm_tail_round_probe.txt'),
                     ('half_guard_cost.txt', 'r06_half_guard_cost_final.txt'), ('half_guard_cost_proteome.txt', 'r06_half_guard_cost_proteome.txt'),
                     ('attn_ragged_table.txt', 'r06_attn_ragged_table.txt'), ('attn_sb_bench.txt', 'r06_attn_sb_bench.txt'),
                     ('attn_lab_d128_s500.txt', 'r06_attn_d128_s500.txt'), ('attn_lab_d128_s2000.txt', 'r06_attn_d128_s2000.txt'),
                     ('token_outlier_guard.txt', 'r06_half_token_outlier_guard.txt'), ('token_outlier_vocab.txt', 'r06_half_token_outlier_vocab.txt')):
        try:
            txt = [l for l in open(os.path.join(O, src)).read().splitlines() if 'amdgpu.ids' not in l]
            open(os.path.join(P, dst), 'w').write('\n'.join(txt) + '\n')
        except Exception as e:
            print('missing', src, e)


if __name__ == '__main__':
    main()
