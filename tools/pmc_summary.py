#!/usr/bin/env python
"""Aggregate rocprofv3 --pmc counter_collection.csv files per kernel (mean per dispatch).
usage: pmc_summary.py out.md dir1 [dir2 ...]"""
import csv, glob, os, sys, collections

def short(name):
    name = name.replace('void esme::', '').replace('esme::', '')
    return name.split('(')[0][:60]

def main(out, *dirs):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in dirs:
        for f in glob.glob(os.path.join(d, '*counter_collection.csv')):
            for row in csv.DictReader(open(f)):
                agg[short(row['Kernel_Name'])][row['Counter_Name']].append(float(row['Counter_Value']))
    counters = sorted({c for k in agg.values() for c in k})
    lines = ['| kernel | dispatches | ' + ' | '.join(counters) + ' |', '|---|---:|' + '---:|' * len(counters)]
    for k, cs in sorted(agg.items(), key=lambda kv: -sum(kv[1].get('SQ_BUSY_CYCLES', kv[1].get('FETCH_SIZE', [0])))):
        n = max(len(v) for v in cs.values())
        if not k.startswith(('gemm', 'attn', 'layernorm', 'rotary', 'embed', 'softmax')):
            continue
        lines.append(f'| `{k}` | {n} | ' + ' | '.join(f'{sum(cs[c]) / len(cs[c]):.4g}' if c in cs else '' for c in counters) + ' |')
    open(out, 'w').write('# rocprofv3 --pmc, mean per dispatch\n\nsources: ' + ', '.join(dirs) + '\n\n' + '\n'.join(lines) + '\n')
    print('\n'.join(lines))

if __name__ == '__main__':
    main(*sys.argv[1:])
