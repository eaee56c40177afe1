#!/usr/bin/env python
"""sample() / run(): loop a kernel for a few seconds while sampling socket power and the shader clock with rocm-smi
(shared by power_probe.py and attn_power_probe.py)."""
import ctypes, json, os, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'esm-efficient_amd'))
import torch
from esme import _hip

def sample(stop, out):
    while not stop.is_set():
        try:
            r = subprocess.run(['rocm-smi', '--showpower', '--showclocks', '--json'], capture_output=True, text=True, timeout=5)
            d = json.loads(r.stdout)
            card = d[next(iter(d))]
            pw = next((float(v) for k, v in card.items() if 'ower' in k and 'W' in k and v not in ('N/A', '')), None)
            sclk = next((v for k, v in card.items() if 'sclk' in k.lower()), None)
            out.append((pw, sclk))
        except Exception as e:
            out.append((None, repr(e)[:60]))
        time.sleep(0.05)

def run(name, fn, flop, secs=3.0):
    fn(); torch.cuda.synchronize()
    stop, out = threading.Event(), []
    th = threading.Thread(target=sample, args=(stop, out)); th.start()
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < secs:
        for _ in range(20): fn()
        torch.cuda.synchronize(); n += 20
    dt = time.perf_counter() - t0
    stop.set(); th.join()
    pws = [p for p, _ in out if p is not None][2:]
    clk = [c for _, c in out][2:]
    print(f'{name:34s} {flop * n / dt / 1e12:8.1f} TF   power mean {sum(pws) / max(len(pws), 1):7.1f} W max {max(pws, default=0):7.1f} W   sclk samples {clk[:3]} .. {clk[-2:]}', flush=True)

