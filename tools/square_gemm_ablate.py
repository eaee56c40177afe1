import os, sys, statistics
ROOT='/root/repo'
os.environ['ESME_HIP_LIB']=os.path.join(ROOT,'esm-efficient_amd','esme','libesme_hip_trace.so')
sys.path.insert(0, os.path.join(ROOT,'esm-efficient_amd'))
import torch
from esme import _hip
lib=_hip.load()
lib.esme_hip_debug_set_gemm_tile(2)
torch.manual_seed(0)
for n in (4096, 8192):
    a = (torch.rand(n, n, device='cuda') * 2 - 1).to(torch.bfloat16); w = (torch.rand(n, n, device='cuda') * 2 - 1).to(torch.bfloat16)
    c = torch.empty(n, n, device='cuda', dtype=torch.bfloat16)
    for mode,label in ((0,'full'),(2,'no C store'),(3,'loop only')):
        lib.esme_hip_debug_set_gemm_nt(mode)
        for _ in range(3): _hip.gemm(a, w, None, out=c)
        ts=[]
        for r in range(5):
            st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            st.record()
            for _ in range(20): _hip.gemm(a, w, None, out=c)
            en.record(); torch.cuda.synchronize()
            ts.append(st.elapsed_time(en)/20)
        ms=statistics.median(ts)
        print(f'{n}^3 {label:10s}: {ms*1e3:8.1f} us {2*n**3/ms/1e9:7.1f} TF', flush=True)
lib.esme_hip_debug_set_gemm_nt(0)
