import sys, math, torch, numpy as np
sys.path[:0]=['/root/repo','/root/repo/esm-efficient_amd']
from esme import _hip
from esme.attention import _fold_layernorm
dev=torch.device('cuda',0)
def rnd(shape, seed, scale=1.0):
    rng=np.random.Generator(np.random.PCG64(seed)); return (torch.from_numpy(rng.standard_normal(shape,dtype=np.float32))*scale).to(torch.bfloat16).to(dev)
lib=_hip.load()
M,K,N=1500,1280,5120
x=rnd((M,K),1); w=rnd((N,K),2,1/math.sqrt(K)); b=rnd((N,),3,0.1); g=(1+0.1*rnd((K,),4).float()).to(torch.bfloat16); be=rnd((K,),5,0.1)
wf,c1,c2=_fold_layernorm(w,b,g,be)
res={}
for tile in (1,2):
    _hip.set_gemm_options(tile=tile)
    sums=_hip.row_sums(x)
    plain=_hip.gemm(x,w,b)
    gelu=_hip.gemm_fused(x,wf,None,_hip.EPI_GELU,ln=(sums,K,1e-5,c1,c2))
    w2=rnd((K,N),6,1/math.sqrt(N)); b2=rnd((K,),7,0.1)
    part=torch.empty(_hip.stats_blocks(M,K),M,2,dtype=torch.float32,device=dev)
    y=_hip.gemm_fused(gelu,w2,b2,_hip.EPI_RESIDUAL,x.clone(),0.5,stats_out=part)
    gelu2=_hip.gemm_fused(y,wf,None,_hip.EPI_GELU,ln=(part,K,1e-5,c1,c2))
    res[tile]=(plain,gelu,y,part,gelu2)
_hip.set_gemm_options(tile=0)
for name,a,b_ in zip(('plain','gelu+lnf','resid','stats','gelu2 via stats'),res[1],res[2]):
    if name == 'stats':      # (N/128, M, 2) vs (N/256, M, 2): different granularity by design; the consumers must agree (next line)
        print(name, 'blocks', a.shape[0], b_.shape[0], 'row totals close', bool(torch.allclose(a.sum(0), b_.sum(0), rtol=1e-5, atol=1e-3)))
        continue
    print(name, 'bit-equal', bool(torch.equal(a,b_)), 'max diff', float((a.float()-b_.float()).abs().max()))
# fused QKV: LN fold + rotary epilogue, ragged positions
from esme.rotary import RotaryEmbedding
lengths=[500,500,377,123]
cu=torch.tensor(np.cumsum([0]+lengths),dtype=torch.int32,device=dev)
pos,_=_hip.seq_positions(cu,M)
H,d=20,64
rot=RotaryEmbedding(d); cos,sin=rot.tables(500,dev,torch.bfloat16)
wq=rnd((3*K,K),8,1/math.sqrt(K)); bq=rnd((3*K,),9,0.1)
wqf,c1q,c2q=_fold_layernorm(wq,bq,g,be)
out={}
for tile in (1,2):
    _hip.set_gemm_options(tile=tile)
    sums=_hip.row_sums(x)
    out[tile]=_hip.gemm_fused(x,wqf,None,ln=(sums,K,1e-5,c1q,c2q),rot=(cos,sin,pos,d,2*K))
_hip.set_gemm_options(tile=0)
print('qkv lnf+rotary bit-equal', bool(torch.equal(out[1],out[2])), float((out[1].float()-out[2].float()).abs().max()))
# whole model: 3 sequences alone with each tile config, and inside a big batch
import tempfile, os
from esme import ESM, synthetic as syn
with tempfile.TemporaryDirectory() as td:
    path=syn.write_checkpoint(os.path.join(td,'m.safetensors'),'esm2_t',2,1280,20,seed=3)
    model=ESM.from_pretrained(path,device='cuda:0')
tokens,cu2,ml,lens=syn.uniform_batch(20000,500,seed=0)
big=model(tokens.to(dev),(cu2.to(dev),ml))
sub_t=tokens[:1500]; sub_cu=cu2[:4]
for tile in (0,1,2):
    _hip.set_gemm_options(tile=tile)
    alone=model(sub_t.to(dev),(sub_cu.to(dev),500))
    print('model alone tile',tile,'== packed:',bool(torch.equal(alone,big[:1500])), float((alone.float()-big[:1500].float()).abs().max()))
    rep_a=model.forward_representation(sub_t.to(dev),(sub_cu.to(dev),500))
_hip.set_gemm_options(tile=0)
rep_b=model.forward_representation(tokens.to(dev),(cu2.to(dev),ml))[:1500]
print('rep equal (tile2 alone vs packed)', bool(torch.equal(rep_a,rep_b)))
# stage by stage
x0=model._embedding_phys(sub_t.to(dev)); x1=model._embedding_phys(tokens.to(dev))[:1500]
print('embed equal', bool(torch.equal(x0,x1)))
