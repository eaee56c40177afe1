#!/usr/bin/env python
"""sharded_forward (esme/shard.py) around the real HIP model on the nccl (= RCCL) backend.

    python tools/shard_check.py                       # world size 1 on a single-GPU box
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/shard_check.py

Every rank builds the same small ESM-2, runs its share of a ragged packed batch, all-gathers the logits
and rank 0 compares them with the plain single-process forward of the whole batch (bit-exact: a
sequence's logits do not depend on what it is packed with)."""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'esm-efficient_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch
import torch.distributed as dist


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def main():
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', str(free_port()))
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    rank, world = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
    from esme import ESM, shard, synthetic as syn
    lengths = [70, 33, 150, 12, 97, 64, 5, 201]
    tokens, cu = syn.random_tokens(lengths, seed=3), syn.cu_lens_of(lengths)
    with tempfile.TemporaryDirectory() as td:
        path = syn.write_checkpoint(os.path.join(td, f'm{rank}.safetensors'), 'esm2_shard', 3, 320, 20, seed=11)
        model = ESM.from_pretrained(path, device=str(dev))
    ok = True
    for mode in ('fast', 'half', 'exact'):              # bf16 logits, then the two fp32-logit modes through the same gather
        model.set_precision(mode)
        with torch.no_grad():
            full = shard.sharded_forward(model, tokens, cu, dev)
            single = model(tokens.to(dev), (cu.to(dev), max(lengths)))
            graphed = model.graphed(tokens.to(dev), (cu.to(dev), max(lengths)), 'forward')
        torch.cuda.synchronize()
        ok_mode = bool(torch.equal(full, single)) and full.dtype == single.dtype and bool(torch.equal(graphed, single))
        ok = ok and ok_mode
        if rank == 0:
            print(f'world {world}, precision {mode}: sharded == single == hipGraph replay: {ok_mode}; logits {tuple(full.shape)} {full.dtype}')
    if rank == 0:
        print(f'world {world}: sharded == single: {ok}; logits {tuple(full.shape)}')
    dist.barrier()
    dist.destroy_process_group()
    if not ok:
        raise SystemExit(1)


if __name__ == '__main__':
    main()
