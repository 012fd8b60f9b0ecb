"""CPU, world_size 2, gloo: the N>1 path — weight-blob broadcast, utterance sharding, result gather.
The per-rank 'engine' here is the numpy oracle (this is tests/: allowed); on the GPU box the same
shard/broadcast code drives the HIP engine with backend nccl (= RCCL)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mi355tts.config import BigVGANConfig
from mi355tts import weights as W
from mi355tts import shard as S


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _unpack(cfg, blob):
    st, off = {}, 0
    for name, shape, _ in W.bigvgan_spec(cfg):
        n = int(np.prod(shape))
        st[name] = blob[off:off + n].reshape(shape)
        off += n
    return st


def _worker(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "text-to-speech-tts-onnx_amd")]
    from oracle import bigvgan_np as O
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = BigVGANConfig.small()
    spec = W.bigvgan_spec(cfg)
    blob = W.pack_bigvgan(cfg, W.synth_state(spec, 9527)) if rank == 0 else 0
    blob = S.broadcast_blob(blob, src=0)
    st = _unpack(cfg, blob)
    n_utts = 5
    a, b = S.shard_range(n_utts, world, rank)
    local = []
    for u in range(a, b):
        mel = W.synth_normal(100 + u, "mel", (1, cfg.num_mels, 6 + u))
        local.append(O.bigvgan_int16(cfg, st, mel)[0, 0])
    allw = S.gather_waveforms(local, dst=0)
    checksum = float(np.abs(blob).sum())
    t = torch.tensor([checksum], dtype=torch.float64)
    lst = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(lst, t)
    if rank == 0:
        q.put(([w.tolist() for w in allw], [float(x) for x in lst]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_broadcast_shard_gather():
    from oracle import bigvgan_np as O
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    waves, sums = q.get(timeout=240)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert sums[0] == sums[1]                                   # every rank holds the same weights
    cfg = BigVGANConfig.small()
    st = W.synth_state(W.bigvgan_spec(cfg), 9527)
    assert len(waves) == 5
    for u, w in enumerate(waves):                               # concatenation == single-rank result
        mel = W.synth_normal(100 + u, "mel", (1, cfg.num_mels, 6 + u))
        assert np.array_equal(np.asarray(w, np.int16), O.bigvgan_int16(cfg, st, mel)[0, 0])
