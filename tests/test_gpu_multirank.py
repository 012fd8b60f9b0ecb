"""GPU: the N > 1 path with the HIP engine.  Two ranks share the one GPU of the test box (backend gloo, because RCCL
refuses two ranks on one device; MI355TTS_BENCH_ONE_GPU=1 maps both ranks to cuda:0) and run bench.py's own multi-rank
branches — weight blob built on rank 0, broadcast, consumed from device memory, per-rank utterance shards, barrier + max
over ranks timing, one JSON line from rank 0 — on the reduced model (MI355TTS_BENCH_SMALL=1).  The per-rank waveforms must
equal what one process computes for the same utterances (utterances are independent, no data-path
collective)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from mi355tts.config import F5Config
from mi355tts import weights as W
from mi355tts.f5 import F5Engine

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.timeout(900)
def test_bench_two_ranks_on_one_gpu_equals_single_process(tmp_path):
    env = dict(os.environ, MI355TTS_BENCH_BACKEND="gloo", MI355TTS_BENCH_ONE_GPU="1", MI355TTS_BENCH_SMALL="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
           "--batch", "3", "--dtype", "f32", "--dump-dir", str(tmp_path)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=800, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                                   # exactly one JSON line, from rank 0
    assert r.stdout.strip() == lines[0] and len(lines[0]) < 4096       # round 5: stdout is that line and nothing else, compact
    assert "BENCH_DETAIL {" in r.stderr                                # the full record travels on stderr / bench_detail.json
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["value"] > 0 and line["dtype"] == "f32"
    assert "PLUMBING TEST ONLY" in line["config"]["workload"] and line["config"]["utterances_per_gpu"] == 3
    assert line["config"]["weight_bcast_ms"] > 0
    # round 4: the job-level utterance list, the per-rank times behind the maximum, and which device every rank drove
    assert line["config"]["utterances_total"] == 6 and line["config"]["utterance_seeds"] == [9527, 9532]
    assert len(line["config"]["per_rank_ms"]) == 2 and max(line["config"]["per_rank_ms"]) == pytest.approx(line["ms_per_step"], rel=1e-9)
    devs = line["config"]["rank_devices"]
    assert len(devs) == 2 and devs[0] == devs[1]            # this plumbing test shares the box's one GPU, and the line says so
    # value = audio of BOTH ranks / max-over-ranks time
    per_gpu = line["config"]["audio_seconds_per_step_per_gpu"]
    assert abs(line["value"] - 2 * per_gpu / (line["ms_per_step"] * 1e-3)) < 1e-6 * line["value"]
    cfg = F5Config.small()
    eng = F5Engine(cfg, W.synth_state(W.f5_spec(cfg), 9527), dtype="f32")
    for rank in range(2):
        got = np.load(tmp_path / f"f5_f32_u3_rank{rank}.npy")
        audio, ids, N, noise = W.f5_synthetic_inputs(cfg, 3, rank, L=24000, first=3 * rank)      # the job-level list: rank r owns utterances 3 r .. 3 r + 2
        want = eng.synthesize(audio, ids, N, noise=noise)
        assert got.shape == want.shape and np.array_equal(got, want), rank
        assert np.sqrt(np.mean(want.astype(np.float64) ** 2)) > 100
    assert not np.array_equal(np.load(tmp_path / "f5_f32_u3_rank0.npy"), np.load(tmp_path / "f5_f32_u3_rank1.npy"))
    eng.close()


@pytest.mark.timeout(900)
def test_bench_launches_its_own_ranks_without_torchrun(tmp_path):
    """`python bench.py --gpus 2` with no launcher and no WORLD_SIZE (the driver's form): bench.py re-executes itself under
    torch.distributed.run, rank 0 prints the one JSON line.  Both ranks share the box's one GPU (gloo)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(MI355TTS_BENCH_BACKEND="gloo", MI355TTS_BENCH_ONE_GPU="1", MI355TTS_BENCH_SMALL="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--batch", "2",
           "--dtype", "bf16", "--dump-dir", str(tmp_path)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=800, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["config"]["weight_bcast_ms"] > 0
    assert (tmp_path / "f5_bf16_u2_rank0.npy").exists() and (tmp_path / "f5_bf16_u2_rank1.npy").exists()


@pytest.mark.timeout(1800)
def test_bench_two_gpus_rccl():
    """The real thing, when the box has two devices: one rank per GPU, backend nccl (= RCCL over xGMI), the weight blob
    broadcast device to device, configs[3] shard per rank."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=1700, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert line["n_gpus"] == 2 and line["dtype"] == "bf16" and "configs[3] shard" in line["config"]["workload"]
    assert line["config"]["weight_bcast_ms"] > 0 and line["config"]["collective_backend"] == "nccl"
    assert line["config"]["utterances_total"] == 16 and len(set(line["config"]["rank_devices"])) == 2


@pytest.mark.timeout(900)
def test_rccl_one_rank_group_broadcasts_the_device_blob_and_the_engine_runs_from_it(tmp_path):
    """RCCL itself, on the one GPU of the test box: a one-rank process group with backend nccl, the weight blob sent
    through shard.broadcast_blob_device(force=True) — communicator, stream and broadcast kernel on the device tensor, as every rank
    of an 8-GPU run does — and an engine built from the blob that came out of the collective.  Twice: in a child process that
    checks the blob and the engine's waveform against an engine that never saw a collective, and through
    `bench.py --gpus 1 --force-collective`, whose line must say backend nccl and a measured broadcast time."""
    code = r"""
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.path.join(sys.argv[1], "text-to-speech-tts-onnx_amd"))
from mi355tts.config import F5Config
from mi355tts import weights as W
from mi355tts.f5 import F5Engine
from mi355tts.shard import broadcast_blob_device
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
assert dist.get_backend() == "nccl"
cfg = F5Config.small()
blob = W.pack_f5(cfg, W.synth_state(W.f5_spec(cfg), 9527))
t = torch.from_numpy(blob).cuda()
out = broadcast_blob_device(t, src=0, force=True)
torch.cuda.synchronize()
assert out.data_ptr() == t.data_ptr() and np.array_equal(out.cpu().numpy(), blob)
probe = torch.ones(1 << 20, device="cuda"); dist.all_reduce(probe); torch.cuda.synchronize()      # a second collective kind on the same communicator
assert float(probe.sum()) == float(1 << 20)
audio, ids, N, noise = W.f5_synthetic_inputs(cfg, 2, 0, L=24000)
a = F5Engine(cfg, blob_device=out, dtype="f32").synthesize(audio, ids, N, noise=noise)
b = F5Engine(cfg, blob=blob, dtype="f32").synthesize(audio, ids, N, noise=noise)
assert np.array_equal(a, b)
maps = open("/proc/self/maps").read()
libs = sorted({l.split("/")[-1] for l in maps.splitlines() if "librccl" in l or "libnccl" in l})
print("RCCL_OK", dist.get_backend(), libs, flush=True)
assert libs, "no RCCL library mapped"
dist.destroy_process_group()
"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", code, ROOT], env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0 and "RCCL_OK nccl" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
    env.pop("MASTER_PORT")
    env["MI355TTS_BENCH_SMALL"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-collective", "--steps", "1", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert line["n_gpus"] == 1 and line["config"]["collective_backend"] == "nccl" and line["config"]["weight_bcast_ms"] > 0


def test_two_ranks_on_one_device_are_refused_without_the_plumbing_switch():
    """One process per GPU is the contract: `bench.py --gpus 2` whose ranks resolve to the SAME physical device must fail loudly
    (a scaling line measured that way would be two ranks time-slicing one GPU) unless MI355TTS_BENCH_ONE_GPU=1 says it is a
    plumbing test.  Forced here by hiding all but one device from both ranks."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MI355TTS_BENCH_ONE_GPU")}
    env.update(MI355TTS_BENCH_BACKEND="gloo", MI355TTS_BENCH_SMALL="1", HSA_ENABLE_IPC_MODE_LEGACY="0", HIP_VISIBLE_DEVICES="0",
               MI355TTS_BENCH_SAME_DEVICE_TEST="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode != 0
    assert "only 1 device(s) visible" in (r.stdout + r.stderr) or "ranks share a physical GPU" in (r.stdout + r.stderr)


def test_engine_from_device_blob_equals_engine_from_host_blob():
    """mi_f5_create_mem(MI_DEVICE) (device-to-device conversion of the DiT / Vocos matrices, bf16 rounding done by a kernel)
    builds the same engine as the host-blob path, for every engine dtype."""
    import torch
    cfg = F5Config.small()
    raw = W.synth_state(W.f5_spec(cfg), 9527)
    blob = W.pack_f5(cfg, raw)
    audio, ids, N, noise = W.f5_synthetic_inputs(cfg, 2, 0, L=24000)
    for dtype in ("f32", "bf16", "f16"):
        e_host = F5Engine(cfg, blob=blob, dtype=dtype)
        e_dev = F5Engine(cfg, blob_device=torch.from_numpy(blob).cuda(), dtype=dtype)
        a = e_host.synthesize(audio, ids, N, noise=noise)
        b = e_dev.synthesize(audio, ids, N, noise=noise)
        assert np.array_equal(a, b), dtype
        e_host.close(); e_dev.close()
    from mi355tts.config import BigVGANConfig
    from mi355tts.bigvgan import BigVGANVocoder
    vcfg = BigVGANConfig.small()
    vblob = W.pack_bigvgan(vcfg, W.synth_state(W.bigvgan_spec(vcfg), 9527))
    mel = W.synth_normal(5, "mel", (2, vcfg.num_mels, 9))
    v1 = BigVGANVocoder(vcfg, blob=vblob, dtype="f16")
    v2 = BigVGANVocoder(vcfg, blob_device=torch.from_numpy(vblob).cuda(), dtype="f16")
    assert np.array_equal(v1.run(mel), v2.run(mel))
    v1.close(); v2.close()
