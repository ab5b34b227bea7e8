"""The multi-GPU entry points rehearsed on the one-GPU box: `bench.py --gpus 2` and `tools/run_config5.py` launched as FRESH child
processes under torch.distributed.run (gloo rendezvous, both ranks on cuda:0 -- KGMA_BENCH_BACKEND / KGMA_BENCH_DEVICE exist for
exactly this), a small genome (--gb).  Rank 0's hit list must equal the one-process run's, and the JSON line must carry what the
driver reads (n_gpus, roofline, per-rank bases).  The driver's real runs use RCCL with one rank per GPU; nothing here re-executes
a process that has touched the GPU: the launcher and the ranks are children of the test process."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(cmd, env, timeout=900):
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, f"{' '.join(cmd)} failed:\n{r.stdout[-2000:]}\n{r.stderr[-4000:]}"
    return r.stdout


def _env():
    env = dict(os.environ)
    env.update(KGMA_BENCH_BACKEND="gloo", KGMA_BENCH_DEVICE="0", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    return env


def _launcher(n, port):
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
            "--master-port", str(port)]


def test_bench_two_ranks_equal_one_process(tmp_path):
    common = ["--steps", "2", "--warmup", "1", "--gb", "2", "--plants", "200", "--no-cpu-baseline", "--no-secondary"]
    h1, h2 = str(tmp_path / "h1.json"), str(tmp_path / "h2.json")
    out1 = _run([sys.executable, "bench.py", "--gpus", "1", "--dump-hits", h1] + common, _env())
    out2 = _run(_launcher(2, _free_port()) + ["bench.py", "--gpus", "2", "--dump-hits", h2] + common, _env())
    j1 = json.loads([ln for ln in out1.splitlines() if ln.startswith("{")][-1])
    j2 = json.loads([ln for ln in out2.splitlines() if ln.startswith("{")][-1])
    assert j1["n_gpus"] == 1 and j2["n_gpus"] == 2
    for j in (j1, j2):
        assert j["unit"] == "Mbp/s" and j["value"] > 0 and j["scaling"] == "strong" and j["steps"] == 2
        assert set(("bound", "achieved", "peak", "frac", "traffic", "unit")) <= set(j["roofline"])
        assert j["config"]["genome_bases"] == 2_000_000_000
    assert j1["per_rank"]["rank0_bases"] == 2_000_000_000 and j2["per_rank"]["rank0_bases"] == 1_000_000_000
    assert j2["config"]["bases_per_gpu_rank0"] == 1_000_000_000
    a, b = json.load(open(h1)), json.load(open(h2))
    assert len(a["hits"]) > 20
    assert a["hits"] == b["hits"]
    assert j1["timed_step"]["n_hits"] == j2["timed_step"]["n_hits"] == len(a["hits"])


def test_config5_three_ranks_equal_one_process(tmp_path):
    common = ["--gb", "2", "--plants", "300", "--no-chain"]
    h1, h3 = str(tmp_path / "c1.json"), str(tmp_path / "c3.json")
    o1, o3 = str(tmp_path / "o1.json"), str(tmp_path / "o3.json")
    _run([sys.executable, "tools/run_config5.py", "--dump-hits", h1, "--out", o1] + common, _env())
    _run(_launcher(3, _free_port()) + ["tools/run_config5.py", "--dump-hits", h3, "--out", o3] + common, _env())
    a, b = json.load(open(h1)), json.load(open(h3))
    assert len(a["hits"]) > 20
    assert sorted(a["hits"]) == sorted(b["hits"])
    j1, j3 = json.load(open(o1)), json.load(open(o3))
    assert j1["n_ranks"] == 1 and j3["n_ranks"] == 3 and j1["bases"] == j3["bases"] == 2_000_000_000
    assert j1["n_hits"] == j3["n_hits"] == len(a["hits"])
