"""`python3 bench.py --gpus N` starts its own ranks (the driver's command shape has no launcher in it).

CPU part: the parent must not touch the GPU, must not exec, and must hand the child exactly its own arguments.
GPU part (one card): `RTXN_REHEARSE_ON_ONE_GPU=1 python3 bench.py --gpus 2 ...` with no torchrun in the command prints ONE
JSON line with n_gpus == 2 and a gathered frame bit-identical to the single-process one."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_self_launch_builds_a_child_torchrun(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7

    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    for name in ("execv", "execve", "execvp", "execvpe", "execl", "execlp"):
        monkeypatch.setattr(os, name, lambda *a, **k: pytest.fail("bench.py must never exec"))
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7                                   # the child's status is the parent's
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_launched_rank_with_wrong_world_size_exits_2():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 2 and "--nproc-per-node must equal --gpus" in p.stderr and p.stdout == ""


@pytest.mark.gpu
def test_bench_gpus_2_without_a_launcher_rehearsed_on_one_gpu(gpu):
    env = dict(os.environ, RTXN_REHEARSE_ON_ONE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--width", "200", "--height", "150", "--grid", "64", "--kernel-steps", "1"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 2 and rec["value"] > 0
    assert rec["gather_check"].startswith("bit-identical")
    assert rec["scaling"] == "strong" and "ray-shard x2" in rec["config"]["parallelism"]
