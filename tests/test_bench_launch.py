"""bench.py launch contract (CPU): `--gpus N` can never silently measure one rank.
* WORLD_SIZE set and != --gpus  -> error exit with a message, before any device work;
* WORLD_SIZE unset and --gpus N>1 -> bench.py becomes the launcher: N children with the torchrun environment on 127.0.0.1."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_world_size_mismatch_is_an_error():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr and r.stdout.strip() == ""


def test_self_spawn_builds_one_rank_per_gpu(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    seen = []

    class FakeProc:
        def __init__(self, cmd, env):
            seen.append((cmd, env))

        def wait(self):
            return 0

        def poll(self):
            return 0
    monkeypatch.setattr(subprocess, "Popen", lambda cmd, env=None: FakeProc(cmd, env))
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    assert bench.self_spawn(4) == 0
    assert [e["RANK"] for _, e in seen] == ["0", "1", "2", "3"] and [e["LOCAL_RANK"] for _, e in seen] == ["0", "1", "2", "3"]
    assert all(e["WORLD_SIZE"] == "4" and e["MASTER_ADDR"] == "127.0.0.1" and e["MASTER_PORT"] == seen[0][1]["MASTER_PORT"] for _, e in seen)
    assert all(c[-4:] == ["--gpus", "4", "--steps", "3"] and c[1].endswith("bench.py") for c, _ in seen)
