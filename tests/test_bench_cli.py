"""bench.py's launcher behaviour, without a GPU: `--gpus N` started bare spawns N ranks itself (the driver's own launcher
line), and a launcher whose world size differs from --gpus is refused instead of reported under the wrong n_gpus."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    e.update(kw)
    return e


def test_bare_gpus_n_spawns_n_ranks_through_torch_distributed_run():
    out = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--steps", "7", "--warmup", "3"], capture_output=True, text=True,
                         timeout=120, env=_env(BENCH_SPAWN_DRYRUN="1"))
    assert out.returncode == 0, out.stderr[-2000:]
    cmd = json.loads(out.stdout.strip().splitlines()[-1])["spawn"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nnodes=1" in cmd
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    i = cmd.index(BENCH)
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "7", "--warmup", "3"]


def test_world_size_that_differs_from_gpus_is_refused():
    out = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--steps", "2", "--warmup", "1"], capture_output=True, text=True,
                         timeout=120, env=_env(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
    assert out.returncode != 0 and "refusing" in out.stderr
    out = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--steps", "2", "--warmup", "1"], capture_output=True, text=True,
                         timeout=120, env=_env(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"))
    assert out.returncode != 0 and "refusing" in out.stderr
