"""-m gpu: the N > 1 branch of bench.py on a one-GPU box — `python bench.py --gpus 2` (the driver's own invocation form, no
torchrun: bench.py starts its ranks itself) with two ranks sharing device 0 (SRLHIP_SINGLE_DEVICE=1) and exchanging over
gloo instead of RCCL.  Checks the launch contract (one JSON line from rank 0,
n_gpus = 2, weak scaling: env ids [0, 4096) and [4096, 8192)) and the path's only collective: the all-gather of per-env
episode returns narrowed on the device by srlhip_episode_stats_device."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(workload, port, extra=()):
    env = dict(os.environ, SRLHIP_SINGLE_DEVICE="1", SRLHIP_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    launcher = [sys.executable] if port is None else [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                                                      "--master-addr", "127.0.0.1", "--master-port", str(port)]
    cmd = launcher + [os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                      "--workload", workload, "--no-cpu-baseline"] + list(extra)
    env.pop("WORLD_SIZE", None)
    out = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]          # rank 0 prints exactly one line
    return json.loads(lines[0])


def test_kuka_bench_two_ranks_gather_episode_returns():
    line = run_bench("kuka", None, ["--inner-steps", "1100"])        # self-launched; every env finishes at least one episode per rollout
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["config"]["ranks_seen"] == 2 and line["config"]["dist_backend"] == "gloo"
    assert line["config"]["env_steps_per_bench_step"] == 2 * 4096 * 1100
    g = line["config"]["episode_returns_allgathered"]
    assert g["count"] == 2 * 4096
    assert -1.0 <= g["mean"] <= 5.0 and g["mean"] != 0.0             # sparse Kuka returns: -1 (table / out of reach) ... +5 (button pressed)
    assert line["value"] > 0 and line["roofline"]["kernel"] == "kuka_tree_rollout_k"


def test_mobile_bench_two_ranks_under_torchrun():
    line = run_bench("mobile", 29533, ["--inner-steps", "512"])      # the driver's N > 1 form: torch.distributed.run starts the ranks
    assert line["n_gpus"] == 2 and line["config"]["episode_returns_allgathered"]["count"] == 2 * 4096
