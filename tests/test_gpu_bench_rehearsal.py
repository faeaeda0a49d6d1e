"""bench.py's N > 1 code paths as far as one GPU lets them run: the native collective with a one-rank communicator, and
the torch / peer exchanges as two real processes that share cuda:0 (BBR_BENCH_SINGLE_DEVICE=1; gloo stands in for RCCL,
which refuses two ranks on one GPU).  Every run ends with --verify: the gathered frame equals the unpartitioned render,
bit for bit.  These are rehearsals of the control flow, not measurements."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON = ["--workload", "c2", "--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--verify"]


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _line(proc):
    assert proc.returncode == 0, proc.stderr[-3000:]
    lines = [l for l in proc.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, proc.stdout[-2000:]
    out = json.loads(lines[0])
    # what the supervisor said about attempts it gave up travels with the line, so that a failing assertion shows it
    out["_supervisor_stderr"] = [l for l in proc.stderr.splitlines() if "bench supervisor" in l or "Error" in l or "differs" in l][-12:]
    return out


@pytest.mark.parametrize("extra", [[], ["--gather", "rgba32f"], ["--present"], ["--gather", "rgba16f"]])
def test_native_collective_with_one_rank(extra):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    out = _line(subprocess.run([sys.executable, "bench.py", "--gpus", "1", "--force-dist", "--exchange", "native"] + COMMON + extra,
                               cwd=ROOT, capture_output=True, text=True, timeout=600, env=env))
    assert out["verified_against_unpartitioned_render"] is True and out["n_gpus"] == 1
    ex = out["exchange"]
    assert ex["who"] == "native" and ex["rccl_ranks"] == 1 and ex["attempts"] == []      # ncclCommCount on the library's communicator
    assert ex["link_bound_ms_estimate"] == 0.0 and ex["link_rate_gbs_assumed"] > 0       # (one rank receives nothing)
    if extra == ["--gather", "rgba16f"]:
        assert ex["form"] == "rgba16f" and "binary16" in out["config"]["output"] and "LOSSY" in out["config"]["output"]
        assert ex["bytes_per_rank_block"] == 1920 * 1080 * 8   # (half of the RGBA32F block)


@pytest.mark.parametrize("exchange,extra", [("peer", []), ("peer", ["--present"]), ("torch", []), ("torch", ["--gather", "rgba16f"])])
def test_two_ranks_on_one_gpu(exchange, extra):
    env = dict(os.environ, BBR_BENCH_BACKEND="gloo", BBR_BENCH_SINGLE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), "bench.py", "--gpus", "2", "--exchange", exchange] + COMMON + extra
    out = _line(subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env))
    assert out["verified_against_unpartitioned_render"] is True and out["n_gpus"] == 2
    assert out["scaling"] == "strong"
    assert ("bbr_push_shard" if exchange == "peer" else "torch.distributed") in out["config"]["partition"], out["_supervisor_stderr"]
    ex = out["exchange"]
    assert ex["attempts"] == [] and ex["who"] == exchange and ex["rccl_ranks"] is None
    # the ceiling a scaling record can be held against: a block over one link (two ranks: ring and direct are the same)
    assert ex["link_bound_ms_estimate"] == pytest.approx(ex["bytes_per_rank_block"] / (ex["link_rate_gbs_assumed"] * 1e9) * 1e3, rel=1e-3)
    assert ex["link_bound_ms_estimate_ring"] == pytest.approx(ex["link_bound_ms_estimate"], rel=1e-3)
    for pr in out["roofline"]["per_rank"]:
        assert pr["alone"]["avg_geometry_ms"] > 0 and pr["alone"]["avg_raster_ms"] > 0 and pr["alone"]["avg_shade_ms"] > 0


def test_a_stalled_exchange_is_given_up_and_the_torch_attempt_completes():
    """the one combination no rehearsal can reach is a native / peer exchange that HANGS on real
    hardware.  Here rank 1 is made to stall (BBR_BENCH_STALL) in the peer attempt: the supervisor -- the parent, which never
    touches the GPU -- sees no milestone for BBR_BENCH_STALL_LIMIT seconds, kills that attempt's whole process group and
    starts the torch attempt in fresh processes; its line says what was given up and why.  Exit code 0, ONE JSON line."""
    env = dict(os.environ, BBR_BENCH_BACKEND="gloo", BBR_BENCH_SINGLE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0",
               BBR_BENCH_STALL="peer:1:first_frame", BBR_BENCH_STALL_LIMIT="45")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = _line(subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--exchange", "peer"] + COMMON,
                               cwd=ROOT, capture_output=True, text=True, timeout=900, env=env))
    ex = out["exchange"]
    assert out["n_gpus"] == 2 and out["verified_against_unpartitioned_render"] is True and ex["who"] == "torch"
    assert len(ex["attempts"]) == 1 and ex["attempts"][0]["exchange"] == "peer"
    assert "no progress" in ex["attempts"][0]["gave_up_because"] and "first_frame" in ex["attempts"][0]["gave_up_because"]


def test_a_stalled_rank_under_a_launcher_is_given_up_by_every_ranks_supervisor():
    """The same hang in the form the DRIVER starts N > 1 runs: `python -m torch.distributed.run ... bench.py --gpus 2`.  There
    is no common parent then: every rank is its own supervisor of its own worker.  Rank 1's worker stalls in the peer attempt;
    its supervisor gives the attempt up and leaves a flag file, rank 0's supervisor -- whose worker hangs in the exchange
    waiting for rank 1 -- sees the flag (or its own stall limit), kills its worker too, and both start the torch attempt,
    which meets on the attempt's own port.  Rank 0 prints the ONE line."""
    env = dict(os.environ, BBR_BENCH_BACKEND="gloo", BBR_BENCH_SINGLE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0",
               BBR_BENCH_STALL="peer:1:first_frame", BBR_BENCH_STALL_LIMIT="45")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), "bench.py", "--gpus", "2", "--exchange", "peer"] + COMMON
    out = _line(subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env))
    ex = out["exchange"]
    assert out["n_gpus"] == 2 and out["verified_against_unpartitioned_render"] is True and ex["who"] == "torch", out["_supervisor_stderr"]
    assert len(ex["attempts"]) == 1 and ex["attempts"][0]["exchange"] == "peer"


def test_a_rank_that_stalls_behind_the_line_does_not_walk_the_ladder_alone():
    """under a launcher the ranks agree on success too.  Rank 1's worker stalls in front of its last
    milestone -- behind the final barrier, i.e. after rank 0 has printed the line.  Rank 0's supervisor leaves the run's .done
    flag; rank 1's supervisor gives its worker up after the stall limit, finds the flag and leaves with 0 instead of starting
    the next rung against a rendezvous nobody else will join.  One line, exit code 0, no second attempt."""
    env = dict(os.environ, BBR_BENCH_BACKEND="gloo", BBR_BENCH_SINGLE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0",
               BBR_BENCH_STALL="torch:1:done", BBR_BENCH_STALL_LIMIT="45")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), "bench.py", "--gpus", "2", "--exchange", "torch"] + COMMON
    proc = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    out = _line(proc)
    assert out["n_gpus"] == 2 and out["verified_against_unpartitioned_render"] is True and out["exchange"]["attempts"] == []
    assert "another rank's worker completed it" in proc.stderr and "attempt 2/" not in proc.stderr, proc.stderr[-2000:]


def test_bare_bench_command_with_two_gpus_launches_its_own_ranks():
    """`python3 bench.py --gpus 2 --steps 20 --warmup 5` with no launcher around it (what a driver that reuses its N = 1
    command form would run): bench.py starts the two ranks itself and relays rank 0's line.  Rehearsed on the one GPU."""
    env = dict(os.environ, BBR_BENCH_BACKEND="gloo", BBR_BENCH_SINGLE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = _line(subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "20", "--warmup", "5", "--workload", "c2",
                                "--verify"], cwd=ROOT, capture_output=True, text=True, timeout=900, env=env))
    assert out["n_gpus"] == 2 and out["verified_against_unpartitioned_render"] is True
    assert out["exchange"]["ranks_in_communicator"] == 2 and out["exchange"]["bytes_per_rank_block"] > 0
    assert out["exchange"]["who"] == "torch" and out["exchange"]["attempts"] == []   # (gloo rehearsal: the ladder starts at torch)
    assert [p["rank"] for p in out["roofline"]["per_rank"]] == [0, 1]
    assert sum(p["n_shaded"] for p in out["roofline"]["per_rank"]) == out["roofline"]["n_shaded"]
