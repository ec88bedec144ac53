"""RCCL on the one GPU that is reachable (VERDICT r3 'missing' 1-2, 'next' 5): a world of ONE rank on backend "nccl" with the
collectives forced on, in subprocesses -- (a) tests/nccl_world1_worker.py: gather / overlapped gather / a face-swap graph captured
under a live process group / eager and GRAPH-CAPTURED data-parallel G steps (bucket all-reduces inside the HIP graph) / torch's own
DistributedDataParallel around Net3, each equal bit for bit to the collective-free computation; (b) bench.py's own main() launched the way the driver launches N ranks."""
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
    port = s.getsockname()[1]
    s.close()
    return port


def _run(cmd, timeout=900):
    """Run a one-rank job on a fresh rendezvous port.  No retry: round 4 answered one unexplained failure of this file with a second
    attempt; round 5 made the worker say WHAT differs, stress-ran it (tools/nccl_world1_stress.py, profiles/r05_nccl_world1_stress.json)
    and took the retry out -- a failure here is a defect to be read, not a transient to be absorbed."""
    return subprocess.run(cmd, env=_env(_free_port()), cwd=ROOT, capture_output=True, text=True, timeout=timeout)


def _env(port):
    env = dict(os.environ)
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", E4S_FORCE_COLLECTIVES="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    return env


KEYS = ("gather_outputs_equal", "overlapped_gather_equal", "overlapped_gather_works_were_real", "graphed_swap_equal_eager",
        "graphed_swap_gather_equal", "averager_active", "eager_averaged_step_equal", "graphed_averaged_step_equal", "torch_ddp_step_equal")


def test_world1_nccl_collectives_gather_and_graph_captured_ddp_step():
    """train_G=True in the data-parallel sections (the reference's default): GradAverager eager / captured, and torch's own
    DistributedDataParallel(find_unused_parameters=True, broadcast_buffers=False) around Net3 as coach.py:74-85 wraps it."""
    p = _run([sys.executable, os.path.join(ROOT, "tests", "nccl_world1_worker.py")])
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("NCCL_WORLD1 ")][-1]
    res = json.loads(line[len("NCCL_WORLD1 "):])
    print(res)
    bad = [k for k in KEYS if res.get(k) is not True]
    assert not bad, (bad, res["diffs"], res)
    assert res["backend"] == "nccl" and res["train_G"] is True
    assert res["buckets_fired_during_backward"] >= 1, res          # at least one all-reduce left while the backward was still running


def test_world1_nccl_bench_main_runs_the_multi_rank_path():
    """bench.py under the driver's launch protocol (RANK / WORLD_SIZE / MASTER_* in the environment) with one rank and forced
    collectives: init_process_group('nccl', device_id=...), GraphedFaceSwap captured in thread_local mode, OverlappedGather of the uint8
    images on RCCL's stream, barrier, MAX all-reduce of the time, the JSON line."""
    p = _run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--steps-only"])
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["value"] > 50 and "forced collectives" in line["config"]["parallelism"], line
    assert "RCCL all_gather of the uint8" in line["config"]["parallelism"]
