"""bench.py's OWN main() on two CPU ranks over gloo (E4S_DIST_BACKEND=gloo, --stub-swap): the N>1 path the driver
launches at round end -- per-rank shard seeds, the overlapped uint8 all-gather, drain + barrier inside the timed region, the
MAX-reduced clock and the JSON line -- executed end to end.  The swap itself is a stub (the HIP path needs a GPU)."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _launch(extra, world=2, batch=3):
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), E4S_DIST_BACKEND="gloo")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "4",
                                       "--warmup", "1", "--batch", str(batch), "--stub-swap"] + extra, env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-2000:]
    lines = [l for l in outs[0][0].splitlines() if l.startswith("{")]
    assert len(lines) == 1, outs[0][0]
    assert not [l for l in outs[1][0].splitlines() if l.startswith("{")]        # only rank 0 prints the line
    return json.loads(lines[0])


def test_bench_main_world2_overlapped_uint8_gather():
    rec = _launch([])
    assert rec["n_gpus"] == 2 and rec["steps"] == 4 and rec["warmup"] == 1
    assert rec["config"]["per_gpu_batch"] == 3 and rec["config"]["global_batch"] == 6
    assert rec["stub_gather_ok"] is True and rec["stub_gather_shape"] == [6, 8, 8, 3]      # rank 1's shard at rows 3..5
    assert rec["scaling"] == "weak" and rec["higher_is_better"] is True and rec["vs_baseline"] is None
    assert rec["value"] > 0 and abs(rec["value"] - 6 * 4 / (rec["ms_per_step"] * 4 / 1e3)) < 1e-2 * rec["value"]
    assert "stub" in rec["data"]


def test_bench_main_world2_sync_fp32_gather():
    rec = _launch(["--sync-gather", "--gather-fp32"])
    assert rec["stub_gather_ok"] is True and rec["stub_gather_shape"] == [6, 3, 8, 8]


def test_stub_swap_is_refused_beside_the_hip_path():
    env = dict(os.environ, E4S_DIST_BACKEND="nccl")
    pr = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--stub-swap"], env=env, capture_output=True, text=True)
    assert pr.returncode != 0 and "gloo" in (pr.stderr + pr.stdout)
