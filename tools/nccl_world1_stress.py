"""Stress harness for tests/nccl_world1_worker.py (VERDICT r4 'next' 1a: root-cause the one unexplained failure of the RCCL world-1
test instead of retrying it).  Launches the worker `--procs` times as fresh one-rank jobs, each looping its sections `--repeat`
times in-process, optionally while THIS process holds a GPU context, a block of memory and a live HIP graph the way the pytest
parent does in the full suite (`--hold-gb`).  Every failing key arrives with the tensor that differed, its max-abs difference and
the number of differing elements; non-zero exits arrive with their stderr tail.  One JSON line per worker + a summary line."""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--procs", type=int, default=4)
    ap.add_argument("--repeat", type=int, default=3)
    ap.add_argument("--sections", default="gather,swap,ddp_eager,ddp_graphed,torch_ddp")
    ap.add_argument("--train-g", type=int, default=1)
    ap.add_argument("--hold-gb", type=float, default=0.0, help="hold this much GPU memory + a live graph in the parent, like pytest does")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    held = None
    if args.hold_gb > 0:
        import torch
        held = torch.empty(int(args.hold_gb * (1 << 30)), dtype=torch.uint8, device="cuda")
        g = torch.cuda.CUDAGraph()
        x = torch.zeros(1 << 20, device="cuda")
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            x.add_(1)
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            x.add_(1)
        g.replay()
        torch.cuda.synchronize()
        held = (held, g, x)
    keys = ("gather_outputs_equal", "overlapped_gather_equal", "overlapped_gather_works_were_real", "graphed_swap_equal_eager",
            "graphed_swap_gather_equal", "averager_active", "eager_averaged_step_equal", "graphed_averaged_step_equal",
            "torch_ddp_step_equal")
    summary = {"procs": args.procs, "repeat": args.repeat, "sections": args.sections, "train_G": bool(args.train_g),
               "hold_gb": args.hold_gb, "crashes": 0, "failed_keys": {}, "iterations": 0, "seconds": []}
    lines = []
    for i in range(args.procs):
        env = dict(os.environ)
        env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1",
                   E4S_FORCE_COLLECTIVES="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
        t0 = time.time()
        p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "nccl_world1_worker.py"), "--repeat", str(args.repeat),
                            "--sections", args.sections, "--train-g", str(args.train_g)], env=env, cwd=ROOT, capture_output=True,
                           text=True, timeout=1500)
        dt = round(time.time() - t0, 1)
        summary["seconds"].append(dt)
        out = [ln for ln in p.stdout.splitlines() if ln.startswith("NCCL_WORLD1 ")]
        if p.returncode != 0 or not out:
            summary["crashes"] += 1
            marks = [ln for ln in p.stdout.splitlines() if ln.startswith("W1 ")]
            what = [ln for ln in p.stderr.splitlines() if "what():" in ln or "HIP error" in ln]
            rec = {"proc": i, "returncode": p.returncode, "last_mark": marks[-1] if marks else None, "error_lines": what[:4],
                   "stderr_tail": p.stderr[-1500:]}
        else:
            res = json.loads(out[-1][len("NCCL_WORLD1 "):])
            summary["iterations"] += res["repeat"]
            for k in keys:
                if k in res and res[k] is not True:
                    summary["failed_keys"][k] = summary["failed_keys"].get(k, 0) + 1
            rec = {"proc": i, "seconds": dt, **res}
        lines.append(rec)
        print(json.dumps(rec), flush=True)
    print("STRESS_SUMMARY " + json.dumps(summary), flush=True)
    if args.out:
        with open(args.out, "w") as fh:
            json.dump({"summary": summary, "workers": lines}, fh, indent=1)
    del held


if __name__ == "__main__":
    main()
