"""bench.py --gpus N without torchrun around it: the flag itself starts the ranks."""
import json
import socket
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")          # the entry point the child processes of a run re-enter


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(args):
    """`python bench.py --gpus N` without torchrun: start the N ranks here."""
    import torch
    backend = os.environ.get("ACG_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if ndev < args.gpus and backend != "gloo" and not getattr(args, "dry_run", False):
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible (ACG_BENCH_BACKEND=gloo rehearses the launch path "
                         "with several ranks per GPU)" % (args.gpus, ndev))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus,
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), BENCH] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))
