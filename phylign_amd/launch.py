"""One rank per GPU from a plain command line: `bench.py --gpus 8` or
`python -m phylign_amd.match_stage --gpus 8 ...` without a launcher around them
start their own ranks here.

The parent is a process that has NOT touched the GPU (no torch import, no HIP
call): it starts N fresh children of the same command with RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_ADDR / MASTER_PORT set -- the environment
`torch.distributed.run` would give them -- relays their output (stdout and
stderr are inherited, so rank 0's JSON line is the parent's), and exits with
the first non-zero status; the other ranks are ended then.  Nothing is
exec'ed over a process that initialised a device."""
import os
import signal
import socket
import subprocess
import sys
import time


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def wants_self_launch(n_gpus, environ=None):
    """True when the command line asks for several ranks and no launcher provided them"""
    env = os.environ if environ is None else environ
    return n_gpus > 1 and "WORLD_SIZE" not in env and "RANK" not in env


def rank_env(rank, world, port, environ=None):
    env = dict(os.environ if environ is None else environ)
    env.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_WORLD_SIZE": str(world),
                "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: what RCCL needs on these hosts
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // world)))
    return env


def _child_setup():
    """between fork and exec of a rank: own process group (ended as a group), and SIGTERM should the parent die first
    (a killed parent must not leave ranks holding GPUs)"""
    os.setsid()
    try:
        import ctypes
        ctypes.CDLL(None, use_errno=True).prctl(1, int(signal.SIGTERM), 0, 0, 0)      # PR_SET_PDEATHSIG
    except Exception:
        pass


def spawn_ranks(cmd, world, poll_s=0.2, grace_s=10.0, extra_env=None):
    """Runs `cmd` (argv list) once per rank and waits.  Returns the exit status of the job: 0 when every rank
    returned 0, else the first failing rank's status (a rank killed by signal n counts as 128 + n)."""
    port = free_port()
    procs = [subprocess.Popen(cmd, env=dict(rank_env(r, world, port), **(extra_env or {})), preexec_fn=_child_setup)
             for r in range(world)]

    def end_all(sig):
        for p in procs:
            if p.poll() is None:
                try:
                    os.killpg(p.pid, sig)                 # exactly the process groups started above
                except (ProcessLookupError, PermissionError):
                    pass

    def on_signal(signum, _frame):
        end_all(signal.SIGTERM)
        sys.exit(128 + signum)
    old = {s: signal.signal(s, on_signal) for s in (signal.SIGINT, signal.SIGTERM)}
    status = 0
    try:
        while True:
            codes = [p.poll() for p in procs]
            bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if bad:
                r, c = bad[0]
                status = 128 - c if c < 0 else c
                sys.stderr.write(f"[launch] rank {r} exited with status {c}: ending the other ranks\n")
                end_all(signal.SIGTERM)
                t_end = time.time() + grace_s
                while time.time() < t_end and any(p.poll() is None for p in procs):
                    time.sleep(poll_s)
                end_all(signal.SIGKILL)
                break
            if all(c == 0 for c in codes):
                break
            time.sleep(poll_s)
    finally:
        for p in procs:
            try:
                p.wait(timeout=grace_s)
            except subprocess.TimeoutExpired:
                pass
        for s, h in old.items():
            signal.signal(s, h)
    return status


def self_launch_script(script, argv, world):
    """`python <script> <argv>` once per rank"""
    return spawn_ranks([sys.executable, script] + list(argv), world)


def self_launch_module(module, argv, world):
    """`python -m <module> <argv>` once per rank (the package stays importable whatever the working directory)"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = root + (os.pathsep + os.environ["PYTHONPATH"] if os.environ.get("PYTHONPATH") else "")
    return spawn_ranks([sys.executable, "-m", module] + list(argv), world, extra_env={"PYTHONPATH": path})
