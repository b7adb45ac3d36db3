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


def reserve_port():
    """(socket, port): a free port that STAYS bound -- not listening, SO_REUSEADDR -- for as long as the socket is open.  The
    rendezvous store of rank 0 (which binds with SO_REUSEADDR too) can still listen on it, while a process that merely asks
    for a free port, or binds without the option, cannot take it between this call and rank 0's bind."""
    s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    s.bind(("127.0.0.1", 0))
    return s, s.getsockname()[1]


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


def arm_parent_death_signal():
    """called by a rank at its own start-up (bench.py, match_stage: before anything else runs) when the launcher below
    started it: SIGTERM should the launcher die first -- a killed parent must not leave ranks holding GPUs.  Done here, in
    the child's own interpreter, not between fork and exec of a parent that may already run threads."""
    if os.environ.get("PHYLIGN_LAUNCHER_PID"):
        try:
            import ctypes
            ctypes.CDLL(None, use_errno=True).prctl(1, int(signal.SIGTERM), 0, 0, 0)      # PR_SET_PDEATHSIG
            if os.getppid() != int(os.environ["PHYLIGN_LAUNCHER_PID"]):                   # it died before we got here
                os.kill(os.getpid(), signal.SIGTERM)
        except Exception:
            pass


def spawn_ranks(cmd, world, poll_s=0.2, grace_s=10.0, extra_env=None):
    """Runs `cmd` (argv list) once per rank and waits.  Returns the exit status of the job: 0 when every rank
    returned 0, else the first failing rank's status (a rank killed by signal n counts as 128 + n).  A failing rank, or
    SIGINT / SIGTERM to the launcher, ends the others the same way: SIGTERM to their process groups, `grace_s` seconds,
    then SIGKILL -- a rank stuck in a collective does not keep its GPU.  MASTER_PORT is held bound by the launcher
    (reserve_port) until the ranks are gone, so nobody else is handed it between its choice and rank 0's bind."""
    holder, port = reserve_port()
    launcher = {"PHYLIGN_LAUNCHER_PID": str(os.getpid())}
    # start_new_session: every rank leads its own process group (ended as a group); no code runs between fork and exec
    procs = [subprocess.Popen(cmd, env=dict(rank_env(r, world, port), **launcher, **(extra_env or {})), start_new_session=True)
             for r in range(world)]
    stop = []                                             # signal numbers received by the launcher

    def end_all(sig):
        for p in procs:
            if p.poll() is None:
                try:
                    os.killpg(p.pid, sig)                 # exactly the process groups started above
                except (ProcessLookupError, PermissionError):
                    pass

    def term_then_kill():
        end_all(signal.SIGTERM)
        t_end = time.time() + grace_s
        while time.time() < t_end and any(p.poll() is None for p in procs):
            time.sleep(poll_s)
        end_all(signal.SIGKILL)

    old = {s: signal.signal(s, lambda signum, _frame: stop.append(signum)) for s in (signal.SIGINT, signal.SIGTERM)}
    status = 0
    try:
        while True:
            if stop:
                status = 128 + stop[0]
                sys.stderr.write(f"[launch] signal {stop[0]}: ending the ranks\n")
                term_then_kill()
                break
            codes = [p.poll() for p in procs]
            bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if bad:
                r, c = bad[0]
                status = 128 - c if c < 0 else c
                sys.stderr.write(f"[launch] rank {r} exited with status {c}: ending the other ranks\n")
                term_then_kill()
                break
            if all(c == 0 for c in codes):
                break
            time.sleep(poll_s)
    finally:
        if any(p.poll() is None for p in procs):          # an exception in the loop above: nobody is left behind
            term_then_kill()
        for p in procs:
            try:
                p.wait(timeout=grace_s)
            except subprocess.TimeoutExpired:
                pass
        for s_, h in old.items():
            signal.signal(s_, h)
        holder.close()
    return status


def self_launch_script(script, argv, world):
    """`python <script> <argv>` once per rank"""
    return spawn_ranks([sys.executable, script] + list(argv), world)


def self_launch_module(module, argv, world):
    """`python -m <module> <argv>` once per rank (the package stays importable whatever the working directory)"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = root + (os.pathsep + os.environ["PYTHONPATH"] if os.environ.get("PYTHONPATH") else "")
    return spawn_ranks([sys.executable, "-m", module] + list(argv), world, extra_env={"PYTHONPATH": path})
