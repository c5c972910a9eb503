"""Self-launch of a one-process-per-GPU job from a plain `python script.py --gpus N`.

The reference has no launcher to mirror: `NeuralSLAM` is single-device (atdn_vslam/slam_framework/neural_slam.py:51) and its
scripts are started as `python evaluate_odometry.py` / `python train_odometry.py`. The multi-GPU contract is this build's own:
one process per GPU under `python -m torch.distributed.run`. A harness that types `python bench.py --gpus 8` (the way it types
`--gpus 1`) must get the same job, so the script calls `spawn_ranks_if_needed()` FIRST — before torch is imported, and long
before any HIP call: the parent only starts the N ranks as a child process, lets them write straight to its stdout / stderr
(rank 0's one JSON line) and exits with their return code. It never re-execs and never touches the GPU (on the GPU pool a
process that has initialised HIP must not exec another program). A rank that fails ends the job non-zero: inside the ranks
through `sharding.rendezvous` (every rank raises), here through torch.distributed.run's own exit code.

Import cost: standard library only.
"""
import os
import socket
import subprocess
import sys


def requested_gpus(argv, default=1):
    """Value of `--gpus N` / `--gpus=N` in an argument list, without argparse (nothing else is parsed here)."""
    for i, a in enumerate(argv):
        if a == "--gpus" and i + 1 < len(argv):
            return int(argv[i + 1])
        if a.startswith("--gpus="):
            return int(a.split("=", 1)[1])
    return default


def free_port():
    """A TCP port nobody listens on right now, on the loopback interface the rendezvous uses."""
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def host_cores():
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def launch_command(script, argv, gpus, port):
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
            "--master-addr", "127.0.0.1", "--master-port", str(port), script] + list(argv)


def spawn_ranks_if_needed(script, argv=None, env=None):
    """If `--gpus N` with N > 1 was asked for and this process is not already a rank of a launched job (WORLD_SIZE unset),
    start the N ranks and return their exit code; otherwise return None and let the caller run as a rank / single process.

        rc = spawn_ranks_if_needed(__file__)
        if rc is not None:
            sys.exit(rc)
    """
    argv = list(sys.argv[1:] if argv is None else argv)
    env = dict(os.environ if env is None else env)
    gpus = requested_gpus(argv)
    if gpus <= 1 or "WORLD_SIZE" in env:
        return None
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["ATDN_SELF_LAUNCHED"] = "1"
    # the ranks share this host's cores: each gets its share unless the user chose a figure (torch.distributed.run would
    # otherwise export OMP_NUM_THREADS=1, which the rank could not tell from a user's explicit 1)
    if "OMP_NUM_THREADS" in env:
        env["ATDN_OMP_FROM_USER"] = "1"
    else:
        env["OMP_NUM_THREADS"] = str(max(1, host_cores() // gpus))
    cmd = launch_command(os.path.abspath(script), argv, gpus, free_port())
    sys.stdout.flush()
    sys.stderr.flush()
    # stdout / stderr are inherited: rank 0's JSON line reaches the caller's stdout as it is written
    return subprocess.run(cmd, env=env).returncode
