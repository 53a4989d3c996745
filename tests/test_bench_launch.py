"""bench.py --gpus N without a launcher: the torch.distributed.run child runs under --launch-timeout in a process group of its own; a rank that
never arrives (a hung RCCL bootstrap on first multi-GPU contact) ends the launch inside the timeout with a non-zero code and the tail of every
rank's stderr, instead of costing the driver its whole time limit.  The launch path is the same with and without GPUs (the ranks of the CPU
run fail at "needs a GPU"; the stalled one never gets there)."""
import os
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env, timeout_s, launch_timeout):
    env = dict(os.environ, FENRIS_BENCH_SHARE_DEVICE="1", **extra_env)
    t0 = time.time()
    pr = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "ns", "--cells", "8", "--steps", "1", "--warmup", "0",
                         "--no-cpu-baseline", "--no-traffic", "--no-secondary", "--launch-timeout", str(launch_timeout)],
                        env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout_s)
    return pr, time.time() - t0


def _check_timed_out(pr, seconds):
    assert pr.returncode == 124, (pr.returncode, pr.stderr[-2000:])
    assert seconds < 200
    assert "--launch-timeout" in pr.stderr and "process group was killed" in pr.stderr
    # the stalled rank's own stderr is in the report
    assert "FENRIS_BENCH_TEST_STALL_RANK: stalling before the first barrier" in pr.stderr
    assert '"metric"' not in pr.stdout


def test_stalled_ranks_end_the_launch_inside_the_timeout():
    # (without a GPU a rank that does arrive fails at once and takes the launch down with it: here every rank stalls)
    _check_timed_out(*_run({"FENRIS_BENCH_TEST_STALL_RANK": "all"}, 300, 30))


@pytest.mark.gpu
def test_one_stalled_rank_ends_the_launch_inside_the_timeout():
    """rank 1 never joins, rank 0 waits for it in init_process_group: what a hung bootstrap on first multi-GPU contact looks like"""
    _check_timed_out(*_run({"FENRIS_BENCH_TEST_STALL_RANK": "1"}, 400, 90))


@pytest.mark.gpu
def test_share_device_launch_prints_the_bootstrap_lines():
    """two ranks on one device (validation mode): the N = 2 line comes back, and a failed launch would have shown these lines per rank"""
    pr, _ = _run({}, 900, 600)
    assert pr.returncode == 0, pr.stderr[-3000:]
    assert '"n_gpus": 2' in pr.stdout
