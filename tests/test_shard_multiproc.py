"""The library's N > 1 path with REAL second (third, fourth) processes, on one GPU (run on the GPU box: -m gpu).

RCCL refuses two ranks on one device, so tests/rccl_standin (a test-only re-statement of the eight RCCL entry points over
POSIX shared memory, loaded through rpf_shard.cpp's RPF_RCCL_LIB hook) stands in for it: rpf_shard_create(world > 1), both
communicators of the two-pipeline schedule, rpf_shard_develop / rpf_shard_develop_async and bench.py's library branch run end
to end with one process per rank, bit-identical to the oracle's bands.  Also here: the bounded waits (a wedged exchange ends
in an error that names the rank, not in a hang) and setters / destroy while async steps are still queued."""
import os
import re
import subprocess
import sys

import pytest

from tests import util

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(world, script_args, port, env_extra=None, timeout=600):
    env = dict(os.environ, RPF_RCCL_LIB=util.build_standin(), HSA_ENABLE_IPC_MODE_LEGACY="0", **(env_extra or {}))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
           "127.0.0.1", "--master-port", str(port)] + script_args
    return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_step_with_real_processes(gpu, world):
    """6000x4000 in `world` bands, one PROCESS per band: exchange-then-launch and the two-pipeline schedule (two
    communicators per rank, a thread each), halo rows zeroed beforehand so that only the exchange can have filled them;
    then a curve setter and add_mask while async steps are queued, and an editor destroyed with steps queued."""
    out = _run(world, [os.path.join(ROOT, "tests", "mp_shard_worker.py")], 29541 + world)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    # (by rank, over the whole text: the ranks share one pipe, and two of them may land on one line)
    oks = {int(r) for r in re.findall(r"rank (\d+) ok: band ", out.stdout)}
    assert oks == set(range(world)), out.stdout[-2000:]


def test_a_wedged_exchange_ends_in_an_error_not_a_hang(gpu):
    """The stand-in drops rank 0's sends: rank 1's halo exchange never completes.  rpf_shard_drain on rank 1 must give up at
    the shard's deadline (1.5 s here) with an error that names the rank, its neighbours and the phase."""
    out = _run(2, [os.path.join(ROOT, "tests", "mp_shard_worker.py"), "--mode", "wedge", "--width", "1024", "--height", "256"],
               29549, {"RPF_STANDIN_DROP_SENDS_FROM": "0", "RPF_STANDIN_TIMEOUT_MS": "6000"}, timeout=300)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    line = re.search(r"rank 1 wedge: ([^\n]*)", out.stdout)
    assert line and "deadline fired" in line.group(1) and "rank 1 of 2" in line.group(1), out.stdout[-2000:]


@pytest.mark.parametrize("schedule", ["serial", "pingpong"])
def test_bench_library_branch_with_two_processes(gpu, schedule):
    """`bench.py --gpus 2` as the driver launches it, the LIBRARY's communicators created on both ranks (through the
    stand-in; the process group itself over gloo): the branch that runs on the driver's multi-GPU node."""
    out = _run(2, [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--preheat", "4", "--backend",
                   "gloo", "--exchange", "rccl", "--schedule", schedule], 29551 + (schedule == "serial"),
               {"RPF_BENCH_TRY_RCCL": "1"})
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    r = util.bench_line(out.stdout)   # ONE line, from rank 0
    c = r["config"]
    assert r["n_gpus"] == 2 and c["world_size"] == 2 and c["mode"] == "strong" and c["schedule"] == schedule
    assert "librpf_hip.so" in c["exchange"] and "fallback" not in c["exchange"]
    assert [x["band"] for x in c["ranks"]] == [[0, 2000], [2000, 2000]]
    assert [x["neighbours"] for x in c["ranks"]] == [[-1, 1], [0, -1]]
    assert r["value"] > 0
