"""GPU: the RCCL ("nccl") branch of ppmstereo_amd.dist on real hardware, as far as ONE GPU can take it.  A fresh child process forms a world-size-1
nccl process group on the MI355X (rendezvous before any GPU call, the environment bench.launch_ranks sets) and runs the frame-sharded cascade with
every exchange forced through the communicator (dist.FrameShard(force_comm=True)): librccl loads, the communicator initialises, device tensors take
the nccl branch of comm_device(), all_gather_into_tensor / batch_isend_irecv / the asynchronous handles and stream waits run -- everything short of a
peer.  Multi-GPU stays "unmeasured on hardware" until the driver's 8-GPU node has produced a SCALE record."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_world_size_one_nccl_group_runs_the_sharded_cascade_bit_identically():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.pop("PPMS_DIST_BACKEND", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_world1_child.py")], capture_output=True, text=True, cwd=ROOT, env=env, timeout=420)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    print("world-size-1 RCCL run:", res)
    assert res["backend"] == "nccl" and res["max_over_ranks"] == 3.5 and res["sum_over_ranks"] == 2.25
    assert res["all_gather_ok"] and res["gather_many_ok"] and res["gather_kept_ok"]
    assert res["engine_sharded"], "the engine dropped the shard: the exchanges did not run"
    assert res["finite"] and res["npred"] == 8
    # one rank holds every frame: same tiles, same launch plan, same summation order as the unsharded cascade -- the same bits
    assert res["disp_equal"] and res["unc_equal"], res
