"""RCCL on the one GPU this project can reach: every collective of the multi-GPU design issued through
`init_process_group("nccl", world_size=1)` on device tensors, next to a hipGraph capture with the process group's watchdog alive,
bitwise equal to the non-distributed computation.  No scaling curve can come out of one GPU -- this proves the code path executes
on RCCL, nothing about its speed."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_collectives_and_graph_capture_under_a_one_rank_nccl_group():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    env.pop("SUPIR_GRAPH_CAPTURE_MODE", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_single_rank_worker.py")], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=900)
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("RCCL_RESULT ")]
    assert p.returncode == 0 and line, p.stdout[-2000:] + "\n" + p.stderr[-4000:]
    r = json.loads(line[-1][len("RCCL_RESULT "):])
    print(r)
    assert r["backend"] == "nccl" and r["capture_mode"] == "thread_local"
    assert r["broadcast_buckets"][0] >= 2 and r["broadcast_buckets"][1] >= 2 and r["broadcast_identity"]
    assert r["sync_autotune_changed"] == 0 and r["sync_autotune_identity"] and r["max_over_ranks"] == 1.25
    assert r["eager_equal_after_broadcast"] and r["graph_equal_eager"]
    assert r["tile_parallel_sampler_equal"] and r["tile_parallel_vae_equal"]
    # round 5: a meta-constructed replica that received its state computes the same bits; a host-resident buffer survives the broadcast
    assert r["replica_on_device"] and r["replica_equal"] and r["host_buffer_buckets"] >= 2 and r["host_buffer_identity"]
