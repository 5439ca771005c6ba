"""Data parallelism of the PRODUCT path on the GPU: two ranks (gloo, sharing the one GPU of the test box; on a
node it is one RCCL rank per GPU, same code) train on the two halves of a batch; their weights after three
steps must equal the single-process run on the whole batch -- instance-norm models, so that per-replica
statistics do not change the maths.  Run eagerly and as the multi-graph replay (four graphs around the all-reduces)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import report

HERE = os.path.dirname(os.path.abspath(__file__))
WORKER = os.path.join(HERE, "_dp_gpu_worker.py")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(world, mode, out):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(r), LOCAL_RANK=str(r),
                   WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, WORKER, out, mode], env=env))
    # collect every rank's exit code; on any failure kill the others (a rank left blocked in a gloo collective would hold
    # the GPU until the process-group timeout)
    codes = []
    try:
        for p in procs:
            codes.append(p.wait(timeout=600))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
                p.wait()
    assert codes == [0] * world, codes
    return dict(np.load(out))


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["eager", "graph", "graph-bf16", "graph-rel", "graph-fused"])
def test_two_rank_step_equals_whole_batch_step(tmp_path, mode):
    """fp32 eager / recorded; the all-bf16 models (config C3's arithmetic); the relativistic losses, whose non-linearity needs
    the GLOBAL means (a 2-float all-reduce inside the step: six graphs per step); the fused step (no predict pass).  In every
    mode the critic's bucket is issued before the generator's training-mode forward and waited for after it."""
    ref_mode = "eager" + ("-" + mode.split("-")[1] if "-" in mode else "")
    one = _run(1, ref_mode, str(tmp_path / "one.npz"))
    two = _run(2, mode, str(tmp_path / "two.npz"))
    worst = 0.0
    names = [k for k in one if k != "losses" and not k.startswith("init/")]
    for m in ("G/", "D/"):
        # error of the three-step UPDATE relative to the model's largest update
        upd = max(float(np.max(np.abs(one[k] - one["init/" + k]))) for k in names if k.startswith(m))
        assert upd > 0
        for k in names:
            if k.startswith(m):
                e = float(np.max(np.abs(two[k] - one[k]))) / upd
                worst = max(worst, e)
                assert e < 5e-3, (k, e)
    le = float(np.max(np.abs(two["losses"] - one["losses"]) / (np.abs(one["losses"]) + 1e-3)))
    report("dp2 (%s) vs whole batch: worst weight err=%.2e  loss err=%.2e" % (mode, worst, le))
    assert le < 1e-3


@pytest.mark.gpu
def test_rccl_one_rank_group_is_bit_identical():
    """scripts/dp_nccl_smoke.py: the DP code path over a real (1-rank) RCCL communicator -- eager and as the four
    hipGraphs around the all-reduces -- reproduces the single-process step bit for bit"""
    script = os.path.join(os.path.dirname(HERE), "scripts", "dp_nccl_smoke.py")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, script], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.strip().endswith("ok")
