"""world_size-2 gloo run on CPU of every collective on the hot path: advantage-statistics merge,
flat gradient averaging (aliasing buffer and cat/copy-back forms), parameter broadcast, metric averaging."""

import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

import oracle

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.fixture(scope="module")
def ranks(tmp_path_factory):
    out = tmp_path_factory.mktemp("dist")
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(ROOT / "tests" / "_dist_worker.py"), str(out)]
    done = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert done.returncode == 0, done.stdout[-2000:] + done.stderr[-4000:]
    return [json.loads((out / f"rank{r}.json").read_text()) for r in range(2)]


def test_ranks_are_seeded_differently(ranks):
    assert ranks[0]["first_randperm"] != ranks[1]["first_randperm"]
    assert ranks[0]["first_randperm"] == oracle.Mt19937(7).randperm(16).tolist()
    assert ranks[1]["first_randperm"] == oracle.Mt19937(8).randperm(16).tolist()


def test_advantage_statistics_merge_matches_the_oracle(ranks):
    means = np.array([r["local_mean"] for r in ranks], np.float32)
    vars_ = np.array([r["local_var"] for r in ranks], np.float32)
    mean, var = oracle.merge_mean_var(means, vars_)
    for r in ranks:  # every rank ends with the same merged statistics
        np.testing.assert_allclose(r["merged_mean"], mean, rtol=1e-6)
        np.testing.assert_allclose(r["merged_var"], var, rtol=1e-6)
    # equal-weight formula: var = avg(var_r + (mean_r - mean)^2), not the pooled variance
    np.testing.assert_allclose(var, (vars_ + (means - means.mean(0)) ** 2).mean(0), rtol=1e-6)


def test_gradient_averaging_flat_and_reference_style_agree(ranks):
    expect = (np.array(ranks[0]["flat_local"]) + np.array(ranks[1]["flat_local"])) / 2
    for r in ranks:
        np.testing.assert_allclose(r["flat_reduced"], expect, rtol=1e-6)
        np.testing.assert_allclose(r["cat_reduced"], expect, rtol=1e-6)


def test_parameter_broadcast_and_metric_averaging(ranks):
    assert ranks[0]["params_before"] != ranks[1]["params_before"]
    assert ranks[0]["params_after"] == ranks[0]["params_before"] == ranks[1]["params_after"]
    for r in ranks:
        assert r["averaged"] == {"shared": 0.5, "only0": 1.0, "only1": 1.0}
        assert r["averaged_same_keys"] == {"a": 5.0, "b": 0.5, "c": 3.0} and r["fast_path_taken"] is True
        assert r["averaged_with_none"] == {"a": 1.5, "b": 2.0}
        assert r["gathered"] == [0, 10] and r["stack"] == [[0.0], [1.0]]


def test_kl_driven_schedules_act_on_the_mean_kl_over_ranks(ranks):
    """Fed different KLs, both ranks take the decisions of the averaged sequence (lr_schedule.py:60-62, 287-296)."""
    import sys

    import cusrl_amd

    sys.path.insert(0, str(ROOT / "tests" / "golden"))
    from make_golden import ScheduleProbe

    assert ranks[0]["threshold_lrs"] == ranks[1]["threshold_lrs"]
    assert ranks[0]["mini_batch_wise_lrs"] == ranks[1]["mini_batch_wise_lrs"]
    mean_kls = [(0.004 + 0.006) / 2, (0.05 + 0.01) / 2, 0.011, (0.0001 + 0.0301) / 2]
    single = ScheduleProbe().run(cusrl_amd.hook.ThresholdLRSchedule(desired_kl_divergence=0.01), mean_kls, False)
    np.testing.assert_allclose(ranks[0]["threshold_lrs"], single[:, 0], rtol=1e-9)
    single = ScheduleProbe().run_mini_batch_wise(cusrl_amd.hook.MiniBatchWiseLRSchedule(desired_kl_divergence=0.01),
                                                 cusrl_amd.hook.OnPolicyPreparation(), mean_kls, 2)
    np.testing.assert_allclose(ranks[0]["mini_batch_wise_lrs"], single[:, 0], rtol=1e-9)
    assert len(set(ranks[0]["threshold_lrs"])) > 1  # the sequence does move the learning rate


def test_native_communicator_start_up_is_collective_safe_at_every_stage(ranks):
    """establish_native_comm with a fault injected on one rank at one stage (CUSRL_COMM_FAULT): both ranks finish (no rank
    is left inside a collective its peer never enters), report the same outcome, the faulty rank names the fault, a
    communicator with a possibly half-issued collective is aborted (not destroyed), and the process group stays in step."""
    for scenario in ("none", "create:0", "create:1", "probe:0", "probe:1", "capture:0", "capture:1", "replay:0", "replay:1"):
        a, b = (r["comm_protocol"][scenario] for r in ranks)
        assert a["ok"] == b["ok"] == (scenario == "none"), scenario
        assert a["in_step"] and b["in_step"], scenario
        if scenario == "none":
            assert not (a["closed"] or a["aborted"] or b["closed"] or b["aborted"])
            continue
        stage, faulty = scenario.split(":")
        reasons = [a["reason"], b["reason"]]
        assert "injected fault" in reasons[int(faulty)] and reasons[1 - int(faulty)], scenario
        for rank, outcome in enumerate((a, b)):
            if stage == "create":  # the healthy rank created a communicator nothing was enqueued on: plain destroy
                assert outcome["closed"] == (rank != int(faulty)) and not outcome["aborted"], scenario
            else:  # the healthy rank's probe / replay is enqueued and will never complete (or its capture broke): abort, on every rank
                assert outcome["aborted"] and not outcome["closed"], scenario
