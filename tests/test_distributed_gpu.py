"""The data-parallel path on RCCL (backend ``nccl``): one torchrun rank on the single GPU a test box has, two ranks when
two GPUs are visible.  Both collective routes are exercised: torch.distributed's process group (eager all-reduce between
two graphs) and the C-ABI communicator (``cusrl_allreduce_mean`` captured inside the minibatch step's hipGraph)."""

import json
import math
import os
import socket
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _run(tmp_path, world: int, compile_: str, native: str, share_gpu: bool = False, extra_env=None, want_output=False):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    if share_gpu:
        env["CUSRL_SHARE_GPU"] = "1"
    env.update(extra_env or {})
    for _ in range(3):  # a rendezvous port picked free may be taken by the time the store binds it: retry that, only that
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
               "127.0.0.1", "--master-port", str(port), str(ROOT / "tests" / "_dist_gpu_worker.py"), str(tmp_path), compile_, native]
        done = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=420)
        if done.returncode == 0 or "EADDRINUSE" not in done.stderr:
            break
    # a rank that has written its results and passed the last barrier has done everything this test checks; an abort of
    # ProcessGroupNCCL's watchdog thread while the process goes away (a torch / RCCL teardown race) is not this package's
    finished = all(f"WORKER_RESULTS_WRITTEN rank {r}" in done.stdout for r in range(world))
    assert done.returncode == 0 or (finished and "ProcessGroupNCCL" in done.stderr), done.stdout[-2000:] + done.stderr[-6000:]
    results = [json.loads((tmp_path / f"rank{r}.json").read_text()) for r in range(world)]
    return (results, done.stdout + done.stderr) if want_output else results


@pytest.mark.parametrize("compile_,native", [("0", "0"), ("1", "0"), ("0", "1"), ("1", "1")])
def test_ppo_preset_over_rccl_single_rank(tmp_path, compile_, native):
    (result,) = _run(tmp_path, 1, compile_, native)
    assert result["world"] == 1 and result["mean"] == result["local_mean"] and result["var"] == result["local_var"]
    assert result["flat_mean"] == 1.0
    for key in ("Agent/value_loss", "Agent/surrogate_loss", "Agent/entropy_loss", "Agent/kl_divergence"):
        assert math.isfinite(result["info"][key]), key
    assert result["native"] == (native == "1")
    # the log average ran over RCCL on the process-group stream and did NOT queue behind the ~80 ms kernel parked on the caller's
    # stream (the worker asserts its values): the pipelined trainer reads a log while the next rollout runs
    assert result["log_average_s"] < 0.05, result["log_average_s"]
    if native == "1":  # the C-ABI entry points really carried the collectives
        assert result["allreduce_calls"] > 0 and result["allgather_calls"] > 0
        if compile_ == "1":
            assert result["single_graph"] and all(result["single_graph"])  # all-reduce captured inside the step graph
            assert result["update_graph_replays"] > 0  # ... and the steps replay from ONE graph per update
    elif compile_ == "1":
        assert result["single_graph"] and not any(result["single_graph"])  # eager all-reduce between two graphs
    # second part of round 6: no route launches a squared-norm pass between the all-reduce and the step — the step launch measures
    # the averaged gradients' norm itself
    # (compile=False keeps torch's optimizer and clip_grad_norm_ over the flat buffer)
    assert result["sumsq_launches"] == 0 and (result["normed_steps"] > 0) == (compile_ == "1")


def test_unjoined_step_of_a_multi_rank_job_changes_no_bit_one_rccl_rank(tmp_path):
    """Second part of round 6: with the critic on its own stream-branch (forced here: the worker's minibatches are below the
    size where it pays) the step of a multi-rank job stays UNJOINED inside the whole-update graph — the critic's window assembled
    on the critic's stream, ONE all-reduce over the whole flat buffer on the main stream behind both assemblies, each window
    stepped on its own stream by a launch that measures the averaged gradients' norm itself (``cusrl_adam_step_normed``).  Against
    the joined form of the same job (one assembly, the all-reduce, one step launch over everything): the same parameters to the
    bit — both forms split the norm's sum alike."""
    (tmp_path / "unjoined").mkdir(), (tmp_path / "joined").mkdir()
    (unjoined,) = _run(tmp_path / "unjoined", 1, "1", "1", extra_env={"CUSRL_CONCURRENT_CRITIC": "1"})
    (joined,) = _run(tmp_path / "joined", 1, "1", "1", extra_env={"CUSRL_CONCURRENT_CRITIC": "1", "CUSRL_TWO_WINDOW_STEP": "0"})
    assert unjoined["native"] and unjoined["update_graph_replays"] > 0 and unjoined["two_window_steps"] > 0, unjoined
    assert joined["update_graph_replays"] > 0 and joined["two_window_steps"] == 0
    assert unjoined["sumsq_launches"] == joined["sumsq_launches"] == 0 and unjoined["normed_steps"] > joined["normed_steps"] > 0
    assert unjoined["param_bytes"] == joined["param_bytes"] and unjoined["param_sum"] == joined["param_sum"]
    for key in ("Agent/value_loss", "Agent/surrogate_loss", "Agent/kl_divergence"):
        assert math.isfinite(unjoined["info"][key]) and unjoined["info"][key] == joined["info"][key], key


@pytest.mark.parametrize("fault,needle", [
    ({"CUSRL_RCCL_LIBRARY": "/nonexistent/librccl.so"}, "RCCL is not available"),  # the library the user named cannot be loaded
    ({"CUSRL_COMM_FAULT": "probe:0"}, "injected fault"),    # the eager probe fails: communicator aborted, not destroyed
    ({"CUSRL_COMM_FAULT": "capture:0"}, "injected fault"),  # RCCL "cannot be captured" on this stack
    ({"CUSRL_COMM_FAULT": "replay:0"}, "injected fault"),   # the captured all-reduce cannot be replayed
])
def test_c_abi_route_falls_back_to_torch_distributed_and_says_so(tmp_path, fault, needle):
    """The default route of an RCCL job verifies itself at start-up (establish_native_comm); when a stage fails the job logs
    the switch ONCE and trains through torch.distributed's collectives — eager all-reduce between two graphs per step."""
    (result,), output = _run(tmp_path, 1, "1", "1", extra_env=fault, want_output=True)
    assert not result["native"] and "torch.distributed rccl" in result["route"] and needle in result["route"]
    assert output.count("C-ABI RCCL communicator unavailable") == 1 and needle in output
    # at most the start-up probe (and, for a replay fault, the capture of the second probe) went through the abandoned communicator
    assert result["allreduce_calls"] <= (2 if "replay" in str(fault) else 1)
    assert result["single_graph"] and not any(result["single_graph"])
    for key in ("Agent/value_loss", "Agent/surrogate_loss", "Agent/entropy_loss", "Agent/kl_divergence"):
        assert math.isfinite(result["info"][key]), key


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs")
@pytest.mark.parametrize("fault", ["create:1", "probe:1", "capture:0", "capture:1", "replay:1"])
def test_two_rccl_ranks_agree_on_the_fallback_when_one_rank_fails(tmp_path, fault):
    ranks = _run(tmp_path, 2, "1", "1", extra_env={"CUSRL_COMM_FAULT": fault})
    _assert_lockstep(ranks)
    assert not any(r["native"] for r in ranks)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs")
@pytest.mark.parametrize("compile_,native", [("1", "0"), ("1", "1"), ("0", "1")])
def test_two_rccl_ranks_stay_in_lockstep(tmp_path, compile_, native):
    """Two ranks, 3 iterations of the preset: every rank ends with bit-identical parameters (same broadcast start, same
    averaged gradients), drew different permutations, and merged the advantage statistics like the oracle."""
    ranks = _run(tmp_path, 2, compile_, native)
    _assert_lockstep(ranks)
    for r in ranks:
        assert r["native"] == (native == "1"), r["route"]


def _assert_lockstep(ranks):
    assert ranks[0]["first_perm"] != ranks[1]["first_perm"]  # per-rank generator streams (seed + rank, misc.py:163)
    assert ranks[0]["param_bytes"] == ranks[1]["param_bytes"] and ranks[0]["param_sum"] == ranks[1]["param_sum"]
    assert ranks[0]["advantage_head"] != ranks[1]["advantage_head"]  # different envs, different data ...
    means = np.array([r["local_mean"] for r in ranks], np.float32)
    vars_ = np.array([r["local_var"] for r in ranks], np.float32)
    mean, var = oracle.merge_mean_var(means, vars_)
    for r in ranks:
        np.testing.assert_allclose(r["mean"], mean, rtol=1e-6)  # ... merged by the reference's equal-weight formula
        np.testing.assert_allclose(r["var"], var, rtol=1e-6)
        assert r["flat_mean"] == 1.5
        for key in ("Agent/value_loss", "Agent/surrogate_loss", "Agent/entropy_loss", "Agent/kl_divergence"):
            assert math.isfinite(r["info"][key]) and r["info"][key] == ranks[0]["info"][key]  # rank-averaged logs


@pytest.mark.parametrize("compile_", ["0", "1"])
def test_two_ranks_sharing_one_gpu_over_gloo_stay_in_lockstep(tmp_path, compile_):
    """The whole multi-rank agent path on the ONE GPU a test box has: two torchrun ranks both drive cuda:0, the process
    group is gloo (RCCL refuses two ranks on one device).  3 iterations of the preset: parameter broadcast at init
    (actor_critic.py:224), flat gradient averaging after every backward (distributed.py:145-172), advantage statistics merged
    across ranks (distributed.py:175-183), rank-averaged logs (trainer.py:387) — every rank must end with bit-identical
    parameters although it saw different envs and drew different permutations.  Under compile=True the env steps replay from
    hipGraphs and every minibatch step is two graphs with the (host-staged) all-reduce between them."""
    ranks = _run(tmp_path, 2, compile_, "1", share_gpu=True)
    _assert_lockstep(ranks)
    for r in ranks:
        assert r["world"] == 2 and not r["native"] and "gloo" in r["route"]
        if compile_ == "1":
            assert r["single_graph"] and not any(r["single_graph"])  # collective outside capture on this route
            assert r["captured_env_steps"] == 8


@pytest.mark.parametrize("compile_", ["0", "1"])
def test_split_gradient_allreduce_changes_no_bit_two_ranks_over_gloo(tmp_path, compile_):
    """CONFIG.split_gradient_allreduce (critic differentiated first, its window averaged on the branch stream while the actor's
    backward runs; cusrl/utils/distributed.py:145-172 is one all-reduce behind the whole backward): two ranks on the one GPU
    over gloo end with the very parameters the single collective gives, rank for rank."""
    (tmp_path / "one").mkdir(), (tmp_path / "split").mkdir()
    single = _run(tmp_path / "one", 2, compile_, "1", share_gpu=True)
    split = _run(tmp_path / "split", 2, compile_, "1", share_gpu=True, extra_env={"CUSRL_SPLIT_ALLREDUCE": "1"})
    _assert_lockstep(split)
    for a, b in zip(single, split):
        assert not a["split_backward"] and b["split_backward"] == (compile_ == "1")  # (the branch stream exists under compile=True)
        assert a["param_bytes"] == b["param_bytes"] and a["param_sum"] == b["param_sum"]


@pytest.mark.parametrize("native", ["0", "1"])
def test_split_gradient_allreduce_changes_no_bit_one_rccl_rank(tmp_path, native):
    """The same on one RCCL rank, both collective routes: with the C ABI the two windows' all-reduces are nodes of the step's
    hipGraph (two communicators, two streams), with torch.distributed they run eagerly between the step's two graphs."""
    (tmp_path / "one").mkdir(), (tmp_path / "split").mkdir()
    (single,) = _run(tmp_path / "one", 1, "1", native)
    (split,) = _run(tmp_path / "split", 1, "1", native, extra_env={"CUSRL_SPLIT_ALLREDUCE": "1"})
    assert split["split_backward"] and not single["split_backward"]
    assert split["native"] == (native == "1") and single["param_bytes"] == split["param_bytes"]
    if native == "1":
        assert split["single_graph"] and all(split["single_graph"])  # both windows' collectives captured inside the step
        assert split["allreduce_calls"] > single["allreduce_calls"]


def test_bench_launches_its_own_ranks_and_prints_one_line(tmp_path):
    """``python bench.py --gpus 2 --share-gpu`` (test-only flag): bench.py starts its own two ranks under
    torch.distributed.run, both drive cuda:0 over gloo, rank 0 prints ONE JSON line that says what it is."""
    cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--share-gpu", "--steps", "3", "--warmup", "4", "--envs-per-gpu", "512",
           "--no-cpu-baseline", "--no-kernel-pass", "--no-scale-pass", "--no-pin"]
    done = subprocess.run(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"), capture_output=True, text=True, timeout=420)
    assert done.returncode == 0, done.stdout[-2000:] + done.stderr[-4000:]
    lines = [line for line in done.stdout.splitlines() if line.startswith("{")]
    assert len(lines) == 1, done.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["scaling"] == "weak" and line["value"] > 0
    config = line["config"]
    assert config["share_gpu"] and "NOT a scaling measurement" in config["test_only"]
    assert config["backend"] == "gloo" and config["rccl_ranks"] == 0 and "gloo" in config["collectives"]
    assert config["parallelism"] == "dp2" and config["env_steps_per_iteration"] == 2 * 512 * 24
    assert config["captured_env_steps"] == 24


def test_c_abi_communicator_of_one_rank_inside_a_hipgraph():
    """cusrl_comm_* through ctypes in this process (world of one needs no launcher): the three collectives, their
    argument errors, and an all-reduce captured into a hipGraph and replayed."""
    from cusrl_amd import _native
    from cusrl_amd.utils.distributed import RcclComm

    lib = _native.lib()
    assert lib.cusrl_comm_available() == 1
    comm = RcclComm(1, 0, device=torch.device("cuda:0"))
    assert lib.cusrl_comm_world_size(comm._handle) == 1
    x = torch.arange(1000, dtype=torch.float32, device="cuda:0")
    expect = x.clone()
    comm.allreduce_mean_(x)
    assert torch.equal(x, expect)                                    # the mean over one rank
    stacked = comm.allgather(torch.tensor([1.0, 4.0], device="cuda:0"))
    assert stacked.shape == (1, 2) and stacked.tolist() == [[1.0, 4.0]]
    comm.broadcast_(x, 0)
    assert torch.equal(x, expect)
    stream, graph = torch.cuda.Stream(), torch.cuda.CUDAGraph()
    y = torch.ones(4096, device="cuda:0")
    torch.cuda.synchronize()
    with torch.cuda.graph(graph, stream=stream):
        y.mul_(2.0)
        comm.allreduce_mean_(y)
        y.add_(1.0)
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(y, torch.full_like(y, 15.0))                  # ((1*2+1)*2+1)*2+1
    with pytest.raises(TypeError):
        comm.allreduce_mean_(torch.zeros(4, dtype=torch.float64, device="cuda:0"))
    assert lib.cusrl_allreduce_mean(x.data_ptr(), -1, comm._handle, None) == -1      # CUSRL_E_INVALID
    assert lib.cusrl_allreduce_mean(x.data_ptr(), 4, None, None) == -1
    assert lib.cusrl_broadcast(x.data_ptr(), 4, 3, comm._handle, None) == -1         # root outside the world
    assert lib.cusrl_comm_create(None, 1, 0, None) == -1
    comm.close()
