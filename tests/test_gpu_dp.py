"""BASELINE config 5 (rays sharded data-parallel, one gradient all-reduce per step; R:lse_nerf/lse_pipeline.py:95-98,
R:train.py:104,146-167) with the REAL HIP model on two ranks.  The ranks are fresh child processes that tests/conftest.py
launches from pytest_configure -- before this pytest process makes its first GPU call, because ranks must never be spawned
from a process that has initialised the GPU -- running tools/dp_rehearsal.py: 2 x 2048 rays, the reference's default
configuration (visibility pre-pass on, 4-level 128^3 grid), occupancy refresh at steps 0 and 320, three optimizer steps, for the
plain / pipelined / sharded / overlap exchanges of lsenerf_amd.dist and the step replayed as a HIP graph up to the backward pass
with the exchange and Adam behind it ("graphed"), against ONE process on the full 4096-ray batch.  A one-GPU
box cannot host two RCCL ranks, so the two-rank collectives run over gloo.  The RCCL calls themselves (async all_reduce on slices
of the flat buffer, reduce_scatter_tensor / all_gather_into_tensor of the sharded Adam, broadcasts of the float and bool grids)
are executed by a second child: ONE rank over backend "nccl" with every collective forced (dist.SINGLE_RANK_COLLECTIVES) -- each
is then the identity, so every exchange must reproduce the single process.  Sums over several RCCL ranks run in bench.py --gpus N
on a multi-GPU node.  These tests only read the reports."""
import pytest

pytestmark = pytest.mark.gpu

MODES = ("plain", "pipelined", "sharded", "overlap", "graphed")


def _report(r):
    if not r.get("launched"):
        pytest.skip(f"rehearsal not launched: {r.get('reason')}")
    assert r["returncode"] is not None
    assert r["report"] is not None, f"the rehearsal wrote no report (rc {r['returncode']}):\n{r['log_tail']}"
    return r["report"], r


def test_two_rank_rehearsal_ran_every_exchange(dp_rehearsal):
    rep, r = _report(dp_rehearsal)
    assert rep["world"] == 2 and rep["rays"] == 4096 and rep["steps"] == 3
    assert set(rep["modes"]) == set(MODES), r["log_tail"]
    assert rep["gpu_sharing"].startswith("ranks take turns")
    # the single-process reference rendered a real workload (pre-pass culling on, carved grid)
    assert min(rep["single_process"]["samples_per_step"]) > 100_000
    assert 0.02 < rep["single_process"]["occupied_fraction_after_refresh"] < 0.9


@pytest.mark.parametrize("mode", MODES)
def test_sharded_step_equals_one_process_on_the_full_batch(mode, dp_rehearsal):
    rep, _ = _report(dp_rehearsal)
    e = rep["modes"][mode]
    # two runs of ONE process differ by `floor` (float atomics: 0.7e-6 .. 1.4e-6 over the rounds' runs), and the comparison below is
    # between two such runs with yet another summation order: measured 1.3e-6 .. 3.1e-6.  The bound was max(3e-6, 3 x floor) until a
    # round-4 run measured 3.09e-6 against a floor of 0.73e-6; it is max(6e-6, 4 x floor) now -- a wrong exchange (a rank's
    # contribution missing or counted twice, a stale slice) is off by O(1), five orders of magnitude above either bound.
    floor = rep["single_process"]["run_to_run_first_grad_err"]
    # first-step gradient, rank-averaged, against the single process: per parameter tensor max |g - g_ref| <= 6e-6 max|g_ref|
    assert e["first_step_grad_err_vs_single"] <= max(6e-6, 4 * floor), (mode, e, floor)
    assert e["params_bit_identical_across_ranks"], (mode, e)
    assert e["grids_bit_identical_across_ranks_after_refresh_step0_step320"] == [True, True], (mode, e)
    assert e["samples_match_single_process"], (mode, e)


def test_ranks_compute_the_same_grid_without_communication(dp_rehearsal):
    """The estimator's (update_seed, step) stream + bit-identical parameters: every rank's own refresh already equals rank 0's
    before dist.sync_grid broadcasts it (recorded per refresh and mode)."""
    rep, _ = _report(dp_rehearsal)
    for mode in MODES:
        assert rep["modes"][mode]["grids_identical_before_the_broadcast"] == [True, True], (mode, rep["modes"][mode])


def test_rccl_calls_of_every_exchange_execute_on_hip_tensors(rccl_rehearsal):
    """backend "nccl" (= RCCL), one rank, collectives forced: the native branches of lsenerf_amd.dist run on the MI355X."""
    rep, r = _report(rccl_rehearsal)
    assert rep["world"] == 1 and rep["backend"].startswith("nccl"), rep["backend"]
    assert set(rep["modes"]) == set(MODES), r["log_tail"]
    floor = rep["single_process"]["run_to_run_first_grad_err"]
    for mode in MODES:
        e = rep["modes"][mode]
        assert e["first_step_grad_err_vs_single"] <= max(6e-6, 4 * floor), (mode, e, floor)
        assert e["within_tolerance"] and e["samples_match_single_process"], (mode, e)
        assert e["grids_bit_identical_across_ranks_after_refresh_step0_step320"] == [True, True], (mode, e)
    assert rep["all_ok"], r["log_tail"]
