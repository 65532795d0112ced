"""The rank launcher of ``bench.py --gpus N`` (lsenerf_amd/launch.py; what R:train.py:171-234 ``launch`` + :114-168
``_distributed_worker`` do for the reference): environment contract, rank 0's stdout relayed alone, exit-code propagation, a
dead rank takes the group down, and a rank-count mismatch is fatal instead of a silent one-rank measurement.  Stub children
only -- no GPU, no torch in the parent."""
import io
import json
import os
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launcher():
    import importlib.util
    spec = importlib.util.spec_from_file_location("lse_launch_under_test", os.path.join(ROOT, "lsenerf_amd", "launch.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_needs_launch_only_without_a_launcher():
    L = _launcher()
    assert L.needs_launch(8, {}) and L.needs_launch(2, {"RANK": "0"})
    assert not L.needs_launch(1, {}) and not L.needs_launch(8, {"WORLD_SIZE": "8"}) and not L.needs_launch(8, {"WORLD_SIZE": "1"})


def test_ranks_get_the_torchrun_environment_and_only_rank0_reaches_stdout():
    L = _launcher()
    child = ("import os, json, sys; e = {k: os.environ.get(k) for k in %r}; e['pid'] = os.getpid(); e['sid'] = os.getsid(0); "
             "print(json.dumps(e)); sys.stderr.write('note from ' + e['RANK'] + '\\n')" % (L.RANK_ENV + ("HSA_ENABLE_IPC_MODE_LEGACY",),))
    out, err = io.StringIO(), io.StringIO()
    rc = L.launch_ranks([sys.executable, "-c", child], 3, timeout=60, stdout=out, stderr=err, env={"PATH": os.environ["PATH"]})
    assert rc == 0
    lines = [json.loads(l) for l in out.getvalue().splitlines()]
    assert len(lines) == 1 and lines[0]["RANK"] == "0"                     # the result line of rank 0, nothing else
    others = [json.loads(l.split("] ", 1)[1]) for l in err.getvalue().splitlines() if l.startswith("[rank") and "{" in l]
    envs = lines + others
    assert sorted(e["RANK"] for e in envs) == ["0", "1", "2"]
    for e in envs:
        assert e["LOCAL_RANK"] == e["RANK"] and e["WORLD_SIZE"] == e["LOCAL_WORLD_SIZE"] == "3"
        assert e["MASTER_ADDR"] == "127.0.0.1" and e["MASTER_PORT"] == envs[0]["MASTER_PORT"] and int(e["MASTER_PORT"]) > 1024
        assert e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
        assert e["sid"] == e["pid"] != os.getpid()                         # every rank leads its own session / process group
    assert len({e["pid"] for e in envs}) == 3
    for r in (0, 1, 2):
        assert f"note from {r}" in err.getvalue()


def test_rank0_chatter_is_kept_out_of_the_result_stream():
    L = _launcher()
    child = "print('[Gloo] Rank 0 is connected to 1 peer ranks'); print('{\"metric\": 1}'); print('bye')"
    out, err = io.StringIO(), io.StringIO()
    assert L.launch_ranks([sys.executable, "-c", child], 1, timeout=60, stdout=out, stderr=err,
                          stdout_filter=lambda l: l.lstrip().startswith("{")) == 0
    assert out.getvalue() == '{"metric": 1}\n' and "[Gloo]" in err.getvalue() and "bye" in err.getvalue()


def test_a_dead_rank_terminates_the_others_and_its_code_is_the_exit_code(tmp_path):
    L = _launcher()
    child = ("import os, sys, time; r = int(os.environ['RANK']); open(os.path.join(%r, 'pid%%d' %% r), 'w').write(str(os.getpid())); "
             "time.sleep(0.5 if r == 1 else 120); sys.exit(7 if r == 1 else 0)" % str(tmp_path))
    err = io.StringIO()
    t0 = time.time()
    rc = L.launch_ranks([sys.executable, "-c", child], 3, timeout=100, stdout=io.StringIO(), stderr=err)
    assert rc == 7 and time.time() - t0 < 30
    assert "rank 1 exited with code 7" in err.getvalue()
    for r in (0, 2):                                                       # the sleeping ranks are gone (terminated by pid / group)
        pid = int((tmp_path / f"pid{r}").read_text())
        with pytest.raises(ProcessLookupError):
            os.kill(pid, 0)


def test_timeout_and_signal_codes():
    L = _launcher()
    rc = L.launch_ranks([sys.executable, "-c", "import time; time.sleep(60)"], 2, timeout=1.0, stdout=io.StringIO(), stderr=io.StringIO())
    assert rc == 124
    rc = L.launch_ranks([sys.executable, "-c", "import os, signal; os.kill(os.getpid(), signal.SIGKILL)"], 1, timeout=30,
                        stdout=io.StringIO(), stderr=io.StringIO())
    assert rc == 128 + 9


def test_bench_py_refuses_a_rank_count_that_differs_from_gpus():
    """`--gpus 8` inside a 1-rank environment used to print an `"n_gpus": 1` line; now it is an error before any measurement."""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert p.returncode != 0 and "WORLD_SIZE=1" in p.stderr and "--gpus 8" in p.stderr and p.stdout.strip() == ""


def test_bench_py_without_a_launcher_starts_its_own_ranks_and_propagates_their_failure():
    """No GPU here: both ranks get as far as the process group (gloo, world 2 = --gpus 2, so the mismatch check passes) and then
    fail on the missing GPU -- the parent must report that failure, not succeed and not hang."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env["LSE_BENCH_BACKEND"] = "gloo"
    p = subprocess.run([sys.executable, "-X", "importtime", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--launch-timeout", "240"], env=env, capture_output=True, text=True, timeout=400, cwd=ROOT)
    assert p.returncode not in (0, 124), p.stderr[-2000:]
    assert "[rank 0]" in p.stderr and "[rank 1]" in p.stderr and "needs an MI355X" in p.stderr
    assert "[launch] rank" in p.stderr and p.stdout.strip() == ""
    # the launching parent stays clear of torch and of the package that binds the HIP library: -X importtime lists every module
    # the PARENT imported (the ranks are fresh interpreters without the flag; their stderr arrives tagged "[rank r]")
    imported = [l.rsplit("|", 1)[1].strip() for l in p.stderr.splitlines() if l.startswith("import time:") and "|" in l]
    assert "subprocess" in imported and len(imported) > 20
    assert not [m for m in imported if m == "torch" or m.startswith("torch.") or m.startswith("lsenerf_amd")]


def test_sigterm_to_the_parent_takes_the_ranks_along(tmp_path):
    child = tmp_path / "child.py"
    child.write_text("import os, time\nopen(os.path.join(%r, 'pid' + os.environ['RANK']), 'w').write(str(os.getpid()))\ntime.sleep(120)\n"
                     % str(tmp_path))
    parent = tmp_path / "parent.py"
    parent.write_text("import sys, importlib.util\nspec = importlib.util.spec_from_file_location('L', %r)\n"
                      "L = importlib.util.module_from_spec(spec)\nspec.loader.exec_module(L)\n"
                      "sys.exit(L.launch_ranks([sys.executable, %r], 2, timeout=100))\n"
                      % (os.path.join(ROOT, "lsenerf_amd", "launch.py"), str(child)))
    p = subprocess.Popen([sys.executable, str(parent)], stderr=subprocess.PIPE, text=True)
    for _ in range(400):
        if (tmp_path / "pid0").exists() and (tmp_path / "pid1").exists():
            break
        time.sleep(0.05)
    time.sleep(0.2)
    pids = [int((tmp_path / f"pid{r}").read_text()) for r in (0, 1)]
    p.terminate()
    assert p.wait(timeout=30) == 130
    for pid in pids:
        with pytest.raises(ProcessLookupError):
            os.kill(pid, 0)
