import json
import os
import socket
import subprocess
import sys
import tempfile
import threading
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _gpu_tier_selected(config) -> bool:
    expr = (config.getoption("markexpr", "") or "").replace(" ", "")
    return "gpu" in expr and "notgpu" not in expr


def _dp_test_selected(config) -> bool:
    """False for subset runs that cannot reach tests/test_gpu_dp.py (explicit other files, a -k that does not name it)."""
    kw = config.getoption("keyword", "") or ""
    if kw and "dp" not in kw:
        return False
    files = [a.split("::")[0] for a in config.args]
    for a in files:
        b = os.path.basename(os.path.normpath(a))
        if b == "test_gpu_dp.py" or os.path.isdir(a):
            return True
    return not files


def _launch_dp_rehearsal(config, backend="gloo"):
    """backend "gloo": two fresh child processes (torch.distributed.run, gloo, both on cuda:0) BEFORE this process makes any GPU call: ranks must
    never be spawned from a process that has initialised the GPU.  torch.cuda.device_count() does not initialise it.
    pytest's output capture is suspended meanwhile, so the rehearsal's progress lines reach the terminal (a silent minutes-long
    start looks like a hang to whoever runs the suite); the full log is kept for the assertion messages.
    backend "nccl": ONE fresh child process over RCCL with every collective forced (two RCCL ranks cannot share a card)."""
    import torch
    if torch.cuda.device_count() < 1:
        return {"launched": False, "reason": "no GPU visible"}
    tmp = tempfile.mkdtemp(prefix="lse_dp_")
    out, log = os.path.join(tmp, f"dp_rehearsal_{backend}.json"), os.path.join(tmp, f"dp_rehearsal_{backend}.log")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tools", "dp_rehearsal.py"), "--out", out]
    env = dict(os.environ)
    if backend == "nccl":
        cmd = [sys.executable, os.path.join(ROOT, "tools", "dp_rehearsal.py"), "--out", out, "--backend", "nccl"]
        env.update({"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    capman = config.pluginmanager.getplugin("capturemanager")
    if capman is not None:
        capman.suspend_global_capture(in_=True)
    t0 = time.time()
    rc = None
    try:
        print(f"\n[conftest] data-parallel rehearsal of the HIP model (tools/dp_rehearsal.py, {backend}) ...", flush=True)
        with open(log, "w") as lf:
            p = subprocess.Popen(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                                 start_new_session=True)
            # watchdog: the rehearsal takes 10 - 60 s; a launch that has not finished after 6 minutes (rendezvous stuck, a rank
            # lost) is killed as a group -- the report tests then fail, the rest of the GPU tier still runs
            def _kill():
                try:
                    os.killpg(p.pid, 9)
                except OSError:
                    pass
            dog = threading.Timer(360.0, _kill)
            dog.daemon = True
            dog.start()
            try:
                for line in p.stdout:              # tee: terminal + log file
                    lf.write(line)
                    if line.startswith("[dp_rehearsal]") or "Error" in line or "dp_rehearsal:" in line:
                        print("  " + line.rstrip(), flush=True)
                rc = p.wait(timeout=60)
            except subprocess.TimeoutExpired:
                _kill()
                rc = -9
            finally:
                dog.cancel()
    finally:
        if capman is not None:
            capman.resume_global_capture()
    with open(log) as lf:
        tail = lf.read()[-6000:]
    res = {"launched": True, "returncode": rc, "seconds": time.time() - t0, "log_tail": tail, "report": None, "path": out}
    if os.path.exists(out):
        with open(out) as f:
            res["report"] = json.load(f)
        keep = os.path.join(ROOT, "gpurun_out")
        if os.path.isdir(keep):          # on the GPU box: travels back with the call
            with open(os.path.join(keep, "dp_rehearsal_pytest.json" if backend == "gloo" else "dp_rehearsal_rccl_pytest.json"), "w") as f:
                json.dump(res["report"], f, indent=1)
    return res


def _test_file_selected(config, fname: str, kw_token: str) -> bool:
    kw = config.getoption("keyword", "") or ""
    if kw and kw_token not in kw:
        return False
    files = [a.split("::")[0] for a in config.args]
    return (not files) or any(os.path.basename(os.path.normpath(a)) == fname or os.path.isdir(a) for a in files)


def _launch_capture_probe(config):
    """tools/capture_after_eager_probe.py as a fresh child process, BEFORE this process touches the GPU: it captures a training step
    while earlier eager losses of the same model are still alive -- the situation that crashed the HIP runtime in
    hipStreamEndCapture in round 4.  A regression is a segfault, so it must not run inside the pytest process."""
    import torch
    if torch.cuda.device_count() < 1:
        return {"launched": False, "reason": "no GPU visible"}
    try:
        p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "capture_after_eager_probe.py")], cwd=ROOT, capture_output=True,
                           text=True, timeout=420)
        return {"launched": True, "returncode": p.returncode, "log_tail": (p.stdout + p.stderr)[-3000:]}
    except subprocess.TimeoutExpired as e:
        return {"launched": True, "returncode": -9, "log_tail": f"timeout: {e}"}


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    if _gpu_tier_selected(config) and _test_file_selected(config, "test_gpu_graph.py", "graph"):
        config._lse_capture_probe = _launch_capture_probe(config)
    else:
        config._lse_capture_probe = {"launched": False, "reason": "tests/test_gpu_graph.py not selected"}
    # the outcome of the 2-rank data-parallel rehearsal of the HIP model (tools/dp_rehearsal.py); tests/test_gpu_dp.py asserts on it
    if not _gpu_tier_selected(config):
        config._lse_dp_rehearsal = config._lse_rccl_rehearsal = {"launched": False, "reason": "GPU tier not selected"}
    elif not _dp_test_selected(config):
        config._lse_dp_rehearsal = config._lse_rccl_rehearsal = {
            "launched": False, "reason": "subset run that does not include tests/test_gpu_dp.py"}
    else:
        config._lse_dp_rehearsal = _launch_dp_rehearsal(config)
        config._lse_rccl_rehearsal = _launch_dp_rehearsal(config, backend="nccl")


@pytest.fixture(scope="session")
def dp_rehearsal(request):
    return request.config._lse_dp_rehearsal


@pytest.fixture(scope="session")
def rccl_rehearsal(request):
    return request.config._lse_rccl_rehearsal


@pytest.fixture(scope="session")
def capture_probe(request):
    return request.config._lse_capture_probe


def pytest_collection_modifyitems(config, items):
    import torch
    # the data-parallel report tests run last: under `-x` an environmental failure of the two-rank launch (ports, gloo) must not
    # stop the parity tests that come after test_gpu_dp.py in alphabetical order
    items.sort(key=lambda it: 1 if "test_gpu_dp" in it.nodeid else 0)
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
