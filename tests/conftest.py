import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
if GOLDEN not in sys.path:
    sys.path.insert(0, GOLDEN)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "experimental: exercises libsegdino3d_hip_experimental.so (kernels outside the product library; "
                            "skipped when it is not built)")
    _start_spawn_server(config)


_SPAWN_SERVER = None


def _start_spawn_server(config):
    """tests/_spawn_server.py, started while this process is still GPU-free (configure time: no test module has been imported yet)."""
    global _SPAWN_SERVER
    expr = config.getoption("-m") or ""
    if "gpu" not in expr or "not gpu" in expr:
        return
    import subprocess
    _SPAWN_SERVER = subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_spawn_server.py")], stdin=subprocess.PIPE,
                                     stdout=subprocess.PIPE, cwd=ROOT)


def pytest_unconfigure(config):
    global _SPAWN_SERVER
    if _SPAWN_SERVER is not None:
        try:
            _SPAWN_SERVER.stdin.close()
            _SPAWN_SERVER.wait(timeout=10)
        except Exception:                                       # noqa: BLE001
            _SPAWN_SERVER.kill()
        _SPAWN_SERVER = None


@pytest.fixture(scope="session")
def fresh_process():
    """run(cmd, env, timeout) -> {"rc", "stdout", "stderr"} of a child started by the GPU-free spawn server (never a fork + exec of this
    process once it has initialised the GPU).  Without the server (a run that did not select `-m gpu`) the child is started directly, which
    is only allowed while this process has not initialised the GPU itself."""
    import json

    def run(cmd, env=None, timeout=600):
        if _SPAWN_SERVER is not None and _SPAWN_SERVER.poll() is None:
            _SPAWN_SERVER.stdin.write((json.dumps({"cmd": list(cmd), "env": env, "timeout": timeout, "cwd": ROOT}) + "\n").encode())
            _SPAWN_SERVER.stdin.flush()
            return json.loads(_SPAWN_SERVER.stdout.readline().decode())
        import subprocess
        import torch
        if torch.cuda.is_initialized():
            pytest.skip("no GPU-free spawn server and this process has initialised the GPU: refusing to fork + exec from it")
        r = subprocess.run(list(cmd), env=env, cwd=ROOT, timeout=timeout, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        return {"rc": r.returncode, "stdout": r.stdout.decode(errors="replace"), "stderr": r.stderr.decode(errors="replace")}
    return run


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
