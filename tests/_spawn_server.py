"""A GPU-free helper process that starts child processes on request.

`pytest -m gpu` runs every test in one process that has initialised the GPU long before a multi-process test is reached, and on the GPU
pool a process that has touched the GPU must never be replaced by another program (fork + exec of the pytest process is exactly that
for the forked copy).  `tests/conftest.py` therefore starts THIS program at configure time - before any test module is imported - and
tests that need a fresh process (RCCL ranks, `bench.py` as a rank) ask it to start one: the exec then happens in a process that has never
opened the GPU.  Protocol: one JSON object per line on stdin {"cmd": [...], "env": {...}, "timeout": seconds, "cwd": path} ->
one JSON object per line on stdout {"rc": int, "stdout": str, "stderr": str}.  It imports nothing but the standard library."""
import json
import subprocess
import sys


def main():
    for line in sys.stdin:
        line = line.strip()
        if not line:
            continue
        req = json.loads(line)
        try:
            r = subprocess.run(req["cmd"], env=req.get("env"), cwd=req.get("cwd"), timeout=req.get("timeout", 600),
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            out = {"rc": r.returncode, "stdout": r.stdout.decode(errors="replace"), "stderr": r.stderr.decode(errors="replace")}
        except subprocess.TimeoutExpired as e:
            out = {"rc": 124, "stdout": (e.stdout or b"").decode(errors="replace"), "stderr": (e.stderr or b"").decode(errors="replace") + "\n[timeout]"}
        except Exception as e:                                  # noqa: BLE001 - reported to the test
            out = {"rc": 125, "stdout": "", "stderr": repr(e)}
        sys.stdout.write(json.dumps(out) + "\n")
        sys.stdout.flush()


if __name__ == "__main__":
    main()
