"""bench.py's Watchdog (the thing that keeps the headline line when a multi-GPU extra hangs or a rank fails) without a
GPU: the thread, the flag file and the exit paths, in child processes (it ends its process with os._exit)."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import os, sys, time
sys.path.insert(0, sys.argv[1])
os.environ["MASTER_PORT"] = sys.argv[2]
import bench
rank, mode = int(sys.argv[3]), sys.argv[4]
dog = bench.Watchdog(rank, 0.6 if mode == "budget" else 60.0)
dog.headline = {"metric": "m", "value": 1.5} if rank == 0 else None
dog.partial["cfg3"] = {"ok": 1}
dog.start()
if mode == "flag":
    time.sleep(0.3)
    dog.raise_flag("cfg4 on rank 1: RuntimeError: boom")
    dog.raise_flag("a later reason must not replace the first")
if mode == "clean":
    dog.stop()
    print("finished")
    sys.exit(0)
time.sleep(20)
print("the watchdog did not end this process")
"""


def _run(tmp_path, port, rank, mode):
    script = tmp_path / "dog.py"
    script.write_text(SCRIPT)
    t0 = time.time()
    out = subprocess.run([sys.executable, str(script), ROOT, str(port), str(rank), mode], capture_output=True, text=True,
                         timeout=120)
    return out, time.time() - t0


def test_watchdog_prints_the_headline_when_a_rank_raises_the_flag(tmp_path):
    out, took = _run(tmp_path, 47011, 0, "flag")
    assert out.returncode == 0 and took < 15, (out.stdout, out.stderr)
    rec = json.loads(out.stdout.strip().splitlines()[-1])
    assert rec["value"] == 1.5 and rec["extras_multi_gpu"]["cfg3"] == {"ok": 1}
    assert rec["extras_multi_gpu"]["error"] == "cfg4 on rank 1: RuntimeError: boom"        # the FIRST reason
    # a rank other than 0 prints nothing and leaves with a non-zero status when it sees the flag of a peer
    flag = os.path.join("/tmp", "miso_bench_abort_47012")
    open(flag, "w").write("peer failed")
    try:
        out, took = _run(tmp_path, 47012, 1, "wait")
        assert out.returncode == 3 and out.stdout.strip() == "" and took < 15
    finally:
        if os.path.exists(flag):
            os.remove(flag)


def test_watchdog_budget_and_clean_stop(tmp_path):
    out, took = _run(tmp_path, 47013, 0, "budget")
    assert out.returncode == 0 and took < 15
    rec = json.loads(out.stdout.strip().splitlines()[-1])
    assert "budget" in rec["extras_multi_gpu"]["error"]
    out, _ = _run(tmp_path, 47014, 0, "clean")
    assert out.returncode == 0 and out.stdout.strip() == "finished"
    assert not os.path.exists(os.path.join("/tmp", "miso_bench_abort_47014"))
