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
os.environ["MISO_BENCH_TOKEN"] = sys.argv[5]
import bench
rank, mode = int(sys.argv[3]), sys.argv[4]
dog = bench.Watchdog(rank, 0.6 if mode == "budget" else 60.0)
dog.headline = {"metric": "m", "value": 1.5} if rank == 0 else None
dog.partial["cfg3"] = {"ok": 1}
dog.start()
ready = dog.flag + ".ready"              # (test plumbing: what the barrier after the construction is in bench.py)
if sys.argv[5] == "pair" and rank == 0:
    open(ready, "w").close()
if sys.argv[5] == "pair" and rank == 1:
    while not os.path.exists(ready):
        time.sleep(0.05)
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


def _cmd(tmp_path, port, rank, mode, token):
    script = tmp_path / "dog.py"
    script.write_text(SCRIPT)
    return [sys.executable, str(script), ROOT, str(port), str(rank), mode, token]


def _run(tmp_path, port, rank, mode, token="tok"):
    t0 = time.time()
    out = subprocess.run(_cmd(tmp_path, port, rank, mode, token), capture_output=True, text=True, timeout=120)
    return out, time.time() - t0


def _flag(port, token="tok"):
    return os.path.join("/tmp", f"miso_bench_abort_{port}_{token}")


def test_watchdog_prints_the_headline_when_a_rank_raises_the_flag(tmp_path):
    out, took = _run(tmp_path, 47011, 0, "flag")
    assert out.returncode == 0 and took < 15, (out.stdout, out.stderr)
    rec = json.loads(out.stdout.strip().splitlines()[-1])
    assert rec["value"] == 1.5 and rec["extras_multi_gpu"]["cfg3"] == {"ok": 1}
    assert rec["extras_multi_gpu"]["error"] == "cfg4 on rank 1: RuntimeError: boom"        # the FIRST reason
    assert not os.path.exists(_flag(47011)) and os.path.exists(_flag(47011) + ".done")      # flag gone, rank 0's "printed" mark left
    os.remove(_flag(47011) + ".done")
    # a rank other than 0 prints nothing; with no rank 0 around to print (no .done file within the grace period) it
    # leaves with a non-zero status
    flag = _flag(47012)
    open(flag, "w").write("peer failed")
    try:
        out, took = _run(tmp_path, 47012, 1, "wait")
        assert out.returncode == 3 and out.stdout.strip() == "" and 4.0 < took < 20
    finally:
        if os.path.exists(flag):
            os.remove(flag)


def test_peers_outlive_rank_zeros_print_and_leave_with_status_zero(tmp_path):
    """a launcher ends every rank at the first non-zero status, so a peer must not leave (non-zero) before rank 0
    has printed.  Rank 1 raises the flag; rank 0 prints the headline and marks it; rank 1 leaves AFTER that, status 0."""
    p0 = subprocess.Popen(_cmd(tmp_path, 47015, 0, "wait", "pair"), stdout=subprocess.PIPE, text=True)
    p1 = subprocess.Popen(_cmd(tmp_path, 47015, 1, "flag", "pair"), stdout=subprocess.PIPE, text=True)
    out0, _ = p0.communicate(timeout=120)
    out1, _ = p1.communicate(timeout=120)
    assert p0.returncode == 0 and p1.returncode == 0, (p0.returncode, p1.returncode)      # 0: rank 1 saw rank 0's mark
    rec = json.loads(out0.strip().splitlines()[-1])
    assert rec["extras_multi_gpu"]["error"].startswith("cfg4 on rank 1") and out1.strip() == ""
    for f in (_flag(47015, "pair"), _flag(47015, "pair") + ".done", _flag(47015, "pair") + ".ready"):
        if os.path.exists(f):
            os.remove(f)


def test_a_stale_flag_of_another_run_is_not_seen(tmp_path):
    """The flag's name carries a per-run token: the leftover of a failed run on the same port does not abort the next."""
    stale = _flag(47016, "old")
    open(stale, "w").write("left behind")
    try:
        out, _ = _run(tmp_path, 47016, 1, "clean", token="new")
        assert out.returncode == 0 and out.stdout.strip() == "finished"
    finally:
        os.remove(stale)


def test_watchdog_budget_and_clean_stop(tmp_path):
    out, took = _run(tmp_path, 47013, 0, "budget")
    assert out.returncode == 0 and took < 15
    rec = json.loads(out.stdout.strip().splitlines()[-1])
    assert "budget" in rec["extras_multi_gpu"]["error"]
    out, _ = _run(tmp_path, 47014, 0, "clean")
    assert out.returncode == 0 and out.stdout.strip() == "finished"
    assert not os.path.exists(_flag(47014))
