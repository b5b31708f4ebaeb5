"""bench.py's Lifeline without a GPU: the line is printed (once) when SIGTERM arrives while the main thread sits in a C
call that no Python-level handler can interrupt, and when a block overruns its deadline."""
import json
import os
import signal
import subprocess
import sys
import textwrap
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

HEAD = textwrap.dedent("""
    import sys, os, time, ctypes
    sys.argv = ['bench.py']
    sys.path.insert(0, %r)
    import bench
    life = bench.Lifeline()
    life.result = {"value": 1.0, "tracer_batched": {"value": 2.0}}
""") % ROOT


def test_sigterm_prints_the_line_while_the_main_thread_sits_in_c():
    code = HEAD + textwrap.dedent("""
        life.arm("scatter_gather", 600.0)
        print("READY", flush=True)
        libc = ctypes.CDLL(None)
        while True:
            libc.sleep(30)
    """)
    p = subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, text=True)
    assert p.stdout.readline().strip() == "READY"
    time.sleep(0.3)
    p.send_signal(signal.SIGTERM)
    out = p.stdout.read()
    p.wait(timeout=20)
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert p.returncode == 143 and len(lines) == 1
    d = json.loads(lines[0])
    assert d["value"] == 1.0 and d["tracer_batched"]["value"] == 2.0 and "scatter_gather" in d["terminated"]


def test_watchdog_ends_a_block_that_overruns_and_prints_once():
    code = HEAD + textwrap.dedent("""
        life.arm("reference_layout_device_call", 1.0)
        libc = ctypes.CDLL(None)
        while True:
            libc.sleep(30)
    """)
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert res.returncode == 0 and len(lines) == 1 and "BENCH_WATCHDOG" in res.stderr
    d = json.loads(lines[0])
    assert d["value"] == 1.0 and "watchdog" in d["reference_layout_device_call"]["error"]


def test_the_line_is_printed_once():
    code = HEAD + textwrap.dedent("""
        life.print_line()
        life.print_line({"again": 1})
    """)
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    assert res.returncode == 0 and len([ln for ln in res.stdout.splitlines() if ln.startswith("{")]) == 1
