"""The submit queue behind dsv_enc / dsv_dec (csrc/batch.h: Coalescer) on the CPU: threads that each loop a synchronous call on an
instance of their own are merged into shared steps, never across keys, every caller gets its own result back, a single caller
waits for nobody, and a caller that has said good-bye (dsv_enc_free / end of stream -> forget) is not waited for.

tests/coalescer_harness.cpp includes the library's own header, compiled host-only (make -C oracle coalescer-test); its step
function only sleeps (2 ms), so no GPU is needed.  The GPU-side counterpart is tests/test_gpu_api_threads.py."""
import json
import os
import subprocess

import pytest

import dsvabi as A

pytestmark = pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="no hipcc")


@pytest.fixture(scope="module")
def report(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("coal") / "coalescer_test")
    r = subprocess.run(["make", "-C", os.path.join(A.ROOT, "oracle"), "coalescer-test", "COAL_OUT=" + out], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
    env = {k: v for k, v in os.environ.items() if not k.startswith("DSV2_COALESCE")}
    r = subprocess.run([out, "2000"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_every_caller_gets_its_own_result_and_keys_never_mix(report):
    for name, r in report.items():
        if name == "token_timeout":
            continue
        assert r["wrong"] == 0, name
        assert r["mixed"] == 0, name


def test_single_caller_is_the_plain_path(report):
    r = report["one_caller"]
    assert r["calls"] == 40 and r["steps"] == 40 and r["largest"] == 1
    assert r["waited_us"] < 40 * 50  # nobody to wait for: no window is spent


def test_callers_share_steps(report):
    r = report["four_callers"]
    assert r["calls"] == 160
    assert r["largest"] == 4 and r["mean_batch"] > 2.5  # (callers that start out of phase fall into step after a round)
    r = report["sixteen_callers"]
    assert r["calls"] == 480
    assert 4 <= r["largest"] <= 16 and r["mean_batch"] > 4.0  # a crowd of >= 8 is split into two steps side by side
    r = report["two_keys"]
    assert r["largest"] <= 3 and r["mean_batch"] > 1.8          # three callers a key: merged within the key only


def test_a_caller_that_left_is_not_waited_for(report):
    r = report["one_leaves"]
    assert r["calls"] == 65
    # 55 of the 60 calls of the caller that stayed ran after the other had gone: alone, and without spending the window on it
    assert r["steps"] >= 58
    assert r["waited_us"] < 55 * 400


def test_one_thread_driving_several_instances_does_not_wait_for_itself(report):
    """advisor finding (round 4): a single thread that loops over N same-geometry instances slept the window on every call,
    waiting for instances that only it could have submitted"""
    r = report["one_thread_four_instances"]
    assert r["calls"] == 80 and r["steps"] == 80 and r["largest"] == 1
    assert r["waited_us"] < 80 * 50


def test_an_instance_that_went_silent_ages_out_while_other_steps_run(report):
    """advisor finding (round 4): with steps of any key running back to back nothing ever aged out, and a caller that stopped
    calling without dsv_*_free / end of stream stayed expected: every leader then spent the full window on it"""
    r = report["silent_leaver"]
    assert r["wrong"] == 0
    own = r["calls"]  # (the other key's calls are counted too: the 40 calls of this key are what must not have waited)
    assert own >= 40
    assert r["waited_us"] < 40 * 100


def test_a_token_waiter_that_gave_up_does_not_stop_the_line(report):
    """advisor finding (round 5): SearchToken::acquire threw after its time-out with its FIFO ticket still in line; `serving`
    never passed it and every later search step of the process waited 120 s and failed.  csrc/batch.h: FifoToken retires it."""
    r = report["token_timeout"]
    assert r["steps"] == 2    # the two waiters that were meant to time out, and nobody else
    assert r["calls"] == 4    # everybody behind them got the token once the holder let go
    assert r["wrong"] == 0    # in the order they asked
    assert r["largest"] == 0  # no retired ticket left behind
    assert r["seconds"] < 2.0
