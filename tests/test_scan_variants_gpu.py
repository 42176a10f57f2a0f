"""Every schedule of the scan kernels stays parity-green: the two-stage form of the 256-query tile
(scan.hip, SQE_SCAN=v0) next to the default ping-pong form (scan_pp.hip), and both ring depths of the
128-query tile.  The selectors exist only in the knobs build (libsqe_knobs.so, `make KNOBS=1`; the shipped
libsqe.so reads no environment variable), which a child process loads through SQE_LIB; each must return
the oracle's answer on a multi-chunk index.  GPU only."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KNOBS_LIB = os.path.join(ROOT, "semantic_query_engine_amd", "libsqe_knobs.so")


@pytest.fixture(scope="module")
def knobs_env():
    # always through make: a knobs library left over from an older tree must not be the one that is tested
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "semantic_query_engine_amd", "csrc"), "KNOBS=1", "-j8"],
                          stdout=subprocess.DEVNULL)
    return dict(os.environ, SQE_LIB=KNOBS_LIB)

CHILD = r"""
import json, sys
import numpy as np
sys.path.insert(0, %(root)r)
from oracle import retrieval as R
from semantic_query_engine_amd import Context, VectorIndex
from tests.gpu_util import assert_topk_matches, exact_topk_fast
rng = np.random.default_rng(4)
n, d, b, k = 300000, 256, %(batch)d, 10
x = rng.standard_normal((n, d)).astype(np.float32)
q = rng.standard_normal((b, d)).astype(np.float32)
q[:50] = x[rng.integers(0, n, 50)] + 0.1 * q[:50]
ctx = Context(0)
idx = VectorIndex(ctx, d)
idx.add(x)
cos, ids = idx.search(q, k)
ref_cos, ref_ids = exact_topk_fast(x, q, k, extra=64)
assert_topk_matches(cos, ids, ref_cos, ref_ids, R.normalize_rows(x), R.normalize_rows(q))
print(json.dumps({"ok": True, "uncertified": int(ctx.stats()["uncertified"])}))
"""


@pytest.mark.parametrize("which", ["v0", "pp"])
def test_alternative_scan_kernels(which, knobs_env):
    env = dict(knobs_env, SQE_SCAN=which)
    out = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT, "batch": 300}], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert json.loads(out.stdout.strip().splitlines()[-1])["ok"] is True


@pytest.mark.parametrize("ring", ["2", "3"])
def test_128_query_tile_ring_depths(ring, knobs_env):
    """Batches of 65-128 run on the 128-query tile: the two-stage ring (SQE_SCAN128=2) and the default
    (DB stages three deep, query stages two deep) both return the oracle's answer."""
    env = dict(knobs_env, SQE_SCAN128=ring)
    out = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT, "batch": 100}], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert json.loads(out.stdout.strip().splitlines()[-1])["ok"] is True


@pytest.mark.parametrize("bits,what", [("64", "k-row bound off: thresholds from the kp-row bound and the per-chunk lists only (the r01 filter)"),
                                       ("2048", "k-row bound from the minimum of the 16 group maxima instead of their k-th largest"),
                                       ("4096", "bound table fetched on the r02a schedule (32 / 128 / every second tile)"),
                                       ("128", "every wave issues its DMA pieces before its operand reads"),
                                       ("1024", "all four DMA pieces of a half-step from the memory phase")],
                         ids=["dbg64", "dbg2048", "dbg4096", "dbg128", "dbg1024"])
def test_timing_switches_do_not_change_answers(bits, what, knobs_env):
    """The SQE_DBG bits that select an older form of one mechanism (A/B timing in tools/ab_*.sh) must still return the
    oracle's answer: they change when work happens, never what is computed."""
    env = dict(knobs_env, SQE_DBG=bits)
    out = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT, "batch": 300}], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, what + "\n" + out.stderr[-2000:]
    assert json.loads(out.stdout.strip().splitlines()[-1])["ok"] is True


CHILD_I8 = CHILD.replace("idx = VectorIndex(ctx, d)\n", """idx = VectorIndex(ctx, d)
from semantic_query_engine_amd import SCAN_INT8_RESCORE
idx.set_option("scan_mode", SCAN_INT8_RESCORE); idx.set_option("i8_min_rows", 0)
idx.set_option("i8_sample_step", 8); idx.set_option("i8_sample_m", 64)
""").replace('print(json.dumps({"ok": True,', 'assert ctx.stats()["i8_collected"] > 0\nprint(json.dumps({"ok": True,')


@pytest.mark.parametrize("var,val,what", [("SQE_I8_DBG", "4", "compute parts at normal wave priority"),
                                          ("SQE_I8_DBG", "16", "appends of a finished tile before the barrier"),
                                          ("SQE_I8_DBG", "0", "the shipped schedule, through the knobs build")],
                         ids=["i8dbg4", "i8dbg16", "i8dbg0"])
def test_int8_schedule_variants_do_not_change_answers(var, val, what, knobs_env):
    """The run-time variants of the int8 scan's one-barrier schedule (scan_i8.hip, SQE_I8_DBG) move work inside a period: each
    returns the oracle's answer through the int8 path (batch 300: two 256-query blocks per chunk).  (Where a group issues its DMA
    pieces is a compile-time variant, -DSQE_I8_VARIANT: tools/r04_ab_i8.sh; every form keeps the invariant that a piece is
    retired by its issuing wave in front of a barrier that precedes its read, and tests/test_i8_exact_gpu.py checks the shipped
    one bit for bit.)"""
    assert "i8_sample_step" in CHILD_I8 and "i8_collected" in CHILD_I8
    env = dict(knobs_env, **{var: val})
    out = subprocess.run([sys.executable, "-c", CHILD_I8 % {"root": ROOT, "batch": 300}], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, what + "\n" + out.stderr[-2000:]
    assert json.loads(out.stdout.strip().splitlines()[-1])["ok"] is True
