"""The hand-scheduled one-wave-per-SIMD conv kernels (conv_row_lw_kernel, conv_row_tall_kernel, conv_pw_lw_kernel; csrc/conv_lw.hip,
generated loop csrc/conv_lw_body.inc) against the ping-pong kernels they replace, BIT FOR BIT, on every layer shape of both networks
(models/encoders/wider_resnet.py:124-167, models/deeplabv3/deeplabv3.py:21-75,127-139 of the reference) at the bench's 8 images, with
every operand / output variant of the epilogue.  Both kernels run the same k order into the same fp32 chains, so ANY difference is a
defect -- this is the check that caught a missing barrier ("rare wrong tiles that came and went with the batch size") and a gfx950
store hazard (0.4 % of the elements) in round 4, which the network-level relative-L2 bars would have let through.

The switch KDCC_CONV_LW is read once per process, so each arm is one FRESH child process of tools/lw_check.py (never a re-exec of the
test runner); the children run one after the other."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import lw_check  # noqa: E402  (the case table only: importing it touches neither torch nor the GPU)

BATCH = 8


def _arm(lw, only="", **extra):
    env = dict(os.environ, KDCC_CONV_LW=lw, KDCC_CONV_LW_PW="1", KDCC_CONV_DUO="0")
    env.pop("KDCC_LIB", None)
    env.update(extra)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "lw_check.py"), "--child", "--batch", str(BATCH), "--iters", "1", "--only", only],
                       env=env, capture_output=True, text=True, timeout=900)
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    assert r.returncode == 0 and line, f"child KDCC_CONV_LW={lw} failed rc={r.returncode}\n{r.stdout[-2000:]}\n{r.stderr[-4000:]}"
    return json.loads(line[0][7:])


@pytest.fixture(scope="module")
def arms():
    return {"lw": _arm("1"), "pp": _arm("0")}


def _expected(name, H, W, Cout, d):
    if d == 0:
        return "conv_pw_lw_kernel", "conv_igemm_persist_kernel<pp>"
    if Cout == 128 and d > 16:
        return None, None          # outside both 512 x 128 kernels (dil <= 16): the arms run the same fallback
    if Cout == 128:
        return "conv_row_tall_kernel", "conv_row_pp128_kernel"
    return "conv_row_lw_kernel", "conv_row_persist_kernel<pp>"


@pytest.mark.parametrize("case", lw_check.CASES, ids=[c[0].replace(" ", "_") for c in lw_check.CASES])
def test_lone_wave_kernels_bit_identical_to_ping_pong(arms, case):
    name, H, W, Cin, Cout, d, opnds, outs = case
    lw, pp = arms["lw"][name], arms["pp"][name]
    want_lw, want_pp = _expected(name, H, W, Cout, d)
    if want_lw is None:
        assert lw["kernel"] == pp["kernel"], name
    else:
        assert lw["kernel"] == [want_lw], f"{name}: the lone-wave arm ran {lw['kernel']}"
        assert pp["kernel"] == [want_pp], f"{name}: the ping-pong arm ran {pp['kernel']}"
    assert lw["finite"] and pp["finite"], name
    assert len(lw["digest"]) == len(outs)
    assert lw["digest"] == pp["digest"], f"{name} {opnds} -> {outs}: {want_lw} and {want_pp} differ (sha256 of the outputs at {BATCH} images)"


def test_the_ab_finds_a_deliberately_missing_barrier(arms):
    """The self-test of the check above: the diagnostics build carries the same generated loop with ONE barrier removed and wave 0
    delayed in front of it (tools/gen_conv_lw.py BROKEN, KDCC_CONV_TUNE=32768 with KDCC_LIB=tuning; the shipped library compiles it
    out) -- the defect class that produced round 4's "rare wrong tiles".  Every conv_row_lw_kernel case of the table must then differ
    from the ping-pong arm: a bitwise A/B that stayed green on this schedule would be worth nothing."""
    sys.path.insert(0, ROOT)
    subprocess.check_call(["make", "-s", "-j", "8", "-C", os.path.join(ROOT, "knowledge-distillation-by-replacing-cheap-conv_amd", "csrc"), "TUNING=1"])
    names = [c[0] for c in lw_check.CASES if _expected(c[0], c[1], c[2], c[4], c[5])[0] == "conv_row_lw_kernel"][:6]
    assert len(names) >= 3
    broken = _arm("1", only=",".join(names), KDCC_LIB="tuning", KDCC_CONV_TUNE="32768")
    for name in names:
        assert broken[name]["kernel"] == ["conv_row_lw_kernel"], name
        assert broken[name]["digest"] != arms["pp"][name]["digest"], f"{name}: the A/B did not notice the missing barrier"
    sane = _arm("1", only=names[0], KDCC_LIB="tuning")        # the diagnostics build without the fault: identical again
    assert sane[names[0]]["digest"] == arms["pp"][names[0]]["digest"]
