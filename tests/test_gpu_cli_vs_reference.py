"""GPU box: the drivers in bin/ next to THE REFERENCE'S OWN PROGRAMS (oracle/_ref/bin: the reference sources built for gfx950 by
oracle/build_ref.sh) on the same command lines — SURVEY §8(b) row 1, the entry points: every option the reference's --help lists is
an option of the driver, the headless summaries have the reference's lines, bad command lines end the same way."""
import os
import re
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFBIN = os.path.join(ROOT, "oracle", "_ref", "bin")
BIN = os.path.join(ROOT, "bin")
PAIRS = [("tgs", "tgs"), ("tau_lbm", "tau_lbm"), ("tau_burgers", "tau_burgers"), ("tau_sw", "tau_sw")]


def run(path, *args, timeout=60):
    env = dict(os.environ, TERM="xterm")
    r = subprocess.run([path, *args], capture_output=True, text=True, timeout=timeout, env=env, stdin=subprocess.DEVNULL)
    return r.returncode, r.stdout, r.stderr


@pytest.fixture(scope="module")
def refbin():
    if not os.path.exists(os.path.join(REFBIN, "tgs")):
        pytest.skip("oracle/_ref/bin absent: oracle/build_ref.sh has not run (needs /root/reference)")
    return REFBIN


def options(help_text):
    return set(re.findall(r"^\s+(--[A-Za-z0-9_]+)", help_text, re.M))


@pytest.mark.parametrize("ref,ours", PAIRS)
def test_every_option_of_the_reference_help_is_an_option_of_the_driver(eng, refbin, ref, ours):
    rc_r, out_r, _ = run(os.path.join(refbin, ref), "--help")
    rc_o, out_o, _ = run(os.path.join(BIN, ours), "--help")
    assert rc_r == 0 and rc_o == 0
    want, got = options(out_r), options(out_o)
    assert len(want) >= 8 and want <= got, sorted(want - got)
    assert out_o.splitlines()[0] == out_r.splitlines()[0].replace(os.path.join(refbin, ref), os.path.join(BIN, ours))


def shape(text):
    """a summary with its numbers blanked: what has to be the same between two runs on different code"""
    return [re.sub(r"[-+]?\d+(\.\d+)?([eE][-+]?\d+)?", "#", l).strip() for l in text.splitlines()]


@pytest.mark.parametrize("ref,ours,args,nlines", [
    ("tau_lbm", "tau_lbm", ["--headless", "--steps", "200", "--nx", "256", "--ny", "128"], 1),
    ("tau_burgers", "tau_burgers", ["--headless", "--steps", "60", "--nx", "256", "--ny", "256"], 4),
    ("tau_burgers", "tau_burgers", ["--headless", "--steps", "60", "--nx", "256", "--ny", "256", "--muscl", "--stride", "7"], 4),
    ("tau_sw", "tau_sw", ["--headless", "--steps", "60", "--nx", "256", "--ny", "256"], 4)])
def test_headless_summaries_have_the_reference_lines(eng, refbin, ref, ours, args, nlines):
    rc_r, out_r, err_r = run(os.path.join(refbin, ref), *args)
    rc_o, out_o, err_o = run(os.path.join(BIN, ours), *args)
    assert rc_r == 0 and rc_o == 0, (err_r, err_o)
    sr, so = shape(out_r), shape(out_o)
    assert len(sr) >= nlines and so[:nlines] == sr[:nlines], (sr, so)
    # the counts the summary states (steps, frames, cells) are the same numbers
    ints = lambda t: re.findall(r"(\d+) (?:steps|frames|cells)|(?:Steps|steps): (\d+)", t)
    assert ints(out_r) == ints(out_o)


def test_unknown_options_and_bad_command_lines(eng, refbin):
    # the regression harness: usage on stderr, exit 2 (tau_hypersonic_cuda_tests.cu:50-82)
    for args in (["--steps"], ["--bogus"], ["--baseline"]):
        rc_r, _, err_r = run(os.path.join(refbin, "tau_hypersonic_cuda_tests"), *args)
        rc_o, _, err_o = run(os.path.join(BIN, "tau_hypersonic_cuda_tests"), *args)
        assert rc_r == rc_o == 2 and err_r.startswith("Usage:") and err_o.startswith("Usage:"), (args, rc_r, rc_o, err_r, err_o)
    # an unknown option: getopt complains on stderr and the Burgers / shallow-water programs go on (here: headless, a short run)
    for ref, ours in (("tau_burgers", "tau_burgers"), ("tau_sw", "tau_sw")):
        rc_r, out_r, err_r = run(os.path.join(refbin, ref), "--bogus", "--headless", "--steps", "5", "--nx", "64", "--ny", "64")
        rc_o, out_o, err_o = run(os.path.join(BIN, ours), "--bogus", "--headless", "--steps", "5", "--nx", "64", "--ny", "64")
        assert rc_r == rc_o == 0
        assert "unrecognized option '--bogus'" in err_r and "unrecognized option '--bogus'" in err_o
        assert shape(out_o)[:2] == shape(out_r)[:2]
