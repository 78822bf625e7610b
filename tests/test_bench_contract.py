"""bench.py pieces that need no GPU: the algorithmic-bytes figure of SURVEY.md 8(d) and the loud failure without a device."""
import importlib.util
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_algorithmic_bytes_match_the_survey():
    b = load_bench()
    # C3+C4 at n = 8: 72 MiB (PRE) + 13.5 MiB (aggregation) + 23/8 MiB (rescale) = 88.375 MiB = 92 667 904 B
    assert b.algorithmic_bytes_per_unit(1 << 16, 12, 4, 3, 8) == 92667904
    mib = 1 << 20
    limb = 8 * (1 << 16)
    assert limb * (2 * 12 + 2 * 3 * 16 + 2 * 12) == 72 * mib          # PRE of one ciphertext
    assert limb * 2 * 12 * (1 + 1 / 8) == 13.5 * mib                   # aggregation share
    assert b.HBM_PEAK_GBS == 8000.0


def test_bench_refuses_to_run_without_a_gpu():
    env = dict(os.environ, HIP_VISIBLE_DEVICES="-1", ROCR_VISIBLE_DEVICES="-1", CUDA_VISIBLE_DEVICES="-1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--no-cpu"],
                       capture_output=True, text=True, env=env, timeout=300)
    if r.returncode == 0:  # a visible device despite the masks: then it must have produced the contract line
        assert '"metric"' in r.stdout
    else:
        assert "no HIP device" in (r.stderr + r.stdout) or "MI355X" in (r.stderr + r.stdout)
