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


def test_secondary_ceiling_terms():
    """The integer-issue ceiling in the bench line: its work terms are SURVEY.md 8(d)'s per-PRE figures."""
    b = load_bench()
    N, L, K, beta, alpha = 1 << 16, 12, 4, 3, 4
    ceil, t = b.valu_ceiling(N, 16, L, K, beta, alpha, 8, 11)
    assert t["butterflies_per_limb"] == 524288
    assert t["limb_transforms_int"] + t["limb_transforms_fp64"] == 80          # 12 INTT + 36 NTT + 2 (4 INTT + 12 NTT)
    assert t["limb_transforms_int"] == 1 + (3 * 4 + 2 * 1) + 8 + 2                # q_0 and the 60-bit P limbs
    assert t["base_conv_macs"] == (3 * 4 * 12 + 2 * 4 * 12) * N                  # 9.4 M + 6.3 M
    assert t["inner_product_muladds"] == 2 * 3 * 16 * N                          # 6.3 M
    assert 30e3 < ceil < 60e3                                                    # ct/s per GPU
    assert t["int_butterfly"] == "pseudo-mersenne"
    ceil_shoup, t2 = b.valu_ceiling(N, 16, L, K, beta, alpha, 8, 11, int_pm=False)
    assert t2["int_butterfly"] == "shoup" and ceil_shoup < ceil                  # dearer integer butterflies, lower ceiling


def test_cpu_model_string():
    b = load_bench()
    assert isinstance(b.cpu_model(), str) and b.cpu_model()
