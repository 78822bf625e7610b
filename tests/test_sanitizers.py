"""CPU sanitizer builds (the reference's Makefile:12 has none): the host-only parts of the product and
the oracle under -fsanitize=address,undefined.  GPU code cannot be sanitised on this pool, so the targets are
  * params.cpp (parameter generation, CRT tables)            ppqsflhe_amd/csrc  `make asan`
  * the hosts' parsing paths (JSON, base64, envelopes)       ppqsflhe_amd/host  `make asan`
  * the oracle's C restatement                               oracle             `make asan`
and this file drives them with well-formed and hostile inputs.  A sanitizer report aborts the process (non-zero exit)."""
import json
import os
import struct
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "ppqsflhe_amd", "host")
SELF = os.path.join(HOST, "build", "asan", "hostlib_selftest")
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:exitcode=99", UBSAN_OPTIONS="halt_on_error=1:exitcode=98")


def make(directory, target):
    r = subprocess.run(["make", "-C", directory, "-s", target], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.fixture(scope="module")
def selftest():
    make(HOST, "asan")
    assert os.access(SELF, os.X_OK)

    def run(*args):
        return subprocess.run([SELF, *map(str, args)], capture_output=True, text=True, env=ENV, timeout=300)
    return run


def clean(r):
    return r.returncode in (0, 1) and "Sanitizer" not in r.stderr and "runtime error" not in r.stderr


def test_params_under_asan_ubsan():
    make(os.path.join(ROOT, "ppqsflhe_amd", "csrc"), "asan")
    r = subprocess.run([os.path.join(ROOT, "ppqsflhe_amd", "params_selftest_asan")], capture_output=True, text=True,
                       env=ENV, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("ok log_n=") == 5 and "invalid parameters are refused" in r.stdout
    # the butterflies' pseudo-Mersenne products and folds (modarith.hpp is host + device code): residues and range bounds
    assert "ok pseudo-Mersenne arithmetic" in r.stdout


def test_host_parsers_on_wellformed_inputs(selftest, tmp_path, golden_dir):
    r = selftest("roundtrip", tmp_path / "rt.mkws")
    assert r.returncode == 0 and "ok roundtrip" in r.stdout, r.stderr
    r = selftest("envelope", tmp_path / "rt.mkws")
    assert r.returncode == 0 and "binary 3 ciphertexts" in r.stdout, r.stderr
    r = selftest("json", os.path.join(golden_dir, "cc_params.json"))
    assert r.returncode == 0 and "ok json" in r.stdout, r.stderr
    from tests.test_cli_hosts import openfhe_style_cc
    ref = json.load(open(os.path.join(golden_dir, "cc_params.json")))
    cc = tmp_path / "CC.json"
    cc.write_text(json.dumps(openfhe_style_cc(ref)))
    r = selftest("cc", cc)
    assert r.returncode == 0 and "log_n=14" in r.stdout and "moduli=4" in r.stdout, r.stderr


def test_host_parsers_on_hostile_inputs(selftest, tmp_path):
    """Every malformed input ends in a clean "ERROR" + exit 1: no sanitizer report, no crash, no giant allocation."""
    good = tmp_path / "good.mkws"
    assert selftest("roundtrip", good).returncode == 0
    raw = good.read_bytes()
    cases = {}
    cases["truncated.mkws"] = raw[: len(raw) // 2]
    cases["truncated_header.mkws"] = raw[:10]
    skel = b'{"weights_summary": []}'
    cases["blob_count.mkws"] = b"MKWS" + struct.pack("<IQ", 1, len(skel)) + skel + struct.pack("<Q", 1 << 40)
    cases["skeleton_size.mkws"] = b"MKWS" + struct.pack("<IQ", 1, 1 << 40) + skel
    cases["blob_size.mkws"] = b"MKWS" + struct.pack("<IQ", 1, len(skel)) + skel + struct.pack("<QQ", 1, 1 << 35)
    cases["version.mkws"] = b"MKWS" + struct.pack("<IQ", 9, len(skel)) + skel + struct.pack("<Q", 0)
    bad_ref = b'{"weights_summary": [{"layer": "l", "shape": [], "mean": "@7", "std_dev": "@0", "values": []}]}'
    cases["blob_index.mkws"] = b"MKWS" + struct.pack("<IQ", 1, len(bad_ref)) + bad_ref + struct.pack("<QQ", 1, 4) + b"MKCK"
    for name, data in cases.items():
        p = tmp_path / name
        p.write_bytes(data)
        r = selftest("envelope", p)
        assert r.returncode == 1 and "ERROR" in r.stderr and clean(r), (name, r.returncode, r.stderr[-500:])
    # ciphertext containers inside a JSON envelope: bad base64, short blob, wrong magic, size mismatch
    import base64
    hdr = struct.pack("<4sIIIIIIIdII", b"MKCK", 1, 1, 64, 3, 2, 1, 2, 1.0, 32, 0)
    blobs = {"short": base64.b64encode(hdr[:20]).decode(), "magic": base64.b64encode(b"XXXX" + hdr[4:]).decode(),
             "size": base64.b64encode(hdr + b"\x00" * 100).decode(), "b64": "!!!not base64!!!",
             "limbs": base64.b64encode(struct.pack("<4sIIIIIIIdII", b"MKCK", 1, 1, 64, 0xFFFFFFFF, 2, 1, 2, 1.0, 32, 0)).decode()}
    for name, blob in blobs.items():
        p = tmp_path / f"ct_{name}.json"
        p.write_text(json.dumps({"weights_summary": [{"layer": "l", "shape": [1], "mean": blob, "std_dev": blob, "values": []}]}))
        r = selftest("envelope", p)
        assert r.returncode == 1 and clean(r), (name, r.stderr[-500:])
    # JSON: unbounded nesting, unterminated string, bad escapes, huge numbers
    for name, text in {"deep.json": "[" * 200000, "deep_obj.json": '{"a":' * 100000, "str.json": '"abc', "esc.json": '"\\u12"',
                       "num.json": "1" * 400, "lit.json": "tru", "empty.json": "", "obj.json": '{"a" 1}'}.items():
        p = tmp_path / name
        p.write_text(text)
        r = selftest("json", p)
        assert clean(r), (name, r.returncode, r.stderr[-300:])
        if name not in ("num.json",):
            assert r.returncode == 1, name
    # CryptoContext files with missing fields
    for name, doc in {"cc_empty.json": {}, "cc_partial.json": {"mkckks_cc": {"log_n": 14}},
                      "cc_openfhe_partial.json": {"value0": {"ptr_wrapper": {"data": {"cc": {}}}}}}.items():
        p = tmp_path / name
        p.write_text(json.dumps(doc))
        r = selftest("cc", p)
        assert r.returncode == 1 and clean(r), (name, r.stderr[-300:])


def test_oracle_known_answers_under_asan_ubsan():
    """The oracle's C restatement, sanitizer build, loaded into a child Python with libasan preloaded: the golden-vector
    tests of tests/test_oracle_kat.py must pass without a report (leak detection off: the interpreter itself leaks)."""
    make(os.path.join(ROOT, "oracle"), "asan")
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    assert os.path.exists(libasan)
    env = dict(os.environ, LD_PRELOAD=libasan, ORACLE_LIB="liboracle_asan.so", OMP_NUM_THREADS="4",
               ASAN_OPTIONS="detect_leaks=0:exitcode=99", UBSAN_OPTIONS="halt_on_error=1:exitcode=98")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_oracle_kat.py"), "-x", "-q",
                        "-p", "no:cacheprovider"], capture_output=True, text=True, env=env, cwd=ROOT, timeout=1500)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    assert "passed" in r.stdout and "Sanitizer" not in r.stderr
