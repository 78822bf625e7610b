"""The C++ CLI hosts (ppqsflhe_amd/host): same argv / exit-code / JSON-envelope contract as the reference's mains
(SURVEY.md 8b).  CPU part: usage errors, genCC parameter KAT.  GPU part: a full FL round through the binaries."""
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "ppqsflhe_amd", "host", "build")
PROGS = ["genCC", "keyGen", "REkeyGen", "encryptModelWeights", "changeCipherDomain", "aggregateEncryptedWeights",
         "decryptModelWeights", "serverRound"]


def run(prog, *args, env=None):
    e = dict(os.environ)
    e.update(env or {})
    return subprocess.run([os.path.join(BIN, prog), *map(str, args)], capture_output=True, text=True, env=e)


def test_binaries_exist():
    for p in PROGS:
        assert os.access(os.path.join(BIN, p), os.X_OK), f"{p} not built (make -C ppqsflhe_amd/host)"


@pytest.mark.parametrize("prog,usage", [
    ("changeCipherDomain", "<cc_path> <rekey_path> <input_encfile> <output_encfile>"),        # changeCipherDomain.cpp:19-24
    ("aggregateEncryptedWeights", "<cc_path> <client2_encfile> <client1to2_encfile> <output_aggfile>"),
    ("encryptModelWeights", "<cc_path> <pubkey_path> <input_weights> <output_encfile>"),
    ("decryptModelWeights", "<cc_path> <privkey_path> <input_encfile> <output_file>"),
    ("keyGen", "<cc_path> <pubkey_out> <privkey_out>"),
    ("REkeyGen", "<cc.json> <client_privkey.json> <peer_pubkey.json> <rekey_out.json>"),
    ("serverRound", "<cc_path> <output_aggfile> <rekey_1|-> <encfile_1>"),
])
def test_usage_errors_exit_1(prog, usage):
    r = run(prog)
    assert r.returncode == 1
    assert "Usage:" in r.stderr and usage in r.stderr


def test_unreadable_cc_exits_1(tmp_path):
    r = run("changeCipherDomain", tmp_path / "nope.json", "a", "b", "c")
    assert r.returncode == 1 and "[recrypt] ERROR: Failed to load CryptoContext" in r.stderr
    r = run("aggregateEncryptedWeights", tmp_path / "nope.json", "a", "b", "c")
    assert r.returncode == 1 and "[agg] ERROR: Failed to load CryptoContext" in r.stderr
    r = run("decryptModelWeights", tmp_path / "nope.json", "a", "b", "c")
    assert r.returncode == 1 and "[decrypt] ERROR: Failed to load CryptoContext" in r.stderr


def test_gencc_reproduces_reference_context(tmp_path, golden_dir):
    # server/config/config_cc.json:2-5 -> the parameters of server/storage/CC.json (N, 4 moduli, roots, dnum)
    cfg = tmp_path / "config_cc.json"
    cfg.write_text(json.dumps({"MultiplicativeDepth": 2, "ScalingModSize": 40, "BatchSize": 8192, "PREMode": "INDCPA"}))
    out = tmp_path / "CC.json"
    r = run("genCC", cfg, out)
    assert r.returncode == 0, r.stderr
    assert "CryptoContext Generated and saved to" in r.stdout
    cc = json.load(open(out))["mkckks_cc"]
    ref = json.load(open(os.path.join(golden_dir, "cc_params.json")))
    assert cc["ring_dim"] == ref["ring_dim"]
    assert cc["moduli"] == ref["moduli"]
    assert cc["roots"] == ref["roots"]
    assert cc["NumLargeDigits"] == ref["dnum"] and cc["AuxBits"] == ref["aux_bits"] and cc["ExtraBits"] == ref["extra_bits"]
    assert cc["BatchSize"] == ref["batch_size"]
    # genCC.cpp:62-65: unknown PREMode -> exit 1
    cfg.write_text(json.dumps({"MultiplicativeDepth": 2, "ScalingModSize": 40, "BatchSize": 8192, "PREMode": "BOGUS"}))
    r = run("genCC", cfg, out)
    assert r.returncode == 1 and "Unknown PREMode" in r.stderr
    r = run("genCC", tmp_path / "missing.json", out)
    assert r.returncode == 1 and "Failed to open" in r.stderr


def openfhe_style_cc(ref):
    """A CC.json with the nesting OpenFHE's cereal writer produces, filled from the reference fixture values."""
    limbs = [{"ptr_wrapper": {"data": {"value0": {"co": 2 * ref["ring_dim"], "rd": ref["ring_dim"], "cm": {"v": m},
                                                   "ru": {"v": r}}}}} for m, r in zip(ref["moduli"], ref["roots"])]
    base = {"elp": {"ptr_wrapper": {"data": {"value0": {"co": 2 * ref["ring_dim"], "rd": ref["ring_dim"]}, "p": limbs}}},
            "enp": {"ptr_wrapper": {"data": {"m": ref["scaling_bits"], "bs": ref["batch_size"]}}}}
    rlwe = {"value0": base, "dp": ref["sigma"], "md": ref["mult_depth"], "mo": ref["pre_mode"]}
    rns = {"value0": rlwe, "ks": 2, "rs": 3, "dnum": ref["dnum"], "ab": ref["aux_bits"], "eb": ref["extra_bits"]}
    return {"value0": {"ptr_wrapper": {"data": {"cc": {"ptr_wrapper": {"data": {"value0": rns}}}}}}}


def test_openfhe_cc_json_is_accepted_without_gpu_until_compute(tmp_path, golden_dir):
    ref = json.load(open(os.path.join(golden_dir, "cc_params.json")))
    cc = tmp_path / "CC.json"
    cc.write_text(json.dumps(openfhe_style_cc(ref)))
    # parsing succeeds; with no visible device the program must fail loudly (exit 1), never fall back to a CPU path
    r = run("keyGen", cc, tmp_path / "pk", tmp_path / "sk", env={"HIP_VISIBLE_DEVICES": "-1", "ROCR_VISIBLE_DEVICES": "-1"})
    if r.returncode != 0:
        assert "[keyGen] ERROR" in r.stderr
    else:  # a GPU is present: keys must exist
        assert os.path.getsize(tmp_path / "pk") > 0


@pytest.mark.gpu
def test_full_round_through_the_binaries(tmp_path, golden_dir):
    """orchestration/run.sh:28-44 with the real programs: genCC, keyGen x2, REkeyGen x2, encrypt x2,
    changeCipherDomain c1->c2, aggregate, changeCipherDomain c2->c1, decrypt x2; output = mean within 2^-25 (P8)."""
    W = np.load(os.path.join(golden_dir, "e2e_weights.npz"))
    meta = json.load(open(os.path.join(golden_dir, "e2e_weights_meta.json")))["layers"]
    keep = [m for m in meta if m["layer"] in ("param_2", "param_6", "param_7", "param_1")]

    def weights_file(c):
        layers = []
        for m in keep:
            vals = W[f"sample_c{c}_{m['layer']}_values"]
            ms = W[f"sample_c{c}_{m['layer']}_mean_std"]
            shape = m["shape"] if m["layer"] != "param_1" else [vals.size]
            layers.append({"layer": m["layer"], "shape": shape, "mean": float(ms[0]), "std_dev": float(ms[1]),
                           "values": [float(v) for v in vals]})
        layers.append({"layer": "optimizer/iteration", "shape": [1], "mean": 1.0, "std_dev": 0.0, "values": [3.0]})
        layers.append({"layer": "empty_tensor", "shape": [0], "mean": 0.0, "std_dev": 0.0, "values": []})  # edge: no values
        p = tmp_path / f"sample_weights_c{c}.json"
        p.write_text(json.dumps({"weights_summary": layers}))
        return p

    ref = json.load(open(os.path.join(golden_dir, "cc_params.json")))
    cc = tmp_path / "CC.json"
    cc.write_text(json.dumps(openfhe_style_cc(ref)))  # consume an OpenFHE-written context file

    def ok(r):
        assert r.returncode == 0, r.stdout + r.stderr
        return r

    for c in (1, 2):
        ok(run("keyGen", cc, tmp_path / f"pk{c}", tmp_path / f"sk{c}"))
    ok(run("REkeyGen", cc, tmp_path / "sk1", tmp_path / "pk2", tmp_path / "rk1"))   # c1 -> c2
    ok(run("REkeyGen", cc, tmp_path / "sk2", tmp_path / "pk1", tmp_path / "rk2"))   # c2 -> c1
    for c in (1, 2):
        r = ok(run("encryptModelWeights", cc, tmp_path / f"pk{c}", weights_file(c), tmp_path / f"enc{c}.json"))
        assert "Skipping optimizer layer: optimizer/iteration" in r.stdout
        assert "Batch size from CryptoContext = 8192" in r.stdout
    enc1 = json.load(open(tmp_path / "enc1.json"))
    assert [l["layer"] for l in enc1["weights_summary"]] == [m["layer"] for m in keep] + ["empty_tensor"]
    assert enc1["weights_summary"][-1]["values"] == []
    assert all(isinstance(l["mean"], str) and isinstance(l["values"], list) for l in enc1["weights_summary"])
    r = ok(run("changeCipherDomain", cc, tmp_path / "rk1", tmp_path / "enc1.json", tmp_path / "c1_as_c2.json"))
    assert "[recrypt] Re-encryption completed successfully" in r.stdout
    r = ok(run("aggregateEncryptedWeights", cc, tmp_path / "enc2.json", tmp_path / "c1_as_c2.json", tmp_path / "agg.json"))
    assert "[agg] Aggregation completed successfully" in r.stdout
    # P6: the aggregate has one limb less (4 -> 3): blob size shrinks by 1/4 of the payload
    agg = json.load(open(tmp_path / "agg.json"))
    assert len(agg["weights_summary"][0]["mean"]) < len(enc1["weights_summary"][0]["mean"]) * 0.8
    ok(run("changeCipherDomain", cc, tmp_path / "rk2", tmp_path / "agg.json", tmp_path / "agg_as_c1.json"))
    ok(run("decryptModelWeights", cc, tmp_path / "sk2", tmp_path / "agg.json", tmp_path / "dec2.json"))
    ok(run("decryptModelWeights", cc, tmp_path / "sk1", tmp_path / "agg_as_c1.json", tmp_path / "dec1.json"))
    tol = 2.0 ** -25
    for c in (1, 2):
        dec = json.load(open(tmp_path / f"dec{c}.json"))["weights_summary"]
        assert [l["layer"] for l in dec] == [m["layer"] for m in keep] + ["empty_tensor"]
        assert dec[-1]["values"] == [] and abs(dec[-1]["mean"]) < tol
        for l in dec[:-1]:
            name = l["layer"]
            v1, v2 = W[f"sample_c1_{name}_values"], W[f"sample_c2_{name}_values"]
            got = np.array(l["values"])
            assert got.size == v1.size  # padding trimmed to prod(shape)
            assert np.abs(got - (v1 + v2) / 2).max() < tol
            ms = (W[f"sample_c1_{name}_mean_std"] + W[f"sample_c2_{name}_mean_std"]) / 2
            assert abs(l["mean"] - ms[0]) < tol and abs(l["std_dev"] - ms[1]) < tol
    # f1 binary envelope: the same ciphertexts carried as raw containers behind the JSON skeleton ("MKWS");
    # the server programs keep the input's form, the decrypted document is identical to the JSON route's
    def reenvelope(src, dst):
        import base64, struct
        doc = json.load(open(src))
        blobs = []

        def take(b64):
            blobs.append(base64.b64decode(b64))
            return f"@{len(blobs) - 1}"
        for l in doc["weights_summary"]:
            l["mean"], l["std_dev"] = take(l["mean"]), take(l["std_dev"])
            l["values"] = [take(v) for v in l["values"]]
        text = json.dumps(doc).encode()
        with open(dst, "wb") as f:
            f.write(b"MKWS" + struct.pack("<IQ", 1, len(text)) + text + struct.pack("<Q", len(blobs)))
            for b in blobs:
                f.write(struct.pack("<Q", len(b)) + b)
    reenvelope(tmp_path / "enc1.json", tmp_path / "enc1.mkws")
    reenvelope(tmp_path / "enc2.json", tmp_path / "enc2.mkws")
    ok(run("changeCipherDomain", cc, tmp_path / "rk1", tmp_path / "enc1.mkws", tmp_path / "c1_as_c2.mkws"))
    assert open(tmp_path / "c1_as_c2.mkws", "rb").read(4) == b"MKWS"
    assert os.path.getsize(tmp_path / "c1_as_c2.mkws") < 0.8 * os.path.getsize(tmp_path / "c1_as_c2.json")
    ok(run("aggregateEncryptedWeights", cc, tmp_path / "enc2.mkws", tmp_path / "c1_as_c2.mkws", tmp_path / "agg.mkws"))
    ok(run("decryptModelWeights", cc, tmp_path / "sk2", tmp_path / "agg.mkws", tmp_path / "dec2_bin.json"))
    assert json.load(open(tmp_path / "dec2_bin.json")) == json.load(open(tmp_path / "dec2.json"))  # PRE is deterministic
    # encryptModelWeights writes the binary form on request; it decrypts like the JSON one
    ok(run("encryptModelWeights", cc, tmp_path / "pk2", weights_file(2), tmp_path / "enc2b.mkws"))
    assert open(tmp_path / "enc2b.mkws", "rb").read(4) == b"MKWS"
    ok(run("decryptModelWeights", cc, tmp_path / "sk2", tmp_path / "enc2b.mkws", tmp_path / "dec2b.json"))
    d2b = json.load(open(tmp_path / "dec2b.json"))["weights_summary"][0]
    assert np.abs(np.array(d2b["values"]) - W[f"sample_c2_{d2b['layer']}_values"]).max() < tol
    # truncated binary envelope -> exit 1
    raw = open(tmp_path / "agg.mkws", "rb").read()
    open(tmp_path / "trunc.mkws", "wb").write(raw[:len(raw) // 2])
    r = run("decryptModelWeights", cc, tmp_path / "sk2", tmp_path / "trunc.mkws", tmp_path / "x.json")
    assert r.returncode == 1 and "[decrypt] ERROR" in r.stderr

    # wrong key -> garbage, not the mean (sanity that decryption really depends on the domain change)
    ok(run("decryptModelWeights", cc, tmp_path / "sk1", tmp_path / "agg.json", tmp_path / "bad.json"))
    bad = json.load(open(tmp_path / "bad.json"))["weights_summary"][0]
    name = bad["layer"]
    v = (W[f"sample_c1_{name}_values"] + W[f"sample_c2_{name}_values"]) / 2
    assert np.abs(np.array(bad["values"]) - v).max() > 1.0


def openfhe_style_private_key(limbs, moduli):
    """The nesting of a private key written by Serial::SerializeToFile(.., SerType::JSON) (keyGen.cpp:45;
    client_1-private.key): only the fields an importer needs, filled from the NTT-KAT fixture residues."""
    v = [{"cereal_class_version": 1,
          "v": {"polymorphic_id": 1073741824,
                "ptr_wrapper": {"valid": 1, "data": {"cereal_class_version": 1, "v": [int(x) for x in limb],
                                                     "m": {"v": int(m)}}}},
          "f": 0} for limb, m in zip(limbs, moduli)]
    return {"value0": {"polymorphic_id": 1073741824,
                       "ptr_wrapper": {"id": 2147483649, "data": {"cereal_class_version": 0,
                                                                  "s": {"cereal_class_version": 1, "v": v, "f": 0}}}}}


def write_key_container(path, kind, ring_dim, limbs, parts, data):
    """ppqsflhe_amd/host/hostlib.hpp BlobHeader (48 bytes) + raw residues."""
    import struct
    hdr = struct.pack("<4sIIIIIIIdII", b"MKCK", 1, kind, ring_dim, limbs, parts, 0, 0, 0.0, 0, 0)
    assert len(hdr) == 48
    with open(path, "wb") as f:
        f.write(hdr)
        f.write(np.ascontiguousarray(data, dtype=np.uint64).tobytes())


@pytest.mark.gpu
def test_private_key_written_by_openfhe_is_imported(tmp_path, golden_dir):
    """REkeyGen / decryptModelWeights take the reference's own private-key file
    (client/storage/client_1/private/client_1-private.key, committed as the NTT-KAT fixture): the ternary secret is
    recovered from the EVALUATION limbs, the key is rebuilt over QP, and it decrypts what the matching public key
    encrypted -- directly and after a domain change to a second client."""
    import torch
    from oracle.oracle import OracleContext
    from ppqsflhe_amd import Context

    ref = json.load(open(os.path.join(golden_dir, "cc_params.json")))
    kat = np.load(os.path.join(golden_dir, "sk_ntt_kat.npz"))
    sk_eval, sk_mod = kat["sk1_eval"], kat["sk1_moduli"]
    assert [int(m) for m in sk_mod] == ref["moduli"]
    cc = tmp_path / "CC.json"
    cc.write_text(json.dumps(openfhe_style_cc(ref)))
    skfile = tmp_path / "client_1-private.key"
    skfile.write_text(json.dumps(openfhe_style_private_key(sk_eval, sk_mod)))

    # the matching public key: ternary secret from the fixture (CPU oracle INTT), pk = (-a s + e, a) on the engine
    log_n = int(np.log2(ref["ring_dim"]))
    o = OracleContext(log_n, ref["mult_depth"], ref["scaling_bits"], 60, dnum=ref["dnum"])
    coef = o.ntt_inv(0, sk_eval[0])
    q0 = int(sk_mod[0])
    s = np.where(coef == 0, 0, np.where(coef == 1, 1, -1)).astype(np.int8)
    assert np.all((coef == 0) | (coef == 1) | (coef == q0 - 1))
    ctx = Context(log_n, ref["mult_depth"], ref["scaling_bits"], 60, dnum=ref["dnum"], device=0)
    N, D = ctx.N, ctx.D
    dev = torch.device("cuda", 0)
    a = torch.empty(D, N, dtype=torch.int64, device=dev)
    e = torch.empty(N, dtype=torch.int32, device=dev)
    ctx.sample_uniform(a, 1, ctx.L, True, 77, 0)
    ctx.sample_gauss(e, N, 3.19, 77, 1)
    pk = torch.empty(2, D, N, dtype=torch.int64, device=dev)
    sk_dev = torch.empty(D, N, dtype=torch.int64, device=dev)
    ctx.keygen(torch.from_numpy(s).to(dev), a, e, pk, sk_dev)
    ctx.sync()
    # the engine's transform of the recovered secret reproduces the file's limbs bit for bit
    assert np.array_equal(sk_dev[:ctx.L].cpu().numpy().view(np.uint64), sk_eval)
    write_key_container(tmp_path / "pk1", 2, N, D, 2, pk.cpu().numpy().view(np.uint64))
    ctx.close()

    W = np.load(os.path.join(golden_dir, "e2e_weights.npz"))
    vals = W["sample_c1_param_2_values"]
    wfile = tmp_path / "w.json"
    wfile.write_text(json.dumps({"weights_summary": [{"layer": "param_2", "shape": [int(vals.size)],
                                                      "mean": float(vals.mean()), "std_dev": float(vals.std()),
                                                      "values": [float(v) for v in vals]}]}))

    def ok(r):
        assert r.returncode == 0, r.stdout + r.stderr
        return r

    ok(run("encryptModelWeights", cc, tmp_path / "pk1", wfile, tmp_path / "enc1.json"))
    r = ok(run("decryptModelWeights", cc, skfile, tmp_path / "enc1.json", tmp_path / "dec1.json"))
    assert "[decrypt] Private key loaded" in r.stdout
    got = np.array(json.load(open(tmp_path / "dec1.json"))["weights_summary"][0]["values"])
    assert np.abs(got - vals).max() < 2.0 ** -25
    # domain change c1 -> c2 with a re-key generated from the imported key (needs s over QP)
    ok(run("keyGen", cc, tmp_path / "pk2", tmp_path / "sk2"))
    ok(run("REkeyGen", cc, skfile, tmp_path / "pk2", tmp_path / "rk1"))
    ok(run("changeCipherDomain", cc, tmp_path / "rk1", tmp_path / "enc1.json", tmp_path / "c1_as_c2.json"))
    ok(run("decryptModelWeights", cc, tmp_path / "sk2", tmp_path / "c1_as_c2.json", tmp_path / "dec2.json"))
    got = np.array(json.load(open(tmp_path / "dec2.json"))["weights_summary"][0]["values"])
    assert np.abs(got - vals).max() < 2.0 ** -25
    # a tampered file (one residue changed) is rejected, exit 1
    bad = openfhe_style_private_key(sk_eval, sk_mod)
    bad["value0"]["ptr_wrapper"]["data"]["s"]["v"][1]["v"]["ptr_wrapper"]["data"]["v"][5] ^= 1
    badfile = tmp_path / "bad.key"
    badfile.write_text(json.dumps(bad))
    r = run("decryptModelWeights", cc, badfile, tmp_path / "enc1.json", tmp_path / "x.json")
    assert r.returncode == 1 and "[decrypt] ERROR" in r.stderr


def test_garbage_private_key_exits_1_without_gpu(tmp_path, golden_dir):
    # decryptModelWeights.cpp:45-49: an unreadable key is "[decrypt] ERROR ..." + exit 1 (here also when no device)
    ref = json.load(open(os.path.join(golden_dir, "cc_params.json")))
    cc = tmp_path / "CC.json"
    cc.write_text(json.dumps(openfhe_style_cc(ref)))
    (tmp_path / "junk.key").write_text("not a key")
    r = run("decryptModelWeights", cc, tmp_path / "junk.key", tmp_path / "a", tmp_path / "b")
    assert r.returncode == 1 and "[decrypt] ERROR" in r.stderr


def _small_cc(tmp_path, batch=None):
    """genCC at the reference's depth/scaling; BatchSize as asked (default N/2 = 8192)."""
    cfg = tmp_path / "config_cc.json"
    cfg.write_text(json.dumps({"MultiplicativeDepth": 2, "ScalingModSize": 40, "BatchSize": batch or 8192,
                               "PREMode": "INDCPA"}))
    cc = tmp_path / "CC.json"
    r = run("genCC", cfg, cc)
    assert r.returncode == 0, r.stderr
    return cc


def _weights(tmp_path, name, layers):
    p = tmp_path / name
    p.write_text(json.dumps({"weights_summary": [
        {"layer": n, "shape": [len(v)], "mean": float(np.mean(v)) if len(v) else 0.0,
         "std_dev": float(np.std(v)) if len(v) else 0.0, "values": [float(x) for x in v]} for n, v in layers]}))
    return p


@pytest.mark.gpu
@pytest.mark.parametrize("n", [3, 5])
def test_server_round_n_clients_equals_the_per_client_programs(tmp_path, n):
    """the n-client round.  Route A (the reference's loop, orchestration/server_fns.sh:62-80): one
    changeCipherDomain per client 1..n-1 into client n's domain, then aggregateEncryptedWeights over the n files.
    Route B: ONE serverRound call (mkckks_reencrypt_sum_batch + rescale_mult_const(1/n)).  PRE is deterministic, so
    the two aggregate files must be identical byte for byte; decrypted, they are the plaintext mean within 2^-25."""
    cc = _small_cc(tmp_path)
    rng = np.random.default_rng(40 + n)

    def ok(r):
        assert r.returncode == 0, r.stdout + r.stderr
        return r

    vals = [[("dense", rng.uniform(-0.3, 0.3, 700)), ("bias", rng.uniform(-0.3, 0.3, 5)), ("empty", [])] for _ in range(n)]
    for c in range(n):
        ok(run("keyGen", cc, tmp_path / f"pk{c}", tmp_path / f"sk{c}"))
    for c in range(n):
        ok(run("encryptModelWeights", cc, tmp_path / f"pk{c}", _weights(tmp_path, f"w{c}.json", vals[c]), tmp_path / f"enc{c}.json"))
    target = n - 1
    for c in range(n - 1):
        ok(run("REkeyGen", cc, tmp_path / f"sk{c}", tmp_path / f"pk{target}", tmp_path / f"rk{c}"))
        ok(run("changeCipherDomain", cc, tmp_path / f"rk{c}", tmp_path / f"enc{c}.json", tmp_path / f"pre{c}.json"))
    # route A: target's own file first (the reference's <client2_encfile>), then the re-encrypted ones
    ok(run("aggregateEncryptedWeights", cc, tmp_path / f"enc{target}.json", tmp_path / "pre0.json", tmp_path / "aggA.json",
           *[tmp_path / f"pre{c}.json" for c in range(1, n - 1)]))
    # route B: one program; "-" marks the client that already is in the target domain
    args = []
    for c in range(n - 1):
        args += [tmp_path / f"rk{c}", tmp_path / f"enc{c}.json"]
    r = ok(run("serverRound", cc, tmp_path / "aggB.json", "-", tmp_path / f"enc{target}.json", *args))
    assert "[round] Re-encryption and aggregation completed successfully" in r.stdout
    a, b = json.load(open(tmp_path / "aggA.json")), json.load(open(tmp_path / "aggB.json"))
    assert [l["layer"] for l in a["weights_summary"]] == ["dense", "bias", "empty"]
    # route B orders the re-keyed clients first: the modular sum does not depend on the order, the blobs are identical
    assert a == b
    ok(run("decryptModelWeights", cc, tmp_path / f"sk{target}", tmp_path / "aggB.json", tmp_path / "dec.json"))
    dec = json.load(open(tmp_path / "dec.json"))["weights_summary"]
    for li, (name, _) in enumerate(vals[0][:2]):
        mean = np.mean([np.asarray(vals[c][li][1]) for c in range(n)], axis=0)
        assert np.abs(np.array(dec[li]["values"]) - mean).max() < 2.0 ** -25, name
    assert dec[2]["values"] == []
    # every client re-keyed (no "-"): same sum as changeCipherDomain on all n + aggregate
    r = run("serverRound", cc, tmp_path / "x.json", tmp_path / "rk0")  # odd argument count -> usage, exit 1
    assert r.returncode == 1 and "Usage:" in r.stderr
    r = run("serverRound", cc, tmp_path / "x.json", tmp_path / "nope", tmp_path / "enc0.json")
    assert r.returncode == 1 and "[round] ERROR: Failed to load ReKey" in r.stderr


@pytest.mark.gpu
def test_server_round_io_pipeline_writes_the_same_bytes(tmp_path):
    """SURVEY 8f f1, second half: binary (MKWS) envelopes go through the I/O pipeline of host/iopipe.hpp -- indexed files,
    reader threads into pinned slots, asynchronous uploads (mkckks_upload_async), range check on the device, results
    written with pwrite() from pinned slots -- instead of the synchronous path (read_envelope -> decode_ct -> validate_ct
    -> mkckks_upload; the reference's shape: changeCipherDomain.cpp:61-117, aggregateEncryptedWeights.cpp:54-119).  Both
    must write byte-identical aggregate and --back files; a residue at or above its modulus is refused on both."""
    n = 4
    cc = _small_cc(tmp_path)
    rng = np.random.default_rng(91)

    def ok(r):
        assert r.returncode == 0, r.stdout + r.stderr
        return r

    # 3 layers -> 2 + 2 + 2 + (3 + 1 + 0) = 10 ciphertexts per client: more than one pass of a small slot ring
    vals = [[("dense", rng.uniform(-0.3, 0.3, 3 * 8192 - 5)), ("bias", rng.uniform(-0.3, 0.3, 7)), ("empty", [])] for _ in range(n)]
    for c in range(n):
        ok(run("keyGen", cc, tmp_path / f"pk{c}", tmp_path / f"sk{c}"))
        ok(run("encryptModelWeights", cc, tmp_path / f"pk{c}", _weights(tmp_path, f"w{c}.json", vals[c]), tmp_path / f"enc{c}.mkws"))
        assert open(tmp_path / f"enc{c}.mkws", "rb").read(4) == b"MKWS"
    target = n - 1
    args, back_a, back_b = [], [], []
    for c in range(n - 1):
        ok(run("REkeyGen", cc, tmp_path / f"sk{c}", tmp_path / f"pk{target}", tmp_path / f"rk{c}"))
        args += [tmp_path / f"rk{c}", tmp_path / f"enc{c}.mkws"]
    ok(run("REkeyGen", cc, tmp_path / f"sk{target}", tmp_path / "pk0", tmp_path / "rkback0"))
    back_a = ["--back", tmp_path / "rkback0", tmp_path / "backA.mkws"]
    back_b = ["--back", tmp_path / "rkback0", tmp_path / "backB.mkws"]
    ra = ok(run("serverRound", cc, tmp_path / "aggA.mkws", "-", tmp_path / f"enc{target}.mkws", *args, *back_a,
                env={"MKCKKS_SYNC_IO": "1"}))
    rb = ok(run("serverRound", cc, tmp_path / "aggB.mkws", "-", tmp_path / f"enc{target}.mkws", *args, *back_b,
                env={"MKCKKS_IO_THREADS": "3"}))
    assert "[round] timing:" in rb.stdout and "[round] timing:" not in ra.stdout
    assert open(tmp_path / "aggA.mkws", "rb").read() == open(tmp_path / "aggB.mkws", "rb").read()
    assert open(tmp_path / "backA.mkws", "rb").read() == open(tmp_path / "backB.mkws", "rb").read()
    # the aggregate decrypts to the mean
    ok(run("decryptModelWeights", cc, tmp_path / f"sk{target}", tmp_path / "aggB.mkws", tmp_path / "dec.json"))
    dec = json.load(open(tmp_path / "dec.json"))["weights_summary"]
    mean = np.mean([np.asarray(vals[c][0][1]) for c in range(n)], axis=0)
    assert np.abs(np.array(dec[0]["values"]) - mean).max() < 2.0 ** -25
    # every client re-keyed (no "-"), one I/O thread
    rc = ok(run("serverRound", cc, tmp_path / "aggC.mkws", *args, env={"MKCKKS_IO_THREADS": "1"}))
    rd = ok(run("serverRound", cc, tmp_path / "aggD.mkws", *args, env={"MKCKKS_SYNC_IO": "1"}))
    assert open(tmp_path / "aggC.mkws", "rb").read() == open(tmp_path / "aggD.mkws", "rb").read()
    assert "[round] timing:" in rc.stdout and rd.returncode == 0
    # three rounds in ONE process (serverRound <cc> --rounds <file>; the loop of orchestration/run.sh:37-43): context, keys
    # (by file name), device arrays, pinned buffers and resolved kernels stay; every round writes the bytes of its one-shot run
    rounds = tmp_path / "rounds.txt"
    rounds.write_text(
        " ".join(map(str, [tmp_path / "r1.mkws", "-", tmp_path / f"enc{target}.mkws", *args, "--back", tmp_path / "rkback0",
                           tmp_path / "r1back.mkws"])) + "\n# a comment line\n\n"
        + " ".join(map(str, [tmp_path / "r2.mkws", *args])) + "\n"
        + " ".join(map(str, [tmp_path / "r3.mkws", "-", tmp_path / f"enc{target}.mkws", *args])) + "\n")
    rr = ok(run("serverRound", cc, "--rounds", rounds))
    assert "[round] 3 rounds, " in rr.stdout and rr.stdout.count("[round] timing:") == 3
    same = lambda a, b: open(tmp_path / a, "rb").read() == open(tmp_path / b, "rb").read()  # noqa: E731
    assert same("r1.mkws", "aggB.mkws") and same("r1back.mkws", "backB.mkws")
    assert same("r2.mkws", "aggC.mkws") and same("r3.mkws", "aggB.mkws")
    rounds.write_text(f"{tmp_path / 'x.mkws'} {tmp_path / 'rk0'}\n")  # a key without its ciphertext file
    r = run("serverRound", cc, "--rounds", rounds)
    assert r.returncode == 1 and "malformed round" in r.stderr
    # a residue that is not below its modulus: both paths refuse the file (the pipelined one after the device-side check)
    raw = bytearray(open(tmp_path / "enc0.mkws", "rb").read())
    skel_len = int.from_bytes(raw[8:16], "little")
    first_blob = 16 + skel_len + 8 + 8            # magic/version/skeleton size, skeleton, blob count, first blob size
    raw[first_blob + 48 + 8 * 17:first_blob + 48 + 8 * 18] = (2 ** 64 - 1).to_bytes(8, "little")
    open(tmp_path / "bad.mkws", "wb").write(raw)
    for env in ({"MKCKKS_SYNC_IO": "1"}, {}):
        r = run("serverRound", cc, tmp_path / "x.mkws", tmp_path / "rk0", tmp_path / "bad.mkws", "-", tmp_path / f"enc{target}.mkws", env=env)
        assert r.returncode == 1 and "residue not below its modulus" in r.stderr, r.stdout + r.stderr
    # a truncated file
    open(tmp_path / "trunc.mkws", "wb").write(bytes(raw[:len(raw) // 2]))
    r = run("serverRound", cc, tmp_path / "x.mkws", tmp_path / "rk0", tmp_path / "trunc.mkws", "-", tmp_path / f"enc{target}.mkws")
    assert r.returncode == 1


@pytest.mark.gpu
def test_server_round_sends_the_aggregate_back_to_every_client(tmp_path):
    """The last leg of the round (orchestration/server_fns.sh:76-80, run.sh:42): the aggregate, which lives in the target
    client's key domain at one limb fewer, is re-encrypted into each other client's domain.  `serverRound ... --back`
    does it while the aggregate is still in HBM; every file must equal changeCipherDomain run on the aggregate file,
    and every client must decrypt the mean with its OWN secret key."""
    n = 3
    cc = _small_cc(tmp_path)
    rng = np.random.default_rng(77)

    def ok(r):
        assert r.returncode == 0, r.stdout + r.stderr
        return r

    vals = [[("dense", rng.uniform(-0.3, 0.3, 300)), ("bias", rng.uniform(-0.3, 0.3, 4))] for _ in range(n)]
    for c in range(n):
        ok(run("keyGen", cc, tmp_path / f"pk{c}", tmp_path / f"sk{c}"))
        ok(run("encryptModelWeights", cc, tmp_path / f"pk{c}", _weights(tmp_path, f"w{c}.json", vals[c]), tmp_path / f"enc{c}.json"))
    target = n - 1
    args, back = [], []
    for c in range(n - 1):
        ok(run("REkeyGen", cc, tmp_path / f"sk{c}", tmp_path / f"pk{target}", tmp_path / f"rk{c}"))
        ok(run("REkeyGen", cc, tmp_path / f"sk{target}", tmp_path / f"pk{c}", tmp_path / f"rkback{c}"))
        args += [tmp_path / f"rk{c}", tmp_path / f"enc{c}.json"]
        back += [tmp_path / f"rkback{c}", tmp_path / f"agg_for{c}.json"]
    r = ok(run("serverRound", cc, tmp_path / "agg.json", "-", tmp_path / f"enc{target}.json", *args, "--back", *back))
    assert r.stdout.count("[round] aggregate re-encrypted with") == n - 1
    mean = [np.mean([np.asarray(vals[c][li][1]) for c in range(n)], axis=0) for li in range(2)]
    for c in range(n - 1):
        ok(run("changeCipherDomain", cc, tmp_path / f"rkback{c}", tmp_path / "agg.json", tmp_path / f"ref_for{c}.json"))
        assert json.load(open(tmp_path / f"agg_for{c}.json")) == json.load(open(tmp_path / f"ref_for{c}.json"))
        ok(run("decryptModelWeights", cc, tmp_path / f"sk{c}", tmp_path / f"agg_for{c}.json", tmp_path / f"dec{c}.json"))
        dec = json.load(open(tmp_path / f"dec{c}.json"))["weights_summary"]
        for li in range(2):
            assert np.abs(np.array(dec[li]["values"]) - mean[li]).max() < 2.0 ** -24, (c, li)
    # the wrong key must not decrypt it (sanity of the test itself)
    ok(run("decryptModelWeights", cc, tmp_path / f"sk{target}", tmp_path / "agg_for0.json", tmp_path / "wrong.json"))
    wrong = json.load(open(tmp_path / "wrong.json"))["weights_summary"]
    assert not (np.abs(np.array(wrong[0]["values"], dtype=float) - mean[0]).max() < 1.0)
    # usage: --back with nothing after it, or an odd count
    r = run("serverRound", cc, tmp_path / "x.json", "-", tmp_path / "enc0.json", "--back")
    assert r.returncode == 1 and "Usage:" in r.stderr
    r = run("serverRound", cc, tmp_path / "x.json", "-", tmp_path / "enc0.json", "--back", tmp_path / "rkback0")
    assert r.returncode == 1 and "Usage:" in r.stderr
    r = run("serverRound", cc, tmp_path / "x.json", "-", tmp_path / f"enc{target}.json", "--back", tmp_path / "nope", tmp_path / "y.json")
    assert r.returncode == 1 and "[round] ERROR: Failed to load ReKey" in r.stderr


@pytest.mark.gpu
def test_aggregate_emits_one_entry_per_matching_pair(tmp_path):
    """aggregateEncryptedWeights.cpp:68-72,115: the reference's nested loops emit an output entry for EVERY (w2, w1) pair
    with equal layer and shape -- duplicate layer names included -- and none for unmatched entries."""
    cc = _small_cc(tmp_path)
    rng = np.random.default_rng(7)

    def ok(r):
        assert r.returncode == 0, r.stdout + r.stderr
        return r

    a1, a2, b1 = rng.uniform(-0.3, 0.3, 6), rng.uniform(-0.3, 0.3, 6), rng.uniform(-0.3, 0.3, 6)
    ok(run("keyGen", cc, tmp_path / "pk", tmp_path / "sk"))
    # file 2 (the reference's client 2): layer "dup" twice, and a layer the other file lacks
    ok(run("encryptModelWeights", cc, tmp_path / "pk", _weights(tmp_path, "w2.json", [("dup", a1), ("dup", a2), ("only2", a1)]),
           tmp_path / "enc2.json"))
    ok(run("encryptModelWeights", cc, tmp_path / "pk", _weights(tmp_path, "w1.json", [("dup", b1), ("other_shape", b1[:3])]),
           tmp_path / "enc1.json"))
    ok(run("aggregateEncryptedWeights", cc, tmp_path / "enc2.json", tmp_path / "enc1.json", tmp_path / "agg.json"))
    ok(run("decryptModelWeights", cc, tmp_path / "sk", tmp_path / "agg.json", tmp_path / "dec.json"))
    dec = json.load(open(tmp_path / "dec.json"))["weights_summary"]
    assert [l["layer"] for l in dec] == ["dup", "dup"]            # one entry per matching pair, in file-2 order
    assert np.abs(np.array(dec[0]["values"]) - (a1 + b1) / 2).max() < 2.0 ** -25
    assert np.abs(np.array(dec[1]["values"]) - (a2 + b1) / 2).max() < 2.0 ** -25


@pytest.mark.gpu
def test_batch_size_below_half_ring(tmp_path):
    """genCC accepts BatchSize < N/2 (genCC.cpp:54): encryptModelWeights then packs BatchSize values per ciphertext and
    decryptModelWeights must return exactly those (decryptModelWeights.cpp:109-110), also for a layer that spans several
    ciphertexts with a ragged tail."""
    cc = _small_cc(tmp_path, batch=256)
    rng = np.random.default_rng(9)
    vals = rng.uniform(-0.3, 0.3, 256 * 2 + 77)  # three ciphertexts, the last one ragged

    def ok(r):
        assert r.returncode == 0, r.stdout + r.stderr
        return r

    ok(run("keyGen", cc, tmp_path / "pk", tmp_path / "sk"))
    r = ok(run("encryptModelWeights", cc, tmp_path / "pk", _weights(tmp_path, "w.json", [("big", vals)]), tmp_path / "enc.json"))
    assert "Batch size from CryptoContext = 256" in r.stdout
    assert len(json.load(open(tmp_path / "enc.json"))["weights_summary"][0]["values"]) == 3
    ok(run("decryptModelWeights", cc, tmp_path / "sk", tmp_path / "enc.json", tmp_path / "dec.json"))
    got = np.array(json.load(open(tmp_path / "dec.json"))["weights_summary"][0]["values"])
    assert got.size == vals.size
    assert np.abs(got - vals).max() < 2.0 ** -25


@pytest.mark.gpu
def test_malformed_client_ciphertexts_are_refused(tmp_path):
    """A client file is untrusted input: residues at or above their modulus, a level that contradicts the limb count, or
    a noiseScaleDeg outside {1, 2} must end in "[...] ERROR" + exit 1, never in a silently wrong aggregate."""
    import base64
    import struct
    cc = _small_cc(tmp_path)

    def ok(r):
        assert r.returncode == 0, r.stdout + r.stderr
        return r

    ok(run("keyGen", cc, tmp_path / "pk", tmp_path / "sk"))
    ok(run("keyGen", cc, tmp_path / "pk2", tmp_path / "sk2"))
    ok(run("REkeyGen", cc, tmp_path / "sk", tmp_path / "pk2", tmp_path / "rk"))
    ok(run("encryptModelWeights", cc, tmp_path / "pk", _weights(tmp_path, "w.json", [("l", np.linspace(-0.2, 0.2, 9))]),
           tmp_path / "enc.json"))
    good = json.load(open(tmp_path / "enc.json"))

    def tamper(fn, name):
        doc = json.loads(json.dumps(good))
        blob = bytearray(base64.b64decode(doc["weights_summary"][0]["mean"]))
        fn(blob)
        doc["weights_summary"][0]["mean"] = base64.b64encode(bytes(blob)).decode()
        p = tmp_path / name
        p.write_text(json.dumps(doc))
        return p

    def residue_too_big(b):
        b[48:56] = struct.pack("<Q", (1 << 64) - 1)
    def wrong_level(b):
        b[24:28] = struct.pack("<I", 2)     # header.level, although all 4 limbs are present
    def wrong_deg(b):
        b[28:32] = struct.pack("<I", 7)     # header.noise_deg
    for fn, name in ((residue_too_big, "big.json"), (wrong_level, "lvl.json"), (wrong_deg, "deg.json")):
        bad = tamper(fn, name)
        r = run("changeCipherDomain", cc, tmp_path / "rk", bad, tmp_path / "o.json")
        assert r.returncode == 1 and "[recrypt] ERROR" in r.stderr, name
        r = run("aggregateEncryptedWeights", cc, tmp_path / "enc.json", bad, tmp_path / "o.json")
        assert r.returncode == 1 and "[agg] ERROR" in r.stderr, name
        r = run("serverRound", cc, tmp_path / "o.json", "-", tmp_path / "enc.json", tmp_path / "rk", bad)
        assert r.returncode == 1 and "[round] ERROR" in r.stderr, name


def read_key_container(path):
    """BlobHeader (48 bytes) + residues of a key container written by the hosts: (kind, ring_dim, limbs, parts, data)."""
    import struct
    raw = open(path, "rb").read()
    magic, ver, kind, ring, limbs, parts = struct.unpack("<4sIIIII", raw[:24])
    assert magic == b"MKCK" and ver == 1
    data = np.frombuffer(raw[48:48 + 8 * ring * limbs * parts], dtype=np.uint64).reshape(parts, limbs, ring)
    return kind, ring, limbs, parts, data


def openfhe_style_dcrtpoly(limbs, moduli, fmt=0):
    return {"cereal_class_version": 1, "f": fmt,
            "v": [{"cereal_class_version": 1, "f": fmt,
                   "v": {"polymorphic_id": 1073741824,
                         "ptr_wrapper": {"valid": 1, "data": {"cereal_class_version": 1, "v": [int(x) for x in limb],
                                                              "m": {"v": int(m)}}}}}
                  for limb, m in zip(limbs, moduli)]}


@pytest.mark.gpu
def test_public_and_reencryption_keys_in_openfhe_json_form(tmp_path):
    """SURVEY 8f row f2, the part without a reference fixture (PARITY UNPINNED: the reference's public-key and ReKey
    blobs are missing): keys made by this project's programs are re-written in the nesting OpenFHE's JSON serialiser uses
    for DCRTPoly elements (the private-key fixture's shape; PublicKeyImpl "h" = [b, a], EvalKeyRelinImpl "k" =
    [A vector, B vector]) and must drive encryptModelWeights / REkeyGen / changeCipherDomain / serverRound exactly like
    the containers they came from -- re-encryption is deterministic, so outputs are compared byte for byte."""
    cc = _small_cc(tmp_path)
    moduli_cc = json.load(open(cc))["mkckks_cc"]

    def ok(r):
        assert r.returncode == 0, r.stdout + r.stderr
        return r

    for c in (1, 2):
        ok(run("keyGen", cc, tmp_path / f"pk{c}", tmp_path / f"sk{c}"))
    ok(run("REkeyGen", cc, tmp_path / "sk1", tmp_path / "pk2", tmp_path / "rk1"))
    from ppqsflhe_amd import Context
    g = Context(moduli_cc["log_n"], moduli_cc["MultiplicativeDepth"], moduli_cc["ScalingModSize"], moduli_cc["FirstModSize"],
                dnum=moduli_cc["NumLargeDigits"], device=-1)
    mods = [int(m) for m in g.moduli]
    g.close()
    # public key of client 2 and re-encryption key 1 -> 2 as OpenFHE-shaped JSON
    kind, ring, limbs, parts, pk2 = read_key_container(tmp_path / "pk2")
    assert (kind, parts, limbs) == (2, 2, len(mods))
    pk_json = {"value0": {"polymorphic_id": 1073741824, "ptr_wrapper": {"id": 2147483649, "data": {
        "cereal_class_version": 0, "h": [openfhe_style_dcrtpoly(pk2[0], mods), openfhe_style_dcrtpoly(pk2[1], mods)]}}}}
    (tmp_path / "pk2.json").write_text(json.dumps(pk_json))
    kind, ring, limbs, parts, rk = read_key_container(tmp_path / "rk1")
    beta = parts // 2
    assert kind == 4 and beta >= 2
    rk_json = {"value0": {"polymorphic_id": 1073741824, "ptr_wrapper": {"id": 2147483649, "data": {
        "cereal_class_version": 0,
        "k": [[openfhe_style_dcrtpoly(rk[2 * d + 1], mods) for d in range(beta)],     # A vector
              [openfhe_style_dcrtpoly(rk[2 * d], mods) for d in range(beta)]]}}}}     # B vector
    (tmp_path / "rk1.json").write_text(json.dumps(rk_json))

    vals = np.linspace(-0.25, 0.25, 300)
    w = _weights(tmp_path, "w.json", [("l", vals)])
    ok(run("encryptModelWeights", cc, tmp_path / "pk1", w, tmp_path / "enc1.json"))
    # re-encryption key: JSON form == container form, byte for byte
    ok(run("changeCipherDomain", cc, tmp_path / "rk1", tmp_path / "enc1.json", tmp_path / "a.json"))
    r = ok(run("changeCipherDomain", cc, tmp_path / "rk1.json", tmp_path / "enc1.json", tmp_path / "b.json"))
    assert "[recrypt] ReKey loaded" in r.stdout
    assert open(tmp_path / "a.json").read() == open(tmp_path / "b.json").read()
    ok(run("serverRound", cc, tmp_path / "c.json", tmp_path / "rk1.json", tmp_path / "enc1.json"))
    ok(run("serverRound", cc, tmp_path / "d.json", tmp_path / "rk1", tmp_path / "enc1.json"))
    assert open(tmp_path / "c.json").read() == open(tmp_path / "d.json").read()
    # public key: encrypt under the JSON form, decrypt with the matching secret key
    ok(run("encryptModelWeights", cc, tmp_path / "pk2.json", w, tmp_path / "enc2.json"))
    ok(run("decryptModelWeights", cc, tmp_path / "sk2", tmp_path / "enc2.json", tmp_path / "dec2.json"))
    got = np.array(json.load(open(tmp_path / "dec2.json"))["weights_summary"][0]["values"])
    assert np.abs(got - vals).max() < 2.0 ** -25
    # ... and as the peer key of REkeyGen: client 1's ciphertext, moved with that key, opens under client 2's secret
    ok(run("REkeyGen", cc, tmp_path / "sk1", tmp_path / "pk2.json", tmp_path / "rk1b"))
    ok(run("changeCipherDomain", cc, tmp_path / "rk1b", tmp_path / "enc1.json", tmp_path / "e.json"))
    ok(run("decryptModelWeights", cc, tmp_path / "sk2", tmp_path / "e.json", tmp_path / "dec_e.json"))
    got = np.array(json.load(open(tmp_path / "dec_e.json"))["weights_summary"][0]["values"])
    assert np.abs(got - vals).max() < 2.0 ** -25
    # COEFFICIENT-format elements on file ("f": 1) are transformed on import: same key, same output
    from oracle.oracle import OracleContext
    o = OracleContext(moduli_cc["log_n"], moduli_cc["MultiplicativeDepth"], moduli_cc["ScalingModSize"],
                      moduli_cc["FirstModSize"], dnum=moduli_cc["NumLargeDigits"])
    coef = lambda poly: [o.ntt_inv(i, poly[i]) for i in range(len(mods))]
    rk_c = {"value0": {"ptr_wrapper": {"data": {
        "k": [[openfhe_style_dcrtpoly(coef(rk[2 * d + 1]), mods, 1) for d in range(beta)],
              [openfhe_style_dcrtpoly(coef(rk[2 * d]), mods, 1) for d in range(beta)]]}}}}
    (tmp_path / "rk1c.json").write_text(json.dumps(rk_c))
    ok(run("changeCipherDomain", cc, tmp_path / "rk1c.json", tmp_path / "enc1.json", tmp_path / "f.json"))
    assert open(tmp_path / "f.json").read() == open(tmp_path / "a.json").read()
    # tampering is refused: a residue at its modulus, a missing digit
    bad = json.loads(json.dumps(rk_json))
    bad["value0"]["ptr_wrapper"]["data"]["k"][0][0]["v"][0]["v"]["ptr_wrapper"]["data"]["v"][3] = mods[0]
    (tmp_path / "bad1.json").write_text(json.dumps(bad))
    r = run("changeCipherDomain", cc, tmp_path / "bad1.json", tmp_path / "enc1.json", tmp_path / "x.json")
    assert r.returncode == 1 and "ERROR" in r.stderr
    bad = json.loads(json.dumps(rk_json))
    bad["value0"]["ptr_wrapper"]["data"]["k"][1].pop()
    (tmp_path / "bad2.json").write_text(json.dumps(bad))
    r = run("changeCipherDomain", cc, tmp_path / "bad2.json", tmp_path / "enc1.json", tmp_path / "x.json")
    assert r.returncode == 1 and "ERROR" in r.stderr
