#!/usr/bin/env python3
"""Build tests/golden/*.npz|json from the reference's DATA fixtures.

Run once in the build container (the reference does not travel to the GPU box):
    python tests/golden/make_fixtures.py [/root/reference]

Only JSON data files are parsed; no reference code is imported or executed.
Sources (SURVEY.md 8c "golden vectors"):
  server/storage/CC.json                                   parameter KAT (P1-P3)
  client/storage/client_{1,2}/private/client_{1,2}-private.key   NTT KAT (P4)
  client/storage/client_{1,2}/private/sample_weights_c{1,2}.json  } end-to-end
  client/storage/client_{1,2}/private/decrypted_weights_c{1,2}.json } KAT (P8)
"""
import json
import os
import sys

import numpy as np

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def cc_params():
    cc = json.load(open(os.path.join(REF, "server/storage/CC.json")))
    d = cc["value0"]["ptr_wrapper"]["data"]["cc"]["ptr_wrapper"]["data"]
    rns = d["value0"]  # CryptoParametersRNS level: ks, rs, dnum, ab, eb ...
    lvl1 = rns["value0"]  # CryptoParametersRLWE level: dp (sigma), md, mo ...
    base = lvl1["value0"]  # CryptoParametersBase level: elp, enp
    elp = base["elp"]["ptr_wrapper"]["data"]
    top = elp["value0"]
    limbs = [p["ptr_wrapper"]["data"]["value0"] for p in elp["p"]]
    enp = base["enp"]["ptr_wrapper"]["data"]
    out = {
        "source": "server/storage/CC.json",
        "ring_dim": top["rd"],
        "cyclotomic_order": top["co"],
        "moduli": [int(l["cm"]["v"]) for l in limbs],
        "roots": [int(l["ru"]["v"]) for l in limbs],
        "composite_modulus_words": [int(x) for x in top["cm"]["v"]],
        "composite_modulus_bits": top["cm"]["m"],
        "batch_size": enp["bs"],
        "scaling_bits": enp["m"],
        "sigma": lvl1["dp"],
        "mult_depth": lvl1["md"] if "md" in lvl1 else None,
        "pre_mode": lvl1.get("mo"),
        "key_switch_technique": rns["ks"],
        "scaling_technique": rns["rs"],
        "dnum": rns["dnum"],
        "aux_bits": rns["ab"],
        "extra_bits": rns["eb"],
    }
    json.dump(out, open(os.path.join(OUT, "cc_params.json"), "w"), indent=1)
    return out


def secret_keys():
    arrs = {}
    for c in (1, 2):
        p = os.path.join(REF, f"client/storage/client_{c}/private/client_{c}-private.key")
        sk = json.load(open(p))
        s = sk["value0"]["ptr_wrapper"]["data"]["s"]
        limbs = []
        mods = []
        for el in s["v"]:
            assert el["f"] == 0  # Format::EVALUATION
            dat = el["v"]["ptr_wrapper"]["data"]
            limbs.append(np.array(dat["v"], dtype=np.uint64))
            mods.append(int(dat["m"]["v"]))
        arrs[f"sk{c}_eval"] = np.stack(limbs)
        arrs[f"sk{c}_moduli"] = np.array(mods, dtype=np.uint64)
    np.savez_compressed(os.path.join(OUT, "sk_ntt_kat.npz"), **arrs)


def weights():
    keep_full = {"param_0", "param_2", "param_5", "param_6", "param_7"}
    arrs = {}
    meta = []
    for c in (1, 2):
        for kind in ("sample", "decrypted"):
            p = os.path.join(REF, f"client/storage/client_{c}/private/{kind}_weights_c{c}.json")
            w = json.load(open(p))["weights_summary"]
            for l in w:
                name = l["layer"]
                vals = np.array(l["values"], dtype=np.float64)
                if name not in keep_full:
                    if name != "param_1":
                        continue
                    vals = vals[:8192]
                arrs[f"{kind}_c{c}_{name}_values"] = vals
                arrs[f"{kind}_c{c}_{name}_mean_std"] = np.array([l["mean"], l["std_dev"]], dtype=np.float64)
                if c == 1 and kind == "sample":
                    meta.append({"layer": name, "shape": l["shape"], "kept": int(vals.size)})
    np.savez_compressed(os.path.join(OUT, "e2e_weights.npz"), **arrs)
    json.dump({"layers": meta,
               "note": "param_1 truncated to its first 8192 values (one full ciphertext); "
                       "param_3/param_4 omitted (same shape as param_1)"},
              open(os.path.join(OUT, "e2e_weights_meta.json"), "w"), indent=1)


if __name__ == "__main__":
    print(cc_params())
    secret_keys()
    weights()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))
