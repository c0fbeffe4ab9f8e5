"""Checker infrastructure (build container only): extract the score-network sections of the reference's shipped YAML
configs — model.{n_feats,n_spks,spk_emb_dim,decoder,dit} — into tests/golden/ref_model_sections.json, so the host
mirror's config loader (dex_tts_amd.config.from_reference_yaml / dex_tts_amd.synthesize) can be tested on the GPU box and
on CPU without /root/reference.  Data only; no reference source text is stored.

    python -m oracle.dump_ref_configs
"""
import json
import os

import yaml

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "ref_model_sections.json")
CONFIGS = {
    "GeDEX-TTS/config/LJSpeech/base.yaml": "gedex",
    "GeDEX-TTS/config/VCTK/base.yaml": "gedex",
    "DEX-TTS/config/VCTK/base.yaml": "dex",
    "DEX-TTS/config/ESD/base.yaml": "dex",
    "DEX-TTS/config/LibriTTS/base.yaml": "dex",
}


def main():
    out = {}
    for rel, variant in CONFIGS.items():
        m = yaml.safe_load(open(os.path.join(REF, rel)))["model"]
        out[rel] = {"variant": variant, "n_feats": m["n_feats"], "n_spks": m["n_spks"], "spk_emb_dim": m["spk_emb_dim"],
                    "decoder": m["decoder"], "dit": m["dit"]}
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print(f"wrote {OUT}: {len(out)} configs")


if __name__ == "__main__":
    main()
