"""Generates tests/golden/prng_kat.json: known answers of shaders/random.glsl:6-33 (tea, lcg, rnd)
from a THIRD transcription (pure Python integers), independent of oracle.cpp and the HIP code.
The values of SURVEY.md Appendix C are included verbatim as a second source."""
import json
import os

M = 0xFFFFFFFF


def tea(v0, v1):
    s0 = 0
    for _ in range(16):
        s0 = (s0 + 0x9E3779B9) & M
        v0 = (v0 + ((((v1 << 4) & M) + 0xA341316C) & M ^ ((v1 + s0) & M) ^ (((v1 >> 5) + 0xC8013EA4) & M))) & M
        v1 = (v1 + ((((v0 << 4) & M) + 0xAD90777D) & M ^ ((v0 + s0) & M) ^ (((v0 >> 5) + 0x7E95761E) & M))) & M
    return v0


def lcg(s):
    s = (1664525 * s + 1013904223) & M
    return s, s & 0x00FFFFFF


pairs = [(0, 0), (1, 0), (0, 1), (2, 0), (12345, 6789), (922159, 0), (M, M), (7, 0xDEADBEEF), (1280 * 719 + 1279, 63)]
out = {"tea": [[a, b, tea(a, b)] for a, b in pairs], "lcg": [], "survey_appendix_c": {
    "tea": [[0, 0, 0x741C187D], [1, 0, 0x8DA6B311], [0, 1, 0x70D3AEF1], [2, 0, 0x260277A2], [12345, 6789, 0x2F5102D4],
            [922159, 0, 0xA3EB65C8], [M, M, 0x16E50358]],
    "lcg_from_0": [[0x3C6EF35F, 7271263], [0x47502932, 5253426], [0xD1CCF6E9, 13432553]],
    "rnd_chain_from_tea00": [0.8242144584655762, 0.0048784613609313965, 0.7542978525161743, 0.06636053323745728],
    "distinct_seed_indices": {"1280x720": 283600, "256x256": 17668}}}
for start in (0, tea(0, 0), 0xFFFFFFFF, 123456789):
    s, seq = start, []
    for _ in range(6):
        s, bits = lcg(s)
        seq.append([s, bits])
    out["lcg"].append({"start": start, "seq": seq})
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "prng_kat.json"), "w"), indent=1)
print("wrote prng_kat.json")
