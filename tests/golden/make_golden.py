#!/usr/bin/env python3
"""Generates tests/golden/click_track_v1.npz with the CPU oracle.

The reference (a Rust crate whose FFT and filter live in un-vendored crates)
cannot be built or run here, so these vectors are produced by the build's own
restatement (oracle/), NOT by the reference: they pin the oracle and the HIP
path to each other and to today's behaviour, and record which of the unpinned
third-party choices (divisor / DC / inclusivity / Nyquist packing) each vector
was made with.  Inputs come from the integer-only generator, so the file can be
regenerated bit-identically:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import beat_detector_amd as bd  # noqa: E402  (host generator only, no GPU)
from oracle import bd_oracle as o  # noqa: E402

SR = 44100
x = bd.synth_fill_host(2, SR * 3, first_stream=40, seed=20240117, sampling_rate=SR)
cases = [
    ("spectrum_L512_default", o.KIND_SPECTRUM, 512, {}),
    ("spectrum_L1024_default", o.KIND_SPECTRUM, 1024, {}),
    ("spectrum_L529_default", o.KIND_SPECTRUM, 529, {}),
    ("spectrum_L1024_div_fftlen", o.KIND_SPECTRUM, 1024, {"spectrum_divisor": o.DIV_FFT_LEN}),
    ("spectrum_L1024_nodc", o.KIND_SPECTRUM, 1024, {"spectrum_include_dc": 0}),
    ("spectrum_L512_nyq", o.KIND_SPECTRUM, 512, {"spectrum_dc_packs_nyquist": 1}),
    ("lpf_L1024_default", o.KIND_LPF, 1024, {}),
    ("lpf_L512_default", o.KIND_LPF, 512, {}),
    ("lpf_L317_default", o.KIND_LPF, 317, {}),
]
out = {"samples": x}
meta = {"sampling_rate": SR, "generator": {"first_stream": 40, "seed": 20240117}, "cases": {}}
for name, kind, L, over in cases:
    p = o.default_params(**over)
    vals, cands, beats = [], [], []
    for row in x:
        v, c, b = o.run_stream(kind, SR, row, L, params=p, full=True)
        vals.append(v); cands.append(c); beats.append(b)
    out[name + "__values"] = np.stack(vals)
    out[name + "__candidates"] = np.stack(cands)
    nb = max(len(b) for b in beats)
    bi = np.full((len(beats), max(nb, 1), 2), -1, dtype=np.int64)
    for s, b in enumerate(beats):
        for i, (j, ms) in enumerate(b):
            bi[s, i] = (j, ms)
    out[name + "__beats"] = bi
    meta["cases"][name] = {"kind": kind, "callback_len": L, "params": over}
out["meta_json"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
np.savez_compressed(os.path.join(os.path.dirname(__file__), "click_track_v1.npz"), **out)
print({k: v.shape for k, v in out.items()})
