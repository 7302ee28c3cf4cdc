#!/usr/bin/env python3
"""One case of tools/decision_audit.py (AMP, SNR, TFS from the environment), guard on, both levels: the disagreement counts."""
import json
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import dabtools_amd as dab  # noqa: E402
amp, snr, tfs = float(os.environ.get("AMP", 0.35)), float(os.environ.get("SNR", 5.0)), int(os.environ.get("TFS", 4400))
per_stream = 40
nstreams = (tfs + per_stream - 1) // per_stream
cfgs = [dab.synth_preset(0, seed=9000 + 131 * i + int(snr), cif_count0=(61 * i) % 5000, snr_db=snr, amplitude=amp) for i in range(nstreams)]
nbytes = dab.synth_bytes(cfgs[0], per_stream)
big = torch.empty(nstreams * nbytes, dtype=torch.uint8, device="cuda")
dab.synth_generate_device(cfgs, per_stream, [big.data_ptr() + i * nbytes for i in range(nstreams)])
torch.cuda.synchronize()
eng = dab.Engine(0)
for level in (1, 2):
    eng.set_parity_guard(level)
    for fused in (True, False):
        on = eng.decision_audit(guard=True, fused=fused, device_ptr=big.data_ptr(), nframes=nstreams * per_stream)
        print(json.dumps({"samplewise": os.environ.get("DABHIP_EXACT_SAMPLEWISE", "0"), "level": level, "fused": fused, "disagree": on["disagree"], "listed": on["listed"]}), flush=True)
