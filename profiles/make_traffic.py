"""Turns a counter summary (profiles/summarise_pmc.py) into the record bench.py reads from profiles/hbm_traffic.json:
    python profiles/make_traffic.py <summary.json> <workload> <kernel name as bench.py prints it> [scale]
scale: the counter passes rendered the frame at 1/divisor of its width and height: the per-launch figures are multiplied
by divisor^2 (scale) to stand for the full frame, and the record says so.
hbm_bytes: (2 * FETCH_SIZE + WRITE_SIZE) KiB per launch; FETCH_SIZE doubled as the gfx950 correction for wide coalesced
reads prescribes (this kernel's reads are small cached gathers, for which the factor is not calibrated: an upper bound).
fp32_flops: (ADD + MUL + TRANS + 2 * FMA) wave instructions x 64 lanes x the measured share of active lanes
(SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU))."""
import json
import os
import sys

summary, workload, kernel = sys.argv[1:4]
scale = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0
c = json.load(open(summary))["counters"]
g = lambda k: c[k]["per_launch"]
lanes = g("SQ_THREAD_CYCLES_VALU") / (64.0 * g("SQ_ACTIVE_INST_VALU"))
wave_flop_instr = g("SQ_INSTS_VALU_ADD_F32") + g("SQ_INSTS_VALU_MUL_F32") + g("SQ_INSTS_VALU_TRANS_F32") + 2 * g("SQ_INSTS_VALU_FMA_F32")
rec = {
    "kernel": kernel,
    "hbm_bytes": int((2 * g("FETCH_SIZE") + g("WRITE_SIZE")) * 1024 * scale),
    "fetch_bytes": int(2 * g("FETCH_SIZE") * 1024 * scale),
    "write_bytes": int(g("WRITE_SIZE") * 1024 * scale),
    "fp32_flops": int(wave_flop_instr * 64 * lanes * scale),
    "scaled_by": scale,
    "lanes_active": lanes,
    "valu_wave_instructions": int(g("SQ_INSTS_VALU") * scale),
    "salu_wave_instructions": int(g("SQ_INSTS_SALU") * scale),
    "source": os.path.relpath(os.path.abspath(summary), os.path.dirname(os.path.abspath(__file__))),
}
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hbm_traffic.json")
try:
    data = json.load(open(path))
    data = {k: v for k, v in data.items() if isinstance(v, dict)}
except Exception:
    data = {}
data[workload] = rec
json.dump(data, open(path, "w"), indent=1)
print(json.dumps(rec, indent=1))
