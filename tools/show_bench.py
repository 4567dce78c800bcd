"""Development aid: the timing fields and ray counts of a bench.py line.  usage: python tools/show_bench.py <file.json>"""
import json, sys
d = json.load(open(sys.argv[1]))
print({k: d[k] for k in ("value", "ms_per_step", "frame_ms_alone", "value_one_frame_at_a_time")})
print(d["config"]["rays_per_frame"])
print({k: (round(v["ms"], 4), v["launches"]) for k, v in d["roofline"]["by_kernel"].items()})
