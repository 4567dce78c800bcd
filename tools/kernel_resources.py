"""Prints registers, spills, scratch and occupancy per kernel from `hipcc -Rpass-analysis=kernel-resource-usage` output (stdin or file)."""
import re
import sys

txt = open(sys.argv[1]).read() if len(sys.argv) > 1 else sys.stdin.read()
for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
    name = b.split("\n")[0]

    def g(k):
        m = re.search(k + r": (\d+)", b)
        return m.group(1) if m else "?"

    short = re.sub(r"_ZN\d+_GLOBAL__N_1", "", name)[:64]
    print(f"{short:66s} vgpr {g('VGPRs'):>4s} agpr {g('AGPRs'):>3s} spill {g('VGPRs Spill'):>3s} scratch {g('ScratchSize .bytes/lane.'):>4s} "
          f"occ {g('Occupancy .waves/SIMD.'):>2s} lds {g('LDS Size .bytes/block.')}")
