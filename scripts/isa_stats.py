"""Instruction mix of one kernel in a gfx950 assembly listing (hipcc --save-temps).
usage: isa_stats.py LISTING.s KERNEL_SUBSTRING [--dump OUT.s]"""
import collections
import re
import sys

listing, needle = sys.argv[1], sys.argv[2]
lines = open(listing).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*:", l) and needle in l and "rocprim" not in l)
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
body = lines[start:end + 1]
if "--dump" in sys.argv:
    open(sys.argv[sys.argv.index("--dump") + 1], "w").write("\n".join(body))
mix = collections.Counter()
for l in body:
    t = l.strip()
    if not t or t.startswith((";", ".")) or t.endswith(":"):
        continue
    op = t.split()[0]
    kind = "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "other"
    mix[kind] += 1
    mix[op] += 1
print(lines[start])
print({k: mix[k] for k in ("valu", "salu", "lds", "vmem", "other")})
print(sorted(((v, k) for k, v in mix.items() if k not in ("valu", "salu", "lds", "vmem", "other")), reverse=True)[:45])
for l in lines[end:end + 80]:
    if re.search(r"NumVgprs|NumAgprs|ScratchSize|Occupancy|LDSByteSize|NumSgprs", l):
        print(l.strip())
