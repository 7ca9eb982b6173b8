#!/usr/bin/env python
"""hyena_ct_pmc_traffic.txt (tools/summarize_prof.py pmc over the two counter passes of tools/profile_hyena_ct.py: three launches at 8 x 8,193 and
three at 1 x 131,073) -> the HBM-side bytes per launch of both shapes as JSON, stamped with the commit of the run:
    bytes = TCC_EA0_RDREQ x 128 (gfx950: every read request is 128 B; the 32-byte class is counted separately and is 0 here)
          + TCC_EA0_WRREQ_64B x 64 + (TCC_EA0_WRREQ - TCC_EA0_WRREQ_64B) x 32        (MI355X_MICROARCH.md, HBM / rocprofv3 section)
    python tools/pmc_live.py <txt> <out.json> <commit>"""
import json
import sys

rows = {}
for line in open(sys.argv[1]):
    if "hyena_ct_kernel<false, " not in line:        # <false, true>: the scoring path's launch since round 6 (main tokens, end state out); <false, false>: the ragged-tile form
        continue
    f = line.split()
    name = [x for x in f if x.startswith("TCC_")][0]
    lo, hi = float(f[-2]), float(f[-1])
    rows[name] = (lo, hi)                                          # min = the 8 x 8,193 launches, max = the 1 x 131,073 ones


def total(i):
    rd = rows["TCC_EA0_RDREQ_sum"][i] - rows.get("TCC_EA0_RDREQ_32B_sum", (0, 0))[i]
    rd32 = rows.get("TCC_EA0_RDREQ_32B_sum", (0, 0))[i]
    w64 = rows["TCC_EA0_WRREQ_64B_sum"][i]
    w32 = rows["TCC_EA0_WRREQ_sum"][i] - w64
    return int(rd * 128 + rd32 * 32 + w64 * 64 + w32 * 32)


out = {"hyena_ct_kernel": {"B8_T8193": total(0), "B1_T131073": total(1)}, "_commit": {"hyena_ct_kernel": sys.argv[3]},
       "_source": "rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum / TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum (two passes, --kernel-trace only) of "
                  "tools/profile_hyena_ct.py in THIS tools/gpu_check.sh run"}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out["hyena_ct_kernel"]))
