#!/usr/bin/env python3
"""profiles/traffic_r03.json from the PMC summaries of scripts/profile_r03.sh / profile_mica.sh (gpurun_out/prof_<tag>/summary.json):
HBM bytes per launch of the dominant kernel = (2 x FETCH_SIZE + WRITE_SIZE) KiB, FETCH_SIZE doubled per the gfx950 note of
MI355X_MICROARCH.md, stamped with the sha of the device sources so that bench.py attaches it only to the code it was taken on.
usage: scripts/make_traffic_json.py <round, e.g. r04> target=<tag> cfg3=<tag> cfg4=<tag> mica_cfg5=<tag>   -> profiles/traffic_<round>.json"""
import json
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import bench  # noqa: E402

ROUND = sys.argv[1]
KERNEL = {"target": "map_kernel<20, 1, 1, 4>", "cfg3": "map_kernel<20, 1, 1, 4>", "cfg4": "map_kernel<16, 1, 4, 4>", "mica_cfg5": "mica_mfma4_kernel<8, false>"}
out = {"kernel_source_sha": bench.kernel_source_sha(),
       "note": "HBM bytes per launch of the dominant kernel = (2 x FETCH_SIZE + WRITE_SIZE) KiB from separate rocprofv3 --pmc passes "
               "(profiles/" + ROUND + "_<workload>_pmc_summary.json; scripts/profile_r03.sh, scripts/profile_mica.sh); FETCH_SIZE doubled per the "
               "gfx950 note of MI355X_MICROARCH.md.  Attached by bench.py only while the device sources hash to kernel_source_sha.  "
               "Mapping workloads: measured at N = 1 with sites_per_launch simulated sites per launch, scaled by bench.py to the "
               "launch's own site count.",
       "sites_per_launch": {}}
for arg in sys.argv[2:]:
    w, tag = arg.split("=")
    d = json.load(open(os.path.join(ROOT, "gpurun_out", "prof_" + tag, "summary.json")))
    k = [v for name, v in d["kernels"].items() if KERNEL[w] in name]
    assert len(k) == 1, (w, [n for n in d["kernels"]])
    pmc = k[0]["pmc"]
    out[w] = (2.0 * pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024.0
    if w in bench.WORKLOADS:
        ww = bench.WORKLOADS[w]
        out["sites_per_launch"][w] = 2 * ww["nrep"](1) * ww["rep_ram"]
json.dump(out, open(os.path.join(ROOT, "profiles", "traffic_" + ROUND + ".json"), "w"), indent=1)
print(json.dumps(out, indent=1))
