#!/usr/bin/env python3
"""One-device timing of the C++ driver (cmx::MultiGpu::enqueueIntraStats + fetchRows, host memory to host memory) on a
bench.py workload, beside bench.py's own host_to_host figure.  VERDICT r3 item 3: "within 3 % of bench.py's step".
usage (GPU box, repo root): python scripts/time_multigpu_cpp.py [target|cfg3|cfg4] [reps]"""
import os
import struct
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from comap_amd import engine as E

wl = sys.argv[1] if len(sys.argv) > 1 else "target"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
w = bench.WORKLOADS[wl]
assert w["statistic"] == "Correlation", "the C++ test driver scores with CorrelationStatistic"
parent, blen, lot, mdl, Bk, clamp = bench.build_inputs(w)
eng = E.Engine(parent, blen, lot, mdl["Q"], mdl["pi"], mdl["rates"], mdl["probs"])
aln, _ = eng.simulate(w["seed"] + 1, 0, w["nsites"])
eng.close()
path = "/tmp/mg_case.bin"
nn, T, S, C = len(parent), len(lot), len(mdl["pi"]), len(mdl["rates"])
with open(path, "wb") as f:
    f.write(struct.pack("<8i", nn, T, S, C, w["nsites"], w["nrep"](1), w["rep_ram"], w["nclasses"]) + struct.pack("<Q", w["seed"] + 7))
    f.write(np.asarray(parent, dtype=np.int32).tobytes() + np.asarray(blen, dtype=np.float64).tobytes() + np.asarray(lot, dtype=np.int32).tobytes())
    f.write(np.asarray(mdl["Q"]).tobytes() + np.asarray(mdl["pi"]).tobytes() + np.asarray(mdl["rates"]).tobytes() + np.asarray(mdl["probs"]).tobytes())
    f.write(np.ascontiguousarray(aln).tobytes())
exe = os.path.join(ROOT, "tests", "cpp", "multigpu_main")
src = os.path.join(ROOT, "tests", "cpp", "multigpu_main.cpp")
if not os.path.exists(exe) or os.path.getmtime(exe) < os.path.getmtime(src):
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include",
                           src, "-o", exe, "-L", os.path.dirname(E.LIB_PATH), "-lcomap_mi355x", "-L", "/opt/rocm/lib", "-lrccl", "-lamdhip64",
                           "-Wl,-rpath," + os.path.dirname(E.LIB_PATH), "-Wl,-rpath,/opt/rocm/lib"])
for records in (False, True):   # rows written by the devices / 16-byte records + rows rebuilt on the host (enableCompactTransfer)
    env = dict(os.environ, **({"CMX_MG_RECORDS": "1"} if records else {}))
    env.pop("CMX_MG_RECORDS", None) if not records else None
    for args in (["1", str(reps)], ["2", str(reps), "loopback"]):
        print(("records " if records else "rows    ") + subprocess.check_output([exe, "time", path] + args, text=True, env=env).strip(), flush=True)
