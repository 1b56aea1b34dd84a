#!/bin/bash
# development aid (GPU box): SQ counters and HBM traffic of the fused kernel on one cfg2 batch
export TMPDIR=/tmp
R=$PWD
cat > /tmp/one_batch.py <<'PY'
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"] if "GRAFT_REPO_ROOT" in os.environ else os.getcwd())
import numpy as np
from robot_mpcs_amd._lib import Solver
from robot_mpcs_amd.scenarios import make_scenario
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
sc = make_scenario(cfg, seed=1000)
s = Solver(sc.desc, max_batch=sc.B)
for _ in range(3):
    r = s.solve(sc.xinit, sc.x0, sc.params)
print("iters", r["iters"].mean())
PY
CFG=${1:-cfg2}
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS" \
           "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_WAVES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/pmcf_$i
  rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmcf_$i -- python3 /tmp/one_batch.py $CFG > /dev/null 2> $R/gpurun_out/pmcf_$i.err || tail -3 $R/gpurun_out/pmcf_$i.err
done
python3 - <<'PY'
import csv, glob, collections
per = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmcf_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0]
        name = "k_fused" if "k_fused" in name else ("k_sweep" if "k_sweep" in name else ("k_riccati" if "k_riccati" in name else ("k_step" if "k_step" in name else name[-24:])))
        per[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, c in per.items():
    if name not in ("k_fused", "k_sweep", "k_riccati", "k_step"): continue
    print(name, {k: f"{sum(v)/len(v):.4g} (x{len(v)})" for k, v in sorted(c.items())})
PY
