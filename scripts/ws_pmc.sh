#!/usr/bin/env bash
# PMC pass over one layer of the weight-stationary chain (run through gpurun): scripts/ws_pmc.sh <layer> <out tag> [B] [S]
set -uo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
L=${1:-conv5}; T=${2:-pmc}; B=${3:-8192}; S=${4:-12}
O=gpurun_out/r03/$T
mkdir -p "$O"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d "$O/sq" -- python3 scripts/ws_layer_bench.py $B $S $L > "$O/sq.log" 2>&1 || echo "sq failed"
python3 - "$O" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(sys.argv[1] + "/sq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
        if r["Counter_Name"] == "SQ_WAVE_CYCLES": n[k] += 1
for k, d in acc.items():
    if "conv_ws" not in k: continue
    print(k, "dispatches", n[k])
    for c, v in sorted(d.items()): print(f"   {c:28s} {v / max(n[k], 1):16.0f}")
    if d.get("SQ_LDS_IDX_ACTIVE"): print("   LDS conflict fraction:", d["SQ_LDS_BANK_CONFLICT"] / d["SQ_LDS_IDX_ACTIVE"])
PY
