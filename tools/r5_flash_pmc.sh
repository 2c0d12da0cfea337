#!/bin/bash
# usage: bash tools/r5_flash_pmc.sh [output name]   (bench_flash.py times the general kernels and the ones without a running maximum: both appear)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5_flash_pmc
rm -rf $O; mkdir -p $O
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $O/a -o a -- python3 tools/bench_flash.py > $O/a.log 2>&1 && echo a ok
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $O/b -o b -- python3 tools/bench_flash.py > $O/b.log 2>&1 && echo b ok
python - > $O/${1:-r5_flash_pmc}.txt <<'PY'
import csv, collections
print('# rocprofv3 --pmc passes of tools/bench_flash.py (tools/r5_flash_pmc.sh): per flash kernel instantiation and grid, median per launch')
for f in ("gpurun_out/r5_flash_pmc/a/a_counter_collection.csv", "gpurun_out/r5_flash_pmc/b/b_counter_collection.csv"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if "flash_attn" not in r["Kernel_Name"]: continue
        key = (r["Kernel_Name"].replace("void (anonymous namespace)::", "")[:48], r["Grid_Size"])
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        agg[key]["_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k, v in agg.items():
        print(k, {c: round(sorted(x)[len(x)//2], 1) for c, x in v.items()})
PY
rm -rf $O/a $O/b
cat $O/${1:-r5_flash_pmc}.txt
