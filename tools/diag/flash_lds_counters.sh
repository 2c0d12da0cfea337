#!/bin/bash
# LDS bank-conflict counters of the VGGT flash kernel, one build per kind of LDS access removed (outputs of those builds
# are wrong - only the counters matter; guide 5.4 rule 17). Builds: see the commands in DESIGN.md / this directory's README.
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
for v in 0 1 2 3; do
  lib=$R/vggt_qwen3_amd/libvq3hip.so
  [ $v != 0 ] && lib=$R/tools/diag/libvq3hip_fadiag$v.so
  mkdir -p $R/gpurun_out/fa_diag$v
  VQ3_HIP_LIB=$lib timeout -k 5 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv \
      -d $R/gpurun_out/fa_diag$v -o l -- python $R/tools/bench_flash.py > /dev/null 2>&1
  python - <<PY
import csv, collections
d=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open("$R/gpurun_out/fa_diag$v/l_counter_collection.csv")):
    if "flash" in r["Kernel_Name"]:
        k=r["Kernel_Name"].split("flash_attn_hd64_kernel")[1][:3]
        d[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[(k,r["Counter_Name"])]+=1
for k,vv in sorted(d.items()):
    c=vv.get("SQ_LDS_BANK_CONFLICT",0); a=vv.get("SQ_LDS_IDX_ACTIVE",1); m=n[(k,"SQ_LDS_IDX_ACTIVE")]
    print("variant $v kernel", k, "conflict/launch %.0f  active/launch %.0f  ratio %.4f" % (c/m, a/m, c/a))
PY
done
