#!/bin/bash
# PMC passes over a kernel micro-benchmark (one counter group per pass, no tracing flags): tools/pmc_attn.sh <attn|attn16|gemm16|conv16> [kernel-name filter]
set -e
what=${1:-attn}
filt=${2:-attn}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp; cd /tmp
mkdir -p $R/gpurun_out
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32"; do
    tag=$(echo $grp | cut -d' ' -f1)
    d=/tmp/pmc_attn_$tag; rm -rf $d
    timeout -k 10 240 rocprofv3 --pmc $grp --output-format csv -d $d -- python3 $R/tools/bench_kernels.py $what --iters 2 > /dev/null 2> $d.err || { echo "pass $tag failed"; tail -3 $d.err; continue; }
    f=$(find $d -name '*counter_collection.csv' | head -1)
    python3 - "$f" "$filt" <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k=r['Kernel_Name']
    if sys.argv[2] not in k: continue
    key=(k[:40], r.get('Grid_Size'), r.get('Workgroup_Size'))
    acc[key][r['Counter_Name']].append(float(r['Counter_Value']))
for key,c in acc.items():
    print(key, {n: round(sum(v)/len(v)) for n,v in c.items()}, 'n=',len(next(iter(c.values()))))
PY
done
