#!/bin/bash
# SQ-level counters of the recompute pair's kernels alone on the GPU (tools/pwdwbench.py), per kernel name.
#   bash tools/pwdw_pmc.sh [pairs]        -> gpurun_out/pwdw_pmc/summary.txt
set -u
PAIRS=${1:-1,2}
R=$(pwd); OUT=$R/gpurun_out/pwdw_pmc; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o bench -- python3 $R/tools/pwdwbench.py --pairs $PAIRS --reps 5 > /dev/null 2> $OUT/p$i.err
done
cd $R
python3 - <<'PY'
import csv, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pwdw_pmc/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        if '(anonymous namespace)' not in n:
            continue
        n = re.sub(r'\(.*', '', n.replace('void (anonymous namespace)::', ''))
        agg[n][r['Counter_Name']].append(float(r['Counter_Value']))
names = sorted(agg, key=lambda k: -sum(agg[k].get('GRBM_GUI_ACTIVE', [0])))
cols = ['SQ_WAVES', 'GRBM_GUI_ACTIVE', 'SQ_WAVE_CYCLES', 'SQ_BUSY_CYCLES', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_VALU', 'SQ_WAIT_INST_ANY',
        'SQ_WAIT_ANY', 'SQ_INSTS_VALU', 'SQ_INSTS_MFMA', 'SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_INSTS_LDS', 'SQ_INSTS_VMEM_RD', 'SQ_INSTS_VMEM_WR',
        'SQ_WAIT_INST_LDS', 'SQ_LDS_BANK_CONFLICT', 'SQ_LDS_IDX_ACTIVE', 'SQ_ACTIVE_INST_LDS', 'SQ_ACTIVE_INST_VMEM', 'SQ_INST_CYCLES_VMEM', 'SQ_INSTS_SALU']
with open('gpurun_out/pwdw_pmc/summary.txt', 'w') as out:
    for n in names[:12]:
        a = agg[n]
        m = {c: (sum(a[c]) / len(a[c]) if a.get(c) else float('nan')) for c in cols}
        wc = m['SQ_WAVE_CYCLES']
        line = ("%-44s launches %3d  waves %8.0f  gui %9.0f | per wave-cycle: active %.2f valu %.2f wait_inst %.2f wait_any %.2f lds_wait %.3f"
                " | insts/wave: valu %6.0f salu %6.0f mfma %5.0f lds %5.0f vmem_rd %5.0f vmem_wr %5.0f | lds_conflict/lds_active %.3f | valu wave-instr per gui cycle %.1f"
                % (n[:44], len(a.get('SQ_WAVES', [])), m['SQ_WAVES'], m['GRBM_GUI_ACTIVE'], m['SQ_ACTIVE_INST_ANY'] / wc, m['SQ_ACTIVE_INST_VALU'] / wc,
                   m['SQ_WAIT_INST_ANY'] / wc, m['SQ_WAIT_ANY'] / wc, m['SQ_WAIT_INST_LDS'] / wc, m['SQ_INSTS_VALU'] / m['SQ_WAVES'], m['SQ_INSTS_SALU'] / m['SQ_WAVES'],
                   m['SQ_INSTS_MFMA'] / m['SQ_WAVES'], m['SQ_INSTS_LDS'] / m['SQ_WAVES'], m['SQ_INSTS_VMEM_RD'] / m['SQ_WAVES'],
                   m['SQ_INSTS_VMEM_WR'] / m['SQ_WAVES'],
                   m['SQ_LDS_BANK_CONFLICT'] / max(m['SQ_LDS_IDX_ACTIVE'], 1), m['SQ_INSTS_VALU'] / max(m['GRBM_GUI_ACTIVE'], 1)))
        print(line); out.write(line + "\n")
PY
tail -3 $OUT/p1.err | head -3
