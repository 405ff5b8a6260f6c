mkdir -p gpurun_out/c8; export MDE_PANEL=1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for dbg in 3 0; do
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SMEM"; do
  MDE_RING_DBG=$dbg timeout 120 rocprofv3 --kernel-trace --pmc $grp -d $R/gpurun_out/c8/p_${dbg}_$(echo $grp | cut -c4-12) --output-format csv -- $R/tools/kbench 1000000 50 3 > /dev/null 2>&1
done
done
cd $R
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/c8/*/*/*counter_collection.csv')):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if 'k_fused_ring' in r['Kernel_Name'] and 'Lb1ELb1' in r['Kernel_Name'] or ('k_fused_ring' in r['Kernel_Name'] and 'true, true' in r['Kernel_Name']):
            a = acc[r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
    print(f.split('/')[2], {k: v[0] / max(v[1], 1) for k, v in acc.items()})
PY
