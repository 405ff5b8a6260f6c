# usage: pmc.sh OUTDIR DBG  -- SQ counters of the ring kernel (kbench, config-4 shape)
mkdir -p gpurun_out/$1; export MDE_PANEL=1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SMEM"; do
  MDE_RING_DBG=$2 timeout 120 rocprofv3 --kernel-trace --pmc $grp -d $R/gpurun_out/$1/p_$2_$(echo $grp | cut -c4-12) --output-format csv -- $R/tools/kbench 1000000 50 3 > /dev/null 2>&1
done
cd $R
python3 - $1 <<'PY'
import csv, glob, collections, sys
for f in sorted(glob.glob('gpurun_out/%s/*/*/*counter_collection.csv' % sys.argv[1])):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'k_fused_ring' in k and 'FnSingle<9, 2>, true, true' in k:
            a = acc[r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
    print(f.split('/')[2], {k: '%.4g' % (v[0] / max(v[1], 1)) for k, v in acc.items()})
PY
