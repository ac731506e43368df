# round 4: SQ counters of k_deblock_band, one-row schedule (head) against two-row (lag2): where does the 8 % per step go?  One stream, one picture per launch.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/pm4; rm -f gpurun_out/pm4/*
for w in head lag2; do
  L=$GRAFT_REPO_ROOT/jmcodec_amd/lib/libjm_amd_dec.so; [ $w = lag2 ] && L=$GRAFT_REPO_ROOT/jmcodec_amd/lib_dbg_lag2/libjm_amd_dec.so
  for c in "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INSTS_BRANCH"; do
    n=$(echo $c | tr ' ' '_' | cut -c1-40)
    JM_AMD_DEC_LIB=$L JM_AMD_DEC_CHAIN_DEPTH=1 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/prof_pm -- python3 bench.py --steps 1 --warmup 1 --frames 30 --streams 1 --no-extra --no-cpu-baseline --no-single --device-output > /dev/null 2>&1
    find gpurun_out/prof_pm -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} gpurun_out/pm4/pmc_${w}_$n.csv; rm -rf gpurun_out/prof_pm
  done
done
python3 - <<'PY'
import csv,glob,collections
for f in sorted(glob.glob("gpurun_out/pm4/pmc_*.csv")):
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0].split('::')[-1][:26]
        acc[k][r['Counter_Name']]+=float(r['Counter_Value']); cnt[(k,r['Counter_Name'])]+=1
    for k,d in acc.items():
        if 'deblock_band' in k or 'intra_band' in k:
            print(f.split('/')[-1][4:9], k, {c: round(v/cnt[(k,c)]) for c,v in d.items()}, "launches", max(cnt[(k,c)] for c in d))
PY
