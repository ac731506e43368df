# round 4, VERDICT r3 next 3: (a) does a chain launch of C2 (16 streams of 4K High I B B P) still give up?  Chains forced on for such a load
# (JM_AMD_DEC_CHAIN_STREAMS=64), JM_AMD_DEC_VERBOSE dumps the census and every band counter of a launch that gives up (Engine::dump_chain_state);
# the round-3 library beside HEAD's, six runs each.  (b) SQ / TCC counters of k_chain on the C4 slice (8 streams of 1080p).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/ch; rm -f gpurun_out/ch/*
R3=$GRAFT_REPO_ROOT/scratch/_ab/r3/jmcodec_amd/lib/libjm_amd_dec.so
HEAD=$GRAFT_REPO_ROOT/jmcodec_amd/lib/libjm_amd_dec.so
C2="--tools high_b --width 3840 --height 2160 --streams 16 --frames 24 --steps 3 --no-extra --no-cpu-baseline --no-single"
python bench.py $C2 > /dev/null 2>&1     # streams into the cache
for i in 1 2 3 4 5 6; do
  for w in head r3; do
    L=$HEAD; [ $w = r3 ] && L=$R3
    JM_AMD_DEC_LIB=$L JM_AMD_DEC_CHAIN_STREAMS=64 JM_AMD_DEC_VERBOSE=1 timeout 300 python bench.py $C2 > gpurun_out/ch/c2_${w}_$i.json 2> gpurun_out/ch/c2_${w}_$i.err
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/ch/c2_*.json")):
    try: l=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, "no line", e); continue
    print(f.split('/')[-1], l["value"], l["bit_exact"], "chain batches", l["engine"]["chain_batches_whole_run"], "recoveries", l["engine"]["chain_recoveries_whole_run"], "errors", l["decode_errors"], {k:(v["avg_us"],v["pictures_per_launch"]) for k,v in l["kernels"].items() if v["launches"]})
PY
grep -h -A40 "gave up" gpurun_out/ch/c2_head_*.err | head -120
# (b) counters
for c in "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INST_LEVEL_VMEM" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum"; do
  n=$(echo $c | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/prof_pm -- python3 bench.py --steps 2 --warmup 1 --frames 60 --streams 8 --no-extra --no-cpu-baseline --no-single --device-output > /dev/null 2> gpurun_out/ch/pmc_$n.err
  find gpurun_out/prof_pm -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} gpurun_out/ch/pmc_$n.csv; rm -rf gpurun_out/prof_pm
done
python3 - <<'PY'
import csv,glob,collections
for f in sorted(glob.glob("gpurun_out/ch/pmc_*.csv")):
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0].split('::')[-1][:26]
        acc[k][r['Counter_Name']]+=float(r['Counter_Value']); cnt[(k,r['Counter_Name'])]+=1
    for k,d in acc.items():
        if 'chain' in k or 'recon_inter' in k:
            print(k, {c: round(v/cnt[(k,c)]) for c,v in d.items()}, "launches", max(cnt[(k,c)] for c in d))
PY
