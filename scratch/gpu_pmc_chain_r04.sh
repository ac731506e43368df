# round 4: where do k_chain's fetched bytes come from?  L2 request counters of the one-stream chain run (one --pmc pass per group; kernel trace only)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; P=gpurun_out/pm5; mkdir -p $P
rocprofv3 --list-avail 2>/dev/null | grep -oE "\b(TCC|TCP_TCC|TCP)_[A-Z0-9_]+" | sort -u > $P/avail.txt; wc -l $P/avail.txt
run() {  # run <tag> <counters...>
  tag=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/prof_x -- python3 bench.py --steps 1 --warmup 1 --frames 30 --streams 1 --no-cpu-baseline --no-single --no-extra --device-output > /dev/null 2>$P/err_$tag.txt
  f=$(find gpurun_out/prof_x -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python3 tools/pmc_summary.py $f | grep -E "k_chain|k_recon_inter|k_deblock_band"; else echo "$tag: no csv"; tail -3 $P/err_$tag.txt; fi
  rm -rf gpurun_out/prof_x
}
run a TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum
run b TCC_READ_sum TCC_WRITE_sum TCC_ATOMIC_sum
run c TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
run d TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum TCC_EA_WRREQ_sum TCC_EA_WRREQ_64B_sum
run e TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum
run f TCP_TCC_NC_READ_REQ_sum TCP_TCC_UC_READ_REQ_sum TCP_TCC_CC_READ_REQ_sum TCP_TCC_RW_READ_REQ_sum
run g TCC_STREAMING_REQ_sum TCC_NC_REQ_sum TCC_UC_REQ_sum TCC_CC_REQ_sum
run h TCC_PROBE_sum TCC_TAG_STALL_sum TCC_BUBBLE_sum
