for inf in 2 3 4; do for s in 32 48; do
JM_AMD_DEC_INFLIGHT=$inf timeout 300 python bench.py --streams $s --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('inflight $inf streams $s', d['value'], d['pcie_out']['achieved'], d['host_ms_per_picture']['wait_slot_ns'], {k:(v['avg_us'], v['pictures_per_launch']) for k,v in d['kernels'].items() if k!='k_intra'})"
done; done
