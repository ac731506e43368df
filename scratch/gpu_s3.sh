for s in 32 48 64; do
timeout 300 python bench.py --streams $s --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('streams $s', d['value'], d['pcie_out']['achieved'], d['host_ms_per_picture']['wait_slot_ns'], {k:(v['avg_us'], v['pictures_per_launch']) for k,v in d['kernels'].items() if k!='k_intra'})"
done
timeout 300 python bench.py --codec hevc --streams 32 --frames 32 --width 1920 --height 1080 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('hevc 1080p 32', d['value'], d['host_ms_per_picture'], {k:(v['avg_us'], v['pictures_per_launch']) for k,v in d['kernels'].items()})"
