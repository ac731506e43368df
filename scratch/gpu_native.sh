# the native harness (tools/test_amd_dec, the counterpart of the reference's test_nv_dec) on one stream: what ONE caller thread reaches without Python in the loop
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/nat
python - <<'PY'
import sys; sys.path.insert(0, '.')
from tools import streams
open('/tmp/c1.h264', 'wb').write(streams.generate(**streams.config_c1(stream_id=0, frames=300)))
open('/tmp/c3.h265', 'wb').write(streams.generate_hevc(**streams.config_c3(frames=128, width=1920, height=1080, stream_id=0)))
c = streams.config_c1(stream_id=1, frames=120); c.update(cabac=1, paff=1, num_ref=2, poc_type=0)
open('/tmp/paff.h264', 'wb').write(streams.generate(**c))
PY
export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/jmcodec_amd/lib:$LD_LIBRARY_PATH
for i in 1 2; do ./tools/_build/test_amd_dec /tmp/c1.h264 --loops 3 2>&1 | grep -i "fps\|frame count\|elapsed" | tr '\n' ' '; echo; done
for i in 1 2; do ./tools/_build/test_amd_dec /tmp/c3.h265 --codec 1 --loops 3 2>&1 | grep -i "fps\|frame count\|elapsed" | tr '\n' ' '; echo; done
./tools/_build/test_amd_dec /tmp/paff.h264 --loops 3 2>&1 | grep -i "fps\|frame count\|elapsed" | tr '\n' ' '; echo
