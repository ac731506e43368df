"""tools/streams.py helpers used by bench.py's checker leg: an IDR period cut out of a stream decodes (CPU oracle) to the same frames the
whole stream gives at that position -- what lets the bench compare a LATE period of the timed configuration with the oracle."""
import pytest

from tools import streams


@pytest.mark.parametrize("kw", [dict(width=64, height=48, frames=12, gop=4, mode=1, num_ref=2),
                                dict(width=64, height=48, frames=13, gop=6, mode=1, num_ref=2, cabac=1, t8x8=1, bframes=2, poc_type=0)])
def test_idr_periods_concatenate_to_the_whole_stream_h264(kw):
    data = streams.generate(**kw)
    orc = streams.Oracle()
    full, n, w, h = orc.decode(data, 1)
    fb, pos, k = w * h * 3 // 2, 0, 0
    while (seg := streams.idr_period(data, k, False)) is not None:
        yuv, m, _, _ = orc.decode(seg, 1)
        assert m > 0 and yuv == full[pos * fb:(pos + m) * fb], (k, m)
        pos, k = pos + m, k + 1
    assert pos == n and k == (kw["frames"] + kw["gop"] - 1) // kw["gop"]


def test_idr_periods_concatenate_to_the_whole_stream_hevc():
    cfg = streams.config_c3(frames=20, width=128, height=64)
    cfg.update(intra_period=8)
    data = streams.generate_hevc(**cfg)
    orc = streams.OracleHevc()
    full, n, w, h = orc.decode(data, 1)
    fb, pos, k = w * h * 3 // 2, 0, 0
    while (seg := streams.idr_period(data, k, True)) is not None:
        yuv, m, _, _ = orc.decode(seg, 1)
        assert m > 0 and yuv == full[pos * fb:(pos + m) * fb], (k, m)
        pos, k = pos + m, k + 1
    assert pos == n == 20 and k == 3
