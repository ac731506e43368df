#!/usr/bin/env python3
"""Extract the H.264 track of an MP4/MOV file as an Annex-B elementary stream (what FFmpeg's h264_mp4toannexb filter does
for /root/reference/test_player/test_player.cpp:221-226).  Test tooling: used once to turn imageio's public sample clip into
tests/golden/thirdparty_realshort.h264; the product-side avcC handling lives in jmcodec_amd/api.py (avcc_to_annexb)."""
import struct, sys


def boxes(buf, start, end):
    p = start
    while p + 8 <= end:
        size, typ = struct.unpack(">I4s", buf[p:p + 8])
        hdr = 8
        if size == 1:
            size = struct.unpack(">Q", buf[p + 8:p + 16])[0]; hdr = 16
        elif size == 0:
            size = end - p
        yield typ, p + hdr, p + size
        p += size


def find(buf, start, end, path):
    for typ, a, b in boxes(buf, start, end):
        if typ == path[0]:
            if len(path) == 1:
                return a, b
            r = find(buf, a, b, path[1:])
            if r:
                return r
    return None


def extract(buf):
    moov = find(buf, 0, len(buf), [b"moov"])
    for typ, a, b in boxes(buf, *moov):
        if typ != b"trak":
            continue
        stbl = find(buf, a, b, [b"mdia", b"minf", b"stbl"])
        if not stbl:
            continue
        stsd = find(buf, *stbl, [b"stsd"])
        i = buf.find(b"avcC", stsd[0], stsd[1])
        if i < 0:
            continue
        c = buf[i + 4:stsd[1]]
        nal_len = (c[4] & 3) + 1
        out = bytearray()
        o = 6
        for _ in range(c[5] & 31):
            n = struct.unpack(">H", c[o:o + 2])[0]; out += b"\x00\x00\x00\x01" + c[o + 2:o + 2 + n]; o += 2 + n
        npps = c[o]; o += 1
        for _ in range(npps):
            n = struct.unpack(">H", c[o:o + 2])[0]; out += b"\x00\x00\x00\x01" + c[o + 2:o + 2 + n]; o += 2 + n
        a_, b_ = find(buf, *stbl, [b"stsz"])
        fixed, count = struct.unpack(">II", buf[a_ + 4:a_ + 12])
        sizes = [fixed] * count if fixed else list(struct.unpack(">%dI" % count, buf[a_ + 12:a_ + 12 + 4 * count]))
        co = find(buf, *stbl, [b"stco"])
        if co:
            n = struct.unpack(">I", buf[co[0] + 4:co[0] + 8])[0]
            chunks = list(struct.unpack(">%dI" % n, buf[co[0] + 8:co[0] + 8 + 4 * n]))
        else:
            co = find(buf, *stbl, [b"co64"])
            n = struct.unpack(">I", buf[co[0] + 4:co[0] + 8])[0]
            chunks = list(struct.unpack(">%dQ" % n, buf[co[0] + 8:co[0] + 8 + 8 * n]))
        a_, b_ = find(buf, *stbl, [b"stsc"])
        n = struct.unpack(">I", buf[a_ + 4:a_ + 8])[0]
        stsc = [struct.unpack(">III", buf[a_ + 8 + 12 * k:a_ + 20 + 12 * k]) for k in range(n)]
        si = 0
        for ci, off in enumerate(chunks):
            per = 0
            for first, cnt, _ in stsc:
                if ci + 1 >= first:
                    per = cnt
            for _ in range(per):
                if si >= len(sizes):
                    break
                p, e = off, off + sizes[si]
                while p + nal_len <= e:
                    n = int.from_bytes(buf[p:p + nal_len], "big")
                    out += b"\x00\x00\x00\x01" + buf[p + nal_len:p + nal_len + n]
                    p += nal_len + n
                off = e; si += 1
        return bytes(out), len(sizes)
    raise SystemExit("no avc1 track")


if __name__ == "__main__":
    data, n = extract(open(sys.argv[1], "rb").read())
    open(sys.argv[2], "wb").write(data)
    print("samples", n, "bytes", len(data))
