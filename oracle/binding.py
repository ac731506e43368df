"""ctypes bindings of the CPU oracle (oracle/_build/liborc.so, liborc_hevc.so).

TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline / bit-exact check may import this.
The decode product (jmcodec_amd/, libjm_amd_dec.so) never does (tests/test_abi.py::test_product_does_not_link_the_oracle).
"""
import ctypes as C
import os
import subprocess

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_oracle():
    subprocess.check_call(["make", "-C", os.path.join(_ROOT, "oracle")], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


class Oracle:
    """ctypes binding of oracle/_build/liborc.so (CPU oracle -- checker / cpu_baseline only)."""

    def __init__(self):
        p = os.path.join(_ROOT, "oracle", "_build", "liborc.so")
        if not os.path.exists(p):
            build_oracle()
        L = C.CDLL(p)
        L.orc_decode_stream_to_buffer.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.POINTER(C.POINTER(C.c_ubyte)),
                                                  C.POINTER(C.c_size_t), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_free.argtypes = [C.c_void_p]
        L.orc_open.restype = C.c_void_p
        L.orc_open.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_close.argtypes = [C.c_void_p]
        L.orc_decode_annexb.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        L.orc_flush.argtypes = [C.c_void_p]
        L.orc_digest_enable.argtypes = [C.c_void_p]
        L.orc_digest_value.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        L.orc_digest_value.restype = C.c_uint64
        L.orc_last_error.argtypes = [C.c_void_p]
        L.orc_last_error.restype = C.c_char_p
        L.orc_packout.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p, C.POINTER(C.c_int)]
        self.L = L

    def decode(self, data, out_fmt=1):
        """Returns (frames_bytes, n_frames, width, height): all display-order frames concatenated."""
        buf = C.POINTER(C.c_ubyte)()
        n = C.c_size_t(0)
        w, h = C.c_int(0), C.c_int(0)
        cnt = self.L.orc_decode_stream_to_buffer(data, len(data), out_fmt, C.byref(buf), C.byref(n), C.byref(w), C.byref(h))
        if cnt < 0:
            raise RuntimeError("oracle decode failed")
        out = C.string_at(buf, n.value) if n.value else b""
        self.L.orc_free(buf)
        return out, cnt, w.value, h.value

    def display_pocs(self, data):
        """PicOrderCnt of every output frame, display order (OrcFrame.poc at the moment the frame is handed out)."""
        class Frame(C.Structure):
            _fields_ = [("y", C.c_void_p), ("u", C.c_void_p), ("v", C.c_void_p), ("width", C.c_int), ("height", C.c_int), ("stride_y", C.c_int), ("stride_c",
                C.c_int),
                        ("poc", C.c_int), ("frame_type", C.c_int), ("decode_index", C.c_int)]
        pocs = []
        cb = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(Frame))(lambda user, f: pocs.append(f.contents.poc))
        d = self.L.orc_open(C.cast(cb, C.c_void_p), None)
        rc = self.L.orc_decode_annexb(d, data, len(data))
        err = self.L.orc_last_error(d).decode()
        self.L.orc_flush(d)
        self.L.orc_close(d)
        if rc < 0:
            raise RuntimeError("oracle: " + err)
        return pocs

    def syntax_digest(self, data):
        """(digest, n_macroblocks) of the parsed syntax elements (see oracle/orc_slice.c digest_mb)."""
        d = self.L.orc_open(None, None)
        self.L.orc_digest_enable(d)
        rc = self.L.orc_decode_annexb(d, data, len(data))
        err = self.L.orc_last_error(d).decode()
        self.L.orc_flush(d)
        n = C.c_uint64(0)
        v = self.L.orc_digest_value(d, C.byref(n))
        self.L.orc_close(d)
        if rc < 0:
            raise RuntimeError("oracle: " + err)
        return v, n.value

    def tools(self, data):
        """Which coding tools the stream exercises: {counter name: macroblocks / slices} (non-zero entries)."""
        L = self.L
        L.orc_tool_name.restype = C.c_char_p
        L.orc_tool_name.argtypes = [C.c_int]
        L.orc_tool_count.restype = C.c_long
        L.orc_tool_count.argtypes = [C.c_void_p, C.c_int]
        d = L.orc_open(None, None)
        rc = L.orc_decode_annexb(d, data, len(data))
        err = L.orc_last_error(d).decode()
        L.orc_flush(d)
        out, i = {}, 0
        while L.orc_tool_name(i):
            if L.orc_tool_count(d, i):
                out[L.orc_tool_name(i).decode()] = L.orc_tool_count(d, i)
            i += 1
        L.orc_close(d)
        if rc < 0:
            raise RuntimeError("oracle: " + err)
        return out

    def packout(self, src, pitch, width, height, out_fmt):
        cap = width * height * 3 // 2
        dst = C.create_string_buffer(cap)
        n = C.c_int(cap)
        rc = self.L.orc_packout(src, pitch, width, height, out_fmt, dst, C.byref(n))
        return rc, dst.raw[:n.value]



class OracleHevc:
    """ctypes binding of oracle/_build/liborc_hevc.so (CPU oracle -- checker only)."""

    def __init__(self):
        p = os.path.join(_ROOT, "oracle", "_build", "liborc_hevc.so")
        if not os.path.exists(p):
            build_oracle()
        L = C.CDLL(p)
        L.orch_decode_stream_to_buffer.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.POINTER(C.POINTER(C.c_ubyte)), C.POINTER(C.c_size_t),
            C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orch_free.argtypes = [C.c_void_p]
        L.orch_open.restype = C.c_void_p
        L.orch_open.argtypes = [C.c_void_p, C.c_void_p]
        L.orch_close.argtypes = [C.c_void_p]
        L.orch_decode_annexb.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        L.orch_flush.argtypes = [C.c_void_p]
        L.orch_digest_enable.argtypes = [C.c_void_p]
        L.orch_digest_value.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        L.orch_digest_value.restype = C.c_uint64
        L.orch_last_error.argtypes = [C.c_void_p]
        L.orch_last_error.restype = C.c_char_p
        L.orch_tool_name.restype = C.c_char_p
        L.orch_tool_name.argtypes = [C.c_int]
        L.orch_tool_count.restype = C.c_long
        L.orch_tool_count.argtypes = [C.c_void_p, C.c_int]
        self.L = L

    def decode(self, data, out_fmt=1):
        buf = C.POINTER(C.c_ubyte)()
        n = C.c_size_t(0)
        w, h = C.c_int(0), C.c_int(0)
        cnt = self.L.orch_decode_stream_to_buffer(data, len(data), out_fmt, C.byref(buf), C.byref(n), C.byref(w), C.byref(h))
        if cnt < 0:
            raise RuntimeError("HEVC oracle decode failed")
        out = C.string_at(buf, n.value) if n.value else b""
        self.L.orch_free(buf)
        return out, cnt, w.value, h.value

    def _run(self, data, digest=False):
        d = self.L.orch_open(None, None)
        if digest:
            self.L.orch_digest_enable(d)
        rc = self.L.orch_decode_annexb(d, data, len(data))
        err = self.L.orch_last_error(d).decode()
        self.L.orch_flush(d)
        return d, rc, err

    def display_pocs(self, data):
        """PicOrderCnt of every output frame, display order (OrcFrame.poc at the moment the frame is handed out)."""
        class Frame(C.Structure):
            _fields_ = [("y", C.c_void_p), ("u", C.c_void_p), ("v", C.c_void_p), ("width", C.c_int), ("height", C.c_int), ("stride_y", C.c_int), ("stride_c",
                C.c_int),
                        ("poc", C.c_int), ("slice_type", C.c_int), ("decode_index", C.c_int)]
        pocs = []
        cb = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(Frame))(lambda user, f: pocs.append(f.contents.poc))
        d = self.L.orch_open(C.cast(cb, C.c_void_p), None)
        rc = self.L.orch_decode_annexb(d, data, len(data))
        err = self.L.orch_last_error(d).decode()
        self.L.orch_flush(d)
        self.L.orch_close(d)
        if rc < 0:
            raise RuntimeError("HEVC oracle: " + err)
        return pocs

    def syntax_digest(self, data):
        d, rc, err = self._run(data, True)
        n = C.c_uint64(0)
        v = self.L.orch_digest_value(d, C.byref(n))
        self.L.orch_close(d)
        if rc < 0:
            raise RuntimeError("HEVC oracle: " + err)
        return v, n.value

    def tools(self, data):
        d, rc, err = self._run(data)
        out, i = {}, 0
        while True:
            name = self.L.orch_tool_name(i)
            if name is None:
                break
            c = self.L.orch_tool_count(d, i)
            if c:
                out[name.decode()] = c
            i += 1
        self.L.orch_close(d)
        if rc < 0:
            raise RuntimeError("HEVC oracle: " + err)
        return out
