"""jmcodec_amd -- MI355X-native H.264 decode backend behind the jmcodec ``jm_nvdec_*`` API.

The product is ``lib/libjm_amd_dec.so`` (host C++17 + hand-written gfx950 HIP kernels,
sources in ``csrc/``).  This package is the thin Python mirror of the reference's
operator interface (``/root/reference/nv_dec/jm_nv_dec.h:20-88``): same function names,
argument meaning and return conventions, bound with ctypes.  There is no CPU decode path
in here: if the library is missing or no HIP device is present the calls fail loudly.
"""
from .api import (  # noqa: F401
    JmAmdDec,
    build,
    jm_nvdec_create_handle,
    jm_nvdec_decode_frame,
    jm_nvdec_deinit,
    jm_nvdec_init,
    jm_nvdec_is_exit,
    jm_nvdec_is_hw_support,
    jm_nvdec_output_frame,
    jm_nvdec_set_eof,
    jm_nvdec_show_dec_info,
    jm_nvdec_stream_info,
    lib,
    lib_path,
    split_nalus,
)
