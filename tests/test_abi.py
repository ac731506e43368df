"""The drop-in boundary: the library loads, exports every symbol include/jm_amd_dec.h declares plus the ten
C++-mangled jm_nvdec_* names of the reference header, and the reference's own harness links against it."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

from jmcodec_amd import api
from util import ROOT

MANGLED = ["_Z22jm_nvdec_create_handlev", "_Z13jm_nvdec_initiiPciPv", "_Z15jm_nvdec_deinitPv", "_Z21jm_nvdec_decode_framePhiPiPv",
           "_Z21jm_nvdec_output_framePhPiPv", "_Z20jm_nvdec_stream_infoPiS_Pv", "_Z16jm_nvdec_set_eofbPv", "_Z16jm_nvdec_is_exitPv",
           "_Z22jm_nvdec_show_dec_infoPv", "_Z22jm_nvdec_is_hw_supportv"]


def _exports():
    out = subprocess.check_output(["nm", "-D", "--defined-only", api.lib_path()], text=True)
    return {l.split()[-1] for l in out.splitlines() if l.strip()}


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "jm_amd_dec.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(jm_amddec_\w+)\s*\(", hdr))
    assert len(declared) >= 14
    exp = _exports()
    assert declared <= exp, declared - exp
    assert set(MANGLED) <= exp, set(MANGLED) - exp
    L = api.lib()
    for name in declared:
        assert getattr(L, name)


def test_no_gpu_means_loud_failure_not_fallback():
    """Without a HIP device init must fail (the reference would return 0); nothing decodes on the CPU."""
    if api.jm_nvdec_is_hw_support():
        pytest.skip("a GPU is present")
    h = api.jm_nvdec_create_handle()
    assert api.jm_nvdec_init(0, 1, None, 0, h) != 0
    assert b"no HIP device" in api.lib().jm_amddec_last_error(h)
    assert api.jm_nvdec_decode_frame(b"\x00\x00\x01\x09\x10", 5, h)[0] != 0
    api.jm_nvdec_deinit(h)


def test_product_does_not_link_the_oracle():
    out = subprocess.check_output(["nm", "-D", api.lib_path()], text=True)
    assert "orc_" not in out
    for f in os.listdir(os.path.join(ROOT, "jmcodec_amd", "csrc")):
        src = open(os.path.join(ROOT, "jmcodec_amd", "csrc", f)).read()
        assert "oracle/" not in src and "orc_h264" not in src, f


@pytest.mark.skipif(not os.path.exists("/root/reference/test_nv_dec/test_nv_dec.cpp"), reason="reference sources only exist in the build container")
def test_reference_harness_links_against_the_library():
    """Compile /root/reference/test_nv_dec/test_nv_dec.cpp IN PLACE (never copied) with a two-file shim for
    <Windows.h>/<conio.h> and link it against libjm_amd_dec.so: every jm_nvdec_* symbol it uses must resolve."""
    if shutil.which("g++") is None or not os.path.exists("/root/reference/test_nv_dec/test_nv_dec.cpp"):
        pytest.skip("g++ or the reference tree is missing")
    with tempfile.TemporaryDirectory() as td:
        open(os.path.join(td, "Windows.h"), "w").write("#include <string.h>\n#include <stdlib.h>\n")
        open(os.path.join(td, "conio.h"), "w").write("static inline int _kbhit(void){return 0;}\nstatic inline int getch(void){return 0;}\n")
        exe = os.path.join(td, "test_nv_dec")
        cmd = ["g++", "-w", "-fpermissive", "-I" + td, "-I/root/reference/nv_dec",
               "-DJMDLL_FUNC=__attribute__((visibility(\"default\")))", "-DJMDLL_API=",
               "/root/reference/test_nv_dec/test_nv_dec.cpp", "-o", exe,
               "-L" + os.path.dirname(api.lib_path()), "-ljm_amd_dec", "-Wl,-rpath," + os.path.dirname(api.lib_path())]
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        und = subprocess.check_output(["nm", "-u", exe], text=True)
        used = set(re.findall(r"_Z\d+jm_nvdec_\w+", und))
        assert len(used) == 8 and used <= set(MANGLED)


# ---- push/pull API of intel_dec/jm_intel_dec.h ----
INTEL_MANGLED = ["_Z26jm_intel_dec_create_handlev", "_Z17jm_intel_dec_initiiPv", "_Z19jm_intel_dec_deinitPv",
                 "_Z29jm_intel_dec_set_yuv_callbackPvPFiPhiS_ES_", "_Z23jm_intel_dec_input_dataPhiPv", "_Z25jm_intel_dec_output_framePhPiPv",
                 "_Z20jm_intel_dec_set_eofiPv", "_Z17jm_intel_dec_infoPv", "_Z24jm_intel_get_stream_infoPiS_PfPv",
                 "_Z27jm_intel_dec_need_more_dataPv", "_Z25jm_intel_dec_free_buf_lenPv", "_Z20jm_intel_dec_is_exitPv", "_Z22jm_intel_is_hw_supportv"]


def test_intel_api_symbols():
    hdr = open(os.path.join(ROOT, "include", "jm_amd_intel_dec.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(jm_amdintel_\w+)\s*\(", hdr))
    assert len(declared) == 15         # the reference's 13 + jm_amdintel_run_pushpull + jm_amdintel_decoder (additions, no reference counterpart)
    exp = _exports()
    assert declared <= exp, declared - exp
    assert set(INTEL_MANGLED) <= exp, set(INTEL_MANGLED) - exp


def test_reference_intel_harness_links_against_the_library():
    """/root/reference/test_intel_dec/test_intel_dec.cpp compiled in place (never copied) links against the library."""
    if shutil.which("g++") is None or not os.path.exists("/root/reference/test_intel_dec/test_intel_dec.cpp"):
        pytest.skip("g++ or the reference tree is missing")
    with tempfile.TemporaryDirectory() as td:
        open(os.path.join(td, "Windows.h"), "w").write("#include <string.h>\n#include <stdlib.h>\n")
        open(os.path.join(td, "conio.h"), "w").write("static inline int _kbhit(void){return 0;}\nstatic inline int getch(void){return 0;}\n")
        exe = os.path.join(td, "test_intel_dec")
        cmd = ["g++", "-w", "-fpermissive", "-Wno-format-security", "-I" + td, "-I/root/reference/intel_dec",
               "-DJMDLL_FUNC=__attribute__((visibility(\"default\")))", "-DJMDLL_API=",
               "/root/reference/test_intel_dec/test_intel_dec.cpp", "-o", exe,
               "-L" + os.path.dirname(api.lib_path()), "-ljm_amd_dec", "-Wl,-rpath," + os.path.dirname(api.lib_path())]
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        und = subprocess.check_output(["nm", "-u", exe], text=True)
        used = set(re.findall(r"_Z\d+jm_intel_\w+", und))
        assert len(used) >= 9 and used <= set(INTEL_MANGLED), used - set(INTEL_MANGLED)
        # host-side run of the reference's own loop (parse-only: no GPU here): it must terminate and report the frame count
        from util import golden_stream
        src = os.path.join(td, "in.h264")
        open(src, "wb").write(golden_stream("ip_fuzz_96x80"))
        env = dict(os.environ, JM_AMD_DEC_PARSE_ONLY="1")
        r = subprocess.run([exe, src], capture_output=True, text=True, env=env, timeout=60, cwd=td)   # the harness opens its hard-coded output name in the cwd
        assert "Frame Count:\t8" in r.stdout, r.stdout + r.stderr


def _harness():
    return os.path.join(ROOT, "tools", "_build", "test_amd_dec")


def test_native_harness_fails_loudly_without_a_gpu(tmp_path):
    """tools/test_amd_dec (the test_nv_dec counterpart) links the drop-in jm_nvdec_* symbols; with no HIP device it must
    refuse (exit 3) instead of decoding on the CPU."""
    import jmcodec_amd
    if jmcodec_amd.jm_nvdec_is_hw_support():
        pytest.skip("a GPU is present")
    from tools import streams
    p = tmp_path / "s.h264"
    p.write_bytes(streams.generate(width=96, height=80, frames=4, gop=4))
    r = subprocess.run([_harness(), str(p)], capture_output=True, text=True)
    assert r.returncode == 3 and "no HIP device" in r.stderr


def test_native_harness_loop_in_parse_only_mode(tmp_path):
    """The harness's sliding-window NAL loop (small window, multi-loop) drives the library's host stages correctly: frame count and
    the reference's info block format (nv_dec.cpp:666-680)."""
    from tools import streams
    p = tmp_path / "s.h264"
    p.write_bytes(streams.generate(width=96, height=80, frames=5, gop=5, num_ref=2, mode=1))
    env = dict(os.environ, JM_AMD_DEC_PARSE_ONLY="1")
    for extra, frames in ((["--chunk", "64"], 5), (["--loops", "3"], 15), ([], 5)):
        r = subprocess.run([_harness(), str(p), "--no-hw-check"] + extra, capture_output=True, text=True, env=env)
        assert r.returncode == 0, r.stderr
        assert f"Frame Count:\t{frames}\n" in r.stdout and "Display:\t96 x 80\n" in r.stdout and "Pixel Format:\tYV12\n" in r.stdout
        assert f"frames fetched = {frames} " in r.stdout
