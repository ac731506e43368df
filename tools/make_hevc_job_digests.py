#!/usr/bin/env python3
"""Developer tool: FNV-1a digests of the HEVC job lists (option "job_digest": every array of a picture as the device gets it) for the test streams of
tests/test_hevc_oracle.py -> tests/golden/hevc_job_digests.json.  Run it on a tree whose HEVC output the GPU parity tests have just confirmed
(python -m pytest tests -m gpu -k hevc): tests/test_hevc_host_parser.py::test_job_lists_are_the_ones_the_gpu_tests_confirmed then holds later changes
of the host parser (which are made, and timed, without a GPU) to exactly those lists.
    python tools/make_hevc_job_digests.py [--check]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import jmcodec_amd                              # noqa: E402
from tools import streams                       # noqa: E402
from test_hevc_oracle import HEVC_CASES         # noqa: E402

PATH = os.path.join(ROOT, "tests", "golden", "hevc_job_digests.json")


def digest(data):
    with jmcodec_amd.JmAmdDec(1, 1, options={"parse_only": 1, "job_digest": 1}) as d:
        n = d.decode_stream(data, keep=False)
        return "%016x" % (d.stat("job_digest") & (2 ** 64 - 1)), n, d.stat("errors")


def main():
    out = {name: digest(streams.generate_hevc(**HEVC_CASES[name]))[0] for name in sorted(HEVC_CASES)}
    if "--check" in sys.argv:
        want = json.load(open(PATH))
        bad = [k for k in out if want.get(k) != out[k]]
        print("differ:", bad)
        sys.exit(1 if bad else 0)
    with open(PATH, "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print(len(out), "digests ->", PATH)


if __name__ == "__main__":
    main()
