#!/usr/bin/env python3
"""Writes tests/golden/table_digests.json: SHA-256 of every constant table of the product headers with the clause of the standard it
restates (tests/test_table_provenance.py::table_digests).  Re-run only after reviewing a table change against that clause."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
from test_table_provenance import table_digests  # noqa: E402

if __name__ == "__main__":
    path = os.path.join(ROOT, "tests", "golden", "table_digests.json")
    json.dump(table_digests(), open(path, "w"), indent=1, sort_keys=True)
    print("wrote", path)
