#!/usr/bin/env python3
"""run under `rocprofv3 --pmc FETCH_SIZE --kernel-include-regex k_stream`: streams 2 GiB with
4-byte and with 16-byte loads per lane, 3 launches each, so FETCH_SIZE per launch can be
compared with the known byte count (MI355X_MICROARCH.md: gfx950 reports 1/2 for wide reads)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry

pkg = entry.load_package()
ctx = pkg.capi.Context(0)
N = 2 << 30
ctx.selftest_stream(N, 0, 3)
ctx.selftest_stream(N, 1, 3)
print("streamed", N, "bytes x3 (4 B/lane) and x3 (16 B/lane)")
